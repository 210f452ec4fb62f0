"""ctypes binding of libpenguin_hip.so -- the same symbols julia/PenguinHIP.jl ccall's.

There is NO CPU fallback: if the shared library is missing, or no HIP device is visible
when a device function is called, the call raises.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["PG_LIB_PATH"]) if os.environ.get("PG_LIB_PATH") else _HERE / "lib" / "libpenguin_hip.so"   # override: A/B runs of two builds
HEADER_PATH = _HERE.parent.parent / "include" / "penguin_hip.h"


class PenguinHipError(RuntimeError):
    pass


# ---- enums (mirror include/penguin_hip.h) ---------------------------------------------------
PG_BODY_BALL, PG_BODY_MULTIBALL, PG_BODY_HALFSPACE, PG_BODY_ELLIPSOID = 1, 2, 3, 4
PG_FLAG_COMPLEMENT, PG_FLAG_NO_CENTROIDS = 1, 2
PG_CAP_V, PG_CAP_GAMMA, PG_CAP_CELL_TYPES, PG_CAP_A, PG_CAP_B, PG_CAP_W, PG_CAP_C_OMEGA, PG_CAP_C_GAMMA = range(8)
PG_CAP_ST_V0, PG_CAP_ST_V1, PG_CAP_ST_CT_OMEGA, PG_CAP_ST_CT_GAMMA = 8, 9, 10, 11   # space-time capacities only
PG_OP_G, PG_OP_H, PG_OP_WINV = 0, 1, 2
PG_OP_C0, PG_OP_K0 = 3, 6            # ConvectionOps: C_d = PG_OP_C0 + d, K_d = PG_OP_K0 + d
PG_BC_NONE, PG_BC_DIRICHLET, PG_BC_NEUMANN, PG_BC_ROBIN, PG_BC_PERIODIC = 0, 1, 2, 3, 4
PG_KEY = {"left": 0, "right": 1, "bottom": 2, "top": 3, "backward": 4, "forward": 5}
PG_SCHEME = {"BE": 0, "CN": 1, "STEADY": 2}
PG_METHOD = {"bicgstab": 0, "cg": 1, "gmres": 2}

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)


class pg_bc_desc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("alpha", C.c_double), ("beta", C.c_double), ("value", C.c_double),
                ("value_array", c_double_p)]


class pg_border_desc(C.Structure):
    _fields_ = [("key", C.c_int32), ("kind", C.c_int32), ("value", C.c_double)]


class pg_jump_desc(C.Structure):
    _fields_ = [("alpha1", C.c_double), ("alpha2", C.c_double), ("g", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("h", C.c_double), ("g_array", c_double_p), ("h_array", c_double_p)]


class pg_motion_desc(C.Structure):
    _fields_ = [("body_kind", C.c_int32), ("flags", C.c_int32), ("axis", C.c_int32), ("nq", C.c_int32), ("sign", C.c_double),
                ("t0", C.c_double), ("t1", C.c_double), ("nodes", c_double_p), ("body0", C.c_double * 4),
                ("body1", C.c_double * 4)]


class pg_krylov_opts(C.Structure):
    _fields_ = [("method", C.c_int32), ("reltol", C.c_double), ("abstol", C.c_double), ("maxiter", C.c_int32),
                ("check_every", C.c_int32), ("warm_start", C.c_int32), ("restart", C.c_int32), ("precond", C.c_int32)]


class pg_step_info(C.Structure):
    _fields_ = [("iters", C.c_int32), ("converged", C.c_int32), ("resnorm", C.c_double), ("bnorm", C.c_double),
                ("extremum", C.c_double), ("time", C.c_double)]


class pg_run_info(C.Structure):
    _fields_ = [("steps", C.c_int64), ("total_iters", C.c_int64), ("t_final", C.c_double), ("extremum", C.c_double),
                ("solve_ms", C.c_double), ("spmv_ms_total", C.c_double), ("spmv_launches", C.c_int64),
                ("unconverged_steps", C.c_int64), ("worst_relres", C.c_double), ("spmv_lean_ms_total", C.c_double),
                ("spmv_lean_launches", C.c_int64), ("poly_degree", C.c_int64), ("half_exits", C.c_int64),
                ("poly_xspace", C.c_int64), ("products", C.c_int64), ("guess_states_read", C.c_int64)]


class pg_system_info(C.Structure):
    _fields_ = [("n_own", C.c_int64), ("nnz", C.c_int64), ("n_ghost", C.c_int64), ("n_omega", C.c_int64),
                ("n_gamma", C.c_int64), ("M_global", C.c_int64), ("spmv_bytes", C.c_int64), ("spmv_slices", C.c_int64),
                ("rows_uniform", C.c_int64), ("rows_pattern", C.c_int64), ("rows_irregular", C.c_int64),
                ("neumann_ok", C.c_int64), ("gershgorin", C.c_double), ("spmv_units", C.c_int64),
                ("rows_marched", C.c_int64), ("rows_matrix", C.c_int64), ("n_ghost_loop", C.c_int64),
                ("loop_is_compact", C.c_int64)]


def declared_symbols() -> list[str]:
    """Every `pg_*` function declared in include/penguin_hip.h."""
    txt = HEADER_PATH.read_text()
    return sorted(set(re.findall(r"\bint32_t\s+(pg_[a-z0-9_]+)\s*\(", txt)))


_lib = None
hip_runtime_source = "system"      # which libamdhip64 the library is bound to ("system", "torch wheel", "already mapped by torch")


def _share_the_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  A PyTorch-ROCm wheel ships its own libamdhip64 / libhsa-runtime64 (SONAMEs
    libamdhip64.so.7, ... -- the system ROCm's SONAMEs) and its libraries ask for them by FILE name (`libamdhip64.so`,
    RPATH $ORIGIN).  Loaded first, torch's copies satisfy libpenguin_hip.so's NEEDED entries by SONAME: one runtime.  Loaded
    second, the file-name lookup does not match the system copy we brought in and a second HIP + HSA runtime is mapped into
    the process.  So when torch is installed but not imported yet, its runtime files are loaded here, before ours resolves
    its dependencies: whichever side comes first, both end up on the same copies (scripts/which_hip_runtime.py lists them).
    PG_HIP_RUNTIME=system keeps the system ROCm.
    (The abort at interpreter exit that round 2 met when torch was imported after the library had another cause -- librccl
    mapped before torch -- and is gone because RCCL is loaded on first use now: csrc/pg_rccl.h.)"""
    global hip_runtime_source
    import sys

    if "torch" in sys.modules:
        hip_runtime_source = "already mapped by torch"
        return
    if os.environ.get("PG_HIP_RUNTIME", "auto") == "system":
        return
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = Path(list(spec.submodule_search_locations)[0]) / "lib"
    if not (libdir / "libamdhip64.so").exists():
        return                                            # a CPU / CUDA build of torch: nothing to share
    mode = C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        if (libdir / name).exists():
            C.CDLL(str(libdir / name), mode=mode)
    hip_runtime_source = "torch wheel"


def lib() -> C.CDLL:
    """Load the shared library (no GPU needed for loading)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise PenguinHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  penguin.jl_amd has no CPU fallback.")
        _share_the_hip_runtime_with_torch()
        _lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
        for name in declared_symbols():
            getattr(_lib, name).restype = C.c_int32
    return _lib


def check(status: int) -> None:
    if status != 0:
        buf = C.create_string_buffer(4096)
        lib().pg_last_error(buf, 4096)
        raise PenguinHipError(buf.value.decode("utf-8", "replace"))


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def iptr(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_i64_p)


_initialised = False


def init(device: int | None = None) -> None:
    """pg_init on LOCAL_RANK (or `device`); idempotent."""
    global _initialised
    if _initialised:
        return
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(lib().pg_init(C.c_int32(device)))
    _initialised = True


def init_distributed(device: int, rank: int, nranks: int, unique_id: bytes | None) -> None:
    global _initialised
    buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
    check(lib().pg_init_distributed(C.c_int32(device), C.c_int32(rank), C.c_int32(nranks), buf))
    _initialised = True


def get_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    check(lib().pg_get_unique_id(buf))
    return buf.raw


def finalize() -> None:
    global _initialised
    if _initialised:
        check(lib().pg_finalize())
        _initialised = False


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(lib().pg_device_name(buf, 256))
    return buf.value.decode()


def config_string() -> str:
    """Every PG_* tuning / variant selector this process runs with (read once by the library), as `name=value ...`."""
    buf = C.create_string_buffer(2048)
    check(lib().pg_config_string(buf, 2048))
    return buf.value.decode() + f" hip_runtime={hip_runtime_source}"
