"""Build libpenguin_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m penguin.jl_amd.build [--force]

Objects go to penguin/jl_amd/_build/, the library to penguin/jl_amd/lib/ (both git-ignored; the .so
travels to the GPU box with the repo snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "_build"
LIBDIR = HERE / "lib"
LIB = LIBDIR / "libpenguin_hip.so"
ARCH = "gfx950"

# per-file extra flags: the geometry must classify cells exactly like the oracle => no FMA contraction
EXTRA = {"pg_capacity.hip": ["-ffp-contract=off"]}


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the sources the library is built from: ties a measurement (bench line, rocprofv3
    trace, PMC pass under profiles/) to the code that produced it -- the GPU box has no git history to ask."""
    import hashlib

    h = hashlib.sha256()
    for p in sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + [HERE.parent.parent / "include" / "penguin_hip.h"]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build penguin.jl_amd)")


def _newer(src: Path, dst: Path, deps) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, *deps])


def build_library(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + [HERE.parent.parent / "include" / "penguin_hip.h"]
    sources = sorted(CSRC.glob("*.hip"))
    jobs = []
    for src in sources:
        obj = OBJ / (src.stem + ".o")
        if force or _newer(src, obj, headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *EXTRA.get(src.name, []),
                   *os.environ.get("PG_HIPCC_FLAGS", "").split(), "-c", str(src),
                   "-o", str(obj)]
            jobs.append((src.name, cmd))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{r.stderr[-4000:]}")
        return name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for name in ex.map(run, jobs):
                if verbose:
                    print(f"[build] compiled {name}", flush=True)
    objs = [str(OBJ / (s.stem + ".o")) for s in sources]
    if jobs or force or not LIB.exists():
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *objs, "-ldl"]   # RCCL is dlopen'ed on first use (csrc/pg_rccl.h)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
