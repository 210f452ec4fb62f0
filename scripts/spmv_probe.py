"""Back-to-back timing of the SpMV launch modes on the run matrix of the n^3 heat problem (developer probe).

    python scripts/spmv_probe.py [n=512] [reps=50]
Modes: 0 plain, 1 (+ r-hat dot), 3 (three dots), 4 (y = 2x - Ax).  Kernel variants via the PG_SPMV_* environment.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
modes = [int(c) for c in sys.argv[3]] if len(sys.argv) > 3 else [0, 1, 3, 4]
modes = [m for m in modes if m <= 4]
pj.init(0)
mesh = pj.Mesh((n, n, n), (4.0, 4.0, 4.0))
cap = pj.Capacity(pj.Sphere((2.01, 2.01, 2.01), 1.0), mesh)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
dt = 0.75 * (4.0 / n) ** 2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, 1)
si = L.pg_step_info()
lib = L.lib()
# diagnostic kernel switches (PG_SPMV_XCD bits 8..) give wrong products: no solve may run with them (it would never
# converge); the constructor's matrix is timed instead of the run matrix
broken = int(os.environ.get("PG_SPMV_XCD", "1")) >= 256
sel = 0 if broken else 1
run = L.pg_run_info()
if not broken:
    L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
    L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(1), C.byref(opts), 0, C.c_int64(2), 0, C.byref(run)))
info = s.system_info(2 + sel)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PG_"))
out = [f"[{tag}] units {info.spmv_units} rows_m {info.rows_marched} slices {info.spmv_slices} bytes {info.spmv_bytes / 1e6:.1f} MB |"]
ms = C.c_double()
print(out[0], file=sys.stderr, flush=True)
for mode in modes:
    print(f"mode {mode} ...", file=sys.stderr, flush=True)
    extra = {0: 0, 1: 8, 3: 16, 4: 0}[mode] * info.n_own
    L.check(lib.pg_solver_time_spmv(s._h, sel | (mode << 4), reps, C.byref(ms)))
    warm = ms.value
    L.check(lib.pg_solver_time_spmv(s._h, sel | (mode << 4) | 256, reps, C.byref(ms)))
    out.append(f"m{mode} warm {warm * 1e3:5.1f} cold {ms.value * 1e3:5.1f} us ({(info.spmv_bytes + extra) / ms.value / 1e6:5.0f} GB/s)")
    if mode == 4:      # the loop's own matrix (Dirichlet interface rows left out), launches chained as in a polynomial chain
        L.check(lib.pg_solver_time_spmv(s._h, sel | (mode << 4) | 512 | 1024, reps, C.byref(ms)))
        li = s.system_info(6 + sel)
        out.append(f"loop-matrix chained {ms.value * 1e3:5.1f} us ({li.spmv_bytes / 1e6:.1f} MB)")
if not broken:
    L.check(lib.pg_set_profiling(1))
    L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(1), C.byref(opts), 0, C.c_int64(10), 0, C.byref(run)))
    out.append(f"| loop: {run.steps / run.solve_ms * 1e3:6.1f} steps/s, {run.total_iters / run.steps:.2f} it/step, m={run.poly_degree}, spmv dots "
               f"{run.spmv_ms_total / max(run.spmv_launches, 1) * 1e3:6.1f} us lean {run.spmv_lean_ms_total / max(run.spmv_lean_launches, 1) * 1e3:6.1f} us")
print(" ".join(out), flush=True)
