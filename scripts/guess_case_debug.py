"""Diagnostic: degrees and extrapolated-start decisions of one small configuration (python scripts/guess_case_debug.py N n steps)."""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
N, n, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pj.init(0)
lib = L.lib()
mesh = pj.Mesh((n,) * N, (4.0,) * N, (0.0,) * N)
cap = pj.Capacity(pj.Sphere((2.01,) * N, 1.0), mesh)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0 if N == 3 else 0.0) for k in ("left", "right", "top", "bottom")})
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
dt = (0.75 if N == 3 else 0.25) * (4.0 / n) ** 2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, 1)
si = L.pg_step_info()
L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
run = L.pg_run_info()
sch = 1 if N == 3 else 0
L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(sch), C.byref(opts), 0, C.c_int64(steps), 0, C.byref(run)))
print("products per step", run.products / run.steps, "states read per step", run.guess_states_read / run.steps, "half exits", run.half_exits, "iters", run.total_iters)
