"""Step rates of the small configurations (SURVEY 8d configs 1-3 shapes): how launch-bound is the loop there?"""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
pj.init(0)
lib = L.lib()
for N, n, steps in ((2, 80, 200), (2, 512, 200), (2, 2048, 100), (3, 64, 200), (3, 128, 100), (3, 256, 50)):
    mesh = pj.Mesh((n,) * N, (4.0,) * N, (0.0,) * N)
    cap = pj.Capacity(pj.Sphere((2.01,) * N, 1.0), mesh)
    op = pj.DiffusionOps(cap)
    M = (n + 1) ** N
    bcb = pj.BorderConditions({k: pj.Dirichlet(1.0 if N == 3 else 0.0) for k in ("left", "right", "top", "bottom")})
    ph = pj.Phase(cap, op, 0.0, 1.0)
    dt = (0.75 if N == 3 else 0.25) * (4.0 / n) ** 2
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
    opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, 1)
    si = L.pg_step_info()
    L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
    run = L.pg_run_info()
    sch = 1 if N == 3 else 0
    L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(sch), C.byref(opts), 0, C.c_int64(5), 0, C.byref(run)))
    L.check(lib.pg_device_synchronize())
    t0 = time.perf_counter()
    L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(sch), C.byref(opts), 0, C.c_int64(steps), 0, C.byref(run)))
    L.check(lib.pg_device_synchronize())
    el = time.perf_counter() - t0
    info = s.system_info(1)
    it = run.total_iters / run.steps
    print(f"{N}D n={n:5d} rows={info.n_own:9d} steps/s={steps/el:9.1f} ms/step={el/steps*1e3:7.3f} iters/step={it:5.2f} us/iter={el/steps*1e6/max(it,1e-9):7.1f}", flush=True)
