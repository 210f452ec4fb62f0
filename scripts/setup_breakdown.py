"""Wall-clock breakdown of the set-up of the 512^3 bench problem (dev diagnostic)."""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pj.init(0)
lib = L.lib()
def tick(label, t0):
    L.check(lib.pg_device_synchronize())
    print(f"{label:34s} {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
for rep in range(2):
    print("--- pass", rep)
    t = time.perf_counter(); mesh = pj.Mesh((n, n, n), (4.0, 4.0, 4.0)); tick("Mesh", t)
    t = time.perf_counter(); cap = pj.Capacity(pj.Sphere((2.01, 2.01, 2.01), 1.0), mesh); tick(f"Capacity (kernels {cap.kernel_ms:.1f} ms)", t)
    t = time.perf_counter(); op = pj.DiffusionOps(cap); tick("DiffusionOps", t)
    bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
    t = time.perf_counter(); ph = pj.Phase(cap, op, 0.0, 1.0)
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), 0.75 * (4.0 / n) ** 2, None, "BE"); tick("DiffusionUnsteadyMono ctor", t)
    opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, 1); info = L.pg_step_info()
    t = time.perf_counter(); L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info))); tick(f"initial solve ({info.iters} its)", t)
    t = time.perf_counter(); L.check(lib.pg_solver_step(s._h, 1, C.byref(opts), C.byref(info))); tick(f"first CN step incl. run matrix ({info.iters} its)", t)
    t = time.perf_counter(); L.check(lib.pg_solver_step(s._h, 1, C.byref(opts), C.byref(info))); tick(f"second CN step ({info.iters} its)", t)
    del s, ph, op, cap, mesh
