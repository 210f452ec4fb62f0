"""PCIe-inclusive rate at 512^3: one CN step + fetching the full padded state (2M doubles = 2.16 GB) to the host, the way the
reference's `push!(s.states, s.x)` keeps every state."""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
pj.init(0)
lib = L.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mesh = pj.Mesh((n, n, n), (4.0, 4.0, 4.0), (0.0, 0.0, 0.0))
cap = pj.Capacity(pj.Sphere((2.01, 2.01, 2.01), 1.0), mesh)
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
dt = 0.75 * (4.0 / n) ** 2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, 1)
si = L.pg_step_info()
L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
M = (n + 1) ** 3
x = np.zeros(2 * M)
for k in range(3):
    L.check(lib.pg_solver_step(s._h, 1, C.byref(opts), C.byref(si)))
t_step = t_fetch = 0.0
K = 5
for k in range(K):
    t0 = time.perf_counter()
    L.check(lib.pg_solver_step(s._h, 1, C.byref(opts), C.byref(si)))
    t1 = time.perf_counter()
    L.check(lib.pg_solver_get_state(s._h, C.c_int64(-1), L.dptr(x), C.c_int64(2 * M)))
    t2 = time.perf_counter()
    t_step += t1 - t0
    t_fetch += t2 - t1
print(f"n={n}: step {t_step / K * 1e3:.2f} ms, state fetch (2M = {2 * M * 8 / 1e9:.2f} GB) {t_fetch / K * 1e3:.1f} ms "
      f"=> {2 * M * 8 / 1e9 / (t_fetch / K):.1f} GB/s, PCIe-inclusive {K / (t_step + t_fetch):.1f} steps/s")
