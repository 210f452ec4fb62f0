"""Developer probe: build the 3-D heat problem at size n and time steps (not the judged bench)."""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pj.init(0)
t0 = time.time()
mesh = pj.Mesh((n, n, n), (4., 4., 4.), (0., 0., 0.))
body = pj.Sphere((2.01, 2.01, 2.01), 1.0)
cap = pj.Capacity(body, mesh)
print(f"capacity: {cap.kernel_ms:.2f} ms kernels, wall {time.time()-t0:.2f}s", flush=True)
op = pj.DiffusionOps(cap)
M = (n + 1) ** 3
bc1 = pj.Dirichlet(1.0)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
ph = pj.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
u0 = np.zeros(2 * M)
dt = 0.75 * (4.0 / n) ** 2
t0 = time.time()
s = pj.DiffusionUnsteadyMono(ph, bcb, bc1, dt, u0, "BE")
print(f"ctor (numbering+assembly): wall {time.time()-t0:.2f}s", flush=True)
info = s.system_info(0)
print("n_own", info.n_own, "nnz", info.nnz, "n_omega", info.n_omega, "n_gamma", info.n_gamma, flush=True)
import os
opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, int(os.environ.get('PG_WARM_START', '1')))
si = L.pg_step_info()
L.check(L.lib().pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
print("initial solve iters", si.iters, "conv", si.converged, "res", si.resnorm, "bnorm", si.bnorm, "ext", si.extremum, flush=True)
L.check(L.lib().pg_set_profiling(1))
run = L.pg_run_info()
for rep in range(2):
    t0 = time.time()
    L.check(L.lib().pg_solver_run(s._h, C.c_double(1e9), C.c_int32(1), C.byref(opts), 0, C.c_int64(steps), 0, C.byref(run)))
    wall = time.time() - t0
    info = s.system_info(1)
    bspmv = 12 * info.nnz + 20 * info.n_own
    avg = run.spmv_ms_total / max(run.spmv_launches, 1)
    print(f"run{rep}: steps {run.steps} iters {run.total_iters} wall {wall*1e3:.1f} ms solve_ms {run.solve_ms:.1f} -> {run.steps/wall:.2f} steps/s; "
          f"spmv avg {avg:.4f} ms x{run.spmv_launches} = {bspmv/avg/1e6:.1f} GB/s ({bspmv/avg/1e6/8000*100:.1f}% of 8TB/s); spmv share {run.spmv_ms_total/run.solve_ms*100:.1f}%", flush=True)
ms = C.c_double()
L.check(L.lib().pg_solver_time_spmv(s._h, 1, 50, C.byref(ms)))
print(f"plain spmv back-to-back: {ms.value:.4f} ms = {bspmv/ms.value/1e6:.1f} GB/s")
