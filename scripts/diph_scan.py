"""Diphasic config-5 family: iterations / convergence per step at growing sizes (dev diagnostic)."""
import sys, ctypes as C
import numpy as np
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L

pj.init(0)
lib = L.lib()
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024]:
    Lx, c, r = 8.0, (4.0, 4.0), 2.0
    M = (n + 1) ** 2
    mesh = pj.Mesh((n, n), (Lx, Lx))
    cap1, cap2 = pj.Capacity(pj.Sphere(c, r), mesh), pj.Capacity(pj.Sphere(c, r, complement=True), mesh)
    p1, p2 = pj.Phase(cap1, pj.DiffusionOps(cap1), 0.0, 1.0), pj.Phase(cap2, pj.DiffusionOps(cap2), 0.0, 1.0)
    ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 1.0, 0.0), pj.FluxJump(1.0, 1.0, 0.0))
    bcb = pj.BorderConditions({})
    dt = 0.5 * (Lx / n) ** 2
    u0 = np.concatenate([np.ones(M), np.ones(M), np.zeros(M), np.zeros(M)])
    s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, u0, "BE")
    opts = L.pg_krylov_opts(L.PG_METHOD["bicgstab"], 1e-12, 0.0, 0, 4, 1)
    info = L.pg_step_info()
    L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
    print(n, "initial", info.iters, info.converged, info.resnorm / max(info.bnorm, 1e-300), info.extremum, flush=True)
    for k in range(4):
        L.check(lib.pg_solver_step(s._h, L.PG_SCHEME["CN"], C.byref(opts), C.byref(info)))
        print(n, "step", k, info.iters, info.converged, info.resnorm / max(info.bnorm, 1e-300), info.extremum, flush=True)
    import time
    import scipy.sparse.linalg as spl
    A, b, idx = s.system(1)
    nr = A.shape[0]
    t0 = time.time()
    x = spl.spsolve(A[:, :nr].tocsc(), b)
    xg = pj_state = None
    L.check(lib.pg_solver_get_state(s._h, C.c_int64(-1), L.dptr(full := np.zeros(4 * M)), C.c_int64(4 * M)))
    xr = full[idx]
    print("   LU %.1fs  rel-L2(gpu, LU) = %.3e   raw residual %.3e   heat %.12f vs %.12f" % (
        time.time() - t0, np.linalg.norm(xr - x) / np.linalg.norm(x), np.linalg.norm(A @ xr - b) / np.linalg.norm(b),
        cap1.V @ full[:M] + cap2.V @ full[2 * M:3 * M], cap1.V.sum()), flush=True)
