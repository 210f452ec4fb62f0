"""Is the Neumann preconditioner admitted on the diphasic config-5 family, and what do the iteration counts look like?"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
pj.init(0)
lib = L.lib()
for n in (128, 512):
    Lx, c, r = 8.0, (4.0, 4.0), 2.0
    M = (n + 1) ** 2
    mesh = pj.Mesh((n, n), (Lx, Lx))
    cap1, cap2 = pj.Capacity(pj.Sphere(c, r), mesh), pj.Capacity(pj.Sphere(c, r, complement=True), mesh)
    p1, p2 = pj.Phase(cap1, pj.DiffusionOps(cap1), 0.0, 1.0), pj.Phase(cap2, pj.DiffusionOps(cap2), 0.0, 1.0)
    ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 1.0, 0.0), pj.FluxJump(1.0, 1.0, 0.0))
    dt = 0.5 * (Lx / n) ** 2
    u0 = np.concatenate([np.ones(M), np.ones(M), np.zeros(M), np.zeros(M)])
    s = pj.DiffusionUnsteadyDiph(p1, p2, pj.BorderConditions({}), ic, dt, u0, "BE")
    info = s.system_info(2)
    opts = L.pg_krylov_opts(L.PG_METHOD["bicgstab"], 1e-12, 0.0, 0, 4, 1)
    si = L.pg_step_info()
    L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
    its = [si.iters]
    for k in range(4):
        L.check(lib.pg_solver_step(s._h, L.PG_SCHEME["CN"], C.byref(opts), C.byref(si)))
        its.append(si.iters)
    print(n, "neumann_ok", info.neumann_ok, "gershgorin %.3f" % info.gershgorin, "run-matrix gersh %.3f" % s.system_info(3).gershgorin, "iters", its, flush=True)
