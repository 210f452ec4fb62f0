import sys, numpy as np, ctypes as C
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
from oracle import penguin_oracle as po
from tests.common import oracle_capacity_from_product, rel_l2
import scipy.sparse.linalg as spla, scipy.sparse as sp
pj.init(0)
n, Lx, c, r = 24, 4.0, (2.0, 2.0), 1.0
M = (n + 1) ** 2
mesh = pj.Mesh((n, n), (Lx, Lx))
cap1, cap2 = pj.Capacity(pj.Sphere(c, r), mesh), pj.Capacity(pj.Sphere(c, r, complement=True), mesh)
f = lambda x, y, z, t: 1.0
D = lambda x, y, z: 1.0
p1, p2 = pj.Phase(cap1, pj.DiffusionOps(cap1), f, D), pj.Phase(cap2, pj.DiffusionOps(cap2), f, D)
ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 1.0, 0.0), pj.FluxJump(1.0, 1.0, 0.0))
bcb = pj.BorderConditions({k: pj.Dirichlet(0.0) for k in ("left","right","top","bottom")})
dt = 0.5 * (Lx / n) ** 2
s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, np.zeros(4*M), "BE")
A,b,idx = s.system(0); A=A[:,:len(idx)]
x_lu = spla.spsolve(A.tocsc(), b)
opts = L.pg_krylov_opts(0, 1e-13, 0.0, 0, 4); info = L.pg_step_info()
L.check(L.lib().pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
print("gpu iters", info.iters, "conv", info.converged, "res", info.resnorm, "bnorm", info.bnorm)
x = s._fetch_state()[idx]
print("gpu vs LU", rel_l2(x, x_lu), "true resid", np.linalg.norm(A@x-b)/np.linalg.norm(b))
d = A.diagonal(); print("diag zeros", np.sum(d==0), "neg", np.sum(d<0), "min abs", np.abs(d).min())
Ds = sp.diags(1/np.sqrt(np.abs(np.where(d==0,1,d)))); As=(Ds@A@Ds).tocsr()
y,it,rn = po.bicgstab_ref(As, Ds@b, reltol=1e-13, maxiter=5000); print("oracle equilibrated bicgstab", it, rn, rel_l2(Ds@y, x_lu))
y,it,rn = po.bicgstab_ref(A.tocsr(), b, reltol=1e-13, maxiter=5000); print("oracle plain bicgstab", it, rn, rel_l2(y, x_lu))
print("cond est: |x_lu|", np.abs(x_lu).max())
