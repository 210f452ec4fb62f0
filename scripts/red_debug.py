import sys, numpy as np
sys.path.insert(0, "/root/repo")
import penguin.jl_amd as pj
pj.init(0)
n, L, N = 20, 4.0, 3
mesh = pj.Mesh((n,)*N, (L,)*N, (0.0,)*N)
cap = pj.Capacity(pj.Sphere((2.03, 1.98, 2.01), 1.1), mesh)
M = (n+1)**N
dt = 0.4*(L/n)**2
ph = pj.Phase(cap, pj.DiffusionOps(cap), lambda x,y,z,t: 0.0, 1.0)
bc = pj.Dirichlet(1.0); bcb = pj.BorderConditions({})
s = pj.DiffusionUnsteadyMono(ph, bcb, bc, dt, np.zeros(2*M), "BE")
pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 3*dt, bcb, bc, "CN", method="bicgstab")
A, b, idx = s.system(3)
info = s.system_info(3)
nw = info.n_omega
A = A.tocsr()
G = A[nw:, :]
print("gamma rows", G.shape, "nnz", G.nnz, "diag==1:", np.sum(G.diagonal(nw) == 1.0), "max|offdiag|", abs(G - __import__('scipy.sparse').sparse.eye(G.shape[0], G.shape[1], k=nw)).max())
cnt = np.diff(G.indptr); print("entries per gamma row: ", np.bincount(cnt))
d = G.diagonal(nw); print("diag range", d.min(), d.max())
