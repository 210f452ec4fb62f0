"""Developer probe: rows with a vanishing diagonal in a moving slab's raw system (slivers of space-time cut cells)."""
import sys, numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
pj.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lx = 16.0
mesh = pj.Mesh((n, n), (lx, lx), (-8.0, -8.0))
c = 1.56
body = pj.MovingSphere(lambda t: (0.0, 0.0), lambda t: 1.0 + c * np.sqrt(t + 0.01), complement=True,
                       dcenter=lambda t: (0.0, 0.0), dradius=lambda t: 0.5 * c / np.sqrt(t + 0.01))
dt = 1.0 * (lx / n) ** 2
t = k * dt
cap = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t, t + dt]))
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
M = (n + 1) ** 2
bcb = pj.BorderConditions({kk: pj.Dirichlet(0.0) for kk in ("left", "right", "top", "bottom")})
s = pj.MovingDiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, mesh, "CN")
A, b, idx = s.system(0)
A = A.tocsr()
n_rows = A.shape[0]
rows = np.repeat(np.arange(n_rows), np.diff(A.indptr))
isd = A.indices == rows
d = np.zeros(n_rows)
d[rows[isd]] = A.data[isd]
absmax = np.zeros(n_rows)
np.maximum.at(absmax, rows, np.abs(A.data))
bad = np.flatnonzero(np.abs(d) <= 1e-14 * absmax)
print("rows", n_rows, "nnz", A.nnz, "rows with |a_ii| <= 1e-14 max|a_ij|:", len(bad), "non-finite entries:", int((~np.isfinite(A.data)).sum()))
V, G, Vn_1, Vn = cap.V, cap.Γ, cap.Vn_1, cap.Vn
for r in bad[:6]:
    j = int(idx[r])
    cell = j % M
    a, e = A.indptr[r], A.indptr[r + 1]
    print("row", r, "unknown", "w" if j < M else "g", "cell", cell, (cell % (n + 1), cell // (n + 1)), "diag", d[r],
          "entries", {(("w" if idx[c] < M else "g"), int(idx[c] % M - cell)): float(v) for c, v in zip(A.indices[a:e], A.data[a:e])})
    print("   V", V[cell], "G", G[cell], "Vn_1", Vn_1[cell], "Vn", Vn[cell], "A", [float(x[cell]) for x in cap.A], "B", [float(x[cell]) for x in cap.B],
          "W", [float(x[cell]) for x in cap.W], "ct", cap.cell_types[cell])
