"""How compressible is the preconditioned run matrix?  Fraction of rows whose stencil (col-row offsets and values)
repeats the previous row's exactly / within 1e-14, and the run-length distribution (dev diagnostic)."""
import sys, ctypes as C
import numpy as np
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L

pj.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L0 = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
mesh = pj.Mesh((n, n, n), (L0, L0, L0), (0.0, 0.0, 0.0))
cap = pj.Capacity(pj.Sphere((L0 / 2 + 0.01,) * 3, L0 / 4), mesh)
op = pj.DiffusionOps(cap)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
dt = 0.75 * (L0 / n) ** 2
s = pj.DiffusionUnsteadyMono(pj.Phase(cap, op, 0.0, 1.0), bcb, pj.Dirichlet(1.0), dt, None, "CN")
A, b, idx = s.system(2)
A = A.tocsr()
nr = A.shape[0]
rp, col, val = A.indptr, A.indices, A.data
cnt = np.diff(rp)
print("n", nr, "nnz", A.nnz, "count histogram", np.bincount(cnt))
same_cnt = np.zeros(nr, bool)
same_cnt[1:] = cnt[1:] == cnt[:-1]
exact = same_cnt.copy()
close = same_cnt.copy()
for c in np.unique(cnt):
    rows = np.flatnonzero((cnt == c) & same_cnt)
    if c == 0 or rows.size == 0:
        continue
    off = col[rp[rows][:, None] + np.arange(c)] - rows[:, None]
    offp = col[rp[rows - 1][:, None] + np.arange(c)] - (rows - 1)[:, None]
    v = val[rp[rows][:, None] + np.arange(c)]
    vp = val[rp[rows - 1][:, None] + np.arange(c)]
    so = np.all(off == offp, axis=1)
    exact[rows] = so & np.all(v == vp, axis=1)
    close[rows] = so & np.all(np.abs(v - vp) <= 1e-14 * np.abs(vp), axis=1)
print("rows repeating the previous row's stencil: exact %.4f  within 1e-14 %.4f" % (exact.mean(), close.mean()))
for name, m in (("exact", exact), ("close", close)):
    starts = np.flatnonzero(~m)
    lens = np.diff(np.append(starts, nr))
    # entries left in CSR if only runs >= 16 rows are compressed
    for thr in (8, 16, 32):
        rows_c = lens[lens >= thr].sum()
        nnz_c = sum(cnt[s0] * l for s0, l in zip(starts[lens >= thr], lens[lens >= thr]))
        print(f"  {name}: runs>={thr}: {np.count_nonzero(lens >= thr)} runs cover {rows_c / nr:.4f} of rows, {nnz_c / A.nnz:.4f} of nnz; mean len {lens[lens >= thr].mean():.1f}")
