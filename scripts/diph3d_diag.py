"""Diagnostic: 3-D diphasic 16^3 case of tests/test_gpu_vtk.py -- per-state differences, residuals of both solutions in the
oracle's reduced system, sensitivity of the oracle's own direct solve to a one-ulp perturbation of the matrix."""
import sys
import numpy as np
import scipy.sparse.linalg as spl
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from oracle import penguin_oracle as po
from tests.common import oracle_capacity_from_product, rel_l2
pj.init(0)
n, Lx, c, r = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 4.0, (2.03, 1.98, 2.01), 1.1
M = (n + 1) ** 3
mesh, omesh = pj.Mesh((n,) * 3, (Lx,) * 3), po.Mesh((n,) * 3, (Lx,) * 3)
cap1, cap2 = pj.Capacity(pj.Sphere(c, r), mesh), pj.Capacity(pj.Sphere(c, r, complement=True), mesh)
oc1, oc2 = oracle_capacity_from_product(cap1, omesh), oracle_capacity_from_product(cap2, omesh)
f = lambda x, y, z, t: 0.0
D1 = lambda x, y, z: 1.0
D2 = lambda x, y, z: 2.0
p1, p2 = pj.Phase(cap1, pj.DiffusionOps(cap1), f, D1), pj.Phase(cap2, pj.DiffusionOps(cap2), f, D2)
q1, q2 = po.Phase(oc1, po.make_diffusion_ops(oc1), f, D1), po.Phase(oc2, po.make_diffusion_ops(oc2), f, D2)
ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 0.5, 0.0), pj.FluxJump(1.0, 1.0, 0.0))
oic = po.InterfaceConditions(po.ScalarJump(1.0, 0.5, 0.0), po.FluxJump(1.0, 1.0, 0.0))
bcb, obcb = pj.BorderConditions({}), po.BorderConditions({})
u0 = np.concatenate([np.ones(M), np.ones(M), np.zeros(M), np.zeros(M)])
dt = 0.5 * (Lx / n) ** 2
s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, u0, "BE")
so = po.DiffusionUnsteadyDiph(q1, q2, obcb, oic, dt, u0, "BE")
A, b, idx = s.system(0)
Ar, br, oidx = po.remove_zero_rows_cols(so.A, so.b)
print("index sets equal", np.array_equal(idx, oidx), "rows", len(idx), "matrix diff", abs(A[:, :len(idx)] - Ar).max() / abs(Ar).max(),
      "rhs diff", np.max(np.abs(b - br)) / np.max(np.abs(br)))
# ---- one CN step: the HIP run system against the oracle's CN system built from the SAME previous state, and the sensitivity
# of that system
s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, u0, "BE")
so = po.DiffusionUnsteadyDiph(q1, q2, obcb, oic, dt, u0, "BE")
pj.solve_DiffusionUnsteadyDiph_b(s, p1, p2, dt, 1 * dt, bcb, ic, "CN", reltol=1e-13)
A1, b1, idx1 = s.system(1)
A2 = po.A_diph_unstead_diff(q1.operator, q2.operator, oc1, oc2, D1, D2, oic, dt, "CN")
b2 = po.b_diph_unstead_diff(q1.operator, q2.operator, f, f, oc1, oc2, D1, D2, oic, s.states[0], dt, dt, "CN")
Ar2, br2, i2 = po.remove_zero_rows_cols(A2, b2)
print("CN run system vs oracle CN system from the same previous state: index sets", np.array_equal(idx1, i2), "matrix diff",
      abs(A1[:, :len(i2)] - Ar2).max() / abs(Ar2).max(), "rhs diff max", np.max(np.abs(b1 - br2)) / np.max(np.abs(br2)),
      "rhs diff rel l2", rel_l2(b1, br2))
rowscale = np.abs(Ar2).max(axis=1).toarray().ravel()
print("rhs diff relative to the row scale, max:", np.max(np.abs(b1 - br2) / rowscale), "at row", int(np.argmax(np.abs(b1 - br2) / rowscale)),
      "block", int(i2[int(np.argmax(np.abs(b1 - br2) / rowscale))] // M))
xo = spl.splu(Ar2.tocsc()).solve(br2)
xh = s.states[1][i2]
print("step 1: hip vs direct solve of the oracle system from the same previous state", rel_l2(xh, xo))
xb = spl.splu(Ar2.tocsc()).solve(b1)
print("direct solve with the HIP rhs vs with the oracle rhs", rel_l2(xb, xo), "; hip vs direct solve with the HIP rhs", rel_l2(xh, xb))
Ap = Ar2.copy().tocsr(); Ap.data = Ap.data * (1.0 + 2.2e-16 * np.sign(np.random.default_rng(1).standard_normal(Ap.nnz)))
print("one-ulp perturbation of the CN matrix moves the direct solution by", rel_l2(spl.splu(Ap.tocsc()).solve(br2), xo))
bp = br2 * (1.0 + 2.2e-16 * np.sign(np.random.default_rng(2).standard_normal(len(br2))))
print("one-ulp perturbation of the rhs moves the direct solution by", rel_l2(spl.splu(Ar2.tocsc()).solve(bp), xo))
sys.exit(0)
import warnings
for scheme, kw in (("BE", {}), ("CN", {}), ("CN", {"warm_start": False}), ("CN", {"precond": -1}), ("CN", {"warm_start": False, "precond": -1}), ("CN", {"method": "gmres", "restart": 60})):
    s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, u0, "BE")
    so = po.DiffusionUnsteadyDiph(q1, q2, obcb, oic, dt, u0, "BE")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        pj.solve_DiffusionUnsteadyDiph_b(s, p1, p2, dt, 3 * dt, bcb, ic, scheme, reltol=1e-13, **kw)
    po.solve_DiffusionUnsteadyDiph(so, q1, q2, dt, 3 * dt, obcb, oic, scheme, method="\\")
    print(scheme, kw, "unconverged", s.unconverged, "warnings", len(w), "rel_l2 per state", ["%.1e" % rel_l2(a, o) for a, o in zip(s.states, so.states)])
    if scheme == "CN" and not kw:
        # true residual of the HIP state of step 1 in the oracle's CN system built from the ORACLE's previous state, and from the HIP one
        A2 = po.A_diph_unstead_diff(q1.operator, q2.operator, oc1, oc2, D1, D2, oic, dt, "CN")
        for name, prev in (("oracle prev", so.states[0]), ("hip prev", s.states[0])):
            b2 = po.b_diph_unstead_diff(q1.operator, q2.operator, f, f, oc1, oc2, D1, D2, oic, prev, dt, dt, "CN")
            Ar2, br2, i2 = po.remove_zero_rows_cols(A2, b2)
            xo = spl.splu(Ar2.tocsc()).solve(br2)
            xh = s.states[1][i2]
            print("   step 1,", name, ": ||b - A x_hip|| / ||b|| = %.2e, oracle's direct solve vs hip %.2e vs oracle state %.2e" % (
                np.linalg.norm(br2 - Ar2 @ xh) / np.linalg.norm(br2), rel_l2(xh, xo), rel_l2(so.states[1][i2], xo)))
