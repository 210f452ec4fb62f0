"""Diagnostic: moving diphasic slabs, HIP system vs the oracle's literal blocks on the same capacities / previous state."""
import sys
import numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from oracle import penguin_oracle as po
from oracle import spacetime as ost
from tests.common import rel_l2
from tests.test_gpu_moving import _cases, _diph_bodies, _oracle_cap
from penguin.jl_amd import moving as mv
pj.init(0)
name, scheme = sys.argv[1], sys.argv[2]
mesh, omesh, _, dt = _cases(pj)[name]
(body, body_c), (obody, obody_c) = _diph_bodies(pj, name)
N, M = omesh.N, int(np.prod(omesh.ext))
f1 = lambda x, y, z, t: 0.3 + 0.2 * x + 0.5 * t
f2 = lambda x, y, z, t: 0.1 - 0.1 * x + 0.2 * t
D1 = lambda x, y, z: 1.0 + 0.1 * x
D2 = lambda x, y, z: 2.0
gj = lambda x, y, z=0.0: 0.2 + 0.1 * x
hj = lambda x, y, z=0.0: 0.5 - 0.05 * x
ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 0.5, gj), pj.FluxJump(1.0, 2.0, hj))
oic = po.InterfaceConditions(po.ScalarJump(1.0, 0.5, gj), po.FluxJump(1.0, 2.0, hj))
keys = ("bottom", "top") if N == 1 else ("left", "right", "top", "bottom")
bcb = pj.BorderConditions({k: pj.Dirichlet(0.3) for k in keys})
obcb = po.BorderConditions({k: po.Dirichlet(0.3) for k in keys})
T = np.random.default_rng(11).random(4 * M)
s = None
for k in range(4):
    t = k * dt
    c1, c2 = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t, t + dt])), pj.Capacity(body_c, pj.SpaceTimeMesh(mesh, [t, t + dt]))
    p1, p2 = pj.Phase(c1, pj.DiffusionOps(c1), f1, D1), pj.Phase(c2, pj.DiffusionOps(c2), f2, D2)
    if s is None:
        s = pj.MovingDiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, T, mesh, scheme)
    else:
        mv._create_step_diph(s, p1, p2, bcb, ic, dt, T, mesh, scheme, t)
    A, b, idx = s.system(0)
    k1, k2 = _oracle_cap(c1, omesh, t, t + dt, obody), _oracle_cap(c2, omesh, t, t + dt, obody_c)
    o1, o2 = po.make_diffusion_ops(k1), po.make_diffusion_ops(k2)
    Ao = ost.A_diph_unstead_diff_moving(o1, o2, k1, k2, D1, D2, oic, scheme)
    bo = ost.b_diph_unstead_diff_moving(o1, o2, k1, k2, D1, D2, f1, f2, oic, T, dt, t, scheme)
    Ao, bo = ost._border_diph(Ao, bo, obcb, k1, k2, omesh, None)
    Ar, br, oidx = po.remove_zero_rows_cols(Ao, bo)
    same = np.array_equal(idx, oidx)
    print(f"slab {k}: rows hip {len(idx)} oracle {len(oidx)} index sets equal {same}")
    if same:
        dA = abs(A[:, :len(idx)] - Ar)
        i, j = np.unravel_index(np.argmax(dA.toarray() if dA.shape[0] < 3000 else dA.todense()), dA.shape)
        print("   matrix diff", dA.max() / abs(Ar).max(), "at row", i, "block", int(idx[i] // M), "cell", int(idx[i] % M), "col block", int(idx[j] // M), "col cell", int(idx[j] % M),
              "hip", A[i, j], "oracle", Ar[i, j])
        db = np.abs(b - br)
        r = int(np.argmax(db))
        print("   rhs diff", db.max() / np.abs(br).max(), "at row", r, "block", int(idx[r] // M), "cell", int(idx[r] % M), "hip", b[r], "oracle", br[r])
    else:
        only_h, only_o = np.setdiff1d(idx, oidx), np.setdiff1d(oidx, idx)
        print("   only hip", [(int(v // M), int(v % M)) for v in only_h[:8]], "only oracle", [(int(v // M), int(v % M)) for v in only_o[:8]])
    import ctypes as C
    from penguin.jl_amd import _lib as L
    opts = pj.api._krylov_opts("bicgstab", {"reltol": 1e-14})
    info = L.pg_step_info()
    L.check(L.lib().pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
    T = s._fetch_state(-1)
    import scipy.sparse.linalg as spla
    xo = np.zeros(4 * M); xo[oidx] = spla.spsolve(Ar.tocsc(), br)
    print("   solve: converged", info.converged, "iters", info.iters, "resnorm/bnorm", info.resnorm / max(info.bnorm, 1e-300), "hip vs oracle direct", rel_l2(T, xo))
    res = br - Ar @ T[oidx]
    rowmax = np.abs(Ar).max(axis=1).toarray().ravel()
    d = np.abs(Ar.diagonal())
    worst = np.argsort(-np.abs(res) / rowmax)[:4]
    print("   true residual / ||b||:", np.linalg.norm(res) / np.linalg.norm(br), " worst rows (block, cell, |res|/rowmax, |a_ii|/rowmax):",
          [(int(oidx[r] // M), int(oidx[r] % M), float(abs(res[r]) / rowmax[r]), float(d[r] / rowmax[r])) for r in worst])
    print("   smallest |a_ii| / rowmax over the rows:", float(np.min(d / rowmax)), "rows with |a_ii| < 1e-6 rowmax:", int(np.sum(d < 1e-6 * rowmax)))
    Ah, bh, _ = s.system(2)
    Ah = Ah[:, :len(idx)]
    xs = T[oidx] / 1.0
    print("   preconditioned system: max |entry|", float(abs(Ah).max()), "min/max |diag|", float(np.min(np.abs(Ah.diagonal()))), float(np.max(np.abs(Ah.diagonal()))),
          "row scaling S min/max", float(s.row_scaling(0).min()), float(s.row_scaling(0).max()))
    y = spla.spsolve(Ah.tocsc(), bh)
    xs2 = s.row_scaling(0) * y
    print("   direct solve of the EXPORTED PRECONDITIONED system: vs hip", rel_l2(T[oidx], xs2), " vs oracle direct", rel_l2(xs2, xo[oidx]))
    # rows of the raw exported system vs oracle at the worst jump rows
    for r in worst[:2]:
        print("      row", r, "kind", int(oidx[r] // M), "cell", int(oidx[r] % M), "hip b", b[r], "oracle b", br[r], "hip row", A[r].toarray().ravel()[A[r].toarray().ravel() != 0][:8], "oracle row", Ar[r].toarray().ravel()[Ar[r].toarray().ravel() != 0][:8])
    cw, cg = c1.C_γ_st, k1.C_g
    cell = int(oidx[worst[0]] % M)
    print("      C_γ_st hip", cw[cell], "oracle cap C_g", cg[cell], "Γ1", c1.Γ[cell], "Γ2", c2.Γ[cell], "g hip-side eval", gj(*cw[cell][:2]))
