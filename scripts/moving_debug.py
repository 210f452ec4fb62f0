import sys, numpy as np
sys.path.insert(0, "/root/repo")
import penguin.jl_amd as pj
from oracle import penguin_oracle as po, spacetime as ost
from tests.test_gpu_moving import _cases, _oracle_cap
from tests.common import rel_l2
import scipy.sparse.linalg as spla
pj.init(0)
for name, scheme, kind in (("2d","CN","robin"),("2d","BE","robin"),("1d","CN","robin"),("2d","CN","dirichlet")):
    mesh, omesh, (body, obody), dt = _cases(pj)[name]
    N, M = omesh.N, int(np.prod(omesh.ext))
    f = lambda x, y, z, t: 0.3 + 0.2 * x + 0.5 * t
    D = lambda x, y, z: 1.0 + 0.1 * x
    if kind == "robin":
        g = lambda x, y, z=0.0: 0.5 + 0.1 * x
        bc, obc = pj.Robin(0.7, 1.3, g), po.Robin(0.7, 1.3, g)
    else:
        g = lambda x, y, z=0.0: 1.0 + 0.2 * x + 0.3 * y
        bc, obc = pj.Dirichlet(g), po.Dirichlet(g)
    keys = ("bottom",) if N == 1 else ("left", "right", "top", "bottom")
    bval = lambda *a: 0.2 + 0.1 * a[-1]
    bcb = pj.BorderConditions({k: pj.Dirichlet(bval) for k in keys})
    obcb = po.BorderConditions({k: po.Dirichlet(bval) for k in keys})
    T0 = np.random.default_rng(7).random(2 * M)
    cap0 = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [0.0, dt]))
    ph = pj.Phase(cap0, pj.DiffusionOps(cap0), f, D)
    s = pj.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, T0, mesh, scheme)
    A, b, idx = s.system(0)
    ocap0 = _oracle_cap(cap0, omesh, 0.0, dt, obody)
    oph = po.Phase(ocap0, po.make_diffusion_ops(ocap0), f, D)
    so = ost.MovingDiffusionUnsteadyMono(oph, obcb, obc, dt, T0, omesh, scheme)
    Ar, br, oidx = po.remove_zero_rows_cols(so.A, so.b)
    print(name, scheme, kind, "idx equal", np.array_equal(idx, oidx), "n", len(idx))
    A = A[:, :len(idx)]
    dA = abs(A - Ar)
    rowscale = np.asarray(abs(Ar).max(axis=1).todense()).ravel()
    print("  max |dA|/rowmax", (np.asarray(dA.max(axis=1).todense()).ravel() / rowscale).max(), " max|db|/|b|", np.abs(b - br).max() / np.abs(br).max())
    x1 = spla.spsolve(A.tocsc(), b); x2 = spla.spsolve(Ar.tocsc(), br)
    print("  direct solves of the two systems differ by", rel_l2(x1, x2), " cond est", np.linalg.cond(Ar.toarray()))
    x3 = spla.spsolve(Ar.tocsc(), b)
    print("  same A, the two b:", rel_l2(x3, x2))
    # row-wise b difference relative to row scale
    db = np.abs(b - br) / rowscale
    k = np.argmax(db); print("  worst b row", k, "kind", "w" if idx[k] < M else "g", db[k], b[k], br[k])
