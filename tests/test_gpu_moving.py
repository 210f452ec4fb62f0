"""GPU parity of the prescribed-motion path: pg_capacity_create_spacetime + pg_solver_create_moving_mono
through penguin.jl_amd.moving against oracle/spacetime.py.

* capacities: same time rule on both sides -> agreement to rounding (the spatial code and every padding convention);
  a 4x finer rule on the oracle's side -> the time-quadrature error of the default rule, bounded here;
* algebra: the oracle assembles and solves (direct `\\`) with the capacities the HIP path computed -> states <= 1e-10."""
import numpy as np
import pytest

from oracle import penguin_oracle as po
from oracle import spacetime as ost
from tests.common import rel_l2

pytestmark = pytest.mark.gpu
TOL_T = 1e-10


def _bodies_1d(pj):
    pos, dpos = (lambda t: 0.21 + 0.9 * t + 0.5 * t * t), (lambda t: 0.9 + t)
    return pj.MovingHalfSpace(0, pos, 1.0, dposition=dpos), ost.MovingHalfSpace(0, pos, 1.0, dposition=dpos)


def _bodies_2d(pj, complement=False):
    cen, dcen = (lambda t: (2.01 + 0.8 * t, 1.97 - 0.5 * t)), (lambda t: (0.8, -0.5))
    rad, drad = (lambda t: 1.0 + 0.4 * t), (lambda t: 0.4)
    return (pj.MovingSphere(cen, rad, complement, dcenter=dcen, dradius=drad),
            ost.MovingBall(cen, rad, complement, dcenter=dcen, dradius=drad))


def _cases(pj):
    return {
        "1d": (pj.Mesh((40,), (1.0,), (0.0,)), po.Mesh((40,), (1.0,), (0.0,)), _bodies_1d(pj), 0.01),
        "2d": (pj.Mesh((16, 16), (4.0, 4.0), (0.0, 0.0)), po.Mesh((16, 16), (4.0, 4.0), (0.0, 0.0)), _bodies_2d(pj), 0.05),
        "2d_outside": (pj.Mesh((16, 16), (4.0, 4.0), (0.0, 0.0)), po.Mesh((16, 16), (4.0, 4.0), (0.0, 0.0)),
                       _bodies_2d(pj, True), 0.05),
    }


def _oracle_cap(cap, omesh, t0, t1, obody=None) -> po.Capacity:
    """The (N+1)-D oracle capacity holding the fields the HIP path computed (second time layer: A_(N+1) only)."""
    N = omesh.N
    z = np.zeros_like(cap.V)
    two = lambda a: np.concatenate([a, z])
    A = tuple(two(a) for a in cap.A) + (np.concatenate([cap.Vn_1, cap.Vn]),)
    B = tuple(two(b) for b in cap.B) + (two(z),)
    W = tuple(two(w) for w in cap.W) + (two(z),)
    zz = np.zeros((len(z), N + 1))
    return po.Capacity(A, B, two(cap.V), W, np.vstack([cap.C_ω_st, zz]), np.vstack([cap.C_γ_st, zz]), two(cap.Γ),
                       two(cap.cell_types), ost.SpaceTimeMesh(omesh, [t0, t1]), obody)


@pytest.mark.parametrize("name", ["1d", "2d", "2d_outside"])
def test_spacetime_capacities_match_oracle(pj, name):
    mesh, omesh, (body, obody), dt = _cases(pj)[name]
    N, M = omesh.N, int(np.prod(omesh.ext))
    h = float(omesh.nodes[0][1] - omesh.nodes[0][0])
    t0 = 0.1
    cap = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t0, t0 + dt]), time_panels=8, time_order=4)
    assert isinstance(cap, pj.SpaceTimeCapacity)
    same = ost.make_spacetime_capacity(obody, omesh, t0, t0 + dt, panels=8, order=4)
    lay = ost.spatial_layer(same, omesh)
    cellm, facem = h ** N * dt, h ** (N - 1) * dt
    assert np.array_equal(cap.cell_types, lay.cell_types)                       # classification bit for bit
    assert np.array_equal(np.flatnonzero(cap.Γ > 0), np.flatnonzero(lay.G > 0))
    assert np.max(np.abs(cap.V - lay.V)) <= 1e-12 * cellm
    assert np.max(np.abs(cap.Vn_1 - same.A[N][:M])) <= 1e-12 * h ** N and np.max(np.abs(cap.Vn - same.A[N][M:])) <= 1e-12 * h ** N
    assert np.max(np.abs(cap.Γ - lay.G)) <= 1e-11 * facem
    big = lay.V > 1e-3 * cellm
    assert np.max(np.abs(cap.C_ω_st[big] - same.C_w[:M][big])) <= 1e-9 * h
    cut = lay.G > 1e-3 * facem
    assert np.max(np.abs(cap.C_γ_st[cut] - same.C_g[:M][cut])) <= 1e-9 * h
    for d in range(N):
        assert np.max(np.abs(cap.A[d] - lay.A[d])) <= 1e-12 * facem
        assert np.max(np.abs(cap.B[d] - lay.B[d])) <= 1e-7 * facem               # through C_ω of tiny cells
        assert np.max(np.abs(cap.W[d] - lay.W[d])) <= 1e-7 * cellm
    # reference layout: N+1 face capacities of length 2M, the last one = [Vn_1; Vn]
    assert len(cap.A_st) == N + 1 and all(len(a) == 2 * M for a in cap.A_st)
    # the default time rule (16 x 4) against a 4x finer one: kinks of V(τ) where the interface passes a cell corner
    dflt = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t0, t0 + dt]))
    fine = ost.spatial_layer(ost.make_spacetime_capacity(obody, omesh, t0, t0 + dt, panels=64, order=4), omesh)
    assert np.max(np.abs(dflt.V - fine.V)) <= 2e-4 * cellm
    for d in range(N):
        assert np.max(np.abs(dflt.A[d] - fine.A[d])) <= (2e-2 if N == 1 else 2e-3) * facem   # 1-D: A_d(τ) is a step function


@pytest.mark.parametrize("name,scheme,bc_kind", [
    ("1d", "BE", "dirichlet"), ("1d", "CN", "dirichlet"), ("1d", "BE", "robin"),
    ("2d", "BE", "dirichlet"), ("2d", "CN", "dirichlet"), ("2d", "CN", "robin"), ("2d_outside", "BE", "dirichlet"),
])
def test_moving_steps_match_oracle(pj, name, scheme, bc_kind):
    """MovingDiffusionUnsteadyMono + solve_MovingDiffusionUnsteadyMono! (diffusion.jl:16-35, 227-268): 1 + 4 slabs, sources,
    variable D, time-dependent border and interface data, fresh and dead cells (the interface crosses cell faces)."""
    mesh, omesh, (body, obody), dt = _cases(pj)[name]
    N, M = omesh.N, int(np.prod(omesh.ext))
    f = lambda x, y, z, t: 0.3 + 0.2 * x + 0.5 * t            # (x, t_c, 0, t) in 1-D+t, (x, y, t_c, t) in 2-D+t
    D = lambda x, y, z: 1.0 + 0.1 * x
    if bc_kind == "robin":
        g = lambda x, y, z=0.0: 0.5 + 0.1 * x
        bc, obc = pj.Robin(0.7, 1.3, g), po.Robin(0.7, 1.3, g)
    else:
        g = lambda x, y, z=0.0: 1.0 + 0.2 * x + 0.3 * y       # y = t_c in 1-D+t
        bc, obc = pj.Dirichlet(g), po.Dirichlet(g)
    keys = ("bottom",) if N == 1 else ("left", "right", "top", "bottom")
    bval = lambda *a: 0.2 + 0.1 * a[-1]                       # value(x.., t)
    bcb = pj.BorderConditions({k: pj.Dirichlet(bval) for k in keys})
    obcb = po.BorderConditions({k: po.Dirichlet(bval) for k in keys})
    rng = np.random.default_rng(7)
    T0 = rng.random(2 * M)
    caps = {}

    def capacity_fn(t0, t1):
        c = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t0, t1]))
        caps[t0] = c
        return _oracle_cap(c, omesh, t0, t1, obody)

    cap0 = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [0.0, dt]))
    ph = pj.Phase(cap0, pj.DiffusionOps(cap0), f, D)
    s = pj.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, T0, mesh, scheme)
    pj.solve_MovingDiffusionUnsteadyMono_b(s, ph, body, dt, 0.0, 3.5 * dt, bcb, bc, mesh, scheme, method="bicgstab", reltol=1e-14)
    ocap0 = _oracle_cap(cap0, omesh, 0.0, dt, obody)
    oph = po.Phase(ocap0, po.make_diffusion_ops(ocap0), f, D)
    so = ost.MovingDiffusionUnsteadyMono(oph, obcb, obc, dt, T0, omesh, scheme)
    # the loop of diffusion.jl:227-268 written out (ost.solve_MovingDiffusionUnsteadyMono), so that every slab's system can
    # be probed: `sens` = how far the ORACLE's own direct solution moves when its matrix entries move by one rounding error.
    # Robin / Neumann rows on the arbitrarily small space-time cut cells a moving interface leaves behind are sensitive
    # (the small-cell problem); the parity bar is 1e-10 wherever the system itself is determined that well.
    sens, t = [], 0.0

    def solve_and_probe():
        po.solve_system(so)
        so.states.append(so.x)
        A, b = so.last_A_reduced, so.last_b_reduced
        Ap = A.copy()
        Ap.data = Ap.data * (1.0 + 2.2e-16 * np.random.default_rng(1).standard_normal(len(Ap.data)))
        import scipy.sparse.linalg as spla
        sens.append(rel_l2(spla.spsolve(Ap.tocsc(), b), spla.spsolve(A.tocsc(), b)))

    solve_and_probe()
    while t < 3.5 * dt:
        t += dt
        ocap = capacity_fn(t, t + dt)
        oop = po.make_diffusion_ops(ocap)
        so.A = ost.A_mono_unstead_diff_moving(oop, ocap, D, obc, scheme)
        so.b = ost.b_mono_unstead_diff_moving(oop, ocap, D, f, obc, so.states[-1], dt, t, scheme)
        so.A, so.b = po.BC_border_mono(so.A, so.b, obcb, omesh, t=t)
        solve_and_probe()
    assert len(s.states) == len(so.states) == 5
    assert s.unconverged == 0
    nact = set()
    for k, (x, xo) in enumerate(zip(s.states, so.states)):
        assert np.array_equal(np.flatnonzero(x != 0.0), np.flatnonzero(xo != 0.0)), f"active set of state {k}"
        tol = max(TOL_T, 50.0 * max(sens[: k + 1]))
        if bc_kind == "dirichlet":
            assert max(sens[: k + 1]) < 1e-10, f"state {k}: a Dirichlet system this sensitive ({max(sens):.1e}) is not expected"
        assert rel_l2(x, xo) <= tol, f"state {k}: {rel_l2(x, xo):.2e} (bar {tol:.1e})"
        nact.add(int(np.count_nonzero(xo)))
    assert len(nact) > 1          # the active set did change from slab to slab (fresh / dead cells were exercised)


def test_moving_uniform_state_is_preserved(pj):
    """no cell changes phase during these slabs: T = Tγ = g = border value must stay exactly that (Vn_1 - (Vn_1 - Vn) = Vn)."""
    mesh = pj.Mesh((20,), (1.0,), (0.0,))
    body = pj.MovingHalfSpace(0, lambda t: 0.301 + 0.3 * t, 1.0, dposition=lambda t: 0.3)
    dt, M = 0.01, 21
    for scheme in ("BE", "CN"):
        cap = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [0.0, dt]))
        ph = pj.Phase(cap, pj.DiffusionOps(cap), lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
        bc, bcb = pj.Dirichlet(1.0), pj.BorderConditions({"bottom": pj.Dirichlet(1.0)})
        s = pj.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, np.ones(2 * M), mesh, scheme)
        pj.solve_MovingDiffusionUnsteadyMono_b(s, ph, body, dt, 0.0, 3 * dt, bcb, bc, mesh, scheme)
        assert len(s.states) == 4
        for x in s.states:
            act = x != 0.0
            assert act.sum() >= 6 and np.abs(x[act] - 1.0).max() < 1e-11


def test_moving_solver_refuses_what_it_cannot_do(pj):
    mesh = pj.Mesh((8, 8, 8), (1.0,) * 3, (0.0,) * 3)
    body = pj.MovingSphere(lambda t: (0.5, 0.5, 0.5), lambda t: 0.3 + t)
    with pytest.raises(pj.PenguinHipError, match="1-D\\+t and 2-D\\+t"):
        pj.Capacity(body, pj.SpaceTimeMesh(mesh, [0.0, 0.1]))
    m1 = pj.Mesh((20,), (1.0,), (0.0,))
    static = pj.Capacity(pj.HalfSpace(0, 0.3), m1)
    ph = pj.Phase(static, pj.DiffusionOps(static), lambda x, y, z, t: 0.0, 1.0)
    with pytest.raises(pj.PenguinHipError, match="space-time capacity"):
        pj.MovingDiffusionUnsteadyMono(ph, pj.BorderConditions({}), pj.Dirichlet(0.0), 0.1, np.zeros(42), m1, "BE")


@pytest.mark.parametrize("scheme", ["BE", "CN"])
def test_moving_interface_similarity_solution(pj, scheme):
    """The HIP path against the analytic field of a growing half line (tests/test_oracle_spacetime.py::similarity_problem):
    second order in h with Δt = 2h², the same error levels as the oracle's run."""
    from tests.test_oracle_spacetime import similarity_problem
    pos, dpos, exact = similarity_problem()
    errs = []
    for nx in (40, 80):
        h = 1.0 / nx
        dt = 2.0 * h * h
        mesh, M = pj.Mesh((nx,), (1.0,), (0.0,)), nx + 1
        body = pj.MovingHalfSpace(0, pos, 1.0, dposition=dpos)
        cap = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [0.0, dt]), time_panels=32)
        ph = pj.Phase(cap, pj.DiffusionOps(cap), lambda x, y, z, t: 0.0, 1.0)
        bcb = pj.BorderConditions({"bottom": pj.Dirichlet(lambda x, t: exact(x + h, t + dt))})
        c0 = pj.Capacity(pj.HalfSpace(0, pos(0.0)), mesh)
        T0 = np.concatenate([np.where(c0.V > 0, exact(c0.C_ω[:, 0], 0.0), 0.0), np.zeros(M)])
        s = pj.MovingDiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(0.0), dt, T0, mesh, scheme)
        pj.solve_MovingDiffusionUnsteadyMono_b(s, ph, body, dt, 0.0, 0.1, bcb, pj.Dirichlet(0.0), mesh, scheme, time_panels=32)
        assert s.unconverged == 0
        tf = dt * len(s.states)
        cf = pj.Capacity(pj.HalfSpace(0, pos(tf)), mesh)
        x = s.states[-1][:M]
        sel = (cf.V > 0.5 * h) & (x != 0.0)
        errs.append(float(np.abs(x[sel] - exact(cf.C_ω[sel, 0], tf)).max()))
    if scheme == "BE":
        assert errs[0] < 2.5e-3 and errs[1] < 0.4 * errs[0], errs
    else:       # an order below BE at these sizes; the error then stalls near the interface (fresh cells restart from 0)
        assert errs[0] < 3e-4 and errs[1] < 3e-4, errs


def test_moving_states_handed_over_on_the_device(pj):
    """save_states=False: pg_solver_create_moving_mono_next takes the previous slab's state on the device; the last state is the
    one the host-state loop ends with, bit for bit (same systems, same start vectors)."""
    mesh, omesh, (body, obody), dt = _cases(pj)["2d"]
    M = int(np.prod(omesh.ext))
    f = lambda x, y, z, t: 0.3 + 0.2 * x + 0.5 * t
    bc = pj.Dirichlet(lambda x, y, z=0.0: 1.0 + 0.2 * x)
    bcb = pj.BorderConditions({k: pj.Dirichlet(0.2) for k in ("left", "right", "top", "bottom")})
    T0 = np.random.default_rng(9).random(2 * M)
    out = []
    for keep in (True, False):
        cap0 = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [0.0, dt]))
        ph = pj.Phase(cap0, pj.DiffusionOps(cap0), f, 1.0)
        s = pj.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, T0, mesh, "CN")
        pj.solve_MovingDiffusionUnsteadyMono_b(s, ph, body, dt, 0.0, 3.5 * dt, bcb, bc, mesh, "CN", method="bicgstab", save_states=keep)
        assert s.unconverged == 0 and len(s.states) == (5 if keep else 1)
        out.append(s.states[-1])
    assert np.array_equal(out[0], out[1])


@pytest.mark.parametrize("variant", ["hs_minus", "hs_complement", "ball_1d", "sqrt_from_zero", "on_nodes"])
def test_spacetime_capacity_variants(pj, variant):
    """Half space with the fluid on the other side / as a complement, a moving interval (1-D ball), the reference's own motion
    `xf + c sqrt(t)` started at t = 0 (infinite interface speed at the slab's lower face: Γ only, everything else exact), and
    an interface that sits exactly on mesh nodes at both time faces."""
    import math
    n, L = 40, 1.0
    mesh, omesh = pj.Mesh((n,), (L,), (0.0,)), po.Mesh((n,), (L,), (0.0,))
    h, t0, dt = L / n, 0.0, 0.01
    if variant == "hs_minus":
        pos, dpos = (lambda t: 0.62 - 0.8 * t), (lambda t: -0.8)
        body, obody = pj.MovingHalfSpace(0, pos, -1.0, dposition=dpos), ost.MovingHalfSpace(0, pos, -1.0, dposition=dpos)
    elif variant == "hs_complement":
        pos, dpos = (lambda t: 0.33 + 0.5 * t), (lambda t: 0.5)
        body = pj.MovingHalfSpace(0, pos, 1.0, complement=True, dposition=dpos)
        obody = ost.MovingHalfSpace(0, pos, 1.0, complement=True, dposition=dpos)
    elif variant == "ball_1d":
        cen, rad = (lambda t: (0.48 + 0.6 * t,)), (lambda t: 0.2 + 0.3 * t)
        body = pj.MovingSphere(cen, rad, dcenter=lambda t: (0.6,), dradius=lambda t: 0.3)
        obody = ost.MovingBall(cen, rad, dcenter=lambda t: (0.6,), dradius=lambda t: 0.3)
    elif variant == "sqrt_from_zero":
        pos = lambda t: 0.01 + 1.0 * math.sqrt(t)                       # examples/1D/SolidMoving/MovingHeat.jl:16-18
        dpos = lambda t: 0.5 / math.sqrt(t)
        body, obody = pj.MovingHalfSpace(0, pos, 1.0, dposition=dpos), ost.MovingHalfSpace(0, pos, 1.0, dposition=dpos)
    else:
        x_a, x_b = float(omesh.nodes[0][12]), float(omesh.nodes[0][13])  # on a node at t0, on the next one at t0 + dt
        pos, dpos = (lambda t: x_a + (x_b - x_a) * (t - t0) / dt), (lambda t: (x_b - x_a) / dt)
        body, obody = pj.MovingHalfSpace(0, pos, 1.0, dposition=dpos), ost.MovingHalfSpace(0, pos, 1.0, dposition=dpos)
    cap = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t0, t0 + dt]), time_panels=8, time_order=4)
    same = ost.make_spacetime_capacity(obody, omesh, t0, t0 + dt, panels=8, order=4)
    lay = ost.spatial_layer(same, omesh)
    M = n + 1
    assert np.array_equal(cap.cell_types, lay.cell_types)
    assert np.max(np.abs(cap.V - lay.V)) <= 1e-13 * h * dt
    assert np.max(np.abs(cap.Vn_1 - same.A[1][:M])) <= 1e-14 * h and np.max(np.abs(cap.Vn - same.A[1][M:])) <= 1e-14 * h
    assert np.max(np.abs(cap.A[0] - lay.A[0])) <= 1e-13 * dt and np.max(np.abs(cap.B[0] - lay.B[0])) <= 1e-13 * dt
    assert np.max(np.abs(cap.W[0] - lay.W[0])) <= 1e-8 * h * dt
    assert np.max(np.abs(cap.Γ - lay.G)) <= 1e-12 * max(lay.G.max(), dt)
    # total fluid measure of the slab: the time integral of the fluid length (closed form where the motion is linear)
    if variant in ("hs_minus", "hs_complement", "on_nodes"):
        lo_d, hi_d = float(omesh.nodes[0][0]), float(omesh.nodes[0][-1])
        p0, p1 = pos(t0), pos(t0 + dt)
        mean_pos = 0.5 * (p0 + p1)
        fluid_right = (variant == "hs_minus") or (variant == "hs_complement")
        exact = (hi_d - mean_pos) * dt if fluid_right else (mean_pos - lo_d) * dt
        assert abs(cap.V.sum() - exact) <= 1e-14


@pytest.mark.parametrize("seed", range(8))
def test_moving_random_problems(pj, seed):
    """seeded random moving problems (1-D half lines and intervals, 2-D discs inside / outside; BE / CN; Dirichlet / Robin
    interface data; random border sets): 1 + 3 slabs against the literal oracle fed with the HIP capacities."""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(2000 + seed)
    N = 1 if seed % 3 == 0 else 2
    n = int(rng.integers(14, 26)) if N == 1 else int(rng.integers(10, 18))
    L, x0 = float(rng.uniform(1.0, 3.0)), float(rng.uniform(-1.0, 0.5))
    mesh, omesh = pj.Mesh((n,) * N, (L,) * N, (x0,) * N), po.Mesh((n,) * N, (L,) * N, (x0,) * N)
    h = L / n
    dt = float(rng.uniform(0.3, 1.5)) * h * h
    speed = float(rng.uniform(0.2, 0.9)) * h / dt                   # crosses most of a cell per slab: fresh / dead cells
    comp = bool(rng.integers(0, 2))
    if N == 1 and seed % 2 == 0:
        p0, sgn = x0 + L * float(rng.uniform(0.3, 0.7)), float(rng.choice([-1.0, 1.0]))
        v = speed * float(rng.choice([-1.0, 1.0]))
        pos, dpos = (lambda t: p0 + v * t), (lambda t: v)
        body, obody = pj.MovingHalfSpace(0, pos, sgn, complement=comp, dposition=dpos), ost.MovingHalfSpace(0, pos, sgn, comp, dposition=dpos)
    else:
        c0 = x0 + L * rng.uniform(0.4, 0.6, N)
        vel = rng.uniform(-1, 1, N)
        vel = vel / np.linalg.norm(vel) * speed * 0.5
        r0, dr = L * float(rng.uniform(0.18, 0.28)), speed * 0.5 * float(rng.choice([-1.0, 1.0]))
        cen, rad = (lambda t: tuple(c0 + vel * t)), (lambda t: r0 + dr * t)
        body = pj.MovingSphere(cen, rad, comp, dcenter=lambda t: tuple(vel), dradius=lambda t: dr)
        obody = ost.MovingBall(cen, rad, comp, dcenter=lambda t: tuple(vel), dradius=lambda t: dr)
    scheme = "CN" if rng.integers(0, 2) else "BE"
    kind = int(rng.integers(0, 3))
    a, b, c = rng.uniform(0.2, 1.0, 3)
    g = lambda x, y, z=0.0: a + b * x + c * y
    if kind == 0:
        bc, obc = pj.Dirichlet(g), po.Dirichlet(g)
    elif kind == 1:
        bc, obc = pj.Robin(1.0, 0.3, g), po.Robin(1.0, 0.3, g)
    else:      # (pure Neumann data on the tiny fresh cells of a slab make the reference's system singular to working precision)
        bc, obc = pj.Robin(0.6, 1.0, g), po.Robin(0.6, 1.0, g)
    keys = [k for k in (("bottom", "top") if N == 1 else ("left", "right", "top", "bottom")) if rng.random() < 0.7]
    bval = lambda *p: 0.3 + 0.2 * p[-1]
    bcb, obcb = pj.BorderConditions({k: pj.Dirichlet(bval) for k in keys}), po.BorderConditions({k: po.Dirichlet(bval) for k in keys})
    f = lambda x, y, z, t: 0.1 + 0.3 * x + t
    D = lambda x, y, z: 1.0 + 0.2 * x * x
    M = int(np.prod(omesh.ext))
    T0 = rng.random(2 * M)
    cap0 = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [0.0, dt]))
    ph = pj.Phase(cap0, pj.DiffusionOps(cap0), f, D)
    s = pj.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, T0, mesh, scheme)
    pj.solve_MovingDiffusionUnsteadyMono_b(s, ph, body, dt, 0.0, 2.5 * dt, bcb, bc, mesh, scheme, method="bicgstab", reltol=1e-14)
    assert s.unconverged == 0 and len(s.states) == 4
    ocap0 = _oracle_cap(cap0, omesh, 0.0, dt, obody)
    so = ost.MovingDiffusionUnsteadyMono(po.Phase(ocap0, po.make_diffusion_ops(ocap0), f, D), obcb, obc, dt, T0, omesh, scheme)
    t, sens = 0.0, []

    systems = []

    def solve_and_probe():
        po.solve_system(so)
        so.states.append(so.x)
        A, bb = so.last_A_reduced, so.last_b_reduced
        systems.append((A, bb, so.last_idx))
        Ap = A.copy()
        Ap.data = Ap.data * (1.0 + 2.2e-16 * np.random.default_rng(1).standard_normal(len(Ap.data)))
        sens.append(rel_l2(spla.spsolve(Ap.tocsc(), bb), spla.spsolve(A.tocsc(), bb)))

    solve_and_probe()
    while t < 2.5 * dt:
        t += dt
        ocap = _oracle_cap(pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t, t + dt])), omesh, t, t + dt, obody)
        oop = po.make_diffusion_ops(ocap)
        so.A = ost.A_mono_unstead_diff_moving(oop, ocap, D, obc, scheme)
        so.b = ost.b_mono_unstead_diff_moving(oop, ocap, D, f, obc, so.states[-1], dt, t, scheme)
        so.A, so.b = po.BC_border_mono(so.A, so.b, obcb, omesh, t=t)
        solve_and_probe()
    for k, (x, xo) in enumerate(zip(s.states, so.states)):
        assert np.array_equal(np.flatnonzero(x != 0.0), np.flatnonzero(xo != 0.0)), f"active set of state {k}"
        tol = max(TOL_T, 50.0 * max(sens[: k + 1]))
        if tol < 1e-7:
            assert rel_l2(x, xo) <= tol, f"state {k}: {rel_l2(x, xo):.2e} (bar {tol:.1e})"
        else:
            # an isolated sliver cell with Robin data (all its faces closed) leaves the reference's system singular to working
            # precision: its own direct solution is not determined; the HIP state must still SOLVE that system -- componentwise
            # backward error of the oracle's matrix and right-hand side (built from the previous ORACLE state, which is why this
            # ends the comparison of the run)
            A, bb, idx = systems[k]
            xr = x[idx]
            berr = np.abs(A @ xr - bb) / (abs(A) @ np.abs(xr) + np.abs(bb) + 1e-300)
            assert berr.max() <= 1e-9, f"state {k}: backward error {berr.max():.2e}"
            break


# ------------------------------------------------------------------------------------------------ two phases
def _diph_bodies(pj, name):
    """(body, complement) on the product's side and on the oracle's, for the moving two-phase cases."""
    if name == "1d":
        pos, dpos = (lambda t: 0.21 + 0.9 * t + 0.5 * t * t), (lambda t: 0.9 + t)
        return ((pj.MovingHalfSpace(0, pos, 1.0, dposition=dpos), pj.MovingHalfSpace(0, pos, 1.0, complement=True, dposition=dpos)),
                (ost.MovingHalfSpace(0, pos, 1.0, dposition=dpos), ost.MovingHalfSpace(0, pos, 1.0, complement=True, dposition=dpos)))
    cen, dcen = (lambda t: (2.01 + 0.8 * t, 1.97 - 0.5 * t)), (lambda t: (0.8, -0.5))
    rad, drad = (lambda t: 1.0 + 0.4 * t), (lambda t: 0.4)
    return ((pj.MovingSphere(cen, rad, False, dcenter=dcen, dradius=drad), pj.MovingSphere(cen, rad, True, dcenter=dcen, dradius=drad)),
            (ost.MovingBall(cen, rad, False, dcenter=dcen, dradius=drad), ost.MovingBall(cen, rad, True, dcenter=dcen, dradius=drad)))


@pytest.mark.parametrize("name,scheme", [("1d", "BE"), ("1d", "CN"), ("2d", "BE"), ("2d", "CN")])
def test_moving_diphasic_steps_match_oracle(pj, name, scheme):
    """MovingDiffusionUnsteadyDiph + solve_MovingDiffusionUnsteadyDiph! (prescribedmotionsolver/diffusion.jl:272-535): 1 + 4
    slabs of a body and its complement that move through the mesh (cells change phase: fresh and dead cells on both sides),
    sources, two diffusivities, jump data that vary along the interface, border rows in BOTH phases (the driver calls
    BC_border_diph! without capacities).  The oracle assembles the literal (N+1)-D blocks from the capacities the HIP path
    computed and solves directly; the bar is 1e-10 wherever the oracle's own direct solution is determined that well
    (`sens`: how far it moves under a one-ulp perturbation of its matrix -- the small-cell problem of moving cut cells)."""
    import scipy.sparse.linalg as spla

    mesh, omesh, _, dt = _cases(pj)[name]
    (body, body_c), (obody, obody_c) = _diph_bodies(pj, name)
    N, M = omesh.N, int(np.prod(omesh.ext))
    f1 = lambda x, y, z, t: 0.3 + 0.2 * x + 0.5 * t
    f2 = lambda x, y, z, t: 0.1 - 0.1 * x + 0.2 * t
    D1 = lambda x, y, z: 1.0 + 0.1 * x
    D2 = lambda x, y, z: 2.0
    gj = lambda x, y, z=0.0: 0.2 + 0.1 * x                         # jump data at the space-time interface centroids
    hj = lambda x, y, z=0.0: 0.5 - 0.05 * x
    ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 0.5, gj), pj.FluxJump(1.0, 2.0, hj))
    oic = po.InterfaceConditions(po.ScalarJump(1.0, 0.5, gj), po.FluxJump(1.0, 2.0, hj))
    keys = ("bottom", "top") if N == 1 else ("left", "right", "top", "bottom")
    bcb = pj.BorderConditions({k: pj.Dirichlet(0.3) for k in keys})
    obcb = po.BorderConditions({k: po.Dirichlet(0.3) for k in keys})
    T0 = np.random.default_rng(11).random(4 * M)

    def hip_caps(t0, t1):
        return pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t0, t1])), pj.Capacity(body_c, pj.SpaceTimeMesh(mesh, [t0, t1]))

    c1, c2 = hip_caps(0.0, dt)
    p1, p2 = pj.Phase(c1, pj.DiffusionOps(c1), f1, D1), pj.Phase(c2, pj.DiffusionOps(c2), f2, D2)
    s = pj.MovingDiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, T0, mesh, scheme)
    pj.solve_MovingDiffusionUnsteadyDiph_b(s, p1, p2, body, body_c, dt, 3.5 * dt, bcb, ic, mesh, scheme, method="bicgstab", reltol=1e-14)
    assert s.unconverged == 0 and len(s.states) == 5
    # the oracle, slab by slab, on the HIP capacities
    oc1, oc2 = _oracle_cap(c1, omesh, 0.0, dt, obody), _oracle_cap(c2, omesh, 0.0, dt, obody_c)
    q1, q2 = po.Phase(oc1, po.make_diffusion_ops(oc1), f1, D1), po.Phase(oc2, po.make_diffusion_ops(oc2), f2, D2)
    so = ost.MovingDiffusionUnsteadyDiph(q1, q2, obcb, oic, dt, T0, omesh, scheme)
    sens = []

    def solve_and_probe():
        po.solve_system(so)
        so.states.append(so.x)
        A, b = so.last_A_reduced, so.last_b_reduced
        Ap = A.copy()
        Ap.data = Ap.data * (1.0 + 2.2e-16 * np.random.default_rng(1).standard_normal(len(Ap.data)))
        sens.append(rel_l2(spla.spsolve(Ap.tocsc(), b), spla.spsolve(A.tocsc(), b)))

    solve_and_probe()
    t = 0.0
    while t < 3.5 * dt:
        t += dt
        h1, h2 = hip_caps(t, t + dt)
        k1, k2 = _oracle_cap(h1, omesh, t, t + dt, obody), _oracle_cap(h2, omesh, t, t + dt, obody_c)
        o1, o2 = po.make_diffusion_ops(k1), po.make_diffusion_ops(k2)
        so.A = ost.A_diph_unstead_diff_moving(o1, o2, k1, k2, D1, D2, oic, scheme)
        # (previous state: the HIP one -- every slab is compared on identical inputs, as the step-by-step diphasic test does)
        so.b = ost.b_diph_unstead_diff_moving(o1, o2, k1, k2, D1, D2, f1, f2, oic, s.states[len(so.states) - 1], dt, t, scheme)
        so.A, so.b = ost._border_diph(so.A, so.b, obcb, k1, k2, omesh, None)
        solve_and_probe()
    assert len(so.states) == 5
    nact = set()
    for k, (x, xo) in enumerate(zip(s.states, so.states)):
        assert np.array_equal(np.flatnonzero(x != 0.0), np.flatnonzero(xo != 0.0)), f"active set of state {k}"
        tol = max(TOL_T, 50.0 * sens[k])
        assert rel_l2(x, xo) <= tol, f"state {k}: {rel_l2(x, xo):.2e} (bar {tol:.1e}, sensitivity {sens[k]:.1e})"
        nact.add(int(np.count_nonzero(xo)))
    assert len(nact) > 1          # cells did change phase from slab to slab
