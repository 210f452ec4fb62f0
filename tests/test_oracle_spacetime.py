"""Pins of the ORACLE's prescribed-motion restatement (oracle/spacetime.py) -- CPU, no GPU.

What the reference itself holds for this row: the SpaceTimeMesh vectors of test/mesh_test.jl:57-80 (golden, checked here) and
the algebra of prescribedmotionsolver/diffusion.jl (restated literally on the (N+1)-D Kronecker operators).  The capacities
come from libvofi there: unpinned, checked here through their defining properties only."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import penguin_oracle as po
from oracle import spacetime as ost


def test_spacetime_mesh_golden_vectors():
    """test/mesh_test.jl:57-66."""
    mesh = po.Mesh((5,), (1.0,), (0.0,))
    st = ost.SpaceTimeMesh(mesh, [0.0, 0.1])
    assert [list(c) for c in st.centers] == [[0.0, 0.2, 0.4, 0.6000000000000001, 0.8], [0.05]]
    assert [list(c) for c in st.nodes] == [[0.1, 0.30000000000000004, 0.5, 0.7000000000000001, 0.9, 1.1], [0.0, 0.1]]
    assert st.nC() == 5 and st.dims == (5, 1)


def test_psi_tables():
    """diffusion.jl:55-98: (Vn, Vn_1) -> weight; fresh = empty at the first face, dead = empty at the second."""
    assert [ost.psip_cn(*a) for a in ((0, 0), (1, 1), (0, 1), (1, 0))] == [0.0, 0.5, 0.5, 1.0]
    assert [ost.psim_cn(*a) for a in ((0, 0), (1, 1), (0, 1), (1, 0))] == [0.0, 0.5, 0.5, 0.0]
    assert [ost.psip_be(*a) for a in ((0, 0), (1, 1), (0, 1), (1, 0))] == [0.0, 1.0, 1.0, 1.0]
    assert [ost.psim_be(*a) for a in ((0, 0), (1, 1), (0, 1), (1, 0))] == [0.0] * 4


def _case_1d():
    mesh = po.Mesh((20,), (1.0,), (0.0,))
    body = ost.MovingHalfSpace(0, lambda t: 0.301 + 0.3 * t, 1.0, dposition=lambda t: 0.3)
    return mesh, body, 0.01


def _case_2d():
    mesh = po.Mesh((10, 10), (4.0, 4.0), (0.0, 0.0))
    body = ost.MovingBall(lambda t: (2.01 + 0.8 * t, 1.97 - 0.5 * t), lambda t: 1.0 + 0.4 * t,
                          dcenter=lambda t: (0.8, -0.5), dradius=lambda t: 0.4)
    return mesh, body, 0.05


def test_spacetime_capacity_defining_properties():
    """static body: every first-layer field is Δt times the static one; moving: ∫V = what the two faces and the swept volume
    say, ∫Γ√(1+v²) of a moving point = the length of its (x, t) curve."""
    mesh = po.Mesh((10, 10), (4.0, 4.0), (0.0, 0.0))
    still = ost.MovingBall(lambda t: (2.01, 1.97), lambda t: 1.0, dcenter=lambda t: (0.0, 0.0), dradius=lambda t: 0.0)
    dt = 0.05
    cst = ost.make_spacetime_capacity(still, mesh, 0.2, 0.2 + dt, panels=2, order=3)
    c0 = po.make_capacity(still.at(0.2), mesh)
    M = int(np.prod(mesh.ext))
    lay = ost.spatial_layer(cst, mesh)
    for d in range(2):
        assert np.allclose(lay.A[d], dt * c0.A[d], rtol=1e-13, atol=1e-16)
        assert np.allclose(lay.B[d], dt * c0.B[d], rtol=1e-13, atol=1e-16)
        assert np.allclose(lay.W[d], dt * c0.W[d], rtol=1e-13, atol=1e-16)
    assert np.allclose(lay.V, dt * c0.V, rtol=1e-13) and np.allclose(lay.G, dt * c0.G, rtol=1e-13)
    assert np.allclose(lay.C_w, c0.C_w, atol=1e-13) and np.array_equal(lay.cell_types, c0.cell_types)
    assert np.allclose(cst.A[2][:M], c0.V) and np.allclose(cst.A[2][M:], c0.V)
    # second time layer: only the time-face capacity
    assert not cst.V[M:].any() and not cst.A[0][M:].any() and not cst.B[0][M:].any() and not cst.W[1][M:].any()

    mesh1, body1, dt1 = _case_1d()
    c1 = ost.make_spacetime_capacity(body1, mesh1, 0.0, dt1)
    M1 = 21
    # total fluid measure: ∫(s(t) - x_lo)dt over the slab, the domain starting at the first node
    x_lo = mesh1.nodes[0][0]
    assert abs(c1.V[:M1].sum() - ((0.301 - x_lo) * dt1 + 0.3 * dt1 ** 2 / 2)) < 1e-15
    assert abs(c1.G[:M1].sum() - dt1 * np.sqrt(1 + 0.3 ** 2)) < 1e-15     # the curve x = s(t) stays in one cell here
    assert abs((c1.A[1][M1:] - c1.A[1][:M1]).sum() - 0.3 * dt1) < 1e-15   # V(t1) - V(t0) = swept length


@pytest.mark.parametrize("case,scheme", [(_case_1d, "BE"), (_case_1d, "CN"), (_case_2d, "BE"), (_case_2d, "CN")])
def test_half_selection_equals_first_layer_space_operators(case, scheme):
    """The `[1:end÷2]` selections of diffusion.jl:113-151 on the (N+1)-D operators leave exactly the N-D operators built from
    the first time layer -- the form the HIP path assembles (pg_stencil.h, eval_row)."""
    mesh, body, dt = case()
    cap = ost.make_spacetime_capacity(body, mesh, 0.1, 0.1 + dt, panels=4, order=3)
    op = po.make_diffusion_ops(cap)
    bc = po.Robin(0.7, 1.3, 0.4)
    D = lambda x, y, z: 1.0 + 0.1 * x
    A = ost.A_mono_unstead_diff_moving(op, cap, D, bc, scheme)
    M = int(np.prod(mesh.ext))
    assert A.shape == (2 * M, 2 * M)
    lay = ost.spatial_layer(cap, mesh)
    ops = po.make_diffusion_ops(lay)
    L_, Mx, P, Q = po._blocks(ops)
    Vn_1, Vn = cap.A[mesh.N][:M], cap.A[mesh.N][M:]
    psip = ost.psip_cn if scheme == "CN" else ost.psip_be
    Psi = sp.diags(np.array([psip(a, b) for a, b in zip(Vn, Vn_1)]))
    Id = sp.diags(np.array([D(*c) for c in po.get_all_coordinates(cap.C_w[:M])]))
    ref = sp.bmat([[sp.diags(Vn_1) + Id @ L_ @ Psi, -(sp.diags(Vn_1) - sp.diags(Vn)) + Id @ Mx @ Psi],
                   [1.3 * P, 1.3 * Q + 0.7 * sp.diags(lay.G)]], format="csr")
    assert abs(A - ref).max() <= 1e-14 * abs(ref).max()


@pytest.mark.parametrize("scheme", ["BE", "CN"])
def test_uniform_state_is_preserved_while_no_cell_changes_phase(scheme):
    """Vn_1 T - (Vn_1 - Vn) Tγ = Vn T_old and G + H annihilate constants: T = Tγ = g = border value stays put, to rounding,
    as long as no cell becomes fluid (a fresh cell carries the old value 0 of an eliminated unknown -- reference behaviour)."""
    mesh, body, dt = _case_1d()
    M = 21
    cap = ost.make_spacetime_capacity(body, mesh, 0.0, dt)
    ph = po.Phase(cap, po.make_diffusion_ops(cap), lambda x, y, z, t=0.0: 0.0, lambda x, y, z: 1.0)
    bc, bcb = po.Dirichlet(1.0), po.BorderConditions({"bottom": po.Dirichlet(1.0)})
    s = ost.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, np.ones(2 * M), mesh, scheme)
    ost.solve_MovingDiffusionUnsteadyMono(s, ph, body, dt, 0.0, 3 * dt, bcb, bc, mesh, scheme)
    assert len(s.states) == 4
    for x in s.states:
        act = x != 0.0
        assert act.sum() >= 6 and np.abs(x[act] - 1.0).max() < 1e-13


def test_moving_interface_heats_the_fluid_it_uncovers():
    """a half line that grows with a hot interface (examples/1D/SolidMoving/MovingHeat.jl, smooth motion): monotone profile
    between the border value and the interface value, bounded by both."""
    mesh = po.Mesh((40,), (1.0,), (0.0,))
    body = ost.MovingHalfSpace(0, lambda t: 0.21 + 1.0 * t, 1.0, dposition=lambda t: 1.0)
    dt, M = 0.01, 41
    cap = ost.make_spacetime_capacity(body, mesh, 0.0, dt, panels=16)
    ph = po.Phase(cap, po.make_diffusion_ops(cap), lambda x, y, z, t=0.0: 0.0, lambda x, y, z: 1.0)
    bc, bcb = po.Dirichlet(1.0), po.BorderConditions({"bottom": po.Dirichlet(0.0)})
    s = ost.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, np.zeros(2 * M), mesh, "BE")
    ost.solve_MovingDiffusionUnsteadyMono(s, ph, body, dt, 0.0, 0.2, bcb, bc, mesh, "BE",
                                          capacity_fn=lambda a, b: ost.make_spacetime_capacity(body, mesh, a, b, panels=16))
    Tw = s.states[-1][:M]
    act = np.flatnonzero(s.states[-1][:M] != 0.0)
    assert len(act) > 15                                   # the fluid domain grew from 4 to ~8 cells... and beyond
    assert Tw.min() >= -1e-12 and Tw.max() <= 1.0 + 1e-12
    assert np.all(np.diff(Tw[act]) > -1e-12)               # rises towards the hot interface


def similarity_problem(lam=0.6, t_shift=0.05):
    """Heat conduction in 0 < x < s(t) = 2λ√(t + t_shift) with T(0) = 1, T(s) = 0: T = 1 - erf(x / 2√(t + t_shift)) / erf(λ)
    (the temperature field of the one-phase Stefan problem; examples/1D/SolidMoving/MovingHeat.jl moves its interface the
    same way, `xf + c√t`).  The clock is shifted so that the run starts at t = 0 with a smooth state."""
    import math
    from scipy.special import erf
    pos = lambda t: 2 * lam * math.sqrt(t + t_shift)
    dpos = lambda t: lam / math.sqrt(t + t_shift)
    exact = lambda x, t: 1.0 - erf(np.asarray(x) / (2 * np.sqrt(t + t_shift))) / erf(lam)
    return pos, dpos, exact


def run_similarity_oracle(nx, scheme, Te=0.1):
    pos, dpos, exact = similarity_problem()
    h = 1.0 / nx
    dt = 2.0 * h * h
    mesh, M = po.Mesh((nx,), (1.0,), (0.0,)), nx + 1
    body = ost.MovingHalfSpace(0, pos, 1.0, dposition=dpos)
    mk = lambda a, b: ost.make_spacetime_capacity(body, mesh, a, b, panels=32)
    cap = mk(0.0, dt)
    ph = po.Phase(cap, po.make_diffusion_ops(cap), lambda x, y, z, t=0.0: 0.0, lambda x, y, z: 1.0)
    # the first cell spans [h/2, 3h/2] although mesh.centers[0] = 0 (src/mesh.jl:49-50): its unknown sits at x = h; the border
    # rows are applied with the time of the slab's START while the unknown lives at its end
    bcb = po.BorderConditions({"bottom": po.Dirichlet(lambda x, t: float(exact(x + h, t + dt)))})
    c0 = po.make_capacity(body.at(0.0), mesh)
    T0 = np.concatenate([np.where(c0.V > 0, exact(c0.C_w[:, 0], 0.0), 0.0), np.zeros(M)])
    s = ost.MovingDiffusionUnsteadyMono(ph, bcb, po.Dirichlet(0.0), dt, T0, mesh, scheme)
    ost.solve_MovingDiffusionUnsteadyMono(s, ph, body, dt, 0.0, Te, bcb, po.Dirichlet(0.0), mesh, scheme, capacity_fn=mk)
    tf = dt * len(s.states)
    cf = po.make_capacity(body.at(tf), mesh)
    x = s.states[-1][:M]
    sel = (cf.V > 0.5 * h) & (x != 0.0)
    return float(np.abs(x[sel] - exact(cf.C_w[sel, 0], tf)).max())


@pytest.mark.parametrize("scheme", ["BE", "CN"])
def test_moving_interface_similarity_solution(scheme):
    """known answer for the moving blocks as restated (names Vn_1 / Vn, Ψ, the swept-volume term): the error against the
    similarity solution falls at second order in h with Δt = 2h²."""
    e20, e40 = run_similarity_oracle(20, scheme), run_similarity_oracle(40, scheme)
    assert e20 < 8e-3 and e40 < 2.5e-3
    assert e40 < 0.4 * e20


def test_reference_3d_plus_time_selection_keeps_x_and_y_only():
    """Why the HIP path refuses 3-D+t (pg_capacity_create_spacetime: N = 1, 2): on a 3-D space mesh the `[1:end÷2]` halves of
    diffusion.jl:145-151 are the first half of 4 blocks of 2M rows = the x and y blocks (both time layers), never z.  Shown on
    the literal (N+1)-D restatement with arbitrary positive capacities: the moving blocks equal the operators built from the
    x and y capacities of the first layer alone, and differ from the full 3-D ones."""
    rng = np.random.default_rng(4)
    mesh = po.Mesh((3, 3, 3), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    st = ost.SpaceTimeMesh(mesh, [0.0, 0.1])
    M = int(np.prod(mesh.ext))
    M2 = 2 * M
    pos = lambda: rng.uniform(0.5, 1.5, M2)
    cap = po.Capacity(tuple(pos() for _ in range(4)), tuple(pos() for _ in range(4)), pos(), tuple(pos() for _ in range(4)),
                      rng.random((M2, 4)), rng.random((M2, 4)), pos(), -np.ones(M2), st, None)
    op = po.make_diffusion_ops(cap)
    assert op.G.shape == (4 * M2, M2)
    A = ost.A_mono_unstead_diff_moving(op, cap, 1.0, po.Robin(0.7, 1.3, 0.0), "BE")
    assert A.shape == (2 * M, 2 * M)

    def blocks(dims):
        lay = po.Capacity(tuple(cap.A[d][:M] for d in dims), tuple(cap.B[d][:M] for d in dims), cap.V[:M],
                          tuple(cap.W[d][:M] for d in dims), cap.C_w[:M, :3], cap.C_g[:M, :3], cap.G[:M], cap.cell_types[:M], mesh, None)
        # Kronecker operators of the chosen space directions on the 3-D mesh
        D_m = [po.build_differential_operator(po.delta_m, mesh, d) for d in dims]
        G = sp.vstack([D_m[i] @ sp.diags(lay.B[i]) for i in range(len(dims))], format="csr")
        H = sp.vstack([sp.diags(lay.A[i]) @ D_m[i] - D_m[i] @ sp.diags(lay.B[i]) for i in range(len(dims))], format="csr")
        w = np.concatenate([lay.W[i] for i in range(len(dims))])
        Wi = sp.diags(1.0 / w)
        Vn_1, Vn = cap.A[3][:M], cap.A[3][M:]
        return sp.bmat([[sp.diags(Vn_1) + G.T @ Wi @ G, -(sp.diags(Vn_1) - sp.diags(Vn)) + G.T @ Wi @ H],
                        [1.3 * (H.T @ Wi @ G), 1.3 * (H.T @ Wi @ H) + 0.7 * sp.diags(lay.G)]], format="csr")

    xy, xyz = blocks((0, 1)), blocks((0, 1, 2))
    assert abs(A - xy).max() <= 1e-13 * abs(xy).max()
    assert abs(A - xyz).max() > 1e-2 * abs(xyz).max()
