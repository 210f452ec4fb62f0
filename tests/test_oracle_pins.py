"""Pin the ORACLE on the reference's own golden vectors and known answers (CPU, no GPU).

Every expected value below is data taken from the reference's tests / examples (cited per test);
tests/golden/reference_pins.json holds the same vectors as a committed fixture.
"""
import json
import math
from pathlib import Path

import numpy as np
import pytest

from oracle import penguin_oracle as po
from oracle.geometry import Ball

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_pins.json").read_text())


# ---------------------------------------------------------------- test/mesh_test.jl:4-55
@pytest.mark.parametrize("N", [1, 2, 3])
def test_mesh_centers_and_borders(N):
    g = GOLD["mesh"][str(N)]
    mesh = po.Mesh((5,) * N, (1.0,) * N, (0.0,) * N)
    for d in range(N):
        assert mesh.centers[d].tolist() == g["centers"]          # exact ==, incl. 0.6000000000000001
    assert mesh.nC() == g["nC"]
    assert len(mesh.border_cells) == g["n_border"]               # 2 / 16 / 98
    assert [list(mesh.border_cells[0][0]), list(mesh.border_cells[0][1])] == g["border0"]
    assert [list(mesh.border_cells[1][0]), list(mesh.border_cells[1][1])] == g["border1"]


def test_mesh_4d_border_count_formula():
    # test/mesh_test.jl:52: 544 border cells for 5^4 = 5^4 - 3^4
    assert 5 ** 4 - 3 ** 4 == GOLD["mesh"]["4"]["n_border"]


# ---------------------------------------------------------------- test/mesh_test.jl:57-80
def test_mesh_nodes_half_cell_shift():
    mesh = po.Mesh((5,), (1.0,), (0.0,))
    assert mesh.nodes[0].tolist() == GOLD["mesh"]["nodes_1d"]    # [0.1, 0.30000000000000004, ...]


# ---------------------------------------------------------------- test/operators_test.jl:4-56
def test_gradient_divergence_of_constants():
    mesh = po.Mesh((20, 20), (1.0, 1.0), (0.0, 0.0))
    cap = po.make_capacity(Ball((0.0, 0.0), 0.5), mesh)
    op = po.make_diffusion_ops(cap)
    n2 = 2 * 21 * 21
    assert po.grad(op, np.ones(n2))[1] == 0.0                    # grad[2] == 0.0
    assert po.div(op, np.ones(n2), np.ones(n2))[1] == 0.0        # div[2] == 0.0


@pytest.mark.parametrize("N,n", [(1, 20), (2, 20), (3, 8)])
def test_operator_sizes(N, n):
    mesh = po.Mesh((n,) * N, (1.0,) * N, (0.0,) * N)
    cap = po.make_capacity(Ball((0.0,) * N, 0.5), mesh)
    op = po.make_diffusion_ops(cap)
    M = (n + 1) ** N
    assert (op.G.T @ op.Winv @ op.G).shape == (M, M)
    assert op.size == (n + 1,) * N
    assert op.G.shape == (N * M, M) and op.H.shape == (N * M, M)


def test_delta_m_last_row_quirk():
    # src/operators.jl:9: D[n,n] = 0  =>  (D p)[n] = -p[n-1]
    D = po.delta_m(5).toarray()
    assert D[4, 4] == 0.0 and D[4, 3] == -1.0 and D[0, 0] == 1.0


# ---------------------------------------------------------------- test/capacity_test.jl
def test_capacity_circle_known_measures():
    # :6-84  circle r = 0.3 on 20^2: sum V ~ pi r^2, sum Gamma ~ 2 pi r (reference: rtol 0.05 / 0.1)
    mesh = po.Mesh((20, 20), (1.0, 1.0), (0.0, 0.0))
    cap = po.make_capacity(Ball((0.5, 0.5), 0.3), mesh)
    assert cap.V.sum() == pytest.approx(math.pi * 0.09, rel=1e-12)
    assert cap.G.sum() == pytest.approx(2 * math.pi * 0.3, rel=1e-12)
    cut = np.flatnonzero(cap.cell_types == -1)
    d = np.linalg.norm(cap.C_g[cut] - 0.5, axis=1)
    assert np.all(np.abs(d - 0.3) < 0.05)                        # :81 interface centroids on the circle


def test_capacity_cut_set_equals_gamma_support():
    # :228-258 circle r=0.3 @ (0.51,0.51) on 30^2: sort(findall(cell_types .== -1)) == sort(findall(diag(Gamma) .> 0))
    mesh = po.Mesh((30, 30), (1.0, 1.0), (0.0, 0.0))
    cap = po.make_capacity(Ball((0.51, 0.51), 0.3), mesh)
    assert np.array_equal(np.flatnonzero(cap.cell_types == -1), np.flatnonzero(cap.G > 0))
    cut = np.flatnonzero(cap.cell_types == -1)
    assert len(cut) > 0
    d = np.linalg.norm(cap.C_g[cut] - 0.51, axis=1)
    assert np.all(np.abs(d - 0.3) < 0.05)


def test_capacity_1d_two_interfaces():
    # :192-226 |x - 0.5| - 0.3 on 20 cells: exactly 2 cells with Gamma > 0, centroids near 0.2 / 0.8
    mesh = po.Mesh((20,), (1.0,), (0.0,))
    cap = po.make_capacity(Ball((0.5,), 0.3), mesh)
    has = np.flatnonzero(cap.G > 0)
    assert len(has) == 2
    cg = cap.C_g[has, 0]
    assert any(abs(c - 0.2) < 0.05 for c in cg) and any(abs(c - 0.8) < 0.05 for c in cg)
    cap2 = po.make_capacity(Ball((0.5,), 0.3), mesh, compute_centroids=False)
    assert cap2.C_g.shape[0] == 0                                 # isempty(C_γ)


def test_capacity_sphere_known_measures():
    # :86-145 sphere r = 0.3 on 10^3
    mesh = po.Mesh((10, 10, 10), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    cap = po.make_capacity(Ball((0.5, 0.5, 0.5), 0.3), mesh)
    assert cap.V.sum() == pytest.approx(4 / 3 * math.pi * 0.027, rel=1e-11)
    assert cap.G.sum() == pytest.approx(4 * math.pi * 0.09, rel=1e-11)
    for d in range(3):
        assert cap.W[d].sum() == pytest.approx(cap.V.sum(), rel=1e-11)   # staggered volumes tile the fluid
    assert np.array_equal(np.flatnonzero(cap.cell_types == -1), np.flatnonzero(cap.G > 0))


# ---------------------------------------------------------------- test/solver/diffusion_test.jl:57-80
def _heat_mono(n, method):
    mesh = po.Mesh((n, n), (4.0, 4.0), (0.0, 0.0))
    cap = po.make_capacity(Ball((2.0, 2.0), 1.0), mesh)
    op = po.make_diffusion_ops(cap)
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
    bcb = po.BorderConditions({k: po.Dirichlet(0.0) for k in ("left", "right", "top", "bottom")})
    M = (n + 1) ** 2
    u0 = np.concatenate([np.zeros(M), np.ones(M)])
    dt = 0.25 * (4.0 / n) ** 2
    s = po.DiffusionUnsteadyMono(ph, bcb, po.Dirichlet(1.0), dt, u0, "BE")
    po.solve_DiffusionUnsteadyMono(s, ph, dt, 0.01, bcb, po.Dirichlet(1.0), "BE", method=method, **({} if method == "\\" else {"reltol": 1e-13}))
    return s, M


def test_heat_monophasic_known_answer():
    s, M = _heat_mono(20, "\\")
    assert s.x[M:].max() == pytest.approx(1.0, abs=1e-2)          # maximum(ug) ≈ 1.0 atol=1e-2
    assert len(s.states) == 2
    s2, _ = _heat_mono(20, "bicgstab")
    assert np.linalg.norm(s2.x - s.x) <= 1e-9 * np.linalg.norm(s.x)


# ---------------------------------------------------------------- test/convergence_test.jl:72-98
def test_unsteady_mono_1d_zero_stays_zero():
    nx, lx = 40, 1.0
    mesh = po.Mesh((nx,), (lx,), (0.0,))
    cap = po.make_capacity(Ball((0.5,), 0.25), mesh)
    op = po.make_diffusion_ops(cap)
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
    bcb = po.BorderConditions({"top": po.Dirichlet(0.0), "bottom": po.Dirichlet(0.0)})
    dt = 0.5 * (lx / nx) ** 2
    s = po.DiffusionUnsteadyMono(ph, bcb, po.Dirichlet(0.0), dt, np.zeros(2 * (nx + 1)), "BE")
    po.solve_DiffusionUnsteadyMono(s, ph, dt, 5 * dt, bcb, po.Dirichlet(0.0), "BE", method="bicgstab")
    assert np.max(np.abs(s.x)) < 1e-8


# ---------------------------------------------------------------- test/convergence_test.jl:30-49
def test_poisson_2d_known_error_bound():
    n = 40
    mesh = po.Mesh((n, n), (4.0, 4.0), (0.0, 0.0))
    c = (2.01, 2.01)
    cap = po.make_capacity(Ball(c, 1.0), mesh)
    op = po.make_diffusion_ops(cap)
    ph = po.Phase(cap, op, lambda x, y, z=0.0: 4.0, lambda x, y, z=0.0: 1.0)
    bcb = po.BorderConditions({k: po.Dirichlet(0.0) for k in ("left", "right", "top", "bottom")})
    s = po.DiffusionSteadyMono(ph, bcb, po.Dirichlet(0.0))
    po.solve_system(s, method="\\")
    u = lambda x, y: 1.0 - (x - c[0]) ** 2 - (y - c[1]) ** 2
    _, _, global_err, *_ = po.check_convergence(u, s, cap, 2)
    assert global_err < 1e-2                                       # reference threshold


# ---------------------------------------------------------------- examples/2D/Diffusion/Heat.jl:63-100
def test_config1_disc_cooling_against_bessel_series():
    """80x80 is config 1; the analytic series is the reference's own (radial_heat_xy)."""
    from scipy.special import j0, j1, jn_zeros

    n = 40  # coarser than config 1 to keep the CPU suite short; error scales as h^2
    mesh = po.Mesh((n, n), (4.0, 4.0), (0.0, 0.0))
    c = (2.01, 2.01)
    cap = po.make_capacity(Ball(c, 1.0), mesh)
    op = po.make_diffusion_ops(cap)
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
    bcb = po.BorderConditions({k: po.Dirichlet(0.0) for k in ("left", "right", "top", "bottom")})
    M = (n + 1) ** 2
    u0 = np.concatenate([np.zeros(M), np.ones(M)])
    dt = 0.25 * (4.0 / n) ** 2
    Tend = 0.1
    s = po.DiffusionUnsteadyMono(ph, bcb, po.Dirichlet(1.0), dt, u0, "BE")
    po.solve_DiffusionUnsteadyMono(s, ph, dt, Tend, bcb, po.Dirichlet(1.0), "BE", method="\\")
    t_num = dt * len(s.states)        # states[k] is the solution after k+1 implicit steps
    al = jn_zeros(0, 200)

    def u_ana(x, y):
        r = math.hypot(x - c[0], y - c[1])
        if r >= 1.0:
            return 0.0
        return 1.0 - 2.0 * float(np.sum(np.exp(-al ** 2 * t_num) * j0(al * r) / (al * j1(al))))

    _, _, global_err, full_err, cut_err, _ = po.check_convergence(u_ana, s, cap, 2)
    assert global_err < 1e-2


# ---------------------------------------------------------------- test/solver/diffusion_test.jl:5-56 (steady drivers)
def test_steady_monophasic_known_answer():
    n = 20
    mesh = po.Mesh((n, n), (2.0, 2.0), (0.0, 0.0))
    cap = po.make_capacity(Ball((0.5, 0.5), 0.5), mesh)
    op = po.make_diffusion_ops(cap)
    ph = po.Phase(cap, op, lambda x, y, z=0.0: 0.0, lambda x, y, z=0.0: 1.0)
    bcb = po.BorderConditions({k: po.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
    s = po.DiffusionSteadyMono(ph, bcb, po.Dirichlet(1.0))
    po.solve_DiffusionSteadyMono(s, method="\\")
    M = (n + 1) ** 2
    pin = GOLD["known_answers"]["steady_mono_20x20_max"]
    assert s.x[:M].max() == pytest.approx(pin["value"], abs=pin["atol"])
    assert s.x[M:].max() == pytest.approx(pin["value"], abs=pin["atol"])


def test_steady_diphasic_known_answer():
    n = 80
    mesh = po.Mesh((n, n), (4.0, 4.0), (0.0, 0.0))
    cap1 = po.make_capacity(Ball((2.0, 2.0), 1.0), mesh)
    cap2 = po.make_capacity(Ball((2.0, 2.0), 1.0, complement=True), mesh)
    one = lambda x, y, z=0.0: 1.0
    p1, p2 = po.Phase(cap1, po.make_diffusion_ops(cap1), one, one), po.Phase(cap2, po.make_diffusion_ops(cap2), one, one)
    bcb = po.BorderConditions({k: po.Dirichlet(0.0) for k in ("left", "right", "top", "bottom")})
    ic = po.InterfaceConditions(po.ScalarJump(1.0, 1.0, 0.0), po.FluxJump(1.0, 1.0, 0.0))
    s = po.DiffusionSteadyDiph(p1, p2, bcb, ic)
    po.solve_DiffusionSteadyDiph(s, method="\\")
    M = (n + 1) ** 2
    pin = GOLD["known_answers"]["steady_diph_80x80_max_u1o"]
    assert s.x[:M].max() == pytest.approx(pin["value"], abs=pin["atol"])


# ---------------------------------------------------------------- test/solver/darcy_test.jl (aliases of the diffusion drivers)
def _darcy_setup():
    n = 20
    mesh = po.Mesh((n, n), (2.0, 2.0), (0.0, 0.0))
    cap = po.make_capacity(Ball((0.5, 0.5), 0.5), mesh)
    op = po.make_diffusion_ops(cap)
    bcb = po.BorderConditions({"left": po.Dirichlet(10.0), "right": po.Dirichlet(20.0)})
    return n, mesh, cap, op, bcb


def test_darcy_known_answers():
    n, mesh, cap, op, bcb = _darcy_setup()
    ph = po.Phase(cap, op, lambda x, y, z=0.0: 0.0, lambda x, y, z=0.0: 1.0)
    s = po.DarcyFlow(ph, bcb, po.Neumann(0.0))
    po.solve_DarcyFlow(s, method="\\")
    M = (n + 1) ** 2
    pin = GOLD["known_answers"]["darcy_20x20_max_uo"]
    assert s.x[:M].max() == pytest.approx(pin["value"], abs=pin["atol"])
    u = po.solve_darcy_velocity(s, ph)
    assert np.nanmax(np.abs(u)) < GOLD["known_answers"]["darcy_velocity_max_abs"]["lt"]


def test_darcy_unsteady_known_answer():
    n, mesh, cap, op, bcb = _darcy_setup()
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z=0.0: 1.0)
    M = (n + 1) ** 2
    dt = 0.1 * (2.0 / n) ** 2
    s = po.DiffusionUnsteadyMono(ph, bcb, po.Neumann(0.0), dt, np.full(2 * M, 10.0), "BE")   # DarcyFlowUnsteady (:46-58)
    po.solve_DiffusionUnsteadyMono(s, ph, dt, 0.2, bcb, po.Neumann(0.0), "BE", method="\\")
    pin = GOLD["known_answers"]["darcy_unsteady_20x20_max_uo"]
    assert s.states[-1][:M].max() == pytest.approx(pin["value"], abs=pin["atol"])


# ---------------------------------------------------------------- advection-diffusion (no reference test: analytic known answer)
def test_advection_diffusion_1d_exponential_profile():
    """Steady u T' = T'' on the whole line segment, T = 0 / 1 at the ends: T = (exp(Pe x) - 1) / (exp(Pe L) - 1) between the
    border cell centres.  Pins the oracle's ConvectionOps / A_mono_stead_advdiff restatement on an analytic solution
    (the reference ships no test for its advection-diffusion drivers)."""
    n, Pe = 80, 5.0
    mesh = po.Mesh((n,), (1.0,), (0.0,))
    cap = po.make_capacity(Ball((10.0,), 20.0), mesh)                # the whole domain is fluid
    M = n + 1
    op = po.make_convection_ops(cap, [np.full(M, Pe)], np.zeros(M))
    ph = po.Phase(cap, op, lambda x, y=0.0, z=0.0: 0.0, lambda x, y=0.0, z=0.0: 1.0)
    bcb = po.BorderConditions({"bottom": po.Dirichlet(0.0), "top": po.Dirichlet(1.0)})
    s = po.AdvectionDiffusionSteadyMono(ph, bcb, po.Dirichlet(0.0))
    po.solve_system(s, method="\\")
    xc = np.array(mesh.centers[0])
    exact = (np.exp(Pe * (xc - xc[0])) - 1.0) / (np.exp(Pe * (xc[-1] - xc[0])) - 1.0)
    assert np.abs(s.x[:n] - exact).max() < 5e-4


def test_gmres_restatement_against_direct_solve_and_scipy():
    """gmres_ref (IterativeSolvers.gmres semantics; the package is not vendored) on a cut-cell heat system: same
    solution as the direct solve and as scipy's GMRES, for the default restart and a short one; the reference's own
    pin for this path is test/solver_test.jl:71-72 (bicgstabl vs gmres solutions agree to 1e-2)."""
    import scipy.sparse.linalg as spla

    n = 16
    mesh = po.Mesh((n, n), (4.0, 4.0), (0.0, 0.0))
    cap = po.make_capacity(Ball((2.01, 2.01), 1.0), mesh)
    op = po.make_diffusion_ops(cap)
    M = (n + 1) ** 2
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
    bcb = po.BorderConditions({k: po.Dirichlet(0.0) for k in ("left", "right", "top", "bottom")})
    s = po.DiffusionUnsteadyMono(ph, bcb, po.Dirichlet(1.0), 0.25 * (4.0 / n) ** 2, np.concatenate([np.zeros(M), np.ones(M)]), "BE")
    A, b, _ = po.remove_zero_rows_cols(s.A, s.b)
    x_direct = spla.spsolve(A.tocsc(), b)
    for restart in (20, 5):
        x, it, res = po.gmres_ref(A, b, reltol=1e-13, restart=restart)
        assert res <= 1e-13 * np.linalg.norm(b)
        assert np.linalg.norm(x - x_direct) <= 1e-10 * np.linalg.norm(x_direct)
        xs, info = spla.gmres(A, b, rtol=1e-13, atol=0.0, restart=restart, maxiter=1000)
        assert info == 0 and np.linalg.norm(x - xs) <= 1e-9 * np.linalg.norm(xs)
    _, it20, _ = po.gmres_ref(A, b, reltol=1e-13, restart=20)
    _, it5, _ = po.gmres_ref(A, b, reltol=1e-13, restart=5)
    assert it5 >= it20                       # restarting can only slow GMRES down
    x3, it3, res3 = po.gmres_ref(A, b, reltol=1e-13, maxiter=3)
    assert it3 == 3 and res3 > 1e-13 * np.linalg.norm(b)
    # through solve_system (method="gmres"), the way the reference's default reaches it (src/solver.jl:158)
    po.solve_system(s, method="gmres", reltol=1e-13)
    xg = s.x.copy()
    po.solve_system(s, method="\\")
    assert np.linalg.norm(xg - s.x) <= 1e-10 * np.linalg.norm(s.x)


def test_unsteady_diphasic_advection_diffusion_reduces_to_diffusion_at_zero_velocity():
    """The reference ships no test for AdvectionDiffusionUnsteadyDiph (src/solver/advectiondiffusion.jl:299-418); the
    restatement is tied to the pinned diffusion restatement instead: with u = 0 its matrix equals
    A_diph_unstead_diff for both schemes and its BE right-hand side equals b_diph_unstead_diff; its CN right-hand
    side differs from the diffusion one by exactly the diffusion term the reference leaves out (:375-377)."""
    n = 12
    mesh = po.Mesh((n, n), (4.0, 4.0), (0.0, 0.0))
    c1 = po.make_capacity(Ball((2.0, 2.0), 1.0), mesh)
    c2 = po.make_capacity(Ball((2.0, 2.0), 1.0, complement=True), mesh)
    M = (n + 1) ** 2
    zero = [np.zeros(M), np.zeros(M)]
    f = lambda x, y, z, t: 1.0 + 0.1 * x
    D = lambda x, y, z: 0.7
    q = [po.Phase(c, po.make_convection_ops(c, zero, np.zeros(2 * M)), f, D) for c in (c1, c2)]
    d = [po.Phase(c, po.make_diffusion_ops(c), f, D) for c in (c1, c2)]
    ic = po.InterfaceConditions(po.ScalarJump(1.0, 0.8, 0.1), po.FluxJump(1.0, 2.0, 0.0))
    dt = 0.01
    rng = np.random.default_rng(3)
    Ti = rng.uniform(0.0, 1.0, 4 * M)
    for sch in ("BE", "CN"):
        Aa = po.A_diph_unstead_advdiff(q[0].operator, q[1].operator, c1, c2, D, D, ic, dt, sch)
        Ad = po.A_diph_unstead_diff(d[0].operator, d[1].operator, c1, c2, D, D, ic, dt, sch)
        assert abs(Aa - Ad).max() <= 1e-15 * abs(Ad).max()
    args = lambda ph: (ph[0].operator, ph[1].operator, f, f, c1, c2, D, D, ic, Ti, dt, 0.3)
    assert np.allclose(po.b_diph_unstead_advdiff(*args(q), "BE"), po.b_diph_unstead_diff(*args(d), "BE"), rtol=0, atol=1e-15)
    ba, bd = po.b_diph_unstead_advdiff(*args(q), "CN"), po.b_diph_unstead_diff(*args(d), "CN")
    L1, M1, _, _ = po._blocks(d[0].operator)
    missing = -dt / 2 * 0.7 * (L1 @ Ti[:M] + M1 @ Ti[M:2 * M])
    assert np.allclose(bd[:M] - ba[:M], missing, rtol=0, atol=1e-13)
    with pytest.raises(ValueError):
        po.A_diph_unstead_advdiff(q[0].operator, q[1].operator, c1, c2, D, D, ic, dt, "RK4")


def test_polynomial_right_preconditioner_divides_bicgstab_iterations():
    """What pg_krylov.hip relies on, checked on the oracle's system of a 3-D CN step (benchmark/Heat3D.jl shape at 20^3):
    with the point-equilibrated Â and the Chebyshev residual polynomial R of degree m on the Gershgorin interval,
    BiCGStab on C = I - R(Â) reaches the same solution in about 1/m of the iterations (about the same number of products
    with Â) -- m = 2 with both roots at 1 is round 1's Neumann preconditioner 2I - Â -- and the Gershgorin radius that
    admits it (columns of identity rows left out, rows taken in the D⁻¹A scaling) is below 0.95 although the plain row
    sums are not."""
    import scipy.sparse as sp

    n = 20
    mesh = po.Mesh((n, n, n), (4.0, 4.0, 4.0), (0.0, 0.0, 0.0))
    cap = po.make_capacity(Ball((2.01, 2.01, 2.01), 1.0), mesh)
    op = po.make_diffusion_ops(cap)
    M = (n + 1) ** 3
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
    bcb = po.BorderConditions({k: po.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
    dt = 0.75 * (4.0 / n) ** 2
    A = po.A_mono_unstead_diff(op, cap, ph.Diffusion_coeff, po.Dirichlet(1.0), dt, "CN")
    b = po.b_mono_unstead_diff(op, ph.source, ph.Diffusion_coeff, cap, po.Dirichlet(1.0), np.zeros(2 * M), dt, dt, "CN")
    A, b = po.BC_border_mono(A, b, bcb, mesh, t=dt)
    Ar, br, _ = po.remove_zero_rows_cols(A, b)
    d = 1.0 / np.sqrt(np.abs(Ar.diagonal()))
    Ah = (sp.diags(d) @ Ar @ sp.diags(d)).tocsr()
    bh = d * br
    rl = np.diff(Ar.indptr)
    ident = rl == 1
    DA = (sp.diags(1.0 / np.abs(Ar.diagonal())) @ Ar).tocsr()
    off = np.asarray(abs(DA @ sp.diags((~ident).astype(float))).sum(axis=1)).ravel() - np.abs(DA.diagonal())
    g = off[~ident].max()

    def chain(taus):
        class Pre:                               # v -> v - Π_k (I - τ_k Â) v: only `@` is asked of it by bicgstab_ref
            shape = Ah.shape

            def __matmul__(self, v):
                w = v
                for tk in taus:
                    w = w - tk * (Ah @ w)
                return v - w

        def recover(y):                          # x = q(Â) y = Σ_k τ_k w_(k-1)
            u, w = np.zeros_like(y), y
            for k, tk in enumerate(taus):
                u = u + tk * w
                if k + 1 < len(taus):
                    w = w - tk * (Ah @ w)
            return u

        return Pre(), recover

    x_plain, it_plain, _ = po.bicgstab_ref(Ah, bh, reltol=1e-12)
    pre, recover = chain([1.0, 1.0])             # Neumann: C = Â(2I - Â) = I - (I - Â)²
    y, it_pre, _ = po.bicgstab_ref(pre, bh, reltol=1e-12)
    assert np.linalg.norm(recover(y) - x_plain) <= 1e-10 * np.linalg.norm(x_plain)
    assert np.allclose(recover(y), 2.0 * y - Ah @ y, rtol=1e-13, atol=1e-15)
    assert it_pre <= it_plain // 2 + 1, (it_pre, it_plain)
    for m in (3, 4, 6):
        lam = sorted(1.0 + g * math.cos(math.pi * (2 * k + 1) / (2 * m)) for k in range(m))
        pre, recover = chain([1.0 / l for l in lam])
        y, it_m, _ = po.bicgstab_ref(pre, bh, reltol=1e-12)
        assert np.linalg.norm(recover(y) - x_plain) <= 1e-10 * np.linalg.norm(x_plain)
        assert it_m <= it_plain // m + 2, (m, it_m, it_plain)
    # the admissibility test of pg_precond.hip (k_gershgorin)
    off_all = np.asarray(abs(DA).sum(axis=1)).ravel() - np.abs(DA.diagonal())
    assert g < 0.95 <= off_all.max()


def test_c_krylov_port_matches_the_numpy_restatement():
    """oracle/krylov_ref.c (what bench.py times on the host cores) against oracle/penguin_oracle.py on a cut-cell system:
    plain BiCGStab takes the same number of iterations as bicgstab_ref, the polynomially preconditioned variant (the
    iteration of pg_krylov.hip: product form, weighted test after both halves, optional warm start) about 1/m of them and
    the same solution; OpenMP threads do not change the iteration count."""
    import scipy.sparse as sp

    from oracle import krylov_c

    n = 16
    mesh = po.Mesh((n, n, n), (4.0, 4.0, 4.0), (0.0, 0.0, 0.0))
    cap = po.make_capacity(Ball((2.01, 2.01, 2.01), 1.0), mesh)
    op = po.make_diffusion_ops(cap)
    M = (n + 1) ** 3
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
    bcb = po.BorderConditions({k: po.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
    s = po.DiffusionUnsteadyMono(ph, bcb, po.Dirichlet(1.0), 0.75 * (4.0 / n) ** 2, np.zeros(2 * M), "CN")
    Ar, br, _ = po.remove_zero_rows_cols(s.A, s.b)
    d = 1.0 / np.sqrt(np.abs(Ar.diagonal()))
    Ah = (sp.diags(d) @ Ar @ sp.diags(d)).tocsr()
    bh = d * br
    x_ref, it_ref, _ = po.bicgstab_ref(Ah, bh, reltol=1e-12)
    x_c, it_c, res_c = krylov_c.solve(Ah, bh, "bicgstab", reltol=1e-12)
    assert it_c == it_ref and np.linalg.norm(x_c - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
    x_0, it_0, _, nmv_0 = krylov_c.solve_poly(Ah, bh, 0, 0.0, reltol=1e-12)          # plain, through the same routine
    assert abs(it_0 - it_ref) <= 1 and nmv_0 <= 2 * it_0 and np.linalg.norm(x_0 - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
    for m in (2, 4, 6):
        x_n, it_n, res_n, nmv = krylov_c.solve_poly(Ah, bh, m, 0.72, reltol=1e-12)
        assert it_n <= it_ref // m + 2 and res_n <= 1e-12 * np.linalg.norm(bh), (m, it_n, it_ref)
        assert nmv <= 2 * m * it_n + (m - 1)
        assert np.linalg.norm(x_n - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
        _, it_n4, _, _ = krylov_c.solve_poly(Ah, bh, m, 0.72, reltol=1e-12, nthreads=4)
        assert abs(it_n4 - it_n) <= 1
    # weighted test and warm start: a tighter criterion in the units of x, fewer iterations from a good guess
    x_w, it_w, _, _ = krylov_c.solve_poly(Ah, bh, 4, 0.72, weights=d, reltol=1e-12)
    assert np.linalg.norm(d * (x_w - x_ref)) <= 1e-11 * np.linalg.norm(d * x_ref)
    x_s, it_s, _, _ = krylov_c.solve_poly(Ah, bh, 4, 0.72, x0=x_ref * (1.0 + 1e-6), weights=d, reltol=1e-12)
    assert it_s < it_w and np.linalg.norm(d * (x_s - x_ref)) <= 1e-11 * np.linalg.norm(d * x_ref)
    y = krylov_c.spmv(Ah, bh, nthreads=2)
    assert np.allclose(y, Ah @ bh, rtol=1e-14, atol=1e-14 * np.abs(bh).max())
