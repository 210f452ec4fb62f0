"""Second randomised sweep, along the axes tests/test_gpu_fuzz.py leaves fixed: unions of balls (the weak-scaling body),
time-dependent closures for source / interface / border data, periodic borders, steady solves, and diphasic problems on
lattice-aligned (degenerate) geometry.  HIP path against the oracle; deterministic seeds."""
import numpy as np
import pytest

from oracle import penguin_oracle as po
from oracle.geometry import Ball, MultiBall
from tests.common import oracle_capacity_from_product, rel_l2

pytestmark = pytest.mark.gpu

KEYS = {1: ("bottom", "top"), 2: ("left", "right", "top", "bottom"), 3: ("left", "right", "top", "bottom", "forward", "backward")}


def _cond_tol(Ar, nstates):
    cond = np.linalg.cond(Ar.toarray(), 1)
    return max(1e-10, 10.0 * cond * np.finfo(float).eps * nstates), cond


@pytest.mark.parametrize("seed", range(10))
def test_random_union_of_balls(pj, seed):
    """MultiSphere (f = min_s f_s over pairwise disjoint balls, >= 3 cells apart): capacities against the oracle's MultiBall
    (classification bit-exact), then BE + CN steps on the product's capacities."""
    rng = np.random.default_rng(9100 + seed)
    N = int(rng.integers(2, 4))
    nb = int(rng.integers(2, 4))
    n = tuple(int(v) for v in rng.integers(10, 15 if N == 3 else 33, size=N))
    L = tuple(float(v) for v in rng.uniform(2.0, 4.0, size=N))
    h = max(L[d] / n[d] for d in range(N))
    r = float(rng.uniform(0.12, 0.2)) * min(L)
    centers = []
    for _ in range(200):
        c = tuple(float(rng.uniform(0.1, 0.9)) * L[d] for d in range(N))
        if all(np.linalg.norm(np.subtract(c, o)) >= 2 * r + 3.5 * h * np.sqrt(N) for o in centers):
            centers.append(c)
        if len(centers) == nb:
            break
    if len(centers) < 2:
        pytest.skip("could not place two disjoint balls")
    mesh, omesh = pj.Mesh(n, L), po.Mesh(n, L, (0.0,) * N)
    cap = pj.Capacity(pj.MultiSphere(centers, r), mesh)
    ocap = po.make_capacity(MultiBall(centers, r), omesh)
    info = (N, n, L, centers, r)
    assert np.array_equal(cap.cell_types, ocap.cell_types), info
    hh = min(L[d] / n[d] for d in range(N))
    assert np.max(np.abs(cap.V - ocap.V)) <= 1e-10 * h ** N, info
    assert np.max(np.abs(cap.Γ - ocap.G)) <= 1e-10 * h ** (N - 1), info
    for d in range(N):
        assert np.max(np.abs(cap.A[d] - ocap.A[d])) <= 1e-10 * h ** (N - 1), info
        assert np.max(np.abs(cap.B[d] - ocap.B[d])) <= 1e-7 * h ** (N - 1), info
        assert np.max(np.abs(cap.W[d] - ocap.W[d])) <= 1e-7 * h ** N, info
    ocap2 = oracle_capacity_from_product(cap, omesh)
    one = lambda *a: 1.0
    ph, oph = pj.Phase(cap, pj.DiffusionOps(cap), one, one), po.Phase(ocap2, po.make_diffusion_ops(ocap2), one, one)
    borders = {k: float(rng.uniform(0.0, 1.0)) for k in KEYS[N]}
    bcb = pj.BorderConditions({k: pj.Dirichlet(v) for k, v in borders.items()})
    obcb = po.BorderConditions({k: po.Dirichlet(v) for k, v in borders.items()})
    M = int(np.prod([v + 1 for v in n]))
    dt = 0.4 * hh ** 2
    u0 = rng.uniform(0.0, 1.0, 2 * M)
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, u0, "BE")
    so = po.DiffusionUnsteadyMono(oph, obcb, po.Dirichlet(1.0), dt, u0, "BE")
    _, _, idx = s.system(0)
    Ar, _, oidx = po.remove_zero_rows_cols(so.A, so.b)
    assert np.array_equal(idx, oidx), info
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 3 * dt, bcb, pj.Dirichlet(1.0), "CN", reltol=1e-14)
    po.solve_DiffusionUnsteadyMono(so, oph, dt, 3 * dt, obcb, po.Dirichlet(1.0), "CN", method="\\")
    tol, cond = _cond_tol(Ar, len(so.states))
    for a, b in zip(s.states, so.states):
        assert rel_l2(a, b) <= tol, (rel_l2(a, b), tol, cond, info)


@pytest.mark.parametrize("seed", range(10))
def test_random_time_dependent_data(pj, seed):
    """Closures for the source f(x,y,z,t), the interface value g(x,y,z,t) and the border values v(x,y,t) -- evaluated by the
    host layer at the reference's points and times (C_ω, C_γ, mesh.centers; t + Δt with t already advanced) -- plus a
    variable diffusivity; periodic left/right on some 2-D draws."""
    rng = np.random.default_rng(9300 + seed)
    N = int(rng.integers(1, 4))
    n = tuple(int(v) for v in rng.integers(8, 13 if N == 3 else 25, size=N))
    L = tuple(float(v) for v in rng.uniform(1.0, 2.5, size=N))
    c = tuple(L[d] * float(rng.uniform(0.35, 0.65)) for d in range(N))
    r = float(rng.uniform(0.2, 0.4)) * min(L)
    a1, a2, a3, w = (float(v) for v in rng.uniform(0.2, 1.5, size=4))
    f = lambda x, y, z, t: a1 * np.cos(w * t) + a2 * x - 0.3 * y * t + 0.1 * z
    g = lambda x, y, z, t: a3 + 0.5 * np.sin(w * t) * x + 0.2 * y
    D = lambda x, y, z: 1.0 + 0.3 * np.cos(x + y) ** 2
    kind = ["dirichlet", "robin", "neumann"][int(rng.integers(0, 3))]
    mk = {"dirichlet": lambda m: m.Dirichlet(g), "neumann": lambda m: m.Neumann(g), "robin": lambda m: m.Robin(1.0, 0.6, g)}[kind]
    bval = {1: (lambda x, t: 0.4 + 0.2 * t + 0.1 * x), 2: (lambda x, y, t: 0.4 + 0.2 * t + 0.1 * x * y),
            3: (lambda x, y, z, t: 0.4 + 0.2 * t + 0.1 * x * y - 0.05 * z)}[N]
    periodic = N == 2 and bool(rng.integers(0, 2))
    def borders(m):
        out = {}
        for k in KEYS[N]:
            if periodic and k in ("left", "right"):
                out[k] = m.Periodic()
            elif rng_keys[k]:
                out[k] = m.Dirichlet(bval)
        return out
    rng_keys = {k: bool(rng.random() < 0.8) for k in KEYS[N]}
    if kind == "neumann" and not any(rng_keys.values()):
        rng_keys[KEYS[N][-1]] = True
    mesh, omesh = pj.Mesh(n, L), po.Mesh(n, L, (0.0,) * N)
    cap = pj.Capacity(pj.Sphere(c, r), mesh)
    ocap = oracle_capacity_from_product(cap, omesh)
    ph, oph = pj.Phase(cap, pj.DiffusionOps(cap), f, D), po.Phase(ocap, po.make_diffusion_ops(ocap), f, D)
    bcb, obcb = pj.BorderConditions(borders(pj)), po.BorderConditions(borders(po))
    M = int(np.prod([v + 1 for v in n]))
    dt = 0.4 * min(L[d] / n[d] for d in range(N)) ** 2
    u0 = rng.uniform(0.0, 1.0, 2 * M)
    sch0, sch = [("BE", "BE"), ("BE", "CN"), ("CN", "CN"), ("CN", "BE")][int(rng.integers(0, 4))]
    s = pj.DiffusionUnsteadyMono(ph, bcb, mk(pj), dt, u0, sch0)
    so = po.DiffusionUnsteadyMono(oph, obcb, mk(po), dt, u0, sch0)
    info = (seed, N, n, kind, periodic, sch0, sch, rng_keys)
    A, b, idx = s.system(0)
    Ar, br, oidx = po.remove_zero_rows_cols(so.A, so.b)
    assert np.array_equal(idx, oidx), info
    assert abs(A[:, :len(idx)] - Ar).max() <= 1e-12 * max(abs(Ar).max(), 1e-300), info
    assert np.max(np.abs(b - br)) <= 1e-11 * max(np.max(np.abs(br)), 1e-300), info
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 4 * dt, bcb, mk(pj), sch, reltol=1e-14)
    po.solve_DiffusionUnsteadyMono(so, oph, dt, 4 * dt, obcb, mk(po), sch, method="\\")
    assert len(s.states) == len(so.states)
    tol, cond = _cond_tol(Ar, len(so.states))
    for k, (a, bb) in enumerate(zip(s.states, so.states)):
        assert rel_l2(a, bb) <= tol, (k, rel_l2(a, bb), tol, cond, info)


@pytest.mark.parametrize("seed", range(8))
def test_random_steady_problem(pj, seed):
    """DiffusionSteadyMono on random geometry: Dirichlet / Robin interface, random Dirichlet borders, variable source."""
    rng = np.random.default_rng(9500 + seed)
    N = int(rng.integers(1, 4))
    n = tuple(int(v) for v in rng.integers(8, 13 if N == 3 else 25, size=N))
    L = tuple(float(v) for v in rng.uniform(1.0, 2.5, size=N))
    c = tuple(L[d] * float(rng.uniform(0.3, 0.7)) for d in range(N))
    r = float(rng.uniform(0.2, 0.45)) * min(L)
    comp = bool(rng.integers(0, 2))
    q = float(rng.uniform(-2.0, 2.0))
    f = lambda x, y=0.0, z=0.0, t=0.0: q * (1.0 + 0.5 * x - 0.2 * y)
    D = lambda x, y=0.0, z=0.0: 1.0 + 0.2 * x
    robin = bool(rng.integers(0, 2))
    mk = (lambda m: m.Robin(1.0, 0.7, 0.3)) if robin else (lambda m: m.Dirichlet(0.3))
    borders = {k: float(rng.uniform(-1.0, 1.0)) for k in KEYS[N] if comp or rng.random() < 0.5}
    mesh, omesh = pj.Mesh(n, L), po.Mesh(n, L, (0.0,) * N)
    cap = pj.Capacity(pj.Sphere(c, r, complement=comp), mesh)
    ocap = oracle_capacity_from_product(cap, omesh)
    ph, oph = pj.Phase(cap, pj.DiffusionOps(cap), f, D), po.Phase(ocap, po.make_diffusion_ops(ocap), f, D)
    bcb = pj.BorderConditions({k: pj.Dirichlet(v) for k, v in borders.items()})
    obcb = po.BorderConditions({k: po.Dirichlet(v) for k, v in borders.items()})
    s, so = pj.DiffusionSteadyMono(ph, bcb, mk(pj)), po.DiffusionSteadyMono(oph, obcb, mk(po))
    info = (seed, N, n, comp, robin, borders)
    A, b, idx = s.system(0)
    Ar, br, oidx = po.remove_zero_rows_cols(so.A, so.b)
    assert np.array_equal(idx, oidx), info
    assert abs(A[:, :len(idx)] - Ar).max() <= 1e-12 * max(abs(Ar).max(), 1e-300), info
    pj.solve_DiffusionSteadyMono_b(s, reltol=1e-14)
    po.solve_DiffusionSteadyMono(so, method="\\")
    tol, cond = _cond_tol(Ar, 1)
    if cond > 1e12:
        pytest.skip(f"numerically singular random system (cond_1 = {cond:.1e})")
    assert rel_l2(s.x, so.x) <= tol, (rel_l2(s.x, so.x), tol, cond, info)


@pytest.mark.parametrize("seed", range(8))
def test_diphasic_on_lattice_aligned_interface(pj, seed):
    """Diphasic heat with the interface through mesh nodes / tangent to faces (centre on the half-spacing lattice, radius
    a multiple of h/2): both phases' capacities against the oracle, they tile the box, and BE + CN states."""
    rng = np.random.default_rng(9700 + seed)
    N = int(rng.integers(2, 4))
    n = tuple(int(v) for v in rng.integers(6, 9 if N == 3 else 17, size=N))
    h = float(rng.choice([0.25, 0.5]))
    L = tuple(h * v for v in n)
    c = tuple(0.5 * h * float(rng.integers(n[d] - 2, n[d] + 3)) for d in range(N))
    r = 0.5 * h * float(rng.integers(2, max(3, min(n) - 2)))
    mesh, omesh = pj.Mesh(n, L), po.Mesh(n, L, (0.0,) * N)
    c1, c2 = pj.Capacity(pj.Sphere(c, r), mesh), pj.Capacity(pj.Sphere(c, r, complement=True), mesh)
    o1, o2 = po.make_capacity(Ball(c, r), omesh), po.make_capacity(Ball(c, r, complement=True), omesh)
    info = (seed, N, n, h, c, r)
    for cap, ocap in ((c1, o1), (c2, o2)):
        assert np.array_equal(cap.cell_types, ocap.cell_types), info
        assert np.max(np.abs(cap.V - ocap.V)) <= 1e-10 * h ** N, info
        for d in range(N):
            assert np.max(np.abs(cap.A[d] - ocap.A[d])) <= 1e-10 * h ** (N - 1), info
            assert np.max(np.abs(cap.W[d] - ocap.W[d])) <= 1e-7 * h ** N, info
    real = np.ones(tuple(v + 1 for v in n)[::-1], dtype=bool)
    for d in range(N):
        sl = [slice(None)] * N
        sl[N - 1 - d] = n[d]
        real[tuple(sl)] = False
    assert np.allclose((c1.V + c2.V)[real.ravel()], h ** N, rtol=0, atol=1e-12 * h ** N), info      # the phases tile every real cell
    assert np.allclose(c1.Γ, c2.Γ, rtol=0, atol=1e-11 * h ** (N - 1)), info                        # one interface, seen from both sides
    q1, q2 = oracle_capacity_from_product(c1, omesh), oracle_capacity_from_product(c2, omesh)
    one = lambda *a: 1.0
    p1, p2 = pj.Phase(c1, pj.DiffusionOps(c1), one, one), pj.Phase(c2, pj.DiffusionOps(c2), one, one)
    r1, r2 = po.Phase(q1, po.make_diffusion_ops(q1), one, one), po.Phase(q2, po.make_diffusion_ops(q2), one, one)
    ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 0.8, 0.1), pj.FluxJump(1.0, 1.5, 0.05))
    oic = po.InterfaceConditions(po.ScalarJump(1.0, 0.8, 0.1), po.FluxJump(1.0, 1.5, 0.05))
    borders = {k: 0.2 for k in KEYS[N]}
    bcb = pj.BorderConditions({k: pj.Dirichlet(v) for k, v in borders.items()})
    obcb = po.BorderConditions({k: po.Dirichlet(v) for k, v in borders.items()})
    M = int(np.prod([v + 1 for v in n]))
    u0 = rng.uniform(0.0, 1.0, 4 * M)
    dt = 0.4 * h ** 2
    s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, u0, "BE")
    so = po.DiffusionUnsteadyDiph(r1, r2, obcb, oic, dt, u0, "BE")
    _, _, idx = s.system(0)
    Ar, _, oidx = po.remove_zero_rows_cols(so.A, so.b)
    assert np.array_equal(idx, oidx), info
    pj.solve_DiffusionUnsteadyDiph_b(s, p1, p2, dt, 3 * dt, bcb, ic, "CN", reltol=1e-14)
    po.solve_DiffusionUnsteadyDiph(so, r1, r2, dt, 3 * dt, obcb, oic, "CN", method="\\")
    tol, cond = _cond_tol(Ar, len(so.states))
    if cond > 1e12:
        pytest.skip(f"numerically singular system (cond_1 = {cond:.1e})")
    for k, (a, b) in enumerate(zip(s.states, so.states)):
        assert rel_l2(a, b) <= tol, (k, rel_l2(a, b), tol, cond, info)
