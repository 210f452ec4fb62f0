"""Degenerate inputs the random sweep never draws: interfaces that pass EXACTLY through mesh nodes / faces / cell
corners (classification must still be bit-exact against the oracle, which uses the same +,-,*,compare arithmetic),
bodies smaller than a cell, one- and two-cell meshes.  The reference's tests use such round numbers freely
(test/capacity_test.jl: centre 0.5, radius 0.21 / 0.25 on n = 10..40 meshes of [0,1])."""
import numpy as np
import pytest

from oracle import penguin_oracle as po
from oracle.geometry import Ball
from tests.common import oracle_capacity_from_product, rel_l2

pytestmark = pytest.mark.gpu

CASES = [
    # N, n, L, x0, centre, radius, complement
    (1, (8,), (1.0,), (0.0,), (0.5,), 0.25, False),            # interface on a cell centre line: x = 0.25, 0.75 are centres
    (1, (8,), (1.0,), (0.0,), (0.5625,), 0.25, False),         # ... on nodes: nodes = (j + 1/2) h
    (2, (16, 16), (4.0, 4.0), (0.0, 0.0), (2.125, 2.125), 1.0, False),    # centre on a node, r = 4h: circle through nodes
    (2, (16, 16), (4.0, 4.0), (0.0, 0.0), (2.0, 2.0), 1.0, False),        # centre on a cell centre, r = 4h
    (2, (16, 16), (4.0, 4.0), (0.0, 0.0), (2.125, 2.0), 0.875, True),     # tangent to faces, complement
    (2, (10, 10), (1.0, 1.0), (0.0, 0.0), (0.5, 0.5), 0.25, False),       # test/capacity_test.jl numbers
    (3, (8, 8, 8), (4.0, 4.0, 4.0), (0.0, 0.0, 0.0), (2.25, 2.25, 2.25), 1.0, False),   # centre on a node, r = 2h
    (3, (8, 8, 8), (4.0, 4.0, 4.0), (0.0, 0.0, 0.0), (2.0, 2.0, 2.0), 1.5, True),
    (2, (12, 12), (3.0, 3.0), (-1.5, -1.5), (0.0, 0.0), 0.05, False),     # body far smaller than a cell (h = 0.25)
    (2, (12, 12), (3.0, 3.0), (-1.5, -1.5), (0.125, 0.125), 0.125, False),  # a disc inscribed in ONE cell
    (1, (1,), (1.0,), (0.0,), (0.7,), 0.2, False),             # one-cell mesh
    (2, (2, 2), (1.0, 1.0), (0.0, 0.0), (0.6, 0.55), 0.3, False),         # two-cell mesh
    (3, (1, 3, 2), (1.0, 3.0, 2.0), (0.0, 0.0, 0.0), (0.6, 1.4, 1.1), 0.45, False),   # ragged tiny 3-D mesh
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}D-n{'x'.join(map(str, c[1]))}-c{c[4][0]}-r{c[5]}{'-comp' if c[6] else ''}" for c in CASES])
def test_degenerate_geometry_capacities_and_solve(pj, case):
    N, n, L, x0, c, r, comp = case
    mesh, omesh = pj.Mesh(n, L, x0), po.Mesh(n, L, x0)
    cap = pj.Capacity(pj.Sphere(c, r, complement=comp), mesh)
    ocap = po.make_capacity(Ball(c, r, complement=comp), omesh)
    assert np.array_equal(cap.cell_types, ocap.cell_types)                       # bit-exact classification
    assert np.array_equal(cap.Γ > 0, ocap.G > 0)                                 # cut set == {Γ > 0} on both sides
    vol = float(np.prod(L)) / float(np.prod(n))
    for a, b in [(cap.V, ocap.V), (cap.Γ, ocap.G)] + [(cap.A[d], ocap.A[d]) for d in range(N)] + \
                [(cap.B[d], ocap.B[d]) for d in range(N)] + [(cap.W[d], ocap.W[d]) for d in range(N)]:
        assert np.max(np.abs(a - b)) <= 2e-10 * max(vol ** ((N - 1) / N) if N > 1 else 1.0, vol, 1e-300) + 1e-13, (np.max(np.abs(a - b)))
    # one BE + two CN steps on the product's own capacities: same active set, same states
    ocap2 = oracle_capacity_from_product(cap, omesh)
    op, oop = pj.DiffusionOps(cap), po.make_diffusion_ops(ocap2)
    f, D = (lambda x, y, z, t: 1.0), (lambda x, y, z: 1.0)
    ph, oph = pj.Phase(cap, op, f, D), po.Phase(ocap2, oop, f, D)
    M = int(np.prod([v + 1 for v in n]))
    keys = {1: ("bottom", "top"), 2: ("left", "right", "top", "bottom"), 3: ("left", "right", "top", "bottom", "forward", "backward")}[N]
    bcb, obcb = pj.BorderConditions({k: pj.Dirichlet(0.5) for k in keys}), po.BorderConditions({k: po.Dirichlet(0.5) for k in keys})
    dt = 0.1 * min(L[d] / n[d] for d in range(N)) ** 2
    u0 = np.zeros(2 * M)
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, u0, "BE")
    so = po.DiffusionUnsteadyMono(oph, obcb, po.Dirichlet(1.0), dt, u0, "BE")
    A, b, idx = s.system(0)
    Ar, br, oidx = po.remove_zero_rows_cols(so.A, so.b)
    assert np.array_equal(idx, oidx)
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 2 * dt, bcb, pj.Dirichlet(1.0), "CN", reltol=1e-13)
    po.solve_DiffusionUnsteadyMono(so, oph, dt, 2 * dt, obcb, po.Dirichlet(1.0), "CN", method="\\")
    assert len(s.states) == len(so.states)
    for a, b in zip(s.states, so.states):
        if np.linalg.norm(b) == 0.0:
            assert np.linalg.norm(a) == 0.0
        else:
            assert rel_l2(a, b) <= 1e-10


def _lattice_case(seed):
    """Centre on the half-spacing lattice of the mesh (nodes, cell centres, face centres, corners) and a radius that is a
    multiple of h/2: every draw has exact tangencies and interfaces through nodes; all numbers are dyadic, so the
    arithmetic of the classification is exact on both sides."""
    rng = np.random.default_rng(4200 + seed)
    N = int(rng.integers(1, 4))
    n = tuple(int(v) for v in rng.integers(4, 9 if N == 3 else 17, size=N))
    h = float(rng.choice([0.25, 0.5, 0.125]))
    L = tuple(h * v for v in n)
    x0 = tuple(float(rng.integers(-4, 5)) * h for _ in range(N))
    c = tuple(x0[d] + 0.5 * h * float(rng.integers(2, 2 * n[d] - 1)) for d in range(N))
    r = 0.5 * h * float(rng.integers(1, max(2, min(n))))
    return N, n, L, x0, c, r, bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", range(40))
def test_lattice_aligned_bodies_capacities(pj, seed):
    N, n, L, x0, c, r, comp = _lattice_case(seed)
    mesh, omesh = pj.Mesh(n, L, x0), po.Mesh(n, L, x0)
    cap = pj.Capacity(pj.Sphere(c, r, complement=comp), mesh)
    ocap = po.make_capacity(Ball(c, r, complement=comp), omesh)
    info = (N, n, L, x0, c, r, comp)
    assert np.array_equal(cap.cell_types, ocap.cell_types), info
    assert np.array_equal(cap.Γ > 0, ocap.G > 0), info
    h = L[0] / n[0]
    for name, a, b, scale in [("V", cap.V, ocap.V, h ** N), ("Γ", cap.Γ, ocap.G, h ** (N - 1))] + \
            [(f"A{d}", cap.A[d], ocap.A[d], h ** (N - 1)) for d in range(N)] + \
            [(f"B{d}", cap.B[d], ocap.B[d], h ** (N - 1)) for d in range(N)] + \
            [(f"W{d}", cap.W[d], ocap.W[d], h ** N) for d in range(N)]:
        err = np.max(np.abs(a - b))
        # B_d / W_d go through the centroid of the cell (a ratio of moments): sliver cells amplify its rounding
        assert err <= (1e-7 if name[0] in "BW" else 1e-10) * scale, (name, err, info)
    # the unknowns a solver built on them keeps: bit-exact index set
    op, oop = pj.DiffusionOps(cap), po.make_diffusion_ops(oracle_capacity_from_product(cap, omesh))
    M = int(np.prod([v + 1 for v in n]))
    one = lambda *a: 1.0
    ph = pj.Phase(cap, op, one, one)
    oph = po.Phase(oracle_capacity_from_product(cap, omesh), oop, one, one)
    s = pj.DiffusionUnsteadyMono(ph, pj.BorderConditions({}), pj.Robin(1.0, 1.0, 0.5), 0.01, np.zeros(2 * M), "BE")
    so = po.DiffusionUnsteadyMono(oph, po.BorderConditions({}), po.Robin(1.0, 1.0, 0.5), 0.01, np.zeros(2 * M), "BE")
    _, _, idx = s.system(0)
    _, _, oidx = po.remove_zero_rows_cols(so.A, so.b)
    assert np.array_equal(idx, oidx), info
