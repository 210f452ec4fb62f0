"""CPU unit test of the geometry DEVICE functions (penguin/jl_amd/csrc/pg_geom.h) compiled for the host
with g++ (tests/geom_host.cpp -> tests/_build/libgeom_host.so; test-only build, never loaded by the
product) against the oracle's independent formulation (oracle/geometry.py)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle.geometry import Ball, MultiBall

ROOT = Path(__file__).resolve().parent.parent
P = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lib():
    out = ROOT / "tests" / "_build" / "libgeom_host.so"
    out.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "geom_host.cpp"
    hdr = ROOT / "penguin" / "jl_amd" / "csrc" / "pg_geom.h"
    if not out.exists() or out.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(out), str(src)], check=True)
    l = C.CDLL(str(out))
    l.geom_section.restype = C.c_double
    return l


def cbox(lib, N, centers, r, lo, hi, comp=0, surf=1):
    out = np.zeros(9)
    cc = np.ascontiguousarray(centers, dtype=float).reshape(-1)
    lo = np.ascontiguousarray(lo, dtype=float)
    hi = np.ascontiguousarray(hi, dtype=float)
    lib.geom_box(N, len(cc) // N, comp, C.c_double(r), cc.ctypes.data_as(P), lo.ctypes.data_as(P), hi.ctypes.data_as(P), surf,
                 out.ctypes.data_as(P))
    return out


def csec(lib, N, centers, r, d, s, lo, hi, comp=0):
    cc = np.ascontiguousarray(centers, dtype=float).reshape(-1)
    lo = np.ascontiguousarray(lo, dtype=float)
    hi = np.ascontiguousarray(hi, dtype=float)
    return lib.geom_section(N, len(cc) // N, comp, C.c_double(r), cc.ctypes.data_as(P), d, C.c_double(s), lo.ctypes.data_as(P),
                            hi.ctypes.data_as(P))


@pytest.mark.parametrize("N,trials", [(1, 300), (2, 1500), (3, 150)])
def test_box_and_section_measures_match_oracle(lib, N, trials):
    rng = np.random.default_rng(10 + N)
    ncut = 0
    for _ in range(trials):
        c = rng.uniform(-0.2, 0.2, N)
        r = rng.uniform(0.5, 1.5)
        h = rng.choice([0.05, 0.2, 0.01])
        dirv = rng.normal(size=N)
        dirv /= np.linalg.norm(dirv)
        p = c + dirv * r + rng.uniform(-h, h, N) * 0.7
        lo, hi = p - h / 2, p + h / 2
        for comp in (0, 1):
            m = Ball(c, r, complement=bool(comp)).box(list(lo), list(hi))
            o = cbox(lib, N, c, r, lo, hi, comp)
            assert int(o[0]) == m.type                                  # bit-exact classification
            full = h ** N
            if m.type == -1:
                ncut += 1
                assert abs(o[1] - m.vol) <= 2e-11 * full
                assert abs(o[5] - m.gamma) <= 2e-11 * max(h ** (N - 1), 1.0)
                # centroids are moment / measure: the absolute moment error (cancellation ~1e-11 full*h) is
                # amplified by full/vol in nearly empty cells
                R = r + 0.2
                if m.vol > 1e-3 * full:
                    assert np.max(np.abs(o[2:2 + N] - np.array(m.centroid))) <= 2e-15 * R ** (N + 1) / m.vol + 1e-14
                if m.gamma > 1e-3 * h ** (N - 1):
                    assert np.max(np.abs(o[6:6 + N] - np.array(m.cgamma))) <= 2e-15 * R ** N / m.gamma + 1e-14
            else:
                assert o[1] == m.vol                                    # full/empty: identical expression
            for d in range(N):
                s = rng.uniform(lo[d], hi[d])
                ref = Ball(c, r, complement=bool(comp)).section(d, s, list(lo), list(hi))
                assert abs(csec(lib, N, c, r, d, s, lo, hi, comp) - ref) <= 1e-11 * max(h ** (N - 1), 1e-300)
    assert ncut > trials // 4


def test_faces_of_full_cells_are_full_bit_for_bit(lib):
    """H = A D⁻ − D⁻ B vanishes exactly away from the interface only if A == B bitwise there."""
    rng = np.random.default_rng(3)
    c, r = np.array([2.01, 2.01, 2.01]), 1.0
    nodes = 0.0 + (np.arange(33) + 0.5) * (4.0 / 32)
    for _ in range(300):
        i = rng.integers(0, 32, 3)
        lo = nodes[i]
        hi = nodes[i + 1]
        o = cbox(lib, 3, c, r, lo, hi)
        if int(o[0]) != 1:
            continue
        for d in range(3):
            others = [k for k in range(3) if k != d]
            full = (hi[others[0]] - lo[others[0]]) * (hi[others[1]] - lo[others[1]])
            assert csec(lib, 3, c, r, d, lo[d], lo, hi) == full
            assert csec(lib, 3, c, r, d, hi[d], lo, hi) == full
            assert csec(lib, 3, c, r, d, o[2 + d], lo, hi) == full


def test_multiball_picks_the_right_ball(lib):
    centers = np.array([[2.01, 2.01, 2.01], [2.01, 2.01, 6.01]])
    mb = MultiBall(centers, 1.0)
    for z in (1.2, 2.9, 5.1, 6.9, 4.0):
        lo, hi = [2.0, 2.0, z], [2.1, 2.1, z + 0.1]
        m = mb.box(lo, hi)
        o = cbox(lib, 3, centers, 1.0, lo, hi)
        assert int(o[0]) == m.type
        assert abs(o[1] - m.vol) <= 1e-14


@pytest.mark.parametrize("N", [2, 3])
def test_boxes_tangent_to_the_ball_at_an_interior_point_of_a_face(lib, N):
    """A box face that touches the ball exactly (|t| == rho: a double root of the chord function, met by every mesh whose
    nodes are round numbers -- centre on a node, r a multiple of h) must not flip the closed-form integration to the full
    rectangle: found by tests/test_gpu_degenerate.py (W_d of the two cells around the south pole came out as the full
    staggered box)."""
    c, r = np.array([2.125, 2.125, 2.125])[:N], 1.0
    # staggered box around the south pole of the ball: symmetric in x, bottom face tangent at its midpoint
    for half in (0.12225, 0.125, 0.03):
        for (t0, t1) in ((1.125, 1.375), (0.875, 1.125), (3.125, 3.375), (2.875, 3.125)):      # tangent from inside / outside, south / north
            lo = [2.125 - half, t0] + ([2.125 - 0.1] if N == 3 else [])
            hi = [2.125 + half, t1] + ([2.125 + 0.15] if N == 3 else [])
            for comp in (0, 1):
                m = Ball(c, r, complement=bool(comp)).box(lo, hi)
                o = cbox(lib, N, c, r, np.array(lo), np.array(hi), comp)
                assert int(o[0]) == m.type
                full = float(np.prod(np.array(hi) - np.array(lo)))
                assert abs(o[1] - m.vol) <= 2e-11 * full, (half, t0, comp, o[1], m.vol)
                if m.type == -1:
                    assert abs(o[5] - m.gamma) <= 2e-10 * max(full ** ((N - 1) / N), 1e-300)
    # sections through the tangent plane (A_d / B_d of such cells)
    for d in range(N):
        lo = list(c - 0.2)
        hi = list(c + 0.3)
        for s in (c[d] - r, c[d] + r):
            for comp in (0, 1):
                ref = Ball(c, r, complement=bool(comp)).section(d, s, lo, hi)
                assert abs(csec(lib, N, c, r, d, s, np.array(lo), np.array(hi), comp) - ref) <= 1e-12


# ------------------------------------------------------------------------------------------------ ellipsoids
def _ebox(lib, N, c, ax, lo, hi, comp=0, surf=1):
    out = np.zeros(9)
    arrs = [np.ascontiguousarray(v, dtype=float) for v in (c, ax, lo, hi)]
    lib.geom_ellipsoid_box(N, comp, *[a.ctypes.data_as(P) for a in arrs], surf, out.ctypes.data_as(P))
    return out


def _esec(lib, N, c, ax, d, s, lo, hi, comp=0):
    lib.geom_ellipsoid_section.restype = C.c_double
    arrs = [np.ascontiguousarray(v, dtype=float) for v in (c, ax, lo, hi)]
    return lib.geom_ellipsoid_section(N, comp, arrs[0].ctypes.data_as(P), arrs[1].ctypes.data_as(P), d, C.c_double(s),
                                      arrs[2].ctypes.data_as(P), arrs[3].ctypes.data_as(P))


def test_ellipse_perimeter_and_area_are_the_closed_forms(lib):
    """whole ellipse in one box cut into four quadrant boxes: area π a b, perimeter 4 a E(e) (scipy.special.ellipe)."""
    from scipy.special import ellipe
    from oracle.geometry import Ellipsoid
    c, ax = (0.3, -0.2), (1.7, 0.6)
    area = per = 0.0
    for lo, hi in (((-3, -2), (0.3, -0.2)), ((0.3, -2), (3, -0.2)), ((-3, -0.2), (0.3, 2)), ((0.3, -0.2), (3, 2))):
        o = _ebox(lib, 2, c, ax, lo, hi)
        area += o[1]
        per += o[5]
        ob = Ellipsoid(c, ax).box(lo, hi)
        assert abs(o[1] - ob.vol) < 1e-13 and abs(o[5] - ob.gamma) < 1e-12
    assert abs(area - np.pi * 1.7 * 0.6) < 1e-13
    assert abs(per - 4 * 1.7 * ellipe(1 - (0.6 / 1.7) ** 2)) < 1e-11     # 16 nodes over a whole quadrant of an e = 0.94 ellipse


@pytest.mark.parametrize("N", [2, 3])
def test_ellipsoid_boxes_and_sections_match_oracle(lib, N):
    """random boxes around the surface: classification identical; volumes, centroids, sections to rounding; the interface
    measure and its centroid to the accuracy of the two (different) quadratures."""
    from oracle.geometry import Ellipsoid
    rng = np.random.default_rng(5 + N)
    c = (0.11, -0.23, 0.31)[:N]
    ax = (1.3, 0.7, 0.45)[:N]
    E, Ec = Ellipsoid(c, ax), Ellipsoid(c, ax, True)
    ncut = 0
    for _ in range(60 if N == 2 else 14):
        u = rng.standard_normal(N)
        u /= np.linalg.norm(u)
        p = np.array(c) + np.array(ax) * u                     # a point of the surface
        h = rng.uniform(0.05, 0.4, N)
        lo = p - h * rng.uniform(0.1, 0.9, N)
        hi = lo + h
        for comp, body in ((0, E), (1, Ec)):
            o, ob = _ebox(lib, N, c, ax, lo, hi, comp), body.box(list(lo), list(hi))
            assert int(o[0]) == ob.type
            if ob.type != -1:
                continue
            ncut += 1
            cell = float(np.prod(h))
            assert abs(o[1] - ob.vol) <= 1e-11 * cell
            if ob.vol > 1e-3 * cell:
                assert np.max(np.abs(o[2:2 + N] - np.array(ob.centroid))) <= 1e-9 * h.max()
            assert abs(o[5] - ob.gamma) <= 1e-9 * max(ob.gamma, cell ** ((N - 1) / N))
            if ob.gamma > 1e-3 * cell ** ((N - 1) / N):
                assert np.max(np.abs(o[6:6 + N] - np.array(ob.cgamma))) <= 1e-8 * h.max()
            for d in range(N):
                for s in (lo[d], 0.5 * (lo[d] + hi[d]), hi[d]):
                    assert abs(_esec(lib, N, c, ax, d, s, lo, hi, comp) - body.section(d, float(s), list(lo), list(hi))) <= 1e-12 * cell / h[d]
    assert ncut > 20


def test_ellipsoid_with_equal_axes_is_the_ball(lib):
    for N in (2, 3):
        c, r = (0.2, 0.1, -0.3)[:N], 0.9
        lo, hi = [0.5, 0.3, 0.1][:N], [1.1, 0.9, 0.7][:N]
        e, b = _ebox(lib, N, c, (r,) * N, lo, hi), cbox(lib, N, [c], r, lo, hi)
        assert int(e[0]) == int(b[0])
        assert np.allclose(e[1:], b[1:], rtol=1e-11, atol=1e-14)
