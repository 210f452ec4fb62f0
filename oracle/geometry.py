"""ORACLE (test infrastructure) -- cut-cell geometry for closed-form bodies.

This file is part of the CPU oracle.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product path never does.

What it restates
----------------
The reference obtains its capacities from the un-vendored C library libvofi
2.0.0 (via Vofinit 0.1.0 / CartesianGeometry 0.1.1, pins in
/root/reference/Manifest.toml:223-229,1995-1999,2136-2140), called at
/root/reference/src/capacity.jl:90-92,103-105.  libvofi integrates the
"heights" of an implicit interface with Gauss-Legendre quadrature between the
kinks of the height function (Bna et al., Comput. Fluids 2015; Chierici et al.,
Comput. Phys. Commun. 2022 for 2.0's centroid / interface measure).  Its source is not in
the tree and Julia cannot run here, so per-cell capacity values are
**parity unpinned**: this oracle computes the *same geometric quantities* for
bodies with a closed form (ball in 1/2/3-D, its complement, unions of disjoint
balls) exactly in 1-D/2-D and by adaptive Gauss-Kronrod quadrature
(scipy quad_vec, tolerance 1e-13) of exact 2-D sections in 3-D.

Definition of every capacity follows the in-tree specification
`GeometricMoments`, /root/reference/src/capacity.jl:264-430:
  V    fluid (f<=0) volume of the cell                       (:273-274)
  type 1 full / 0 empty / -1 cut                             (:277-292)
  C_w  fluid centroid; cell centre for full and empty cells  (:280-299)
  G    interface measure, cut cells only                     (:302)
  A_d  fluid measure of the lower face x_d = nodes_d[i_d]    (:355-371)
  B_d  fluid measure of the section through C_w[d]           (:373-391)
  W_d  fluid volume between the centroids of i-e_d and i     (:396-429)
  C_g  centroid of the interface inside the cell             (:316-347, capacity.jl:137-197)
Formulation used here is deliberately different from the HIP kernels' (signed
quadrant decomposition + angular interval arithmetic here, chord/breakpoint
integration there) so that the parity tests compare two independent derivations.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------
# 1-D / 2-D exact primitives for a ball of radius rho centred at the origin
# ----------------------------------------------------------------------------


def seg_overlap(rho: float, a: float, b: float) -> Tuple[float, float]:
    """Length and first moment of [a,b] ∩ [-rho,rho]."""
    lo = max(a, -rho)
    hi = min(b, rho)
    if hi <= lo:
        return 0.0, 0.0
    return hi - lo, 0.5 * (hi * hi - lo * lo)


def _P(rho: float, u: float) -> float:
    """Antiderivative of sqrt(rho^2-u^2)."""
    s2 = (rho - u) * (rho + u)
    s = math.sqrt(s2) if s2 > 0.0 else 0.0
    return 0.5 * (u * s + rho * rho * math.atan2(u, s))


def _Q(rho: float, a: float, b: float) -> Tuple[float, float, float]:
    """Area, x-moment, y-moment of disc(rho) ∩ [0,a]x[0,b], a,b >= 0."""
    a = min(a, rho)
    b = min(b, rho)
    if a <= 0.0 or b <= 0.0:
        return 0.0, 0.0, 0.0
    r2 = rho * rho
    if a * a + b * b <= r2:
        return a * b, 0.5 * a * a * b, 0.5 * a * b * b
    xb = math.sqrt(max(r2 - b * b, 0.0))  # circle height equals b at x = xb  (xb < a)
    yb = math.sqrt(max(r2 - a * a, 0.0))  # circle height at x = a
    area = b * xb + (_P(rho, a) - _P(rho, xb))
    # int_xb^a x sqrt(r2-x^2) dx = [-(r2-x^2)^{3/2}/3]
    mx = 0.5 * b * xb * xb + (b ** 3 - yb ** 3) / 3.0
    # y-moment: int_0^xb b^2/2 dx + int_xb^a (r2-x^2)/2 dx
    my = 0.5 * b * b * xb + 0.5 * (r2 * (a - xb) - (a ** 3 - xb ** 3) / 3.0)
    return area, mx, my


def _q(rho: float, x: float, y: float) -> Tuple[float, float, float]:
    """F(x,y) = int_0^x int_0^y 1_disc dY dX and the X-, Y-weighted versions."""
    ar, mx, my = _Q(rho, abs(x), abs(y))
    sx = 1.0 if x >= 0 else -1.0
    sy = 1.0 if y >= 0 else -1.0
    return sx * sy * ar, sy * mx, sx * my


def disc_rect(rho: float, a: float, b: float, t0: float, t1: float) -> Tuple[float, float, float]:
    """Area and first moments (about the disc centre) of disc(rho) ∩ [a,b]x[t0,t1]."""
    if rho <= 0.0 or b <= a or t1 <= t0:
        return 0.0, 0.0, 0.0
    f11 = _q(rho, b, t1)
    f01 = _q(rho, a, t1)
    f10 = _q(rho, b, t0)
    f00 = _q(rho, a, t0)
    return tuple(f11[i] - f01[i] - f10[i] + f00[i] for i in range(3))  # type: ignore


def disc_arcs(rho: float, a: float, b: float, t0: float, t1: float) -> Tuple[float, float, float]:
    """Total angle, int cos, int sin over the parts of the circle of radius rho
    that lie inside [a,b]x[t0,t1] (angular interval arithmetic)."""
    if rho <= 0.0:
        return 0.0, 0.0, 0.0
    cand = []
    for xv in (a, b):
        if -rho < xv < rho:
            al = math.atan2(math.sqrt((rho - xv) * (rho + xv)), xv)  # acos(xv/rho), well conditioned
            cand += [al, -al]
    for yv in (t0, t1):
        if -rho < yv < rho:
            al = math.atan2(yv, math.sqrt((rho - yv) * (rho + yv)))  # asin(yv/rho)
            cand += [al, (math.pi - al) if al >= 0 else (-math.pi - al)]

    def inside(phi: float) -> bool:
        x = rho * math.cos(phi)
        y = rho * math.sin(phi)
        return a <= x <= b and t0 <= y <= t1

    if not cand:
        return (2.0 * math.pi, 0.0, 0.0) if inside(0.3) else (0.0, 0.0, 0.0)
    cand = sorted(set(cand))
    tot = ic = isn = 0.0
    m = len(cand)
    for k in range(m):
        p1 = cand[k]
        p2 = cand[(k + 1) % m] + (2.0 * math.pi if k == m - 1 else 0.0)
        if p2 <= p1:
            continue
        if inside(0.5 * (p1 + p2)):
            tot += p2 - p1
            ic += math.sin(p2) - math.sin(p1)
            isn += math.cos(p1) - math.cos(p2)
    return tot, ic, isn


# ----------------------------------------------------------------------------
# N-D measures of ball ∩ box
# ----------------------------------------------------------------------------

FULL, EMPTY, CUT = 1, 0, -1


def _prod(ext: Sequence[float]) -> float:
    p = ext[0]
    for e in ext[1:]:
        p = p * e
    return p


def ball_box_type(c: Sequence[float], r: float, lo: Sequence[float], hi: Sequence[float]) -> int:
    """FULL if every corner satisfies f<=0, EMPTY if the closest point of the box has
    d^2 >= r^2, else CUT.  Sums are accumulated in dimension order (x first) so the HIP
    kernel can reproduce the decision bit for bit."""
    n = len(c)
    r2 = r * r
    # farthest corner: per-dimension max of squared offsets
    far = 0.0
    near = 0.0
    for d in range(n):
        dl = lo[d] - c[d]
        dh = hi[d] - c[d]
        far = far + max(dl * dl, dh * dh)
        if c[d] < lo[d]:
            near = near + dl * dl
        elif c[d] > hi[d]:
            near = near + dh * dh
        else:
            near = near + 0.0
    if far <= r2:
        return FULL
    if near >= r2:
        return EMPTY
    return CUT


@dataclass
class BoxMeasure:
    type: int
    vol: float
    centroid: Tuple[float, ...]
    gamma: float
    cgamma: Tuple[float, ...]


def _ball_box_moments(c, r, lo, hi, want_surface=True):
    """Volume, first moments (about the ball centre), interface measure and its first
    moments for ball(c,r) ∩ box in 1/2/3-D.  Box is assumed CUT (no shortcuts)."""
    n = len(c)
    a = [lo[d] - c[d] for d in range(n)]
    b = [hi[d] - c[d] for d in range(n)]
    if n == 1:
        L, m1 = seg_overlap(r, a[0], b[0])
        g = 0.0
        gm = 0.0
        for p in (-r, r):  # interface points: measure 1 each (0-D)
            if a[0] <= p <= b[0]:
                g += 1.0
                gm += p
        return L, (m1,), g, (gm,)
    if n == 2:
        ar, mx, my = disc_rect(r, a[0], b[0], a[1], b[1])
        ph, ic, isn = disc_arcs(r, a[0], b[0], a[1], b[1])
        return ar, (mx, my), r * ph, (r * r * ic, r * r * isn)
    if n == 3:
        from scipy.integrate import quad_vec

        z0 = max(a[2], -r)
        z1 = min(b[2], r)
        if z1 <= z0:
            return 0.0, (0.0, 0.0, 0.0), 0.0, (0.0, 0.0, 0.0)
        # breakpoints: rho(z) equals the distance to a corner / an edge line of the rectangle
        crit = [0.0]
        for xv in (a[0], b[0]):
            crit.append(abs(xv))
            for yv in (a[1], b[1]):
                crit.append(math.hypot(xv, yv))
        for yv in (a[1], b[1]):
            crit.append(abs(yv))
        pts = set()
        for rc in crit:
            if rc < r:
                zz = math.sqrt(r * r - rc * rc)
                for s in (zz, -zz):
                    if z0 < s < z1:
                        pts.add(s)

        def integrand(z):
            rho2 = r * r - z * z
            rho = math.sqrt(rho2) if rho2 > 0 else 0.0
            ar, mx, my = disc_rect(rho, a[0], b[0], a[1], b[1])
            if want_surface:
                ph, ic, isn = disc_arcs(rho, a[0], b[0], a[1], b[1])
            else:
                ph = ic = isn = 0.0
            return np.array([ar, mx, my, z * ar, ph, rho * ic, rho * isn, z * ph])

        res, _err = quad_vec(integrand, z0, z1, epsabs=1e-15, epsrel=1e-13,
                             points=sorted(pts) if pts else None, limit=400)
        vol, mx, my, mz, ph, gx, gy, gz = (float(v) for v in res)
        return vol, (mx, my, mz), r * ph, (r * gx, r * gy, r * gz)
    raise ValueError("dimension must be 1, 2 or 3")


class Ball:
    """Level set f(x) = |x-c| - r ; fluid where f <= 0 (inside), or outside if complement."""

    def __init__(self, center: Sequence[float], radius: float, complement: bool = False):
        self.c = tuple(float(v) for v in center)
        self.r = float(radius)
        self.complement = bool(complement)
        self.N = len(self.c)

    # ---- the level-set itself (for host-side checks / plots) -------------------------
    def __call__(self, *x):
        s = 0.0
        for d in range(self.N):
            s += (x[d] - self.c[d]) ** 2
        f = math.sqrt(s) - self.r
        return -f if self.complement else f

    # ---- N-D box ------------------------------------------------------------------
    def box(self, lo: Sequence[float], hi: Sequence[float], want_surface: bool = True) -> BoxMeasure:
        n = len(lo)
        assert n == self.N
        ext = [hi[d] - lo[d] for d in range(n)]
        ctr = tuple(0.5 * (lo[d] + hi[d]) for d in range(n))
        zero = tuple(0.0 for _ in range(n))
        if any(e <= 0.0 for e in ext):
            # degenerate (zero-width) box: volume 0; type from the point/face itself
            t = ball_box_type(self.c, self.r, lo, hi)
            if self.complement and t != CUT:
                t = 1 - t
            return BoxMeasure(t, 0.0, ctr, 0.0, zero)
        t = ball_box_type(self.c, self.r, lo, hi)
        full = _prod(ext)
        if t != CUT:
            if self.complement:
                t = 1 - t
            return BoxMeasure(t, full if t == FULL else 0.0, ctr, 0.0, zero)
        vol, mom, g, gm = _ball_box_moments(self.c, self.r, lo, hi, want_surface)
        if self.complement:
            vc = full - vol
            momc = tuple(full * (ctr[d] - self.c[d]) - mom[d] for d in range(n))
            vol, mom = vc, momc
        if vol > 0.0:
            cen = tuple(self.c[d] + mom[d] / vol for d in range(n))
        else:
            cen = ctr
        cg = tuple(self.c[d] + gm[d] / g for d in range(n)) if g > 0.0 else zero
        return BoxMeasure(CUT, vol, cen, g, cg)

    # ---- (N-1)-D section x_d = s of a box ----------------------------------------
    def section(self, d: int, s: float, lo: Sequence[float], hi: Sequence[float]) -> float:
        """Fluid measure of {x_d = s} ∩ box (lo/hi hold all N dims; entry d is ignored)."""
        n = self.N
        if n == 1:
            f = abs(s - self.c[0]) - self.r
            f = -f if self.complement else f
            return 1.0 if f <= 0.0 else 0.0
        others = [k for k in range(n) if k != d]
        ext = [hi[k] - lo[k] for k in others]
        full = _prod(ext)
        # classify with the N-D level set so that faces of a full cell are full bit-for-bit
        plo = list(lo)
        phi = list(hi)
        plo[d] = s
        phi[d] = s
        t = ball_box_type(self.c, self.r, plo, phi)
        if t != CUT:
            if self.complement:
                t = 1 - t
            return full if t == FULL else 0.0
        dz = s - self.c[d]
        rho2 = self.r * self.r - dz * dz
        rho = math.sqrt(rho2) if rho2 > 0.0 else 0.0
        a = [lo[k] - self.c[k] for k in others]
        b = [hi[k] - self.c[k] for k in others]
        if n == 2:
            m, _ = seg_overlap(rho, a[0], b[0])
        else:
            m, _, _ = disc_rect(rho, a[0], b[0], a[1], b[1])
        return (full - m) if self.complement else m


class MultiBall:
    """Union of pairwise-disjoint balls (weak-scaling body):
    f = min_s f_s.  A box is assumed to meet at most one ball."""

    def __init__(self, centers: Sequence[Sequence[float]], radius: float):
        self.balls = [Ball(c, radius) for c in centers]
        self.N = self.balls[0].N
        self.complement = False

    def __call__(self, *x):
        return min(b(*x) for b in self.balls)

    def _pick(self, lo, hi) -> Ball:
        best = None
        for b in self.balls:
            t = ball_box_type(b.c, b.r, lo, hi)
            if t != EMPTY:
                return b
            best = b
        return best  # all empty: any ball gives the empty answer

    def box(self, lo, hi, want_surface=True):
        return self._pick(lo, hi).box(lo, hi, want_surface)

    def section(self, d, s, lo, hi):
        plo = list(lo)
        phi = list(hi)
        plo[d] = s
        phi[d] = s
        return self._pick(plo, phi).section(d, s, lo, hi)


class HalfSpace:
    """Level set f(x) = sign * (x[axis] - pos); fluid where f < 0 (f > 0 if complement) -- the reference's 1-D diphasic
    bodies `(x, _=0) -> (x - xint)` (/root/reference/test/convergence_test.jl:111-112,230-231) and their extrusions.
    Capacities are exact: a cut cell is the box cut by one axis-aligned plane."""

    def __init__(self, axis: int, pos: float, sign: float = 1.0, complement: bool = False, N: int = 1):
        self.axis, self.pos = int(axis), float(pos)
        self.sign = -1.0 if sign < 0 else 1.0
        self.complement = bool(complement)
        self.N = N

    def __call__(self, *x):
        f = self.sign * (x[self.axis] - self.pos)
        return -f if self.complement else f

    def _interval(self, lo: float, hi: float):
        below = (self.sign > 0.0) != self.complement          # fluid = {x < pos}
        if below:
            if hi <= self.pos:
                return FULL, lo, hi
            if lo >= self.pos:
                return EMPTY, lo, hi
            return CUT, lo, self.pos
        if lo >= self.pos:
            return FULL, lo, hi
        if hi <= self.pos:
            return EMPTY, lo, hi
        return CUT, self.pos, hi

    def box(self, lo, hi, want_surface=True) -> BoxMeasure:
        n = len(lo)
        ext = [hi[d] - lo[d] for d in range(n)]
        ctr = tuple(0.5 * (lo[d] + hi[d]) for d in range(n))
        zero = tuple(0.0 for _ in range(n))
        t, flo, fhi = self._interval(lo[self.axis], hi[self.axis])
        if any(e <= 0.0 for e in ext):
            return BoxMeasure(t, 0.0, ctr, 0.0, zero)
        if t != CUT:
            return BoxMeasure(t, _prod(ext) if t == FULL else 0.0, ctr, 0.0, zero)
        others = [ext[d] for d in range(n) if d != self.axis]
        cross = _prod(others) if others else 1.0
        cen = list(ctr)
        cen[self.axis] = 0.5 * (flo + fhi)
        cg = list(ctr)
        cg[self.axis] = self.pos
        return BoxMeasure(CUT, (fhi - flo) * cross, tuple(cen), cross, tuple(cg))

    def section(self, d: int, s: float, lo, hi) -> float:
        n = len(lo)
        others = [k for k in range(n) if k != d]
        full = _prod([hi[k] - lo[k] for k in others]) if others else 1.0
        if d == self.axis:
            f = self.sign * (s - self.pos)
            f = -f if self.complement else f
            return full if f <= 0.0 else 0.0
        t, flo, fhi = self._interval(lo[self.axis], hi[self.axis])
        if t == FULL:
            return full
        if t == EMPTY:
            return 0.0
        m = fhi - flo
        for k in others:
            if k != self.axis:
                m = m * (hi[k] - lo[k])
        return m


# ----------------------------------------------------------------------------
# axis-aligned ellipsoid
# ----------------------------------------------------------------------------
_GL24 = np.polynomial.legendre.leggauss(24)


def _arcs_weighted(rho: float, a: float, b: float, t0: float, t1: float, w) -> Tuple[float, float, float]:
    """disc_arcs with a weight: int w, int w cos, int w sin over the parts of the circle of radius rho inside
    [a,b]x[t0,t1]; w(phi) is smooth, 24-point Gauss-Legendre per angular interval."""
    if rho <= 0.0:
        return 0.0, 0.0, 0.0
    cand = []
    for xv in (a, b):
        if -rho < xv < rho:
            al = math.atan2(math.sqrt((rho - xv) * (rho + xv)), xv)
            cand += [al, -al]
    for yv in (t0, t1):
        if -rho < yv < rho:
            al = math.atan2(yv, math.sqrt((rho - yv) * (rho + yv)))
            cand += [al, (math.pi - al) if al >= 0 else (-math.pi - al)]

    def inside(phi: float) -> bool:
        return a <= rho * math.cos(phi) <= b and t0 <= rho * math.sin(phi) <= t1

    if not cand:
        if not inside(0.3):
            return 0.0, 0.0, 0.0
        cand = [-math.pi, 0.0, math.pi]      # the whole circle, in two pieces
        pieces = [(-math.pi, 0.0), (0.0, math.pi)]
    else:
        cand = sorted(set(cand))
        m = len(cand)
        pieces = []
        for k in range(m):
            p1 = cand[k]
            p2 = cand[(k + 1) % m] + (2.0 * math.pi if k == m - 1 else 0.0)
            if p2 > p1 and inside(0.5 * (p1 + p2)):
                pieces.append((p1, p2))
    x, wt = _GL24
    tot = ic = isn = 0.0
    for p1, p2 in pieces:
        ph = 0.5 * (p1 + p2) + 0.5 * (p2 - p1) * x
        ww = 0.5 * (p2 - p1) * wt * w(ph)
        tot += float(np.sum(ww))
        ic += float(np.sum(ww * np.cos(ph)))
        isn += float(np.sum(ww * np.sin(ph)))
    return tot, ic, isn


class Ellipsoid:
    """Level set f(x) = sqrt(sum ((x_d - c_d)/a_d)^2) - 1 (axis-aligned semi-axes a_d), fluid where f <= 0.

    Everything with the dimension of a volume / a section is the unit ball's measure in the scaled coordinates
    x' = (x - c)/a times the product of the semi-axes involved (the map is affine).  The interface measure is not affine
    invariant: a surface element of the unit sphere with normal n maps to one of area sqrt(sum_d (n_d prod_(k!=d) a_k)^2)
    (Nanson), integrated here along the arcs of the scaled sections (2-D: the ellipse's arc length)."""

    def __init__(self, center: Sequence[float], semi_axes: Sequence[float], complement: bool = False):
        self.c = tuple(float(v) for v in center)
        self.a = tuple(float(v) for v in semi_axes)
        assert len(self.c) == len(self.a) and all(v > 0 for v in self.a)
        self.complement = bool(complement)
        self.N = len(self.c)
        self._unit = Ball(tuple(0.0 for _ in self.c), 1.0, False)

    def __call__(self, *x):
        f = math.sqrt(sum(((x[d] - self.c[d]) / self.a[d]) ** 2 for d in range(self.N))) - 1.0
        return -f if self.complement else f

    def _scaled(self, lo, hi):
        n = self.N
        return ([(lo[d] - self.c[d]) / self.a[d] for d in range(n)], [(hi[d] - self.c[d]) / self.a[d] for d in range(n)])

    def _surface(self, slo, shi):
        """interface measure and its first moments (scaled coordinates about the centre) inside the scaled box"""
        n = self.N
        if n == 2:
            k0, k1 = self.a[1], self.a[0]
            w = lambda ph: np.sqrt((k0 * np.cos(ph)) ** 2 + (k1 * np.sin(ph)) ** 2)
            g, gx, gy = _arcs_weighted(1.0, slo[0], shi[0], slo[1], shi[1], w)
            return g, (gx, gy)
        from scipy.integrate import quad_vec
        k = (self.a[1] * self.a[2], self.a[0] * self.a[2], self.a[0] * self.a[1])
        z0, z1 = max(slo[2], -1.0), min(shi[2], 1.0)
        if z1 <= z0:
            return 0.0, (0.0, 0.0, 0.0)
        crit = [0.0]
        for xv in (slo[0], shi[0]):
            crit.append(abs(xv))
            for yv in (slo[1], shi[1]):
                crit.append(math.hypot(xv, yv))
        for yv in (slo[1], shi[1]):
            crit.append(abs(yv))
        pts = sorted({s for rc in crit if rc < 1.0 for s in (math.sqrt(1 - rc * rc), -math.sqrt(1 - rc * rc)) if z0 < s < z1})

        def integrand(z):
            rho = math.sqrt(max(1.0 - z * z, 0.0))
            w = lambda ph: np.sqrt((k[0] * rho * np.cos(ph)) ** 2 + (k[1] * rho * np.sin(ph)) ** 2 + (k[2] * z) ** 2)
            g, ic, isn = _arcs_weighted(rho, slo[0], shi[0], slo[1], shi[1], w)
            return np.array([g, rho * ic, rho * isn, z * g])

        res, _ = quad_vec(integrand, z0, z1, epsabs=1e-15, epsrel=1e-12, points=pts or None, limit=400)
        return float(res[0]), (float(res[1]), float(res[2]), float(res[3]))

    def box(self, lo, hi, want_surface: bool = True) -> BoxMeasure:
        n = self.N
        if n == 1:
            return Ball(self.c, self.a[0], self.complement).box(lo, hi, want_surface)
        ext = [hi[d] - lo[d] for d in range(n)]
        ctr = tuple(0.5 * (lo[d] + hi[d]) for d in range(n))
        zero = tuple(0.0 for _ in range(n))
        slo, shi = self._scaled(lo, hi)
        t = ball_box_type(self._unit.c, 1.0, slo, shi)
        if any(e <= 0.0 for e in ext) or t != CUT:
            if self.complement and t != CUT:
                t = 1 - t
            return BoxMeasure(t, _prod(ext) if (t == FULL and all(e > 0.0 for e in ext)) else 0.0, ctr, 0.0, zero)
        full = _prod(ext)
        J = _prod(list(self.a))
        vol, mom, _, _ = _ball_box_moments(self._unit.c, 1.0, slo, shi, want_surface=False)
        vol = vol * J
        mom = tuple(mom[d] * J * self.a[d] for d in range(n))
        if self.complement:
            vol, mom = full - vol, tuple(full * (ctr[d] - self.c[d]) - mom[d] for d in range(n))
        cen = tuple(self.c[d] + mom[d] / vol for d in range(n)) if vol > 0.0 else ctr
        g, cg = 0.0, zero
        if want_surface:
            g, gm = self._surface(slo, shi)
            if g > 0.0:
                cg = tuple(self.c[d] + self.a[d] * gm[d] / g for d in range(n))
        return BoxMeasure(CUT, vol, cen, g, cg)

    def section(self, d: int, s: float, lo, hi) -> float:
        n = self.N
        if n == 1:
            return Ball(self.c, self.a[0], self.complement).section(d, s, lo, hi)
        others = [k for k in range(n) if k != d]
        full = _prod([hi[k] - lo[k] for k in others])
        plo, phi = list(lo), list(hi)
        plo[d] = phi[d] = s
        slo, shi = self._scaled(plo, phi)
        t = ball_box_type(self._unit.c, 1.0, slo, shi)
        if t != CUT:
            if self.complement:
                t = 1 - t
            return full if t == FULL else 0.0
        dz = (s - self.c[d]) / self.a[d]
        rho2 = (1.0 - dz) * (1.0 + dz)
        rho = math.sqrt(rho2) if rho2 > 0.0 else 0.0
        a = [slo[k] for k in others]
        b = [shi[k] for k in others]
        if n == 2:
            m, _ = seg_overlap(rho, a[0], b[0])
        else:
            m, _, _ = disc_rect(rho, a[0], b[0], a[1], b[1])
        m = m * _prod([self.a[k] for k in others])
        return (full - m) if self.complement else m
