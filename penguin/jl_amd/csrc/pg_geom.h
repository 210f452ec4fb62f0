// pg_geom.h -- per-cell cut-cell geometry for closed-form level sets (ball / union of disjoint
// balls / complement), usable from HIP kernels and from a host (g++) unit-test build.
//
// Replaces what the reference obtains from libvofi through CartesianGeometry.integrate
// (src/capacity.jl:90-92,103-105) and ImplicitIntegration (src/capacity.jl:182-187); capacity
// definitions follow GeometricMoments, src/capacity.jl:264-430.
//
// Formulation (deliberately different from oracle/geometry.py):
//   2-D  : disc ∩ rectangle integrated along x between the kinks of the chord function with
//          closed-form antiderivatives; the same sweep yields the arcs inside the rectangle.
//   3-D  : Gauss-Legendre quadrature in z of the exact 2-D sections, on the sub-intervals between
//          the kinks of rho(z) (circle meets a corner / an edge line of the rectangle), with the
//          substitution z = zm + zh*(3t - t^3)/2 that removes the half-integer endpoint
//          singularities (libvofi integrates heights between kinks with plain Gauss-Legendre).
// Classification (full / empty / cut) uses only +,-,*,compare in a fixed order so that the host
// oracle reproduces it bit for bit (compile with -ffp-contract=off).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PG_HD __host__ __device__ inline
#else
#define PG_HD inline
#endif

namespace pggeom {

constexpr int PG_FULL = 1, PG_EMPTY = 0, PG_CUT = -1;
constexpr int NGL = 16;       // Gauss-Legendre points per z sub-interval
constexpr int MAX_BALLS = 16;

struct GLTable {
  double x[NGL];
  double w[NGL];
};

constexpr int BODY_BALLS = 0, BODY_HALFSPACE = 1, BODY_ELLIPSOID = 2;

struct BallSet {
  int N;          // spatial dimension 1..3
  int nballs;
  int complement; // fluid outside the ball(s) / on the other side of the plane
  double r;
  double c[MAX_BALLS][3];
  // axis-aligned half space (kind == BODY_HALFSPACE): level set f(x) = sgn (x_axis - pos), fluid where f < 0 -- the
  // reference's 1-D diphasic bodies `(x, _=0) -> x - xint` (test/convergence_test.jl:111,230) and their N-D extrusions
  int kind;
  int axis;
  double pos, sgn;
  // axis-aligned ellipsoid (kind == BODY_ELLIPSOID): f(x) = sqrt(sum ((x_d - c_d) / ax_d)^2) - 1, centre c[0].  Everything is
  // computed on the unit ball in the scaled coordinates x' = (x - c) / ax: volumes, sections and their moments scale with
  // products of the semi-axes; the interface measure does not (it is not affine invariant) and carries the weight
  // |F^-T n| det F = sqrt(sum_d (n_d prod_(k != d) ax_k)^2) of the unit normal n along the arcs (ellipse_weight)
  double ax[3];
};

struct BoxMeasure {
  int type;
  double vol;
  double cen[3];
  double gamma;
  double cg[3];
};

// ---------------------------------------------------------------------------------------------
PG_HD double dmax(double a, double b) { return a > b ? a : b; }
PG_HD double dmin(double a, double b) { return a < b ? a : b; }

// FULL if the farthest corner is inside (f <= 0), EMPTY if the closest point has d^2 >= r^2.
PG_HD int ball_box_type(const double* c, double r, const double* lo, const double* hi, int N) {
  const double r2 = r * r;
  double far = 0.0, near = 0.0;
  for (int d = 0; d < N; ++d) {
    const double dl = lo[d] - c[d];
    const double dh = hi[d] - c[d];
    const double l2 = dl * dl, h2 = dh * dh;
    far = far + dmax(l2, h2);
    if (c[d] < lo[d]) near = near + l2;
    else if (c[d] > hi[d]) near = near + h2;
    else near = near + 0.0;
  }
  if (far <= r2) return PG_FULL;
  if (near >= r2) return PG_EMPTY;
  return PG_CUT;
}

// antiderivatives on the disc of radius rho
// sqrt(rho^2-u^2) as sqrt((rho-u)(rho+u)): accurate when |u| ~ rho
PG_HD double sroot(double rho, double u) {
  const double s2 = (rho - u) * (rho + u);
  return s2 > 0.0 ? sqrt(s2) : 0.0;
}
// asin(u/rho) = atan2(u, sqrt(rho^2-u^2)): well conditioned near |u| = rho
PG_HD double asin_r(double u, double s) { return atan2(u, s); }

struct Sec2 {
  double area, mx, my;   // area and first moments about the disc centre
  double phi, ic, is;    // total angle of the circle inside, int cos, int sin over it
};

// disc(rho) ∩ [a,b] x [t0,t1], coordinates relative to the disc centre.
// ew != nullptr (ellipsoids): the arc integrals carry the weight w = sqrt((ew[0] u)^2 + (ew[1] t)^2 + ew[2]) of the point
// (u, t) = rho (sin α, ±cos α) of the circle, integrated with the Gauss-Legendre rule `gl` in α over every arc piece
// (ew[0], ew[1] = the products of the OTHER semi-axes, ew[2] = (ew_z z)^2 of the section's height on the unit sphere)
PG_HD Sec2 disc_rect_all(double rho, double a, double b, double t0, double t1, bool want_arcs, const double* ew = nullptr,
                         const GLTable* gl = nullptr) {
  Sec2 o;
  o.area = o.mx = o.my = o.phi = o.ic = o.is = 0.0;
  if (!(rho > 0.0)) return o;
  const double rho2 = rho * rho;
  double ua = dmax(a, -rho), ub = dmin(b, rho);
  if (!(ub > ua) || !(t1 > t0)) return o;
  // kinks of the chord function: |u| = sqrt(rho^2 - t^2) for t in {t0,t1}
  double bp[6];
  int nb = 0;
  bp[nb++] = ua;
  double e[2];
  int ne = 0;
  // <=: a side that touches the circle (|t| == rho) has its double root u = 0 as a kink -- the chord limit switches
  // between the side and the arc there, and the midpoint test below would otherwise sit exactly on the tangent point
  if (fabs(t1) <= rho) e[ne++] = sroot(rho, t1);
  if (fabs(t0) <= rho) e[ne++] = sroot(rho, t0);
  double cand[4];
  int nc = 0;
  for (int k = 0; k < ne; ++k) {
    cand[nc++] = e[k];
    cand[nc++] = -e[k];
  }
  // insertion sort of the interior candidates
  for (int i = 1; i < nc; ++i) {
    double v = cand[i];
    int j = i - 1;
    while (j >= 0 && cand[j] > v) { cand[j + 1] = cand[j]; --j; }
    cand[j + 1] = v;
  }
  for (int k = 0; k < nc; ++k)
    if (cand[k] > ua && cand[k] < ub && cand[k] > bp[nb - 1]) bp[nb++] = cand[k];
  bp[nb++] = ub;
  const double inv_rho = 1.0 / rho;
  for (int k = 0; k + 1 < nb; ++k) {
    const double x0 = bp[k], x1 = bp[k + 1];
    if (!(x1 > x0)) continue;
    const double um = 0.5 * (x0 + x1);
    const double sm = sroot(rho, um);
    const bool up_arc = sm < t1;       // upper limit is the circle
    const bool lo_arc = -sm > t0;      // lower limit is the circle
    const double upper = up_arc ? sm : t1;
    const double lower = lo_arc ? -sm : t0;
    if (!(upper > lower)) continue;
    const double du = x1 - x0;
    const double du2 = 0.5 * (x1 * x1 - x0 * x0);
    const double s0 = sroot(rho, x0), s1 = sroot(rho, x1);
    double dP = 0.0, dS3 = 0.0, dR = 0.0, dAs = 0.0;
    if (up_arc || lo_arc) {
      dAs = asin_r(x1, s1) - asin_r(x0, s0);
      dP = 0.5 * ((x1 * s1 - x0 * s0) + rho2 * dAs);
      dS3 = (s0 * s0 * s0 - s1 * s1 * s1) * (1.0 / 3.0);
      dR = rho2 * du - (x1 * x1 * x1 - x0 * x0 * x0) * (1.0 / 3.0);
    }
    const double Iu = up_arc ? dP : t1 * du;        // int upper du
    const double Il = lo_arc ? -dP : t0 * du;       // int lower du
    o.area += Iu - Il;
    const double Mu = up_arc ? dS3 : t1 * du2;      // int u*upper du
    const double Ml = lo_arc ? -dS3 : t0 * du2;
    o.mx += Mu - Ml;
    const double Qu = up_arc ? dR : t1 * t1 * du;   // int upper^2 du
    const double Ql = lo_arc ? dR : t0 * t0 * du;
    o.my += 0.5 * (Qu - Ql);
    if (want_arcs && ew && (up_arc || lo_arc)) {
      const double al0 = asin_r(x0, s0), al1 = asin_r(x1, s1);
      const double am = 0.5 * (al0 + al1), ah = 0.5 * (al1 - al0);
      double p = 0.0, pc = 0.0, ps = 0.0;
      for (int q = 0; q < NGL; ++q) {
        const double al = am + ah * gl->x[q];
        const double sa = sin(al), ca = cos(al);
        const double u = rho * sa, t = rho * ca;
        const double wq = gl->w[q] * ah * sqrt(ew[0] * ew[0] * (u * u) + ew[1] * ew[1] * (t * t) + ew[2]);
        p += wq; pc += wq * sa; ps += wq * ca;
      }
      if (up_arc) { o.phi += p; o.ic += pc; o.is += ps; }
      if (lo_arc) { o.phi += p; o.ic += pc; o.is -= ps; }
    } else if (want_arcs) {
      if (up_arc) { o.phi += dAs; o.ic += (s0 - s1) * inv_rho; o.is += du * inv_rho; }
      if (lo_arc) { o.phi += dAs; o.ic += (s0 - s1) * inv_rho; o.is -= du * inv_rho; }
    }
  }
  return o;
}

// [a,b] ∩ [-rho,rho]: length and first moment
PG_HD void seg_overlap(double rho, double a, double b, double& len, double& m1) {
  const double lo = dmax(a, -rho), hi = dmin(b, rho);
  if (hi <= lo) { len = 0.0; m1 = 0.0; return; }
  len = hi - lo;
  m1 = 0.5 * (hi * hi - lo * lo);
}

struct Mom {
  double vol, m[3], gamma, gm[3];
};

// ball(c,r) ∩ box moments about the ball centre (box assumed CUT).
// qlane / qstride: this caller evaluates only the Gauss-Legendre nodes q = qlane, qlane + qstride, ... of every z piece
// (the piece structure is a function of the box alone, so cooperating lanes walk it identically); the returned moments
// are then PARTIAL sums that the caller adds up over its group.  The closed-form 1-D / 2-D cases have no quadrature:
// lane 0 returns the whole result, the others zero.  (0, 1) = everything, the host / oracle-checked form.
// ek != nullptr: unit ball in an ellipsoid's scaled coordinates (r == 1), ek[d] = prod_(k != d) ax_k: the interface
// measure and its moments are those of the ELLIPSOID's surface (in scaled coordinates about the centre)
PG_HD Mom ball_box_moments(const double* c, double r, const double* lo, const double* hi, int N,
                           bool want_surface, const GLTable& gl, int qlane = 0, int qstride = 1,
                           const double* ek = nullptr) {
  Mom o;
  o.vol = o.gamma = 0.0;
  for (int d = 0; d < 3; ++d) o.m[d] = o.gm[d] = 0.0;
  if (N < 3 && qlane != 0) return o;
  double a[3], b[3];
  for (int d = 0; d < N; ++d) { a[d] = lo[d] - c[d]; b[d] = hi[d] - c[d]; }
  if (N == 1) {
    seg_overlap(r, a[0], b[0], o.vol, o.m[0]);
    if (-r >= a[0] && -r <= b[0]) { o.gamma += 1.0; o.gm[0] += -r; }
    if (r >= a[0] && r <= b[0]) { o.gamma += 1.0; o.gm[0] += r; }
    return o;
  }
  if (N == 2) {
    double ew[3] = {0.0, 0.0, 0.0};
    if (ek) { ew[0] = ek[0]; ew[1] = ek[1]; }
    Sec2 s = disc_rect_all(r, a[0], b[0], a[1], b[1], want_surface, ek ? ew : nullptr, &gl);
    o.vol = s.area; o.m[0] = s.mx; o.m[1] = s.my;
    o.gamma = r * s.phi; o.gm[0] = r * r * s.ic; o.gm[1] = r * r * s.is;
    return o;
  }
  // N == 3
  const double r2 = r * r;
  const double z0 = dmax(a[2], -r), z1 = dmin(b[2], r);
  if (!(z1 > z0)) return o;
  // singular points of the section functions in z: rho(z) equals the distance from the axis to a
  // corner / an edge line of the rectangle, or zero (poles).  ALL of them are kept, also those
  // outside [z0,z1]: a singularity just outside a piece slows Gauss-Legendre down, so pieces are
  // bisected until the nearest other singularity is at least half a piece away.
  double S[18];
  int ns = 0;
  {
    double crit2[9];
    int ncr = 0;
    crit2[ncr++] = 0.0;
    const double xs[2] = {a[0], b[0]}, ys[2] = {a[1], b[1]};
    for (int i = 0; i < 2; ++i) {
      crit2[ncr++] = xs[i] * xs[i];
      crit2[ncr++] = ys[i] * ys[i];
      for (int j = 0; j < 2; ++j) crit2[ncr++] = xs[i] * xs[i] + ys[j] * ys[j];
    }
    for (int k = 0; k < ncr; ++k)
      if (crit2[k] < r2) {
        const double zz = sqrt(r2 - crit2[k]);
        S[ns++] = zz;
        S[ns++] = -zz;
      }
    for (int i = 1; i < ns; ++i) {
      double v = S[i];
      int j = i - 1;
      while (j >= 0 && S[j] > v) { S[j + 1] = S[j]; --j; }
      S[j + 1] = v;
    }
  }
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double zprev = z0;
  for (int ks = 0; ks <= ns; ++ks) {
    // next sub-interval boundary: interior singular point or z1
    double znext;
    if (ks < ns) {
      if (!(S[ks] > zprev)) continue;
      if (!(S[ks] < z1)) { znext = z1; ks = ns; }
      else znext = S[ks];
    } else {
      znext = z1;
    }
    if (!(znext > zprev)) continue;
    // explicit bisection stack over [zprev, znext]
    double stk_a[64], stk_b[64];
    int sp = 0;
    stk_a[0] = zprev; stk_b[0] = znext; sp = 1;
    while (sp > 0) {
      --sp;
      const double za = stk_a[sp], zb = stk_b[sp];
      const double len = zb - za;
      // distance to the nearest singular point strictly outside [za,zb]
      double dmin_out = 1e300;
      for (int q = 0; q < ns; ++q) {
        if (S[q] < za) dmin_out = dmin(dmin_out, za - S[q]);
        else if (S[q] > zb) dmin_out = dmin(dmin_out, S[q] - zb);
      }
      if (dmin_out < 0.5 * len && sp + 2 <= 64 && len > 1e-13 * r) {
        const double zc = 0.5 * (za + zb);
        stk_a[sp] = za; stk_b[sp] = zc; ++sp;
        stk_a[sp] = zc; stk_b[sp] = zb; ++sp;
        continue;
      }
      const double zm = 0.5 * (za + zb), zh = 0.5 * len;
      for (int q = qlane; q < NGL; q += qstride) {
        const double t = gl.x[q];
        const double g = 0.5 * t * (3.0 - t * t);
        const double jw = gl.w[q] * zh * 1.5 * (1.0 - t * t);
        const double z = zm + zh * g;
        const double rho = sroot(r, z);
        double ew[3] = {0.0, 0.0, 0.0};
        if (ek) { ew[0] = ek[0]; ew[1] = ek[1]; ew[2] = (ek[2] * z) * (ek[2] * z); }
        const Sec2 s = disc_rect_all(rho, a[0], b[0], a[1], b[1], want_surface, ek ? ew : nullptr, &gl);
        acc[0] += jw * s.area;
        acc[1] += jw * s.mx;
        acc[2] += jw * s.my;
        acc[3] += jw * z * s.area;
        if (want_surface) {
          acc[4] += jw * s.phi;
          acc[5] += jw * rho * s.ic;
          acc[6] += jw * rho * s.is;
          acc[7] += jw * z * s.phi;
        }
      }
    }
    zprev = znext;
  }
  o.vol = acc[0]; o.m[0] = acc[1]; o.m[1] = acc[2]; o.m[2] = acc[3];
  o.gamma = r * acc[4]; o.gm[0] = r * acc[5]; o.gm[1] = r * acc[6]; o.gm[2] = r * acc[7];
  return o;
}

PG_HD double prod_ext(const double* lo, const double* hi, int N, int skip) {
  double p = 0.0;
  bool first = true;
  for (int d = 0; d < N; ++d) {
    if (d == skip) continue;
    const double e = hi[d] - lo[d];
    p = first ? e : p * e;
    first = false;
  }
  return first ? 1.0 : p;
}

// ---- axis-aligned half space ---------------------------------------------------------------------------------
// fluid part [flo, fhi] of the interval [lo, hi] along the axis (compare-only: the oracle reproduces it bit for bit)
// with_complement = false: the type with respect to the body as given (what pick_ball returns: its callers apply the
// complement themselves, as for balls)
PG_HD int hs_interval(const BallSet& bs, double lo, double hi, double& flo, double& fhi, bool with_complement = true) {
  const bool below = (bs.sgn > 0.0) != (with_complement && bs.complement != 0);   // fluid = {x < pos} (true) or {x > pos}
  flo = lo; fhi = hi;
  if (below) {
    if (hi <= bs.pos) return PG_FULL;
    if (lo >= bs.pos) return PG_EMPTY;
    fhi = bs.pos;
  } else {
    if (lo >= bs.pos) return PG_FULL;
    if (hi <= bs.pos) return PG_EMPTY;
    flo = bs.pos;
  }
  return PG_CUT;
}

// pick the ball a box can meet (balls are pairwise disjoint): first non-EMPTY, else the last one
PG_HD int pick_ball(const BallSet& bs, const double* lo, const double* hi, int& type) {
  if (bs.kind == BODY_HALFSPACE) {
    double flo, fhi;
    type = hs_interval(bs, lo[bs.axis], hi[bs.axis], flo, fhi, false);
    return 0;
  }
  if (bs.kind == BODY_ELLIPSOID) {
    double slo[3], shi[3];
    const double zero[3] = {0.0, 0.0, 0.0};
    for (int d = 0; d < bs.N; ++d) { slo[d] = (lo[d] - bs.c[0][d]) / bs.ax[d]; shi[d] = (hi[d] - bs.c[0][d]) / bs.ax[d]; }
    type = ball_box_type(zero, 1.0, slo, shi, bs.N);
    return 0;
  }
  int t = PG_EMPTY;
  for (int s = 0; s < bs.nballs; ++s) {
    t = ball_box_type(bs.c[s], bs.r, lo, hi, bs.N);
    if (t != PG_EMPTY) { type = t; return s; }
  }
  type = t;
  return bs.nballs - 1;
}

struct NoGroup {     // single caller: nothing to add up
  PG_HD void operator()(Mom&) const {}
};

// group = functor that sums a partial Mom over the cooperating lanes (device: shuffles), see ball_box_moments
template <class Group = NoGroup>
PG_HD BoxMeasure box_measure(const BallSet& bs, const double* lo, const double* hi, bool want_surface,
                             const GLTable& gl, int qlane = 0, int qstride = 1, Group group = Group()) {
  const int N = bs.N;
  BoxMeasure o;
  o.vol = 0.0; o.gamma = 0.0;
  bool degenerate = false;
  for (int d = 0; d < 3; ++d) { o.cen[d] = 0.0; o.cg[d] = 0.0; }
  for (int d = 0; d < N; ++d) {
    o.cen[d] = 0.5 * (lo[d] + hi[d]);
    if (!(hi[d] - lo[d] > 0.0)) degenerate = true;
  }
  int t;
  const int s = pick_ball(bs, lo, hi, t);
  if (bs.kind == BODY_HALFSPACE) {
    if (bs.complement && t != PG_CUT) t = 1 - t;
    o.type = t;
    if (degenerate || t != PG_CUT) {
      o.vol = (!degenerate && t == PG_FULL) ? prod_ext(lo, hi, N, -1) : 0.0;
      return o;
    }
    double flo, fhi;   // closed form: lane 0 of a cooperating group contributes all of it
    hs_interval(bs, lo[bs.axis], hi[bs.axis], flo, fhi);
    const double cross = prod_ext(lo, hi, N, bs.axis);
    Mom m;
    m.vol = qlane == 0 ? (fhi - flo) * cross : 0.0;
    m.gamma = qlane == 0 ? cross : 0.0;
    for (int d = 0; d < 3; ++d) m.m[d] = m.gm[d] = 0.0;
    group(m);
    o.vol = m.vol;
    o.gamma = m.gamma;
    o.cen[bs.axis] = 0.5 * (flo + fhi);
    for (int d = 0; d < N; ++d) o.cg[d] = o.cen[d];
    o.cg[bs.axis] = bs.pos;
    return o;
  }
  if (degenerate || t != PG_CUT) {
    if (bs.complement && t != PG_CUT) t = 1 - t;
    o.type = t;
    o.vol = (!degenerate && t == PG_FULL) ? prod_ext(lo, hi, N, -1) : 0.0;
    return o;
  }
  const double* c = bs.c[s];
  const double full = prod_ext(lo, hi, N, -1);
  Mom m;
  if (bs.kind == BODY_ELLIPSOID) {
    double slo[3], shi[3], ek[3] = {1.0, 1.0, 1.0};
    const double zero[3] = {0.0, 0.0, 0.0};
    double J = 1.0;
    for (int d = 0; d < N; ++d) {
      slo[d] = (lo[d] - c[d]) / bs.ax[d];
      shi[d] = (hi[d] - c[d]) / bs.ax[d];
      J = J * bs.ax[d];
      for (int k = 0; k < N; ++k)
        if (k != d) ek[d] = ek[d] * bs.ax[k];
    }
    m = ball_box_moments(zero, 1.0, slo, shi, N, want_surface, gl, qlane, qstride, ek);
    m.vol = m.vol * J;                                        // back to physical measures about the centre
    for (int d = 0; d < N; ++d) { m.m[d] = m.m[d] * (J * bs.ax[d]); m.gm[d] = m.gm[d] * bs.ax[d]; }
  } else {
    m = ball_box_moments(c, bs.r, lo, hi, N, want_surface, gl, qlane, qstride);
  }
  group(m);
  if (bs.complement) {
    m.vol = full - m.vol;
    for (int d = 0; d < N; ++d) m.m[d] = full * (o.cen[d] - c[d]) - m.m[d];
  }
  o.type = PG_CUT;
  o.vol = m.vol;
  if (m.vol > 0.0)
    for (int d = 0; d < N; ++d) o.cen[d] = c[d] + m.m[d] / m.vol;
  o.gamma = m.gamma;
  if (m.gamma > 0.0)
    for (int d = 0; d < N; ++d) o.cg[d] = c[d] + m.gm[d] / m.gamma;
  return o;
}

// fluid measure of {x_d = s} ∩ box  (A_d with s = node, B_d with s = centroid coordinate)
// full_measure >= 0: the measure of a FULL section supplied by the caller (the kernels pass the product of the mesh
// spacings so that every full face / section of a uniform mesh has bitwise the same measure: A_d - B_d must be an exact
// zero between full neighbours, or a spurious 1e-17 coupling activates the γ unknown of a full cell)
PG_HD double section_measure(const BallSet& bs, int d, double s, const double* lo, const double* hi,
                             double full_measure = -1.0) {
  const int N = bs.N;
  double plo[3], phi[3];
  for (int k = 0; k < N; ++k) { plo[k] = lo[k]; phi[k] = hi[k]; }
  plo[d] = s; phi[d] = s;
  int t;
  const int sb = pick_ball(bs, plo, phi, t);
  if (bs.kind == BODY_HALFSPACE) {
    // the section is a point (N = 1), or a box of N - 1 dimensions cut by the same plane (d != axis), or lies in a
    // plane x_axis = s that is fluid or not as a whole (a section IN the interface plane counts as fluid, f <= 0, as
    // the ball's 1-D rule does)
    if (d == bs.axis) {
      double f = bs.sgn * (s - bs.pos);
      if (bs.complement) f = -f;
      if (N == 1) return f <= 0.0 ? 1.0 : 0.0;
      return f <= 0.0 ? (full_measure >= 0.0 ? full_measure : prod_ext(lo, hi, N, d)) : 0.0;
    }
    if (bs.complement && t != PG_CUT) t = 1 - t;
    if (t == PG_FULL) return full_measure >= 0.0 ? full_measure : prod_ext(lo, hi, N, d);
    if (t == PG_EMPTY) return 0.0;
    double flo, fhi;
    hs_interval(bs, lo[bs.axis], hi[bs.axis], flo, fhi);
    double m = fhi - flo;
    for (int k = 0; k < N; ++k)
      if (k != d && k != bs.axis) m = m * (hi[k] - lo[k]);
    return m;
  }
  const double* c = bs.c[sb];
  if (bs.kind == BODY_ELLIPSOID) {     // (N >= 2: a 1-D ellipsoid is parsed as a ball)
    const double full = full_measure >= 0.0 ? full_measure : prod_ext(lo, hi, N, d);
    if (t != PG_CUT) {
      if (bs.complement) t = 1 - t;
      return t == PG_FULL ? full : 0.0;
    }
    const double dz = (s - c[d]) / bs.ax[d];
    const double rho = sroot(1.0, dz);
    double a[2], b[2], scale = 1.0;
    int q = 0;
    for (int k = 0; k < N; ++k)
      if (k != d) { a[q] = (lo[k] - c[k]) / bs.ax[k]; b[q] = (hi[k] - c[k]) / bs.ax[k]; scale = scale * bs.ax[k]; ++q; }
    double m;
    if (N == 2) {
      double m1;
      seg_overlap(rho, a[0], b[0], m, m1);
    } else {
      m = disc_rect_all(rho, a[0], b[0], a[1], b[1], false).area;
    }
    m = m * scale;
    return bs.complement ? (full - m) : m;
  }
  if (N == 1) {
    double f = fabs(s - c[0]) - bs.r;
    if (bs.complement) f = -f;
    return f <= 0.0 ? 1.0 : 0.0;
  }
  const double full = full_measure >= 0.0 ? full_measure : prod_ext(lo, hi, N, d);
  if (t != PG_CUT) {
    if (bs.complement) t = 1 - t;
    return t == PG_FULL ? full : 0.0;
  }
  const double dz = s - c[d];
  const double rho2 = bs.r * bs.r - dz * dz;
  const double rho = rho2 > 0.0 ? sqrt(rho2) : 0.0;
  double a[2], b[2];
  int q = 0;
  for (int k = 0; k < N; ++k)
    if (k != d) { a[q] = lo[k] - c[k]; b[q] = hi[k] - c[k]; ++q; }
  double m;
  if (N == 2) {
    double m1;
    seg_overlap(rho, a[0], b[0], m, m1);
  } else {
    m = disc_rect_all(rho, a[0], b[0], a[1], b[1], false).area;
  }
  return bs.complement ? (full - m) : m;
}

// Gauss-Legendre nodes/weights on [-1,1] (host side; Newton on P_n)
inline void gl_init(GLTable& gl) {
  const int n = NGL;
  for (int i = 0; i < (n + 1) / 2; ++i) {
    double z = cos(M_PI * (i + 0.75) / (n + 0.5));
    double pp = 0.0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 0; j < n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      const double z1 = z;
      z = z1 - p1 / pp;
      if (fabs(z - z1) < 1e-16) break;
    }
    gl.x[i] = -z;
    gl.x[n - 1 - i] = z;
    gl.w[i] = 2.0 / ((1.0 - z * z) * pp * pp);
    gl.w[n - 1 - i] = gl.w[i];
  }
}

}  // namespace pggeom
