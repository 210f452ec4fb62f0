// Host (g++) build of the geometry device functions, for CPU unit tests only (tests/test_geom_host.py).
// The product never loads this library.
#include "../penguin/jl_amd/csrc/pg_geom.h"
using namespace pggeom;
static GLTable g_gl;
static bool g_init = false;
static void init() { if (!g_init) { gl_init(g_gl); g_init = true; } }
extern "C" {
// out: type, vol, cen[3], gamma, cg[3]  (9 doubles)
void geom_box(int N, int nballs, int complement, double r, const double* centers, const double* lo,
              const double* hi, int want_surface, double* out) {
  init();
  BallSet bs; bs.kind = BODY_BALLS; bs.ax[0] = bs.ax[1] = bs.ax[2] = 1.0; bs.axis = 0; bs.pos = 0.0; bs.sgn = 1.0; bs.N = N; bs.nballs = nballs; bs.complement = complement; bs.r = r;
  for (int s = 0; s < nballs; ++s) for (int d = 0; d < N; ++d) bs.c[s][d] = centers[s * N + d];
  BoxMeasure m = box_measure(bs, lo, hi, want_surface != 0, g_gl);
  out[0] = m.type; out[1] = m.vol; out[2] = m.cen[0]; out[3] = m.cen[1]; out[4] = m.cen[2];
  out[5] = m.gamma; out[6] = m.cg[0]; out[7] = m.cg[1]; out[8] = m.cg[2];
}
double geom_section(int N, int nballs, int complement, double r, const double* centers, int d, double s,
                    const double* lo, const double* hi) {
  init();
  BallSet bs; bs.kind = BODY_BALLS; bs.ax[0] = bs.ax[1] = bs.ax[2] = 1.0; bs.axis = 0; bs.pos = 0.0; bs.sgn = 1.0; bs.N = N; bs.nballs = nballs; bs.complement = complement; bs.r = r;
  for (int k = 0; k < nballs; ++k) for (int q = 0; q < N; ++q) bs.c[k][q] = centers[k * N + q];
  return section_measure(bs, d, s, lo, hi);
}
// ellipsoid: centre c[N], semi-axes ax[N];  out as geom_box
void geom_ellipsoid_box(int N, int complement, const double* c, const double* ax, const double* lo, const double* hi,
                        int want_surface, double* out) {
  init();
  BallSet bs; bs.kind = BODY_ELLIPSOID; bs.axis = 0; bs.pos = 0.0; bs.sgn = 1.0; bs.N = N; bs.nballs = 1; bs.complement = complement; bs.r = 1.0;
  for (int d = 0; d < 3; ++d) { bs.c[0][d] = d < N ? c[d] : 0.0; bs.ax[d] = d < N ? ax[d] : 1.0; }
  BoxMeasure m = box_measure(bs, lo, hi, want_surface != 0, g_gl);
  out[0] = m.type; out[1] = m.vol; out[2] = m.cen[0]; out[3] = m.cen[1]; out[4] = m.cen[2];
  out[5] = m.gamma; out[6] = m.cg[0]; out[7] = m.cg[1]; out[8] = m.cg[2];
}
double geom_ellipsoid_section(int N, int complement, const double* c, const double* ax, int d, double s, const double* lo,
                              const double* hi) {
  init();
  BallSet bs; bs.kind = BODY_ELLIPSOID; bs.axis = 0; bs.pos = 0.0; bs.sgn = 1.0; bs.N = N; bs.nballs = 1; bs.complement = complement; bs.r = 1.0;
  for (int k = 0; k < 3; ++k) { bs.c[0][k] = k < N ? c[k] : 0.0; bs.ax[k] = k < N ? ax[k] : 1.0; }
  return section_measure(bs, d, s, lo, hi);
}
}
