// AddressSanitizer + UndefinedBehaviorSanitizer build of the library's host-side algorithms (sanitizers
// run on the CPU build only).  Compiled and run by tests/test_host_sanitizers.py:
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tests/host_asan.cpp oracle/krylov_ref.c
// Covers, with seeded random inputs and functional checks (exit code 0 = all passed):
//   1. pghost::partition_planes            -- the slab partition behind pg_partition_planes / the multi-GPU set-up
//   2. pghost::plan_march_units            -- the SpMV's marching-unit planner: a compacted cut-cell numbering of a ball is
//                                             synthesised, the planned records are EXECUTED by a host emulation of the kernel's
//                                             data flow and the result compared with the stencil applied row by row
//   3. pggeom::box_measure / section_measure (pg_geom.h, the device functions of the capacity kernels) for balls, unions,
//                                             complements and half spaces on random and degenerate boxes
//   4. oracle/krylov_ref.c                 -- the C restatement of the Krylov loops on a random diagonally dominant system
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../penguin/jl_amd/csrc/pg_geom.h"
#include "../penguin/jl_amd/csrc/pg_host_algos.h"

extern "C" {
int krylov_ref_bicgstab(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* b, double* x, double reltol,
                        double abstol, int maxiter, int nthreads, double* resnorm_out);
int krylov_ref_bicgstab_poly(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* b, double* x,
                             const double* x0, const double* wts, int m, double g, double reltol, double abstol, int maxiter,
                             int nthreads, double* resnorm_out, int64_t* nmv_out);
int krylov_ref_cg(int64_t n, const int64_t* rp, const int32_t* ci, const double* v, const double* b, double* x, double reltol,
                  double abstol, int maxiter, int nthreads, double* resnorm_out);
}

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      ++g_fail;                                           \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                       \
      fprintf(stderr, "\n");                              \
    }                                                     \
  } while (0)

// ---------------------------------------------------------------------------------------------- 1. partition
static void test_partition(std::mt19937_64& rng) {
  for (int rep = 0; rep < 200; ++rep) {
    const int nranks = 1 + (int)(rng() % 8);
    const int64_t nplanes = nranks + (int64_t)(rng() % 600);
    std::vector<int64_t> w(nplanes), b(nranks + 1, -1);
    const int shape = (int)(rng() % 4);
    for (int64_t k = 0; k < nplanes; ++k)
      w[k] = shape == 0 ? 0 : shape == 1 ? (int64_t)(rng() % 1000) : shape == 2 ? (k < nplanes / 3 ? 0 : 100000) : (int64_t)(rng() % 3 == 0 ? 1000000 : 0);
    pghost::partition_planes(w.data(), nplanes, nranks, b.data());
    CHECK(b[0] == 0 && b[nranks] == nplanes, "partition ends");
    for (int r = 0; r < nranks; ++r) CHECK(b[r + 1] > b[r], "rank %d owns no plane", r);
  }
}

// ---------------------------------------------------------------------------------------------- 2. marching units
// a ball of radius R cells in an n^3 grid, active cells numbered x fastest (the reduced numbering); a row is "uniform" when
// all six neighbours are active; uniform rows of one line with one set of offsets form a run
struct Synth {
  int n;
  std::vector<int> id;   // cell -> row or -1
  std::vector<pghost::MRun> runs;
  std::vector<int> info;
  int64_t nrows = 0;
};
static Synth synth_ball(int n, double R, double cx, double cy, double cz, bool two_d) {
  Synth s;
  s.n = n;
  const int nz = two_d ? 1 : n;
  s.id.assign((size_t)n * n * nz, -1);
  auto at = [&](int i, int j, int k) -> int& { return s.id[(size_t)i + (size_t)n * (j + (size_t)n * k)]; };
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        const double dx = i + 0.5 - cx, dy = j + 0.5 - cy, dz = two_d ? 0.0 : k + 0.5 - cz;
        if (dx * dx + dy * dy + dz * dz < R * R) at(i, j, k) = (int)s.nrows++;
      }
  const int cnt = two_d ? 5 : 7;
  auto offsets = [&](int i, int j, int k, int* o) {   // entry order +1, -1, [+Y, -Y,] +Z, -Z, 0  (2-D: +1, -1, +Y, -Y, 0)
    auto nb = [&](int a, int b, int c) { return (a < 0 || b < 0 || c < 0 || a >= n || b >= n || c >= nz) ? -1 : at(a, b, c); };
    const int r = at(i, j, k);
    const int q[6] = {nb(i + 1, j, k), nb(i - 1, j, k), nb(i, j + 1, k), nb(i, j - 1, k), nb(i, j, k + 1), nb(i, j, k - 1)};
    const int m = two_d ? 4 : 6;
    for (int e = 0; e < m; ++e) {
      if (q[e] < 0) return false;
      o[e] = q[e] - r;
    }
    o[m] = 0;
    return true;
  };
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < n; ++j) {
      int run_start = -1, prev[8] = {0};
      for (int i = 0; i <= n; ++i) {
        int o[8] = {0};
        const bool uni = i < n && at(i, j, k) >= 0 && offsets(i, j, k, o);
        const bool same = uni && run_start >= 0 && std::memcmp(o, prev, sizeof(int) * cnt) == 0;
        if (run_start >= 0 && !same) {
          const int r0 = at(run_start, j, k), len = i - run_start;
          if (len >= 3) {
            s.runs.push_back(pghost::MRun{r0, len, cnt});
            for (int e = 0; e < 8; ++e) s.info.push_back(e < cnt ? prev[e] : 0);
            for (int e = 0; e < 8; ++e) {   // values: the same for every run (one coefficient per entry), as lo / hi dwords
              const double v = e < cnt ? (e + 1 == cnt ? 1.0 : -0.1 - 0.01 * e) : 0.0;
              int w[2];
              std::memcpy(w, &v, 8);
              s.info.push_back(w[0]);
              s.info.push_back(w[1]);
            }
          }
          run_start = -1;
        }
        if (uni && run_start < 0) run_start = i;
        if (uni) std::memcpy(prev, o, sizeof(prev));
      }
    }
  return s;
}

static void test_march(std::mt19937_64& rng) {
  const pghost::MarchGeometry geos[] = {{4, 2, 64, 126, 24, 4}, {4, 2, 64, 126, 24, 3}, {6, 2, 64, 126, 24, 6}, {4, 2, 64, 126, 24, 1}};
  for (int rep = 0; rep < 10; ++rep) {
    const bool two_d = rep % 3 == 2;
    const int n = two_d ? 300 + (int)(rng() % 200) : 40 + (int)(rng() % 30);
    std::uniform_real_distribution<double> U(-3.0, 3.0);
    const double R = two_d ? 0.3 * n + U(rng) : 0.42 * n + U(rng);
    Synth s = synth_ball(n, R, 0.5 * n + U(rng), 0.5 * n + U(rng), 0.5 * n + U(rng), two_d);
    const pghost::MarchGeometry& geo = geos[rep % 4];
    std::vector<int> mrec;
    std::vector<pghost::RowRange> fb;
    int64_t rows_m = 0;
    pghost::plan_march_units(s.nrows, s.runs, s.info, geo, mrec, fb, rows_m);
    const int64_t nunits = (int64_t)mrec.size() / geo.REC;
    CHECK(nunits > 0 && rows_m > 0, "no units planned (n = %d, runs = %zu)", n, s.runs.size());
    {   // the tile table over the units' keys and some slices' keys: 8 parts x T tiles, monotone, ends on the part's ends
      std::vector<int> mrec2;
      std::vector<pghost::RowRange> fb2;
      std::vector<int64_t> uk, sk;
      int64_t rm2 = 0;
      pghost::plan_march_units(s.nrows, s.runs, s.info, geo, mrec2, fb2, rm2, &uk);
      CHECK((int64_t)uk.size() == nunits && std::is_sorted(uk.begin(), uk.end()), "unit keys: %zu of %lld", uk.size(), (long long)nunits);
      for (int q = 0, m = (int)(rng() % 4000); q < m; ++q) sk.push_back((int64_t)(rng() % (uint64_t)(s.nrows + 1)));
      std::sort(sk.begin(), sk.end());
      const int T = 1 + (int)(rng() % 7);
      std::vector<int> tab;
      pghost::plan_tiles(uk, sk, T, tab);
      CHECK((int)tab.size() == 8 * (T + 1) * 2, "tile table size %zu", tab.size());
      const int64_t nu = nunits, ns = (int64_t)sk.size();
      for (int q = 0; q < 8; ++q) {
        const int* tb = tab.data() + (size_t)q * (T + 1) * 2;
        CHECK(tb[0] == nu * q / 8 && tb[2 * T] == nu * (q + 1) / 8 && tb[1] == ns * q / 8 && tb[2 * T + 1] == ns * (q + 1) / 8,
              "tile table: part %d does not end on the part's ends", q);
        for (int t = 0; t < T; ++t)
          CHECK(tb[2 * t] <= tb[2 * t + 2] && tb[2 * t + 1] <= tb[2 * t + 3], "tile table: part %d tile %d not monotone", q, t);
      }
    }
    // execute the records as the kernel does (lane l holds elements 2l, 2l+1 of a 128-element window of every line) on a
    // vector with 8 elements of slack, and compare with the stencil applied row by row
    std::vector<double> x(s.nrows + 8), y(s.nrows, 0.0), yref(s.nrows, 0.0);
    std::vector<int> covered(s.nrows, 0);
    std::uniform_real_distribution<double> V(-1.0, 1.0);
    for (auto& v : x) v = V(rng);
    for (size_t q = 0; q < s.runs.size(); ++q) {
      const pghost::MRun& r = s.runs[q];
      for (int l = 0; l < r.len; ++l) {
        double acc = 0.0;
        for (int e = 0; e < r.cnt; ++e) {
          double c;
          int w[2] = {s.info[24 * q + 8 + 2 * e], s.info[24 * q + 9 + 2 * e]};
          std::memcpy(&c, w, 8);
          acc += c * x[r.r0 + l + s.info[24 * q + e]];
        }
        yref[r.r0 + l] = acc;
      }
    }
    auto xat = [&](int64_t idx) -> double {
      CHECK(idx >= 0 && idx < s.nrows + 8, "unit reads element %lld of a vector of %lld (+8)", (long long)idx, (long long)s.nrows);
      return idx >= 0 && idx < s.nrows + 8 ? x[idx] : 0.0;
    };
    for (int64_t u = 0; u < nunits; ++u) {
      const int* rec = mrec.data() + geo.REC * u;
      const int K = rec[0] & 255, cnt = (rec[0] >> 8) & 255, lanes = rec[0] >> 16;
      CHECK(K == geo.K || K == geo.KS, "unit size %d", K);
      CHECK(lanes >= 1 && lanes <= 64, "active lanes %d", lanes);
      double c[8];
      for (int e = 0; e < cnt; ++e) std::memcpy(&c[e], &rec[2 + 2 * e], 8);
      const bool Y = cnt == 7;
      for (int i = 0; i < K; ++i) {
        const int rb = rec[18 + 4 * i], lo = rec[21 + 4 * i] & 255, hi = rec[21 + 4 * i] >> 8;
        const int rb_dn = i == 0 ? rec[16] : rec[18 + 4 * (i - 1)], rb_up = i + 1 < K ? rec[18 + 4 * (i + 1)] : rec[17];
        CHECK(lo >= 1 && hi <= 127 && lo <= hi, "range [%d, %d)", lo, hi);
        for (int l = 0; l < 64; ++l) {
          // every ACTIVE lane loads both elements of its pair in every line it touches
          if (l < lanes) {
            for (int e = 0; e < 2; ++e) {
              (void)xat((int64_t)rb + 2 * l + e); (void)xat((int64_t)rb_dn + 2 * l + e); (void)xat((int64_t)rb_up + 2 * l + e);
              if (Y) { (void)xat((int64_t)rb + rec[19 + 4 * i] + 2 * l + 1 + e); (void)xat((int64_t)rb + rec[20 + 4 * i] + 2 * l + 1 + e); }
            }
          }
          for (int e = 1; e <= 2; ++e) {
            const int cw = 2 * l + e;                 // window-relative row
            if (cw < lo || cw >= hi) continue;
            CHECK(l < lanes && (cw + 1) / 2 < lanes + (cw + 1 <= hi ? 0 : 0), "row %d computed by a masked lane (lanes = %d)", cw, lanes);
            CHECK((cw + 1) / 2 < lanes, "row %d needs element %d of lane %d >= %d active lanes", cw, cw + 1, (cw + 1) / 2, lanes);
            const int64_t row = (int64_t)rb + cw;
            double acc = 0.0;
            int j = 0;
            acc += c[j++] * xat(row + 1);
            acc += c[j++] * xat(row - 1);
            if (Y) {
              acc += c[j++] * xat(row + rec[20 + 4 * i]);
              acc += c[j++] * xat(row + rec[19 + 4 * i]);
            }
            acc += c[j++] * xat((int64_t)rb_up + cw);
            acc += c[j++] * xat((int64_t)rb_dn + cw);
            acc += c[j++] * xat(row);
            CHECK(row >= 0 && row < s.nrows, "unit writes row %lld", (long long)row);
            if (row >= 0 && row < s.nrows) { y[row] = acc; covered[row] += 1; }
          }
        }
      }
    }
    for (const auto& f : fb)
      for (int64_t r = f.a; r < f.b; ++r) { covered[r] += 1; y[r] = yref[r]; }
    int64_t total = 0;
    for (const auto& r : s.runs) {
      total += r.len;
      for (int l = 0; l < r.len; ++l) {
        CHECK(covered[r.r0 + l] == 1, "row %d covered %d times", r.r0 + l, covered[r.r0 + l]);
        CHECK(y[r.r0 + l] == yref[r.r0 + l], "row %d: %.17g != %.17g", r.r0 + l, y[r.r0 + l], yref[r.r0 + l]);
        if (g_fail > 20) return;
      }
    }
    int64_t fbrows = 0;
    for (const auto& f : fb) fbrows += f.b - f.a;
    CHECK(rows_m + fbrows == total, "rows marched %lld + fallback %lld != %lld", (long long)rows_m, (long long)fbrows, (long long)total);
  }
}

// ---------------------------------------------------------------------------------------------- 3. geometry
static void test_geometry(std::mt19937_64& rng) {
  using namespace pggeom;
  GLTable gl;
  gl_init(gl);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (int rep = 0; rep < 3000; ++rep) {
    BallSet bs;
    std::memset(&bs, 0, sizeof(bs));
    bs.N = 1 + (int)(rng() % 3);
    bs.complement = (int)(rng() % 2);
    const bool hs = rep % 4 == 3;
    if (hs) {
      bs.kind = BODY_HALFSPACE; bs.axis = (int)(rng() % bs.N); bs.pos = U(rng); bs.sgn = rng() % 2 ? 1.0 : -1.0; bs.nballs = 1; bs.r = 1.0;
    } else if (rep % 4 == 2 && bs.N > 1) {
      bs.kind = BODY_ELLIPSOID; bs.nballs = 1; bs.r = 1.0;
      for (int d = 0; d < 3; ++d) { bs.c[0][d] = U(rng); bs.ax[d] = 0.05 + 0.6 * U(rng); }
    } else {
      bs.kind = BODY_BALLS; bs.nballs = 1 + (int)(rng() % 3); bs.r = 0.05 + 0.3 * U(rng);
      if (bs.nballs > 1) bs.complement = 0;
      for (int s = 0; s < bs.nballs; ++s)
        for (int d = 0; d < 3; ++d) bs.c[s][d] = s * 1.5 + U(rng);
    }
    double lo[3], hi[3];
    const int mode = (int)(rng() % 6);
    for (int d = 0; d < 3; ++d) {
      const double a = U(rng) * (hs ? 1.0 : 1.2), h = mode == 0 ? 0.0 : (mode == 1 ? 1e-12 : 0.02 + 0.2 * U(rng));
      lo[d] = a; hi[d] = a + h;
    }
    if (mode == 2 && !hs) for (int d = 0; d < bs.N; ++d) { lo[d] = bs.c[0][d]; hi[d] = bs.c[0][d] + bs.r * (bs.kind == BODY_ELLIPSOID ? bs.ax[d] : 1.0); }   // corner on the surface
    if (mode == 3 && hs) { hi[bs.axis] = bs.pos + (hi[bs.axis] - lo[bs.axis]); lo[bs.axis] = bs.pos; }      // face in the plane
    const BoxMeasure m = box_measure(bs, lo, hi, true, gl);
    double full = 1.0;
    for (int d = 0; d < bs.N; ++d) full *= hi[d] - lo[d];
    CHECK(std::isfinite(m.vol) && m.vol >= -1e-15 && m.vol <= full * (1 + 1e-9) + 1e-300, "vol %g of a box of %g", m.vol, full);
    CHECK(std::isfinite(m.gamma) && m.gamma >= 0.0, "gamma %g", m.gamma);
    CHECK(m.type == PG_FULL || m.type == PG_EMPTY || m.type == PG_CUT, "type %d", m.type);
    for (int d = 0; d < bs.N; ++d) {
      CHECK(std::isfinite(m.cen[d]) && std::isfinite(m.cg[d]), "centroid");
      const double sm = section_measure(bs, d, lo[d] + 0.3 * (hi[d] - lo[d]), lo, hi);
      double fs = 1.0;
      for (int k = 0; k < bs.N; ++k) if (k != d) fs *= hi[k] - lo[k];
      CHECK(std::isfinite(sm) && sm >= -1e-15 && sm <= fs * (1 + 1e-9) + 1e-300, "section %g of %g", sm, fs);
    }
  }
}

// ---------------------------------------------------------------------------------------------- 4. Krylov restatement
static void test_krylov(std::mt19937_64& rng) {
  const int64_t n = 400;
  std::vector<int64_t> rp(n + 1, 0);
  std::vector<int32_t> ci;
  std::vector<double> v, b(n), x(n), w(n, 1.0);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int64_t r = 0; r < n; ++r) {
    const int64_t nb[3] = {r - 1, r, r + 1};
    for (int64_t c : nb) {
      if (c < 0 || c >= n) continue;
      ci.push_back((int32_t)c);
      v.push_back(c == r ? 1.0 : -0.3 + 0.05 * U(rng));
    }
    rp[r + 1] = (int64_t)ci.size();
    b[r] = U(rng);
    w[r] = 0.5 + std::fabs(U(rng));
  }
  auto resid = [&](const std::vector<double>& xx) {
    double s = 0.0, bb = 0.0;
    for (int64_t r = 0; r < n; ++r) {
      double a = b[r];
      for (int64_t k = rp[r]; k < rp[r + 1]; ++k) a -= v[k] * xx[ci[k]];
      s += a * a; bb += b[r] * b[r];
    }
    return std::sqrt(s / bb);
  };
  double rn;
  int64_t nmv;
  int it = krylov_ref_bicgstab(n, rp.data(), ci.data(), v.data(), b.data(), x.data(), 1e-12, 0.0, 500, 1, &rn);
  CHECK(it > 0 && resid(x) < 1e-10, "bicgstab residual %g after %d", resid(x), it);
  for (int m : {0, 2, 3, 6, 16}) {
    it = krylov_ref_bicgstab_poly(n, rp.data(), ci.data(), v.data(), b.data(), x.data(), nullptr, w.data(), m, 0.7, 1e-12, 0.0, 500, 1, &rn, &nmv);
    CHECK(it > 0 && resid(x) < 1e-9, "bicgstab_poly(m = %d) residual %g after %d", m, resid(x), it);
    std::vector<double> x0(x), x2(n);
    for (auto& e : x0) e *= 1.0 + 1e-5;
    const int it2 = krylov_ref_bicgstab_poly(n, rp.data(), ci.data(), v.data(), b.data(), x2.data(), x0.data(), w.data(), m, 0.7, 1e-12, 0.0, 500, 1, &rn, &nmv);
    CHECK(it2 <= it && resid(x2) < 1e-9, "warm start: %d iterations, residual %g", it2, resid(x2));
  }
  // symmetrise for CG: A + A^T has the same pattern here
  for (int64_t r = 0; r < n; ++r)
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) if (ci[k] != r) v[k] = -0.3;
  it = krylov_ref_cg(n, rp.data(), ci.data(), v.data(), b.data(), x.data(), 1e-12, 0.0, 500, 1, &rn);
  CHECK(it > 0 && resid(x) < 1e-10, "cg residual %g", resid(x));
}

int main() {
  std::mt19937_64 rng(20261004);
  test_partition(rng);
  test_march(rng);
  test_geometry(rng);
  test_krylov(rng);
  if (g_fail) {
    fprintf(stderr, "%d check(s) failed\n", g_fail);
    return 1;
  }
  printf("host_asan: all checks passed\n");
  return 0;
}
