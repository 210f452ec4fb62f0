// pg_host_algos.h -- the pure HOST algorithms of the library, free of HIP so that the CPU test-suite can compile them with
// AddressSanitizer + UndefinedBehaviorSanitizer (tests/host_asan.cpp): the slab partition of the
// planes and the planning of the SpMV's marching units (index arithmetic over run descriptors: the place where an
// off-by-one reads outside a vector on the GPU).  Included by pg_context.hip and pg_spmv.hip.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace pghost {

// contiguous plane ranges whose cumulated weight is closest to r / nranks of the total; every rank gets at least one
// plane (+1 per plane so that empty regions are still spread).  bounds: nranks + 1 entries.  nplanes >= nranks >= 1.
inline void partition_planes(const int64_t* weight, int64_t nplanes, int nranks, int64_t* bounds) {
  std::vector<double> cum(nplanes + 1, 0.0);
  for (int64_t k = 0; k < nplanes; ++k) cum[k + 1] = cum[k] + static_cast<double>(weight[k]) + 1.0;
  const double total = cum[nplanes];
  bounds[0] = 0;
  for (int r = 1; r < nranks; ++r) {
    const double target = total * r / nranks;
    int64_t k = std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();
    if (k > 0 && (target - cum[k - 1]) < (cum[k] - target)) --k;
    const int64_t lo = bounds[r - 1] + 1;
    const int64_t hi = nplanes - (nranks - r);
    bounds[r] = std::min(std::max(k, lo), hi);
  }
  bounds[nranks] = nplanes;
}

// ---- local numbering of one slab (K10, the layout pg_comm.hip's halo exchange rests on) -----------------------------
// Local vector layout: [kind 0 owned | kind 1 owned | ... | lower ghosts: kind 0, kind 1, ... | upper ghosts: kind 0, ...].
// Owned unknowns of a kind are ordered by padded cell index, so the actives of the first / last owned plane are the
// contiguous chunks at the start / end of each kind's segment: they are what goes to the lower / upper neighbour, and what
// arrives from a neighbour lands in one contiguous ghost segment per kind -- no pack / unpack kernels.
// Inputs per kind k: the number of active unknowns of the stored planes that lie BEFORE the first owned plane (posO), before
// the end of the last owned plane (posU), before the end of the first owned plane (posFE), before the start of the last
// owned plane (posLB), and in all stored planes (tot); actives outside the one ghost plane each side do not exist (their
// flags are cleared before the count).  build_numbering (device scans) and pg_slab_numbering_host (CPU prefix sums: the
// gloo test of the CPU suite) both end here.
constexpr int SLAB_MAX_KINDS = 4;
struct SlabNumbering {
  int K = 0;
  int64_t n_own = 0, n_ghost = 0;
  int64_t cnt_own[SLAB_MAX_KINDS], off_own[SLAB_MAX_KINDS], cntL[SLAB_MAX_KINDS], offL[SLAB_MAX_KINDS], cntU[SLAB_MAX_KINDS],
      offU[SLAB_MAX_KINDS], sendL_off[SLAB_MAX_KINDS], sendL_cnt[SLAB_MAX_KINDS], sendU_off[SLAB_MAX_KINDS], sendU_cnt[SLAB_MAX_KINDS];
};
inline void slab_numbering(int K, const int64_t* posO, const int64_t* posU, const int64_t* posFE, const int64_t* posLB,
                           const int64_t* tot, bool has_lower, bool has_upper, SlabNumbering& out) {
  out.K = K;
  int64_t off = 0;
  for (int k = 0; k < K; ++k) {
    out.cntL[k] = posO[k];
    out.cnt_own[k] = posU[k] - posO[k];
    out.cntU[k] = tot[k] - posU[k];
    out.off_own[k] = off;
    off += out.cnt_own[k];
  }
  out.n_own = off;
  for (int k = 0; k < K; ++k) { out.offL[k] = off; off += out.cntL[k]; }
  for (int k = 0; k < K; ++k) { out.offU[k] = off; off += out.cntU[k]; }
  out.n_ghost = off - out.n_own;
  for (int k = 0; k < K; ++k) {
    // owned actives of the first owned plane go to the lower neighbour, of the last to the upper one
    out.sendL_off[k] = out.off_own[k];
    out.sendL_cnt[k] = has_lower ? posFE[k] - posO[k] : 0;
    out.sendU_cnt[k] = has_upper ? posU[k] - posLB[k] : 0;
    out.sendU_off[k] = out.off_own[k] + out.cnt_own[k] - out.sendU_cnt[k];
  }
  for (int k = K; k < SLAB_MAX_KINDS; ++k)
    out.cnt_own[k] = out.off_own[k] = out.cntL[k] = out.offL[k] = out.cntU[k] = out.offU[k] = out.sendL_off[k] = out.sendL_cnt[k] =
        out.sendU_off[k] = out.sendU_cnt[k] = 0;
}

// stored / exact-flag plane ranges of a slab that owns planes [p0, p1): stored = [p0 - halo, p1 + halo) clipped, flags are
// exact on [p0 - 1, p1 + 1) clipped (one ghost plane each side)
struct SlabPlanes { int64_t s0, s1, a0, a1; };
inline SlabPlanes slab_planes(int64_t p0, int64_t p1, int64_t nplanes, int halo) {
  SlabPlanes g;
  g.s0 = p0 - halo < 0 ? 0 : p0 - halo;
  g.s1 = p1 + halo > nplanes ? nplanes : p1 + halo;
  g.a0 = p0 - 1 < 0 ? 0 : p0 - 1;
  g.a1 = p1 + 1 > nplanes ? nplanes : p1 + 1;
  return g;
}

struct MRun {
  int r0, len, cnt;   // rows [r0, r0 + len) with one stencil (offsets and values), cnt = 5 / 7 entries
};
struct RowRange {
  int64_t a, b;       // rows [a, b) that cannot march: they go back to the U slices
  int cnt;
};
struct MarchGeometry {
  int K, KS;          // planes per full / short unit (shorter units are closed with empty planes)
  int REC;            // dwords per unit record
  int W;              // rows computed per window (the window holds W + 2 elements of every line)
  int INFO;           // dwords per run descriptor: 8 offsets, 8 values (lo, hi)
  int kmax;           // planes per unit actually used (<= K)
  // optional grid geometry (3-D units only): dword 24 of a run descriptor is then the grid cell of the run's first row,
  // cell = (plane * lines + line) * ext0 + i, and the units are ordered strip by strip (see below)
  int64_t ext0 = 0, lines = 0;
  int strip = 0;      // lateral lines per strip (0: order by first row)
  // 1 (needs the grid geometry): units are cut at COMMON planes (every unit starts at a multiple of kmax or at the start of
  // its chain) and ordered window-major inside a (strip, plane group): the four waves of a block then hold four
  // neighbouring lines of one window over the same planes, and a unit's lateral lines are its block mates' own lines
  int unit_order = 0;
};

// place of a grid cell in the order of the work items: strip of lateral lines, then plane; ties by row number
inline int64_t march_order(int64_t cell, int64_t ext0, int64_t lines, int strip) {
  const int64_t line = (cell / ext0) % lines, plane = cell / (ext0 * lines);
  return (line / strip) * ((int64_t)1 << 20) + plane;
}

// Chains of runs that face each other across the slowest stencil direction (entry cnt-3 / cnt-2 of a row lead to the row of
// the same lateral position one plane up / down), cut into windows of W computed rows and units of <= kmax planes
// (pg_spmv.hip "marching units" has the record layout).  info: INFO dwords per run -- the col - row offsets (8) and the
// values (8 doubles as lo / hi dwords) of the run's first row, in the entry order of the assembled rows:
// +1, -1, [+Y, -Y,] +Z, -Z, 0.  n = rows of the matrix (= guaranteed length of every vector the kernel reads).
// mrec: the unit records, sorted by first row -- or, when the grid geometry is known, 3-D units strip by strip: all planes
// of `strip` neighbouring lateral lines before the next strip.  A unit reads the line below its first and above its last
// plane (6 lines of the chain for 4 computed ones); in first-row order a plane of units lies between a unit and the unit
// above it in the chain, which is more than an XCD's L2 holds at 512^3 (PMC: x crossed the fabric 1.54 times per launch),
// in strip order it is strip x windows units.  fallback: rows that cannot march (another entry order, windows that would read
// outside the vector); rows_m: rows covered by units.
inline void plan_march_units(int64_t n, const std::vector<MRun>& runs, const std::vector<int>& info, const MarchGeometry& geo,
                             std::vector<int>& mrec, std::vector<RowRange>& fallback, int64_t& rows_m,
                             std::vector<int64_t>* okeys = nullptr) {
  typedef int64_t i64;
  rows_m = 0;
  const i64 nr = (i64)runs.size();
  const int RI = geo.INFO;
  auto off = [&](i64 i, int k) { return info[RI * i + k]; };
  auto same_values = [&](i64 i, i64 j, int cnt) {
    for (int k = 0; k < 2 * cnt; ++k)
      if (info[RI * i + 8 + k] != info[RI * j + 8 + k]) return false;
    return true;
  };
  // entries of a marching row, in the order eval_row emits them: +1, -1, [+Y, -Y,] +Z, -Z, 0 with 1 < Y < Z
  std::vector<char> ok(nr, 0);
  for (i64 i = 0; i < nr; ++i) {
    const int cnt = runs[i].cnt;
    bool good = (cnt == 5 || cnt == 7) && off(i, 0) == 1 && off(i, 1) == -1 && off(i, cnt - 1) == 0;
    if (good) {
      const int up_e = cnt - 3, dn_e = cnt - 2;
      good = off(i, up_e) > 1 && off(i, dn_e) < -1;
      if (good && cnt == 7) good = off(i, 2) > 1 && off(i, 2) < off(i, up_e) && off(i, 3) < -1 && off(i, 3) > off(i, dn_e);
    }
    ok[i] = good;
    if (!ok[i]) fallback.push_back(RowRange{runs[i].r0, (i64)runs[i].r0 + runs[i].len, cnt});
  }
  // successor of run i: the run that holds most of the rows r + o[up], r in run i, if it mirrors the offset and carries
  // the same values; every run has at most one predecessor
  std::vector<int> succ(nr, -1), pred(nr, -1);
  for (i64 i = 0; i < nr; ++i) {
    if (!ok[i]) continue;
    const int cnt = runs[i].cnt, up_off = off(i, cnt - 3);
    const i64 lo = (i64)runs[i].r0 + up_off, hi = lo + runs[i].len;
    i64 j = std::upper_bound(runs.begin(), runs.end(), lo, [](i64 v, const MRun& r) { return v < (i64)r.r0; }) - runs.begin();
    if (j > 0) --j;
    i64 best = -1, best_ov = 0;
    for (; j < nr && (i64)runs[j].r0 < hi; ++j) {
      const i64 a = runs[j].r0, b = a + runs[j].len;
      if (b <= lo || j == i || !ok[j] || pred[j] >= 0 || runs[j].cnt != cnt || off(j, cnt - 2) != -up_off || !same_values(i, j, cnt)) continue;
      const i64 ov = std::min(hi, b) - std::max(lo, a);
      if (ov > best_ov) { best_ov = ov; best = j; }
    }
    if (best >= 0) { succ[i] = (int)best; pred[best] = (int)i; }
  }
  struct Unit { int key; int64_t order; int64_t sub; std::vector<int> rec; };
  const bool aligned = geo.unit_order == 1 && geo.strip > 0 && geo.ext0 > 0 && geo.lines > 0 && RI > 24;
  // absolute plane of a run (-1: unknown)
  auto plane_of = [&](i64 run) -> i64 {
    const i64 cell = info[RI * run + 24];
    return cell >= 0 ? cell / (geo.ext0 * geo.lines) : -1;
  };
  std::vector<Unit> units;
  // every load of a unit reads elements idx0 .. idx0 + 129 of a vector of >= n elements (+ 8 of slack, DevBuf)
  auto safe = [&](i64 idx0) { return idx0 >= 0 && idx0 + 130 <= n + 8; };
  std::vector<i64> chain, B;
  for (i64 h = 0; h < nr; ++h) {
    if (!ok[h] || pred[h] >= 0) continue;
    chain.clear(); B.clear();
    const int cnt = runs[h].cnt;
    const bool Y = cnt == 7;
    const int up_e = cnt - 3, dn_e = cnt - 2;   // entries of the +plane / -plane taps
    i64 base = runs[h].r0;
    for (i64 i = h; i >= 0; i = succ[i]) {
      chain.push_back(i);
      B.push_back(base);
      base += off(i, up_e);
    }
    const i64 L = (i64)chain.size();
    i64 cmin = 0, cmax = 0;
    for (i64 k = 0; k < L; ++k) {
      const i64 a = runs[chain[k]].r0 - B[k], b = a + runs[chain[k]].len;
      cmin = k == 0 ? a : std::min(cmin, a);
      cmax = k == 0 ? b : std::max(cmax, b);
    }
    for (i64 W = cmin - 1; W + 1 < cmax; W += geo.W) {
      auto range = [&](i64 k, i64& lo, i64& hi) {
        const i64 a = runs[chain[k]].r0 - B[k], b = a + runs[chain[k]].len;
        lo = std::max(a, W + 1);
        hi = std::min(b, W + 1 + geo.W);
        return lo < hi;
      };
      i64 k = 0;
      while (k < L) {
        i64 lo, hi;
        if (!range(k, lo, hi)) { ++k; continue; }
        i64 k1 = k;
        while (k1 < L && k1 - k < geo.kmax && range(k1, lo, hi)) {
          ++k1;
          if (aligned && Y && k1 < L && plane_of(chain[k1]) >= 0 && plane_of(chain[k1]) % geo.kmax == 0) break;   // common cut
        }
        const int K = (int)(k1 - k);
        bool fits = safe(B[k] + off(chain[k], dn_e) + W) && safe(B[k1 - 1] + off(chain[k1 - 1], up_e) + W);
        for (i64 q = k; q < k1 && fits; ++q) {
          fits = safe(B[q] + W);
          if (Y) fits = fits && safe(B[q] + W + off(chain[q], 2)) && safe(B[q] + W + off(chain[q], 3));
        }
        if (!fits) {
          for (i64 q = k; q < k1; ++q) {
            range(q, lo, hi);
            fallback.push_back(RowRange{B[q] + lo, B[q] + hi, cnt});
          }
          k = k1;
          continue;
        }
        Unit u;
        u.rec.assign(geo.REC, 0);
        range(k, lo, hi);
        u.key = (int)(B[k] + lo);
        u.order = 0;
        u.sub = 0;
        if (Y && geo.strip > 0 && geo.ext0 > 0 && geo.lines > 0 && RI > 24) {
          const i64 cell = info[RI * chain[k] + 24];
          if (cell >= 0) {
            u.order = march_order(cell, geo.ext0, geo.lines, geo.strip);
            if (aligned) {
              // plane GROUP instead of plane; then the window (grid position of the unit's first row along the line), then the line
              const i64 line = (cell / geo.ext0) % geo.lines, plane = cell / (geo.ext0 * geo.lines);
              u.order = (line / geo.strip) * ((i64)1 << 20) + plane / geo.kmax;
              const i64 x0 = cell % geo.ext0 + ((B[k] + lo) - (i64)runs[chain[k]].r0);
              u.sub = (x0 / geo.W) * ((i64)1 << 20) + line;
            }
          }
        }
        const int Kp = K <= geo.KS ? geo.KS : geo.K;   // the kernel's two unit sizes: shorter units end with empty planes
        u.rec[0] = Kp | (cnt << 8);   // | active lanes << 16, below
        u.rec[1] = u.key;
        for (int q = 0; q < 2 * cnt; ++q) u.rec[2 + q] = info[RI * chain[k] + 8 + q];
        u.rec[16] = (int)(B[k] + off(chain[k], dn_e) + W);
        u.rec[17] = (int)(B[k1 - 1] + off(chain[k1 - 1], up_e) + W);
        int hi_max = 0;
        for (int i = 0; i < K; ++i) {
          const i64 q = k + i;
          range(q, lo, hi);
          hi_max = std::max(hi_max, (int)(hi - W));
          u.rec[18 + 4 * i] = (int)(B[q] + W);
          u.rec[19 + 4 * i] = Y ? off(chain[q], 3) : 0;   // -lateral
          u.rec[20 + 4 * i] = Y ? off(chain[q], 2) : 0;   // +lateral
          u.rec[21 + 4 * i] = (int)(lo - W) | ((int)(hi - W) << 8);
          rows_m += hi - lo;
        }
        // rows < hi_max need elements <= hi_max of the lines: lanes 0 .. hi_max / 2
        u.rec[0] |= std::min(64, hi_max / 2 + 1) << 16;
        for (int i = K; i < Kp; ++i) {
          // closing planes compute nothing (lo = hi); their line is the one ABOVE the last real plane -- the line that
          // plane's +plane tap reads, and a valid address for every load of the closing plane
          u.rec[18 + 4 * i] = u.rec[17];
          u.rec[19 + 4 * i] = 0;
          u.rec[20 + 4 * i] = 0;
          u.rec[21 + 4 * i] = 1 | (1 << 8);
        }
        units.push_back(std::move(u));
        k = k1;
      }
    }
  }
  std::sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) {
    if (a.order != b.order) return a.order < b.order;
    return a.sub != b.sub ? a.sub < b.sub : a.key < b.key;
  });
  mrec.reserve(units.size() * geo.REC);
  for (auto& u : units) mrec.insert(mrec.end(), u.rec.begin(), u.rec.end());
  if (okeys) {   // the order as one number per unit (build_tiles lines the slices up with it)
    okeys->clear();
    okeys->reserve(units.size());
    for (auto& u : units) okeys->push_back(u.order * ((i64)1 << 32) + u.key);
  }
}

// Tile table of the slice kernel (pg_spmv.hip k_spmv_s): the units (sorted by ukey) and the slices that need no halo
// (sorted by skey, the same order) are both split into 8 equal parts -- one per XCD, as the kernel without tiles does --
// and every part into `T` tiles: equal numbers of units, and the slices whose keys lie between the tile's first unit and
// the next tile's.  tab[(q (T + 1) + t) 2 + {0, 1}] = first unit / first slice of tile t of XCD q (t = T: the ends).
inline void plan_tiles(const std::vector<int64_t>& ukey, const std::vector<int64_t>& skey, int T, std::vector<int>& tab) {
  typedef int64_t i64;
  const i64 nu = (i64)ukey.size(), ns = (i64)skey.size();
  tab.assign((size_t)8 * (T + 1) * 2, 0);
  for (int q = 0; q < 8; ++q) {
    const i64 u0 = nu * q / 8, u1 = nu * (q + 1) / 8, s0 = ns * q / 8, s1 = ns * (q + 1) / 8;
    i64 sprev = s0;
    for (int t = 0; t <= T; ++t) {
      const i64 u = u0 + (u1 - u0) * t / T;
      i64 sl;
      if (t == 0) sl = s0;
      else if (t == T || u >= nu) sl = s1;
      else sl = std::lower_bound(skey.begin(), skey.end(), ukey[u]) - skey.begin();
      sl = std::min(std::max(sl, sprev), s1);
      sprev = sl;
      tab[((size_t)q * (T + 1) + t) * 2] = (int)u;
      tab[((size_t)q * (T + 1) + t) * 2 + 1] = (int)sl;
    }
  }
}

}  // namespace pghost
