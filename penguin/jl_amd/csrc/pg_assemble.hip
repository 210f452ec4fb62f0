// pg_assemble.hip -- K7 (block system -> CSR), K9 (border rows as a row mask) and K10 (active set)
//
//   reference                                                   here
//   A_mono_unstead_diff / A_diph_unstead_diff (8 sparse products)  diffusion.jl:212-241,334-389   eval_row (pg_stencil.h)
//   BC_border_mono! / BC_border_diph!  (A[row,:] .= 0 per cell)    solver.jl:417-499,540-580       eval_row (row mask)
//   remove_zero_rows_cols!  (sum(abs.(A)) twice, every step)       solver.jl:59-78                 k_flags + scan, ONCE
//
// One thread per padded cell evaluates its rows from the capacities; a row is kept iff it has a non-zero
// entry and some row has a non-zero entry in its column (rows ∩ cols, solver.jl:71).  Flags -> exclusive
// scan -> local numbering; per-row counts -> scan -> rowptr; second evaluation writes col/val.
#include "pg_host_algos.h"
#include "pg_scan.h"
#include "pg_system.h"

using namespace pg;

namespace {

struct Region {
  i64 r0, r1;   // planes whose rows can be evaluated locally
  i64 a0, a1;   // planes whose active flags are exact: [max(p0-1,0), min(p1+1,nplanes))
};

Region regions(const Slab& s) {
  Region g;
  g.r0 = s.s0 == 0 ? 0 : s.s0 + 1;
  g.r1 = s.s1 == s.nplanes ? s.nplanes : s.s1 - 1;
  g.a0 = s.p0 - 1 < 0 ? 0 : s.p0 - 1;
  g.a1 = s.p1 + 1 > s.nplanes ? s.nplanes : s.p1 + 1;
  return g;
}

// rowflag[k*Mloc+lc] = row (k,lc) has a non-zero; colflag[...] = some evaluated row has a non-zero there
__global__ void k_flags(SysParams P, i64 Mloc, i64 lc_begin, i64 lc_end, unsigned char* rowflag,
                        unsigned char* colflag) {
  const int K = nkinds(P);
  const CapView& c = P.cap[0];
  for (i64 lc = lc_begin + blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < lc_end; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    for (int k = 0; k < K; ++k) {
      bool any = false;
      eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
        if (v != 0.0) {
          any = true;
          colflag[(i64)ck * Mloc + cl] = 1;
        }
      });
      rowflag[(i64)k * Mloc + lc] = any ? 1 : 0;
    }
  }
}

__global__ void k_active(int K, i64 Mloc, i64 lc_a0, i64 lc_a1, unsigned char* rowflag, const unsigned char* colflag) {
  const i64 total = (i64)K * Mloc;
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < total; q += (i64)gridDim.x * blockDim.x) {
    const i64 lc = q % Mloc;
    const bool in = lc >= lc_a0 && lc < lc_a1;
    rowflag[q] = (in && rowflag[q] && colflag[q]) ? 1 : 0;
  }
}

struct RedMap {
  i64 lcL, lcO, lcU, lcE;      // local cell boundaries: [lcL,lcO) lower ghost, [lcO,lcU) owned, [lcU,lcE) upper ghost
  i64 posO[MAX_KINDS], posU[MAX_KINDS];   // scan values at lcO and lcU
  i64 off_own[MAX_KINDS], offL[MAX_KINDS], offU[MAX_KINDS];
};

__global__ void k_red(int K, i64 Mloc, RedMap m, const unsigned char* act, const int* pos, int* red, int* row_cell) {
  const i64 total = (i64)K * Mloc;
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < total; q += (i64)gridDim.x * blockDim.x) {
    const int k = (int)(q / Mloc);
    const i64 lc = q % Mloc;
    int r = -1;
    if (act[q]) {
      const i64 p = pos[q];
      if (lc < m.lcO) r = (int)(m.offL[k] + p);
      else if (lc < m.lcU) {
        r = (int)(m.off_own[k] + (p - m.posO[k]));
        row_cell[r] = (int)lc;
      } else r = (int)(m.offU[k] + (p - m.posU[k]));
    }
    red[q] = r;
  }
}

__device__ inline int row_kind(const Numbering* /*unused*/, int K, const i64* off_own, i64 r) {
  int k = 0;
  while (k + 1 < K && r >= off_own[k + 1]) ++k;
  return k;
}

struct RowSegs {
  int K;
  i64 off_own[MAX_KINDS];
};

__global__ void k_count(SysParams P, RowSegs seg, i64 Mloc, i64 n_own, const int* row_cell, const int* red, int* cnt) {
  const CapView& c = P.cap[0];
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_own; r += (i64)gridDim.x * blockDim.x) {
    const int k = row_kind(nullptr, seg.K, seg.off_own, r);
    const i64 lc = row_cell[r];
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    int n = 0;
    eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
      if (v != 0.0 && red[(i64)ck * Mloc + cl] >= 0) ++n;
    });
    cnt[r] = n;
  }
}

__global__ void k_fill(SysParams P, RowSegs seg, i64 Mloc, i64 n_own, const int* row_cell, const int* red,
                       const int* rowptr, int* col, double* val) {
  const CapView& c = P.cap[0];
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_own; r += (i64)gridDim.x * blockDim.x) {
    const int k = row_kind(nullptr, seg.K, seg.off_own, r);
    const i64 lc = row_cell[r];
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    int at = rowptr[r];
    eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
      if (v != 0.0) {
        const int cc = red[(i64)ck * Mloc + cl];
        if (cc >= 0) {
          col[at] = cc;
          val[at] = v;
          ++at;
        }
      }
    });
  }
}

// ds[red] = |a_ii|^-1/2 for every numbered unknown (owned and ghost: ghost rows are evaluable locally)
__global__ void k_diag_scale(SysParams P, int K, i64 Mloc, const int* red, double* ds) {
  const CapView& c = P.cap[0];
  const i64 total = (i64)K * Mloc;
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < total; q += (i64)gridDim.x * blockDim.x) {
    const int r = red[q];
    if (r < 0) continue;
    const int k = (int)(q / Mloc);
    const i64 lc = q % Mloc;
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    double diag = 0.0;
    eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
      if (ck == k && cl == lc) diag = v;
    });
    const double a = fabs(diag);
    ds[r] = (a > 0.0 && a < 1e300) ? 1.0 / sqrt(a) : 1.0;
  }
}

__global__ void k_scale_vals(i64 n, const int* __restrict__ rowptr, const int* __restrict__ col,
                             const double* __restrict__ ds, double* __restrict__ val) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    const double sr = ds[r];
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) val[k] = sr * val[k] * ds[col[k]];
  }
}

// y[k*Mloc+lc] = sum_entries v * x[ck*Mloc+cl] for rows of planes [lc_begin,lc_end)
__global__ void k_apply_padded(SysParams P, i64 Mloc, i64 lc_begin, i64 lc_end, const double* x, double* y) {
  const int K = nkinds(P);
  const CapView& c = P.cap[0];
  for (i64 lc = lc_begin + blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < lc_end; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
        if (v != 0.0) s += v * x[(i64)ck * Mloc + cl];
      });
      y[(i64)k * Mloc + lc] = s;
    }
  }
}

}  // namespace

namespace pg {

void build_numbering(const SysParams& P, const Slab& s, Numbering& nb) {
  hipStream_t st = ctx().stream;
  const int K = nkinds(P);
  const i64 Mloc = s.Mloc();
  PG_REQUIRE((i64)K * Mloc < (i64)2147483647 * 2, "system too large for one GPU");
  nb.K = K;
  nb.Mloc = Mloc;
  const Region g = regions(s);
  DevBuf<unsigned char> rowflag((i64)K * Mloc), colflag((i64)K * Mloc);
  rowflag.zero();
  colflag.zero();
  const i64 lb = (g.r0 - s.s0) * s.plane, le = (g.r1 - s.s0) * s.plane;
  if (le > lb)
    hipLaunchKernelGGL(k_flags, dim3(grid_for(le - lb, 256, 256 * 16)), dim3(256), 0, st, P, Mloc, lb, le, rowflag.p, colflag.p);
  PG_HIP(hipGetLastError());
  const i64 la0 = (g.a0 - s.s0) * s.plane, la1 = (g.a1 - s.s0) * s.plane;
  hipLaunchKernelGGL(k_active, dim3(grid_for((i64)K * Mloc, 256, 256 * 16)), dim3(256), 0, st, K, Mloc, la0, la1, rowflag.p,
                     colflag.p);
  PG_HIP(hipGetLastError());
  colflag.release();

  // positions: exclusive scan per kind
  DevBuf<int> pos((i64)K * Mloc);
  DevBuf<int> totals(K);
  for (int k = 0; k < K; ++k)
    scan_exclusive<unsigned char>(rowflag.p + (i64)k * Mloc, pos.p + (i64)k * Mloc, Mloc, totals.p + k, st);
  std::vector<int> htot(K);
  totals.download(htot.data(), K);

  RedMap m;
  m.lcL = la0;
  m.lcO = (s.p0 - s.s0) * s.plane;
  m.lcU = (s.p1 - s.s0) * s.plane;
  m.lcE = la1;
  auto pos_at = [&](int k, i64 lc) -> i64 {
    if (lc >= Mloc) return htot[k];
    int v;
    pos.download(&v, 1, (i64)k * Mloc + lc);
    return v;
  };
  const i64 lc_first_end = (s.p0 + 1 - s.s0) * s.plane;   // end of the first owned plane
  const i64 lc_last_begin = (s.p1 - 1 - s.s0) * s.plane;  // start of the last owned plane
  static_assert(pghost::SLAB_MAX_KINDS == MAX_KINDS, "kinds");
  i64 pos_first_end[MAX_KINDS], pos_last_begin[MAX_KINDS], tot[MAX_KINDS];
  for (int k = 0; k < K; ++k) {
    m.posO[k] = pos_at(k, m.lcO);
    m.posU[k] = pos_at(k, m.lcU);
    pos_first_end[k] = pos_at(k, lc_first_end);
    pos_last_begin[k] = pos_at(k, lc_last_begin);
    tot[k] = htot[k];
  }
  // segment offsets, ghost segments and send chunks: plain host code shared with the CPU suite (pg_host_algos.h)
  pghost::SlabNumbering sn;
  pghost::slab_numbering(K, m.posO, m.posU, pos_first_end, pos_last_begin, tot, s.p0 > 0, s.p1 < s.nplanes, sn);
  nb.n_own = sn.n_own;
  nb.n_ghost = sn.n_ghost;
  PG_REQUIRE(sn.n_own + sn.n_ghost < (i64)2147483647, "reduced system exceeds int32 indexing");
  for (int k = 0; k < K; ++k) {
    nb.cnt_own[k] = sn.cnt_own[k]; nb.off_own[k] = sn.off_own[k];
    nb.cntL[k] = sn.cntL[k]; nb.offL[k] = sn.offL[k];
    nb.cntU[k] = sn.cntU[k]; nb.offU[k] = sn.offU[k];
    nb.sendL_off[k] = sn.sendL_off[k]; nb.sendL_cnt[k] = sn.sendL_cnt[k];
    nb.sendU_off[k] = sn.sendU_off[k]; nb.sendU_cnt[k] = sn.sendU_cnt[k];
    m.off_own[k] = nb.off_own[k];
    m.offL[k] = nb.offL[k];
    m.offU[k] = nb.offU[k];
  }
  nb.red.alloc((i64)K * Mloc);
  nb.row_cell.alloc(nb.n_own > 0 ? nb.n_own : 1);
  hipLaunchKernelGGL(k_red, dim3(grid_for((i64)K * Mloc, 256, 256 * 16)), dim3(256), 0, st, K, Mloc, m, rowflag.p, pos.p,
                     nb.red.p, nb.row_cell.p);
  PG_HIP(hipGetLastError());
  PG_HIP(hipStreamSynchronize(st));
}

void assemble_csr(const SysParams& P, const Slab& s, const Numbering& nb, CsrMatrix& A, bool scale) {
  hipStream_t st = ctx().stream;
  const i64 n = nb.n_own;
  A.n = n;
  A.rowptr.alloc(n + 1);
  RowSegs seg;
  seg.K = nb.K;
  for (int k = 0; k < MAX_KINDS; ++k) seg.off_own[k] = nb.off_own[k];
  if (n == 0) {
    A.rowptr.zero();
    A.nnz = 0;
    A.ds.alloc(nb.n_vec() > 0 ? nb.n_vec() : 1);
    return;
  }
  DevBuf<int> cnt(n);
  const int gr = grid_for(n, 256, 256 * 16);
  hipLaunchKernelGGL(k_count, dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p, cnt.p);
  PG_HIP(hipGetLastError());
  scan_exclusive<int>(cnt.p, A.rowptr.p, n, A.rowptr.p + n, st);
  int nnz = 0;
  A.rowptr.download(&nnz, 1, n);
  PG_REQUIRE(nnz >= 0, "nnz overflow");
  A.nnz = nnz;
  // +8: the SpMV streams 16-byte-aligned pairs/quads and may touch a few entries past nnz
  A.col.alloc(nnz + 8);
  A.val.alloc(nnz + 8);
  A.col.zero();
  A.val.zero();
  hipLaunchKernelGGL(k_fill, dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p, A.rowptr.p, A.col.p,
                     A.val.p);
  PG_HIP(hipGetLastError());
  if (scale) {
    // symmetric diagonal equilibration folded into the values (see CsrMatrix)
    A.ds.alloc(nb.n_vec());
    hipLaunchKernelGGL(k_diag_scale, dim3(grid_for((i64)nb.K * nb.Mloc, 256, 256 * 16)), dim3(256), 0, st, P, nb.K, nb.Mloc,
                       nb.red.p, A.ds.p);
    PG_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_scale_vals, dim3(gr), dim3(256), 0, st, n, A.rowptr.p, A.col.p, A.ds.p, A.val.p);
    PG_HIP(hipGetLastError());
  }
  PG_HIP(hipStreamSynchronize(st));
  build_spmv_chunks(A);
}

void apply_rows_padded(const SysParams& P, const Slab& s, const double* x, double* y) {
  hipStream_t st = ctx().stream;
  const i64 Mloc = s.Mloc();
  const Region g = regions(s);
  const i64 lb = (g.r0 - s.s0) * s.plane, le = (g.r1 - s.s0) * s.plane;
  if (le > lb)
    hipLaunchKernelGGL(k_apply_padded, dim3(grid_for(le - lb, 256, 256 * 16)), dim3(256), 0, st, P, Mloc, lb, le, x, y);
  PG_HIP(hipGetLastError());
}

}  // namespace pg
