// pg_spmv.hip -- the dominant kernel of the path: y = A x for the reduced cut-cell system
// (fp64 values, int32 column indices; rows in box order, 1 / 2N+1 / up to 2(2N+1) entries).
// Replaces the single-threaded CSC mul! inside IterativeSolvers (src/solver.jl:178-181).  Roofline: HBM.
//
// Three kernels, in the order they were written (PG_SPMV_VARIANT; profiles/r01_spmv_sweeps.txt, r01_spmv_slices.txt):
//
//   1    k_spmv      plain CSR, a block of 256 rows staged through LDS, two block barriers per chunk (2.0 TB/s).
//
//   2|38 k_spmv_cw   plain CSR (12 nnz + 20 n bytes), the best a CSR stream can do here: 61 % of 8 TB/s back-to-back.
//        The rows are cut once per matrix into CHUNKS of <= 64 consecutive rows and <= 508 entries (chunk_desc, the
//        row-block idea of CSR-Adaptive).  One WAVE owns one chunk per iteration: it streams the chunk's contiguous
//        val/col range with coalesced 16-byte loads into its PRIVATE slice of LDS -- no block barrier anywhere -- and
//        lane l walks row l out of LDS: column indices, all x gathers issued together, FMAs.  Everything inside an
//        iteration is branch-free with a FIXED number of vector-memory operations (clamped indices + selects):
//        hipcc puts every predicated load in its own basic block behind `s_waitcnt vmcnt(0)`, and only when every
//        path issues the same number of loads can it emit counted waits.  The chunk descriptor (a scalar load the
//        stream addresses depend on) is fetched two chunks ahead.  +4 = non-temporal hint on the matrix stream.
//
//   70   k_spmv_s    stencil slices (default): see "stencil slices" below.  0.25 GB instead of 0.99 GB at 512^3.
//
// The dots that follow an SpMV in BiCGStab / CG are fused into the epilogue (one partial per block, summed in a fixed
// order: deterministic); the slice kernel's last-arriving block can also evaluate the scalar phase (pg_spmv.h).  The slice
// kernel has one more epilogue, y = 2x - A x: the product with the Neumann preconditioner 2I - A of pg_krylov.hip.
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

#include "pg_host_algos.h"
#include "pg_krylov.h"
#include "pg_spmv.h"

namespace pg {
namespace {

// ---- variant 1 ---------------------------------------------------------------------------------------
constexpr int SPMV_ROWS = 256;          // rows per block iteration
constexpr int SPMV_LDS_ENTRIES = 3584;  // 256 rows x 14 entries (max row: 2*(2N+1) in 3-D)

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_spmv(i64 n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                const double* __restrict__ val, const double* __restrict__ x,
                                                double* __restrict__ y, const double* __restrict__ aux,
                                                double* __restrict__ partials, const double* __restrict__ sc) {
  __shared__ double s_val[SPMV_LDS_ENTRIES];
  __shared__ int s_col[SPMV_LDS_ENTRIES];
  __shared__ double s_red[BLOCK / 64];
  if (sc && sc[S_DONE] != 0.0) return;
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  const i64 nchunks = (n + SPMV_ROWS - 1) / SPMV_ROWS;
  for (i64 chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const i64 r0 = chunk * SPMV_ROWS;
    const i64 r1 = r0 + SPMV_ROWS < n ? r0 + SPMV_ROWS : n;
    const int base = rowptr[r0];
    const int cnt = rowptr[r1] - base;
    const i64 r = r0 + threadIdx.x;
    int a = 0, b = 0;
    if (r < r1) {
      a = rowptr[r] - base;
      b = rowptr[r + 1] - base;
    }
    double sum = 0.0;
    if (cnt <= SPMV_LDS_ENTRIES) {
      for (int k = threadIdx.x; k < cnt; k += BLOCK) {
        s_val[k] = val[base + k];
        s_col[k] = col[base + k];
      }
      __syncthreads();
      for (int k = a; k < b; ++k) sum += s_val[k] * x[s_col[k]];
      __syncthreads();
    } else {
      for (int k = a; k < b; ++k) sum += val[base + k] * x[col[base + k]];
    }
    if (r < r1) {
      y[r] = sum;
      if (MODE == 1) acc0 += aux[r] * sum;
      if (MODE >= 2) {
        acc0 += sum * x[r];
        acc1 += sum * sum;
      }
      if (MODE == 3) acc2 += aux[r] * sum;
    }
  }
  if (MODE >= 1) {
    const double t0 = block_sum(acc0, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  }
  if (MODE >= 2) {
    const double t1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = t1;
  }
  if (MODE == 3) {
    const double t2 = block_sum(acc2, s_red);
    if (threadIdx.x == 0) partials[4 * (size_t)gridDim.x + blockIdx.x] = t2;
  }
}

// ---- chunked wave kernel ---------------------------------------------------------------------------------------
constexpr int WUNR = 8;                          // gathers issued together per row (bulk rows: 2N+1 <= 7)
constexpr int GUNR = 4;                          // slice kernel, irregular rows: LDS (value, x) pairs read per round

template <bool NT, class T>
__device__ inline T stream_load(const T* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

struct Desc {          // wave-uniform (SGPRs): rows [r0,r1), entries [base,end)
  int r0, base, r1, end;
};

__device__ inline Desc load_desc(const int* __restrict__ cd, i64 ch) {
  // 4 consecutive ints at a wave-uniform address: one s_load_dwordx4 (lgkmcnt, independent of vmcnt)
  const int* p = cd + 2 * ch;
  Desc d;
  d.r0 = p[0]; d.base = p[1]; d.r1 = p[2]; d.end = p[3];
  return d;
}

// ---- chunked CSR kernel, 16-byte stream loads ---------------------------------------------------------------------
// values are read as aligned pairs of doubles starting at base & ~1, columns as aligned quads of ints starting at
// base & ~3 (1 KiB per wave instruction, the widest coalesced access; 6 instead of 16 stream instructions per
// chunk); the LDS image keeps the same shift.  Chunks hold <= 508 entries so the shifted image fits 512 slots;
// CsrMatrix pads val/col by 8 entries so the widened reads stay inside the allocation.
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));
constexpr int XV_IT = 4, XC_IT = 2;

struct StreamW {
  int ra, rb;
  d2_t v[XV_IT];
  i4_t c[XC_IT];
};

template <bool NT>
__device__ __forceinline__ void stream_issue_w(StreamW& q, const Desc& d, const int* __restrict__ rowptr,
                                      const int* __restrict__ col, const double* __restrict__ val, int lane) {
  const int r = d.r0 + lane;
  q.ra = stream_load<NT>(rowptr + (r < d.r1 ? r : d.r1));
  q.rb = stream_load<NT>(rowptr + (r + 1 < d.r1 ? r + 1 : d.r1));
  const int basev = d.base & ~1, basec = d.base & ~3;
  const int endm1 = d.end > d.base ? d.end - 1 : d.base;
  const int lastp = (endm1 - basev) >> 1, lastq = (endm1 - basec) >> 2;
  const d2_t* vp = reinterpret_cast<const d2_t*>(val + basev);
  const i4_t* cp = reinterpret_cast<const i4_t*>(col + basec);
#pragma unroll
  for (int j = 0; j < XV_IT; ++j) {
    const int k = lane + 64 * j;
    q.v[j] = stream_load<NT>(vp + (k < lastp ? k : lastp));
  }
#pragma unroll
  for (int j = 0; j < XC_IT; ++j) {
    const int k = lane + 64 * j;
    q.c[j] = stream_load<NT>(cp + (k < lastq ? k : lastq));
  }
}

template <int MODE, bool NT>
__global__ __launch_bounds__(BLOCK) void k_spmv_cw(i64 n, i64 nchunks, const int* __restrict__ chunk_desc,
                                                   const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const double* __restrict__ val, const double* __restrict__ x,
                                                   double* __restrict__ y, const double* __restrict__ aux,
                                                   double* __restrict__ partials, const double* __restrict__ sc) {
  __shared__ __attribute__((aligned(16))) double s_val[BLOCK / 64][512];
  __shared__ __attribute__((aligned(16))) int s_col[BLOCK / 64][512];
  __shared__ double s_red[BLOCK / 64];
  if (sc && sc[S_DONE] != 0.0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* __restrict__ sv = s_val[wave];
  int* __restrict__ scl = s_col[wave];
  d2_t* sv2 = reinterpret_cast<d2_t*>(sv);
  i4_t* sc4 = reinterpret_cast<i4_t*>(scl);
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  const i64 first = (i64)blockIdx.x * (BLOCK / 64) + wave, wstride = (i64)gridDim.x * (BLOCK / 64);
  const i64 lastc = nchunks - 1;
  StreamW q;
  Desc dcur, dnext;
  if (first < nchunks) {
    dcur = load_desc(chunk_desc, first);
    dnext = load_desc(chunk_desc, first + wstride < nchunks ? first + wstride : lastc);
  }
  for (i64 chunk = first; chunk < nchunks; chunk += wstride) {
    stream_issue_w<NT>(q, dcur, rowptr, col, val, lane);
    const int r = dcur.r0 + lane;
    const bool live = r < dcur.r1;
    const int basev = dcur.base & ~1, basec = dcur.base & ~3;
    const int av = q.ra - basev, ac = q.ra - basec, len = q.rb - q.ra;
#pragma unroll
    for (int j = 0; j < XV_IT; ++j) sv2[lane + 64 * j] = q.v[j];
#pragma unroll
    for (int j = 0; j < XC_IT; ++j) sc4[lane + 64 * j] = q.c[j];
    __builtin_amdgcn_wave_barrier();
    double xv[WUNR], vv[WUNR];
    const int c0 = dcur.base - basec;           // first real column slot of the chunk: always valid
#pragma unroll
    for (int j = 0; j < WUNR; ++j) {
      const bool ok = j < len;
      xv[j] = x[scl[ok ? ac + j : c0]];
      vv[j] = ok ? sv[av + j] : 0.0;
    }
    const i64 c2 = chunk + 2 * wstride;
    const Desc dnext2 = load_desc(chunk_desc, c2 < nchunks ? c2 : lastc);
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < WUNR; ++j) sum += vv[j] * xv[j];
    if (__builtin_expect(len > WUNR, 0))
      for (int k = WUNR; k < len; ++k) sum += sv[av + k] * x[scl[ac + k]];
    __builtin_amdgcn_wave_barrier();
    if (live) {
      y[r] = sum;
      if (MODE == 1) acc0 += aux[r] * sum;
      if (MODE >= 2) {
        acc0 += sum * x[r];
        acc1 += sum * sum;
      }
      if (MODE == 3) acc2 += aux[r] * sum;
    }
    dcur = dnext;
    dnext = dnext2;
  }
  if (MODE >= 1) {
    const double t0 = block_sum(acc0, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  }
  if (MODE >= 2) {
    const double t1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = t1;
  }
  if (MODE == 3) {
    const double t2 = block_sum(acc2, s_red);
    if (threadIdx.x == 0) partials[4 * (size_t)gridDim.x + blockIdx.x] = t2;
  }
}

// ---- stencil slices (default) ----------------------------------------------------------------------------------
// The assembler knows what a generic CSR kernel cannot: away from the interface every row of a line carries the
// SAME stencil -- identical (col - row) offsets, and on a uniform mesh bitwise identical (equilibrated) values.
// build_spmv_chunks() detects that once per matrix (k_row_same) and re-slices the rows:
//   U slice   <= 254 consecutive rows, identical offsets and values: the slice is ONE 128-byte record (descriptor,
//             offsets, values); lane l computes rows r0+2l, r0+2l+1 from coalesced 16-byte x loads.  Nothing else
//             of the matrix is streamed.
//   P slice   identical offsets, per-row values: values re-laid slot-major (pval[j*rows + l]) so the 8 B/entry
//             stream is coalesced without LDS; the 4 B/entry column stream disappears.
//   G chunk   the remaining irregular rows (cut cells, their neighbours, line ends, rows with > 8 entries), packed 64
//             at a time into a compact CSR of their own and processed like k_spmv_cw (LDS-staged stream + gathers).
// The products are accumulated in the CSR entry order in all three paths, so y is bitwise what the CSR kernels give.
//
// What bounds the U path (SQ counters + ablations, profiles/r01_spmv_slices.txt): the kernel ran at the same speed with
// every x load forced to hit one L1 line, with the record loads forced to one line, at 4, 5, 6 or 8 waves per SIMD,
// and with 8- or 16-byte accesses: ~22 cycles per vector-memory INSTRUCTION per CU whatever its width -- the
// address unit, not HBM, L2 or latency.  Hence: two rows per lane (16-byte accesses), no padded slots (the slot
// count is a template parameter: 1 border identity rows, 3 / 5 / 7 the 1-D / 2-D / 3-D stencils), and one record
// load per slice instead of descriptor + offsets + values.
//
// Slices are sorted by first row; with `xcd` each XCD (blocks are dispatched round-robin over the 8 XCDs) sweeps one
// contiguous eighth of them, so the +-line / +-plane x neighbours are found in that XCD's own L2.
enum { SL_U = 0, SL_P = 1, SL_G = 2 };
constexpr int SL_MAXROWS_U = 254, SL_MAXROWS_P = 128, SL_MAXCNT = 8, SL_REC = 32;   // record: 32 dwords

struct SDesc {
  int r0, meta, base, aux;
};

// wave-uniform values out of a vector register (the result lives in SGPRs)
__device__ inline int rlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte access, 8-byte aligned

template <bool NT>
__device__ inline d2_t load_pair(const double* p) {
  const d2u_t* q = reinterpret_cast<const d2u_t*>(p);
  d2u_t v = NT ? __builtin_nontemporal_load(q) : *q;
  d2_t r;
  r.x = v.x; r.y = v.y;
  return r;
}

// epilogue of mode 4: y = pc0 x + pc1 (A x) -- one factor I - τÂ (pc0 = 1, pc1 = -τ) of the polynomial preconditioner's
// residual polynomial, or 2x - Âx (pg_krylov.hip); the ONE expression every path of the kernel uses
__device__ inline double mode4_out(double xown, double ax, double pc0, double pc1) { return pc0 * xown + pc1 * ax; }

// Launch modes of the slice kernel (MODE): what the epilogue does with s = (A x)_row.  A, B = operand vectors read at the
// row (pa, pb below); own = x_row.
//   0  y = s
//   1  y = s;                         acc0 += A y                                   A = aux
//   2  y = s;                         acc0 += A y, acc1 += y y                      A = dotx (or x)
//   3  y = s;                         acc0 += A y, acc1 += y y, acc2 += B y         A = dotx (or x), B = aux
//   4  y = pc0 own + pc1 s                                                          (one factor of the polynomial)
//   5  y = B - (pc0 own + pc1 s);     acc0 += A y                                   A = aux, B = base
//   6  y = A - (pc0 own + pc1 s);     acc0 += A y, acc1 += y y, acc2 += B y         A = dotx, B = aux
//   7  y = pc0 own + pc1 s;           A_row += pc2 own  (A = accv, read and written)
//   8  y = B + (pc0 own + pc1 s)                                                     B = base  (Horner step of x = x0 + q(Â)y)
// 5 / 6 close a chain of mode-4 launches: v = p - R(Â)p with (r̂,v), t = s - R(Â)s with (t,s), (t,t), (r̂,t); 7 is a step of
// the same chain that also accumulates q(Â)y = Σ τ_k w_(k-1) (pg_krylov.hip).
template <int MODE>
struct ModeInfo {
  static constexpr bool OWN = MODE >= 4;
  static constexpr bool HAS_A = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7;
  static constexpr bool HAS_B = MODE == 3 || MODE == 5 || MODE == 6 || MODE == 8;
  static constexpr bool DOT0 = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6;
  static constexpr bool DOT1 = MODE == 2 || MODE == 3 || MODE == 6;
  static constexpr bool DOT2 = MODE == 3 || MODE == 6;
};

// y of one row from its product s; a, b = the operands A, B at the row
template <int MODE>
__device__ __forceinline__ double mode_out(double s, double own, double a, double b, double pc0, double pc1, double pc2) {
  if (MODE == 4 || MODE == 7) return mode4_out(own, s, pc0, pc1);
  if (MODE == 5) return b - mode4_out(own, s, pc0, pc1);
  if (MODE == 6) return a - mode4_out(own, s, pc0, pc1);
  if (MODE == 8) return pc2 * b + mode4_out(own, s, pc0, pc1);   // Horner step: pc2 base + pc0 x + pc1 A x
  return s;
}

// rows of one U (PSL = false) or P (PSL = true) slice with CNT stencil slots; rec = the slice's record (lane & 31).
// NB batches of 128 rows: every load of the slice is issued before the first FMA (one exposed latency per slice).
template <int CNT, int NB, bool PSL, int MODE, bool NT>
__device__ __forceinline__ void slice_rows(const SDesc& d, int nrows, int rec, const double* __restrict__ pval,
                                  const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ pa,
                                  const double* __restrict__ pb, int lane, double& acc0, double& acc1, double& acc2, int dbg,
                                  double pc0, double pc1, double pc2) {
  using MI = ModeInfo<MODE>;
  int o[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) o[j] = (dbg & 4) ? 0 : rlane(rec, 4 + j);
  int l0[NB];
  bool live0[NB], live1[NB];
  d2_t xv[NB][CNT];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int l = 128 * b + 2 * lane;
    live0[b] = l < nrows;
    live1[b] = l + 1 < nrows;
    l0[b] = live0[b] ? l : 0;
#pragma unroll
    for (int j = 0; j < CNT; ++j) xv[b][j] = load_pair<false>(x + d.r0 + l0[b] + o[j]);
  }
  d2_t sum[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) { sum[b].x = 0.0; sum[b].y = 0.0; }
  if (!PSL) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      const double c = __hiloint2double(rlane(rec, 13 + 2 * j), rlane(rec, 12 + 2 * j));
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        sum[b].x += c * xv[b][j].x;
        sum[b].y += c * xv[b][j].y;
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double* __restrict__ pv = pval + d.base + l0[b];
      d2_t vv[CNT];
#pragma unroll
      for (int j = 0; j < CNT; ++j) vv[j] = load_pair<NT>(pv + j * nrows);
#pragma unroll
      for (int j = 0; j < CNT; ++j) {
        sum[b].x += vv[j].x * xv[b][j].x;
        sum[b].y += vv[j].y * xv[b][j].y;
      }
    }
  }
  d2_t own[NB], ax[NB], ax2[NB];   // the row's own x, operands A and B (ModeInfo)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    own[b].x = own[b].y = ax[b].x = ax[b].y = ax2[b].x = ax2[b].y = 0.0;
    if (MI::OWN) own[b] = load_pair<false>(x + d.r0 + l0[b]);
    if (MI::HAS_A) ax[b] = load_pair<false>(pa + d.r0 + l0[b]);
    if (MI::HAS_B) ax2[b] = load_pair<false>(pb + d.r0 + l0[b]);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    sum[b].x = mode_out<MODE>(sum[b].x, own[b].x, ax[b].x, ax2[b].x, pc0, pc1, pc2);
    sum[b].y = mode_out<MODE>(sum[b].y, own[b].y, ax[b].y, ax2[b].y, pc0, pc1, pc2);
    const bool st = !(dbg & 8) || sum[b].x == 1.2345e-300;
    double* yp = y + d.r0 + l0[b];
    double* ap = const_cast<double*>(pa) + d.r0 + l0[b];   // mode 7: the accumulated vector
    if (live1[b]) {
      if (st) {
        d2u_t out;
        out.x = sum[b].x; out.y = sum[b].y;
        if (!(dbg & 2048)) __builtin_nontemporal_store(out, reinterpret_cast<d2u_t*>(yp));
        else *reinterpret_cast<d2u_t*>(yp) = out;
      }
      if (MODE == 7) {
        d2u_t out;
        out.x = ax[b].x + pc2 * own[b].x; out.y = ax[b].y + pc2 * own[b].y;
        *reinterpret_cast<d2u_t*>(ap) = out;
      }
      if (MI::DOT0) acc0 += ax[b].x * sum[b].x + ax[b].y * sum[b].y;
      if (MI::DOT1) acc1 += sum[b].x * sum[b].x + sum[b].y * sum[b].y;
      if (MI::DOT2) acc2 += ax2[b].x * sum[b].x + ax2[b].y * sum[b].y;
    } else if (live0[b]) {
      if (st) *yp = sum[b].x;
      if (MODE == 7) *ap = ax[b].x + pc2 * own[b].x;
      if (MI::DOT0) acc0 += ax[b].x * sum[b].x;
      if (MI::DOT1) acc1 += sum[b].x * sum[b].x;
      if (MI::DOT2) acc2 += ax2[b].x * sum[b].x;
    }
  }
}

template <int CNT, bool PSL, int MODE, bool NT>
__device__ __forceinline__ void slice_nb(const SDesc& d, int nrows, int rec, const double* __restrict__ pval,
                                const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ pa,
                                const double* __restrict__ pb, int lane, double& acc0, double& acc1, double& acc2, int dbg,
                                double pc0, double pc1, double pc2) {
  if (!PSL && nrows > 128) slice_rows<CNT, 2, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2);
  else slice_rows<CNT, 1, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2);
}

template <bool PSL, int MODE, bool NT>
__device__ __forceinline__ void slice_dispatch(int cnt, const SDesc& d, int nrows, int rec, const double* __restrict__ pval,
                                      const double* __restrict__ x, double* __restrict__ y,
                                      const double* __restrict__ pa, const double* __restrict__ pb, int lane, double& acc0,
                                      double& acc1, double& acc2, int dbg, double pc0, double pc1, double pc2) {
  switch (cnt) {   // wave-uniform
    case 1: slice_nb<1, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
    case 3: slice_nb<3, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
    case 5: slice_nb<5, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
    case 7: slice_nb<7, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
    case 2: slice_nb<2, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
    case 4: slice_nb<4, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
    case 6: slice_nb<6, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
    default: slice_nb<8, PSL, MODE, NT>(d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, pc0, pc1, pc2); break;
  }
}


// ---- marching units (M): the uniform interior of a 2-D / 3-D stencil ----------------------------------------------
// What bounds the U path is bytes through the L1 / texture-address path: 7 loads of x per row (80 B per row with the dot
// operands and the store) while only 8 B per row have to come from memory.  A U run of a 3-D problem is a piece of a grid
// line, and the +-1 taps of a row are its neighbours in the SAME load, the +-plane taps the rows of the next / previous
// line of the chain of runs that face each other across the slowest stencil direction (col - row offsets o[0] / o[cnt-1]
// lead from a row to the row of the same lateral position one plane down / up).  So one wave MARCHES along such a chain:
// it keeps the lines of planes k-1, k, k+1 in registers, lane l holding elements 2l, 2l+1 of a window of 128 consecutive
// elements; per plane it loads ONE new line of the chain plus the two lateral neighbour lines (3-D only), computes the
// 126 rows 1..126 of the window -- row 2l+1 from (2l, 2l+1, 2l+2), row 2l+2 from (2l+1, 2l+2, 2l+3), the elements of lane
// l+1 arriving by a lane shift -- and stores them: 3 loads of x per 126 rows instead of 7 per 128 (1 instead of 5 in
// 2-D), 24 instead of 56 bytes of x through L1 per row.  The products are accumulated in the rows' entry order as
// everywhere else: y is bitwise the CSR kernels' y.
//
// One 256-byte record per UNIT (<= MARCH_K consecutive planes of one chain x one window), read with one vector load:
//   dword 0            K | cnt << 8 | active lanes << 16   (planes in the unit, MARCH_K or MARCH_KS: shorter units are closed by empty
//                                               planes; 7 = 3-D stencil, 5 = 2-D stencil)
//   dword 1            first computed row (sort key only)
//   dwords 2..15       the cnt values of the rows (all rows of a unit share them bitwise), in the entry order of the
//                      assembled rows: +1, -1, [+lateral, -lateral,] +plane, -plane, diagonal (eval_row, pg_stencil.h)
//   dword 16, 17       row index held by lane 0 (.x) in the line below the first / above the last plane of the unit
//   dwords 18 + 4i..   plane i: row index held by lane 0 (.x); offset to the lower / upper lateral neighbour line
//                      (col - row; 3-D only); lo | hi << 8 = window-relative range [lo, hi) of the rows computed here
// build_march_units() forms chains and units on the host from the same row flags the slices are cut from.
#ifndef PG_MARCH_K
#define PG_MARCH_K 4
#endif
#ifndef PG_MARCH_SPLIT
#define PG_MARCH_SPLIT 1
#endif
constexpr int MARCH_K = PG_MARCH_K, MARCH_KS = 2, MARCH_REC = 64, MARCH_W = 126;   // planes per unit: full / short units
static_assert(MARCH_K > MARCH_KS && 18 + 4 * MARCH_K <= MARCH_REC, "unit record layout");

__device__ inline double lane_up(double v) {   // the value lane l + 1 holds (lane 63: unspecified)
  return __shfl_down(v, 1, 64);
}

struct MarchSide {   // per-plane operands besides the chain's own lines: lateral neighbour lines, dot operands
  d2_t ym, yp, ax, ax2;
};

// rows 2l+1, 2l+2 of one plane from the lines in registers; returns lane l+1's xc.x (the next plane's xm_n)
template <int CNT, int MODE>
__device__ __forceinline__ double march_plane(const double (&c)[CNT], int rng, int l2, double* __restrict__ yrow, const d2_t& xm,
                                              double xm_n, const d2_t& xc, const d2_t& xp, const MarchSide& sd, double pc0,
                                              double pc1, double pc2, double* __restrict__ arow, double& acc0, double& acc1,
                                              double& acc2, int dbg) {
  using MI = ModeInfo<MODE>;
  constexpr bool Y = CNT == 7;
  const double xc_nx = lane_up(xc.x), xc_ny = lane_up(xc.y), xp_n = lane_up(xp.x);
  // entry order of the assembled rows (eval_row, pg_stencil.h): +1, -1, [+lateral, -lateral,] +plane, -plane, diagonal
  double s1 = 0.0, s2 = 0.0;
  int j = 0;
  s1 += c[j] * xc_nx; s2 += c[j] * xc_ny; ++j;
  s1 += c[j] * xc.x;  s2 += c[j] * xc.y;  ++j;
  if (Y) {
    s1 += c[j] * sd.yp.x; s2 += c[j] * sd.yp.y; ++j;
    s1 += c[j] * sd.ym.x; s2 += c[j] * sd.ym.y; ++j;
  }
  s1 += c[j] * xp.y;  s2 += c[j] * xp_n;  ++j;
  s1 += c[j] * xm.y;  s2 += c[j] * xm_n;  ++j;
  s1 += c[j] * xc.y;  s2 += c[j] * xc_nx;
  // (the row's own x: elements 2l+1, 2l+2 of the centre line)
  s1 = mode_out<MODE>(s1, xc.y, sd.ax.x, sd.ax2.x, pc0, pc1, pc2);
  s2 = mode_out<MODE>(s2, xc_nx, sd.ax.y, sd.ax2.y, pc0, pc1, pc2);
  const int lo = rng & 255, hi = rng >> 8;
  const bool nost = (dbg & 32) && s1 != 1.2345e-300;   // diagnostics: no stores
  const bool allst = (dbg & 512) != 0;                  // diagnostics: every lane stores its pair (whole windows)
  const bool v1 = ((l2 + 1 >= lo && l2 + 1 < hi) || allst) && !nost, v2 = ((l2 + 2 >= lo && l2 + 2 < hi) || allst) && !nost;
  if (v1 && v2) {
    d2u_t out;
    out.x = s1; out.y = s2;
    // y leaves with the stream hint: fewer dirty lines for the end-of-kernel write-back, which sits between every two
    // dependent launches of the loop (55.3 -> 52.6 µs per back-to-back launch at 512^3 on one box, within the noise on another; system-scope write-through
    // stores, `sc0 sc1`, double the launch; dbg bit 2048: plain stores)
    if (!(dbg & 2048)) __builtin_nontemporal_store(out, reinterpret_cast<d2u_t*>(yrow));
    else *reinterpret_cast<d2u_t*>(yrow) = out;
  } else if (v1) {
    yrow[0] = s1;
  } else if (v2) {
    yrow[1] = s2;
  }
  if (MODE == 7) {   // the accumulated vector of the chain: A_row += pc2 own
    const double a1 = sd.ax.x + pc2 * xc.y, a2 = sd.ax.y + pc2 * xc_nx;
    if (v1 && v2) {
      d2u_t out;
      out.x = a1; out.y = a2;
      *reinterpret_cast<d2u_t*>(arow) = out;
    } else if (v1) {
      arow[0] = a1;
    } else if (v2) {
      arow[1] = a2;
    }
  }
  if (MI::DOT0) {
    const double w1 = v1 ? s1 : 0.0, w2 = v2 ? s2 : 0.0;
    acc0 += (v1 ? sd.ax.x : 0.0) * w1 + (v2 ? sd.ax.y : 0.0) * w2;
    if (MI::DOT1) acc1 += w1 * w1 + w2 * w2;
    if (MI::DOT2) acc2 += (v1 ? sd.ax2.x : 0.0) * w1 + (v2 ? sd.ax2.y : 0.0) * w2;
  }
  return xc_nx;
}

// One unit of exactly KK planes (the builder closes shorter ones with empty planes), fully unrolled: EVERY load of the
// unit -- KK + 2 lines of the chain, 2 KK lateral lines, the dot operands -- is independent of the arithmetic, so the
// compiler issues them up front (as far as registers allow) and the wave pays ONE memory round trip per unit.  A loop
// over the planes with prefetch registers did not get there: the registers carried around the back edge are copied,
// and a copy waits for the load that fills it (2.2 us per plane, measured).
// (Q0, KTOT: planes [Q0, Q0 + KK) of a unit of KTOT planes -- the launches with fused dots take a full unit in two halves,
// see march_dispatch)
template <int CNT, int MODE, int KK, int Q0 = 0, int KTOT = KK>
__device__ __forceinline__ void march_unit(int rec, int lane, const double* __restrict__ x, double* __restrict__ y,
                                  const double* __restrict__ pa, const double* __restrict__ pb, double pc0, double pc1,
                                  double pc2, double& acc0, double& acc1, double& acc2, int dbg) {
  using MI = ModeInfo<MODE>;
  constexpr bool Y = CNT == 7;
  const int amask = (dbg & 128) ? ~1 : ~0;          // diagnostics: every access 16-byte aligned (wrong results)
  const int one = (dbg & 128) ? 0 : 1;
  const bool nolat = (dbg & 64) != 0;               // diagnostics: no lateral lines
  double c[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) c[j] = __hiloint2double(rlane(rec, 3 + 2 * j), rlane(rec, 2 + 2 * j));
  const int l2 = 2 * lane;
  int rb[KK];
  d2_t ln[KK + 2];          // lines -1 .. KK of the chain
  MarchSide sd[KK];
  // lanes beyond the widest plane of the unit load nothing (a window at the end of a chord is mostly empty: what a load
  // costs the CU's memory pipe goes with its active lanes); they run the arithmetic on zeros and store nothing
  const bool act = lane < (rlane(rec, 0) >> 16);
  d2_t zero;
  zero.x = zero.y = 0.0;
#pragma unroll
  for (int q = 0; q < KK + 2; ++q) ln[q] = zero;
#pragma unroll
  for (int q = 0; q < KK; ++q) {
    rb[q] = rlane(rec, 18 + 4 * (Q0 + q)) & amask;
    sd[q].ym = sd[q].yp = sd[q].ax = sd[q].ax2 = zero;
  }
  if (act) {
    ln[0] = load_pair<false>(x + (rlane(rec, Q0 == 0 ? 16 : 18 + 4 * (Q0 - 1)) & amask) + l2);
#pragma unroll
    for (int q = 0; q < KK; ++q) ln[q + 1] = load_pair<false>(x + rb[q] + l2);
    ln[KK + 1] = load_pair<false>(x + (rlane(rec, Q0 + KK == KTOT ? 17 : 18 + 4 * (Q0 + KK)) & amask) + l2);
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      if (Y && !nolat) {
        sd[q].ym = load_pair<false>(x + rb[q] + (rlane(rec, 19 + 4 * (Q0 + q)) & amask) + l2 + one);
        sd[q].yp = load_pair<false>(x + rb[q] + (rlane(rec, 20 + 4 * (Q0 + q)) & amask) + l2 + one);
      }
      if (MI::HAS_A) sd[q].ax = load_pair<false>(pa + rb[q] + l2 + one);
      if (MI::HAS_B) sd[q].ax2 = load_pair<false>(pb + rb[q] + l2 + one);   // (no stream hint: the chain's input is re-read by every launch of the chain and stays in the Infinity Cache -- with the hint a Horner step took 65.7 instead of 53.8 us)
    }
  }
  double xm_n = lane_up(ln[0].x);
#pragma unroll
  for (int q = 0; q < KK; ++q)
    xm_n = march_plane<CNT, MODE>(c, rlane(rec, 21 + 4 * (Q0 + q)), l2, y + rb[q] + l2 + one, ln[q], xm_n, ln[q + 1], ln[q + 2], sd[q], pc0,
                                  pc1, pc2, const_cast<double*>(pa) + rb[q] + l2 + one, acc0, acc1, acc2, dbg);
}

template <int CNT, int MODE>
__device__ __forceinline__ void march_dispatch(int rec, int lane, const double* __restrict__ x, double* __restrict__ y,
                                      const double* __restrict__ pa, const double* __restrict__ pb, double pc0, double pc1,
                                      double pc2, double& acc0, double& acc1, double& acc2, int dbg) {
  // The launches with fused dots and one or two more operand vectors per plane (modes 1, 3: the closing launch of an
  // application of the operator; 5, 6: the same in the y-space form of the preconditioned loop) do not fit a unit of MARCH_K
  // planes into the 128 VGPRs of four waves per SIMD (10 / 14 VGPRs went to scratch in modes 5 / 6): they take it in two
  // halves -- two more line loads per unit, which hit L1
  constexpr bool SPLIT = (MODE == 1 || MODE == 3 || MODE == 5 || MODE == 6) && MARCH_K == 2 * MARCH_KS && PG_MARCH_SPLIT;
  if ((rlane(rec, 0) & 255) == MARCH_K) {
    if (SPLIT) {
      march_unit<CNT, MODE, MARCH_KS, 0, MARCH_K>(rec, lane, x, y, pa, pb, pc0, pc1, pc2, acc0, acc1, acc2, dbg);
      march_unit<CNT, MODE, MARCH_KS, MARCH_KS, MARCH_K>(rec, lane, x, y, pa, pb, pc0, pc1, pc2, acc0, acc1, acc2, dbg);
    } else {
      march_unit<CNT, MODE, MARCH_K>(rec, lane, x, y, pa, pb, pc0, pc1, pc2, acc0, acc1, acc2, dbg);
    }
  } else {
    march_unit<CNT, MODE, MARCH_KS>(rec, lane, x, y, pa, pb, pc0, pc1, pc2, acc0, acc1, acc2, dbg);
  }
}

// DIAG: the diagnostic build of the kernel (ablations for profiles/: PG_SPMV_XCD bits 8.. skip parts of the work or force
// access patterns and give WRONG products).  The shipped instantiation has DIAG = false: `dbg` is the constant 0 there and
// none of the switches exists in its code.
template <int MODE, bool NT, bool DIAG>
__global__ __launch_bounds__(BLOCK, 4) void k_spmv_s(i64 nslices, const int* __restrict__ srec,
                                                  const double* __restrict__ pval, const int* __restrict__ g_rowid,
                                                  const int* __restrict__ g_rowptr, const int* __restrict__ g_col,
                                                  const double* __restrict__ g_val, const double* __restrict__ x,
                                                  double* __restrict__ y, const double* __restrict__ aux,
                                                  double* __restrict__ partials, const double* __restrict__ sc, int xcd,
                                                  FinArgs fin, int pstride, int accum, const int* __restrict__ mrec,
                                                  i64 nunits, const int* __restrict__ tiles, int tpx) {
  __shared__ __attribute__((aligned(16))) double s_val[BLOCK / 64][512];
  __shared__ __attribute__((aligned(16))) double s_x[BLOCK / 64][512];   // G chunks: x[col] of every entry of the chunk
  __shared__ double s_red[BLOCK / 64];
  if (sc && sc[S_DONE] != 0.0) return;
  using MI = ModeInfo<MODE>;
  constexpr bool DOTS = MI::DOT0;
  const double* __restrict__ dx = fin.dotx ? fin.dotx : x;   // operand of the (y, .) dot of modes 2 / 3 / 6
  // operand vectors A and B of the epilogue (ModeInfo)
  const double* __restrict__ pa = (MODE == 1 || MODE == 5) ? aux : ((MODE == 2 || MODE == 3 || MODE == 6) ? dx : (MODE == 7 ? fin.accv : nullptr));
  const double* __restrict__ pb = (MODE == 3 || MODE == 6) ? aux : ((MODE == 5 || MODE == 8) ? fin.base : nullptr);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* __restrict__ sv = s_val[wave];
  double* __restrict__ sx = s_x[wave];
  d2_t* sv2 = reinterpret_cast<d2_t*>(sv);
  d2_t* sx2 = reinterpret_cast<d2_t*>(sx);
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  // diagnostics (PG_SPMV_XCD bits 8..; results are wrong with any of them): 1 skip G chunks, 2 skip U/P slices,
  // 4 every x load hits one line, 8 no y stores in U/P slices, 16 skip the marching units, 32 no y stores in the units,
  // 64 no lateral lines in the units, 128 every access of the units 16-byte aligned, 256 non-temporal y stores in the
  // units, 512 the units store whole windows, 2048 plain y stores in the units, 4096 slices before units (the latency-bound
  // irregular chunks then run alone at the start instead of under the streaming units of slower waves: 52 vs 45.6 us)
  const int dbg = DIAG ? (xcd >> 8) : 0;
  xcd &= 255;
  // ---- marching units first (the bulk of the rows), then the slices.  (Alternating units and slices in every wave, so
  // that the latency-bound irregular chunks of some waves overlap the streaming of others, measured no faster, and one
  // merged loop costs registers: the live ranges of both bodies add up -- 144 VGPRs, a wave per SIMD less.)
  xcd &= 1;
  // Work of this wave: every (BLOCK / 64 x blocks)-th unit and slice of its XCD's eighth of both lists (XCD-aware: the
  // blocks of an XCD share one L2) -- in TILES when the image has a tile table (build_tiles): tile t of XCD q = units
  // [tb[2t], tb[2t+2]) and the slices [tb[2t+1], tb[2t+3]) that lie next to them in the grid, units first.  The irregular
  // rows of a tile then gather x from lines the tile's units have just pulled into this XCD's L2; with all units before
  // all slices those lines crossed the fabric twice (PMC, 512^3: 28 MB of 234 per launch).  The round robin over the
  // waves runs on across the tiles (a tile holds about one unit per wave).
  const bool xsplit = xcd && gridDim.x >= 8;
  const int xid = xsplit ? (blockIdx.x & 7) : 0;
  const int woff = xsplit ? (int)(blockIdx.x >> 3) * (BLOCK / 64) + wave : (int)blockIdx.x * (BLOCK / 64) + wave;
  const int wstr = xsplit ? (int)(((i64)gridDim.x + 7 - xid) >> 3) * (BLOCK / 64) : (int)gridDim.x * (BLOCK / 64);
  const int* __restrict__ tb = tiles ? tiles + 2 * (tpx + 1) * xid : nullptr;
  const int ntile = tiles ? tpx : 1;
  const int l31 = lane & 31;
  for (int tile = 0; tile < ntile; ++tile) {
  i64 ulo, uhi, slo, shi, ubase, sbase;
  if (tiles) {
    ubase = tb[0]; sbase = tb[1];
    ulo = tb[2 * tile]; slo = tb[2 * tile + 1]; uhi = tb[2 * tile + 2]; shi = tb[2 * tile + 3];
  } else if (xsplit) {
    ubase = ulo = nunits * xid / 8; uhi = nunits * (xid + 1) / 8;
    sbase = slo = nslices * xid / 8; shi = nslices * (xid + 1) / 8;
  } else {
    ubase = ulo = 0; uhi = nunits;
    sbase = slo = 0; shi = nslices;
  }
#pragma nounroll
  for (int phase = 0; phase < 2; ++phase) {
  if ((phase == 0) != ((dbg & 4096) != 0)) {   // (diagnostics, bit 4096: slices before units)
  if (nunits > 0) {
    int d = (woff - (int)(ulo - ubase)) % wstr;
    d = d < 0 ? d + wstr : d;
    i64 ucur = ulo + d;
    const i64 ustride = wstr;
    int urec = ucur < uhi ? mrec[MARCH_REC * ucur + lane] : 0;
    for (; ucur < uhi; ucur += ustride) {
      const i64 un = ucur + ustride;
      const int urec_n = mrec[MARCH_REC * (un < uhi ? un : ucur) + lane];   // the next unit's record is in flight meanwhile
      if (!(dbg & 16)) {
        if (((rlane(urec, 0) >> 8) & 255) == 7)
          march_dispatch<7, MODE>(urec, lane, x, y, pa, pb, fin.pc0, fin.pc1, fin.pc2, acc0, acc1, acc2, dbg);
        else
          march_dispatch<5, MODE>(urec, lane, x, y, pa, pb, fin.pc0, fin.pc1, fin.pc2, acc0, acc1, acc2, dbg);
      }
      urec = urec_n;
    }
  }
  continue;
  }
  int ds_ = (woff - (int)(slo - sbase)) % wstr;
  ds_ = ds_ < 0 ? ds_ + wstr : ds_;
  const i64 first = slo + ds_, wstride = wstr, hi = shi;
  const i64 lastc = hi - 1;
  // the record of a slice is fetched with ONE vector load TWO slices ahead (lane l reads dword l & 31) and its
  // wave-uniform fields are broadcast with v_readlane
  int rec = first < hi ? srec[SL_REC * first + l31] : 0;
  int rec_n = srec[SL_REC * (first + wstride < hi ? first + wstride : (first < hi ? lastc : 0)) + l31];
  for (i64 chunk = first; chunk < hi; chunk += wstride) {
    const i64 cnn = chunk + 2 * wstride;
    const int rec_nn = srec[SL_REC * (cnn < hi ? cnn : lastc) + l31];
    SDesc d;
    d.r0 = rlane(rec, 0); d.meta = rlane(rec, 1); d.base = rlane(rec, 2); d.aux = rlane(rec, 3);
    const int nrows = d.meta & 255, type = (d.meta >> 8) & 3, cnt = d.meta >> 16;
    const bool skip = ((dbg & 1) && type == SL_G) || ((dbg & 2) && type != SL_G);
    if (skip) {
    } else if (type == SL_U) {
      slice_dispatch<false, MODE, NT>(cnt, d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, fin.pc0, fin.pc1, fin.pc2);
    } else if (type == SL_P) {
      slice_dispatch<true, MODE, NT>(cnt, d, nrows, rec, pval, x, y, pa, pb, lane, acc0, acc1, acc2, dbg, fin.pc0, fin.pc1, fin.pc2);
    } else {
      // packed irregular rows: same data flow as k_spmv_cw on the compact CSR, plus the row-id indirection
      Desc dd;
      dd.r0 = d.r0; dd.base = d.base; dd.r1 = d.r0 + nrows; dd.end = d.aux;
      StreamW q;
      stream_issue_w<NT>(q, dd, g_rowptr, g_col, g_val, lane);
      const bool live = lane < nrows;
      const int rid = g_rowid[d.r0 + (live ? lane : 0)];
      const int basev = dd.base & ~1, basec = dd.base & ~3;
      const int av = q.ra - basev, ac = q.ra - basec, len = q.rb - q.ra;
#pragma unroll
      for (int j = 0; j < XV_IT; ++j) sv2[lane + 64 * j] = q.v[j];
      // the dot operands of the row ride with the gathers (issued before the LDS phase, not after it)
      const double opa = MI::HAS_A ? pa[rid] : 0.0, opb = MI::HAS_B ? pb[rid] : 0.0;
      const double xown = MI::OWN ? x[rid] : 0.0;
      // ENTRY-parallel x gathers: lane l holds the columns of entries 4l..4l+3 (and 256 + 4l..), so neighbouring lanes
      // gather for the same or the next row and one gather instruction touches ~20 cache lines -- a row per lane (the CSR
      // kernel's way) touches 64, and the vector-memory unit serves a line per cycle.  The x values go to LDS next to the
      // matrix values; the row sums then read LDS only, in CSR entry order (bitwise the CSR kernels' result).
      double xg[XC_IT][4];
#pragma unroll
      for (int j = 0; j < XC_IT; ++j) {
        xg[j][0] = x[q.c[j].x]; xg[j][1] = x[q.c[j].y]; xg[j][2] = x[q.c[j].z]; xg[j][3] = x[q.c[j].w];
      }
#pragma unroll
      for (int j = 0; j < XC_IT; ++j) {
        d2_t lo2, hi2;
        lo2.x = xg[j][0]; lo2.y = xg[j][1]; hi2.x = xg[j][2]; hi2.y = xg[j][3];
        sx2[2 * (lane + 64 * j)] = lo2;
        sx2[2 * (lane + 64 * j) + 1] = hi2;
      }
      __builtin_amdgcn_wave_barrier();
      double sum = 0.0;
      int maxlen = len;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const int other = __shfl_xor(maxlen, off, 64);
        maxlen = other > maxlen ? other : maxlen;
      }
      maxlen = __builtin_amdgcn_readfirstlane(maxlen);
      for (int j0 = 0; j0 < maxlen; j0 += GUNR) {   // LDS reads of a round issued together
        double vv[GUNR], xv[GUNR];
#pragma unroll
        for (int j = 0; j < GUNR; ++j) {
          const int k = j0 + j < len ? j0 + j : 0;
          vv[j] = sv[av + k];
          xv[j] = sx[ac + k];
        }
#pragma unroll
        for (int j = 0; j < GUNR; ++j)
          if (j0 + j < len) sum += vv[j] * xv[j];
      }
      __builtin_amdgcn_wave_barrier();
      if (live) {
        sum = mode_out<MODE>(sum, xown, opa, opb, fin.pc0, fin.pc1, fin.pc2);
        y[rid] = sum;
        if (MODE == 7) const_cast<double*>(pa)[rid] = opa + fin.pc2 * xown;
        if (MI::DOT0) acc0 += opa * sum;
        if (MI::DOT1) acc1 += sum * sum;
        if (MI::DOT2) acc2 += opb * sum;
      }
    }
    rec = rec_n;
    rec_n = rec_nn;
  }
  }   // phases
  }   // tiles
  // (write-through stores: the last block of this launch may read them, see fold_scalar_phase)
  // pstride = blocks per partial slot (the grid the Krylov workspace was sized for; >= gridDim.x).  accum: this launch
  // covers the rows left out by an earlier launch of the same product (spmv_with_halo: the rows that had to wait for
  // the halo) and adds its sums to that launch's -- which has completed: same stream.
  if (DOTS) {
    const double t0 = block_sum(acc0, s_red);
    if (threadIdx.x == 0) put_partial(partials + blockIdx.x, t0, accum);
  }
  if (MI::DOT1) {
    const double t1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) put_partial(partials + pstride + blockIdx.x, t1, accum);
  }
  if (MI::DOT2) {
    const double t2 = block_sum(acc2, s_red);
    if (threadIdx.x == 0) put_partial(partials + 4 * (size_t)pstride + blockIdx.x, t2, accum);
  }
  if (DOTS) fold_scalar_phase(fin, partials, s_red, pstride);
}

// flags[r]: bit 0 = row r has the count and (col - row) offsets of row r-1, bit 1 = and bitwise the same values,
// bit 2 = row r references a ghost column (col >= n: its product needs the halo exchange of x to have landed)
__global__ void k_row_same(i64 n, const int* __restrict__ rowptr, const int* __restrict__ col,
                           const double* __restrict__ val, unsigned char* __restrict__ flags) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    unsigned char f = 0;
    bool ghost = false;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) ghost = ghost || col[k] >= n;
    if (r > 0) {
      const int a = rowptr[r], b = rowptr[r + 1], pa = rowptr[r - 1];
      if (b - a == a - pa) {
        bool same = true, vsame = true;
        for (int k = 0; k < b - a; ++k) {
          same = same && (col[a + k] - 1 == col[pa + k]);
          vsame = vsame && (__double_as_longlong(val[a + k]) == __double_as_longlong(val[pa + k]));
        }
        f = same ? (vsame ? 3 : 1) : 0;
      }
    }
    flags[r] = f | (ghost ? 4 : 0);
  }
}

// P slices: pval[base + j*rows + l] = val[rowptr[r0+l] + j]
__global__ void k_pval_fill(i64 nslices, const int* __restrict__ srec, const int* __restrict__ rowptr,
                            const double* __restrict__ val, double* __restrict__ pval) {
  for (i64 sidx = blockIdx.x; sidx < nslices; sidx += gridDim.x) {
    const int r0 = srec[SL_REC * sidx], meta = srec[SL_REC * sidx + 1], base = srec[SL_REC * sidx + 2];
    if (((meta >> 8) & 3) != SL_P) continue;
    const int nrows = meta & 255, cnt = meta >> 16;
    for (int q = threadIdx.x; q < nrows * cnt; q += blockDim.x) {
      const int l = q / cnt, j = q % cnt;
      pval[(i64)base + (i64)j * nrows + l] = val[rowptr[r0 + l] + j];
    }
  }
}

// packed CSR of the irregular rows
__global__ void k_gpack(i64 ng, const int* __restrict__ g_rowid, const int* __restrict__ g_rowptr,
                        const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                        int* __restrict__ g_col, double* __restrict__ g_val) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < ng; q += (i64)gridDim.x * blockDim.x) {
    const int r = g_rowid[q], a = rowptr[r], len = rowptr[r + 1] - a, ga = g_rowptr[q];
    for (int k = 0; k < len; ++k) {
      g_col[ga + k] = col[a + k];
      g_val[ga + k] = val[a + k];
    }
  }
}

// offsets (dwords 4..11) and values (dwords 12..27) of every U / P record, from the slice's first row
__global__ void k_fill_records(i64 nslices, int* __restrict__ srec, const int* __restrict__ rowptr,
                               const int* __restrict__ col, const double* __restrict__ val) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nslices; q += (i64)gridDim.x * blockDim.x) {
    int* rec = srec + SL_REC * q;
    const int type = (rec[1] >> 8) & 3;
    if (type == SL_G) continue;
    const int r = rec[0], a = rowptr[r], len = rowptr[r + 1] - a;
    for (int k = 0; k < SL_MAXCNT; ++k) {
      rec[4 + k] = col[a + (k < len ? k : 0)] - r;
      const double v = (k < len && type == SL_U) ? val[a + k] : 0.0;
      rec[12 + 2 * k] = __double2loint(v);
      rec[13 + 2 * k] = __double2hiint(v);
    }
  }
}

int variant() {
  // default 70 = stencil slices + non-temporal streams; 38 = the chunked CSR kernel + non-temporal streams; 1 = first kernel
  return config().spmv_variant;
}

int xcd_map() {
  return config().spmv_xcd;
}

}  // namespace
void ensure_csr_chunks(const CsrMatrix& A);
namespace {

#define PG_LAUNCH_C(KERNEL)                                                                                    \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL), dim3(grid), dim3(BLOCK), 0, st, A.n, A.nchunks, A.chunk_desc.p, \
                     A.rowptr.p, A.col.p, A.val.p, x, y, aux, partials, sc)

// slices [s0, s0 + ns) of the slice image (the whole image: 0, A.nslices); `grid` blocks, partial slots `pstride` apart
template <int MODE>
bool launch_slices(int v, const CsrMatrix& A, i64 s0, i64 ns, const double* x, double* y, const double* aux, double* partials,
                   const double* sc, int grid, int pstride, int accum, hipStream_t st, const FinArgs* fin, bool units = true) {
  FinArgs fa{nullptr, nullptr, PH_NONE, 0, 0, nullptr};
  if (fin && MODE >= 1) fa = *fin;
  const int* rec = A.srec.p + SL_REC * s0;
  const i64 nu = units ? A.nunits : 0;   // the marching units ride with the launch that covers the interior slices
  const bool tiled = units && A.tiles_per_xcd > 0 && s0 == 0 && ns == A.tile_ns && (xcd_map() & 1) && grid >= 8;
  const int* tiles = tiled ? (const int*)A.tiles.p : nullptr;
  const int tpx = tiled ? A.tiles_per_xcd : 0;
  if (xcd_map() >= 256)   // diagnostic switches set: the diagnostic build (wrong products; ablation runs only)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_s<MODE, true, true>), dim3(grid), dim3(BLOCK), 0, st, ns, rec, A.pval.p,
                       A.g_rowid.p, A.g_rowptr.p, A.g_col.p, A.g_val.p, x, y, aux, partials, sc, xcd_map(), fa, pstride, accum,
                       (const int*)A.mrec.p, nu, tiles, tpx);
  else if (v & 4)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_s<MODE, true, false>), dim3(grid), dim3(BLOCK), 0, st, ns, rec, A.pval.p,
                       A.g_rowid.p, A.g_rowptr.p, A.g_col.p, A.g_val.p, x, y, aux, partials, sc, xcd_map(), fa, pstride, accum,
                       (const int*)A.mrec.p, nu, tiles, tpx);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_s<MODE, false, false>), dim3(grid), dim3(BLOCK), 0, st, ns, rec, A.pval.p,
                       A.g_rowid.p, A.g_rowptr.p, A.g_col.p, A.g_val.p, x, y, aux, partials, sc, xcd_map(), fa, pstride, accum,
                       (const int*)A.mrec.p, nu, tiles, tpx);
  return fa.ticket != nullptr;
}

template <int MODE>
bool launch_mode(int v, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
                 const double* sc, int grid, hipStream_t st, const FinArgs* fin) {
  if (v & 64) return launch_slices<MODE>(v, A, 0, A.nslices, x, y, aux, partials, sc, grid, grid, 0, st, fin);
  PG_REQUIRE(MODE < 4 && !(fin && fin->dotx), "the preconditioner products and separate dot operands need the slice kernel");
  if (v != 1) ensure_csr_chunks(A);
  if (v == 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv<MODE>), dim3(grid), dim3(BLOCK), 0, st, A.n, A.rowptr.p, A.col.p, A.val.p, x, y,
                       aux, partials, sc);
    return false;
  }
  // every other variant: the chunked CSR kernel (bit 2), with (bit 4) or without the non-temporal stream hint
  if (v & 4) PG_LAUNCH_C((k_spmv_cw<MODE, true>));
  else PG_LAUNCH_C((k_spmv_cw<MODE, false>));
  return false;
}

}  // namespace


namespace {

struct Slice {
  int r0, meta, base, aux;
  int key;   // first matrix row (sort key)
  int bnd;   // 1: some row references a ghost column (the slice waits for the halo exchange)
  i64 okey = 0;   // place in the order of the work items (march_order of the first row's cell, then the row)
};

using pghost::MRun;   // rows [r0, r0 + len) with one stencil (offsets and values), cnt = 5 / 7 entries

constexpr int RUN_INFO = 25;   // dwords per run: 8 offsets, 8 values (lo, hi), grid cell of the first row (-1: unknown)

// col - row offsets and values of the first row of every run
__global__ void k_run_info(i64 nruns, const int* __restrict__ run_r0, const int* __restrict__ rowptr,
                           const int* __restrict__ col, const double* __restrict__ val, const int* __restrict__ cell,
                           const int* __restrict__ map, int* __restrict__ out) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nruns; q += (i64)gridDim.x * blockDim.x) {
    const int r = run_r0[q], a = rowptr[r], len = rowptr[r + 1] - a;
    out[RUN_INFO * q + 24] = cell ? cell[map ? map[r] : r] : -1;
    for (int k = 0; k < 8; ++k) {
      out[RUN_INFO * q + k] = k < len ? col[a + k] - r : 0;
      const double v = k < len ? val[a + k] : 0.0;
      out[RUN_INFO * q + 8 + 2 * k] = __double2loint(v);
      out[RUN_INFO * q + 9 + 2 * k] = __double2hiint(v);
    }
  }
}

// values (dwords 2..15) of every unit record from its first computed row (dword 1)
__global__ void k_fill_units(i64 nunits, int* __restrict__ mrec, const int* __restrict__ rowptr,
                             const double* __restrict__ val) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nunits; q += (i64)gridDim.x * blockDim.x) {
    int* rec = mrec + MARCH_REC * q;
    const int cnt = (rec[0] >> 8) & 255, a = rowptr[rec[1]];
    for (int k = 0; k < cnt; ++k) {
      rec[2 + 2 * k] = __double2loint(val[a + k]);
      rec[3 + 2 * k] = __double2hiint(val[a + k]);
    }
  }
}

// do the rows of every unit still share one stencil?  (rows after the first of a plane repeat their predecessor --
// row flags -- and the first row of every plane carries the values of the unit's first row)
__global__ void k_units_invalid(i64 nunits, const int* __restrict__ mrec, const int* __restrict__ rowptr,
                                const double* __restrict__ val, const unsigned char* __restrict__ flags,
                                unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (i64 q = blockIdx.x; q < nunits; q += gridDim.x) {
    const int* rec = mrec + MARCH_REC * q;
    const int K = rec[0] & 255, cnt = (rec[0] >> 8) & 255, a0 = rowptr[rec[1]];
    for (int i = 0; i < K; ++i) {
      const int rb = rec[18 + 4 * i], lo = rec[21 + 4 * i] & 255, hi = rec[21 + 4 * i] >> 8;
      for (int l = lo + threadIdx.x; l < hi; l += blockDim.x) {
        const int r = rb + l;
        if (l > lo) {
          c += (flags[r] & 3) != 3 ? 1 : 0;
        } else {
          const int a = rowptr[r];
          bool same = rowptr[r + 1] - a == cnt;
          for (int k = 0; same && k < cnt; ++k)
            same = __double_as_longlong(val[a + k]) == __double_as_longlong(val[a0 + k]);
          c += same ? 0 : 1;
        }
      }
    }
  }
  if (c) atomicAdd(out, c);
}

__global__ void k_row_cells(i64 nrows, const int* __restrict__ rows, const int* __restrict__ cell, const int* __restrict__ map,
                            int* __restrict__ out) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nrows; q += (i64)gridDim.x * blockDim.x) {
    const int r = rows[q];
    out[q] = cell[map ? map[r] : r];
  }
}

int march_strip() {
  return config().spmv_strip;
}
bool has_geo(const CsrMatrix& A) { return A.geo_cell && A.geo_ext0 > 0 && A.geo_lines > 0 && march_strip() > 0; }

// march_order of the given rows (0 for all without grid geometry)
std::vector<i64> row_orders(const CsrMatrix& A, const std::vector<int>& rows) {
  std::vector<i64> ord(rows.size(), 0);
  if (!has_geo(A) || rows.empty()) return ord;
  const i64 nr = (i64)rows.size();
  DevBuf<int> d_rows(nr), d_cells(nr);
  d_rows.upload(rows.data(), nr);
  hipLaunchKernelGGL(k_row_cells, dim3(grid_for(nr, 256)), dim3(256), 0, ctx().stream, nr, d_rows.p, A.geo_cell, A.geo_map, d_cells.p);
  PG_HIP(hipGetLastError());
  std::vector<int> cells(rows.size());
  d_cells.download(cells.data(), nr);
  for (i64 q = 0; q < nr; ++q) ord[q] = pghost::march_order(cells[q], A.geo_ext0, A.geo_lines, march_strip());
  return ord;
}

void emit_u_rows(std::vector<Slice>& up, i64 a, i64 b, int cnt) {
  for (i64 r = a; r < b; r += SL_MAXROWS_U) {
    const int rows = (int)std::min<i64>(SL_MAXROWS_U, b - r);
    up.push_back(Slice{(int)r, rows | (SL_U << 8) | (cnt << 16), 0, 0, (int)r, 0});
  }
}

// Chains of runs -> unit records: the planning itself is plain host code (pg_host_algos.h: also compiled, with
// AddressSanitizer / UBSan, into the CPU test-suite); here: the run descriptors come from the device and the result goes
// back.  Rows that cannot march are appended to `up` as U slices.
void build_march_units(CsrMatrix& A, const std::vector<MRun>& runs, std::vector<int>& mrec, std::vector<Slice>& up,
                       std::vector<i64>& ukeys) {
  A.rows_m = 0;
  const i64 nr = (i64)runs.size();
  if (nr == 0) return;
  hipStream_t st = ctx().stream;
  std::vector<int> info((size_t)RUN_INFO * nr);
  {
    std::vector<int> r0s(nr);
    for (i64 i = 0; i < nr; ++i) r0s[i] = runs[i].r0;
    DevBuf<int> d_r0(nr), d_info(RUN_INFO * nr);
    d_r0.upload(r0s.data(), nr);
    hipLaunchKernelGGL(k_run_info, dim3(grid_for(nr, 256)), dim3(256), 0, st, nr, d_r0.p, A.rowptr.p, A.col.p, A.val.p,
                       A.geo_cell, A.geo_map, d_info.p);
    PG_HIP(hipGetLastError());
    d_info.download(info.data(), RUN_INFO * nr);
  }
  const int kmax = config().spmv_march_k > 0 ? std::max(1, std::min(MARCH_K, config().spmv_march_k)) : MARCH_K;
  pghost::MarchGeometry geo{MARCH_K, MARCH_KS, MARCH_REC, MARCH_W, RUN_INFO, kmax, A.geo_ext0, A.geo_lines, march_strip(),
                            has_geo(A) ? config().unit_order : 0};
  std::vector<pghost::RowRange> fallback;
  i64 rows_m = 0;
  pghost::plan_march_units(A.n, runs, info, geo, mrec, fallback, rows_m, &ukeys);
  A.rows_m = rows_m;
  for (const auto& f : fallback) emit_u_rows(up, f.a, f.b, f.cnt);
}

// stencil-slice image of A (see "stencil slices" above); rp = host copy of A.rowptr
void build_slices(CsrMatrix& A, const int* rp) {
  hipStream_t st = ctx().stream;
  Laps laps;
  const i64 n = A.n;
  const int minrun = config().spmv_minrun;
  unsigned char* fl = static_cast<unsigned char*>(pinned_scratch(1, (size_t)n));
  A.rowflags.alloc(n);
  hipLaunchKernelGGL(k_row_same, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, n, A.rowptr.p, A.col.p, A.val.p,
                     A.rowflags.p);
  PG_HIP(hipGetLastError());
  A.rowflags.download(fl, n);
  laps.lap("    slices: row flags");
  // run classification over [lo, hi) (lo, hi on pattern-run boundaries): independent per range, so the 10^7-row
  // sweep is split over a few host threads and the pieces are concatenated in row order
  struct Part {
    std::vector<Slice> up;      // U and P slices
    std::vector<int> grows;     // irregular rows, ascending
    std::vector<MRun> runs;     // U runs of 2-D / 3-D stencil rows: candidates for marching units
    i64 rows_u = 0, rows_p = 0, nnz_p = 0;
  };
  const bool march_env = config().spmv_march;
  const bool march_on = march_env && A.want_units;
  auto classify = [&](i64 lo, i64 hi, Part& out) {
    auto emit = [&](int type, i64 a, i64 b, int cnt) {
      if (march_on && type == SL_U && (cnt == 5 || cnt == 7)) {
        // a piece of a grid line with one stencil: kept whole, turned into marching units (or U slices) below
        bool ghost = false;
        for (i64 q = a; q < b; ++q) ghost = ghost || ((fl[q] >> 2) & 1);
        if (!ghost) {
          out.runs.push_back(MRun{(int)a, (int)(b - a), cnt});
          out.rows_u += b - a;
          return;
        }
      }
      const i64 maxrows = type == SL_U ? SL_MAXROWS_U : SL_MAXROWS_P;
      for (i64 r = a; r < b; r += maxrows) {
        const int rows = (int)std::min<i64>(maxrows, b - r);
        int bnd = 0;
        for (i64 q = r; q < r + rows; ++q) bnd |= (fl[q] >> 2) & 1;
        out.up.push_back(Slice{(int)r, rows | (type << 8) | (cnt << 16), 0, 0, (int)r, bnd});
      }
      (type == SL_U ? out.rows_u : out.rows_p) += b - a;
      if (type == SL_P) out.nnz_p += (b - a) * cnt;
    };
    auto irregular = [&](i64 a, i64 b) {
      for (i64 r = a; r < b; ++r) out.grows.push_back((int)r);
    };
    i64 i = lo;
    while (i < hi) {
      i64 j = i + 1;
      while (j < n && (fl[j] & 1)) ++j;                 // pattern run [i, j)
      const int cnt = rp[i + 1] - rp[i];
      if (j - i >= minrun && cnt >= 1 && cnt <= SL_MAXCNT) {   // rows with more entries than a record holds stay irregular
        i64 pstart = i, k = i;
        auto flush_p = [&](i64 a, i64 b) {
          if (b - a >= minrun) emit(SL_P, a, b, cnt);
          else irregular(a, b);
        };
        while (k < j) {
          i64 m = k + 1;
          while (m < j && (fl[m] & 2)) ++m;             // value run [k, m)
          if (m - k >= minrun) {
            flush_p(pstart, k);
            emit(SL_U, k, m, cnt);
            pstart = m;
          }
          k = m;
        }
        flush_p(pstart, j);
      } else {
        irregular(i, j);
      }
      i = j;
    }
  };
  const int nthreads = n > (i64)1 << 20 ? (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
  std::vector<i64> cut(nthreads + 1, n);
  cut[0] = 0;
  for (int t = 1; t < nthreads; ++t) {
    i64 c = n * t / nthreads;
    while (c < n && (fl[c] & 1)) ++c;   // move to the start of a pattern run
    cut[t] = std::max(c, cut[t - 1]);
  }
  std::vector<Part> parts(nthreads);
  if (nthreads == 1) {
    classify(0, n, parts[0]);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t] { classify(cut[t], cut[t + 1], parts[t]); });
    for (auto& x : th) x.join();
  }
  std::vector<Slice> up;
  std::vector<int> grows;
  std::vector<MRun> runs;
  A.rows_u = A.rows_p = A.rows_g = A.nnz_p = A.nnz_g = 0;
  {
    size_t nu = 0, ngr = 0, nr = 0;
    for (auto& pt : parts) { nu += pt.up.size(); ngr += pt.grows.size(); nr += pt.runs.size(); }
    up.reserve(nu);
    grows.reserve(ngr);
    runs.reserve(nr);
    for (auto& pt : parts) {
      up.insert(up.end(), pt.up.begin(), pt.up.end());
      grows.insert(grows.end(), pt.grows.begin(), pt.grows.end());
      runs.insert(runs.end(), pt.runs.begin(), pt.runs.end());
      A.rows_u += pt.rows_u; A.rows_p += pt.rows_p; A.nnz_p += pt.nnz_p;
    }
  }
  laps.lap("    slices: host classification");
  // marching units out of the stencil runs; what cannot march comes back as U slices
  std::vector<int> mrec;
  std::vector<i64> ukeys;
  build_march_units(A, runs, mrec, up, ukeys);
  A.nunits = (i64)mrec.size() / MARCH_REC;
  A.mrec.alloc(std::max<i64>((i64)mrec.size(), 1));
  if (!mrec.empty()) A.mrec.upload(mrec.data(), (i64)mrec.size());
  laps.lap("    slices: marching units");
  // P value bases
  {
    i64 pbase = 0;
    for (auto& sl : up) {
      if (((sl.meta >> 8) & 3) != SL_P) continue;
      const int rows = sl.meta & 255, cnt = sl.meta >> 16;
      PG_REQUIRE(pbase + (i64)rows * cnt < (i64)2147483647, "P-slice value array exceeds int32 indexing");
      sl.base = (int)pbase;
      pbase += (i64)rows * cnt;
    }
  }
  A.pval.alloc(A.nnz_p + 8);
  // packed CSR of the irregular rows + its chunks: the rows that reference a ghost column go last, so that no chunk
  // mixes rows that can be multiplied before the halo has landed with rows that cannot
  // ... in the order of the work items (strip of lines, plane, row: the chunks of a tile gather x where the tile's units
  // have just read it) -- a counting sort, the rows arrive in ascending order
  std::vector<i64> g_ord = row_orders(A, grows);
  if (has_geo(A) && !grows.empty()) {
    std::vector<i64> uniq(g_ord);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    std::vector<i64> start(uniq.size() + 1, 0);
    std::vector<int> bucket(grows.size());
    for (size_t q = 0; q < grows.size(); ++q) {
      bucket[q] = (int)(std::lower_bound(uniq.begin(), uniq.end(), g_ord[q]) - uniq.begin());
      ++start[bucket[q] + 1];
    }
    for (size_t b = 0; b < uniq.size(); ++b) start[b + 1] += start[b];
    std::vector<int> sorted(grows.size());
    std::vector<i64> sorted_ord(grows.size());
    for (size_t q = 0; q < grows.size(); ++q) {
      const i64 pos = start[bucket[q]]++;
      sorted[pos] = grows[q];
      sorted_ord[pos] = g_ord[q];
    }
    grows.swap(sorted);
    g_ord.swap(sorted_ord);
  }
  {   // (the order keys travel with the rows through the partition)
    std::vector<std::pair<int, i64>> both(grows.size());
    for (size_t q = 0; q < grows.size(); ++q) both[q] = {grows[q], g_ord[q]};
    std::stable_partition(both.begin(), both.end(), [&](const std::pair<int, i64>& e) { return (fl[e.first] & 4) == 0; });
    for (size_t q = 0; q < grows.size(); ++q) { grows[q] = both[q].first; g_ord[q] = both[q].second; }
  }
  const i64 ng = (i64)grows.size();
  i64 ng_int = 0;
  while (ng_int < ng && (fl[grows[ng_int]] & 4) == 0) ++ng_int;
  A.rows_g = ng;
  std::vector<int> grp(ng + 1, 0);
  for (i64 q = 0; q < ng; ++q) grp[q + 1] = grp[q] + (rp[grows[q] + 1] - rp[grows[q]]);
  A.nnz_g = grp[ng];
  A.g_rowid.alloc(ng > 0 ? ng : 1);
  A.g_rowptr.alloc(ng + 1);
  A.g_col.alloc(A.nnz_g + 8);
  A.g_val.alloc(A.nnz_g + 8);
  A.g_col.zero();
  A.g_val.zero();
  A.g_rowptr.upload(grp.data(), ng + 1);
  {
    std::vector<int> firsts(up.size());
    for (size_t q = 0; q < up.size(); ++q) firsts[q] = up[q].key;
    const std::vector<i64> ord = row_orders(A, firsts);
    for (size_t q = 0; q < up.size(); ++q) up[q].okey = ord[q] * ((i64)1 << 32) + up[q].key;
  }
  std::vector<Slice> all(up);
  if (ng > 0) {
    A.g_rowid.upload(grows.data(), ng);
    hipLaunchKernelGGL(k_gpack, dim3(grid_for(ng, 256)), dim3(256), 0, st, ng, A.g_rowid.p, A.g_rowptr.p, A.rowptr.p, A.col.p,
                       A.val.p, A.g_col.p, A.g_val.p);
    PG_HIP(hipGetLastError());
    i64 q = 0;
    while (q < ng) {
      i64 e = q + 1;
      PG_REQUIRE(grp[e] - grp[q] <= SPMV_CHUNK_ENTRIES, "CSR row longer than an SpMV chunk");
      const i64 stop = q < ng_int ? ng_int : ng;   // chunks end at the interior / boundary seam
      while (e < stop && e - q < 64 && grp[e + 1] - grp[q] <= SPMV_CHUNK_ENTRIES) ++e;
      all.push_back(Slice{(int)q, (int)(e - q) | (SL_G << 8), grp[q], grp[e], grows[q], q >= ng_int ? 1 : 0,
                          g_ord[q] * ((i64)1 << 32) + grows[q]});
      q = e;
    }
  }
  // by first row within each group; the slices that wait for the halo form the tail [nslices_int, nslices)
  std::stable_sort(all.begin(), all.end(), [](const Slice& a, const Slice& b) { return a.bnd != b.bnd ? a.bnd < b.bnd : a.okey < b.okey; });
  A.nslices = (i64)all.size();
  A.nslices_int = 0;
  while (A.nslices_int < A.nslices && !all[A.nslices_int].bnd) ++A.nslices_int;
  // tiles (PG_SPMV_TILE_UNITS = units per tile, e.g. 512 = one per resident wave of an XCD; default off): measured at
  // 512^3, tiles of 512 units take 5 MB off the 159 MB a lean launch reads but cost 6 us of its 45 -- a wave then has one
  // unit and about one slice per tile and nothing to prefetch the next record behind
  {
    const int tile_units = config().spmv_tile_units;
    const i64 T = tile_units > 0 ? std::min<i64>(256, (A.nunits / 8 + tile_units / 2) / tile_units) : 0;
    A.tiles_per_xcd = 0;
    A.tile_ns = 0;
    A.tiles.release();
    if (T >= 2 && A.nslices_int > 0 && (i64)ukeys.size() == A.nunits) {
      std::vector<i64> skeys((size_t)A.nslices_int);
      for (i64 q = 0; q < A.nslices_int; ++q) skeys[q] = all[q].okey;
      std::vector<int> tab;
      pghost::plan_tiles(ukeys, skeys, (int)T, tab);
      A.tiles.alloc((i64)tab.size());
      A.tiles.upload(tab.data(), (i64)tab.size());
      A.tiles_per_xcd = (int)T;
      A.tile_ns = A.nslices_int;
    }
  }
  std::vector<int> sd(SL_REC * (A.nslices + 1), 0);
  for (i64 q = 0; q < A.nslices; ++q) {
    int* rec = sd.data() + SL_REC * q;
    rec[0] = all[q].r0; rec[1] = all[q].meta; rec[2] = all[q].base; rec[3] = all[q].aux;
  }
  A.srec.alloc((i64)sd.size());
  A.srec.upload(sd.data(), (i64)sd.size());
  if (A.nslices > 0) {
    hipLaunchKernelGGL(k_fill_records, dim3(grid_for(A.nslices, 256)), dim3(256), 0, st, A.nslices, A.srec.p, A.rowptr.p, A.col.p,
                       A.val.p);
    PG_HIP(hipGetLastError());
  }
  if (A.nnz_p > 0) {
    hipLaunchKernelGGL(k_pval_fill, dim3((unsigned)std::min<i64>(A.nslices, 65535)), dim3(256), 0, st, A.nslices, A.srec.p,
                       A.rowptr.p, A.val.p, A.pval.p);
    PG_HIP(hipGetLastError());
  }
  PG_HIP(hipStreamSynchronize(st));
  laps.lap("    slices: packing + records");
  A.spmv_bytes = 4 * SL_REC * A.nslices + 4 * MARCH_REC * A.nunits + 8 * A.nnz_p + 12 * A.nnz_g + 8 * ng + 16 * n;
  if (config().debug)
    fprintf(stderr, "[pg_spmv] slices %lld (%lld wait for the halo), marching units %lld (%lld rows): rows U %lld P %lld G %lld of %lld; nnz P %lld G %lld of %lld; bytes/launch %lld (CSR %lld)\n",
            (long long)A.nslices, (long long)(A.nslices - A.nslices_int), (long long)A.nunits, (long long)A.rows_m, (long long)A.rows_u, (long long)A.rows_p, (long long)A.rows_g,
            (long long)n, (long long)A.nnz_p, (long long)A.nnz_g, (long long)A.nnz, (long long)A.spmv_bytes,
            (long long)(12 * A.nnz + 20 * n));
}

}  // namespace

// rows -> chunks of <= 64 rows and <= SPMV_CHUNK_ENTRIES entries for the plain-CSR kernels (greedy, host side).  Only
// the CSR variants need it: built on first use.
void ensure_csr_chunks(const CsrMatrix& Ac) {
  CsrMatrix& A = const_cast<CsrMatrix&>(Ac);
  if (A.chunk_desc.p || A.n == 0) return;
  std::vector<int> rp(A.n + 1);
  A.rowptr.download(rp.data(), A.n + 1);
  std::vector<int> cd;   // {first row, first entry} per chunk, closed by {n, nnz}; padded for the 4-int descriptor read
  cd.reserve(2 * (A.n / 60 + 16));
  i64 r = 0;
  while (r < A.n) {
    i64 e = r + 1;   // a chunk holds at least one row (rows never exceed the slice: <= 4(2N+1) entries)
    PG_REQUIRE(rp[e] - rp[r] <= SPMV_CHUNK_ENTRIES, "CSR row longer than an SpMV chunk");
    while (e < A.n && e - r < 64 && rp[e + 1] - rp[r] <= SPMV_CHUNK_ENTRIES) ++e;
    cd.push_back((int)r);
    cd.push_back(rp[r]);
    r = e;
  }
  A.nchunks = (i64)cd.size() / 2;
  cd.push_back((int)A.n);
  cd.push_back(rp[A.n]);
  cd.push_back((int)A.n);   // pad
  cd.push_back(rp[A.n]);
  A.chunk_desc.alloc((i64)cd.size());
  A.chunk_desc.upload(cd.data(), (i64)cd.size());
}

// SpMV image of a freshly assembled matrix (once per matrix)
void build_spmv_chunks(CsrMatrix& A) {
  Laps laps;
  int* rp = static_cast<int*>(pinned_scratch(0, sizeof(int) * (size_t)(A.n + 1)));
  A.rowptr.download(rp, A.n + 1);
  laps.lap("    chunks: rowptr download");
  A.chunk_desc.release();
  A.nchunks = 0;
  build_slices(A, rp);
}

// do the new row flags still support every slice?  (U: rows 2.. repeat offsets and values; P: offsets.)  More repetition
// than before is fine -- it is just not exploited; rows outside the slices (G) need nothing.
__global__ void k_slices_invalid(i64 nslices, const int* __restrict__ srec, const unsigned char* __restrict__ flags,
                                 unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (i64 q = blockIdx.x; q < nslices; q += gridDim.x) {
    const int r0 = srec[SL_REC * q], meta = srec[SL_REC * q + 1];
    const int nrows = meta & 255, type = (meta >> 8) & 3;
    const unsigned char need = type == SL_U ? 3 : 1;
    for (int l = 1 + threadIdx.x; l < nrows; l += blockDim.x) c += (flags[r0 + l] & need) != need ? 1 : 0;
  }
  if (c) atomicAdd(out, c);
}

template <class T>
static void clone_buf(DevBuf<T>& dst, const DevBuf<T>& src, hipStream_t st) {
  dst.alloc(src.n > 0 ? src.n : 1);
  if (src.n > 0) PG_HIP(hipMemcpyAsync(dst.p, src.p, sizeof(T) * (size_t)src.n, hipMemcpyDeviceToDevice, st));
}

bool build_slices_like(const CsrMatrix& T, CsrMatrix& A) {
  hipStream_t st = ctx().stream;
  const i64 n = A.n;
  if (!T.rowflags.p || T.n != n || !T.srec.p) return false;
  A.rowflags.alloc(n);
  DevBuf<unsigned long long> diff(1);
  diff.zero();
  hipLaunchKernelGGL(k_row_same, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, n, A.rowptr.p, A.col.p, A.val.p,
                     A.rowflags.p);
  if (T.nslices > 0)
    hipLaunchKernelGGL(k_slices_invalid, dim3((unsigned)std::min<i64>(T.nslices, 16384)), dim3(64), 0, st, T.nslices, T.srec.p,
                       A.rowflags.p, diff.p);
  if (T.nunits > 0)
    hipLaunchKernelGGL(k_units_invalid, dim3((unsigned)std::min<i64>(T.nunits, 16384)), dim3(64), 0, st, T.nunits, T.mrec.p,
                       A.rowptr.p, A.val.p, A.rowflags.p, diff.p);
  PG_HIP(hipGetLastError());
  unsigned long long h = 0;
  diff.download(&h, 1);
  if (h != 0) return false;
  // every slice still holds: copy the structure, refill the values
  A.chunk_desc.release();
  A.nchunks = 0;
  A.nslices = T.nslices;
  A.nslices_int = T.nslices_int;
  A.rows_u = T.rows_u; A.rows_p = T.rows_p; A.rows_g = T.rows_g; A.nnz_p = T.nnz_p; A.nnz_g = T.nnz_g;
  A.spmv_bytes = T.spmv_bytes;
  A.nunits = T.nunits; A.rows_m = T.rows_m;
  A.tiles_per_xcd = T.tiles_per_xcd; A.tile_ns = T.tile_ns;
  if (T.tiles_per_xcd > 0) clone_buf(A.tiles, T.tiles, st);
  else A.tiles.release();
  clone_buf(A.mrec, T.mrec, st);
  if (A.nunits > 0)
    hipLaunchKernelGGL(k_fill_units, dim3(grid_for(A.nunits, 256)), dim3(256), 0, st, A.nunits, A.mrec.p, A.rowptr.p, A.val.p);
  clone_buf(A.srec, T.srec, st);
  clone_buf(A.g_rowid, T.g_rowid, st);
  clone_buf(A.g_rowptr, T.g_rowptr, st);
  A.pval.alloc(A.nnz_p + 8);
  A.g_col.alloc(A.nnz_g + 8);
  A.g_val.alloc(A.nnz_g + 8);
  A.g_col.zero();
  A.g_val.zero();
  if (A.nslices > 0)
    hipLaunchKernelGGL(k_fill_records, dim3(grid_for(A.nslices, 256)), dim3(256), 0, st, A.nslices, A.srec.p, A.rowptr.p, A.col.p,
                       A.val.p);
  if (A.nnz_p > 0)
    hipLaunchKernelGGL(k_pval_fill, dim3((unsigned)std::min<i64>(A.nslices, 65535)), dim3(256), 0, st, A.nslices, A.srec.p,
                       A.rowptr.p, A.val.p, A.pval.p);
  if (A.rows_g > 0)
    hipLaunchKernelGGL(k_gpack, dim3(grid_for(A.rows_g, 256)), dim3(256), 0, st, A.rows_g, A.g_rowid.p, A.g_rowptr.p, A.rowptr.p,
                       A.col.p, A.val.p, A.g_col.p, A.g_val.p);
  PG_HIP(hipGetLastError());
  PG_HIP(hipStreamSynchronize(st));
  return true;
}

bool spmv_supports_preconditioner_product() { return (variant() & 64) != 0; }

int spmv_default_grid(i64 n) {
  // resident blocks per CU: the slice kernel holds ~100 VGPRs (4 waves / SIMD), the CSR kernels 24.6 KB of LDS (6 blocks)
  const int per_cu = config().spmv_blocks_per_cu > 0 ? config().spmv_blocks_per_cu : ((variant() & 64) ? 4 : 6);
  static int cus = 0;
  if (cus == 0) {
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDeviceProperties(&prop, ctx().device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  return grid_for(n, BLOCK, cus * per_cu);
}

bool launch_spmv(int mode, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
                 const double* sc, int grid, hipStream_t st, const FinArgs* fin) {
  if (A.n == 0) return false;
  const int v = variant();
  if (mode == 0) return launch_mode<0>(v, A, x, y, aux, partials, sc, grid, st, fin);
  if (mode == 1) return launch_mode<1>(v, A, x, y, aux, partials, sc, grid, st, fin);
  if (mode == 2) return launch_mode<2>(v, A, x, y, aux, partials, sc, grid, st, fin);
  if (mode == 4) return launch_mode<4>(v, A, x, y, aux, partials, sc, grid, st, fin);
  if (mode == 5) return launch_mode<5>(v, A, x, y, aux, partials, sc, grid, st, fin);
  if (mode == 6) return launch_mode<6>(v, A, x, y, aux, partials, sc, grid, st, fin);
  if (mode == 7) return launch_mode<7>(v, A, x, y, aux, partials, sc, grid, st, fin);
  if (mode == 8) return launch_mode<8>(v, A, x, y, aux, partials, sc, grid, st, fin);
  PG_REQUIRE(mode == 3, "unknown SpMV launch mode");
  return launch_mode<3>(v, A, x, y, aux, partials, sc, grid, st, fin);
}

bool spmv_with_halo(int mode, const CsrMatrix& A, const Numbering& nb, const Slab& slab, double* x, double* y, const double* aux,
                    double* partials, const double* sc, int grid, hipStream_t st, const FinArgs* fin) {
  if (A.n == 0 || !A.halo_needed) {
    // (a rank without rows still takes part in the exchange: its neighbours receive from it)
    if (A.halo_needed) halo_exchange(nb, slab, x, st);
    return launch_spmv(mode, A, x, y, aux, partials, sc, grid, st, fin);
  }
  const int v = variant();
  const bool overlap = config().halo_overlap;
  const i64 ni = A.nslices_int, nbnd = A.nslices - A.nslices_int;
  if (!(v & 64) || !overlap || (ni == 0 && A.nunits == 0) || nbnd == 0) {
    halo_exchange(nb, slab, x, st);
    return launch_spmv(mode, A, x, y, aux, partials, sc, grid, st, fin);
  }
  // blocks of the second launch: one wave per slice, never more blocks than partial-sum entries per slot
  const int grid2 = (int)std::max<i64>(1, std::min<i64>(grid, (nbnd + BLOCK / 64 - 1) / (BLOCK / 64)));
  halo_begin(nb, slab, x, st);                      // x's owned part is final on `st`; ghosts arrive on the comm stream
  bool folded;
  FinArgs fin_nofold{nullptr, nullptr, PH_NONE, 0, 0, fin ? fin->dotx : nullptr};   // first launch: dot operand, no scalar phase
  if (fin) { fin_nofold.pc0 = fin->pc0; fin_nofold.pc1 = fin->pc1; fin_nofold.pc2 = fin->pc2; fin_nofold.base = fin->base; fin_nofold.accv = fin->accv; }
#define PG_SPLIT(MODE_)                                                                                              \
  launch_slices<MODE_>(v, A, 0, ni, x, y, aux, partials, sc, grid, grid, 0, st, fin ? &fin_nofold : nullptr);      \
  halo_end(st);                                                                                                      \
  folded = launch_slices<MODE_>(v, A, ni, nbnd, x, y, aux, partials, sc, grid2, grid, 1, st, fin, false)
  if (mode == 0) { PG_SPLIT(0); }
  else if (mode == 1) { PG_SPLIT(1); }
  else if (mode == 2) { PG_SPLIT(2); }
  else if (mode == 4) { PG_SPLIT(4); }
  else if (mode == 5) { PG_SPLIT(5); }
  else if (mode == 6) { PG_SPLIT(6); }
  else if (mode == 7) { PG_SPLIT(7); }
  else if (mode == 8) { PG_SPLIT(8); }
  else { PG_SPLIT(3); }
#undef PG_SPLIT
  return folded;
}

void launch_spmv_variant(int v, const CsrMatrix& A, const double* x, double* y, hipStream_t st) {
  if (A.n == 0) return;
  launch_mode<0>(v, A, x, y, nullptr, nullptr, nullptr, spmv_default_grid(A.n), st, nullptr);
}

}  // namespace pg
