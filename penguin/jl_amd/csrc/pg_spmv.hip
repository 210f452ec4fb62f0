// pg_spmv.hip -- the dominant kernel of the path: y = A x for the reduced cut-cell system in CSR
// (fp64 values, int32 column indices, int32 row pointers; rows in box order, 1 / 2N+1 / up to 2(2N+1) entries).
// Replaces the single-threaded CSC mul! inside IterativeSolvers (src/solver.jl:178-181).
//
// Roofline: HBM.  Algorithmic bytes per launch = 12 nnz + 20 n; x (8 n bytes, 82 MB at 512^3)
// is gathered out of L2 / Infinity Cache.
//
// Design (CDNA4): the rows are cut once per matrix into CHUNKS of <= 64 consecutive rows and <= 512 entries
// (CsrMatrix::chunk_desc = {first row, first entry} per chunk, the row-block idea of CSR-Adaptive).  One WAVE
// owns one chunk per iteration.  The chunk's val/col entries are one contiguous range of the CSR arrays; the wave
// streams it with fully coalesced loads into its PRIVATE slice of LDS -- no block barrier anywhere, a wave
// stalls only on its own loads -- and then lane l walks row l out of LDS: column indices, all x gathers issued
// together, FMAs.  For a fixed stencil slot consecutive lanes gather consecutive x entries, so the gathers
// coalesce too.
//
// Everything inside an iteration is branch-free with a FIXED number of vector-memory operations (clamped
// indices + selects).  Reasons, all seen in the ISA / PMC (profiles/):
//   (1) hipcc puts every predicated load in its own basic block behind `s_waitcnt vmcnt(0)`: the gathers serialise;
//   (2) vector-memory operations retire in order, so overlapping the stream of chunk i+1 with the gathers of chunk
//       i needs a COUNTED wait -- `s_waitcnt vmcnt(N)` lets the gathers complete while the N younger stream loads
//       stay in flight -- and the compiler can only count when every path issues the same number of loads;
//   (3) the chunk descriptor is a scalar load the stream addresses depend on: it is fetched TWO chunks ahead
//       (3-stage software pipeline: descriptor(i+2) | stream(i+1) | gathers+FMA(i)), otherwise its latency sits in
//       front of every stream (ablation: the kernel ran at the same 0.20 ms with the gathers removed).
//
// The dots that follow an SpMV in BiCGStab / CG are fused into the epilogue (one partial per block, summed in
// fixed order by k_finalize: deterministic).
//
// Variants (PG_SPMV_VARIANT; default = best measured, see profiles/):
//   1   block of 256 rows staged through LDS, two block barriers per chunk     (first version, 2.0 TB/s)
//   2   chunked, wave-private LDS slices, branch-free, no pipelining           (bit 1)
//   +4  non-temporal hint on the matrix stream (read once; keeps x in L2 / Infinity Cache)
//   +16 3-stage software pipeline inside the wave
//   +32 16-byte (aligned pair / quad) stream loads
#include <algorithm>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "pg_spmv.h"

#ifndef PG_SPMV_ABLATE
#define PG_SPMV_ABLATE 0   // diagnostics builds only (scripts/spmv_ablate.sh)
#endif

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
  double acc0 = 0.0, acc1 = 0.0;
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
      if (MODE == 2) {
        acc0 += sum * x[r];
        acc1 += sum * sum;
      }
    }
  }
  if (MODE >= 1) {
    const double t0 = block_sum(acc0, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  }
  if (MODE == 2) {
    const double t1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = t1;
  }
}

// ---- chunked wave kernel ---------------------------------------------------------------------------------------
constexpr int WITER = 8;                         // staging loads per lane: 512 slots >= SPMV_CHUNK_ENTRIES
constexpr int WUNR = 8;                          // gathers issued together per row (bulk rows: 2N+1 <= 7)

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

struct Stream {        // registers holding one chunk's raw loads
  int ra, rb;          // this lane's row bounds, ABSOLUTE: nothing consumes them in the issue phase
  double tv[WITER];
  int tc[WITER];
};

// issue every vector load of a chunk -- always 2 + 2*WITER loads, branch-free, no dependent scalar load
template <bool NT>
__device__ inline void stream_issue(Stream& q, const Desc& d, const int* __restrict__ rowptr, const int* __restrict__ col,
                                    const double* __restrict__ val, int lane) {
  const int r = d.r0 + lane;
  q.ra = stream_load<NT>(rowptr + (r < d.r1 ? r : d.r1));
  q.rb = stream_load<NT>(rowptr + (r + 1 < d.r1 ? r + 1 : d.r1));
  const int last = d.end - d.base > 0 ? d.end - d.base - 1 : 0;
#pragma unroll
  for (int j = 0; j < WITER; ++j) {
    const int k = lane + 64 * j;
    const int kk = k < last ? k : last;         // tail lanes re-read the last entry (one cache line)
    q.tv[j] = stream_load<NT>(val + d.base + kk);
    q.tc[j] = stream_load<NT>(col + d.base + kk);
  }
}

template <int MODE, bool NT, bool PIPE>
__global__ __launch_bounds__(BLOCK) void k_spmv_c(i64 n, i64 nchunks, const int* __restrict__ chunk_desc,
                                                  const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const double* __restrict__ val, const double* __restrict__ x,
                                                  double* __restrict__ y, const double* __restrict__ aux,
                                                  double* __restrict__ partials, const double* __restrict__ sc) {
  __shared__ double s_val[BLOCK / 64][64 * WITER];
  __shared__ int s_col[BLOCK / 64][64 * WITER];
  __shared__ double s_red[BLOCK / 64];
  if (sc && sc[S_DONE] != 0.0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* __restrict__ sv = s_val[wave];
  int* __restrict__ scl = s_col[wave];
  double acc0 = 0.0, acc1 = 0.0;
  const i64 first = (i64)blockIdx.x * (BLOCK / 64) + wave, wstride = (i64)gridDim.x * (BLOCK / 64);
  const i64 lastc = nchunks - 1;
  Stream q;
  Desc dcur, dnext;
  if (first < nchunks) {
    dcur = load_desc(chunk_desc, first);
    dnext = load_desc(chunk_desc, first + wstride < nchunks ? first + wstride : lastc);
    if (PIPE) stream_issue<NT>(q, dcur, rowptr, col, val, lane);
  }
  for (i64 chunk = first; chunk < nchunks; chunk += wstride) {
    if (!PIPE) stream_issue<NT>(q, dcur, rowptr, col, val, lane);
    const int r = dcur.r0 + lane;
    const bool live = r < dcur.r1;
    const int a = q.ra - dcur.base, b = q.rb - dcur.base;
#if PG_SPMV_ABLATE == 3
    // ablation: stream only -- no LDS, no gathers: consume the registers directly
    double sum3 = 0.0;
#pragma unroll
    for (int j = 0; j < WITER; ++j) sum3 += q.tv[j] * (double)q.tc[j];
    const i64 c23 = chunk + 2 * wstride;
    const Desc dn23 = load_desc(chunk_desc, c23 < nchunks ? c23 : lastc);
    if (PIPE) stream_issue<NT>(q, dnext, rowptr, col, val, lane);
    if (live) y[r] = sum3 + a + b;
    dcur = dnext;
    dnext = dn23;
    continue;
#endif
    // stream -> LDS slice (fixed 2*WITER stores; slots past the chunk hold copies of its last entry)
#pragma unroll
    for (int j = 0; j < WITER; ++j) {
      sv[lane + 64 * j] = q.tv[j];
      scl[lane + 64 * j] = q.tc[j];
    }
    __builtin_amdgcn_wave_barrier();
    double xv[WUNR], vv[WUNR];
#pragma unroll
    for (int j = 0; j < WUNR; ++j) {
      const int p = a + j < b ? a + j : 0;      // slot 0 always holds a valid column
#if PG_SPMV_ABLATE == 1
      xv[j] = 1.0 + p;                          // ablation: no gathers
#elif PG_SPMV_ABLATE == 2
      xv[j] = x[scl[p] & 1023];                 // ablation: gathers always hit L1/L2
#else
      xv[j] = x[scl[p]];
#endif
      vv[j] = a + j < b ? sv[p] : 0.0;
    }
    // stage 1 of the NEXT iteration and stage 0 of the one after: both behind this chunk's gathers
    const i64 c2 = chunk + 2 * wstride;
    const Desc dnext2 = load_desc(chunk_desc, c2 < nchunks ? c2 : lastc);
    if (PIPE) stream_issue<NT>(q, dnext, rowptr, col, val, lane);   // counted vmcnt keeps these in flight
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < WUNR; ++j) sum += vv[j] * xv[j];
    if (__builtin_expect(b - a > WUNR, 0))
      for (int k = a + WUNR; k < b; ++k) sum += sv[k] * x[scl[k]];   // cut-cell rows (> 8 entries)
    __builtin_amdgcn_wave_barrier();
    if (live) {
#if PG_SPMV_ABLATE == 4
      if (sum == 1.2345e-300) y[r] = sum;   // ablation: no y store
#elif PG_SPMV_ABLATE == 5
      __builtin_nontemporal_store(sum, &y[r]);
#else
      y[r] = sum;
#endif
      if (MODE == 1) acc0 += aux[r] * sum;
      if (MODE == 2) {
        acc0 += sum * x[r];
        acc1 += sum * sum;
      }
    }
    dcur = dnext;
    dnext = dnext2;
  }
  if (MODE >= 1) {
    const double t0 = block_sum(acc0, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  }
  if (MODE == 2) {
    const double t1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = t1;
  }
}

// ---- same kernel with 16-byte stream loads (bit 32) ------------------------------------------------------------
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
__device__ inline void stream_issue_w(StreamW& q, const Desc& d, const int* __restrict__ rowptr,
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

template <int MODE, bool NT, bool PIPE>
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
  double acc0 = 0.0, acc1 = 0.0;
  const i64 first = (i64)blockIdx.x * (BLOCK / 64) + wave, wstride = (i64)gridDim.x * (BLOCK / 64);
  const i64 lastc = nchunks - 1;
  StreamW q;
  Desc dcur, dnext;
  if (first < nchunks) {
    dcur = load_desc(chunk_desc, first);
    dnext = load_desc(chunk_desc, first + wstride < nchunks ? first + wstride : lastc);
    if (PIPE) stream_issue_w<NT>(q, dcur, rowptr, col, val, lane);
  }
  for (i64 chunk = first; chunk < nchunks; chunk += wstride) {
    if (!PIPE) stream_issue_w<NT>(q, dcur, rowptr, col, val, lane);
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
    if (PIPE) stream_issue_w<NT>(q, dnext, rowptr, col, val, lane);
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < WUNR; ++j) sum += vv[j] * xv[j];
    if (__builtin_expect(len > WUNR, 0))
      for (int k = WUNR; k < len; ++k) sum += sv[av + k] * x[scl[ac + k]];
    __builtin_amdgcn_wave_barrier();
    if (live) {
      y[r] = sum;
      if (MODE == 1) acc0 += aux[r] * sum;
      if (MODE == 2) {
        acc0 += sum * x[r];
        acc1 += sum * sum;
      }
    }
    dcur = dnext;
    dnext = dnext2;
  }
  if (MODE >= 1) {
    const double t0 = block_sum(acc0, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  }
  if (MODE == 2) {
    const double t1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = t1;
  }
}

// ---- stencil slices (default) ----------------------------------------------------------------------------------
// The assembler knows what a generic CSR kernel cannot: away from the interface every row of a line carries the
// SAME stencil -- identical (col - row) offsets, and on a uniform mesh bitwise identical (equilibrated) values.
// build_spmv_chunks() detects that once per matrix (k_row_same) and re-slices the rows:
//   U slice   <= 255 consecutive rows, identical offsets and values: the slice is 16 bytes of descriptor + a shared
//             stencil-table entry read through the scalar cache; lane l computes row r0+l from COALESCED x loads.
//             Nothing of the matrix is streamed.
//   P slice   identical offsets, per-row values: values re-laid slot-major (pval[j*rows + l]) so the 8 B/entry
//             stream is coalesced without LDS; the 4 B/entry column stream disappears.
//   G chunk   the remaining irregular rows (cut cells, their neighbours, line ends), packed 64 at a time into a
//             compact CSR of their own and processed exactly like k_spmv_cw (LDS-staged stream + gathers).
// The products are accumulated in the CSR entry order in all three paths, so y is bitwise what the CSR kernels give.
// Slices are sorted by first row; with `xcd` each XCD (blocks are dispatched round-robin over the 8 XCDs) sweeps one
// contiguous eighth of them, so the +-line / +-plane x neighbours are found in that XCD's own L2.
enum { SL_U = 0, SL_P = 1, SL_G = 2 };
constexpr int SL_MAXROWS = 255, SL_MAXCNT = 16;

struct SDesc {
  int r0, meta, base, aux;
};

__device__ inline SDesc load_sdesc(const int* __restrict__ sd, i64 ch) {
  const int* p = sd + 4 * ch;   // wave-uniform address: one s_load_dwordx4
  SDesc d;
  d.r0 = p[0]; d.meta = p[1]; d.base = p[2]; d.aux = p[3];
  return d;
}

template <int MODE, bool NT>
__global__ __launch_bounds__(BLOCK) void k_spmv_s(i64 nslices, const int* __restrict__ sdesc,
                                                  const int* __restrict__ stab_off, const double* __restrict__ stab_val,
                                                  const double* __restrict__ pval, const int* __restrict__ g_rowid,
                                                  const int* __restrict__ g_rowptr, const int* __restrict__ g_col,
                                                  const double* __restrict__ g_val, const double* __restrict__ x,
                                                  double* __restrict__ y, const double* __restrict__ aux,
                                                  double* __restrict__ partials, const double* __restrict__ sc, int xcd) {
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
  double acc0 = 0.0, acc1 = 0.0;
  i64 first, wstride, hi;
  if (xcd && gridDim.x >= 8) {
    const int xid = blockIdx.x & 7;
    const i64 nbx = ((i64)gridDim.x + 7 - xid) >> 3;          // blocks resident on this XCD
    const i64 lo = nslices * xid / 8;
    hi = nslices * (xid + 1) / 8;
    first = lo + (i64)(blockIdx.x >> 3) * (BLOCK / 64) + wave;
    wstride = nbx * (BLOCK / 64);
  } else {
    first = (i64)blockIdx.x * (BLOCK / 64) + wave;
    wstride = (i64)gridDim.x * (BLOCK / 64);
    hi = nslices;
  }
  const i64 lastc = hi - 1;
  SDesc dcur, dnext;
  if (first < hi) {
    dcur = load_sdesc(sdesc, first);
    dnext = load_sdesc(sdesc, first + wstride < hi ? first + wstride : lastc);
  }
  for (i64 chunk = first; chunk < hi; chunk += wstride) {
    const i64 c2 = chunk + 2 * wstride;
    const SDesc dnext2 = load_sdesc(sdesc, c2 < hi ? c2 : lastc);   // two slices ahead: off the critical path
    const int nrows = dcur.meta & 255, type = (dcur.meta >> 8) & 3, cnt = dcur.meta >> 16;
    if (type != SL_G) {
      const int* __restrict__ so = stab_off + 16 * (i64)dcur.aux;
      const double* __restrict__ svl = stab_val + 16 * (i64)dcur.aux;
      int o[WUNR];
      double cf[WUNR];
#pragma unroll
      for (int j = 0; j < WUNR; ++j) { o[j] = so[j]; cf[j] = svl[j]; }   // uniform: scalar loads; padded table
      const double* __restrict__ pv = pval + (type == SL_P ? dcur.base : 0);
      for (int ofs = 0; ofs < nrows; ofs += 64) {
        const int l = ofs + lane;
        const bool live = l < nrows;
        const int ll = live ? l : 0;
        const int r = dcur.r0 + ll;
        double xv[WUNR], vv[WUNR];
#pragma unroll
        for (int j = 0; j < WUNR; ++j) xv[j] = x[r + o[j]];
        if (type == SL_P) {
#pragma unroll
          for (int j = 0; j < WUNR; ++j) {
            const double v = stream_load<NT>(pv + (j < cnt ? j : 0) * nrows + ll);
            vv[j] = j < cnt ? v : 0.0;
          }
        } else {
#pragma unroll
          for (int j = 0; j < WUNR; ++j) vv[j] = cf[j];
        }
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < WUNR; ++j) sum += vv[j] * xv[j];
        if (__builtin_expect(cnt > WUNR, 0))
          for (int j = WUNR; j < cnt; ++j) sum += (type == SL_P ? pv[j * nrows + ll] : svl[j]) * x[r + so[j]];
        if (live) {
          y[r] = sum;
          if (MODE == 1) acc0 += aux[r] * sum;
          if (MODE == 2) {
            acc0 += sum * x[r];
            acc1 += sum * sum;
          }
        }
      }
    } else {
      // packed irregular rows: same data flow as k_spmv_cw on the compact CSR, plus the row-id indirection
      Desc d;
      d.r0 = dcur.r0; d.base = dcur.base; d.r1 = dcur.r0 + nrows; d.end = dcur.aux;
      StreamW q;
      stream_issue_w<NT>(q, d, g_rowptr, g_col, g_val, lane);
      const bool live = lane < nrows;
      const int rid = g_rowid[dcur.r0 + (live ? lane : 0)];
      const int basev = d.base & ~1, basec = d.base & ~3;
      const int av = q.ra - basev, ac = q.ra - basec, len = q.rb - q.ra;
#pragma unroll
      for (int j = 0; j < XV_IT; ++j) sv2[lane + 64 * j] = q.v[j];
#pragma unroll
      for (int j = 0; j < XC_IT; ++j) sc4[lane + 64 * j] = q.c[j];
      __builtin_amdgcn_wave_barrier();
      double xv[WUNR], vv[WUNR];
      const int c0 = d.base - basec;
#pragma unroll
      for (int j = 0; j < WUNR; ++j) {
        const bool ok = j < len;
        xv[j] = x[scl[ok ? ac + j : c0]];
        vv[j] = ok ? sv[av + j] : 0.0;
      }
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < WUNR; ++j) sum += vv[j] * xv[j];
      if (len > WUNR)
        for (int k = WUNR; k < len; ++k) sum += sv[av + k] * x[scl[ac + k]];
      __builtin_amdgcn_wave_barrier();
      if (live) {
        y[rid] = sum;
        if (MODE == 1) acc0 += aux[rid] * sum;
        if (MODE == 2) {
          acc0 += sum * x[rid];
          acc1 += sum * sum;
        }
      }
    }
    dcur = dnext;
    dnext = dnext2;
  }
  if (MODE >= 1) {
    const double t0 = block_sum(acc0, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  }
  if (MODE == 2) {
    const double t1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = t1;
  }
}

// flags[r]: bit 0 = row r has the count and (col - row) offsets of row r-1, bit 1 = and bitwise the same values
__global__ void k_row_same(i64 n, const int* __restrict__ rowptr, const int* __restrict__ col,
                           const double* __restrict__ val, unsigned char* __restrict__ flags) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    unsigned char f = 0;
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
    flags[r] = f;
  }
}

// P slices: pval[base + j*rows + l] = val[rowptr[r0+l] + j]
__global__ void k_pval_fill(i64 nslices, const int* __restrict__ sdesc, const int* __restrict__ rowptr,
                            const double* __restrict__ val, double* __restrict__ pval) {
  for (i64 sidx = blockIdx.x; sidx < nslices; sidx += gridDim.x) {
    const int r0 = sdesc[4 * sidx], meta = sdesc[4 * sidx + 1], base = sdesc[4 * sidx + 2];
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

// offsets / values of the first row of every U / P slice (for the stencil table)
__global__ void k_fetch_stencil(i64 ns, const int* __restrict__ rows, const int* __restrict__ rowptr,
                                const int* __restrict__ col, const double* __restrict__ val, int* __restrict__ off16,
                                double* __restrict__ val16) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < ns; q += (i64)gridDim.x * blockDim.x) {
    const int r = rows[q], a = rowptr[r], len = rowptr[r + 1] - a;
    for (int k = 0; k < SL_MAXCNT; ++k) {
      off16[q * SL_MAXCNT + k] = col[a + (k < len ? k : 0)] - r;
      val16[q * SL_MAXCNT + k] = k < len ? val[a + k] : 0.0;
    }
  }
}

int variant() {
  // default 70 = stencil slices + non-temporal streams.  38 = chunked CSR + non-temporal + 16-byte stream loads (the
  // best plain-CSR kernel; pipelining measured neutral: profiles/r01_spmv_sweeps.txt)
  static const int v = getenv("PG_SPMV_VARIANT") ? atoi(getenv("PG_SPMV_VARIANT")) : 70;
  return v;
}

int xcd_map() {
  static const int v = getenv("PG_SPMV_XCD") ? atoi(getenv("PG_SPMV_XCD")) : 1;
  return v;
}

#define PG_LAUNCH_C(KERNEL)                                                                                    \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL), dim3(grid), dim3(BLOCK), 0, st, A.n, A.nchunks, A.chunk_desc.p, \
                     A.rowptr.p, A.col.p, A.val.p, x, y, aux, partials, sc)

template <int MODE>
void launch_mode(int v, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
                 const double* sc, int grid, hipStream_t st) {
  if (v & 64) {
    if (v & 4)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_s<MODE, true>), dim3(grid), dim3(BLOCK), 0, st, A.nslices, A.sdesc.p,
                         A.stab_off.p, A.stab_val.p, A.pval.p, A.g_rowid.p, A.g_rowptr.p, A.g_col.p, A.g_val.p, x, y, aux,
                         partials, sc, xcd_map());
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_s<MODE, false>), dim3(grid), dim3(BLOCK), 0, st, A.nslices, A.sdesc.p,
                         A.stab_off.p, A.stab_val.p, A.pval.p, A.g_rowid.p, A.g_rowptr.p, A.g_col.p, A.g_val.p, x, y, aux,
                         partials, sc, xcd_map());
    return;
  }
  if (v == 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv<MODE>), dim3(grid), dim3(BLOCK), 0, st, A.n, A.rowptr.p, A.col.p, A.val.p, x, y,
                       aux, partials, sc);
    return;
  }
  const bool nt = (v & 4) != 0, pipe = (v & 16) != 0;
  if (v & 32) {
    if (nt && pipe) PG_LAUNCH_C((k_spmv_cw<MODE, true, true>));
    else if (nt) PG_LAUNCH_C((k_spmv_cw<MODE, true, false>));
    else if (pipe) PG_LAUNCH_C((k_spmv_cw<MODE, false, true>));
    else PG_LAUNCH_C((k_spmv_cw<MODE, false, false>));
    return;
  }
  if (nt && pipe) PG_LAUNCH_C((k_spmv_c<MODE, true, true>));
  else if (nt) PG_LAUNCH_C((k_spmv_c<MODE, true, false>));
  else if (pipe) PG_LAUNCH_C((k_spmv_c<MODE, false, true>));
  else PG_LAUNCH_C((k_spmv_c<MODE, false, false>));
}

}  // namespace


namespace {

struct Slice {
  int r0, meta, base, aux;
  int key;   // first matrix row (sort key)
};

// stencil-slice image of A (see "stencil slices" above); rp = host copy of A.rowptr
void build_slices(CsrMatrix& A, const std::vector<int>& rp) {
  hipStream_t st = ctx().stream;
  const i64 n = A.n;
  static const int minrun = getenv("PG_SPMV_MINRUN") ? atoi(getenv("PG_SPMV_MINRUN")) : 12;
  std::vector<unsigned char> fl(n);
  {
    DevBuf<unsigned char> flags(n);
    hipLaunchKernelGGL(k_row_same, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, n, A.rowptr.p, A.col.p, A.val.p, flags.p);
    PG_HIP(hipGetLastError());
    flags.download(fl.data(), n);
  }
  std::vector<Slice> up;       // U and P slices
  std::vector<int> grows;      // irregular rows, ascending
  auto emit = [&](int type, i64 a, i64 b, int cnt) {
    for (i64 r = a; r < b; r += SL_MAXROWS) {
      const int rows = (int)std::min<i64>(SL_MAXROWS, b - r);
      up.push_back(Slice{(int)r, rows | (type << 8) | (cnt << 16), 0, 0, (int)r});
    }
    (type == SL_U ? A.rows_u : A.rows_p) += b - a;
    if (type == SL_P) A.nnz_p += (b - a) * cnt;
  };
  auto irregular = [&](i64 a, i64 b) {
    for (i64 r = a; r < b; ++r) grows.push_back((int)r);
  };
  A.rows_u = A.rows_p = A.rows_g = A.nnz_p = A.nnz_g = 0;
  i64 i = 0;
  while (i < n) {
    i64 j = i + 1;
    while (j < n && (fl[j] & 1)) ++j;                 // pattern run [i, j)
    const int cnt = rp[i + 1] - rp[i];
    if (j - i >= minrun && cnt >= 1 && cnt <= SL_MAXCNT) {
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
  // (a row flushed out of order above keeps `grows` ascending only within a run: restore the global order)
  std::sort(grows.begin(), grows.end());
  // stencil table: first row of every U / P slice, de-duplicated
  const i64 nup = (i64)up.size();
  std::vector<int> toff;
  std::vector<double> tval;
  if (nup > 0) {
    std::vector<int> first(nup);
    for (i64 q = 0; q < nup; ++q) first[q] = up[q].r0;
    DevBuf<int> dfirst(nup), doff(nup * SL_MAXCNT);
    DevBuf<double> dval(nup * SL_MAXCNT);
    dfirst.upload(first.data(), nup);
    hipLaunchKernelGGL(k_fetch_stencil, dim3(grid_for(nup, 256)), dim3(256), 0, st, nup, dfirst.p, A.rowptr.p, A.col.p, A.val.p,
                       doff.p, dval.p);
    PG_HIP(hipGetLastError());
    std::vector<int> hoff(nup * SL_MAXCNT);
    std::vector<double> hval(nup * SL_MAXCNT);
    doff.download(hoff.data(), nup * SL_MAXCNT);
    dval.download(hval.data(), nup * SL_MAXCNT);
    std::unordered_map<std::string, int> dict;
    i64 pbase = 0;
    for (i64 q = 0; q < nup; ++q) {
      const int type = (up[q].meta >> 8) & 3, rows = up[q].meta & 255, cnt = up[q].meta >> 16;
      double* v = hval.data() + q * SL_MAXCNT;
      if (type == SL_P)
        for (int k2 = 0; k2 < SL_MAXCNT; ++k2) v[k2] = 0.0;
      std::string key(reinterpret_cast<const char*>(hoff.data() + q * SL_MAXCNT), sizeof(int) * SL_MAXCNT);
      key.append(reinterpret_cast<const char*>(v), sizeof(double) * SL_MAXCNT);
      auto it = dict.find(key);
      int id;
      if (it == dict.end()) {
        id = (int)dict.size();
        dict.emplace(std::move(key), id);
        toff.insert(toff.end(), hoff.begin() + q * SL_MAXCNT, hoff.begin() + (q + 1) * SL_MAXCNT);
        tval.insert(tval.end(), v, v + SL_MAXCNT);
      } else id = it->second;
      up[q].aux = id;
      if (type == SL_P) {
        PG_REQUIRE(pbase + (i64)rows * cnt < (i64)2147483647, "P-slice value array exceeds int32 indexing");
        up[q].base = (int)pbase;
        pbase += (i64)rows * cnt;
      }
    }
  }
  A.nstencils = (i64)toff.size() / SL_MAXCNT;
  A.stab_off.alloc(toff.empty() ? SL_MAXCNT : (i64)toff.size());
  A.stab_val.alloc(tval.empty() ? SL_MAXCNT : (i64)tval.size());
  if (!toff.empty()) {
    A.stab_off.upload(toff.data(), (i64)toff.size());
    A.stab_val.upload(tval.data(), (i64)tval.size());
  }
  A.pval.alloc(A.nnz_p + 8);
  // packed CSR of the irregular rows + its chunks
  const i64 ng = (i64)grows.size();
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
      while (e < ng && e - q < 64 && grp[e + 1] - grp[q] <= SPMV_CHUNK_ENTRIES) ++e;
      all.push_back(Slice{(int)q, (int)(e - q) | (SL_G << 8), grp[q], grp[e], grows[q]});
      q = e;
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const Slice& a, const Slice& b) { return a.key < b.key; });
  A.nslices = (i64)all.size();
  std::vector<int> sd(4 * (A.nslices + 1), 0);
  for (i64 q = 0; q < A.nslices; ++q) {
    sd[4 * q] = all[q].r0; sd[4 * q + 1] = all[q].meta; sd[4 * q + 2] = all[q].base; sd[4 * q + 3] = all[q].aux;
  }
  A.sdesc.alloc((i64)sd.size());
  A.sdesc.upload(sd.data(), (i64)sd.size());
  if (A.nnz_p > 0) {
    hipLaunchKernelGGL(k_pval_fill, dim3((unsigned)std::min<i64>(A.nslices, 65535)), dim3(256), 0, st, A.nslices, A.sdesc.p,
                       A.rowptr.p, A.val.p, A.pval.p);
    PG_HIP(hipGetLastError());
  }
  PG_HIP(hipStreamSynchronize(st));
  A.spmv_bytes = 16 * A.nslices + 8 * A.nnz_p + 12 * A.nnz_g + 8 * ng + 16 * n;
  if (getenv("PG_DEBUG"))
    fprintf(stderr, "[pg_spmv] slices %lld (stencils %lld): rows U %lld P %lld G %lld of %lld; nnz P %lld G %lld of %lld; bytes/launch %lld (CSR %lld)\n",
            (long long)A.nslices, (long long)A.nstencils, (long long)A.rows_u, (long long)A.rows_p, (long long)A.rows_g,
            (long long)n, (long long)A.nnz_p, (long long)A.nnz_g, (long long)A.nnz, (long long)A.spmv_bytes,
            (long long)(12 * A.nnz + 20 * n));
}

}  // namespace

// rows -> chunks of <= 64 rows and <= SPMV_CHUNK_ENTRIES entries (greedy; once per matrix, host side)
void build_spmv_chunks(CsrMatrix& A) {
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
  build_slices(A, rp);
}

int spmv_default_grid(i64 n) {
  static const int per_cu = getenv("PG_SPMV_BLOCKS_PER_CU") ? atoi(getenv("PG_SPMV_BLOCKS_PER_CU")) : 6;
  static int cus = 0;
  if (cus == 0) {
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDeviceProperties(&prop, ctx().device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  return grid_for(n, BLOCK, cus * per_cu);
}

void launch_spmv(int mode, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
                 const double* sc, int grid, hipStream_t st) {
  if (A.n == 0) return;
  const int v = variant();
  if (mode == 0) launch_mode<0>(v, A, x, y, aux, partials, sc, grid, st);
  else if (mode == 1) launch_mode<1>(v, A, x, y, aux, partials, sc, grid, st);
  else launch_mode<2>(v, A, x, y, aux, partials, sc, grid, st);
}

void launch_spmv_variant(int v, const CsrMatrix& A, const double* x, double* y, hipStream_t st) {
  if (A.n == 0) return;
  launch_mode<0>(v, A, x, y, nullptr, nullptr, nullptr, spmv_default_grid(A.n), st);
}

}  // namespace pg
