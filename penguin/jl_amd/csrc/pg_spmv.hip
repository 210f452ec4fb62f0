// pg_spmv.hip -- the dominant kernel of the path: y = A x for the reduced cut-cell system in CSR
// (fp64 values, int32 column indices, int32 row pointers; rows in box order, 1 / 2N+1 / up to 2(2N+1) entries).
// Replaces the single-threaded CSC mul! inside IterativeSolvers (src/solver.jl:178-181).
//
// Roofline: HBM.  Algorithmic bytes per launch = 12 nnz + 20 n; x (8 n bytes, 82 MB at 512^3)
// is gathered out of L2 / Infinity Cache.
//
// Design (CDNA4): the rows are cut once per matrix into CHUNKS of <= 64 consecutive rows and <= 512 entries
// (CsrMatrix::chunk_start, the row-block idea of CSR-Adaptive).  One WAVE owns one chunk per iteration.  The
// chunk's val/col entries are one contiguous range of the CSR arrays; the wave streams it with fully coalesced
// loads into its PRIVATE slice of LDS -- no block barrier anywhere, a wave stalls only on its own loads -- and
// then lane l walks row l out of LDS: column indices, all x gathers issued together, FMAs.  For a fixed
// stencil slot consecutive lanes gather consecutive x entries, so the gathers coalesce too.
//
// Everything inside an iteration is branch-free with a FIXED number of vector-memory operations (clamped
// indices + selects).  Two reasons, both seen in the ISA: (1) hipcc puts every predicated load in its own basic
// block behind `s_waitcnt vmcnt(0)`, which serialises the gathers; (2) vector-memory operations retire in
// order, so the software pipeline below needs a COUNTED wait -- `s_waitcnt vmcnt(N)` lets the gathers of chunk
// i complete while the N younger stream loads of chunk i+1 stay in flight -- and the compiler can only count
// when every path issues the same number of loads.
//
// The dots that follow an SpMV in BiCGStab / CG are fused into the epilogue (one partial per block, summed in
// fixed order by k_finalize: deterministic).
//
// Variants (PG_SPMV_VARIANT; default = best measured, see profiles/):
//   1   block of 256 rows staged through LDS, two block barriers per chunk     (first version, 2.0 TB/s)
//   2   chunked, wave-private LDS slices, branch-free                          (bit 1)
//   +4  non-temporal hint on the matrix stream (read once; keeps x in L2 / Infinity Cache)
//   +8  XCD-contiguous chunk ranges (blocks sharing an XCD sweep one eighth of the chunks)
//   +16 software pipeline inside the wave: the stream of chunk i+1 is issued behind the gathers of chunk i
#include <cstdlib>
#include <vector>

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
constexpr int WITER = SPMV_CHUNK_ENTRIES / 64;   // staging loads per lane (8)
constexpr int WUNR = 8;                          // gathers issued together per row (bulk rows: 2N+1 <= 7)

template <bool NT, class T>
__device__ inline T stream_load(const T* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

struct Chunk {
  int r0, nrows;       // wave-uniform (scalar loads)
  int base, cnt;       // wave-uniform: first entry / number of entries
  int ra, rb;          // this lane's row bounds, ABSOLUTE and raw: nothing consumes them in the issue phase
  double tv[WITER];
  int tc[WITER];
};

// issue every load of a chunk -- always 2 + 2*WITER vector loads, branch-free
template <bool NT>
__device__ inline void chunk_issue(Chunk& q, i64 ch, const int* __restrict__ chunk_start, const int* __restrict__ rowptr,
                                   const int* __restrict__ col, const double* __restrict__ val, int lane) {
  q.r0 = chunk_start[ch];                       // uniform address -> s_load (lgkmcnt, not vmcnt)
  const int r1 = chunk_start[ch + 1];
  q.nrows = r1 - q.r0;
  q.base = rowptr[q.r0];
  q.cnt = rowptr[r1] - q.base;
  const int r = q.r0 + lane;
  q.ra = stream_load<NT>(rowptr + (r < r1 ? r : r1));
  q.rb = stream_load<NT>(rowptr + (r + 1 < r1 ? r + 1 : r1));
  const int last = q.cnt > 0 ? q.cnt - 1 : 0;
#pragma unroll
  for (int j = 0; j < WITER; ++j) {
    const int k = lane + 64 * j;
    const int kk = k < last ? k : last;         // tail lanes re-read the last entry (one cache line)
    q.tv[j] = stream_load<NT>(val + q.base + kk);
    q.tc[j] = stream_load<NT>(col + q.base + kk);
  }
}

template <int MODE, bool NT, bool XCD, bool PIPE>
__global__ __launch_bounds__(BLOCK) void k_spmv_c(i64 n, i64 nchunks, const int* __restrict__ chunk_start,
                                                  const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const double* __restrict__ val, const double* __restrict__ x,
                                                  double* __restrict__ y, const double* __restrict__ aux,
                                                  double* __restrict__ partials, const double* __restrict__ sc) {
  __shared__ double s_val[BLOCK / 64][SPMV_CHUNK_ENTRIES];
  __shared__ int s_col[BLOCK / 64][SPMV_CHUNK_ENTRIES];
  __shared__ double s_red[BLOCK / 64];
  if (sc && sc[S_DONE] != 0.0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* __restrict__ sv = s_val[wave];
  int* __restrict__ scl = s_col[wave];
  double acc0 = 0.0, acc1 = 0.0;
  i64 c_hi = nchunks, first = (i64)blockIdx.x * (BLOCK / 64) + wave, wstride = (i64)gridDim.x * (BLOCK / 64);
  if (XCD && (gridDim.x & 7) == 0) {
    // blocks b and b+8 share an XCD (observed round-robin placement: speed only, never correctness)
    const int xcd = blockIdx.x & 7, g = blockIdx.x >> 3, gpx = gridDim.x >> 3;
    const i64 c_lo = nchunks * xcd / 8;
    c_hi = nchunks * (xcd + 1) / 8;
    first = c_lo + (i64)g * (BLOCK / 64) + wave;
    wstride = (i64)gpx * (BLOCK / 64);
  }
  Chunk q;
  if (PIPE && first < c_hi) chunk_issue<NT>(q, first, chunk_start, rowptr, col, val, lane);
  for (i64 chunk = first; chunk < c_hi; chunk += wstride) {
    if (!PIPE) chunk_issue<NT>(q, chunk, chunk_start, rowptr, col, val, lane);
    const int r = q.r0 + lane;
    const bool live = lane < q.nrows;
    const int a = q.ra - q.base, b = q.rb - q.base;
    // stream -> LDS slice (fixed 2*WITER stores; slots past cnt hold copies of the last entry)
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
      xv[j] = x[scl[p]];
      vv[j] = a + j < b ? sv[p] : 0.0;
    }
    if (PIPE) {
      // behind the gathers: these 2 + 2*WITER loads stay in flight while the gathers retire (counted vmcnt)
      const i64 next = chunk + wstride < c_hi ? chunk + wstride : chunk;
      chunk_issue<NT>(q, next, chunk_start, rowptr, col, val, lane);
    }
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < WUNR; ++j) sum += vv[j] * xv[j];
    if (__builtin_expect(b - a > WUNR, 0))
      for (int k = a + WUNR; k < b; ++k) sum += sv[k] * x[scl[k]];            // cut-cell rows (> 8 entries)
    __builtin_amdgcn_wave_barrier();
    if (live) {
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

int variant() {
  static const int v = getenv("PG_SPMV_VARIANT") ? atoi(getenv("PG_SPMV_VARIANT")) : 22;   // chunked + NT + pipelined
  return v;
}

#define PG_LAUNCH_C(KERNEL)                                                                                     \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL), dim3(grid), dim3(BLOCK), 0, st, A.n, A.nchunks, A.chunk_start.p, \
                     A.rowptr.p, A.col.p, A.val.p, x, y, aux, partials, sc)

template <int MODE, bool NT, bool XCD>
void launch_c(bool pipe, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
              const double* sc, int grid, hipStream_t st) {
  if (pipe) PG_LAUNCH_C((k_spmv_c<MODE, NT, XCD, true>));
  else PG_LAUNCH_C((k_spmv_c<MODE, NT, XCD, false>));
}

template <int MODE>
void launch_mode(const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials, const double* sc,
                 int grid, hipStream_t st) {
  const int v = variant();
  if (v == 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv<MODE>), dim3(grid), dim3(BLOCK), 0, st, A.n, A.rowptr.p, A.col.p, A.val.p, x, y,
                       aux, partials, sc);
    return;
  }
  const bool nt = (v & 4) != 0, xcd = (v & 8) != 0, pipe = (v & 16) != 0;
  if (nt && xcd) launch_c<MODE, true, true>(pipe, A, x, y, aux, partials, sc, grid, st);
  else if (nt) launch_c<MODE, true, false>(pipe, A, x, y, aux, partials, sc, grid, st);
  else if (xcd) launch_c<MODE, false, true>(pipe, A, x, y, aux, partials, sc, grid, st);
  else launch_c<MODE, false, false>(pipe, A, x, y, aux, partials, sc, grid, st);
}

}  // namespace

// rows -> chunks of <= 64 rows and <= SPMV_CHUNK_ENTRIES entries (greedy; once per matrix, host side)
void build_spmv_chunks(CsrMatrix& A) {
  std::vector<int> rp(A.n + 1);
  A.rowptr.download(rp.data(), A.n + 1);
  std::vector<int> cs;
  cs.reserve(A.n / 60 + 16);
  i64 r = 0;
  cs.push_back(0);
  while (r < A.n) {
    i64 e = r + 1;   // a chunk holds at least one row (rows never exceed the slice: <= 4(2N+1) entries)
    PG_REQUIRE(rp[e] - rp[r] <= SPMV_CHUNK_ENTRIES, "CSR row longer than an SpMV chunk");
    while (e < A.n && e - r < 64 && rp[e + 1] - rp[r] <= SPMV_CHUNK_ENTRIES) ++e;
    cs.push_back((int)e);
    r = e;
  }
  A.nchunks = (i64)cs.size() - 1;
  A.chunk_start.alloc((i64)cs.size());
  A.chunk_start.upload(cs.data(), (i64)cs.size());
}

int spmv_default_grid(i64 n) {
  static const int per_cu = getenv("PG_SPMV_BLOCKS_PER_CU") ? atoi(getenv("PG_SPMV_BLOCKS_PER_CU")) : 5;
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
  if (mode == 0) launch_mode<0>(A, x, y, aux, partials, sc, grid, st);
  else if (mode == 1) launch_mode<1>(A, x, y, aux, partials, sc, grid, st);
  else launch_mode<2>(A, x, y, aux, partials, sc, grid, st);
}

}  // namespace pg
