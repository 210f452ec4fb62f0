// pg_spmv.hip -- the dominant kernel of the path: y = A x for the reduced cut-cell system in CSR
// (fp64 values, int32 column indices, int32 row pointers; rows in box order, 1 / 2N+1 / up to 2(2N+1) entries).
// Replaces the single-threaded CSC mul! inside IterativeSolvers (src/solver.jl:178-181).
//
// Roofline: HBM.  Algorithmic bytes per launch = 12 nnz + 20 n; x (8 n bytes, 82 MB at 512^3)
// is gathered out of L2 / Infinity Cache.
//
// Variants (PG_SPMV_VARIANT, default = best measured):
//   1  block of 256 rows staged through LDS, two block barriers per chunk            (first version)
//   2  one wave owns 64 rows: private LDS slice, no block barrier, unrolled staging loads and gathers
#include <cstdlib>

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

// ---- variant 2: one WAVE owns 64 consecutive rows -----------------------------------------------------------
// No block barrier: each of the 4 waves of a block has a private LDS slice, so a wave stalls only on its own
// loads.  Per wave-iteration: (1) all val/col loads of the 64-row chunk are issued back to back (up to 10 per
// lane; fully coalesced 512-B / 256-B wave accesses), (2) written to the wave's LDS slice, (3) lane l walks row
// l: columns out of LDS, all x gathers issued together, then the FMAs.  LDS slice = 640 entries (7.5 KB per
// wave -> 20 waves per CU); the rare chunk with more entries (runs of 14-entry cut rows) reads global memory
// directly.
constexpr int WROWS = 64;
constexpr int WCAP = 640;
constexpr int WITER = WCAP / 64;   // staging loads per lane
constexpr int WUNROLL = 8;         // gathers issued together per row

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_spmv_w(i64 n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const double* __restrict__ val, const double* __restrict__ x,
                                                  double* __restrict__ y, const double* __restrict__ aux,
                                                  double* __restrict__ partials, const double* __restrict__ sc) {
  __shared__ double s_val[BLOCK / 64][WCAP];
  __shared__ int s_col[BLOCK / 64][WCAP];
  __shared__ double s_red[BLOCK / 64];
  if (sc && sc[S_DONE] != 0.0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* __restrict__ sv = s_val[wave];
  int* __restrict__ scl = s_col[wave];
  double acc0 = 0.0, acc1 = 0.0;
  const i64 nchunks = (n + WROWS - 1) / WROWS;
  const i64 wstride = (i64)gridDim.x * (BLOCK / 64);
  for (i64 chunk = (i64)blockIdx.x * (BLOCK / 64) + wave; chunk < nchunks; chunk += wstride) {
    const i64 r0 = chunk * WROWS;
    const i64 r1 = r0 + WROWS < n ? r0 + WROWS : n;
    const i64 r = r0 + lane;
    const int base = rowptr[r0];
    const int cnt = rowptr[r1] - base;
    int a = 0, b = 0;
    if (r < r1) {
      a = rowptr[r] - base;
      b = rowptr[r + 1] - base;
    }
    double sum = 0.0;
    if (cnt <= WCAP) {
      double tv[WITER];
      int tc[WITER];
#pragma unroll
      for (int j = 0; j < WITER; ++j) {
        const int k = lane + 64 * j;
        if (k < cnt) {
          tv[j] = val[base + k];
          tc[j] = col[base + k];
        }
      }
#pragma unroll
      for (int j = 0; j < WITER; ++j) {
        const int k = lane + 64 * j;
        if (k < cnt) {
          sv[k] = tv[j];
          scl[k] = tc[j];
        }
      }
      __builtin_amdgcn_wave_barrier();
      double xv[WUNROLL];
#pragma unroll
      for (int j = 0; j < WUNROLL; ++j)
        if (a + j < b) xv[j] = x[scl[a + j]];
#pragma unroll
      for (int j = 0; j < WUNROLL; ++j)
        if (a + j < b) sum += sv[a + j] * xv[j];
      for (int k = a + WUNROLL; k < b; ++k) sum += sv[k] * x[scl[k]];
      __builtin_amdgcn_wave_barrier();
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

int variant() {
  static const int v = getenv("PG_SPMV_VARIANT") ? atoi(getenv("PG_SPMV_VARIANT")) : 2;
  return v;
}

template <int MODE>
void launch_mode(const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials, const double* sc,
                 int grid, hipStream_t st) {
  if (variant() == 1)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv<MODE>), dim3(grid), dim3(BLOCK), 0, st, A.n, A.rowptr.p, A.col.p, A.val.p, x, y,
                       aux, partials, sc);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_w<MODE>), dim3(grid), dim3(BLOCK), 0, st, A.n, A.rowptr.p, A.col.p, A.val.p, x,
                       y, aux, partials, sc);
}

}  // namespace

int spmv_default_grid(i64 n) { return grid_for(n, BLOCK, 256 * 8); }

void launch_spmv(int mode, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
                 const double* sc, int grid, hipStream_t st) {
  if (A.n == 0) return;
  if (mode == 0) launch_mode<0>(A, x, y, aux, partials, sc, grid, st);
  else if (mode == 1) launch_mode<1>(A, x, y, aux, partials, sc, grid, st);
  else launch_mode<2>(A, x, y, aux, partials, sc, grid, st);
}

}  // namespace pg
