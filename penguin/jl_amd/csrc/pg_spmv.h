// pg_spmv.h -- shared between the Krylov driver and the SpMV kernels: device scalar slots, block reductions.
#pragma once
#include "pg_system.h"

namespace pg {

// device scalar block of one Krylov solve (KrylovWork::sc)
enum { S_RHO = 0, S_RHO_OLD, S_ALPHA, S_OMEGA, S_BETA, S_RR, S_BB, S_TOL2, S_DONE, S_ITERS, S_RELTOL2, S_ABSTOL2,
       S_RESTART, S_RHAT2, S_FORCE, S_PENDING3,
       S_RED0, S_RED1, S_RED2, S_RED3, S_RED4, S_COUNT };

constexpr int BLOCK = 256;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// sum over the block; result valid in thread 0
__device__ inline double block_sum(double v, double* sh /*BLOCK/64*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += sh[w];
  }
  return s;
}

// y = A x on rows [0, A.n).  mode 0: plain; 1: partials[0..grid) = aux . y; 2: partials[0..grid) = y . x and
// partials[grid..2grid) = y . y; 3: mode 2 plus partials[4grid..5grid) = aux . y.  `sc` (may be NULL): kernels return immediately when sc[S_DONE] != 0.
// `grid` must be the value used to size `partials` (KrylovWork::grid) for modes 1/2.
void launch_spmv(int mode, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
                 const double* sc, int grid, hipStream_t st);
// plain y = A x with an explicit kernel variant (PG_SPMV_VARIANT numbering): kernel-vs-kernel parity checks
void launch_spmv_variant(int variant, const CsrMatrix& A, const double* x, double* y, hipStream_t st);
int spmv_default_grid(i64 n);

}  // namespace pg
