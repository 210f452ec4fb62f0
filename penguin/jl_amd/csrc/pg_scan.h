// pg_scan.h -- device-wide exclusive scan (int32 result) used by the active-set compaction (K10) and the
// CSR row pointers (K7).  Wave64 shuffles inside a wave, LDS across the 4 waves of a 256-thread block,
// recursion over block sums across blocks.  Setup-time only; not on the per-step path.
#pragma once
#include "pg_common.h"

namespace pg {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// exclusive scan of one int per thread over the block; *total = block sum (valid in every thread)
__device__ inline int block_exclusive_scan(int v, int* total) {
  __shared__ int wave_sums[SCAN_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    const int s = wave_sums[w];
    if (w < wave) wave_off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wave_off + incl - v;
}

template <class T>
__global__ void k_scan_reduce(const T* __restrict__ in, i64 n, int* __restrict__ bsum) {
  const i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) s += (int)in[base + k];
  int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

template <class T>
__global__ void k_scan_apply(const T* __restrict__ in, i64 n, const int* __restrict__ boff, int* __restrict__ out,
                             int* __restrict__ total_out) {
  const i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? (int)in[base + k] : 0;
    s += v[k];
  }
  int tot;
  int run = block_exclusive_scan(s, &tot) + (boff ? boff[blockIdx.x] : 0);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) *total_out = run;
}

// out[i] = sum_{k<i} in[k]; *d_total (device int, optional) = sum of everything.
template <class T>
void scan_exclusive(const T* in, int* out, i64 n, int* d_total, hipStream_t st) {
  if (n <= 0) {
    if (d_total) PG_HIP(hipMemsetAsync(d_total, 0, sizeof(int), st));
    return;
  }
  const i64 nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<T>), dim3(1), dim3(SCAN_BLOCK), 0, st, in, n,
                       (const int*)nullptr, out, d_total);
    PG_HIP(hipGetLastError());
    return;
  }
  DevBuf<int> bsum(nb), boff(nb);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<T>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, n, bsum.p);
  PG_HIP(hipGetLastError());
  scan_exclusive<int>(bsum.p, boff.p, nb, nullptr, st);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<T>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, n,
                     (const int*)boff.p, out, d_total);
  PG_HIP(hipGetLastError());
  PG_HIP(hipStreamSynchronize(st));  // bsum/boff are freed on return
}

}  // namespace pg
