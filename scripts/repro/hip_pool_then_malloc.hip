// Reproducer for DESIGN.md "stream-ordered pool": does a plain hipMalloc hand out memory that a hipFreeAsync has released
// to the default pool in STREAM order only -- i.e. while the kernel that still uses the block is running?
//   hipcc --offload-arch=gfx950 -o /tmp/pool_repro scripts/repro/hip_pool_then_malloc.hip && /tmp/pool_repro
// Prints, per trial, the two address ranges and whether they overlap while the first block's last kernel has not finished.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void slow_fill(double* p, size_t n, long long spin) {
  for (size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}                     // the block stays in use for ~ spin / 100 MHz seconds
}
__global__ void count_not_one(const double* p, size_t n, unsigned long long* bad) {
  unsigned long long c = 0;
  for (size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != 1.0;
  if (c) atomicAdd(bad, c);
}
int main() {
  hipStream_t s;
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipMemPool_t pool;
  CK(hipDeviceGetDefaultMemPool(&pool, 0));
  uint64_t keep = ~0ull;
  CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
  const size_t bytes = (size_t)96 << 20, n = bytes / 8;
  unsigned long long* bad;
  CK(hipMalloc(&bad, 8));
  int overlaps = 0, corrupt = 0;
  for (int trial = 0; trial < 8; ++trial) {
    double *a = nullptr, *keepalive = nullptr, *b = nullptr;
    CK(hipMallocAsync((void**)&a, bytes, s));
    CK(hipMallocAsync((void**)&keepalive, bytes, s));
    CK(hipMemsetAsync(bad, 0, 8, s));
    hipLaunchKernelGGL(slow_fill, dim3(256), dim3(256), 0, s, a, n, 5000000LL);   // writes, then keeps running ~50 ms
    hipLaunchKernelGGL(count_not_one, dim3(256), dim3(256), 0, s, (const double*)a, n, bad);
    CK(hipFreeAsync(a, s));                       // free point: after count_not_one, in stream order
    CK(hipMalloc((void**)&b, bytes));             // NO synchronisation: the kernels above are still running
    const bool ov = (char*)b < (char*)a + bytes && (char*)a < (char*)b + bytes;
    CK(hipMemset(b, 0, bytes));                   // the new owner writes its block (null stream, returns when done)
    unsigned long long h = 0;
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("trial %d: pool block %p, hipMalloc block %p: %s; first owner saw %llu foreign values\n", trial, (void*)a, (void*)b,
           ov ? "OVERLAP while the first owner's kernels were queued" : "disjoint", h);
    overlaps += ov; corrupt += h != 0;
    CK(hipFree(b));
    CK(hipFreeAsync(keepalive, s));
    CK(hipStreamSynchronize(s));
  }
  printf("%d of 8 trials overlapped, %d corrupted the first owner's data\n", overlaps, corrupt);
  return 0;
}
