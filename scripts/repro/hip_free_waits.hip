// Does hipFree wait for work queued on a NON-BLOCKING stream that still uses the block?  (DESIGN.md "stream-ordered pool")
//   hipcc --offload-arch=gfx950 -o /tmp/free_waits scripts/repro/hip_free_waits.hip && /tmp/free_waits
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_then_write(double* p, long long spin) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  p[threadIdx.x] = 1.0;
}
int main() {
  hipStream_t s;
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int trial = 0; trial < 3; ++trial) {
    double* a = nullptr;
    CK(hipMalloc((void**)&a, (size_t)64 << 20));
    hipLaunchKernelGGL(spin_then_write, dim3(1), dim3(64), 0, s, a, 20000000LL);   // ~200 ms at 100 MHz
    const auto t0 = std::chrono::steady_clock::now();
    CK(hipFree(a));
    const auto t1 = std::chrono::steady_clock::now();
    const hipError_t q = hipStreamQuery(s);
    printf("trial %d: hipFree returned after %.1f ms; the kernel that uses the block was %s\n", trial,
           std::chrono::duration<double, std::milli>(t1 - t0).count(), q == hipSuccess ? "finished" : "STILL RUNNING");
    CK(hipStreamSynchronize(s));
  }
  return 0;
}
