// pg_probe.hip -- diagnostics only: read-bandwidth probes used to place the SpMV kernel against what the
// memory system delivers for ITS access shapes (8-byte and 4-byte per lane streams), not just the float4 copy
// figure of the micro-architecture guide.  Not on the product path.
#include "pg_common.h"

using namespace pg;

namespace {
template <class T, bool NT>
__global__ __launch_bounds__(256) void k_probe_read(const T* __restrict__ p, i64 n, double* out) {
  double acc = 0.0;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
  // 4 independent loads in flight per lane
  for (; i + 3 * stride < n; i += 4 * stride) {
    T a, b, c, d;
    if (NT) {
      a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride);
      c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride);
    } else {
      a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride];
    }
    acc += (double)a + (double)b + (double)c + (double)d;
  }
  for (; i < n; i += stride) acc += (double)p[i];
  if (acc == 1.2345e-300) out[0] = acc;   // keep the loads alive
}
}  // namespace

extern "C" int32_t pg_debug_read_probe(int64_t bytes, int32_t elem_bytes, int32_t nt, int32_t blocks, int32_t reps,
                                       double* gbs) {
  PG_API_BEGIN
  require_init();
  hipStream_t st = ctx().stream;
  DevBuf<double> buf(bytes / 8), out(1);
  buf.zero();
  hipEvent_t e0, e1;
  PG_HIP(hipEventCreate(&e0));
  PG_HIP(hipEventCreate(&e1));
  auto launch = [&]() {
    if (elem_bytes == 8) {
      if (nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_read<double, true>), dim3(blocks), dim3(256), 0, st, buf.p, bytes / 8, out.p);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_read<double, false>), dim3(blocks), dim3(256), 0, st, buf.p, bytes / 8, out.p);
    } else {
      const int* ip = reinterpret_cast<const int*>(buf.p);
      if (nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_read<int, true>), dim3(blocks), dim3(256), 0, st, ip, bytes / 4, out.p);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_read<int, false>), dim3(blocks), dim3(256), 0, st, ip, bytes / 4, out.p);
    }
  };
  for (int i = 0; i < 2; ++i) launch();
  PG_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  PG_HIP(hipEventRecord(e1, st));
  PG_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PG_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbs = (double)bytes * reps / (ms * 1e-3) / 1e9;
  PG_API_END
}
