// Stress of the stream-ordered allocator alone (no library code): blocks of 1 .. 280 MB come and go on one stream, every
// block is filled with its own number when it is allocated and checked just before it is freed.  A block that does not hold
// its own number any more was given to two owners (or written by a third party) by the allocator.
//   hipcc --offload-arch=gfx950 -o /tmp/pool_stress scripts/repro/hip_pool_stress.hip && /tmp/pool_stress [max_live_GB=12] [plain_every=0] [policy=0] [small=0]
// plain_every = k > 0: every k-th allocation is a plain hipMalloc / hipFree instead (the mixed use of round 2).
// policy bits: 1 hipMemPoolReuseAllowOpportunistic off, 2 hipMemPoolReuseAllowInternalDependencies off,
// 4 hipMemPoolReuseFollowEventDependencies off, 8 release threshold left at its default (0).  small = 1: blocks <= 32 MB only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned long long* p, size_t n, unsigned long long id) {
  for (size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = id * 1000003ull + i;
}
// bad[0]: words that do not hold their owner's value; bad[1]: ... and are ZERO (a late clear of the pages); bad[2]: ... and hold
// the pattern of another block (two owners)
__global__ void check(const unsigned long long* p, size_t n, unsigned long long id, unsigned long long* bad) {
  unsigned long long c = 0, z = 0, o = 0;
  for (size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long v = p[i];
    if (v != id * 1000003ull + i) { ++c; if (v == 0) ++z; else ++o; }
  }
  if (c) { atomicAdd(bad, c); atomicAdd(bad + 1, z); atomicAdd(bad + 2, o); }
}
struct Blk { unsigned long long* p; size_t n; unsigned long long id; bool plain; };
int main(int argc, char** argv) {
  const double max_gb = argc > 1 ? atof(argv[1]) : 12.0;
  const int plain_every = argc > 2 ? atoi(argv[2]) : 0;
  const int policy = argc > 3 ? atoi(argv[3]) : 0;
  const int small = argc > 4 ? atoi(argv[4]) : 0;
  hipStream_t s;
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipMemPool_t pool;
  CK(hipDeviceGetDefaultMemPool(&pool, 0));
  uint64_t keep = ~0ull;
  if (!(policy & 8)) CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
  int off = 0;
  if (policy & 1) CK(hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowOpportunistic, &off));
  if (policy & 2) CK(hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowInternalDependencies, &off));
  if (policy & 4) CK(hipMemPoolSetAttribute(pool, hipMemPoolReuseFollowEventDependencies, &off));
  int overlaps = 0;
  unsigned long long* bad;
  CK(hipMalloc(&bad, 48));     // [0..2]: pooled blocks, [3..5]: plain blocks
  CK(hipMemset(bad, 0, 48));
  const int wa = argc > 6 ? atoi(argv[6]) : 0;
  const double reserve_gb = argc > 5 ? atof(argv[5]) : 0.0;   // > 0: grow the pool ONCE up front and wait for it
  if (reserve_gb > 0) {
    void* r = nullptr;
    CK(hipMallocAsync(&r, (size_t)(reserve_gb * 1e9), s));
    CK(hipMemsetAsync(r, 0, (size_t)(reserve_gb * 1e9), s));
    CK(hipStreamSynchronize(s));
    CK(hipFreeAsync(r, s));
    CK(hipStreamSynchronize(s));
    CK(hipDeviceSynchronize());
  }
  std::vector<Blk> live;
  size_t live_bytes = 0;
  unsigned long long next_id = 1, rng = 88172645463325252ull;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  int allocs = 0, frees = 0;
  for (int it = 0; it < 3000; ++it) {
    const bool grow = live.empty() || (live_bytes < (size_t)(max_gb * 1e9) && rnd() % 100 < 55);
    if (grow) {
      const size_t sizes[6] = {(size_t)134283336, (size_t)268566608, (size_t)67141704, (size_t)1 << 20, (size_t)33570824, (size_t)4096};
      const size_t small_sizes[6] = {(size_t)33570824, (size_t)16785416, (size_t)8 << 20, (size_t)1 << 20, (size_t)65536, (size_t)4096};
      const size_t bytes = small ? small_sizes[rnd() % 6] : sizes[rnd() % 6];
      Blk b{nullptr, bytes / 8, next_id++, plain_every > 0 && allocs % plain_every == plain_every - 1};
      if (b.plain) {
        // workarounds tried around the plain allocation (6th argument, bits): 1 stream sync before, 2 device sync before,
        // 4 device sync after, 8 device sync before the matching hipFree
        if (wa & 1) CK(hipStreamSynchronize(s));
        if (wa & 2) CK(hipDeviceSynchronize());
        CK(hipMalloc((void**)&b.p, bytes));
        if (wa & 4) CK(hipDeviceSynchronize());
      } else {
        CK(hipMallocAsync((void**)&b.p, bytes, s));
      }
      for (auto& o : live)      // two LIVE blocks must never share an address
        if ((char*)o.p < (char*)b.p + bytes && (char*)b.p < (char*)o.p + o.n * 8) ++overlaps;
      hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, b.p, b.n, b.id);
      live.push_back(b);
      live_bytes += bytes;
      ++allocs;
    } else {
      const size_t k = rnd() % live.size();
      Blk b = live[k];
      live[k] = live.back();
      live.pop_back();
      hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, s, (const unsigned long long*)b.p, b.n, b.id, bad + (b.plain ? 3 : 0));
      if (b.plain) { if (wa & 8) CK(hipDeviceSynchronize()); CK(hipFree(b.p)); } else CK(hipFreeAsync(b.p, s));
      live_bytes -= b.n * 8;
      ++frees;
    }
  }
  for (auto& b : live) hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, s, (const unsigned long long*)b.p, b.n, b.id, bad + (b.plain ? 3 : 0));
  CK(hipStreamSynchronize(s));
  unsigned long long hh[6] = {0, 0, 0, 0, 0, 0};
  CK(hipMemcpy(hh, bad, 48, hipMemcpyDeviceToHost));
  const unsigned long long h = hh[0] + hh[3];
  printf("pooled blocks: %llu wrong words (%llu zero, %llu foreign); plain blocks: %llu wrong words (%llu zero, %llu foreign)\n", hh[0], hh[1],
         hh[2], hh[3], hh[4], hh[5]);
  hh[1] += hh[4]; hh[2] += hh[5];
  printf("policy %d small %d overlapping live blocks %d | max live %.1f GB, plain_every %d: %d allocations, %d frees, %zu blocks live at the end: %llu words did not hold their owner's value (%llu of them zero, %llu foreign), reserve %.1f GB, workaround %d\n",
         policy, small, overlaps, max_gb, plain_every, allocs, frees, live.size(), h, hh[1], hh[2], reserve_gb, wa);
  return h != 0;
}
