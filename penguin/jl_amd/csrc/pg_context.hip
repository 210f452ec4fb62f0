// pg_context.hip -- device / stream / RCCL communicator lifetime, error channel, Mesh.
#include "pg_common.h"

#include <cxxabi.h>
#include <execinfo.h>

#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include "pg_host_algos.h"

#include <algorithm>
#include <array>
#include <memory>
#include <set>

namespace pg {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

Context& ctx() {
  static Context c;
  Context* t = thread_context();
  return t ? *t : c;
}

const Config& config() {
  static const Config c = [] {
    Config k;
    auto geti = [](const char* name, long long dflt) { const char* v = getenv(name); return v ? atoll(v) : dflt; };
    auto getd = [](const char* name, double dflt) { const char* v = getenv(name); return v ? atof(v) : dflt; };
    k.debug = getenv("PG_DEBUG") != nullptr;
    k.alloc_poison = (int)geti("PG_ALLOC_POISON", 0);
    k.async_alloc = (int)geti("PG_ASYNC_ALLOC", 0);
    k.alloc_guard = (int)geti("PG_ALLOC_GUARD", 0);
    k.pool_limit_mb = geti("PG_POOL_LIMIT_MB", 0);
    k.poly = geti("PG_POLY", 1) != 0;
    k.poly_degree = (int)geti("PG_POLY_DEGREE", 6);
    k.poly_adapt = geti("PG_POLY_ADAPT", getenv("PG_POLY_DEGREE") ? 0 : 1) != 0;
    k.poly_xspace = geti("PG_POLY_XSPACE", 1) != 0;
    k.half_test = (int)geti("PG_HALF_TEST", -1);
    k.half_batch = geti("PG_HALF_BATCH", 1) != 0;
    k.krylov_nt = geti("PG_KRYLOV_NT", 1) != 0;
    k.fuse_half_update = geti("PG_FUSE_HALF", 1) != 0;
    k.poly_margin = getd("PG_POLY_MARGIN", 1.0);
    k.poly_slack = getd("PG_POLY_SLACK", -1.0);
    k.poly_hist = (int)std::max<long long>(1, std::min<long long>(3, geti("PG_POLY_HIST", 3)));
    k.poly_maxdeg = (int)geti("PG_POLY_MAXDEG", 0);
    k.poly_mindeg = (int)std::max<long long>(2, std::min<long long>(8, geti("PG_POLY_MINDEG", 4)));
    k.poly_trend = geti("PG_POLY_TREND", 1) != 0;
    k.recovery_horner = geti("PG_RECOVERY_HORNER", 1) != 0;
    k.profile_sample = (int)std::max<long long>(1, geti("PG_PROFILE_SAMPLE", 3));
    k.gamma_elim = geti("PG_GAMMA_ELIM", 1) != 0;
    k.diag_elim = geti("PG_DIAG_ELIM", 1) != 0;
    k.diag_elim_frac = getd("PG_DIAG_ELIM_FRAC", 0.0005);
    k.spmv_variant = (int)geti("PG_SPMV_VARIANT", 70);
    k.spmv_xcd = (int)geti("PG_SPMV_XCD", 1);
    k.spmv_strip = (int)geti("PG_SPMV_STRIP", 16);
    k.spmv_march_k = (int)geti("PG_SPMV_MARCH_K", 0);
    k.spmv_minrun = (int)geti("PG_SPMV_MINRUN", 24);
    k.spmv_march = geti("PG_SPMV_MARCH", 1) != 0;
    k.spmv_tile_units = (int)geti("PG_SPMV_TILE_UNITS", 0);
    k.spmv_blocks_per_cu = (int)geti("PG_SPMV_BLOCKS_PER_CU", 0);
    k.halo_overlap = geti("PG_HALO_OVERLAP", 1) != 0;
    k.speculate_product = geti("PG_SPECULATE", 1) != 0;
    k.guess_n = (int)std::max<long long>(0, std::min<long long>(4, geti("PG_GUESS_STATES", 4)));
    k.guess_depth = (int)std::max<long long>(1, std::min<long long>(7, geti("PG_GUESS_DEPTH", 7)));
    k.guess_monitor = (int)std::max<long long>(1, geti("PG_GUESS_MONITOR", 256));
    k.guess_pass_cost = getd("PG_GUESS_PASS_COST", 0.34);
    k.guess_async = geti("PG_GUESS_ASYNC", 0) != 0;
    k.guess_always = geti("PG_GUESS_ALWAYS", 0) != 0;
    k.guess_defer = geti("PG_GUESS_DEFER", 1) != 0;
    k.guess_gain = getd("PG_GUESS_GAIN", 0.8);
    k.unit_order = (int)geti("PG_SPMV_UNIT_ORDER", 0);
    return k;
  }();
  return c;
}

void require_init() {
  if (!ctx().inited)
    throw Error("penguin_hip: not initialised (call pg_init / pg_init_distributed first)");
}

static void init_device(int device_id) {
  Context& c = ctx();
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    throw Error("penguin_hip: no HIP device visible -- this library has no CPU fallback");
  PG_REQUIRE(device_id >= 0 && device_id < ndev, "penguin_hip: device id out of range");
  PG_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  PG_HIP(hipGetDeviceProperties(&prop, device_id));
  c.device = device_id;
  c.device_name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
  if (!c.stream) PG_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  if (!c.comm_stream) PG_HIP(hipStreamCreateWithFlags(&c.comm_stream, hipStreamNonBlocking));
  c.inited = true;
}

}  // namespace pg

using namespace pg;

namespace pg {
namespace {
// ---- device memory ---------------------------------------------------------------------------------------------------
// hipMalloc / hipFree synchronise the device, and a caller that rebuilds capacities and systems every time step (the
// moving-body solver: ~60 buffers per slab) pays that ~120 times per slab.  Inside an AsyncAllocScope the library therefore
// keeps the blocks it frees in a CACHE of its own and hands them out again without any synchronisation.  That is safe for
// the same reason a stream-ordered allocator is: every kernel, memset and copy of the path is queued on ONE stream (the
// context's compute stream), so whatever the next owner queues runs after everything the previous owner queued.
//
// Why not the runtime's stream-ordered allocator (hipMallocAsync / hipFreeAsync), which round 2 used: on this runtime
// (ROCm 7.2, gfx950) live blocks of that pool lose their contents -- whole ranges read back as ZERO -- as soon as plain
// hipMalloc / hipFree calls are interleaved with it, whatever synchronisation is put around them, and a pool used alone
// occasionally hands one range to two owners in the first process after a box comes up.  scripts/repro/hip_pool_stress.hip
// shows both on the bare runtime (no library code; results in profiles/r03_allocator_findings.txt).  That, not the size of
// the pooled blocks, is what corrupted the 3072² / 4096² slabs of round 2 (its 64 MB limit made every run a mixed one); the
// library cannot avoid plain allocations altogether (constructors outside a scope, blocks of a GiB and more), so it does
// not use that allocator at all.
thread_local int t_async_scope = 0;
std::mutex g_alloc_mutex;
struct BlockCache {
  std::multimap<size_t, void*> free_blocks;        // size -> block, blocks that may be handed out again
  std::unordered_map<void*, size_t> owned;         // every block this cache has allocated (free or in use) -> its size
  size_t free_bytes = 0;
};
BlockCache g_cache;
constexpr size_t CACHE_TRIM_BYTES = (size_t)48 << 30;   // more than this idle: give everything idle back (one device sync)

size_t round_block(size_t bytes) {                      // few distinct sizes: 4 KiB steps, 2 MiB steps from 2 MiB on
  const size_t q = bytes >= ((size_t)2 << 20) ? ((size_t)2 << 20) : 4096;
  return (bytes + q - 1) / q * q;
}

bool cache_wanted(const Context& c, size_t bytes) {
  const int mode = config().async_alloc;             // 1 everywhere, 0 inside an AsyncAllocScope, -1 never
  if (mode < 0 || !c.inited || !c.stream || c.local) return false;   // (virtual ranks run one stream per host thread)
  if (!(t_async_scope > 0 || mode > 0)) return false;
  const long long lim = config().pool_limit_mb;
  return lim <= 0 || bytes < ((size_t)lim << 20);
}

// PG_ALLOC_POISON=1 (debugging): every block starts as 0xFF bytes -- NaN as a double, -1 as an int -- on the stream its first
// user is ordered on.  A buffer that is read before it is written then shows in the results instead of passing by the
// accident of fresh (zeroed) device pages; cached blocks and fresh ones alike.
void poison(void* p, size_t bytes) {
  if (!config().alloc_poison || !p || bytes == 0) return;
  Context& c = ctx();
  if (c.inited && c.stream) PG_HIP(hipMemsetAsync(p, 0xFF, bytes, c.stream));
  else PG_HIP(hipMemset(p, 0xFF, bytes));
}

void cache_trim_locked() {                              // idle blocks back to the runtime
  if (g_cache.free_blocks.empty()) return;
  (void)hipDeviceSynchronize();
  for (auto& kv : g_cache.free_blocks) {
    g_cache.owned.erase(kv.second);
    (void)hipFree(kv.second);
  }
  g_cache.free_blocks.clear();
  g_cache.free_bytes = 0;
}

void* dev_alloc_raw(size_t bytes) {
  void* p = nullptr;
  Context& c = ctx();
  if (cache_wanted(c, bytes)) {
    const size_t want = round_block(bytes);
    {
      std::lock_guard<std::mutex> lk(g_alloc_mutex);
      auto it = g_cache.free_blocks.lower_bound(want);
      // a block of the size class, or one at most an eighth larger (the per-slab sizes of a moving problem drift by a few
      // rows from slab to slab)
      if (it != g_cache.free_blocks.end() && it->first <= want + want / 8) {
        p = it->second;
        g_cache.free_bytes -= it->first;
        g_cache.free_blocks.erase(it);
      }
    }
    if (!p) {
      if (hipMalloc(&p, want) != hipSuccess || !p) {     // out of memory with idle blocks around: give them back, once
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lk(g_alloc_mutex); cache_trim_locked(); }
        PG_HIP(hipMalloc(&p, want));
      }
      std::lock_guard<std::mutex> lk(g_alloc_mutex);
      g_cache.owned[p] = want;
    }
    poison(p, bytes);
    return p;
  }
  PG_HIP(hipMalloc(&p, bytes));
  poison(p, bytes);
  return p;
}

void dev_free_raw(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_alloc_mutex);
    auto it = g_cache.owned.find(p);
    if (it != g_cache.owned.end()) {
      // back into the cache without synchronisation: the next owner's work is queued behind this owner's on the same stream
      g_cache.free_blocks.emplace(it->second, p);
      g_cache.free_bytes += it->second;
      if (g_cache.free_bytes > CACHE_TRIM_BYTES) cache_trim_locked();
      return;
    }
  }
  (void)hipFree(p);
}

// PG_ALLOC_GUARD=1 (debugging): every block sits between two 4 KiB guard bands of 0xA5 bytes that are checked when the block
// is freed -- an out-of-bounds WRITE shows with the size of the block it ran off, the side, the first damaged byte and the
// call stack that allocated the block.  (Plain hipMalloc blocks are padded to large pages, so an overrun is silent there.)
constexpr size_t GUARD = 4096;
struct GuardInfo { char* raw; size_t bytes; unsigned long long seq; std::string where; };
std::unordered_map<void*, GuardInfo> g_guarded;
unsigned long long g_alloc_seq = 0;
std::string call_stack() {
  void* frames[24];
  const int nf = backtrace(frames, 24);
  char** sym = backtrace_symbols(frames, nf);
  std::string s;
  for (int i = 2; sym && i < nf && i < 12; ++i) {
    std::string f = sym[i];
    const size_t a = f.find('('), b = f.find('+', a == std::string::npos ? 0 : a);
    if (a != std::string::npos && b != std::string::npos && b > a + 1) {
      std::string mangled = f.substr(a + 1, b - a - 1);
      int st = 0;
      char* dem = abi::__cxa_demangle(mangled.c_str(), nullptr, nullptr, &st);
      f = st == 0 && dem ? dem : mangled;
      free(dem);
      const size_t par = f.find('(');
      if (par != std::string::npos) f = f.substr(0, par);
    }
    s += (s.empty() ? "" : " <- ") + f;
  }
  free(sym);
  return s;
}
}  // namespace

AsyncAllocScope::AsyncAllocScope() { ++t_async_scope; }
AsyncAllocScope::~AsyncAllocScope() { --t_async_scope; }

void dev_cache_release() {
  std::lock_guard<std::mutex> lk(g_alloc_mutex);
  cache_trim_locked();
}

void* dev_alloc(size_t bytes) {
  if (!config().alloc_guard) return dev_alloc_raw(bytes);
  char* raw = static_cast<char*>(dev_alloc_raw(bytes + 2 * GUARD));
  Context& c = ctx();
  hipStream_t st = (c.inited && c.stream) ? c.stream : nullptr;
  PG_HIP(hipMemsetAsync(raw, 0xA5, GUARD, st));
  PG_HIP(hipMemsetAsync(raw + GUARD + bytes, 0xA5, GUARD, st));
  void* user = raw + GUARD;
  std::lock_guard<std::mutex> lk(g_alloc_mutex);
  g_guarded[user] = GuardInfo{raw, bytes, ++g_alloc_seq, call_stack()};
  return user;
}

void dev_free(void* p) {
  if (!config().alloc_guard) { dev_free_raw(p); return; }
  GuardInfo gi;
  {
    std::lock_guard<std::mutex> lk(g_alloc_mutex);
    auto it = g_guarded.find(p);
    if (it == g_guarded.end()) { gi.raw = nullptr; }
    else { gi = it->second; g_guarded.erase(it); }
  }
  if (!gi.raw) { dev_free_raw(p); return; }
  (void)hipDeviceSynchronize();
  std::vector<unsigned char> lo(GUARD), hi(GUARD);
  (void)hipMemcpy(lo.data(), gi.raw, GUARD, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hi.data(), gi.raw + GUARD + gi.bytes, GUARD, hipMemcpyDeviceToHost);
  auto first_bad = [](const std::vector<unsigned char>& g, bool from_end) -> long {
    if (from_end) { for (long i = (long)g.size() - 1; i >= 0; --i) if (g[i] != 0xA5) return i; }
    else { for (size_t i = 0; i < g.size(); ++i) if (g[i] != 0xA5) return (long)i; }
    return -1;
  };
  const long bl = first_bad(lo, true), bh = first_bad(hi, false);
  if (bl >= 0 || bh >= 0) {
    long nl = 0, nh = 0;
    for (unsigned char v : lo) nl += v != 0xA5;
    for (unsigned char v : hi) nh += v != 0xA5;
    fprintf(stderr, "[pg_alloc] GUARD DAMAGED: block #%llu of %zu bytes (incl. 64 of slack): %ld bytes written BELOW it (nearest %ld bytes "
            "before the start), %ld bytes written ABOVE it (first %ld bytes past the end); allocated by: %s\n", gi.seq, gi.bytes, nl,
            bl >= 0 ? (long)GUARD - bl : 0, nh, bh >= 0 ? bh : 0, gi.where.c_str());
  }
  dev_free_raw(gi.raw);
}
}  // namespace pg

extern "C" {

int32_t pg_last_error(char* buf, size_t n) {
  if (buf && n > 0) {
    std::strncpy(buf, g_last_error.c_str(), n - 1);
    buf[n - 1] = 0;
  }
  return 0;
}

int32_t pg_init(int32_t device_id) {
  PG_API_BEGIN
  Context& c = ctx();
  init_device(device_id);
  c.rank = 0;
  c.nranks = 1;
  PG_API_END
}

int32_t pg_get_unique_id(void* out128) {
  PG_API_BEGIN
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  PG_NCCL(rccl::api().GetUniqueId(&id));
  std::memcpy(out128, &id, sizeof(id));
  PG_API_END
}

int32_t pg_init_distributed(int32_t device_id, int32_t rank, int32_t nranks, const void* unique_id128) {
  PG_API_BEGIN
  Context& c = ctx();
  PG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "pg_init_distributed: bad rank / nranks");
  init_device(device_id);
  c.rank = rank;
  c.nranks = nranks;
  if (nranks > 1 || unique_id128 != nullptr) {   // a 1-rank communicator is legal (used to smoke-test the RCCL path)
    PG_REQUIRE(unique_id128 != nullptr, "pg_init_distributed: unique id required when nranks > 1");
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof(id));
    const rccl::Api& R = rccl::api();
    PG_NCCL(R.CommInitRank(&c.comm, nranks, id, rank));
    // A communicator of its own for the halos: each communicator is then driven from exactly ONE stream (comm: compute
    // stream, comm_halo: communication stream), so nothing rests on how RCCL orders operations that one communicator
    // receives from two streams.  Both are used by every rank in the same program order, and a halo exchange is never in
    // flight together with a reduction (halo_end joins the compute stream before the launch whose dots are reduced; the
    // next exchange forks off the compute stream after that reduction): no two RCCL kernels can wait for each other.
    // (an RCCL without ncclCommSplit, or one that refuses it: a second communicator from a second id, which rank 0
    //  hands round through the first one)
    bool split_ok = false;
    if (R.CommSplit) {
      split_ok = R.CommSplit(c.comm, 0, rank, &c.comm_halo, nullptr) == ncclSuccess && c.comm_halo != nullptr;
      if (!split_ok) c.comm_halo = nullptr;
    }
    if (!split_ok) {
      ncclUniqueId id2;
      if (rank == 0) PG_NCCL(R.GetUniqueId(&id2));
      DevBuf<char> buf((i64)sizeof(id2));
      if (rank == 0) buf.upload(reinterpret_cast<const char*>(&id2), (i64)sizeof(id2));
      PG_NCCL(R.Broadcast(buf.p, buf.p, sizeof(id2), ncclChar, 0, c.comm, c.stream));
      buf.download(reinterpret_cast<char*>(&id2), (i64)sizeof(id2));
      PG_NCCL(R.CommInitRank(&c.comm_halo, nranks, id2, rank));
    }
  }
  PG_API_END
}

int32_t pg_finalize(void) {
  PG_API_BEGIN
  Context& c = ctx();
  if (c.inited) {
    (void)hipDeviceSynchronize();
    if (c.comm_halo) { (void)rccl::api().CommDestroy(c.comm_halo); c.comm_halo = nullptr; }
    if (c.comm) { (void)rccl::api().CommDestroy(c.comm); c.comm = nullptr; }
    dev_cache_release();
    if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    if (c.comm_stream) { (void)hipStreamDestroy(c.comm_stream); c.comm_stream = nullptr; }
    if (c.ev_fork) { (void)hipEventDestroy(c.ev_fork); c.ev_fork = nullptr; }
    if (c.ev_join) { (void)hipEventDestroy(c.ev_join); c.ev_join = nullptr; }
    c.inited = false;
  }
  PG_API_END
}

int32_t pg_device_synchronize(void) {
  PG_API_BEGIN
  require_init();
  PG_HIP(hipDeviceSynchronize());
  PG_API_END
}

int32_t pg_set_profiling(int32_t on) {
  PG_API_BEGIN
  ctx().profiling = on != 0;
  PG_API_END
}

int32_t pg_config_string(char* buf, size_t n) {
  PG_API_BEGIN
  const Config& k = config();
  char tmp[1024];
  snprintf(tmp, sizeof(tmp),
           "spmv_variant=%d spmv_xcd=%d spmv_strip=%d spmv_unit_order=%d spmv_march=%d spmv_march_k=%d spmv_minrun=%d spmv_tile_units=%d "
           "spmv_blocks_per_cu=%d halo_overlap=%d speculate=%d poly=%d poly_degree=%d poly_adapt=%d poly_xspace=%d half_test=%d half_batch=%d "
           "krylov_nt=%d fuse_half=%d poly_margin=%g poly_slack=%g poly_hist=%d poly_trend=%d poly_maxdeg=%d recovery_horner=%d gamma_elim=%d diag_elim=%d "
           "guess_states=%d guess_depth=%d guess_async=%d guess_defer=%d async_alloc=%d cache_limit_mb=%lld alloc_poison=%d alloc_guard=%d profile_sample=%d debug=%d",
           k.spmv_variant, k.spmv_xcd, k.spmv_strip, k.unit_order, (int)k.spmv_march, k.spmv_march_k, k.spmv_minrun, k.spmv_tile_units,
           k.spmv_blocks_per_cu, (int)k.halo_overlap, (int)k.speculate_product, (int)k.poly, k.poly_degree, (int)k.poly_adapt, (int)k.poly_xspace, k.half_test,
           (int)k.half_batch, (int)k.krylov_nt, (int)k.fuse_half_update, k.poly_margin, k.poly_slack, k.poly_hist, (int)k.poly_trend, k.poly_maxdeg, (int)k.recovery_horner,
           (int)k.gamma_elim, (int)k.diag_elim, k.guess_n, k.guess_depth, (int)k.guess_async, (int)k.guess_defer, k.async_alloc, k.pool_limit_mb, k.alloc_poison, k.alloc_guard, k.profile_sample, (int)k.debug);
  if (buf && n > 0) {
    std::strncpy(buf, tmp, n - 1);
    buf[n - 1] = 0;
  }
  PG_API_END
}

int32_t pg_device_name(char* buf, size_t n) {
  PG_API_BEGIN
  require_init();
  if (buf && n > 0) {
    std::strncpy(buf, ctx().device_name.c_str(), n - 1);
    buf[n - 1] = 0;
  }
  PG_API_END
}

// ---------------------------------------------------------------------------------------------
// Mesh                                                              src/mesh.jl:47-78
// ---------------------------------------------------------------------------------------------
int32_t pg_mesh_create(int32_t N, const int64_t* n, const double* L, const double* x0, pg_mesh** out) {
  PG_API_BEGIN
  PG_REQUIRE(N >= 1 && N <= 3, "pg_mesh_create: N must be 1, 2 or 3");
  PG_REQUIRE(out != nullptr, "pg_mesh_create: out is NULL");
  PG_REQUIRE(n != nullptr && L != nullptr, "pg_mesh_create: n / L is NULL");
  std::unique_ptr<pg_mesh> guard(new pg_mesh());   // released on success; an argument error below frees it
  pg_mesh* m = guard.get();
  m->N = N;
  for (int d = 0; d < N; ++d) {
    PG_REQUIRE(n[d] >= 1, "pg_mesh_create: n_d must be >= 1");
    m->n[d] = n[d];
    m->L[d] = L[d];
    m->x0[d] = x0 ? x0[d] : 0.0;
    const double h = L[d] / static_cast<double>(n[d]);  // (domain_size[i] / n[i])
    m->centers[d].resize(n[d]);
    m->nodes[d].resize(n[d] + 1);
    for (i64 j = 0; j < n[d]; ++j) m->centers[d][j] = m->x0[d] + static_cast<double>(j) * h;       // :49
    for (i64 j = 0; j <= n[d]; ++j) m->nodes[d][j] = m->x0[d] + (static_cast<double>(j) + 0.5) * h; // :50
  }
  if (ctx().inited) {
    for (int d = 0; d < N; ++d) {
      m->d_nodes[d].alloc(n[d] + 1);
      m->d_nodes[d].upload(m->nodes[d].data(), n[d] + 1);
    }
  }
  *out = guard.release();
  PG_API_END
}

int32_t pg_mesh_destroy(pg_mesh* m) {
  PG_API_BEGIN
  delete m;
  PG_API_END
}

int32_t pg_mesh_get_centers(const pg_mesh* m, int32_t d, double* out, int64_t len) {
  PG_API_BEGIN
  PG_REQUIRE(m && d >= 0 && d < m->N && len == m->n[d], "pg_mesh_get_centers: bad arguments");
  std::memcpy(out, m->centers[d].data(), sizeof(double) * len);
  PG_API_END
}

int32_t pg_mesh_get_nodes(const pg_mesh* m, int32_t d, double* out, int64_t len) {
  PG_API_BEGIN
  PG_REQUIRE(m && d >= 0 && d < m->N && len == m->n[d] + 1, "pg_mesh_get_nodes: bad arguments");
  std::memcpy(out, m->nodes[d].data(), sizeof(double) * len);
  PG_API_END
}

int32_t pg_mesh_num_border_cells(const pg_mesh* m, int64_t* out) {
  PG_API_BEGIN
  PG_REQUIRE(m && out, "pg_mesh_num_border_cells: bad arguments");
  // prod(n) - prod(max(n-2,0))
  i64 all = 1, inner = 1;
  for (int d = 0; d < m->N; ++d) {
    all *= m->n[d];
    inner *= std::max<i64>(m->n[d] - 2, 0);
  }
  *out = all - inner;
  PG_API_END
}

int32_t pg_mesh_get_border_cells(const pg_mesh* mc, int64_t* idx, double* pos, int32_t* key) {
  PG_API_BEGIN
  PG_REQUIRE(mc, "pg_mesh_get_border_cells: mesh is NULL");
  pg_mesh* m = const_cast<pg_mesh*>(mc);
  mesh_build_border(m);
  const size_t nb = m->border_key.size();
  if (idx) std::memcpy(idx, m->border_idx.data(), sizeof(i64) * nb * m->N);
  if (pos) std::memcpy(pos, m->border_pos.data(), sizeof(double) * nb * m->N);
  if (key) std::memcpy(key, m->border_key.data(), sizeof(i32) * nb);
  PG_API_END
}

int32_t pg_partition_planes(const int64_t* weight, int64_t nplanes, int32_t nranks, int64_t* bounds) {
  PG_API_BEGIN
  PG_REQUIRE(weight && bounds && nplanes >= 1 && nranks >= 1, "pg_partition_planes: bad arguments");
  PG_REQUIRE(nplanes >= nranks, "pg_partition_planes: fewer planes than ranks");
  pghost::partition_planes(weight, nplanes, nranks, bounds);   // pg_host_algos.h (also built with ASan / UBSan for the CPU suite)
  PG_API_END
}

int32_t pg_slab_numbering_host(int32_t K, int64_t plane, int64_t nplanes, int64_t p0, int64_t p1, const uint8_t* active,
                               int64_t* out) {
  PG_API_BEGIN
  PG_REQUIRE(K >= 1 && K <= pghost::SLAB_MAX_KINDS && plane >= 1 && nplanes >= 1 && p0 >= 0 && p1 > p0 && p1 <= nplanes && active && out,
             "pg_slab_numbering_host: bad arguments");
  const pghost::SlabPlanes g = pghost::slab_planes(p0, p1, nplanes, CAP_HALO);
  const i64 Mloc = (g.s1 - g.s0) * plane;
  const i64 lcA0 = (g.a0 - g.s0) * plane, lcA1 = (g.a1 - g.s0) * plane;        // flags count inside the one ghost plane each side
  const i64 lcO = (p0 - g.s0) * plane, lcU = (p1 - g.s0) * plane, lcFE = (p0 + 1 - g.s0) * plane, lcLB = (p1 - 1 - g.s0) * plane;
  i64 posO[pghost::SLAB_MAX_KINDS], posU[pghost::SLAB_MAX_KINDS], posFE[pghost::SLAB_MAX_KINDS], posLB[pghost::SLAB_MAX_KINDS],
      tot[pghost::SLAB_MAX_KINDS];
  for (int k = 0; k < K; ++k) {
    i64 c = 0;
    posO[k] = posU[k] = posFE[k] = posLB[k] = -1;
    for (i64 lc = 0; lc <= Mloc; ++lc) {
      if (lc == lcO) posO[k] = c;
      if (lc == lcU) posU[k] = c;
      if (lc == lcFE) posFE[k] = c;
      if (lc == lcLB) posLB[k] = c;
      if (lc < Mloc && lc >= lcA0 && lc < lcA1 && active[(i64)k * Mloc + lc]) ++c;
    }
    tot[k] = c;
  }
  pghost::SlabNumbering sn;
  pghost::slab_numbering(K, posO, posU, posFE, posLB, tot, p0 > 0, p1 < nplanes, sn);
  i64 q = 0;
  out[q++] = sn.n_own;
  out[q++] = sn.n_ghost;
  const i64* arrays[10] = {sn.cnt_own, sn.off_own, sn.cntL, sn.offL, sn.cntU, sn.offU, sn.sendL_off, sn.sendL_cnt, sn.sendU_off, sn.sendU_cnt};
  for (const i64* a : arrays)
    for (int k = 0; k < K; ++k) out[q++] = a[k];
  PG_API_END
}

}  // extern "C"

namespace pg {

// src/mesh.jl:57-74: for d, for face in (1, dims[d]), Iterators.product (dim 1 fastest), unique!
void mesh_build_border(pg_mesh* m) {
  if (m->border_built) return;
  const int N = m->N;
  std::vector<i64>& bi = m->border_idx;
  std::set<std::array<i64, 3>> seen;
  auto emit = [&](i64 i0, i64 i1, i64 i2) {
    std::array<i64, 3> key{i0, i1, i2};
    if (!seen.insert(key).second) return;
    const i64 id[3] = {i0, i1, i2};
    for (int d = 0; d < N; ++d) {
      bi.push_back(id[d] + 1);
      m->border_pos.push_back(m->centers[d][id[d]]);
    }
    m->border_key.push_back(border_key_of(N, m->n, i0, i1, i2));
  };
  for (int d = 0; d < N; ++d) {
    const i64 faces[2] = {0, m->n[d] - 1};
    for (int f = 0; f < 2; ++f) {
      i64 lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
      for (int k = 0; k < N; ++k) hi[k] = m->n[k];
      lo[d] = faces[f];
      hi[d] = faces[f] + 1;
      for (i64 i2 = lo[2]; i2 < hi[2]; ++i2)
        for (i64 i1 = lo[1]; i1 < hi[1]; ++i1)
          for (i64 i0 = lo[0]; i0 < hi[0]; ++i0) emit(i0, i1, i2);
    }
  }
  m->border_built = true;
}

}  // namespace pg

pg::Slab pg_mesh::base_slab() const {
  pg::Slab s;
  s.N = N;
  pg::i64 st = 1;
  for (int d = 0; d < 3; ++d) {
    s.n[d] = d < N ? n[d] : 1;
    s.ext[d] = d < N ? n[d] + 1 : 1;
    s.stride[d] = st;
    st *= s.ext[d];
  }
  s.M = st;
  s.nplanes = s.ext[N - 1];
  s.plane = s.M / s.nplanes;
  s.set_own(0, s.nplanes);
  return s;
}
