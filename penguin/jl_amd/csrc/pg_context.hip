// pg_context.hip -- device / stream / RCCL communicator lifetime, error channel, Mesh.
#include "pg_common.h"

#include <mutex>
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
thread_local int t_async_scope = 0;
std::mutex g_pool_mutex;
std::unordered_set<void*> g_pool_ptrs;
int async_mode() {      // PG_ASYNC_ALLOC: 1 every allocation from the pool, -1 none (not even inside an AsyncAllocScope)
  static const int m = getenv("PG_ASYNC_ALLOC") ? atoi(getenv("PG_ASYNC_ALLOC")) : 0;
  return m;
}
bool async_everywhere() { return async_mode() > 0; }
// Requests of this size and more go the ordinary way (PG_POOL_LIMIT_MB, default 64).  Measured on this runtime: with blocks of
// 370 MB in the pool (a 3072² slab) buffers came back with stale zeros in them and slabs took 100+ ms, with > 2 GiB requests
// the process aborted; up to 165 MB (2048²) everything was fine.  The gain is in the dozens of SMALL buffers a slab creates
// and frees (lists, counters, per-row flags): 64 MB keeps all of it (1024²: 5.5 ms per slab against 17 without the pool).
size_t pool_limit() {
  static const size_t lim = (size_t)(getenv("PG_POOL_LIMIT_MB") ? atoi(getenv("PG_POOL_LIMIT_MB")) : 64) << 20;
  return lim;
}
}  // namespace
AsyncAllocScope::AsyncAllocScope() { ++t_async_scope; }
AsyncAllocScope::~AsyncAllocScope() { --t_async_scope; }

void* dev_alloc(size_t bytes) {
  void* p = nullptr;
  Context& c = ctx();
  if ((t_async_scope > 0 || async_everywhere()) && async_mode() >= 0 && c.inited && c.stream && !c.local && bytes < pool_limit()) {
    static bool pool_set = false;
    if (!pool_set) {
      hipMemPool_t pool;
      if (hipDeviceGetDefaultMemPool(&pool, c.device) == hipSuccess) {
        uint64_t keep = ~0ull;     // never hand memory back at synchronisation points
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
      }
      pool_set = true;
    }
    if (hipMallocAsync(&p, bytes, c.stream) == hipSuccess && p) {
      std::lock_guard<std::mutex> lk(g_pool_mutex);
      g_pool_ptrs.insert(p);
      return p;
    }
    (void)hipGetLastError();
    p = nullptr;
  }
  PG_HIP(hipMalloc(&p, bytes));
  return p;
}

void dev_free(void* p) {
  bool pooled = false;
  {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    pooled = g_pool_ptrs.erase(p) > 0;
  }
  if (pooled) {
    Context* c = thread_context();
    Context& cc = c ? *c : ctx();
    if (cc.inited && cc.stream && hipFreeAsync(p, cc.stream) == hipSuccess) return;
    (void)hipGetLastError();
  }
  (void)hipFree(p);
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
  PG_NCCL(ncclGetUniqueId(&id));
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
    PG_NCCL(ncclCommInitRank(&c.comm, nranks, id, rank));
    // A communicator of its own for the halos: each communicator is then driven from exactly ONE stream (comm: compute
    // stream, comm_halo: communication stream), so nothing rests on how RCCL orders operations that one communicator
    // receives from two streams.  Both are used by every rank in the same program order, and a halo exchange is never in
    // flight together with a reduction (halo_end joins the compute stream before the launch whose dots are reduced; the
    // next exchange forks off the compute stream after that reduction): no two RCCL kernels can wait for each other.
    PG_NCCL(ncclCommSplit(c.comm, 0, rank, &c.comm_halo, nullptr));
  }
  PG_API_END
}

int32_t pg_finalize(void) {
  PG_API_BEGIN
  Context& c = ctx();
  if (c.inited) {
    (void)hipDeviceSynchronize();
    if (c.comm_halo) { (void)ncclCommDestroy(c.comm_halo); c.comm_halo = nullptr; }
    if (c.comm) { (void)ncclCommDestroy(c.comm); c.comm = nullptr; }
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
