// pg_common.h -- shared internals of libpenguin_hip.so (gfx950 only; no CUDA paths, no shims).
#pragma once
#include <hip/hip_runtime.h>
#include "pg_rccl.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/penguin_hip.h"

namespace pg {

using i64 = int64_t;
using i32 = int32_t;

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

void set_last_error(const std::string& msg);

#define PG_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      throw pg::Error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                      std::to_string(__LINE__) + " (" #expr ")");                                 \
  } while (0)

#define PG_NCCL(expr)                                                                             \
  do {                                                                                            \
    ncclResult_t _e = (expr);                                                                     \
    if (_e != ncclSuccess)                                                                        \
      throw pg::Error(std::string("RCCL error: ") + pg::rccl::api().GetErrorString(_e) + " at " + __FILE__ + ":" + \
                      std::to_string(__LINE__) + " (" #expr ")");                                 \
  } while (0)

#define PG_REQUIRE(cond, msg)                                   \
  do {                                                          \
    if (!(cond)) throw pg::Error(std::string(msg));             \
  } while (0)

// every extern "C" entry point is wrapped:  PG_API_BEGIN ... PG_API_END
#define PG_API_BEGIN try {
#define PG_API_END                                  \
  return 0;                                         \
  }                                                 \
  catch (const std::exception& e) {                 \
    pg::set_last_error(e.what());                   \
    return 1;                                       \
  }                                                 \
  catch (...) {                                     \
    pg::set_last_error("unknown exception");        \
    return 2;                                       \
  }

// Every tuning / variant selector of the library, read from the environment ONCE (first use) into this struct: nothing
// else in the library calls getenv, no selector is re-read per solve, and pg_config_string() reports what is active so that
// a measurement can name exactly what ran.  Defaults = the shipped configuration; the variants exist for A/B measurements
// and for the kernel-vs-kernel parity tests.
struct Config {
  bool debug = false;             // PG_DEBUG: set-up laps and solver diagnostics on stderr
  int alloc_poison = 0;           // PG_ALLOC_POISON=1: every device allocation is filled with a NaN pattern (uninitialised reads show)
  int async_alloc = 0;            // PG_ASYNC_ALLOC: 1 every allocation through the library's block cache, -1 none, 0 inside AsyncAllocScopes
  int alloc_guard = 0;            // PG_ALLOC_GUARD=1: guard bands around every device block, checked at free (out-of-bounds writes)
  long long pool_limit_mb = 0;    // PG_POOL_LIMIT_MB: requests of this size and more bypass the block cache (0: no limit)
  // Krylov driver (pg_krylov.hip)
  bool poly = true;               // PG_POLY: polynomial right preconditioner where admissible
  int poly_degree = 6;            // PG_POLY_DEGREE: degree of the first solve on a matrix / of every solve when adaptation is off
  bool poly_adapt = true;         // PG_POLY_ADAPT (default: on unless PG_POLY_DEGREE is given)
  bool poly_xspace = true;        // PG_POLY_XSPACE
  int half_test = -1;             // PG_HALF_TEST: -1 automatic (degree >= 3), 0 off, 1 on
  bool half_batch = true;         // PG_HALF_BATCH
  bool krylov_nt = true;          // PG_KRYLOV_NT: stream hints on the dead vectors of the vector kernels
  bool fuse_half_update = true;   // PG_FUSE_HALF: x-space loop: x += α M⁻¹p inside the s kernel (k_bicg_s_x) instead of k_bicg_half
  double poly_margin = 1.0;       // PG_POLY_MARGIN
  double poly_slack = -1.0;       // PG_POLY_SLACK (< 0: 0.3 products in the x-space form, 0 otherwise)
  int poly_hist = 3;              // PG_POLY_HIST
  bool poly_trend = true;         // PG_POLY_TREND: the degree estimate follows the trend of the last estimates
  int poly_maxdeg = 0;            // PG_POLY_MAXDEG (0: 32 x-space / 10 y-space)
  int poly_mindeg = 4;            // PG_POLY_MINDEG: the smallest degree the per-solve choice takes (2 .. 8)
  bool recovery_horner = true;    // PG_RECOVERY_HORNER (y-space form)
  int profile_sample = 3;         // PG_PROFILE_SAMPLE
  bool gamma_elim = true;         // PG_GAMMA_ELIM
  bool diag_elim = true;          // PG_DIAG_ELIM
  double diag_elim_frac = 0.0005; // PG_DIAG_ELIM_FRAC: the compact loop system is built when at least this share of the rows is alone on its
                                  // diagonal (it was 2 %: 2-D problems, whose share is below that, then never reached the quiet steps'
                                  // folded start and extrapolated start -- 2048^2: 3420 -> 5370 (BE) / 3450 -> 4440 (CN) steps/s)
  // SpMV (pg_spmv.hip)
  int spmv_variant = 70;          // PG_SPMV_VARIANT
  int spmv_xcd = 1;               // PG_SPMV_XCD (bits 8..: diagnostics that give WRONG products, in the diagnostic build of the kernel only)
  int spmv_strip = 16;            // PG_SPMV_STRIP
  int spmv_march_k = 0;           // PG_SPMV_MARCH_K (0: the compiled-in unit depth)
  int spmv_minrun = 24;           // PG_SPMV_MINRUN
  bool spmv_march = true;         // PG_SPMV_MARCH
  int spmv_tile_units = 0;        // PG_SPMV_TILE_UNITS
  int spmv_blocks_per_cu = 0;     // PG_SPMV_BLOCKS_PER_CU (0: 4 slice kernel / 6 CSR kernels)
  bool halo_overlap = true;       // PG_HALO_OVERLAP
  // the start of a quiet time step extrapolated from older states (pg_solver.hip, GuessArgs / k_guess_fit):
  int guess_n = 4;                // PG_GUESS_STATES: older states read at most (0: off, <= 4)
  int guess_depth = 7;            // PG_GUESS_DEPTH: older states kept to choose from (<= 7)
  bool guess_defer = true;        // PG_GUESS_DEFER: compact x-space loop: the extrapolated state is formed by the solve's first update of x
  bool guess_always = false;      // PG_GUESS_ALWAYS=1: also where the fit's launch is not expected to pay (small, easy systems)
  bool guess_async = false;       // PG_GUESS_ASYNC=1: one rank: the fit runs on a stream of its own beside the solve (measured: the
                                  // Horner launches beside it slow down by more than the 21 us it takes off the stream: 586 vs 592 steps/s)
  int guess_monitor = 256;        // PG_GUESS_MONITOR: the fit samples every n-th chunk of 256 rows
  double guess_pass_cost = 0.34;  // PG_GUESS_PASS_COST: one more vector read of k_rhs_init_c, in products of the loop
  double guess_gain = 0.8;        // PG_GUESS_GAIN: products saved per product the fit predicts (what is left after the
                                  // extrapolation sits where the polynomial is weakest)
  bool speculate_product = true;  // PG_SPECULATE: pg_solver_run queues the next step's first product behind a solve's first batch
  int unit_order = 0;             // PG_SPMV_UNIT_ORDER: 0 by first row within (strip, plane); 1 units cut at common planes, window-major
                                  // (a block's four waves on four neighbouring lines; measured 59.0 vs 54.6 us per Horner launch: off)
};
const Config& config();              // pg_context.hip

struct LocalComm;   // in-process "virtual ranks" backend (pg_comm.hip), diagnostics only

struct Context {
  bool inited = false;
  LocalComm* local = nullptr;        // non-null: this thread is a virtual rank of a LocalComm
  int device = 0;
  hipStream_t stream = nullptr;      // compute stream: every kernel of the path runs here
  hipStream_t comm_stream = nullptr; // halo exchange stream (overlaps interior SpMV rows)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // compute stream -> comm stream -> compute stream (halo_begin / halo_end)
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;         // reductions of the Krylov scalars and set-up collectives: compute stream ONLY
  ncclComm_t comm_halo = nullptr;    // ghost-plane send / recv: communication stream ONLY (ncclCommSplit of comm)
  bool profiling = false;
  std::string device_name;
};
Context& ctx();                      // the calling thread's context (virtual ranks override the process context)
void require_init();
void set_thread_context(Context* c);
Context* thread_context();
LocalComm* local_comm_create(int nranks);
void local_comm_destroy(LocalComm* c);
// collectives on device buffers, enqueued on `st` (no-ops on one rank)            C2 / C3 of SURVEY.md 2.3
void comm_allreduce_sum_f64(double* dev, int count, hipStream_t st);
void comm_allreduce_max_f64(double* dev, int count, hipStream_t st);
void comm_allreduce_sum_u64(unsigned long long* dev, i64 count, hipStream_t st);

// ---- device buffer -------------------------------------------------------------------------
// Allocation: hipMalloc / hipFree, or -- inside an AsyncAllocScope (the moving path's entry points; everywhere with
// PG_ASYNC_ALLOC=1) -- the library's own cache of freed blocks, which hands memory out again without synchronising the
// device: everything of the path is queued on one stream (pg_context.hip "device memory" has the reasoning, and why the
// runtime's hipMallocAsync pool is not used).  17 -> 5 ms per slab of the moving solver at 1024².
void* dev_alloc(size_t bytes);   // pg_context.hip
void dev_free(void* p);
void dev_cache_release();        // idle cached blocks back to the runtime (one device synchronisation)
struct AsyncAllocScope {
  AsyncAllocScope();
  ~AsyncAllocScope();
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  i64 n = 0;
  DevBuf() = default;
  explicit DevBuf(i64 n_) { alloc(n_); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) dev_free(p);
    p = nullptr; n = 0;
  }
  void alloc(i64 n_) {
    release();
    n = n_;
    // 64 bytes of slack: 16-byte loads that start at the last element (SpMV pair loads) stay inside the allocation
    if (n > 0) p = static_cast<T*>(dev_alloc(sizeof(T) * static_cast<size_t>(n) + 64));
  }
  void zero() {
    if (n > 0) PG_HIP(hipMemsetAsync(p, 0, sizeof(T) * static_cast<size_t>(n), ctx().stream));
  }
  void upload(const T* h, i64 count, i64 offset = 0) {
    if (count > 0) {
      PG_HIP(hipMemcpyAsync(p + offset, h, sizeof(T) * static_cast<size_t>(count), hipMemcpyHostToDevice, ctx().stream));
      PG_HIP(hipStreamSynchronize(ctx().stream));
    }
  }
  void download(T* h, i64 count, i64 offset = 0) const {
    if (count > 0) {
      PG_HIP(hipMemcpyAsync(h, p + offset, sizeof(T) * static_cast<size_t>(count), hipMemcpyDeviceToHost, ctx().stream));
      PG_HIP(hipStreamSynchronize(ctx().stream));
    }
  }
};

// grow-only pinned host staging buffers (two slots): device -> host copies of tens of MB run at PCIe speed instead of
// the pageable-memory rate; allocating pinned memory per call would cost more than it saves
inline void* pinned_scratch(int slot, size_t bytes) {
  static thread_local void* buf[2] = {nullptr, nullptr};
  static thread_local size_t cap[2] = {0, 0};
  if (bytes > cap[slot]) {
    if (buf[slot]) (void)hipHostFree(buf[slot]);
    PG_HIP(hipHostMalloc(&buf[slot], bytes + (bytes >> 2)));
    cap[slot] = bytes + (bytes >> 2);
  }
  return buf[slot];
}

// PG_DEBUG=1: wall-clock laps of the set-up phases on stderr (device synchronised at each lap)
struct Laps {
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit Laps() : on(config().debug), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[pg_laps] %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// two HIP events that are destroyed on every exit path (timed regions whose body may throw)
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  EventPair() {
    PG_HIP(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
      (void)hipEventDestroy(e0);
      throw Error("hipEventCreate failed");
    }
  }
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
  EventPair(const EventPair&) = delete;
  EventPair& operator=(const EventPair&) = delete;
};

inline int grid_for(i64 n, int block, int cap = 256 * 8) {
  i64 g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

// ---- slab of the padded grid owned / stored by this rank --------------------------------------
// The padded grid has ext[d] = n_d + 1 nodes per dimension (d < N), linear index dim-0 fastest.
// The slowest dimension (N-1) is cut into planes; a rank owns planes [p0,p1) and stores planes
// [s0,s1) = [p0-HALO, p1+HALO) ∩ [0,nplanes) so that every stencil / active-set decision for its
// owned rows and its one ghost plane each side can be taken locally (capacities are pure functions
// of position: ghost planes are recomputed, never communicated).
constexpr int CAP_HALO = 3;

struct Slab {
  int N = 0;
  i64 n[3] = {1, 1, 1};     // cells per dim
  i64 ext[3] = {1, 1, 1};   // n+1 for d<N, 1 otherwise
  i64 stride[3] = {1, 1, 1};
  i64 plane = 1;            // cells per plane of the slowest dim
  i64 nplanes = 1;          // ext[N-1]
  i64 M = 1;                // global padded size
  i64 p0 = 0, p1 = 1;       // owned planes
  i64 s0 = 0, s1 = 1;       // stored planes
  i64 Mloc() const { return (s1 - s0) * plane; }
  i64 first_cell() const { return s0 * plane; }  // global linear index of local cell 0
  void set_own(i64 a, i64 b) {
    p0 = a; p1 = b;
    s0 = p0 - CAP_HALO; if (s0 < 0) s0 = 0;
    s1 = p1 + CAP_HALO; if (s1 > nplanes) s1 = nplanes;
  }
};

}  // namespace pg

// ---- opaque handle definitions -------------------------------------------------------------
struct pg_mesh {
  int N = 0;
  pg::i64 n[3] = {1, 1, 1};
  double L[3] = {0, 0, 0};
  double x0[3] = {0, 0, 0};
  std::vector<double> centers[3];
  std::vector<double> nodes[3];
  // border cells, reference order (src/mesh.jl:57-74 + unique!): built lazily
  bool border_built = false;
  std::vector<pg::i64> border_idx;   // nb*N, 1-based
  std::vector<double> border_pos;    // nb*N
  std::vector<pg::i32> border_key;   // nb
  pg::DevBuf<double> d_nodes[3];     // device copies of the node coordinates
  pg::Slab base_slab() const;
};

namespace pg {
void mesh_build_border(pg_mesh* m);
// classify_boundary_cell_fast, src/solver.jl:379-409, 0-based cell indices, returns PG_KEY_* or -1
__host__ __device__ inline int border_key_of(int N, const i64* n, i64 i0, i64 i1, i64 i2) {
  if (N >= 2) {
    if (i1 == 0) return PG_KEY_LEFT;
    if (i1 == n[1] - 1) return PG_KEY_RIGHT;
  }
  if (i0 == 0) return PG_KEY_BOTTOM;
  if (i0 == n[0] - 1) return PG_KEY_TOP;
  if (N >= 3) {
    if (i2 == 0) return PG_KEY_BACKWARD;
    if (i2 == n[2] - 1) return PG_KEY_FORWARD;
  }
  return -1;
}
}  // namespace pg
