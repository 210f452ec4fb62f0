// pg_comm.hip -- the only place that talks to RCCL (C1/C2/C3 of SURVEY.md section 2.3).
//
// Two backends behind the same four calls:
//   * RCCL over xGMI (production): one process per GPU, two communicators each driven from one stream: `comm` for the
//     reductions (compute stream) and `comm_halo` for the nearest-neighbour send/recv (communication stream, forked off
//     and joined back to the compute stream with events); no host synchronisation.
//   * LocalComm (diagnostics): N "virtual ranks" = N host threads of ONE process sharing one GPU, exchanging
//     through device-to-device copies and pthread barriers.  It exists so that the slab partition, the ghost
//     numbering and every send/recv offset of the multi-GPU path can be verified bit for bit on a 1-GPU box
//     (pg_debug_run_virtual_ranks, tests/test_gpu_virtual_ranks.py).  Same reduction order as RCCL is NOT
//     assumed anywhere: results only need to be identical on all ranks, which both backends guarantee.
#include <dlfcn.h>
#include <pthread.h>

#include <mutex>

#include "pg_krylov.h"

namespace pg {

struct LocalComm {
  int nranks = 1;
  pthread_barrier_t bar;
  std::vector<std::vector<double>> slot_f64;                 // [rank][count]
  std::vector<std::vector<unsigned long long>> slot_u64;     // [rank][count]
  struct Post { const double* sendL[MAX_KINDS]; const double* sendU[MAX_KINDS]; i64 cntL[MAX_KINDS]; i64 cntU[MAX_KINDS]; };
  std::vector<Post> posts;
  explicit LocalComm(int n) : nranks(n), slot_f64(n), slot_u64(n), posts(n) { pthread_barrier_init(&bar, nullptr, n); }
  ~LocalComm() { pthread_barrier_destroy(&bar); }
  void barrier() { pthread_barrier_wait(&bar); }
};

namespace rccl {
namespace {
Api g_api;
bool g_loaded = false;
std::mutex g_load_mutex;
}  // namespace
bool loaded() { return g_loaded; }
const Api& api() {
  std::lock_guard<std::mutex> lk(g_load_mutex);
  if (g_loaded) return g_api;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) throw Error(std::string("penguin_hip: RCCL (librccl.so.1) cannot be loaded -- needed for runs on several GPUs: ") + dlerror());
  auto need = [&](const char* sym, bool required = true) -> void* {
    void* p = dlsym(h, sym);
    if (!p && required) throw Error(std::string("penguin_hip: RCCL lacks ") + sym);
    return p;
  };
  g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(need("ncclGetUniqueId"));
  g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(need("ncclCommInitRank"));
  g_api.CommSplit = reinterpret_cast<decltype(g_api.CommSplit)>(need("ncclCommSplit", false));
  g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(need("ncclCommDestroy"));
  g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(need("ncclAllReduce"));
  g_api.Broadcast = reinterpret_cast<decltype(g_api.Broadcast)>(need("ncclBroadcast"));
  g_api.Send = reinterpret_cast<decltype(g_api.Send)>(need("ncclSend"));
  g_api.Recv = reinterpret_cast<decltype(g_api.Recv)>(need("ncclRecv"));
  g_api.GroupStart = reinterpret_cast<decltype(g_api.GroupStart)>(need("ncclGroupStart"));
  g_api.GroupEnd = reinterpret_cast<decltype(g_api.GroupEnd)>(need("ncclGroupEnd"));
  g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(need("ncclGetErrorString"));
  g_loaded = true;
  return g_api;
}
}  // namespace rccl

static thread_local Context* tl_ctx = nullptr;
void set_thread_context(Context* c) { tl_ctx = c; }
Context* thread_context() { return tl_ctx; }

LocalComm* local_comm_create(int nranks) { return new LocalComm(nranks); }
void local_comm_destroy(LocalComm* c) { delete c; }

template <class T, class Slots>
static void local_allreduce(LocalComm* lc, int rank, Slots& slots, T* dev, i64 count, hipStream_t st, bool do_max) {
  std::vector<T>& mine = slots[rank];
  mine.resize(count);
  PG_HIP(hipMemcpyAsync(mine.data(), dev, sizeof(T) * count, hipMemcpyDeviceToHost, st));
  PG_HIP(hipStreamSynchronize(st));
  lc->barrier();
  std::vector<T> res(count);
  for (i64 i = 0; i < count; ++i) {
    T acc = slots[0][i];
    for (int r = 1; r < lc->nranks; ++r) acc = do_max ? (slots[r][i] > acc ? slots[r][i] : acc) : acc + slots[r][i];
    res[i] = acc;
  }
  lc->barrier();   // everyone has read every slot
  PG_HIP(hipMemcpyAsync(dev, res.data(), sizeof(T) * count, hipMemcpyHostToDevice, st));
  PG_HIP(hipStreamSynchronize(st));
}

void comm_allreduce_sum_f64(double* dev, int count, hipStream_t st) {
  Context& cx = ctx();
  if (cx.nranks == 1 && !cx.comm) return;
  if (cx.local) local_allreduce<double>(cx.local, cx.rank, cx.local->slot_f64, dev, count, st, false);
  else PG_NCCL(rccl::api().AllReduce(dev, dev, count, ncclDouble, ncclSum, cx.comm, st));
}

void comm_allreduce_max_f64(double* dev, int count, hipStream_t st) {
  Context& cx = ctx();
  if (cx.nranks == 1 && !cx.comm) return;
  if (cx.local) local_allreduce<double>(cx.local, cx.rank, cx.local->slot_f64, dev, count, st, true);
  else PG_NCCL(rccl::api().AllReduce(dev, dev, count, ncclDouble, ncclMax, cx.comm, st));
}

void comm_allreduce_sum_u64(unsigned long long* dev, i64 count, hipStream_t st) {
  Context& cx = ctx();
  if (cx.nranks == 1 && !cx.comm) return;
  if (cx.local) local_allreduce<unsigned long long>(cx.local, cx.rank, cx.local->slot_u64, dev, count, st, false);
  else PG_NCCL(rccl::api().AllReduce(dev, dev, count, ncclUint64, ncclSum, cx.comm, st));
}

static void rccl_halo(const Numbering& nb, const Slab& slab, double* vec, hipStream_t st);

// C1: ghost segments of `vec` <- boundary chunks of the neighbours' owned parts.  The chunks are contiguous
// (Numbering), so there is nothing to pack: one send and one recv per unknown kind and neighbour, grouped.
void halo_exchange(const Numbering& nb, const Slab& slab, double* vec, hipStream_t st) {
  Context& cx = ctx();
  if (cx.nranks == 1) return;
  const bool has_lo = slab.p0 > 0, has_hi = slab.p1 < slab.nplanes;
  if (cx.local) {
    LocalComm* lc = cx.local;
    LocalComm::Post& me = lc->posts[cx.rank];
    for (int k = 0; k < MAX_KINDS; ++k) {
      me.sendL[k] = vec + nb.sendL_off[k]; me.cntL[k] = k < nb.K ? nb.sendL_cnt[k] : 0;
      me.sendU[k] = vec + nb.sendU_off[k]; me.cntU[k] = k < nb.K ? nb.sendU_cnt[k] : 0;
    }
    PG_HIP(hipStreamSynchronize(st));   // my boundary values are final
    lc->barrier();
    for (int k = 0; k < nb.K; ++k) {
      if (has_lo && nb.cntL[k] > 0) {   // what the lower neighbour sends UP is my lower ghost
        const LocalComm::Post& o = lc->posts[cx.rank - 1];
        PG_REQUIRE(o.cntU[k] == nb.cntL[k], "halo size mismatch with the lower neighbour");
        PG_HIP(hipMemcpyAsync(vec + nb.offL[k], o.sendU[k], sizeof(double) * nb.cntL[k], hipMemcpyDeviceToDevice, st));
      }
      if (has_hi && nb.cntU[k] > 0) {
        const LocalComm::Post& o = lc->posts[cx.rank + 1];
        PG_REQUIRE(o.cntL[k] == nb.cntU[k], "halo size mismatch with the upper neighbour");
        PG_HIP(hipMemcpyAsync(vec + nb.offU[k], o.sendL[k], sizeof(double) * nb.cntU[k], hipMemcpyDeviceToDevice, st));
      }
    }
    PG_HIP(hipStreamSynchronize(st));
    lc->barrier();                      // nobody overwrites a source before it has been copied
    return;
  }
  // RCCL: always on the communication stream (the halo communicator is never driven from the compute stream)
  halo_begin(nb, slab, vec, st);
  halo_end(st);
}

void halo_begin(const Numbering& nb, const Slab& slab, double* vec, hipStream_t st) {
  Context& cx = ctx();
  if (cx.nranks == 1) return;
  if (cx.local) { halo_exchange(nb, slab, vec, st); return; }
  if (!cx.ev_fork) PG_HIP(hipEventCreateWithFlags(&cx.ev_fork, hipEventDisableTiming));
  if (!cx.ev_join) PG_HIP(hipEventCreateWithFlags(&cx.ev_join, hipEventDisableTiming));
  PG_HIP(hipEventRecord(cx.ev_fork, st));                      // the producer of vec's owned part has run
  PG_HIP(hipStreamWaitEvent(cx.comm_stream, cx.ev_fork, 0));
  rccl_halo(nb, slab, vec, cx.comm_stream);
  PG_HIP(hipEventRecord(cx.ev_join, cx.comm_stream));
}

void halo_end(hipStream_t st) {
  Context& cx = ctx();
  if (cx.nranks == 1 || cx.local) return;
  PG_HIP(hipStreamWaitEvent(st, cx.ev_join, 0));
}

static void rccl_halo(const Numbering& nb, const Slab& slab, double* vec, hipStream_t st) {
  Context& cx = ctx();
  const bool has_lo = slab.p0 > 0, has_hi = slab.p1 < slab.nplanes;
  // ranks own increasing plane ranges: the lower neighbour is rank-1, the upper one rank+1
  const rccl::Api& R = rccl::api();
  PG_NCCL(R.GroupStart());
  for (int k = 0; k < nb.K; ++k) {
    if (has_lo) {
      if (nb.sendL_cnt[k] > 0) PG_NCCL(R.Send(vec + nb.sendL_off[k], nb.sendL_cnt[k], ncclDouble, cx.rank - 1, cx.comm_halo, st));
      if (nb.cntL[k] > 0) PG_NCCL(R.Recv(vec + nb.offL[k], nb.cntL[k], ncclDouble, cx.rank - 1, cx.comm_halo, st));
    }
    if (has_hi) {
      if (nb.sendU_cnt[k] > 0) PG_NCCL(R.Send(vec + nb.sendU_off[k], nb.sendU_cnt[k], ncclDouble, cx.rank + 1, cx.comm_halo, st));
      if (nb.cntU[k] > 0) PG_NCCL(R.Recv(vec + nb.offU[k], nb.cntU[k], ncclDouble, cx.rank + 1, cx.comm_halo, st));
    }
  }
  PG_NCCL(R.GroupEnd());
}

}  // namespace pg
