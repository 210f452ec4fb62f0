// pg_debug.hip -- diagnostics: run the slab-decomposed path with N "virtual ranks" on ONE GPU.
//
// Every virtual rank is a host thread with its own Context (rank, nranks, stream) and the LocalComm backend of
// pg_comm.hip; it executes exactly the code a real rank executes (partition by per-plane weights, capacities on
// its stored planes, numbering with ghost segments, assembly, BiCGStab with halo exchange and all-reduced
// dots).  The result must equal the single-rank result to solver tolerance; partition bounds and per-rank sizes
// are returned for inspection.  Used by tests/test_gpu_virtual_ranks.py; never on the product path.
#include <pthread.h>

#include <memory>
#include <string>
#include <thread>
#include <unistd.h>

#include "pg_common.h"

using namespace pg;

namespace {

struct VArgs {
  int nranks, N;
  const int64_t* n;
  const double* L;
  int body_kind;
  const double* params;
  int nparams;
  double interface_value, border_value;
  int nkeys;
  const int32_t* keys;
  double dt;
  int scheme_ctor, scheme_run;
  int64_t steps;
  double* x_out;
  int64_t* n_own_out;
  int64_t* nnz_out;
  int64_t* n_ghost_out;
  int64_t* iters_out;
};

// Krylov method of the virtual-rank runs (pg_debug_set_virtual_rank_method): BiCGStab unless a test asks otherwise
int g_method = PG_METHOD_BICGSTAB, g_restart = 0;
// pg_debug_set_virtual_rank_ramp: != 0: the interface value changes from step to step, g = value (1 + ramp step) -- the rows
// alone on their diagonal then MOVE in every step and the compact loop has to carry their change to the rows coupled to
// them, across slab faces too (pg_reduce.hip diag_fix with the exchange of the deltas)
double g_ramp = 0.0;
// what the loop of each rank iterated on in the last run (pg_debug_virtual_rank_info)
int64_t g_loop_rows[8], g_loop_bytes[8], g_loop_ghosts[8], g_full_rows[8];

void check(int32_t st, const char* what) {
  if (st != 0) {
    char buf[1024];
    pg_last_error(buf, sizeof(buf));
    throw Error(std::string(what) + ": " + buf);
  }
}

void rank_main(const VArgs& a, int rank, LocalComm* lc, int device, std::string* err) {
  Context c;
  try {
    PG_HIP(hipSetDevice(device));
    c.inited = true;
    c.device = device;
    c.rank = rank;
    c.nranks = a.nranks;
    c.local = lc;
    PG_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    PG_HIP(hipStreamCreateWithFlags(&c.comm_stream, hipStreamNonBlocking));
    set_thread_context(&c);

    pg_mesh* mesh = nullptr;
    pg_capacity* cap = nullptr;
    pg_diffops* ops = nullptr;
    pg_solver* sol = nullptr;
    check(pg_mesh_create(a.N, a.n, a.L, nullptr, &mesh), "mesh");
    check(pg_capacity_create_levelset(mesh, a.body_kind, a.params, a.nparams, 0, &cap), "capacity");
    check(pg_diffops_create(cap, &ops), "diffops");
    pg_bc_desc bc{};
    bc.kind = PG_BC_DIRICHLET;
    bc.value = a.interface_value;
    std::vector<pg_border_desc> borders(a.nkeys);
    for (int k = 0; k < a.nkeys; ++k) borders[k] = pg_border_desc{a.keys[k], PG_BC_DIRICHLET, a.border_value};
    int64_t M = 1;
    for (int d = 0; d < a.N; ++d) M *= a.n[d] + 1;
    // T0 = zeros stays implicit (NULL): 2M doubles per rank would be 17 GB of host memory at 8 x 512^3
    check(pg_solver_create_unsteady_mono(cap, ops, &bc, borders.data(), a.nkeys, nullptr, nullptr, a.dt, nullptr,
                                         a.scheme_ctor, &sol), "solver");
    pg_krylov_opts o{g_method, 1e-13, 0.0, 0, 4, g_method == PG_METHOD_BICGSTAB ? 1 : 0, g_restart};
    pg_run_info info{};
    if (g_ramp == 0.0) {
      check(pg_solver_run(sol, 1e300, a.scheme_run, &o, 1, a.steps, 0, &info), "run");
    } else {
      pg_step_info sti{};
      check(pg_solver_initial_solve(sol, &o, &sti), "initial solve");
      info.total_iters = sti.iters;
      std::vector<double> g0(M), g1(M);
      for (int64_t k = 0; k < a.steps; ++k) {
        std::fill(g0.begin(), g0.end(), a.interface_value * (1.0 + g_ramp * (double)k));
        std::fill(g1.begin(), g1.end(), a.interface_value * (1.0 + g_ramp * (double)(k + 1)));
        check(pg_solver_set_interface_value(sol, g0.data(), g1.data()), "interface value");
        check(pg_solver_step(sol, a.scheme_run, &o, &sti), "step");
        info.total_iters += sti.iters;
      }
    }
    pg_system_info si{};
    check(pg_solver_system_info(sol, 1, &si), "info");
    {
      pg_system_info li{};
      check(pg_solver_system_info(sol, 1 | 4 | 2, &li), "loop info");
      g_loop_rows[rank] = li.rows_matrix;
      g_loop_bytes[rank] = li.spmv_bytes;
      g_loop_ghosts[rank] = li.n_ghost_loop;
      g_full_rows[rank] = si.n_own;
    }
    a.n_own_out[rank] = si.n_own;
    a.nnz_out[rank] = si.nnz;
    a.n_ghost_out[rank] = si.n_ghost;
    a.iters_out[rank] = info.total_iters;
    if (a.x_out) check(pg_solver_get_state(sol, -1, a.x_out, 2 * M), "state");   // writes the owned planes only
    pg_solver_destroy(sol);
    pg_diffops_destroy(ops);
    pg_capacity_destroy(cap);
    pg_mesh_destroy(mesh);
  } catch (const std::exception& e) {
    *err = e.what();
    // a failed rank must not leave the others stuck in a barrier: there is no clean recovery, report loudly
    fprintf(stderr, "[pg_debug] virtual rank %d failed: %s\n", rank, e.what());
    fflush(stderr);
    _exit(3);   // the other ranks would wait in a barrier forever
  }
  set_thread_context(nullptr);
  if (c.stream) (void)hipStreamDestroy(c.stream);
  if (c.comm_stream) (void)hipStreamDestroy(c.comm_stream);
}

}  // namespace

extern "C" int32_t pg_debug_set_virtual_rank_method(int32_t method, int32_t restart) {
  PG_API_BEGIN
  PG_REQUIRE(method == PG_METHOD_BICGSTAB || method == PG_METHOD_CG || method == PG_METHOD_GMRES, "unknown Krylov method");
  g_method = method;
  g_restart = restart;
  PG_API_END
}

extern "C" int32_t pg_debug_set_virtual_rank_ramp(double ramp) {
  PG_API_BEGIN
  g_ramp = ramp;
  PG_API_END
}

extern "C" int32_t pg_debug_virtual_rank_info(int32_t rank, int64_t* full_rows, int64_t* loop_rows, int64_t* loop_bytes,
                                              int64_t* loop_ghosts) {
  PG_API_BEGIN
  PG_REQUIRE(rank >= 0 && rank < 8, "pg_debug_virtual_rank_info: rank out of range");
  if (full_rows) *full_rows = g_full_rows[rank];
  if (loop_rows) *loop_rows = g_loop_rows[rank];
  if (loop_bytes) *loop_bytes = g_loop_bytes[rank];
  if (loop_ghosts) *loop_ghosts = g_loop_ghosts[rank];
  PG_API_END
}

extern "C" int32_t pg_debug_run_virtual_ranks(int32_t nranks, int32_t N, const int64_t* n, const double* L, int32_t body_kind,
                                              const double* params, int32_t nparams, double interface_value,
                                              double border_value, int32_t nkeys, const int32_t* keys, double dt,
                                              int32_t scheme_ctor, int32_t scheme_run, int64_t steps, double* x_out,
                                              int64_t* n_own_out, int64_t* nnz_out, int64_t* n_ghost_out,
                                              int64_t* iters_out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(nranks >= 1 && nranks <= 8, "1..8 virtual ranks");
  VArgs a{nranks, N, n, L, body_kind, params, nparams, interface_value, border_value, nkeys, keys, dt,
          scheme_ctor, scheme_run, steps, x_out, n_own_out, nnz_out, n_ghost_out, iters_out};
  std::unique_ptr<LocalComm, void (*)(LocalComm*)> lc(local_comm_create(nranks), local_comm_destroy);
  std::vector<std::string> errs(nranks);
  std::vector<std::thread> th;
  const int device = ctx().device;
  for (int r = 0; r < nranks; ++r) th.emplace_back(rank_main, std::cref(a), r, lc.get(), device, &errs[r]);
  for (auto& t : th) t.join();
  for (int r = 0; r < nranks; ++r)
    if (!errs[r].empty()) throw Error("virtual rank " + std::to_string(r) + ": " + errs[r]);
  PG_API_END
}
