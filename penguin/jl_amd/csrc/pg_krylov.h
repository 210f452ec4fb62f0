// pg_krylov.h -- K11: Krylov solve of the reduced system on one slab (BiCGStab / CG / GMRES).
#pragma once
#include <functional>

#include "pg_system.h"

namespace pg {

struct XGuess {
  const double* zbase = nullptr;
  const double* zr[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const double* coef = nullptr;
};

struct KrylovWork {
  i64 n = 0, nvec = 0;
  DevBuf<double> r, rhat, p, v, t;      // n_vec each (p and r -- which holds s between the two SpMVs -- carry ghosts)
  DevBuf<double> partials;              // per-block partial sums, 5 slots x grid
  DevBuf<double> sc;                    // device scalars (see pg_krylov.hip)
  double* h_sc = nullptr;               // pinned host mirror of sc
  int grid = 1;
  DevBuf<unsigned> ticket;              // arrival counter of the in-launch scalar phases (pg_spmv.h)
  double last_rate2 = 0.0;              // what one product took off log (r,r)_W in the previous polynomial solve (0: unknown)
  int last_iters = 0;                   // iterations of the previous solve (sizes the first launch batch)
  // degree of the preconditioner polynomial chosen from the previous solve on the same matrix (auto mode, pg_krylov.hip):
  // adapt_m products per application, adapt_h applications expected
  const void* adapt_matrix = nullptr;
  int adapt_m = 0, adapt_h = 0;
  const void* need_matrix = nullptr;    // the last estimates of the products a solve on that matrix needs (newest first)
  double need_hist[3] = {0.0, 0.0, 0.0};
  const void* fail_matrix = nullptr;    // x-space, one application per solve: the last degree that fell short (with the
  int fail_deg = 0, good_deg = 0;       // log of start residual / tolerance it fell short at) and the last that sufficed
  double fail_L = 0.0;
  int fail_wait = 0, fail_backoff = 1;  // solves left before that degree may be tried again; doubles when it falls short again
  // set by the caller around one krylov_solve: the system is a compact image of the caller's (pg_reduce.hip, DiagElim) and
  // x is the caller's FULL vector -- the solution update x += q(Â)y lands at x[scatter[i]].  Needs the polynomial path.
  const int* scatter = nullptr;
  // set by the caller of a compact solve whose start was extrapolated from older states (pg_solver.hip, GuessArgs) and who left
  // the extrapolated state unformed: the FIRST update of x then writes  x[map] = z_g[map] + α M⁻¹p,  z_g = zbase + Σ c_j (zr[o_j] -
  // zbase)  (coef: [0..3] c_j, [4] how many, [5..8] which of zr), instead of adding to x.  Reset by krylov_solve.
  XGuess xguess;
  // set by the caller of a prepared start that wrote r̂ only: p = r = r̂ at the start, so the first iteration reads r̂
  // wherever it needs p or r (k_bicg_s writes r = s, k_bicg_xrp writes p, as always) and the start kernel writes two
  // vectors less.  Reset by krylov_solve.
  bool p_in_rhat = false;
  // set by the caller whose start kernel has already reset the scalars, summed the start sums and derived PH_INIT
  // (k_rhs_init_c with a ticket): krylov_solve launches no start kernel.  Reset by krylov_solve.
  bool start_folded = false;
  // polynomial right preconditioner (pg_krylov.hip), n_vec each, on first use: the accumulated solution of the
  // preconditioned system (x = x0 + q(Â) ya) and the two work vectors the chain of products alternates between
  DevBuf<double> ya, wa, wb;
  DevBuf<double> yb;        // x-space form: ya = M⁻¹p, yb = M⁻¹s of the running iteration
  // GMRES(m) only, allocated on first use (pg_gmres.hip): m+1 basis vectors, H / rotations / g, per-block partial sums
  DevBuf<double> gm_basis, gm, gm_partials;
  int gm_m = -1;
  // Set by the caller around one krylov_solve: work to queue right AFTER the first batch of iterations and the copy of the
  // scalars, BEFORE the host waits for that copy -- the next time step's first product, speculatively: the device then
  // has something to run while the host wakes up, reads the scalars and queues the next step (the wait is on an event
  // recorded behind the copy, not on the stream).  Called at most once per solve; reset by krylov_solve.
  std::function<void()> after_first_batch;
  hipEvent_t ev_poll = nullptr;
  void init(i64 n_own, i64 n_vec);
  ~KrylovWork();
};

struct SolveStats {
  int iters = 0;
  int converged = 0;
  double resnorm = 0.0, bnorm = 0.0;
  // profiling: HIP-event time of the sampled SpMV launches, by kind -- launches that carry fused dots (and their operand
  // vectors) and lean ones (plain products and the factors of the preconditioner polynomial: x, y and the matrix only)
  double spmv_ms = 0.0, spmv_lean_ms = 0.0;
  i64 spmv_launches = 0, spmv_lean_launches = 0;
  int poly_degree = 0;      // products with Â per application of the preconditioned operator (0: plain iteration)
  int half_exit = 0;        // 1: the solve ended at the half step of its last iteration (counted as an iteration)
  int poly_xspace = 0;      // 1: x-space form of the preconditioned loop (Horner chains, no recovery), pg_krylov.hip
  i64 products = 0;         // products with Â inside the solve (the caller's start product not counted)
  int polls = 0;            // host waits of the solve (1: it ended inside the first batch of queued launches)
};

// halo exchange of the ghost segments of `vec` (no-op on one rank)
void halo_exchange(const Numbering& nb, const Slab& slab, double* vec, hipStream_t st);
// the same exchange, split: halo_begin forks it off `st` onto the communication stream (everything enqueued on `st` so
// far is waited for there), halo_end makes `st` wait for its completion; work enqueued on `st` in between overlaps it.
// (Virtual ranks exchange synchronously inside halo_begin.)
void halo_begin(const Numbering& nb, const Slab& slab, double* vec, hipStream_t st);
void halo_end(hipStream_t st);
// y[0..n) = A * x   (x must hold valid ghosts)
void spmv(const CsrMatrix& A, const double* x, double* y, hipStream_t st);
// halo exchange of x (when the matrix needs one) overlapped with y[0..n) = A * x
void spmv_halo(const CsrMatrix& A, const Numbering& nb, const Slab& slab, double* x, double* y, hipStream_t st);
// x (n_vec, overwritten) = A^{-1} b.  x0 == nullptr: zero initial guess; else start from x0 with Ax0 = A*x0 given
// (BiCGStab only).  x must not alias x0.
// preinit (BiCGStab only): the caller has already written x, w.r = w.rhat = w.p = r0 and the partial sums of (r0,r0)
// / (b,b) in slots 0 / 1 of w.partials with a w.grid-block launch (pg_solver.hip fuses that with the right-hand side).
void krylov_solve(const CsrMatrix& A, const Numbering& nb, const Slab& slab, const double* b, double* x,
                  KrylovWork& w, const pg_krylov_opts& opts, SolveStats& stats, const double* x0 = nullptr,
                  const double* Ax0 = nullptr, bool preinit = false);

// will krylov_solve(A, ..., opts) run BiCGStab right-preconditioned with the polynomial (the y-space iteration)?
bool krylov_uses_polynomial(const CsrMatrix& A, const pg_krylov_opts& opts);

// restarted GMRES (pg_gmres.hip): zero initial guess, x (n_vec, overwritten) = A^{-1} b
void gmres_solve(const CsrMatrix& A, const Numbering& nb, const Slab& slab, const double* b, double* x, KrylovWork& w,
                 const pg_krylov_opts& opts, SolveStats& stats);

}  // namespace pg
