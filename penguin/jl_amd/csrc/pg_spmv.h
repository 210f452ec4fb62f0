// pg_spmv.h -- shared between the Krylov driver and the SpMV kernels: device scalar slots, block reductions.
#pragma once
#include "pg_system.h"

namespace pg {

// device scalar block of one Krylov solve (KrylovWork::sc)
enum { S_RHO = 0, S_RHO_OLD, S_ALPHA, S_OMEGA, S_BETA, S_RR, S_BB, S_TOL2, S_DONE, S_ITERS, S_RELTOL2, S_ABSTOL2,
       S_RESTART, S_RHAT2, S_FORCE, S_PENDING3, S_RRW, S_HALF, S_RR0 /* initial (r,r)_W: diagnostics */,
       S_MOVED /* folded start of a compact step: a row alone on its diagonal moved although the data did not (pg_solver.hip) */,
       S_RED0, S_RED1, S_RED2, S_RED3, S_RED4, S_COUNT };

constexpr int BLOCK = 256;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// sum over the block; result valid in thread 0
__device__ inline double block_sum(double v, double* sh /*BLOCK/64*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += sh[w];
  }
  return s;
}

// ---- scalar phases of the Krylov drivers (shared: evaluated by k_finalize / k_derive or by the LAST block of the
// producing SpMV launch, see last_block_arrives) ------------------------------------------------------------------
enum { PH_NONE = -1, PH_INIT = 0, PH_BICG_1, PH_BICG_2, PH_BICG_3, PH_CG_INIT, PH_CG_1, PH_CG_2, PH_BICG_S };

// Convergence is tested in the units of x.  The loop iterates on the equilibrated system Â = B⁻¹ S A S, y = S⁻¹x, whose
// residual r̂ = B⁻¹S(b − A x) weighs every row by |a_ii|^-½: at 512^3 that is 1 for the (decoupled) Dirichlet border rows
// but 800 for the bulk rows, so ||r̂|| ≤ reltol ||b̂|| -- a norm dominated by the trivial rows -- left the temperature
// field at 2e-9 of the direct solve for reltol = 1e-12 (scripts/config4_accuracy.py).  The test is therefore
//       ||S r̂||₂ ≤ max(reltol ||S b̂||₂, abstol)
// i.e. the residual of the ROW-scaled system D⁻¹A x = D⁻¹b, whose entries are errors of x up to the O(1) factor
// ||(D⁻¹A)⁻¹||: the weighted sums (.., .)_W = Σ s_i² · · ride along with the plain ones (which the BiCG recurrences and
// the restart rule keep using); S_RR / S_BB hold the weighted values.
//
// end of a BiCGStab iteration: (r,r)_W -> convergence; plain (r,r) -> restart bookkeeping.  The reduction rides with the
// next iteration's (r̂,v) (PH_BICG_1) -- one scalar kernel and, with several ranks, one all-reduce less per iteration --
// or stands alone before the host polls (PH_BICG_3).  S_RED1 = (r,r)_W, S_RED2 = (r,r).
__device__ inline void end_of_iteration(double* sc) {
  const double rrw = sc[S_RED1], rr = sc[S_RED2];
  sc[S_PENDING3] = 0.0;
  sc[S_RR] = rrw;
  sc[S_ITERS] += 1.0;
  if (rrw <= sc[S_TOL2]) sc[S_DONE] = 1.0;
  else if (sc[S_RESTART] != 0.0) { sc[S_RHO] = rr; sc[S_RHAT2] = rr; }
}

__device__ inline void derive(int phase, double* sc) {
  const double r0 = sc[S_RED0], r1 = sc[S_RED1];
  switch (phase) {
    case PH_INIT: {
      // S_RED0 = (r0,r0), S_RED1 = (b,b)_W (warm start: r0 != b), S_RED2 = (r0,r0)_W
      const double bbw = r1, rrw = sc[S_RED2];
      sc[S_BB] = bbw; sc[S_RR] = rrw; sc[S_RHO] = r0; sc[S_RHO_OLD] = 1.0;
      sc[S_ALPHA] = 1.0; sc[S_OMEGA] = 1.0; sc[S_BETA] = 0.0; sc[S_ITERS] = 0.0;
      sc[S_RESTART] = 0.0; sc[S_RHAT2] = r0; sc[S_FORCE] = 0.0; sc[S_PENDING3] = 0.0;
      const double t2 = sc[S_RELTOL2] * bbw;
      sc[S_TOL2] = t2 > sc[S_ABSTOL2] ? t2 : sc[S_ABSTOL2];
      sc[S_DONE] = (rrw <= sc[S_TOL2]) ? 1.0 : 0.0;
      sc[S_RR0] = rrw;
      break;
    }
    case PH_CG_INIT: {
      // S_RED0 = (b,b), S_RED1 = (b,b)_W
      sc[S_BB] = r1; sc[S_RR] = r0; sc[S_RHO] = r0; sc[S_RHO_OLD] = 1.0;
      sc[S_ALPHA] = 1.0; sc[S_OMEGA] = 1.0; sc[S_BETA] = 0.0; sc[S_ITERS] = 0.0;
      sc[S_RESTART] = 0.0; sc[S_RHAT2] = r0; sc[S_FORCE] = 0.0; sc[S_PENDING3] = 0.0;
      const double t2 = sc[S_RELTOL2] * r1;
      sc[S_TOL2] = t2 > sc[S_ABSTOL2] ? t2 : sc[S_ABSTOL2];
      sc[S_DONE] = (r1 <= sc[S_TOL2]) ? 1.0 : 0.0;
      sc[S_RRW] = r1;
      break;
    }
    case PH_BICG_1:
      if (sc[S_PENDING3] != 0.0) end_of_iteration(sc);
      if (sc[S_DONE] != 0.0) break;
      // (r̂, A p) == 0: take a pure minimal-residual half step (alpha = 0) and restart afterwards
      if (r0 == 0.0) { sc[S_ALPHA] = 0.0; sc[S_FORCE] = 1.0; } else sc[S_ALPHA] = sc[S_RHO] / r0;
      break;
    case PH_BICG_2: {
      // r0 = (t,s), r1 = (t,t), RED2 = (r̂,s), RED3 = (s,s), RED4 = (r̂,t)
      const double ts = r0, tt = r1, rs = sc[S_RED2], ss = sc[S_RED3], rt = sc[S_RED4];
      const double omega = tt != 0.0 ? ts / tt : 0.0;
      const double rho_old = sc[S_RHO], rho_new = rs - omega * rt;
      double rr_pred = ss - 2.0 * omega * ts + omega * omega * tt;   // (r,r) of the coming update, restart test only
      rr_pred = rr_pred > 0.0 ? rr_pred : 0.0;
      sc[S_OMEGA] = omega;
      sc[S_RHO_OLD] = rho_old;
      sc[S_RHO] = rho_new;
      if (omega == 0.0 || sc[S_FORCE] != 0.0 || rho_new * rho_new < 1e-20 * sc[S_RHAT2] * rr_pred) {
        // (r̂,r) collapsed -- r̂ = b is often supported on a few identity rows (T⁰ = 0) and r leaves that
        // support: restart with r̂ := r (the remedy Eigen's BiCGSTAB uses, with a relative threshold:
        // cos(r̂,r) < 1e-10).  k_bicg_xrp copies r into r̂ and p; PH_BICG_3 sets ρ = (r,r).
        sc[S_FORCE] = 0.0;
        sc[S_RESTART] = 1.0;
        sc[S_BETA] = 0.0;
      } else {
        sc[S_RESTART] = 0.0;
        sc[S_BETA] = (rho_new / rho_old) * (sc[S_ALPHA] / omega);
      }
      break;
    }
    case PH_BICG_3:
      if (sc[S_PENDING3] != 0.0) end_of_iteration(sc);
      break;
    case PH_BICG_S:
      // half step: s = r - αv already meets the tolerance (S_RED4 = (s,s)_W): x += αp and stop (k_bicg_half) -- with a
      // polynomial of degree m in every application of the operator a whole iteration is 2m products, worth testing in
      // the middle
      // (S_HALF = the number of the iteration that ended this way: k_bicg_half applies x += αp in that iteration only --
      // iterations queued past it find S_DONE set and must not repeat the update)
      if (sc[S_DONE] == 0.0 && sc[S_RED4] <= sc[S_TOL2]) {
        sc[S_RR] = sc[S_RED4];
        sc[S_ITERS] += 1.0;
        sc[S_PENDING3] = 0.0;
        sc[S_DONE] = 1.0;
        sc[S_HALF] = sc[S_ITERS];
      }
      break;
    case PH_CG_1:
      if (r0 == 0.0) sc[S_DONE] = 2.0; else sc[S_ALPHA] = sc[S_RR] / r0;
      break;
    case PH_CG_2: {
      // S_RED0 = (r,r) (the recurrence's), S_RED1 = (r,r)_W (the convergence test's)
      const double rr_old = sc[S_RR];
      sc[S_RR] = r0;
      sc[S_RRW] = r1;
      sc[S_ITERS] += 1.0;
      if (r1 <= sc[S_TOL2]) sc[S_DONE] = 1.0;
      else sc[S_BETA] = r0 / rr_old;
      break;
    }
  }
}


// In-launch hand-off to the last-arriving block (cdna_hip_programming.md "In-launch split-K reduction", the write-through
// form): a block stores its few partials with agent-scope (sc1, write-through) stores -- no release fence, which would
// write back the XCD's whole dirty L2 (the y vector) once per block: measured +45 us per launch -- all its waves drain
// their stores, one lane draws a ticket; the block that draws the last ticket reads every block's partials with
// agent-scope loads.  Correct for any placement of the blocks over XCDs.  The ticket word is reset by the last arriver
// (it is zeroed once at allocation; launches that return at the done flag never touch it).  Returns the same value in
// every thread of the block.
__device__ inline void store_partial(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double load_partial(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}

// store (accum == 0) or add to (accum != 0) one partial sum
__device__ inline void put_partial(double* p, double v, int accum) { store_partial(p, accum ? v + load_partial(p) : v); }

__device__ inline bool last_block_arrives(unsigned* ticket, unsigned nblocks, double* sh /* >= 1 double of LDS */) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == nblocks - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[0] = last ? 1.0 : 0.0;
  }
  __syncthreads();
  const bool last = sh[0] != 0.0;
  __syncthreads();   // sh is reused by the block sums that follow
  return last;
}

// scalar phase folded into the producing launch: what to reduce and derive once the last block has arrived
struct FinArgs {
  unsigned* ticket;   // nullptr: no in-launch phase
  double* sc;         // scalar block (read-write)
  int phase;          // PH_*
  int nslots;         // partial slots [0, nslots) to sum into S_RED0..
  int do_derive;      // 0: sums only (several ranks: an all-reduce follows)
  const double* dotx; // operand of the (y, .) dot of modes 2 / 3; nullptr: the input vector x itself
  double pc0 = 2.0, pc1 = -1.0;   // modes 4..7: pc0 x + pc1 A x  (default: the Neumann product 2x - Âx)
  const double* base = nullptr;   // mode 5: y = base - (pc0 x + pc1 A x);  mode 8: y = pc2 base + pc0 x + pc1 A x
  double* accv = nullptr;         // mode 7: accv += pc2 x
  double pc2 = 0.0;
};

// executed by every block at the end of a producing launch (after its partials are stored); nparts = partial sums per
// slot (the slot stride; == gridDim.x unless the launch completes an earlier, larger one)
__device__ inline void fold_scalar_phase(const FinArgs& fin, const double* __restrict__ partials, double* s_red, int nparts) {
  if (!fin.ticket) return;
  if (!last_block_arrives(fin.ticket, gridDim.x, s_red)) return;
  for (int sl = 0; sl < fin.nslots; ++sl) {
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += BLOCK) a += load_partial(partials + (size_t)sl * nparts + i);
    const double t = block_sum(a, s_red);
    if (threadIdx.x == 0) fin.sc[S_RED0 + sl] = t;
  }
  if (threadIdx.x == 0 && fin.do_derive) derive(fin.phase, fin.sc);
}

// y = A x on rows [0, A.n).  mode 0: plain; 1: partials[0..grid) = aux . y; 2: partials[0..grid) = y . x and
// partials[grid..2grid) = y . y; 3: mode 2 plus partials[4grid..5grid) = aux . y (the operand of the (y, .) dot of modes
// 2 / 3 is fin->dotx when set); 4 (slice kernel only): y = 2x - A x, i.e. u = M⁻¹x for the Neumann preconditioner
// M⁻¹ = 2I - Â of the BiCGStab driver, no dots.  `sc` (may be NULL): kernels return immediately when sc[S_DONE] != 0.
// `grid` must be the value used to size `partials` (KrylovWork::grid) for modes 1/2.
// `fin` (optional): scalar phase evaluated by the last block of the launch; returns true when the launched kernel
// does that (the stencil-slice kernel), false when the caller still has to launch the scalar kernel itself.
bool launch_spmv(int mode, const CsrMatrix& A, const double* x, double* y, const double* aux, double* partials,
                 const double* sc, int grid, hipStream_t st, const FinArgs* fin = nullptr);
// y = A x on a slab whose x needs a halo exchange first (C1 of SURVEY.md 2.3): the exchange of x's ghost segments runs on
// the communication stream while the slices that reference no ghost column are multiplied; the slices that do (rows of
// the first / last owned plane) follow in a second, small launch once the halo has landed, adding their partial sums
// to the first launch's.  Falls back to exchange-then-multiply for the CSR kernels or when nothing can be split off.
// Same arguments and return value as launch_spmv; x is non-const because its ghost segments are received into.
bool spmv_with_halo(int mode, const CsrMatrix& A, const Numbering& nb, const Slab& slab, double* x, double* y, const double* aux,
                    double* partials, const double* sc, int grid, hipStream_t st, const FinArgs* fin = nullptr);
// plain y = A x with an explicit kernel variant (PG_SPMV_VARIANT numbering): kernel-vs-kernel parity checks
void launch_spmv_variant(int variant, const CsrMatrix& A, const double* x, double* y, hipStream_t st);
int spmv_default_grid(i64 n);
bool spmv_supports_preconditioner_product();   // mode 4 and FinArgs::dotx exist in the slice kernel only (PG_SPMV_VARIANT)

}  // namespace pg
