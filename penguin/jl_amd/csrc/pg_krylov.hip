// pg_krylov.hip -- K11 of SURVEY.md section 2.3: the Krylov inner loop of solve_system! (src/solver.jl:158-188)
//
//   reference (IterativeSolvers 0.9.4, single thread)          here
//   CSC mul!                                                   pg_spmv.hip (stencil slices; dots fused into the launch)
//   dot / axpy! / norm (OpenBLAS BLAS-1)                       two fused vector kernels per BiCGStab iteration
//   method(A_reduced, b_reduced; kwargs...)                    device-resident BiCGStab / CG: all scalars stay on the
//                                                              GPU; the host polls a done flag between batches of
//                                                              queued iterations (no per-iteration sync)
//
// One BiCGStab iteration = 4 launches on one rank:
//   SpMV (v = Â p, (r̂,v); its last block evaluates: previous iteration's (r,r) -> convergence / restart, then α)
//   k_bicg_s   (s = r - αv over r; (r̂,s), (s,s))
//   SpMV (t = Â s, (t,s), (t,t), (r̂,t); last block: ω, ρ' = (r̂,s) - ω(r̂,t), β / restart decision)
//   k_bicg_xrp (x += αp + ωs; r = s - ωt; p = r + β(p - ωv); (r,r))
// With several ranks the SpMV's last block leaves the local sums, an RCCL all-reduce and k_derive follow.
//
// Polynomial right preconditioner (default where admissible, CsrMatrix::poly_ok; opts.precond = -1 or PG_POLY=0 turns it
// off).  The equilibrated systems of the time loop have their spectrum inside [1 - g, 1 + g], g = the Gershgorin radius
// pg_precond.hip computes (0.72 for the benchmark's Crank-Nicolson matrix).  With λ_1..λ_m the Chebyshev nodes of that
// interval, R(Â) = Π_k (I - Â/λ_k) is the degree-m residual polynomial of smallest maximum on it (0.29, 0.11, 0.043,
// 0.016 for m = 2, 3, 4, 5 at g = 0.72), M⁻¹ = q(Â) = Â⁻¹(I - R(Â)) a polynomial approximation of Â⁻¹, and BiCGStab runs on
//     C y = b̂ - Â x0,   C = Â M⁻¹ = I - R(Â),   x = x0 + q(Â) y
// whose spectrum lies within that maximum of 1: iterations fall like 1/m while the number of products with Â stays about
// the same -- and all but the last product of an application of C are LEAN launches (mode 4: w <- w - Âw/λ_k, one vector
// in, one out, no dots, no BLAS-1 pass), the last one closes the chain (v = p - w_m, t = s - w_m) and carries the dots.
// The vector passes, the reductions and, with several ranks, the all-reduces of a time step fall with the iteration
// count.  R is applied in product form (Richardson steps), roots taken alternately from both ends of the interval so
// that no partial product grows; y is accumulated in place of x and x recovered once per solve with the same chain
// (mode 7 accumulates Σ_k w_(k-1)/λ_k = q(Â) y on the way).  The residual BiCGStab sees is the true residual of Âx = b̂,
// the stopping test is unchanged.  m = 2 with both roots at 1 is the Neumann preconditioner 2I - Â of round 1.
#include "pg_krylov.h"
#include "pg_spmv.h"

using namespace pg;

namespace pg { extern double g_host_wait_us; }

namespace {


// ---- fused vector kernels ---------------------------------------------------------------------------
// x0 == nullptr: zero initial guess (IterativeSolvers' default).  Otherwise x = x0 and r = b - Ax0 (warm start with
// the previous time level; Ax0 is the SpMV the CN right-hand side needs anyway).  partials: slot 0 = r.r, slot 1 =
// (b,b)_W, slot 2 = (r,r)_W with the weights ds² of the convergence test (pg_spmv.h; ds == nullptr: unweighted)
__global__ __launch_bounds__(BLOCK) void k_bicg_init(i64 n, i64 nvec, const double* __restrict__ b,
                                                     const double* __restrict__ x0, const double* __restrict__ Ax0,
                                                     double* __restrict__ x, double* __restrict__ r,
                                                     double* __restrict__ rhat, double* __restrict__ p,
                                                     double* __restrict__ v, double* __restrict__ partials,
                                                     const double* __restrict__ ds) {
  __shared__ double s_red[BLOCK / 64];
  double acc = 0.0, accb = 0.0, accw = 0.0;
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < nvec; i += (i64)gridDim.x * BLOCK) {
    if (i < n) {
      const double bi = b[i];
      const double ri = x0 ? bi - Ax0[i] : bi;
      const double d = ds ? ds[i] : 1.0;
      x[i] = x0 ? x0[i] : 0.0;
      r[i] = ri; rhat[i] = ri; p[i] = ri; v[i] = 0.0;   // p₀ = r₀ (β = 0)
      acc += ri * ri;
      accb += (d * bi) * (d * bi);
      accw += (d * ri) * (d * ri);
    } else {
      x[i] = 0.0; p[i] = 0.0;
    }
  }
  const double t = block_sum(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  const double tb = block_sum(accb, s_red);
  if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = tb;
  const double tw = block_sum(accw, s_red);
  if (threadIdx.x == 0) partials[2 * (size_t)gridDim.x + blockIdx.x] = tw;
}

// s = r - αv written over r (r is not needed again: r_new = s - ωt); partials slots 2, 3 = (r̂,s), (s,s): they are
// summed together with the dots of the SpMV that follows (one scalar kernel instead of two)
// NTV: stream hints on the vectors that are dead after this kernel (v here; x, t, v in k_bicg_xrp), so that they do not
// push the SpMV's matrix data (records, packed irregular rows: ~90 MB) out of the 256 MB Infinity Cache between launches
// slot 4 = (s,s)_W, the weighted norm of the half-step convergence test (PH_BICG_S; the second product overwrites the slot
// afterwards)
template <bool NTV>
__global__ __launch_bounds__(BLOCK) void k_bicg_s(i64 n, double* __restrict__ sc, const double* __restrict__ v,
                                                  const double* __restrict__ rhat, double* __restrict__ r,
                                                  double* __restrict__ partials, const double* __restrict__ ds,
                                                  unsigned* __restrict__ ticket, int r_in_rhat) {
  __shared__ double s_red[BLOCK / 64];
  if (sc[S_DONE] != 0.0) return;
  const double alpha = sc[S_ALPHA];
  const bool rrhat = r_in_rhat != 0 && sc[S_ITERS] == 0.0;   // first iteration of a start that left r = r̂ unwritten
  double a0 = 0.0, a1 = 0.0, aw = 0.0;
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const double rh = rhat[i];
    const double si = (rrhat ? rh : r[i]) - alpha * (NTV ? __builtin_nontemporal_load(v + i) : v[i]);
    r[i] = si;
    a0 += rh * si;
    a1 += si * si;
    const double ws = ds[i] * si;
    aw += ws * ws;
  }
  const double t0 = block_sum(a0, s_red);
  if (threadIdx.x == 0) partials[2 * (size_t)gridDim.x + blockIdx.x] = t0;
  const double t1 = block_sum(a1, s_red);
  if (threadIdx.x == 0) partials[3 * (size_t)gridDim.x + blockIdx.x] = t1;
  const double tw = block_sum(aw, s_red);
  if (!ticket) {
    if (threadIdx.x == 0) partials[4 * (size_t)gridDim.x + blockIdx.x] = tw;
    return;
  }
  // the half-step test (PH_BICG_S) inside this launch: the last block to arrive sums slot 4 in k_finalize's order and
  // derives (one rank: no all-reduce in between) -- a scalar kernel and its gap less per half step
  double* slot4 = partials + 4 * (size_t)gridDim.x;
  if (threadIdx.x == 0) store_partial(slot4 + blockIdx.x, tw);
  if (!last_block_arrives(ticket, gridDim.x, s_red)) return;
  double a = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += BLOCK) a += load_partial(slot4 + i);
  const double t = block_sum(a, s_red);
  if (threadIdx.x == 0) {
    sc[S_RED0 + 4] = t;
    derive(PH_BICG_S, sc);
  }
}

// k_bicg_s and the x update of the first half in ONE pass (x-space form of the preconditioned loop): s = r - αv over r with its
// dots and the half-step test, and x += α M⁻¹p through the index map of a compact system.  BiCGStab adds α M⁻¹p to x in every
// iteration whether or not it ends at the half step (x_{i+1} = x_i + α p̂ + ω ŝ, and nothing reads x in between), so the
// update does not wait for the verdict of the test: k_bicg_half -- a second pass over p̂ and x, and a launch -- is not
// needed, and k_bicg_xrp of a solve that goes on adds ω ŝ only (`alpha_done`).  Two rows per lane (16-byte accesses).
template <bool NTV>
__global__ __launch_bounds__(BLOCK) void k_bicg_s_x(i64 n, double* __restrict__ sc, const double* __restrict__ v,
                                                    const double* __restrict__ rhat, double* __restrict__ r,
                                                    double* __restrict__ partials, const double* __restrict__ ds,
                                                    unsigned* __restrict__ ticket, int r_in_rhat, const double* __restrict__ phat,
                                                    double* x, const int* __restrict__ map, XGuess xg, int first_launch) {
  __shared__ double s_red[BLOCK / 64];
  const bool done = sc[S_DONE] != 0.0;
  if (done && !(xg.zbase != nullptr && first_launch != 0)) return;
  const double alpha = sc[S_ALPHA];
  const bool rrhat = r_in_rhat != 0 && sc[S_ITERS] == 0.0;   // first iteration of a start that left r = r̂ unwritten
  const double* __restrict__ rsrc = rrhat ? rhat : r;
  // first update of an extrapolated start whose state was left unformed (XGuess): x is written, not added to
  const bool fresh = xg.zbase != nullptr && rrhat && map != nullptr;
  int ku = 0;
  double cj[4] = {0.0, 0.0, 0.0, 0.0};
  const double* zo[4] = {xg.zbase, xg.zbase, xg.zbase, xg.zbase};
  if (fresh) {
    ku = min(4, (int)xg.coef[4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < ku) {
        const int sel = (int)xg.coef[5 + j];
        cj[j] = xg.coef[j];
        const double* p = xg.zr[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) p = sel == q ? xg.zr[q] : p;
        zo[j] = p;
      }
    }
  }
  auto base_of = [&](int j) {
    const double b = xg.zbase[j];
    double g = b;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < ku) g += cj[q] * (zo[q][j] - b);
    return g;
  };
  if (done) {
    // the start met the tolerance: no update will run, and this -- the first s kernel queued for the solve, ahead of whatever
    // the caller queues behind the first batch -- is where the extrapolated state of the loop's rows gets written
    if (fresh)
      for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) { const int j = map[i]; x[j] = base_of(j); }
    return;
  }
  double a0 = 0.0, a1 = 0.0, aw = 0.0;
  typedef double dd2 __attribute__((ext_vector_type(2)));
  const i64 npair = n / 2;
  for (i64 q = blockIdx.x * (i64)BLOCK + threadIdx.x; q < npair; q += (i64)gridDim.x * BLOCK) {
    const i64 i = 2 * q;
    const dd2 rh = *reinterpret_cast<const dd2*>(rhat + i);
    const dd2 rr = rrhat ? rh : *reinterpret_cast<const dd2*>(rsrc + i);
    const dd2 vv = NTV ? __builtin_nontemporal_load(reinterpret_cast<const dd2*>(v + i)) : *reinterpret_cast<const dd2*>(v + i);
    const dd2 dw = *reinterpret_cast<const dd2*>(ds + i);
    const dd2 ph = NTV ? __builtin_nontemporal_load(reinterpret_cast<const dd2*>(phat + i)) : *reinterpret_cast<const dd2*>(phat + i);
    dd2 si;
    si.x = rr.x - alpha * vv.x;
    si.y = rr.y - alpha * vv.y;
    *reinterpret_cast<dd2*>(r + i) = si;
    if (map) {
      const int j0 = map[i], j1 = map[i + 1];
      if (fresh) {
        const double g0 = base_of(j0), g1 = base_of(j1);
        x[j0] = g0 + alpha * ph.x;
        x[j1] = g1 + alpha * ph.y;
      } else {
        x[j0] += alpha * ph.x;
        x[j1] += alpha * ph.y;
      }
    } else {
      dd2 xx = *reinterpret_cast<dd2*>(x + i);
      xx.x += alpha * ph.x;
      xx.y += alpha * ph.y;
      *reinterpret_cast<dd2*>(x + i) = xx;
    }
    a0 += rh.x * si.x + rh.y * si.y;
    a1 += si.x * si.x + si.y * si.y;
    const double w0 = dw.x * si.x, w1 = dw.y * si.y;
    aw += w0 * w0 + w1 * w1;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // the odd last row
    const i64 i = n - 1;
    const double rh = rhat[i], si = rsrc[i] - alpha * v[i];
    r[i] = si;
    if (fresh) x[map[i]] = base_of(map[i]) + alpha * phat[i];
    else x[map ? map[i] : i] += alpha * phat[i];
    a0 += rh * si;
    a1 += si * si;
    aw += (ds[i] * si) * (ds[i] * si);
  }
  const double t0 = block_sum(a0, s_red);
  if (threadIdx.x == 0) partials[2 * (size_t)gridDim.x + blockIdx.x] = t0;
  const double t1 = block_sum(a1, s_red);
  if (threadIdx.x == 0) partials[3 * (size_t)gridDim.x + blockIdx.x] = t1;
  const double tw = block_sum(aw, s_red);
  if (!ticket) {
    if (threadIdx.x == 0) partials[4 * (size_t)gridDim.x + blockIdx.x] = tw;
    return;
  }
  double* slot4 = partials + 4 * (size_t)gridDim.x;
  if (threadIdx.x == 0) store_partial(slot4 + blockIdx.x, tw);
  if (!last_block_arrives(ticket, gridDim.x, s_red)) return;
  double a = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += BLOCK) a += load_partial(slot4 + i);
  const double t = block_sum(a, s_red);
  if (threadIdx.x == 0) {
    sc[S_RED0 + 4] = t;
    derive(PH_BICG_S, sc);
  }
}

// the half step accepted (PH_BICG_S: derive has set S_DONE and S_HALF = this iteration's number): x += αp; r = s stands
// fresh != 0: x is the accumulated solution of the preconditioned system, which starts at zero and has not been written yet
// in the first iteration -- assigned instead of read (its memset and its first read are saved)
// (x-space form of the preconditioned loop: p = M⁻¹p of this iteration, x the solution itself -- the caller's full vector
// through `map` when the system is a compact image of it)
__global__ __launch_bounds__(BLOCK) void k_bicg_half(i64 n, const double* __restrict__ sc, const double* __restrict__ p,
                                                     double* __restrict__ x, int fresh, int iteration,
                                                     const int* __restrict__ map) {
  if (sc[S_HALF] != (double)iteration) return;
  const double alpha = sc[S_ALPHA];
  const bool first = fresh != 0 && iteration == 1;
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const i64 j = map ? map[i] : i;
    x[j] = (first ? 0.0 : x[j]) + alpha * p[i];
  }
}

// x += αp + ωs;  r = s - ωt;  p = r + β(p - ωv)  (restart: p = r̂ = r);  partial slot 1 = (r,r)_W (convergence, weights
// ds²: pg_spmv.h), slot 2 = (r,r) (restart bookkeeping).
// β is known before r exists because ρ_new = (r̂,r) = (r̂,s) - ω(r̂,t) comes out of the dots of k_bicg_s and of the
// second SpMV: the classical p-update kernel (4 vector passes) and one scalar kernel per iteration disappear.
// (with the polynomial preconditioner x is the accumulated solution y of the preconditioned system)
template <bool NTV>
__global__ __launch_bounds__(BLOCK) void k_bicg_xrp(i64 n, double* sc, const double* __restrict__ t,
                                                    const double* __restrict__ v, double* __restrict__ x,
                                                    double* __restrict__ r, double* __restrict__ p,
                                                    double* __restrict__ rhat, double* __restrict__ partials,
                                                    const double* __restrict__ ds, int fresh, int p_in_rhat,
                                                    const double* __restrict__ phat, const double* __restrict__ shat,
                                                    const int* __restrict__ map, int alpha_done) {
  __shared__ double s_red[BLOCK / 64];
  if (sc[S_DONE] != 0.0) return;
  const double alpha = sc[S_ALPHA], omega = sc[S_OMEGA], beta = sc[S_BETA];
  const bool restart = sc[S_RESTART] != 0.0;
  const bool first = fresh != 0 && sc[S_ITERS] == 0.0;   // (see k_bicg_half)
  const bool prhat = p_in_rhat != 0 && sc[S_ITERS] == 0.0;   // first iteration of a start that left p = r̂ unwritten
  double a0 = 0.0, aw = 0.0;
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const double si = r[i], pi = prhat ? rhat[i] : p[i];
    if (phat) {
      // x-space form: x += α M⁻¹p + ω M⁻¹s with the two preconditioned vectors the applications of the operator left
      const i64 j = map ? map[i] : i;
      // (alpha_done: the first half has added α M⁻¹p already, k_bicg_s_x)
      const double ap = alpha_done ? 0.0 : alpha * (NTV ? __builtin_nontemporal_load(phat + i) : phat[i]);
      x[j] += ap + omega * (NTV ? __builtin_nontemporal_load(shat + i) : shat[i]);
    } else {
      const double xi = (first ? 0.0 : (NTV ? __builtin_nontemporal_load(x + i) : x[i])) + alpha * pi + omega * si;
      if (NTV) __builtin_nontemporal_store(xi, x + i); else x[i] = xi;
    }
    const double ri = si - omega * (NTV ? __builtin_nontemporal_load(t + i) : t[i]);
    r[i] = ri;
    if (restart) {
      p[i] = ri;
      rhat[i] = ri;
    } else {
      p[i] = ri + beta * (pi - omega * (NTV ? __builtin_nontemporal_load(v + i) : v[i]));
    }
    a0 += ri * ri;
    const double wr = ds[i] * ri;
    aw += wr * wr;
  }
  // slots 1, 2: summed together with the next SpMV's (r̂,v) in slot 0 (or alone, before a host poll)
  const double tw = block_sum(aw, s_red);
  if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = tw;
  const double t0 = block_sum(a0, s_red);
  if (threadIdx.x == 0) partials[2 * (size_t)gridDim.x + blockIdx.x] = t0;
  if (blockIdx.x == 0 && threadIdx.x == 0) sc[S_PENDING3] = 1.0;   // (only the scalar kernels read it)
}

__global__ __launch_bounds__(BLOCK) void k_cg_init(i64 n, i64 nvec, const double* __restrict__ b, double* __restrict__ x,
                                                   double* __restrict__ r, double* __restrict__ p,
                                                   double* __restrict__ partials, const double* __restrict__ ds) {
  __shared__ double s_red[BLOCK / 64];
  double acc = 0.0, accw = 0.0;
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < nvec; i += (i64)gridDim.x * BLOCK) {
    if (i < n) {
      const double bi = b[i];
      x[i] = 0.0; r[i] = bi; p[i] = bi;
      acc += bi * bi;
      accw += (ds[i] * bi) * (ds[i] * bi);
    } else {
      x[i] = 0.0; p[i] = 0.0;
    }
  }
  const double t = block_sum(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  const double tw = block_sum(accw, s_red);
  if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = tw;
}

__global__ __launch_bounds__(BLOCK) void k_cg_xr(i64 n, const double* __restrict__ sc, const double* __restrict__ p,
                                                 const double* __restrict__ q, double* __restrict__ x,
                                                 double* __restrict__ r, double* __restrict__ partials,
                                                 const double* __restrict__ ds) {
  __shared__ double s_red[BLOCK / 64];
  if (sc[S_DONE] != 0.0) return;
  const double alpha = sc[S_ALPHA];
  double a0 = 0.0, aw = 0.0;
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    a0 += ri * ri;
    aw += (ds[i] * ri) * (ds[i] * ri);
  }
  const double t0 = block_sum(a0, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  const double tw = block_sum(aw, s_red);
  if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = tw;
}

__global__ __launch_bounds__(BLOCK) void k_cg_p(i64 n, const double* __restrict__ sc, const double* __restrict__ r,
                                                double* __restrict__ p) {
  if (sc[S_DONE] != 0.0) return;
  const double beta = sc[S_BETA];
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) p[i] = r[i] + beta * p[i];
}

// x += a w  (last term of x = x0 + q(Â) y)
__global__ __launch_bounds__(BLOCK) void k_axpy_scatter(i64 n, double a, const double* __restrict__ w, double* __restrict__ x,
                                                        const int* __restrict__ map) {
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) x[map[i]] += a * w[i];
}

__global__ __launch_bounds__(BLOCK) void k_axpy(i64 n, double a, const double* __restrict__ w, double* __restrict__ x) {
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) x[i] += a * w[i];
}

// ---- scalar phase: sum block partials (deterministic order), then derive the iteration scalars (pg_spmv.h) ----
__global__ __launch_bounds__(BLOCK) void k_finalize(int phase, int slot0, int nslots, int grid,
                                                    const double* __restrict__ partials, double* __restrict__ sc,
                                                    int do_derive, int check_done) {
  __shared__ double s_red[BLOCK / 64];
  if (check_done && sc[S_DONE] != 0.0) return;
  for (int s = slot0; s < slot0 + nslots; ++s) {
    double a = 0.0;
    for (int i = threadIdx.x; i < grid; i += BLOCK) a += partials[(size_t)s * grid + i];
    const double t = block_sum(a, s_red);
    if (threadIdx.x == 0) sc[S_RED0 + s] = t;
  }
  if (do_derive && threadIdx.x == 0) derive(phase, sc);
}

// fresh scalar block of a solve: everything zero but the squared tolerances
__global__ void k_sc_reset(double* sc, double reltol2, double abstol2) {
  const int i = threadIdx.x;
  if (i < S_COUNT) sc[i] = i == S_RELTOL2 ? reltol2 : (i == S_ABSTOL2 ? abstol2 : 0.0);
}

// k_sc_reset + the start phase's k_finalize in one launch (one rank, BiCGStab with a prepared start: nothing reads the
// scalars in between)
__global__ __launch_bounds__(BLOCK) void k_start(int phase, int nslots, int grid, const double* __restrict__ partials,
                                                 double* __restrict__ sc, double reltol2, double abstol2) {
  __shared__ double s_red[BLOCK / 64];
  if (threadIdx.x < S_COUNT) sc[threadIdx.x] = threadIdx.x == S_RELTOL2 ? reltol2 : (threadIdx.x == S_ABSTOL2 ? abstol2 : 0.0);
  __syncthreads();
  for (int s = 0; s < nslots; ++s) {
    double a = 0.0;
    for (int i = threadIdx.x; i < grid; i += BLOCK) a += partials[(size_t)s * grid + i];
    const double t = block_sum(a, s_red);
    if (threadIdx.x == 0) sc[S_RED0 + s] = t;
  }
  if (threadIdx.x == 0) derive(phase, sc);
}

__global__ void k_derive(int phase, double* sc, int check_done) {
  if (check_done && sc[S_DONE] != 0.0) return;
  derive(phase, sc);
}

// after a launch whose last block already summed (and, on one rank, derived) the phase: only what is left to do
void finalize_folded(int phase, int nslots, KrylovWork& w, hipStream_t st) {
  Context& cx = ctx();
  if (cx.nranks == 1 && !cx.comm) return;
  comm_allreduce_sum_f64(w.sc.p + S_RED0, nslots, st);
  hipLaunchKernelGGL(k_derive, dim3(1), dim3(1), 0, st, phase, w.sc.p, 1);
}

void finalize(int phase, int nslots, KrylovWork& w, hipStream_t st, bool check_done, int slot0 = 0) {
  Context& cx = ctx();
  if (cx.nranks == 1 && !cx.comm) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, st, phase, slot0, nslots, w.grid, w.partials.p, w.sc.p, 1,
                       check_done ? 1 : 0);
  } else {
    // local sums -> RCCL all-reduce of <= 5 doubles over xGMI -> identical scalars on every rank
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, st, phase, slot0, nslots, w.grid, w.partials.p, w.sc.p, 0,
                       check_done ? 1 : 0);
    comm_allreduce_sum_f64(w.sc.p + S_RED0 + slot0, nslots, st);
    hipLaunchKernelGGL(k_derive, dim3(1), dim3(1), 0, st, phase, w.sc.p, check_done ? 1 : 0);
  }
}

struct SpmvTimer {
  bool on;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
  std::vector<int> iter_of;   // Krylov iteration each launch belongs to
  std::vector<char> lean_of;  // 1: lean launch (no dots), 0: launch with fused dots
  std::vector<char> second_of;   // 1: launch of the second half of its iteration (skipped when the half step was accepted)
  std::vector<int> count_of;     // launches between the two events (a chain of lean launches is bracketed as a whole)
  int cur = 0, cur_count = 1;
  bool cur_lean = false, cur_second = false;
  bool armed = false;
  explicit SpmvTimer(bool on_) : on(on_) {}
  // events are recycled through a per-thread pool: nothing is created or destroyed inside the time loop after the
  // first profiled solve
  static std::vector<hipEvent_t>& pool() {
    static thread_local std::vector<hipEvent_t> p;
    return p;
  }
  static hipEvent_t take() {
    auto& p = pool();
    if (!p.empty()) {
      hipEvent_t e = p.back();
      p.pop_back();
      return e;
    }
    hipEvent_t e;
    PG_HIP(hipEventCreate(&e));
    return e;
  }
  void begin(hipStream_t st, int iteration, bool lean = false, bool second = false, int count = 1) {
    cur = iteration;
    cur_lean = lean;
    cur_second = second;
    cur_count = count;
    // every `sample`-th bracket is timed (PG_PROFILE_SAMPLE, default 3 -- coprime with the 4 brackets of an iteration: the
    // chain of lean launches and the closing launch of each half, so all four are sampled in turn).  A pair of event
    // records costs the stream ~10 µs (the kernel behind a record starts late): a lean chain is bracketed as a WHOLE so
    // that this cost is spread over its m - 1 launches; rocprofv3's per-kernel durations (profiles/) are the cross-check
    const int sample = config().profile_sample;
    static thread_local unsigned long long counter = 0;
    armed = on && (counter++ % sample == 0);
    if (!armed) return;
    e0 = take();
    e1 = take();
    PG_HIP(hipEventRecord(e0, st));
  }
  void end(hipStream_t st) {
    if (!armed) return;
    PG_HIP(hipEventRecord(e1, st));
    pairs.emplace_back(e0, e1);
    iter_of.push_back(cur);
    lean_of.push_back(cur_lean ? 1 : 0);
    second_of.push_back(cur_second ? 1 : 0);
    count_of.push_back(cur_count);
  }
  // launches queued after convergence return at their first instruction (done flag): they are not SpMVs and are
  // left out of the launch count and of the average
  void collect(SolveStats& s, int iters_done, bool half_exit) {
    for (size_t q = 0; q < pairs.size(); ++q) {
      auto& pr = pairs[q];
      float ms = 0.f;
      (void)hipEventSynchronize(pr.second);
      const bool ran = iter_of[q] < iters_done && !(half_exit && iter_of[q] == iters_done - 1 && second_of[q]);
      if (ran && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        if (lean_of[q]) { s.spmv_lean_ms += ms; s.spmv_lean_launches += count_of[q]; }
        else { s.spmv_ms += ms; s.spmv_launches += 1; }
      }
      pool().push_back(pr.first);
      pool().push_back(pr.second);
    }
    pairs.clear();
    iter_of.clear();
    lean_of.clear();
    second_of.clear();
    count_of.clear();
  }
};

}  // namespace

namespace pg {

double g_host_wait_us = 0.0;

void KrylovWork::init(i64 n_own, i64 n_vec) {
  n = n_own;
  nvec = n_vec;
  const i64 a = n_vec > 0 ? n_vec : 1;
  r.alloc(a); rhat.alloc(a); p.alloc(a); v.alloc(a); t.alloc(a);
  grid = spmv_default_grid(n_own);
  partials.alloc(5 * (i64)grid);
  sc.alloc(S_COUNT);
  sc.zero();
  ticket.alloc(1);
  ticket.zero();
  if (!h_sc) PG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_sc), sizeof(double) * S_COUNT));
}

KrylovWork::~KrylovWork() {
  if (h_sc) (void)hipHostFree(h_sc);
  if (ev_poll) (void)hipEventDestroy(ev_poll);
}

void spmv_halo(const CsrMatrix& A, const Numbering& nb, const Slab& slab, double* x, double* y, hipStream_t st) {
  spmv_with_halo(0, A, nb, slab, x, y, nullptr, nullptr, nullptr, spmv_default_grid(A.n), st);
  PG_HIP(hipGetLastError());
}

void spmv(const CsrMatrix& A, const double* x, double* y, hipStream_t st) {
  if (A.n == 0) return;
  launch_spmv(0, A, x, y, nullptr, nullptr, nullptr, spmv_default_grid(A.n), st);
  PG_HIP(hipGetLastError());
}

bool krylov_uses_polynomial(const CsrMatrix& A, const pg_krylov_opts& opts) {
  const bool poly_env = config().poly;
  const int degree_env = config().poly_degree;
  if (!(poly_env && opts.precond >= 0 && opts.method == PG_METHOD_BICGSTAB && A.poly_ok && spmv_supports_preconditioner_product() && A.n > 0))
    return false;
  return (opts.precond > 0 ? opts.precond : degree_env) >= 2;
}

constexpr int MAX_POLY_DEGREE = 40;   // products per application of the polynomial preconditioner

void krylov_solve(const CsrMatrix& A, const Numbering& nb, const Slab& slab, const double* b, double* x, KrylovWork& w,
                  const pg_krylov_opts& opts, SolveStats& stats, const double* x0, const double* Ax0, bool preinit) {
  Context& cx = ctx();
  hipStream_t st = cx.stream;
  const i64 n = A.n, nvec = nb.n_vec();
  PG_REQUIRE(w.n >= n && w.nvec >= nvec, "krylov workspace too small");   // (a reduced system works on prefixes, pg_reduce.hip)
  PG_REQUIRE(opts.method == PG_METHOD_BICGSTAB || opts.method == PG_METHOD_CG || opts.method == PG_METHOD_GMRES,
             "unknown Krylov method");
  if (opts.method == PG_METHOD_GMRES) {
    PG_REQUIRE(!preinit && !x0, "GMRES starts from zero (IterativeSolvers' default)");
    gmres_solve(A, nb, slab, b, x, w, opts, stats);
    return;
  }
  const int G = w.grid;
  const int check_every = opts.check_every > 0 ? opts.check_every : 4;
  int maxiter = opts.maxiter;
  if (maxiter <= 0) {
    // IterativeSolvers default: size of the system (global)
    maxiter = 100000;
  }
  // tolerances -> device scalars (a one-thread kernel: a host-to-device copy out of pageable memory stalls the stream)
  struct HookReset { KrylovWork& w; ~HookReset() { w.after_first_batch = nullptr; } } hook_reset{w};   // (every exit path)
  const bool p_in_rhat = w.p_in_rhat && preinit && opts.method == PG_METHOD_BICGSTAB;
  w.p_in_rhat = false;
  const XGuess xg = w.xguess;      // (first update of x out of place: see KrylovWork::xguess)
  w.xguess = XGuess();
  const bool start_folded = w.start_folded && preinit && opts.method == PG_METHOD_BICGSTAB;
  w.start_folded = false;
  const bool fused_start = !start_folded && preinit && opts.method == PG_METHOD_BICGSTAB && cx.nranks == 1 && !cx.comm;   // k_start below
  if (!fused_start && !start_folded)
    hipLaunchKernelGGL(k_sc_reset, dim3(1), dim3(S_COUNT), 0, st, w.sc.p, opts.reltol * opts.reltol, opts.abstol * opts.abstol);
  SpmvTimer timer(cx.profiling);

  const bool cg = opts.method == PG_METHOD_CG;
  const Config& cfg = config();
  const bool ntv = cfg.krylov_nt;
  const bool poly_env = cfg.poly;
  const int degree_env = cfg.poly_degree;
  // polynomial right preconditioner: BiCGStab on the slice kernel, where Gershgorin bounds the spectrum inside
  // |λ - 1| < 0.95; m products with Â per application of C = I - R(Â)  (m < 2: the plain iteration)
  int m = 0;
  if (poly_env && opts.precond >= 0 && !cg && A.poly_ok && spmv_supports_preconditioner_product() && n > 0)
    m = std::min(opts.precond > 0 ? opts.precond : degree_env, MAX_POLY_DEGREE);
  // auto mode: the degree that fits the number of products the previous solve on this matrix would have needed (below)
  const bool adapt_env = cfg.poly_adapt;
  const bool adaptive = adapt_env && m >= 2 && opts.precond == 0 && preinit;
  int expect_halves = 0;
  if (adaptive && w.adapt_matrix == &A && w.adapt_m >= 2) { m = w.adapt_m; expect_halves = w.adapt_h; }
  if (m < 2) m = 0;
  const bool poly = m > 0;
  stats.poly_degree = m;
  PG_REQUIRE(!w.scatter || (poly && preinit), "a compact system needs the polynomial path and a prepared start");
  double tau[MAX_POLY_DEGREE];
  if (poly) {
    // 1 / Chebyshev nodes of [1 - g, 1 + g], the largest and the smallest remaining root in turn
    const double g = std::min(std::max(A.gersh, 0.05), 0.95);
    double lam[MAX_POLY_DEGREE];
    for (int k = 0; k < m; ++k) lam[k] = 1.0 + g * std::cos(M_PI * (2.0 * k + 1.0) / (2.0 * m));   // descending
    for (int k = 0, lo = 0, hi = m - 1; k < m; ++k) tau[k] = 1.0 / ((k & 1) ? lam[hi--] : lam[lo++]);
    if (w.ya.n < nvec) {
      w.ya.alloc(nvec); w.wa.alloc(nvec); w.wb.alloc(nvec); w.yb.alloc(nvec);
      w.ya.zero(); w.wa.zero(); w.wb.zero(); w.yb.zero();
    }
    // y0 = 0 (x = x0 + q(Â) y): not written here -- the first update of y assigns (k_bicg_half / k_bicg_xrp, `fresh`)
  }
  // x-space form (default): every application of the operator computes M⁻¹in = q(Â) in by Horner's rule (m - 1 launches
  // of three streams, the first of two) and closes with the plain product Â(M⁻¹in) and its dots; the iteration updates x
  // itself.  The y-space form (PG_POLY_XSPACE=0) applies C = I - R(Â) in product form (m - 1 launches of two streams),
  // accumulates the solution y of the preconditioned system and recovers x = x0 + q(Â) y at the end: m - 1 Horner launches
  // and an update per SOLVE -- with 3 applications of degree 9 that recovery was 0.49 ms of a 2.4 ms step, the third stream
  // of 21 launches costs 0.17.
  const bool xspace_env = cfg.poly_xspace;
  const bool xspace = poly && xspace_env;
  stats.poly_xspace = xspace ? 1 : 0;
  double* const xit = (poly && !xspace) ? w.ya.p : x;   // what the iteration updates
  const int* const xmap = xspace ? w.scatter : nullptr; // ... through the caller's index map when the system is a compact image
  // convergence is also tested after the first half of an iteration when a half costs several products (the test itself
  // costs two small launches); PG_HALF_TEST=0/1 forces it off / on
  const int half_env = cfg.half_test;
  const bool half_test = !cg && (half_env < 0 ? m >= 3 : half_env != 0);
  // x-space: the first half's update of x rides with k_bicg_s (k_bicg_s_x): one pass and one launch less per iteration
  const bool fused_x = xspace && cfg.fuse_half_update;
  PG_REQUIRE(!xg.zbase || (fused_x && p_in_rhat && w.scatter), "an unformed extrapolated state needs the fused x-space update on a compact system");
  PG_REQUIRE(!preinit || !cg, "preinit is a BiCGStab path");
  if (!cg) {
    if (!preinit)
      hipLaunchKernelGGL(k_bicg_init, dim3(G), dim3(BLOCK), 0, st, n, nvec, b, x0, Ax0, x, w.r.p, w.rhat.p, w.p.p, w.v.p,
                       w.partials.p, (const double*)A.ds.p);
    static_assert(S_COUNT <= BLOCK, "k_start resets the scalar block with one thread per scalar");
    if (start_folded) {
    } else if (fused_start)
      hipLaunchKernelGGL(k_start, dim3(1), dim3(BLOCK), 0, st, (int)PH_INIT, 3, w.grid, (const double*)w.partials.p, w.sc.p,
                         opts.reltol * opts.reltol, opts.abstol * opts.abstol);
    else
      finalize(PH_INIT, 3, w, st, false);
  } else {
    hipLaunchKernelGGL(k_cg_init, dim3(G), dim3(BLOCK), 0, st, n, nvec, b, x, w.r.p, w.p.p, w.partials.p, (const double*)A.ds.p);
    finalize(PH_CG_INIT, 2, w, st, false);
  }
  PG_HIP(hipGetLastError());

  // Iterations are queued in batches and the done flag is polled in between.  Consecutive time steps need almost
  // the same number of iterations, so the first batch runs up to one short of the previous solve's count and the
  // next few polls come after every iteration: no iterations are queued past convergence (each would still cost
  // its launch overheads), at the price of two or three extra stream syncs per solve.
  // With the half-step test a batch is counted in HALVES and may end in the middle of an iteration: when the previous
  // solve ended at a half step (the usual case: 3 applications of the polynomial), the second half of the last iteration
  // is not queued at all -- its 1 + m launches would each find the done flag and return, at ~5 us apiece.
  int launched = 0, polls = 0;          // launched: iterations whose first half is queued
  bool mid = false;                     // ... and the second half of the last one is not
  bool done = false, poly_failed = false;
  const int poly_give_up = 40 + 400 / std::max(m, 1);   // iterations; an admitted system needs 2 .. 10
  const int derive_here = (cx.nranks == 1 && !cx.comm) ? 1 : 0;
  // out = C in = in - R(Â) in with the dots of `mode_last` (5: (r̂,out); 6: (out,in), (out,out), (r̂,out)); the plain
  // iteration applies Â itself (modes 1 / 3)
  auto apply = [&](double* in, double* out, int phase, int nslots, int itn) {
    const bool second = phase == PH_BICG_2;
    if (xspace) {
      // hat = q(Â) in:  u_(m-1) = τ_(m-1) in,  u_k = τ_k in + (I - τ_k Â) u_(k+1),  hat = u_0   (mode 8: pc2 base + pc0 x + pc1 Â x;
      // the first launch has x = base = in)
      double* hat = second ? w.yb.p : w.ya.p;
      double* src = in;
      timer.begin(st, itn, true, second, m - 1);
      for (int k = m - 2; k >= 0; --k) {
        double* dst = k == 0 ? hat : ((k & 1) ? w.wb.p : w.wa.p);
        FinArgs f{nullptr, nullptr, PH_NONE, 0, 0, nullptr};
        const double lead = k == m - 2 ? tau[m - 1] : 1.0;
        f.pc0 = lead; f.pc1 = -lead * tau[k]; f.pc2 = tau[k]; f.base = in;
        spmv_with_halo(8, A, nb, slab, src, dst, nullptr, nullptr, w.sc.p, G, st, &f);
        src = dst;
      }
      timer.end(st);
      FinArgs f{w.ticket.p, w.sc.p, phase, nslots, derive_here, nullptr};
      f.dotx = in;      // mode 3: `in` (= s) is the operand of the (out, .) dot, not the launch's x
      timer.begin(st, itn, false, second);
      const bool folded = spmv_with_halo(phase == PH_BICG_1 ? 1 : 3, A, nb, slab, hat, out, w.rhat.p, w.partials.p, w.sc.p, G, st, &f);
      timer.end(st);
      if (folded) finalize_folded(phase, nslots, w, st); else finalize(phase, nslots, w, st, true);
      return;
    }
    double* src = in;
    if (m > 1) timer.begin(st, itn, true, second, m - 1);
    for (int k = 0; k + 1 < m; ++k) {
      double* dst = (k & 1) ? w.wb.p : w.wa.p;
      FinArgs f{nullptr, nullptr, PH_NONE, 0, 0, nullptr};
      f.pc0 = 1.0; f.pc1 = -tau[k];
      spmv_with_halo(4, A, nb, slab, src, dst, nullptr, nullptr, w.sc.p, G, st, &f);   // w <- w - τ_k Â w
      src = dst;
    }
    if (m > 1) timer.end(st);
    // the scalar phase that follows is evaluated by the last block of the launch (stencil-slice kernel); with
    // several ranks the halo exchange of the input overlaps the rows that need no ghost value (spmv_with_halo)
    FinArgs f{w.ticket.p, w.sc.p, phase, nslots, derive_here, nullptr};
    int mode = phase == PH_BICG_1 ? 1 : 3;
    if (poly) {
      mode = phase == PH_BICG_1 ? 5 : 6;
      f.pc0 = 1.0; f.pc1 = -tau[m - 1];
      f.base = in;      // mode 5: out = in - (w - τ_m Â w)
      f.dotx = in;      // mode 6: the same, `in` is the operand of the (out, in) dot as well
    }
    timer.begin(st, itn, false, second);
    const bool folded = spmv_with_halo(mode, A, nb, slab, src, out, w.rhat.p, w.partials.p, w.sc.p, G, st, &f);
    timer.end(st);
    if (folded) finalize_folded(phase, nslots, w, st); else finalize(phase, nslots, w, st, true);
  };
  // test: make the half-step test in this iteration.  (Not before the half step the previous solve ended at: a solve that
  // could have stopped one application earlier stops at the end of that iteration instead, which happens about never,
  // and the k_bicg_half launches of the earlier iterations -- which return at once -- are not queued.)
  auto first_half = [&](int itn, bool test) {
    double* pvec = (p_in_rhat && itn == 0) ? w.rhat.p : w.p.p;
    apply(pvec, w.v.p, PH_BICG_1, 3, itn);     // v = C p, (r̂,v); previous iteration's (r,r): convergence / restart; then α
    const bool half_test = test;
    unsigned* tk = (half_test && derive_here) ? w.ticket.p : nullptr;   // the half-step test inside k_bicg_s
    if (fused_x) {
      // s, its dots, the half-step test AND x += α M⁻¹p in one pass; without the test in this iteration the sums of slot 4
      // are simply not looked at
      if (ntv) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bicg_s_x<true>), dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.v.p, w.rhat.p, w.r.p, w.partials.p, (const double*)A.ds.p, tk, p_in_rhat ? 1 : 0, (const double*)w.ya.p, xit, xmap, xg, itn == 0 ? 1 : 0);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bicg_s_x<false>), dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.v.p, w.rhat.p, w.r.p, w.partials.p, (const double*)A.ds.p, tk, p_in_rhat ? 1 : 0, (const double*)w.ya.p, xit, xmap, xg, itn == 0 ? 1 : 0);
      if (half_test && !tk) finalize(PH_BICG_S, 1, w, st, true, 4);
      return;
    }
    if (ntv) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bicg_s<true>), dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.v.p, w.rhat.p, w.r.p, w.partials.p, (const double*)A.ds.p, tk, p_in_rhat ? 1 : 0);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bicg_s<false>), dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.v.p, w.rhat.p, w.r.p, w.partials.p, (const double*)A.ds.p, tk, p_in_rhat ? 1 : 0);
    if (half_test) {   // does s already meet the tolerance?  then x += αp and stop: the second half is 1 + m launches
      if (!tk) finalize(PH_BICG_S, 1, w, st, true, 4);
      hipLaunchKernelGGL(k_bicg_half, dim3(G), dim3(BLOCK), 0, st, n, (const double*)w.sc.p,
                         xspace ? (const double*)w.ya.p : (const double*)pvec, xit, (poly && !xspace) ? 1 : 0, itn + 1, xmap);
    }
  };
  auto second_half = [&](int itn) {
    apply(w.r.p, w.t.p, PH_BICG_2, 5, itn);     // t = C s (r holds s), (t,s), (t,t), (r̂,t); then ω, ρ, β / restart
    if (ntv) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bicg_xrp<true>), dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.t.p, w.v.p, xit, w.r.p,
                                w.p.p, w.rhat.p, w.partials.p, (const double*)A.ds.p, (poly && !xspace) ? 1 : 0, p_in_rhat ? 1 : 0,
                                xspace ? (const double*)w.ya.p : nullptr, xspace ? (const double*)w.yb.p : nullptr, xmap, fused_x ? 1 : 0);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bicg_xrp<false>), dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.t.p, w.v.p, xit, w.r.p,
                            w.p.p, w.rhat.p, w.partials.p, (const double*)A.ds.p, (poly && !xspace) ? 1 : 0, p_in_rhat ? 1 : 0,
                                xspace ? (const double*)w.ya.p : nullptr, xspace ? (const double*)w.yb.p : nullptr, xmap, fused_x ? 1 : 0);
  };
  const bool half_batches = cfg.half_batch;   // (0: whole iterations, for A/B runs)
  while (!done) {
    int want = check_every;
    if (w.last_iters > 1) want = polls == 0 ? w.last_iters - 1 : (polls <= 4 ? 1 : check_every);
    if (expect_halves > 0) want = polls == 0 ? (expect_halves + 1) / 2 : (polls <= 4 ? 1 : check_every);   // the predicted count at once
    const int batch = std::max(1, std::min(want, maxiter - launched));
    int halves = 2 * batch;
    if (!cg && half_test && half_batches && expect_halves > 0 && polls == 0) halves = std::max(1, std::min(expect_halves, 2 * (maxiter - launched)));
    ++polls;
    if (!cg) {
      const bool predicted = half_test && half_batches && expect_halves > 0 && polls == 1;
      for (int h = 0; h < halves; ++h) {
        if (!mid) {
          if (launched >= maxiter) break;
          first_half(launched, half_test && (!predicted || h + 2 >= halves));
          ++launched;
          mid = true;
        } else {
          second_half(launched - 1);
          mid = false;
        }
      }
    } else {
      for (int it = 0; it < batch; ++it) {
        timer.begin(st, launched + it);
        spmv_with_halo(2, A, nb, slab, w.p.p, w.v.p, nullptr, w.partials.p, w.sc.p, G, st);   // v = A p ; (v.p), (v.v)
        timer.end(st);
        finalize(PH_CG_1, 1, w, st, true);
        hipLaunchKernelGGL(k_cg_xr, dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.p.p, w.v.p, x, w.r.p, w.partials.p, (const double*)A.ds.p);
        finalize(PH_CG_2, 2, w, st, true);
        hipLaunchKernelGGL(k_cg_p, dim3(G), dim3(BLOCK), 0, st, n, w.sc.p, w.r.p, w.p.p);
      }
      launched += batch;
    }
    if (!cg && !mid) finalize(PH_BICG_3, 2, w, st, true, 1);   // the last iteration's (r,r), not yet folded into a next one
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemcpyAsync(w.h_sc, w.sc.p, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, st));
    const auto t_wait0 = std::chrono::steady_clock::now();
    struct WaitClock {   // PG_DEBUG: where the host's time goes (pg_solver_run prints the totals)
      std::chrono::steady_clock::time_point t0;
      ~WaitClock() { g_host_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    } wait_clock{t_wait0};
    if (w.after_first_batch && polls == 1) {
      // the caller's speculative work goes behind the copy; the host waits for the COPY only
      if (!w.ev_poll) PG_HIP(hipEventCreateWithFlags(&w.ev_poll, hipEventDisableTiming));
      PG_HIP(hipEventRecord(w.ev_poll, st));
      std::function<void()> hook;
      hook.swap(w.after_first_batch);
      hook();
      PG_HIP(hipEventSynchronize(w.ev_poll));
    } else {
      PG_HIP(hipStreamSynchronize(st));   // (spinning on hipStreamQuery instead: no measurable difference)
    }
    if (w.h_sc[S_DONE] != 0.0 || (launched >= maxiter && !mid)) done = true;
    // safety net of the polynomial preconditioner: its roots assume a (nearly) real spectrum inside the Gershgorin
    // interval; a matrix that defeats that assumption shows as stagnation, and the solve falls back to the plain iteration
    // from the iterate reached (the matrix keeps the verdict)
    if (!done && poly && launched >= poly_give_up) { poly_failed = true; done = true; }
  }
  if (xspace && poly_failed) {              // x is the iterate reached: nothing to recover
    const_cast<CsrMatrix&>(A).poly_ok = false;
    timer.collect(stats, launched, false);
    if (w.scatter) {                        // (the caller solves again on its full system, which has a right-hand side vector)
      stats.iters = launched;
      stats.converged = 0;
      stats.poly_degree = -1;
      return;
    }
    if (cfg.debug) fprintf(stderr, "[pg_krylov] polynomial preconditioner (m = %d) stagnated after %d iterations: plain iteration\n", m, launched);
    spmv_with_halo(0, A, nb, slab, x, w.t.p, nullptr, nullptr, nullptr, G, st);   // Â x of the iterate reached
    SolveStats rest;
    krylov_solve(A, nb, slab, b, x, w, opts, rest, x, w.t.p, false);
    stats = rest;
    stats.iters += launched;
    return;
  }
  if (poly && !xspace && w.h_sc[S_ITERS] > 0.0) {      // (no iteration: the start already met the tolerance, y was never written)
    // x = x0 + q(Â) y,  q(Â) y = Σ_k τ_k w_k,  w_0 = y, w_(k+1) = (I - τ_k Â) w_k  -- evaluated by Horner's rule from the inside,
    //   u_(m-1) = τ_(m-1) y,   u_k = τ_k y + (I - τ_k Â) u_(k+1),   q(Â) y = u_0,
    // in the scaled variable ũ_k = u_k / τ_k so that a launch needs no scaled copy of y:
    //   ũ_(m-1) = y,   ũ_k = y + c_k (ũ_(k+1) - τ_k Â ũ_(k+1)),  c_k = τ_(k+1) / τ_k      (mode 8: x in, y-vector, out)
    // m - 1 launches of three vector streams each (the accumulating form, mode 7, read and wrote x as a fourth in every
    // launch); the factors are applied in the reverse of the chain's order, which alternates between the two ends of the
    // spectrum either way.  (No done flag here: it is set.)
    const bool horner_env = cfg.recovery_horner;
    const bool horner = horner_env || w.scatter != nullptr;
    double* src = w.ya.p;
    if (horner) {
      for (int k = m - 2; k >= 0; --k) {
        double* dst = (k & 1) ? w.wb.p : w.wa.p;
        FinArgs f{nullptr, nullptr, PH_NONE, 0, 0, nullptr};
        const double c = tau[k + 1] / tau[k];
        f.pc0 = c; f.pc1 = -c * tau[k]; f.pc2 = 1.0; f.base = w.ya.p;
        spmv_with_halo(8, A, nb, slab, src, dst, nullptr, nullptr, nullptr, G, st, &f);
        src = dst;
      }
      if (w.scatter) hipLaunchKernelGGL(k_axpy_scatter, dim3(G), dim3(BLOCK), 0, st, n, tau[0], (const double*)src, x, w.scatter);
      else hipLaunchKernelGGL(k_axpy, dim3(G), dim3(BLOCK), 0, st, n, tau[0], (const double*)src, x);
    } else {
      for (int k = 0; k + 1 < m; ++k) {
        double* dst = (k & 1) ? w.wb.p : w.wa.p;
        FinArgs f{nullptr, nullptr, PH_NONE, 0, 0, nullptr};
        f.pc0 = 1.0; f.pc1 = -tau[k]; f.accv = x; f.pc2 = tau[k];
        spmv_with_halo(7, A, nb, slab, src, dst, nullptr, nullptr, nullptr, G, st, &f);
        src = dst;
      }
      hipLaunchKernelGGL(k_axpy, dim3(G), dim3(BLOCK), 0, st, n, tau[m - 1], (const double*)src, x);
    }
    PG_HIP(hipGetLastError());
    if (poly_failed && w.scatter) {      // (the caller solves again on its full system, which has a right-hand side vector)
      const_cast<CsrMatrix&>(A).poly_ok = false;
      stats.iters = launched;
      stats.converged = 0;
      stats.poly_degree = -1;
      timer.collect(stats, launched, false);
      return;
    }
    if (poly_failed) {
      if (cfg.debug) fprintf(stderr, "[pg_krylov] polynomial preconditioner (m = %d) stagnated after %d iterations: plain iteration\n", m, launched);
      const_cast<CsrMatrix&>(A).poly_ok = false;
      spmv_with_halo(0, A, nb, slab, x, w.t.p, nullptr, nullptr, nullptr, G, st);   // Â x of the iterate reached
      timer.collect(stats, launched, false);
      SolveStats rest;
      krylov_solve(A, nb, slab, b, x, w, opts, rest, x, w.t.p, false);
      stats = rest;
      stats.iters += launched;
      return;
    }
  }
  if (cfg.debug)
    fprintf(stderr, "[pg_krylov] done=%g iters=%g rr0=%g rr=%g tol2=%g rho=%g rho_old=%g alpha=%g omega=%g beta=%g red0=%g red1=%g\n",
            w.h_sc[S_DONE], w.h_sc[S_ITERS], w.h_sc[S_RR0], w.h_sc[S_RR], w.h_sc[S_TOL2], w.h_sc[S_RHO], w.h_sc[S_RHO_OLD], w.h_sc[S_ALPHA],
            w.h_sc[S_OMEGA], w.h_sc[S_BETA], w.h_sc[S_RED0], w.h_sc[S_RED1]);
  stats.iters = (int)w.h_sc[S_ITERS];
  stats.polls = polls;
  w.last_iters = stats.iters;
  stats.converged = w.h_sc[S_DONE] == 1.0 ? 1 : 0;
  stats.half_exit = w.h_sc[S_HALF] != 0.0 ? 1 : 0;
  stats.resnorm = std::sqrt(cg ? w.h_sc[S_RRW] : w.h_sc[S_RR]);   // the weighted norms of the convergence test
  stats.bnorm = std::sqrt(w.h_sc[S_BB]);
  timer.collect(stats, stats.iters, w.h_sc[S_HALF] != 0.0);
  // Degree for the next solve on this matrix (the loop's warm solves: the start residual and the tolerance move slowly from
  // step to step).  This solve applied P = (2 iters - half) m products and took (r,r)_W from rr0 to rr; at that rate the
  // tolerance needed  P log(tol²/rr0) / log(rr/rr0)  of them.  An application of the operator costs its m - 1 lean launches
  // + one closing launch with its vector kernel (≈ 3.2 lean launches): take the (applications h, degree m) with h m >= need
  // that is cheapest -- in the y-space form 27 products are 3 x 9, not 5 x 6 (2.5 iterations of 12) or 2 x 16 of degree 8
  // (335 -> 370 steps/s at 512^3), because its recovery costs m - 1 launches per solve.  The x-space form has no such term and
  // no exactness requirement on the polynomial (x and r move by the SAME computed M⁻¹p, so r stays the residual of x whatever
  // rounding does inside the Horner chain): there the cheapest is ONE application of degree ~26 that ends at the half-step
  // test -- a Chebyshev step with BiCGStab's α, its residual test and, if the estimate was short, its continuation
  // (444 -> 546 steps/s at 512^3 against 3 x 9 in the same form).  No safety margin: a miss costs one more application
  // once, and the next estimate is made from that solve.
  stats.products = (i64)(2 * stats.iters - stats.half_exit) * (m >= 2 ? m : 1);
  if (!cg && stats.converged && stats.products > 0) {   // (what a product took off log (r,r)_W: the extrapolated start's cost model)
    const double rr0 = w.h_sc[S_RR0], rr = w.h_sc[S_HALF] != 0.0 ? w.h_sc[S_RED4] : w.h_sc[S_RR];
    if (rr0 > 0.0 && rr > 0.0 && rr < rr0) w.last_rate2 = std::log(rr0 / rr) / (double)stats.products;
  }
  const bool keep_degree = adaptive && stats.converged && stats.iters == 0 && w.adapt_m >= 2;   // met the tolerance at the start:
  w.adapt_matrix = nullptr;                                                                      // the smallest polynomial next time
  if (keep_degree) { w.adapt_matrix = &A; w.adapt_m = cfg.poly_mindeg; w.adapt_h = 1; }
  if (adaptive && stats.converged && stats.iters > 0) {
    const double rr0 = w.h_sc[S_RR0], rr = w.h_sc[S_HALF] != 0.0 ? w.h_sc[S_RED4] : w.h_sc[S_RR], tol2 = w.h_sc[S_TOL2];
    const double P = (2.0 * stats.iters - stats.half_exit) * m;
    if (rr0 > tol2 && rr > 0.0 && rr < rr0 && tol2 > 0.0) {
      w.last_rate2 = std::log(rr0 / rr) / P;
      const double margin = cfg.poly_margin;
      // x-space: a third of a product of slack -- with ONE application per solve a miss costs a whole second one, and the
      // estimate moves by a few tenths from step to step (200 steps at 512^3: 7 % of the solves missed without it, 548 steps/s;
      // 1.75 % with 0.3: 575; none with 0.6: 600 -- but 27 instead of 26 products in the bench's first 20 steps)
      const double slack_env = cfg.poly_slack;
      const double slack = slack_env >= 0.0 ? slack_env : (xspace ? 0.3 : 0.0);
      const double need_now = std::min(P, P * std::log(tol2 / rr0) / std::log(rr / rr0)) * margin + slack;
      // the largest of the last three estimates on this matrix: the estimate wobbles by a few tenths of a product from step
      // to step, and with one application per solve falling short by a tenth costs a whole second application
      const int hist_n = cfg.poly_hist;
      if (w.need_matrix != &A) { w.need_matrix = &A; w.need_hist[0] = w.need_hist[1] = w.need_hist[2] = 0.0; }
      w.need_hist[2] = w.need_hist[1]; w.need_hist[1] = w.need_hist[0]; w.need_hist[0] = need_now;
      double need = need_now;
      // The estimates of a run move along a trend (the start residual falls from step to step -- by a third of a product per
      // step while the extrapolated start of pg_solver.hip settles in) with a wobble of a few tenths on top.  Trend: half the
      // change over the last two steps; the older estimates are carried along it before the largest is taken (the wobble
      // guard), and the next solve gets the value the trend predicts for it.  (Without the trend the guard alone kept the
      // degree 0.6 + a step's change above what was needed all the way down: one product per step in the bench window.)
      double trend = 0.0;
      if (xspace && cfg.poly_trend && hist_n >= 3 && w.need_hist[2] > 0.0) trend = std::max(-0.5, std::min(0.2, 0.5 * (need_now - w.need_hist[2])));
      if (xspace) for (int q = 1; q < hist_n; ++q)
        if (w.need_hist[q] > 0.0) need = std::max(need, std::min(w.need_hist[q] + q * trend, need_now + 0.6));   // (wobble, not trend)
      need += trend;
      double best = 1e300;
      for (int h = 1; h <= 16; ++h) {
        // (a need below the smallest polynomial -- an extrapolated start close to the tolerance, a field close to steady -- takes
        //  the smallest one; it used to find no admissible degree, which sent the next solve back to the first-solve default
        //  and its extra polls: 64^3 near steady state 240 us per step with 3 products, 150 with 16)
        const int mm = std::max(cfg.poly_mindeg, (int)std::ceil(need / h));
        const int maxdeg = cfg.poly_maxdeg > 0 ? std::min(MAX_POLY_DEGREE, std::max(4, cfg.poly_maxdeg)) : (xspace ? 32 : 10);
        if (mm > maxdeg) continue;
        // x-space: a chain = one lean launch + mm - 2 Horner launches (1.18 lean launches each), closing launch + vector
        // kernel ≈ 3.6; y-space: mm - 1 lean launches, closing ≈ 3.2, + the recovery's mm - 1 Horner launches per solve
        const double cost = xspace ? h * (1.0 + 1.18 * (mm - 2) + 3.6) : h * ((mm - 1) + 3.2) + 1.25 * (mm - 1);
        if (cost < best) { best = cost; w.adapt_m = mm; w.adapt_h = h; }
      }
      if (best < 1e300) w.adapt_matrix = &A;
      // One application per solve (x-space): what a degree achieves is not monotone in the degree -- the residual polynomial
      // at the few eigenvalues that carry an extrapolated start's residual goes like |cos(m θ)| on top of the Chebyshev
      // bound, so the estimate made at a lucky degree can undershoot by a product (512^3, step 96: 12, 11, 10 all reached
      // the same residual, 9 fell short; the loop then cycled 11, 11, 10, 9+9).  A degree that fell short is therefore not
      // tried again for a while -- twice what the miss cost, in steps (a miss costs about m + 3 products, a degree less saves
      // one per step), doubled each time the same degree falls short again -- unless the problem has become easier by a
      // product's worth per degree first; and the solve after a miss goes back to the degree that last sufficed.
      if (xspace && stats.iters == 1) {
        const double Lnow = std::log(rr0 / tol2);
        if (w.fail_matrix != &A) { w.fail_matrix = &A; w.fail_deg = 0; w.good_deg = 0; w.fail_wait = 0; w.fail_backoff = 1; }
        if (w.fail_wait > 0) --w.fail_wait;
        if (stats.half_exit) w.good_deg = m;
        else {
          w.fail_backoff = (m == w.fail_deg) ? std::min(w.fail_backoff * 2, 64) : 1;
          w.fail_deg = m; w.fail_L = Lnow;
          w.fail_wait = 2 * w.fail_backoff * (m + 3);
        }
        if (best < 1e300 && w.adapt_h == 1) {
          if (!stats.half_exit && w.good_deg > m) w.adapt_m = std::max(m + 1, std::min(w.adapt_m, w.good_deg));
          const double per_product = std::max(1.0, w.last_rate2);
          if (w.fail_deg > 0 && w.fail_wait > 0 && w.adapt_m <= w.fail_deg &&
              Lnow > w.fail_L - per_product * (w.fail_deg - w.adapt_m + 1))
            w.adapt_m = w.fail_deg + 1;
        }
      }
      if (cfg.debug) fprintf(stderr, "[pg_krylov] degree %d used %g products, needed %.1f -> next: %d applications of degree %d\n", m, P, need, w.adapt_h, w.adapt_m);
    }
  }
}

}  // namespace pg
