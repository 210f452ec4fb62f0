// pg_solver.hip -- Solver / DiffusionUnsteadyMono / DiffusionUnsteadyDiph and their time loops
//
//   reference                                                        here
//   DiffusionUnsteadyMono ctor               diffusion.jl:192-210     pg_solver_create_unsteady_mono
//   DiffusionUnsteadyDiph ctor               diffusion.jl:319-332     pg_solver_create_unsteady_diph
//   b_mono_unstead_diff / b_diph_unstead_diff diffusion.jl:243-265,391-420   k_row_static + k_bconst + k_rhs (K8)
//   solve_DiffusionUnsteadyMono!/Diph!       diffusion.jl:268-301,422-454    pg_solver_initial_solve / _step / _run
//   s.x = zeros(n); s.x[cols_idx] = x_reduced solver.jl:186-187       k_scatter_active (K12)
//
// Everything the loop touches stays in HBM in the reduced (active-set) layout; the padded 2M/4M vector is
// only materialised when the host asks for a state.
#include <algorithm>
#include <cmath>
#include <memory>

#include "pg_krylov.h"
#include "pg_reduce.h"
#include "pg_spmv.h"

using namespace pg;

namespace pg { extern double g_host_wait_us; }

struct pg_solver {
  int nphase = 1;
  pg_capacity* cap[2] = {nullptr, nullptr};
  pg_diffops* ops[2] = {nullptr, nullptr};   // (their convection data, if any, enters the bulk rows)
  Slab slab;
  int K = 2;
  i64 Mloc = 0, M = 0;
  double dt = 0.0, t = 0.0;
  int scheme_ctor = PG_SCHEME_BE;
  // problem description
  pg_bc_desc bc_i{};          // mono interface condition
  pg_jump_desc ic{};          // diph jumps
  int border_kind[6] = {0, 0, 0, 0, 0, 0};
  double border_value[6] = {0, 0, 0, 0, 0, 0};
  double inv_dx = 0.0;
  DevBuf<double> Id[2];       // D(C_ω), optional
  DevBuf<double> f_n[2], f_np1[2];   // source at t and t+Δt (padded local), optional (NULL = 0)
  DevBuf<double> g_n, g_np1;  // mono interface value arrays, optional
  DevBuf<double> g_arr, h_arr;  // diph jump arrays, optional
  // numbering + matrices
  Numbering nb;
  CsrMatrix A_ctor, A_run;
  GammaElim elim_ctor, elim_run;   // the same matrices without the Dirichlet interface unknowns (pg_reduce.hip), on first use
  DiagElim diag_ctor, diag_run;    // ... without any row that is alone on its diagonal, compact vectors (PG_DIAG_ELIM=1)
  bool have_run = false;
  int scheme_run = -1;
  // per-row data
  DevBuf<double> mass, bconst, bcv;
  DevBuf<unsigned char> fixed;
  int bconst_scheme = -1;
  bool bconst_dirty = true;
  long long bconst_version = 0;   // counts the recomputations of bconst
  // block rows of the warm loop's right-hand side with everything that does not depend on the state folded into two tables
  // (k_blk_cache / k_rhs_block_c), valid for one (matrix, scheme, bconst)
  DevBuf<double> blk_wz, blk_c0;
  const CsrMatrix* blk_cache_matrix = nullptr;
  int blk_cache_scheme = -1;
  long long blk_cache_version = -1;
  // vectors
  DevBuf<double> x, b, y;     // n_vec: unscaled state, scaled right-hand side, Â z
  DevBuf<double> z, ysol;     // n_vec: scaled state S⁻¹x (SpMV input), Krylov solution y (x = S y)
  // The loop carries the SCALED state z (the Krylov solution itself); the unscaled x = S z is materialised only when
  // somebody asks for it (state download, saved states, a cold-start step).
  bool x_valid = true;
  const CsrMatrix* z_matrix = nullptr;   // the matrix whose S the current z refers to
  DevBuf<double> T0pad;       // K*Mloc, ctor initial condition
  KrylovWork work;
  // ŷ = Â z of the NEXT step, queued speculatively behind the previous solve's first batch (KrylovWork::after_first_batch):
  // valid when that solve ended inside the batch (nothing moved z afterwards) and the next step runs on the same matrix
  bool spec_y_valid = false, spec_pending = false, hint_more_steps = false;
  const CsrMatrix* spec_matrix = nullptr;
  // the last states z^{n-1}.. and their products ŷ^{n-1}.. = Â z^{n-1}.. (same-size buffers that trade places with z / y):
  // the extrapolated start of a quiet step (GuessArgs).  hist_cnt of them are valid, all for the matrix of hist_de.
  DevBuf<double> zh[8], yh[8], guess_coef, guess_partials;
  DevBuf<unsigned> guess_ticket;
  // k_guess_fit runs beside the solve on a stream of its own (one rank): it reads what the step's first kernel left (b̂, r̂,
  // the products) and writes the coefficients the NEXT step's first kernel reads; fit_join() orders that kernel behind it
  struct FitAsync {
    hipStream_t st = nullptr;
    hipEvent_t ev_rhs = nullptr, ev_fit = nullptr;
    bool pending = false;
    ~FitAsync() {
      if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
      if (ev_rhs) (void)hipEventDestroy(ev_rhs);
      if (ev_fit) (void)hipEventDestroy(ev_fit);
    }
  } fit;
  ~pg_solver() { if (fit.st) (void)hipStreamSynchronize(fit.st); }   // (before any buffer it reads goes back to the cache)
  int hist_cnt = 0, hist_k = 0;     // (hist_k: the ring's depth the count refers to)
  bool plain_guess_on[2] = {false, false};          // the plain warm path's switch and last solve's products, per matrix
  double plain_last_products[2] = {0.0, 0.0};       // (ctor / run; the compact path keeps its own in DiagElim)
  const void* hist_de = nullptr;
  bool initial_done = false;
  std::vector<DevBuf<double>> states;
  i64 steps_done = 0;
  DevBuf<double> red_scratch;  // max-abs partials
  DevBuf<i64> border_cells_lin;  // global linear index of each border cell (mesh order), lazily uploaded
  // moving body (pg_solver_create_moving_mono): one space-time step; Ψn1 = psip.(Vn, Vn_1), Ψn = psim.(Vn, Vn_1)
  bool moving = false;
  DevBuf<double> psi_p[2], psi_m[2];   // per phase
  pg_solver* init_from = nullptr;   // constructor only: the previous slab's solver whose state is this one's initial state
};

namespace {

using pg::BLOCK;   // 256 (pg_spmv.h)

// the compute stream waits for a fit that may still be running beside it (host = true: the host does)
void fit_join(pg_solver* s, bool host = false) {
  if (!s->fit.pending) return;
  if (host) PG_HIP(hipStreamSynchronize(s->fit.st));
  else PG_HIP(hipStreamWaitEvent(ctx().stream, s->fit.ev_fit, 0));
  s->fit.pending = false;
}

struct RowSegs {
  int K;
  i64 off_own[MAX_KINDS];
};

__device__ inline int seg_kind(const RowSegs& s, i64 r) {
  int k = 0;
  while (k + 1 < s.K && r >= s.off_own[k + 1]) ++k;
  return k;
}

RowSegs make_segs(const Numbering& nb) {
  RowSegs s;
  s.K = nb.K;
  for (int k = 0; k < MAX_KINDS; ++k) s.off_own[k] = nb.off_own[k];
  return s;
}

struct KeyVals {
  double v[6];
};

// static per-row data: mass (V for bulk rows that are not overwritten), fixed (b = bconst), border constant
__global__ void k_row_static(SysParams P, RowSegs seg, i64 n_own, const int* row_cell, KeyVals kv, double* mass,
                             unsigned char* fixed, double* bcv) {
  const CapView& c = P.cap[0];
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_own; r += (i64)gridDim.x * blockDim.x) {
    const int k = seg_kind(seg, r);
    const i64 lc = row_cell[r];
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    double m = 0.0, bv = 0.0;
    unsigned char fx = 0;
    if ((k & 1) == 0) {
      const int ph = k >> 1;
      int key = -1;
      const int bk = border_row_kind(P, ph, lc, idx, &key);
      if (bk != PG_BC_NONE) {
        fx = 1;
        bv = (bk == PG_BC_PERIODIC) ? 0.0 : kv.v[key];
      } else {
        m = P.mv_psi_w[0] ? P.mv_v1[ph][lc] : P.mass * P.cap[ph].V[lc];   // moving: b1 = Vn Tω + ...  (diffusion.jl:214,216)
      }
    } else if (P.nphase == 2) {
      fx = 1;   // jump rows: b2 = g, b4 = Γ₂h for both schemes (diffusion.jl:415-416)
    }
    mass[r] = m;
    fixed[r] = fx;
    bcv[r] = bv;
  }
}

struct SrcView {
  const double* f_n[2];
  const double* f_np1[2];
  const double* g_n;     // mono: g(t)      diph: g array
  const double* g_np1;   // mono: g(t+Δt)   diph: h array
  double g_const, h_const;
};

// time-data part of the right-hand side                 diffusion.jl:257-261, 409-416
__global__ void k_bconst(SysParams P, RowSegs seg, i64 n_own, const int* row_cell, SrcView s, int scheme, double dt,
                         const unsigned char* fixed, const double* bcv, double* bconst, int moving) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_own; r += (i64)gridDim.x * blockDim.x) {
    const int k = seg_kind(seg, r);
    const i64 lc = row_cell[r];
    const int ph = k >> 1;
    double v;
    if ((k & 1) == 0) {
      if (fixed[r]) v = bcv[r];
      else {
        const double V = P.cap[ph].V[lc];
        const double f1 = s.f_np1[ph] ? s.f_np1[ph][lc] : 0.0;
        if (scheme == PG_SCHEME_CN) {
          // f(t) not supplied separately (BE constructor, constant-in-time source): f(t) = f(t+Δt)
          const double f0 = s.f_n[ph] ? s.f_n[ph][lc] : f1;
          v = dt / 2 * V * (f0 + f1);
        } else {
          v = dt * V * f1;
        }
      }
    } else if (P.nphase == 1) {
      const double G = P.cap[0].G[lc];
      const double g1 = s.g_np1 ? s.g_np1[lc] : s.g_const;
      if (scheme == PG_SCHEME_CN && !moving) {   // moving: b2 = Γ g under both schemes (diffusion.jl:218)
        const double g0 = s.g_n ? s.g_n[lc] : s.g_const;
        v = dt / 2 * G * (g0 + g1);
      } else {
        v = G * g1;
      }
    } else if (k == 1) {
      v = s.g_n ? s.g_n[lc] : s.g_const;                          // b2 = gᵧ
    } else {
      v = P.cap[1].G[lc] * (s.g_np1 ? s.g_np1[lc] : s.h_const);   // b4 = Γ₂ hᵧ
    }
    bconst[r] = v;
  }
}

// K8, loop form: x is the previous (unscaled) reduced state; yhat = Â (S⁻¹x) (CN only); output is S b
__global__ void k_rhs(i64 n, int scheme, const double* __restrict__ x, const double* __restrict__ yhat,
                      const double* __restrict__ ds, const double* __restrict__ mass, const double* __restrict__ bconst,
                      const unsigned char* __restrict__ fixed, double* __restrict__ b) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    double v;
    if (fixed[r]) v = ds[r] * bconst[r];
    else if (scheme == PG_SCHEME_CN) v = ds[r] * (2.0 * (mass[r] * x[r]) + bconst[r]) - yhat[r];
    else v = ds[r] * (mass[r] * x[r] + bconst[r]);
    b[r] = v;
  }
}

// K8 for the rows of cells with a non-trivial preconditioner block (overwrites what k_rhs wrote there):
//   b̂_r = Σ_j (B⁻¹S)[r,j] c_j  -  Σ_j (B⁻¹MB)[r,j] ŷ_j      c = un-preconditioned constant part, ŷ = Â S⁻¹x
// (x = ds ∘ z when ds != NULL: the loop form carries the scaled state)
__global__ void k_rhs_block(i64 nblk, int scheme, const int* __restrict__ blk_rows, const int* __restrict__ blk_idx,
                            const double* __restrict__ blk_coef, const double* __restrict__ blk_cn,
                            const double* __restrict__ x, const double* __restrict__ ds, const double* __restrict__ yhat,
                            const double* __restrict__ mass, const double* __restrict__ bconst,
                            const unsigned char* __restrict__ fixed, double* __restrict__ b) {
  const double fac = scheme == PG_SCHEME_CN ? 2.0 : 1.0;
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nblk; q += (i64)gridDim.x * blockDim.x) {
    double v = 0.0;
#pragma unroll
    for (int a = 0; a < MAX_KINDS; ++a) {
      const int j = blk_idx[q * MAX_KINDS + a];
      if (j < 0) continue;
      const double xj = ds ? ds[j] * x[j] : x[j];
      const double cj = fixed[j] ? bconst[j] : fac * (mass[j] * xj) + bconst[j];
      v += blk_coef[q * MAX_KINDS + a] * cj;
      if (scheme == PG_SCHEME_CN) v -= blk_cn[q * MAX_KINDS + a] * yhat[j];
    }
    b[blk_rows[q]] = v;
  }
}

// The same rows in the warm loop, state-independent part precomputed:  b̂_r = c0_q + Σ_a (wz_qa z_j - cn_qa ŷ_j)  with
//   wz_qa = (B⁻¹S)[r,j] fac mass_j s_j  (0 for fixed rows),   c0_q = Σ_a (B⁻¹S)[r,j] bconst_j
// -- two gathers per unknown of the cell instead of six (42 -> ~20 us at 512^3)
__global__ void k_blk_cache(i64 nblk, int scheme, const int* __restrict__ blk_idx, const double* __restrict__ blk_coef,
                            const double* __restrict__ ds, const double* __restrict__ mass, const double* __restrict__ bconst,
                            const unsigned char* __restrict__ fixed, double* __restrict__ wz, double* __restrict__ c0) {
  const double fac = scheme == PG_SCHEME_CN ? 2.0 : 1.0;
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nblk; q += (i64)gridDim.x * blockDim.x) {
    double c = 0.0;
#pragma unroll
    for (int a = 0; a < MAX_KINDS; ++a) {
      const int j = blk_idx[q * MAX_KINDS + a];
      double w = 0.0;
      if (j >= 0) {
        const double cf = blk_coef[q * MAX_KINDS + a];
        c += cf * bconst[j];
        if (!fixed[j]) w = cf * (fac * (mass[j] * ds[j]));
      }
      wz[q * MAX_KINDS + a] = w;
    }
    c0[q] = c;
  }
}

__global__ void k_rhs_block_c(i64 nblk, int scheme, const int* __restrict__ blk_rows, const int* __restrict__ blk_idx,
                              const double* __restrict__ wz, const double* __restrict__ c0, const double* __restrict__ blk_cn,
                              const double* __restrict__ z, const double* __restrict__ yhat, double* __restrict__ b) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nblk; q += (i64)gridDim.x * blockDim.x) {
    int j[MAX_KINDS];
    double zz[MAX_KINDS], yy[MAX_KINDS];
#pragma unroll
    for (int a = 0; a < MAX_KINDS; ++a) j[a] = blk_idx[q * MAX_KINDS + a];
#pragma unroll
    for (int a = 0; a < MAX_KINDS; ++a) {   // every gather in flight before the first FMA
      zz[a] = j[a] >= 0 ? z[j[a]] : 0.0;
      yy[a] = (j[a] >= 0 && scheme == PG_SCHEME_CN) ? yhat[j[a]] : 0.0;
    }
    double v = c0[q];
#pragma unroll
    for (int a = 0; a < MAX_KINDS; ++a) v += wz[q * MAX_KINDS + a] * zz[a] - (scheme == PG_SCHEME_CN ? blk_cn[q * MAX_KINDS + a] * yy[a] : 0.0);
    b[blk_rows[q]] = v;
  }
}

// The same rows of a DIPHASIC system under Crank-Nicolson, matrix-free: b̂_r = Σ_a (B⁻¹S)[r,a] (c_a - M_a (A x)_a), with
// (A x)_a evaluated from the capacities (eval_row, the row the assembler would store, un-preconditioned) and x = S z.
// The subtraction happens in the row's own scale and B⁻¹ is applied once -- the table form (B⁻¹S)c - (B⁻¹MB)ŷ loses
// eps·cond(B) of b̂ there (M = diag(1,0,1,0) inside a cell: the product is no identity), 1e-8 in 3-D cut cells.
// One thread per block row evaluates the (<= 4) rows of its cell: 4x redundant, on a few thousand rows.
__global__ void k_rhs_block_mf(SysParams P, RowSegs seg, i64 Mloc, i64 nblk, const int* __restrict__ blk_rows,
                               const int* __restrict__ blk_idx, const double* __restrict__ blk_coef,
                               const int* __restrict__ row_cell, const int* __restrict__ red, const double* __restrict__ ds,
                               const double* __restrict__ z, const double* __restrict__ mass, const double* __restrict__ bconst,
                               const unsigned char* __restrict__ fixed, double* __restrict__ b) {
  const CapView& c = P.cap[0];
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nblk; q += (i64)gridDim.x * blockDim.x) {
    const i64 lc = row_cell[blk_rows[q]];
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    double v = 0.0;
    for (int a = 0; a < MAX_KINDS; ++a) {
      const int j = blk_idx[q * MAX_KINDS + a];
      if (j < 0) continue;
      const int k = seg_kind(seg, j);
      double ca;
      if (fixed[j]) {
        ca = bconst[j];                                         // jump rows, border rows: b = data
      } else {
        double ax = 0.0;                                        // (A x)_j, x = S z
        eval_row(P, k, lc, idx, [&](int ck, i64 cl, double val) {
          const int col = red[(i64)ck * Mloc + cl];
          if (val != 0.0 && col >= 0) ax += val * (ds[col] * z[col]);
        });
        ca = 2.0 * (mass[j] * (ds[j] * z[j])) + bconst[j] - ax;  // CN bulk row: (2 V - A) x + data      diffusion.jl:409-412
      }
      v += blk_coef[q * MAX_KINDS + a] * ca;
    }
    b[blk_rows[q]] = v;
  }
}

// The start of a quiet step extrapolated from older states.  The solver keeps the last R states z^{n-1} .. z^{n-R} and their
// products ŷ^{n-o} = Â z^{n-o} (ring buffers; no extra store: the buffers trade places with z / ŷ).  For a choice of
// offsets o_j, d_j = z^{n-o_j} - z^n and q_j = Â d_j = ŷ^{n-o_j} - ŷ^n cost two reads each, and
//   z_g = z^n + Σ c_j d_j   has the residual   r̂ = (b̂ - ŷ^n) - Σ c_j q_j   exactly (linearity; no product).
// Which (at most KH) offsets and which c_j: the weighted least-squares fit of the PREVIOUS step's plain residual on its
// q_j over a sample of the rows (k_guess_fit, which also weighs the products saved against the reads) -- a second pass for
// this step's own fit would cost what the fit saves, and the c_j move slowly from step to step.  Crank-Nicolson's
// alternating component makes the states of the SAME parity (offsets 1, 3, 5, 7) the useful ones, backward Euler the
// nearest ones; the fit finds that out.  Only the start changes: the iteration and its stopping test see r̂ as ever.
struct GuessArgs {
  const double* zr[8];      // z^{n-1} .. z^{n-R}   (the last one is also znew: a lane reads its element before writing it)
  const double* yr[8];      // ŷ^{n-1} ..
  double* znew;             // z^{n+1}'s buffer: z_g for the rows of the loop, the solved value for a row alone on its diagonal
  double* coef;             // [0..3] c_j, [4] how many, [5..8] their ring indices o_j - 1   (k_guess_fit); [13] += the number of
                            // older states this launch read (a running total for the bench's byte count)
  int defer;                // 1: z_g is NOT formed here -- the older states are not read for it and znew gets the rows alone on
                            // their diagonal only; the solve's first update of x writes z_g + α M⁻¹p (KrylovWork::xguess) from
                            // the copy of [0..8] this launch leaves at coef[GUESS_USED ..] (the fit rewrites [0..8] meanwhile)
};
constexpr int GUESS_USED = 64;

__device__ inline const double* ring_pick(const double* const (&r)[8], int i) {
  const double* p = r[0];
#pragma unroll
  for (int q = 1; q < 8; ++q) p = i == q ? r[q] : p;   // (uniform: scalar selects, no private-memory copy of the argument)
  return p;
}

// K8 + the BiCGStab start in one pass (loop form, warm start): from the scaled state z and ŷ = Âz
//   b̂ = S(2 mass∘(S z) + bconst) - ŷ  (CN) | S(mass∘(S z) + bconst) (BE) | S bconst (fixed rows) | as written by k_rhs_block
//   r = r̂ = p = b̂ - ŷ,  x = z (in place),  partial sums of (r,r) and (b̂,b̂) in slots 0 / 1   (k_rhs_init_c: r̂ only)
// replaces k_rhs + k_bicg_init: 9.3 instead of 14.1 vector passes, and no separate scaling kernels per step
typedef double rd2_t __attribute__((ext_vector_type(2)));
typedef unsigned char ruc2_t __attribute__((ext_vector_type(2)));

// one element of k_rhs_init
__device__ inline void rhs_init_one(int scheme, double z, double yh, double d, double ms, double bc, bool fx, bool blk, double bold,
                                    double& bi, double& ri) {
  if (blk) bi = bold;
  else if (fx) bi = d * bc;
  else if (scheme == PG_SCHEME_CN) bi = d * (2.0 * (ms * (d * z)) + bc) - yh;
  else bi = d * (ms * (d * z) + bc);
  ri = bi - yh;
}

// TWO elements per lane (16-byte accesses: half the vector-memory instructions of the 11 streams; the two flag bytes of
// a pair come with one 2-byte load each); the odd last element and the ghost tail are handled by the same lanes
// KH > 0: the start extrapolated from older states (GuessArgs; every row takes part: no row is left out on this path), the
// state written out of place
template <int KH>
__global__ __launch_bounds__(BLOCK) void k_rhs_init(i64 n, i64 nvec, int scheme, const double* __restrict__ z,
                                                    const double* __restrict__ yhat, const double* __restrict__ ds,
                                                    const double* __restrict__ mass, const double* __restrict__ bconst,
                                                    const unsigned char* __restrict__ fixed,
                                                    const unsigned char* __restrict__ isblk, double* __restrict__ b,
                                                    double* __restrict__ r, double* __restrict__ rhat,
                                                    double* __restrict__ p, double* __restrict__ partials, GuessArgs g) {
  __shared__ double s_red[BLOCK / 64];
  double acc = 0.0, accb = 0.0, accw = 0.0;   // (r,r), (b,b)_W, (r,r)_W: slots 0, 1, 2 as k_bicg_init (weights ds²)
  double cj[KH > 0 ? KH : 1];
  const double* zo[KH > 0 ? KH : 1];
  const double* yo[KH > 0 ? KH : 1];
  int ku = 0;
  if (KH > 0) {
    ku = min(KH, (int)g.coef[4]);
#pragma unroll
    for (int j = 0; j < KH; ++j) {
      const int sel = j < ku ? (int)g.coef[5 + j] : 0;
      cj[j] = j < ku ? g.coef[j] : 0.0;
      zo[j] = ring_pick(g.zr, sel);
      yo[j] = ring_pick(g.yr, sel);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) g.coef[13] += (double)ku;
  }
  const i64 npair = (nvec + 1) / 2;
  for (i64 q = blockIdx.x * (i64)BLOCK + threadIdx.x; q < npair; q += (i64)gridDim.x * BLOCK) {
    const i64 i = 2 * q;
    if (i + 1 < n) {
      // (stream hints: the per-row data is read once per step and b̂ is only kept for inspection -- neither should
      //  displace the SpMV's matrix data from the Infinity Cache)
      const rd2_t yh = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(yhat + i));
      const rd2_t d = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(ds + i));
      const rd2_t bc = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(bconst + i));
      const rd2_t ms = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(mass + i));
      const rd2_t zz = *reinterpret_cast<const rd2_t*>(z + i);
      const ruc2_t fx = *reinterpret_cast<const ruc2_t*>(fixed + i);
      const ruc2_t bk = *reinterpret_cast<const ruc2_t*>(isblk + i);
      rd2_t bold;
      bold.x = 0.0; bold.y = 0.0;
      if (bk.x | bk.y) bold = *reinterpret_cast<const rd2_t*>(b + i);
      double b0, r0, b1, r1;
      rhs_init_one(scheme, zz.x, yh.x, d.x, ms.x, bc.x, fx.x != 0, bk.x != 0, bold.x, b0, r0);
      rhs_init_one(scheme, zz.y, yh.y, d.y, ms.y, bc.y, fx.y != 0, bk.y != 0, bold.y, b1, r1);
      if (KH > 0) {
        rd2_t zg = zz;
#pragma unroll
        for (int j = 0; j < KH; ++j) {
          if (j < ku) {
            const rd2_t a = *reinterpret_cast<const rd2_t*>(zo[j] + i);
            const rd2_t y2 = *reinterpret_cast<const rd2_t*>(yo[j] + i);
            r0 -= cj[j] * (y2.x - yh.x); r1 -= cj[j] * (y2.y - yh.y);
            zg.x += cj[j] * (a.x - zz.x); zg.y += cj[j] * (a.y - zz.y);
          }
        }
        *reinterpret_cast<rd2_t*>(g.znew + i) = zg;
      }
      rd2_t bi, ri;
      bi.x = b0; bi.y = b1; ri.x = r0; ri.y = r1;
      __builtin_nontemporal_store(bi, reinterpret_cast<rd2_t*>(b + i));
      *reinterpret_cast<rd2_t*>(r + i) = ri;
      *reinterpret_cast<rd2_t*>(rhat + i) = ri;
      *reinterpret_cast<rd2_t*>(p + i) = ri;
      acc += ri.x * ri.x + ri.y * ri.y;
      accb += (d.x * bi.x) * (d.x * bi.x) + (d.y * bi.y) * (d.y * bi.y);
      accw += (d.x * ri.x) * (d.x * ri.x) + (d.y * ri.y) * (d.y * ri.y);
    } else {
      for (i64 k = i; k < i + 2 && k < nvec; ++k) {
        if (k < n) {
          double bi, ri;
          rhs_init_one(scheme, z[k], yhat[k], ds[k], mass[k], bconst[k], fixed[k] != 0, isblk[k] != 0, isblk[k] ? b[k] : 0.0, bi, ri);
          if (KH > 0) {
            double zg = z[k];
#pragma unroll
            for (int j = 0; j < KH; ++j)
              if (j < ku) { ri -= cj[j] * (yo[j][k] - yhat[k]); zg += cj[j] * (zo[j][k] - z[k]); }
            g.znew[k] = zg;
          }
          b[k] = bi;
          r[k] = ri; rhat[k] = ri; p[k] = ri;
          acc += ri * ri;
          accb += (ds[k] * bi) * (ds[k] * bi);
          accw += (ds[k] * ri) * (ds[k] * ri);
        } else {
          p[k] = 0.0;
        }
      }
    }
  }
  const double t = block_sum(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  const double tb = block_sum(accb, s_red);
  if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = tb;
  const double tw = block_sum(accw, s_red);
  if (threadIdx.x == 0) partials[2 * (size_t)gridDim.x + blockIdx.x] = tw;
}

// k_rhs_init for a COMPACT loop system (pg_reduce.hip, DiagElim): cmap[i] >= 0: the row stays, r = r̂ = p go to that index of
// the compact vectors and count in the start sums; cmap[i] = -1 - e: the row is alone on its diagonal (entry e of gdiag /
// delta): solved on the spot, left out of the sums.  b̂ is written for every row, (b̂,b̂)_W sums over all of them.
// KH > 0: the extrapolated start above (out of place: the state written is g.znew), up to KH older states.
template <int KH>
__global__ __launch_bounds__(BLOCK) void k_rhs_init_c(i64 n, int scheme, const double* z,
                                                      const double* __restrict__ yhat, const double* __restrict__ ds,
                                                      const double* __restrict__ mass, const double* __restrict__ bconst,
                                                      const unsigned char* __restrict__ fixed,
                                                      const unsigned char* __restrict__ isblk, const int* __restrict__ cmap,
                                                      double* __restrict__ b, const double* __restrict__ gdiag,
                                                      double* __restrict__ delta, int* __restrict__ flag, int stamp,
                                                      double* __restrict__ rhat, double* __restrict__ partials,
                                                      unsigned* __restrict__ ticket, double* __restrict__ sc, double reltol2,
                                                      double abstol2, GuessArgs g) {
  // (p = r = r̂ are NOT written: the first iteration reads r̂ for them, KrylovWork::p_in_rhat)
  // ticket != NULL: the solver's start phase (scalar reset, start sums, PH_INIT) by the last block of this launch -- the
  // caller knows that nothing will touch the start sums afterwards (no diagonal row can move: diag_fix is not launched)
  __shared__ double s_red[BLOCK / 64];
  double acc = 0.0, accb = 0.0, accw = 0.0;
  double cj[KH > 0 ? KH : 1];
  const double* zo[KH > 0 ? KH : 1];
  const double* yo[KH > 0 ? KH : 1];
  int ku = 0;
  const bool defer = KH > 0 && g.defer != 0;
  if (KH > 0) {
    ku = min(KH, (int)g.coef[4]);
#pragma unroll
    for (int j = 0; j < KH; ++j) {
      const int sel = j < ku ? (int)g.coef[5 + j] : 0;
      cj[j] = j < ku ? g.coef[j] : 0.0;
      zo[j] = ring_pick(g.zr, sel);
      yo[j] = ring_pick(g.yr, sel);
    }
    if (defer && blockIdx.x == 0 && threadIdx.x < 9) g.coef[GUESS_USED + threadIdx.x] = g.coef[threadIdx.x];
  }
  int moved = 0;
  double* zw = KH > 0 ? g.znew : const_cast<double*>(z);   // (in place: a lane writes only elements it has read itself)
  // zh / yv: the row's entries of the chosen older states and products
  auto one = [&](i64 i, double zi, double yh, double d, double ms, double bc, bool fx, bool blk, double bold, int c, double& bi,
                 const double* zh, const double* yv) {
    double ri;
    rhs_init_one(scheme, zi, yh, d, ms, bc, fx, blk, bold, bi, ri);
    accb += (d * bi) * (d * bi);
    if (c >= 0) {
      if (KH > 0) {
        double zg = zi, corr = 0.0;
#pragma unroll
        for (int j = 0; j < KH; ++j) {
          corr += cj[j] * (yv[j] - yh);      // (cj = 0, yv = yh, zh = zi beyond ku)
          zg += cj[j] * (zh[j] - zi);
        }
        ri -= corr;
        if (!defer) zw[i] = zg;
      }
      rhat[c] = ri;
      acc += ri * ri;
      accw += (d * ri) * (d * ri);
    } else {
      // a row alone on its diagonal: solved here, x += δ = r / a_ii (δ = 0 once the row's datum has stopped changing);
      // δ is what the rows coupled to it need (diag_fix), the flag says whether any δ of this step is non-zero
      const int e = -1 - c;
      const double dl = ri / gdiag[e];
      delta[e] = dl;
      if (KH > 0 || dl != 0.0) zw[i] = zi + dl;
      if (fabs(dl) > 1e-12 * fabs(zi)) moved = 1;
    }
  };
  // two rows per lane, 16-byte loads of the seven input streams (as k_rhs_init); the compact stores are 8-byte
  const i64 npair = (n + 1) / 2;
  for (i64 q = blockIdx.x * (i64)BLOCK + threadIdx.x; q < npair; q += (i64)gridDim.x * BLOCK) {
    const i64 i = 2 * q;
    double zh0[KH > 0 ? KH : 1], zh1[KH > 0 ? KH : 1], yv0[KH > 0 ? KH : 1], yv1[KH > 0 ? KH : 1];
    if (i + 1 < n) {
      const rd2_t yh = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(yhat + i));
      const rd2_t d = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(ds + i));
      const rd2_t bc = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(bconst + i));
      const rd2_t ms = __builtin_nontemporal_load(reinterpret_cast<const rd2_t*>(mass + i));
      const rd2_t zz = *reinterpret_cast<const rd2_t*>(z + i);
      const ruc2_t fx = *reinterpret_cast<const ruc2_t*>(fixed + i);
      const ruc2_t bk = *reinterpret_cast<const ruc2_t*>(isblk + i);
      const int c0 = cmap[i], c1 = cmap[i + 1];
      if (KH > 0) {
#pragma unroll
        for (int j = 0; j < KH; ++j) {
          rd2_t a = zz, y2 = yh;
          if (j < ku && (c0 >= 0 || c1 >= 0)) {     // (rows alone on their diagonal take nothing from the older states)
            if (!defer) a = *reinterpret_cast<const rd2_t*>(zo[j] + i);
            y2 = *reinterpret_cast<const rd2_t*>(yo[j] + i);
          }
          zh0[j] = a.x; zh1[j] = a.y; yv0[j] = y2.x; yv1[j] = y2.y;
        }
      }
      rd2_t bold;
      bold.x = 0.0; bold.y = 0.0;
      if (bk.x | bk.y) bold = *reinterpret_cast<const rd2_t*>(b + i);
      rd2_t bi;
      double b0, b1;
      one(i, zz.x, yh.x, d.x, ms.x, bc.x, fx.x != 0, bk.x != 0, bold.x, c0, b0, zh0, yv0);
      one(i + 1, zz.y, yh.y, d.y, ms.y, bc.y, fx.y != 0, bk.y != 0, bold.y, c1, b1, zh1, yv1);
      bi.x = b0; bi.y = b1;
      __builtin_nontemporal_store(bi, reinterpret_cast<rd2_t*>(b + i));
    } else if (i < n) {
      const double zi = z[i], yh = yhat[i];
      if (KH > 0) {
#pragma unroll
        for (int j = 0; j < KH; ++j) {
          zh0[j] = (j < ku && !defer) ? zo[j][i] : zi;
          yv0[j] = j < ku ? yo[j][i] : yh;
        }
      }
      double b0;
      one(i, zi, yh, ds[i], mass[i], bconst[i], fixed[i] != 0, isblk[i] != 0, isblk[i] ? b[i] : 0.0, cmap[i], b0, zh0, yv0);
      b[i] = b0;
    }
  }
  if (__any(moved) && (threadIdx.x & 63) == 0) atomicMax(flag, stamp);
  const double t = block_sum(acc, s_red);
  const double tb = block_sum(accb, s_red);
  const double tw = block_sum(accw, s_red);
  if (!ticket) {
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = t;
      partials[gridDim.x + blockIdx.x] = tb;
      partials[2 * (size_t)gridDim.x + blockIdx.x] = tw;
      if (KH > 0 && blockIdx.x == 0) g.coef[13] += (double)ku;
    }
    return;
  }
  if (threadIdx.x == 0) {
    pg::store_partial(partials + blockIdx.x, t);
    pg::store_partial(partials + gridDim.x + blockIdx.x, tb);
    pg::store_partial(partials + 2 * (size_t)gridDim.x + blockIdx.x, tw);
  }
  if (!pg::last_block_arrives(ticket, gridDim.x, s_red)) return;
  // (k_start's work, in its order of summation)
  if (threadIdx.x < pg::S_COUNT) sc[threadIdx.x] = threadIdx.x == pg::S_RELTOL2 ? reltol2 : (threadIdx.x == pg::S_ABSTOL2 ? abstol2 : 0.0);
  __syncthreads();
  for (int sl = 0; sl < 3; ++sl) {
    double a = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += BLOCK) a += pg::load_partial(partials + (size_t)sl * gridDim.x + i);
    const double sum = block_sum(a, s_red);
    if (threadIdx.x == 0) sc[pg::S_RED0 + sl] = sum;
  }
  if (threadIdx.x == 0) {
    pg::derive(pg::PH_INIT, sc);
    // (every block's atomicMax on the flag was drained before it drew its ticket)
    sc[pg::S_MOVED] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == stamp ? 1.0 : 0.0;
    if (KH > 0) g.coef[13] += (double)ku;
  }
}

// The fit behind the extrapolated start (GuessArgs): over every `stride`-th chunk of BLOCK rows (a few hundred blocks share
// them), the normal equations of
//   min || r_u - Σ c_j q_j ||_W ,  r_u = b̂ - ŷ^n (the plain start residual), q_j = ŷ^{n-j-1} - ŷ^n, j < avail <= RM
// (Gram matrix, right-hand side, (r_u,r_u)_W and (r,r)_W of the start actually taken).  The last block then tries every
// subset of at most kmax of the avail older states -- one thread each, a Cholesky factorisation of at most 4 x 4 -- and
// keeps the one whose saving,  gain · log((r_u,r_u) / (left,left)) / rate2  products, exceeds its reads (2 passes per
// state, pass_cost products each) by most: offsets and coefficients for the NEXT step.
constexpr int GUESS_RM = 7;
constexpr int GUESS_NS = GUESS_RM * (GUESS_RM + 1) / 2 + GUESS_RM + 2;
struct GuessFit {
  const double* yr[8];
  double* partials;       // GUESS_NS * grid
  unsigned* ticket;
  double* coef;           // [0..3] c_j, [4] how many, [5..8] ring indices, [9] (r_u,r_u)_W, [10] (r,r)_W, [11] the fit's left-over
  int stride, avail, kmax;
  double rate2, pass_cost, gain;
  double* sums;           // several ranks: where the GUESS_NS local sums go (NULL: one rank, the launch decides itself)
};

__device__ inline int guess_tri(int j, int l) { return j * GUESS_RM - j * (j - 1) / 2 + (l - j); }   // l >= j

// the choice among the subsets (k_guess_fit's doc), from the sums s_g (GUESS_NS of them, in LDS), by the whole block
__device__ inline void guess_decide(const double* s_g, const GuessFit& f) {
  constexpr int RM = GUESS_RM, NS = GUESS_NS;
  __shared__ double s_left[128], s_net[128];
  __shared__ double s_c[128][4];
  const double ru = s_g[NS - 2], rw = s_g[NS - 1];
  // one subset of the older states per thread: bit j of the thread number = state j
  if (threadIdx.x < 128) {
    const int mask = threadIdx.x, m = __popc(mask);
    double left = ru, cc[4] = {0.0, 0.0, 0.0, 0.0};
    if (mask != 0 && m <= f.kmax && m <= 4 && (mask >> f.avail) == 0 && ru > 0.0) {
      int idx[4] = {0, 0, 0, 0};
      for (int j = 0, a = 0; j < RM; ++j) if (mask >> j & 1) { if (a < 4) idx[a] = j; ++a; }
      double L[4][4], wv[4];
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        wv[a] = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) L[a][e] = 0.0;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (a >= m || !ok) continue;
        const double g0 = s_g[guess_tri(idx[a], idx[a])];
        double dj = g0;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < a) dj -= L[a][e] * L[a][e];
        if (!(g0 > 0.0) || !(dj > 1e-12 * g0)) { ok = false; continue; }   // (collinear with the ones before it)
        const double lj = sqrt(dj);
        L[a][a] = lj;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          if (l <= a || l >= m) continue;
          double v = s_g[guess_tri(idx[a], idx[l])];
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e < a) v -= L[l][e] * L[a][e];
          L[l][a] = v / lj;
        }
        double v = s_g[RM * (RM + 1) / 2 + idx[a]];
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < a) v -= L[a][e] * wv[e];
        wv[a] = v / lj;
        left -= wv[a] * wv[a];
      }
      if (ok) {
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {            // Lᵀ c = w
          const int a = 3 - aa;
          if (a >= m) continue;
          double v = wv[a];
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e > a && e < m) v -= L[e][a] * cc[e];
          cc[a] = v / L[a][a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) ok = ok && isfinite(cc[a]) && fabs(cc[a]) < 256.0;
      }
      if (!ok) { left = ru; cc[0] = cc[1] = cc[2] = cc[3] = 0.0; }
      if (!(left > 1e-30 * ru)) left = 1e-30 * ru;
    }
    s_left[threadIdx.x] = left;
#pragma unroll
    for (int a = 0; a < 4; ++a) s_c[threadIdx.x][a] = cc[a];
    // what the subset nets: the products its fit saves less its reads (a start that the extrapolation made worse: none
    // next time)
    const bool live = ru > 0.0 && f.rate2 > 0.0 && rw <= ru * (1.0 + 1e-12) && left < ru;
    s_net[threadIdx.x] = live ? f.gain * log(ru / left) / f.rate2 - 2.0 * m * f.pass_cost : 0.0;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    // the subset that nets most (ties: the lower number), by the first wave
    int pick = threadIdx.x;
    double best = s_net[pick];
    if (s_net[pick + 64] > best) { best = s_net[pick + 64]; pick += 64; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_down(best, off, 64);
      const int op = __shfl_down(pick, off, 64);
      if (ob > best || (ob == best && op < pick)) { best = ob; pick = op; }
    }
    if (!(best > 0.0)) pick = 0;
    if (threadIdx.x != 0) return;
    int a = 0;
    for (int j = 0; j < RM; ++j)
      if (pick >> j & 1) { f.coef[a] = s_c[pick][a]; f.coef[5 + a] = (double)j; ++a; }
    f.coef[4] = (double)a;
    for (; a < 4; ++a) { f.coef[a] = 0.0; f.coef[5 + a] = 0.0; }
    f.coef[9] = ru; f.coef[10] = rw; f.coef[11] = pick ? s_left[pick] : ru;
  }
}

__global__ __launch_bounds__(BLOCK) void k_guess_fit(i64 n, const int* __restrict__ cmap, const double* __restrict__ ds,
                                                     const double* __restrict__ b, const double* __restrict__ yn,
                                                     const double* __restrict__ rhat, GuessFit f) {
  constexpr int RM = GUESS_RM, NS = GUESS_NS;
  __shared__ double s_red[BLOCK / 64];
  __shared__ double s_g[NS];
  double gs[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) gs[j] = 0.0;
  const i64 nchunk = (n + BLOCK - 1) / BLOCK;
  for (i64 ch = (i64)blockIdx.x * f.stride; ch < nchunk; ch += (i64)gridDim.x * f.stride) {
    const i64 i = ch * BLOCK + threadIdx.x;
    const int c = i < n ? (cmap ? cmap[i] : (int)i) : -1;   // (no map: every row is in the loop)
    if (c < 0) continue;
    const double d = ds[i], w = d * d, y0 = yn[i], ru = b[i] - y0, ra = rhat[c];
    double q[RM];
#pragma unroll
    for (int j = 0; j < RM; ++j) q[j] = j < f.avail ? f.yr[j][i] - y0 : 0.0;
    int t = 0;
#pragma unroll
    for (int j = 0; j < RM; ++j)
#pragma unroll
      for (int l = j; l < RM; ++l) gs[t++] += w * q[j] * q[l];
#pragma unroll
    for (int j = 0; j < RM; ++j) gs[t++] += w * q[j] * ru;
    gs[t++] += w * ru * ru;
    gs[t] += w * ra * ra;
  }
  {
    __shared__ double s_w[BLOCK / 64][NS];     // all the sums through one barrier
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (the shuffle tree level by level over all the sums: NS independent shuffles in flight instead of one chain each)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int j = 0; j < NS; ++j) gs[j] += __shfl_down(gs[j], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < NS; ++j) s_w[wave][j] = gs[j];
    }
    __syncthreads();
    if (threadIdx.x < NS) {
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < BLOCK / 64; ++q) a += s_w[q][threadIdx.x];
      pg::store_partial(f.partials + (size_t)threadIdx.x * gridDim.x + blockIdx.x, a);
    }
  }
  if (!pg::last_block_arrives(f.ticket, gridDim.x, s_red)) return;
  // the blocks' partial sums, in a fixed order: a wave per sum, a lane per block (at most 128 blocks), shuffle tree -- every
  // load of a wave in flight before the first is used, the trees level by level
  {
    constexpr int PW = (NS + BLOCK / 64 - 1) / (BLOCK / 64);   // sums per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, G = (int)gridDim.x;
    double a0[PW], a1[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const int sl = wave + k * (BLOCK / 64);
      a0[k] = (sl < NS && lane < G) ? pg::load_partial(f.partials + (size_t)sl * G + lane) : 0.0;
      a1[k] = (sl < NS && lane + 64 < G) ? pg::load_partial(f.partials + (size_t)sl * G + lane + 64) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < PW; ++k) a0[k] += a1[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int k = 0; k < PW; ++k) a0[k] += __shfl_down(a0[k], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const int sl = wave + k * (BLOCK / 64);
        if (sl < NS) s_g[sl] = a0[k];
      }
    }
  }
  __syncthreads();
  if (f.sums) {                     // several ranks: the sums go through an all-reduce, k_guess_decide follows
    if (threadIdx.x < NS) f.sums[threadIdx.x] = s_g[threadIdx.x];
    return;
  }
  guess_decide(s_g, f);
}

__global__ __launch_bounds__(BLOCK) void k_guess_decide(GuessFit f) {
  __shared__ double s_g[GUESS_NS];
  if (threadIdx.x < GUESS_NS) s_g[threadIdx.x] = f.sums[threadIdx.x];
  __syncthreads();
  guess_decide(s_g, f);
}

// z refers to S_old: re-express it with S_new (x = S_old z = S_new z')
__global__ void k_rescale_state(i64 n, const double* __restrict__ ds_old, const double* __restrict__ ds_new,
                                double* __restrict__ z) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x)
    z[r] = (ds_old[r] * z[r]) / ds_new[r];
}

__global__ void k_scale_state(i64 n, const double* __restrict__ ds, const double* __restrict__ in, double* __restrict__ out,
                              int divide) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x)
    out[r] = divide ? in[r] / ds[r] : in[r] * ds[r];
}

// K8, constructor form: T0 and K*T0 live in the padded layout (T0 may be non-zero at eliminated unknowns)
__global__ void k_rhs_first(RowSegs seg, i64 n, i64 Mloc, int scheme, const int* row_cell, const double* T0pad,
                            const double* ypad, const double* mass, const double* bconst,
                            const unsigned char* fixed, double* braw, double* x0, int moving) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    const int k = seg_kind(seg, r);
    const i64 q = (i64)k * Mloc + row_cell[r];
    const double T = T0pad[q];
    double v;
    if (fixed[r]) v = bconst[r];
    else if (moving && (k & 1)) v = bconst[r];                                        // b2 = Γ g
    else if (moving && scheme == PG_SCHEME_CN) v = mass[r] * T - ypad[q] + bconst[r];   // (Vn - Id GᵀWꜝGΨn)Tω - ½Id GᵀWꜝH Tγ + ...
    else if (scheme == PG_SCHEME_CN) v = 2.0 * (mass[r] * T) - ypad[q] + bconst[r];
    else v = mass[r] * T + bconst[r];
    braw[r] = v;
    x0[r] = T;
  }
}

// Ψn1 = psip.(Vn, Vn_1), Ψn = psim.(Vn, Vn_1)      prescribedmotionsolver/diffusion.jl:55-98 (the live definitions)
__global__ void k_psi(i64 Mloc, int scheme, const double* __restrict__ vn_1 /*lower face*/, const double* __restrict__ vn,
                      double* __restrict__ psip, double* __restrict__ psim) {
  for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < Mloc; i += (i64)gridDim.x * blockDim.x) {
    const bool z1 = vn[i] == 0.0, z2 = vn_1[i] == 0.0;   // args = (Vn, Vn_1)
    double pp, pm;
    if (scheme == PG_SCHEME_CN) {
      pp = (z1 && z2) ? 0.0 : (!z1 && !z2) ? 0.5 : (z1 && !z2) ? 0.5 : 1.0;   // psip_cn
      pm = (z1 && z2) ? 0.0 : (!z1 && !z2) ? 0.5 : (z1 && !z2) ? 0.5 : 0.0;   // psim_cn
    } else {
      pp = (z1 && z2) ? 0.0 : 1.0;                                            // psip_be
      pm = 0.0;                                                               // psim_be
    }
    psip[i] = pp;
    psim[i] = pm;
  }
}

// K12: padded[k*Mloc + cell] = x[r]
__global__ void k_scatter_active(RowSegs seg, i64 n, i64 Mloc, const int* row_cell, const double* x, double* padded) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    const int k = seg_kind(seg, r);
    padded[(i64)k * Mloc + row_cell[r]] = x[r];
  }
}

__global__ void k_maxabs(i64 n, const double* x, const double* ds /*NULL: x is unscaled*/, double* partials) {
  __shared__ double sh[BLOCK];
  double m = 0.0;
  for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const double a = fabs(ds ? ds[i] * x[i] : x[i]);
    m = a > m ? a : m;
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] > sh[threadIdx.x + s] ? sh[threadIdx.x] : sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

__global__ void k_set_border_values(i64 nb, const i64* cells_global, const double* values, i64 first_cell, i64 Mloc,
                                    int nbulk, const int* red, i64 n_own, double* bcv) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nb; q += (i64)gridDim.x * blockDim.x) {
    const i64 lc = cells_global[q] - first_cell;
    if (lc < 0 || lc >= Mloc) continue;
    for (int ph = 0; ph < nbulk; ++ph) {
      const int r = red[(i64)(2 * ph) * Mloc + lc];
      if (r >= 0 && r < n_own) bcv[r] = values[q];
    }
  }
}

SysParams make_params(const pg_solver* s, int scheme) {
  SysParams P;
  std::memset(&P, 0, sizeof(P));
  P.nphase = s->nphase;
  for (int q = 0; q < s->nphase; ++q) {
    P.cap[q] = cap_view(s->cap[q]);
    P.ct[q] = s->cap[q]->ct.p;
    P.Id[q] = s->Id[q].p;
    if (s->ops[q] && s->ops[q]->has_velocity) {
      for (int d = 0; d < s->cap[q]->N; ++d) P.conv_a[q][d] = s->ops[q]->conv_a[d].p;
      P.conv_k[q] = s->ops[q]->conv_k.p;
    }
  }
  if (s->nphase == 1) { P.cap[1] = P.cap[0]; P.ct[1] = P.ct[0]; P.Id[1] = nullptr; }
  // steady (diffusion.jl:30-43): the unsteady BE blocks with Δt = 1 and without the V term
  const double th = scheme == PG_SCHEME_CN ? s->dt / 2 : s->dt;
  P.theta = th;
  P.gscale = scheme == PG_SCHEME_CN ? s->dt / 2 : 1.0;
  P.mass = scheme == PG_SCHEME_STEADY ? 0.0 : 1.0;
  if (s->nphase == 1) {
    switch (s->bc_i.kind) {     // build_I_bc, solver.jl:203-223
      case PG_BC_DIRICHLET: P.Ia = 1.0; P.Ib = 0.0; break;
      case PG_BC_NEUMANN: P.Ia = 0.0; P.Ib = 1.0; break;
      case PG_BC_ROBIN: P.Ia = s->bc_i.alpha; P.Ib = s->bc_i.beta; break;
      default: P.Ia = 0.0; P.Ib = 0.0; break;
    }
  } else {
    P.a1 = s->ic.alpha1; P.a2 = s->ic.alpha2; P.b1 = s->ic.beta1; P.b2 = s->ic.beta2;
  }
  for (int k = 0; k < 6; ++k) P.border_kind[k] = s->border_kind[k];
  P.inv_dx = s->inv_dx;
  P.border_both_phases = (s->moving && s->nphase == 2) ? 1 : 0;
  if (s->moving) {            // Δt lives inside the space-time capacities
    P.theta = 1.0;
    P.gscale = 1.0;
    P.mass = 1.0;
    for (int q = 0; q < s->nphase; ++q) {
      P.mv_v0[q] = s->cap[q]->Vt[0].p;
      P.mv_v1[q] = s->cap[q]->Vt[1].p;
      P.mv_psi_w[q] = s->psi_p[q].p;
      P.mv_psi_g[q] = s->psi_p[q].p;
    }
    if (s->nphase == 1) { P.mv_v0[1] = P.mv_v0[0]; P.mv_v1[1] = P.mv_v1[0]; P.mv_psi_w[1] = P.mv_psi_w[0]; P.mv_psi_g[1] = P.mv_psi_g[0]; }
  }
  return P;
}

// the explicit operator of the moving Crank-Nicolson right-hand side (diffusion.jl:214)
SysParams make_params_moving_explicit(const pg_solver* s) {
  SysParams P = make_params(s, PG_SCHEME_CN);
  for (int q = 0; q < 2; ++q) {
    const int src = q < s->nphase ? q : 0;
    P.mv_psi_w[q] = s->psi_m[src].p;
    // mono (:214): the γ term is -½ Id GᵀWꜝH Tγ; diph (:487-488): -Id GᵀWꜝH Ψn Tγ with the same Ψn = psim as the ω term
    P.mv_psi_g[q] = s->nphase == 2 ? s->psi_m[src].p : nullptr;
  }
  P.mv_gconst = 0.5;
  P.mv_explicit = 1;
  return P;
}

void upload_local(DevBuf<double>& dst, const double* host_global, const Slab& slab) {
  dst.alloc(slab.Mloc());
  dst.upload(host_global + slab.first_cell(), slab.Mloc());
}

SrcView src_view(const pg_solver* s) {
  SrcView v;
  for (int q = 0; q < 2; ++q) { v.f_n[q] = s->f_n[q].p; v.f_np1[q] = s->f_np1[q].p; }
  if (s->nphase == 1) {
    v.g_n = s->g_n.p; v.g_np1 = s->g_np1.p;
    v.g_const = s->bc_i.value; v.h_const = 0.0;
  } else {
    v.g_n = s->g_arr.p; v.g_np1 = s->h_arr.p;
    v.g_const = s->ic.g; v.h_const = s->ic.h;
  }
  return v;
}

void ensure_bconst(pg_solver* s, int scheme) {
  if (!s->bconst_dirty && s->bconst_scheme == scheme) return;
  hipStream_t st = ctx().stream;
  const i64 n = s->nb.n_own;
  if (n > 0) {
    const SysParams P = make_params(s, scheme);
    hipLaunchKernelGGL(k_bconst, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, st, P, make_segs(s->nb), n, s->nb.row_cell.p,
                       src_view(s), scheme, s->dt, s->fixed.p, s->bcv.p, s->bconst.p, s->moving ? 1 : 0);
    PG_HIP(hipGetLastError());
  }
  s->bconst_scheme = scheme;
  s->bconst_dirty = false;
  ++s->bconst_version;
}

// first right-hand side: b(t = 0) with the constructor scheme (diffusion.jl:202/205, 328); re-run when the host
// supplies per-cell data after construction but before the first solve
void build_first_rhs(pg_solver* s) {
  hipStream_t st = ctx().stream;
  const i64 n = s->nb.n_own;
  const SysParams P = make_params(s, s->scheme_ctor);
  ensure_bconst(s, s->scheme_ctor);
  DevBuf<double> ypad;
  if (s->scheme_ctor == PG_SCHEME_CN) {
    ypad.alloc((i64)s->K * s->Mloc);
    ypad.zero();
    apply_rows_padded(s->moving ? make_params_moving_explicit(s) : P, s->slab, s->T0pad.p, ypad.p);
  }
  if (n > 0) {
    DevBuf<double> braw(n);
    hipLaunchKernelGGL(k_rhs_first, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, st, make_segs(s->nb), n, s->Mloc,
                       s->scheme_ctor, s->nb.row_cell.p, s->T0pad.p, ypad.p, s->mass.p, s->bconst.p, s->fixed.p, braw.p,
                       s->x.p, s->moving ? 1 : 0);
    PG_HIP(hipGetLastError());
    apply_left(s->A_ctor, braw.p, s->b.p, st);   // b̂ = B⁻¹ S b
    PG_HIP(hipStreamSynchronize(st));
  }
  PG_HIP(hipStreamSynchronize(st));
}

void materialize_x(pg_solver* s);

void setup_common(pg_solver* s, const pg_border_desc* borders, int nborders, const double* T0) {
  hipStream_t st = ctx().stream;
  const Slab& slab = s->slab;
  s->K = s->nphase == 1 ? 2 : 4;
  s->Mloc = slab.Mloc();
  s->M = slab.M;
  for (int i = 0; i < nborders; ++i) {
    const int key = borders[i].key;
    PG_REQUIRE(key >= 0 && key < 6, "border key out of range");
    const int kind = borders[i].kind;
    PG_REQUIRE(kind == PG_BC_DIRICHLET || kind == PG_BC_PERIODIC || kind == PG_BC_NEUMANN || kind == PG_BC_NONE ||
               kind == PG_BC_ROBIN, "border kind not supported");
    // Robin (and Neumann in >1-D) borders are silently ignored by the reference (solver.jl:450-499)
    s->border_kind[key] = (kind == PG_BC_ROBIN) ? PG_BC_NONE : kind;
    s->border_value[key] = borders[i].value;
  }
  if (slab.N == 1) {
    const auto& nd = s->cap[0]->mesh->nodes[0];
    double dx = 1e300;
    for (size_t i = 1; i < nd.size(); ++i) dx = std::min(dx, nd[i] - nd[i - 1]);
    s->inv_dx = 1.0 / dx;   // solver.jl:476
  }
  if (ctx().nranks > 1)
    for (int k = 0; k < 6; ++k)
      PG_REQUIRE(s->border_kind[k] != PG_BC_PERIODIC, "periodic borders are single-rank only");

  // initial condition, padded local layout
  s->T0pad.alloc((i64)s->K * s->Mloc);
  s->T0pad.zero();   // T0 == NULL: zeros(2M) without a host array (multi-GPU sizes)
  if (T0)
    for (int k = 0; k < s->K; ++k) s->T0pad.upload(T0 + (i64)k * s->M + slab.first_cell(), s->Mloc, (i64)k * s->Mloc);
  if (s->init_from) {     // the previous slab's state, device to device (zeros at its eliminated unknowns, solver.jl:186-187)
    pg_solver* p = s->init_from;
    PG_REQUIRE(p->K == s->K && p->Mloc == s->Mloc && p->M == s->M, "previous solver lives on another mesh");
    materialize_x(p);
    if (p->nb.n_own > 0)
      hipLaunchKernelGGL(k_scatter_active, dim3(grid_for(p->nb.n_own, BLOCK)), dim3(BLOCK), 0, st, make_segs(p->nb), p->nb.n_own,
                         p->Mloc, p->nb.row_cell.p, p->x.p, s->T0pad.p);
    PG_HIP(hipGetLastError());
    s->init_from = nullptr;
  }

  // K10 once, K7/K9 for the constructor scheme
  const SysParams P = make_params(s, s->scheme_ctor);
  Laps laps;
  build_numbering(P, slab, s->nb);
  laps.lap("build_numbering");
  assemble_csr_preconditioned(P, slab, s->nb, s->A_ctor);
  laps.lap("assemble_csr_preconditioned");
  s->A_ctor.scheme = s->scheme_ctor;

  const i64 n = s->nb.n_own, nv = s->nb.n_vec();
  const i64 na = n > 0 ? n : 1, nva = nv > 0 ? nv : 1;
  s->mass.alloc(na); s->bconst.alloc(na); s->bcv.alloc(na); s->fixed.alloc(na);
  s->x.alloc(nva); s->b.alloc(nva); s->y.alloc(nva); s->z.alloc(nva); s->ysol.alloc(nva);
  s->x.zero(); s->b.zero(); s->y.zero(); s->z.zero(); s->ysol.zero();
  s->work.init(n, nv);
  s->red_scratch.alloc(2048);
  if (n > 0) {
    KeyVals kv;
    for (int k = 0; k < 6; ++k) kv.v[k] = s->border_value[k];
    hipLaunchKernelGGL(k_row_static, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, st, P, make_segs(s->nb), n, s->nb.row_cell.p,
                       kv, s->mass.p, s->fixed.p, s->bcv.p);
    PG_HIP(hipGetLastError());
  }
  laps.lap("row data, vectors");
  build_first_rhs(s);
  laps.lap("first right-hand side");
  s->t = 0.0;
}

double max_abs(pg_solver* s) {
  hipStream_t st = ctx().stream;
  const i64 n = s->nb.n_own;
  double m = 0.0;
  if (n > 0) {
    const int g = grid_for(n, BLOCK, 2048);
    if (s->x_valid) hipLaunchKernelGGL(k_maxabs, dim3(g), dim3(BLOCK), 0, st, n, s->x.p, (const double*)nullptr, s->red_scratch.p);
    else hipLaunchKernelGGL(k_maxabs, dim3(g), dim3(BLOCK), 0, st, n, s->z.p, s->z_matrix->ds.p, s->red_scratch.p);
    PG_HIP(hipGetLastError());
    std::vector<double> h(g);
    s->red_scratch.download(h.data(), g);
    for (double v : h) m = std::max(m, v);
  }
  if (ctx().nranks > 1 || ctx().comm) {
    DevBuf<double> d(1);
    d.upload(&m, 1);
    comm_allreduce_max_f64(d.p, 1, st);
    d.download(&m, 1);
  }
  return m;
}

void ensure_run_matrix(pg_solver* s, int scheme) {
  if (s->have_run && s->scheme_run == scheme) return;
  if (scheme == s->scheme_ctor && s->A_ctor.n == s->nb.n_own && s->A_ctor.rowptr.p) {
    // same scheme: the loop matrix equals the constructor matrix (border rows are re-applied identically)
    s->scheme_run = scheme;
    s->have_run = true;
    return;
  }
  const SysParams P = make_params(s, scheme);
  s->spec_y_valid = false;          // (a product queued ahead with the matrix that is about to be replaced)
  s->elim_run = GammaElim();        // (belongs to the matrix that is about to be replaced)
  s->diag_run = DiagElim();
  if (s->blk_cache_matrix == &s->A_run) s->blk_cache_matrix = nullptr;
  assemble_csr_like(P, s->slab, s->nb, s->A_ctor, s->A_run);
  s->A_run.scheme = scheme;
  s->scheme_run = scheme;
  s->have_run = true;
}

const CsrMatrix& run_matrix(const pg_solver* s) {
  return (s->scheme_run == s->scheme_ctor) ? s->A_ctor : s->A_run;
}

void fill_info(pg_solver* s, const SolveStats& st, pg_step_info* info) {
  if (!info) return;
  info->iters = st.iters;
  info->converged = st.converged;
  info->resnorm = st.resnorm;
  info->bnorm = st.bnorm;
  info->extremum = max_abs(s);
  info->time = s->t;
}

// x = S z, on demand
void materialize_x(pg_solver* s) {
  if (s->x_valid) return;
  const i64 n = s->nb.n_own;
  if (n > 0) {
    hipLaunchKernelGGL(k_scale_state, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx().stream, n, s->z_matrix->ds.p, s->z.p,
                       s->x.p, 0);
    PG_HIP(hipGetLastError());
  }
  s->x_valid = true;
}

void keep_state(pg_solver* s) {
  materialize_x(s);
  DevBuf<double> cp(s->nb.n_own > 0 ? s->nb.n_own : 1);
  if (s->nb.n_own > 0)
    PG_HIP(hipMemcpyAsync(cp.p, s->x.p, sizeof(double) * s->nb.n_own, hipMemcpyDeviceToDevice, ctx().stream));
  s->states.emplace_back(std::move(cp));
}

pg_krylov_opts default_opts() {
  pg_krylov_opts o;
  o.method = PG_METHOD_BICGSTAB;
  o.reltol = 1e-12;
  o.abstol = 0.0;
  o.maxiter = 0;
  o.check_every = 4;
  o.warm_start = 1;
  o.restart = 0;
  o.precond = 0;
  return o;
}

void do_initial(pg_solver* s, const pg_krylov_opts* opts, SolveStats& st) {
  const pg_krylov_opts o = opts ? *opts : default_opts();
  fit_join(s);
  // solve_system!(s) with the constructor's A and b (diffusion.jl:275)
  const i64 n = s->nb.n_own;
  if (s->moving && o.warm_start != 0 && o.method == PG_METHOD_BICGSTAB && n > 0) {
    // a slab of the moving solver: start from the previous state on this slab's active set (s->x, written by k_rhs_first;
    // fresh cells start from the 0 their eliminated unknown held) -- a time step, as pg_solver_step's warm start
    hipStream_t stream = ctx().stream;
    hipLaunchKernelGGL(k_scale_state, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, stream, n, s->A_ctor.ds.p, s->x.p, s->z.p, 1);
    spmv_halo(s->A_ctor, s->nb, s->slab, s->z.p, s->y.p, stream);
    PG_HIP(hipGetLastError());
    krylov_solve(s->A_ctor, s->nb, s->slab, s->b.p, s->ysol.p, s->work, o, st, s->z.p, s->y.p, false);
  } else {
    krylov_solve(s->A_ctor, s->nb, s->slab, s->b.p, s->ysol.p, s->work, o, st);
  }
  std::swap(s->z.p, s->ysol.p);   // the scaled solution becomes the state (same-size buffers trade roles)
  s->z_matrix = &s->A_ctor;
  s->x_valid = false;
  s->initial_done = true;
  s->diag_ctor.snapped_version = s->diag_run.snapped_version = -1;   // z was replaced behind the compact path's back
  s->hist_cnt = 0;
  s->spec_y_valid = false;
}

// ---- host side of the extrapolated start (GuessArgs), shared by the compact and the plain warm path ---------------------
// Worth its launch?  The fit costs about 20 us per step whatever the size, a product max(5 us, its bytes at 5 TB/s): the
// mechanism is switched on (for good, on this system) once a solve has used so many products that a few of them saved pay
// for it -- 512^3: from 7 products (the loop needs 26), 64^3 ... 2048^2: from 18 (launch-bound: the 26 products of the 3-D CN
// loops qualify, +15 ... +50 %) -- and not on systems of a few thousand rows, where every launch is latency and the fit's
// is one more (80^2, 1815 rows: 19 -> 11 products per step and still 6400 instead of 7800 steps/s).
void guess_policy(bool& on, double last_products, double bytes_per_product) {
  const Config& cfg = config();
  if (on || cfg.guess_n <= 0 || last_products <= 0.0) return;
  const double t_prod_us = std::max(5.0, bytes_per_product / 5.0e6);
  if (cfg.guess_always || (bytes_per_product >= 256.0 * 1024.0 && last_products >= 6.0 + 3.0 * (20.0 / t_prod_us))) on = true;
}

struct GuessPlan {
  int KH = 0, R = 0, hist_had = 0;
  bool refit = true;      // false: this step keeps the coefficients it has (see guess_prepare)
  GuessArgs ga{};
};

// before the step's first kernel: the ring buffers and the kernel's arguments; the count of valid older states is taken
// and reset (guess_commit restores it once the step has gone its way to the end)
GuessPlan guess_prepare(pg_solver* s, const void* owner, bool allowed, double last_products, hipStream_t stream) {
  const Config& cfg = config();
  GuessPlan gp;
  // a loop that is down to the smallest polynomial (or meets the tolerance at the start) has nothing left to save: the fit
  // then runs every eighth step only -- its launch is a third of such a step on small systems
  gp.refit = last_products > 4.0 || s->steps_done % 8 == 0;
  gp.KH = allowed ? cfg.guess_n : 0;                           // older states read at most
  gp.R = gp.KH > 0 ? cfg.guess_depth : 0;                      // older states kept
  gp.hist_had = (s->hist_de == owner && s->hist_k == gp.R) ? s->hist_cnt : 0;
  s->hist_cnt = 0;
  if (gp.KH == 0) return gp;
  const int R = gp.R;
  const i64 nva = s->z.n;
  if (s->guess_coef.n == 0) { s->guess_coef.alloc(GUESS_USED + 16); s->guess_coef.zero(); s->guess_ticket.alloc(1); s->guess_ticket.zero(); }
  for (int j = 0; j <= R; ++j) {      // (one more product buffer than states: the one retired a step ago takes the next
    if (j < R && s->zh[j].n != nva) { s->zh[j].alloc(nva); s->zh[j].zero(); }   //  product while the fit still reads the others)
    if (s->yh[j].n != nva) { s->yh[j].alloc(nva); s->yh[j].zero(); }
  }
  if (gp.hist_had == 0) PG_HIP(hipMemsetAsync(s->guess_coef.p, 0, 12 * sizeof(double), stream));   // (the counters stay)
  for (int j = 0; j < 8; ++j) { gp.ga.zr[j] = s->zh[std::min(j, R - 1)].p; gp.ga.yr[j] = s->yh[std::min(j, R - 1)].p; }
  gp.ga.znew = s->zh[R - 1].p;
  gp.ga.coef = s->guess_coef.p;
  return gp;
}

// after the step's first kernel: the fit for the next step (k_guess_fit; cmap == NULL: every row is in the loop), then the
// buffers trade places
void guess_after_rhs(pg_solver* s, const GuessPlan& gp, i64 n, const int* cmap, const double* ds, const double* rhat, bool single,
                     KrylovWork& w, hipStream_t stream) {
  if (gp.KH == 0) return;
  const Config& cfg = config();
  const int R = gp.R;
  if (gp.hist_had > 0 && gp.refit) {
    GuessFit gf{};
    for (int j = 0; j < 8; ++j) gf.yr[j] = s->yh[std::min(j, R - 1)].p;
    const i64 nchunk = (n + BLOCK - 1) / BLOCK;
    gf.stride = (int)std::max<i64>(1, std::min<i64>(cfg.guess_monitor, nchunk / 128));   // (small systems: every chunk)
    const int gfit = (int)std::min<i64>(128, (nchunk + gf.stride - 1) / gf.stride);
    if (s->guess_partials.n != (i64)GUESS_NS * std::max(gfit, 1)) s->guess_partials.alloc((i64)GUESS_NS * std::max(gfit, 1));
    gf.partials = s->guess_partials.p;
    gf.ticket = s->guess_ticket.p;
    gf.coef = s->guess_coef.p;
    gf.avail = std::min(gp.hist_had, GUESS_RM);
    gf.kmax = gp.KH;
    gf.rate2 = w.last_rate2;
    gf.pass_cost = cfg.guess_pass_cost;
    gf.gain = cfg.guess_gain;
    gf.sums = single ? nullptr : s->guess_coef.p + 16;
    hipStream_t fst = stream;
    if (single && cfg.guess_async) {
      if (!s->fit.st) {
        PG_HIP(hipStreamCreateWithFlags(&s->fit.st, hipStreamNonBlocking));
        PG_HIP(hipEventCreateWithFlags(&s->fit.ev_rhs, hipEventDisableTiming));
        PG_HIP(hipEventCreateWithFlags(&s->fit.ev_fit, hipEventDisableTiming));
      }
      PG_HIP(hipEventRecord(s->fit.ev_rhs, stream));
      PG_HIP(hipStreamWaitEvent(s->fit.st, s->fit.ev_rhs, 0));
      fst = s->fit.st;
    }
    hipLaunchKernelGGL(k_guess_fit, dim3(std::max(gfit, 1)), dim3(BLOCK), 0, fst, n, cmap, ds, (const double*)s->b.p,
                       (const double*)s->y.p, rhat, gf);
    if (fst != stream) {
      PG_HIP(hipEventRecord(s->fit.ev_fit, fst));
      s->fit.pending = true;
    }
    if (!single) {
      comm_allreduce_sum_f64(gf.sums, GUESS_NS, stream);
      hipLaunchKernelGGL(k_guess_decide, dim3(1), dim3(BLOCK), 0, stream, gf);
    }
    PG_HIP(hipGetLastError());
  }
  // the buffers trade places (stream order keeps the kernels above ahead of whatever writes them next): the state written
  // becomes z, z and ŷ become the newest kept pair, the product buffer retired a step ago is the next product's output
  double* znew = s->zh[R - 1].p;
  double* yspare = s->yh[R].p;
  double* yretire = s->yh[R - 1].p;
  for (int j = R - 1; j > 0; --j) { s->zh[j].p = s->zh[j - 1].p; s->yh[j].p = s->yh[j - 1].p; }
  s->zh[0].p = s->z.p; s->z.p = znew;
  s->yh[0].p = s->y.p; s->y.p = yspare;
  s->yh[R].p = yretire;
}

// the step has gone its way to the end: one more older state stands
void guess_commit(pg_solver* s, const GuessPlan& gp, const void* owner) {
  if (gp.KH == 0) return;
  if (config().debug) {
    double hc[16];
    fit_join(s, true);
    s->guess_coef.download(hc, 16);
    fprintf(stderr, "[pg_solver] extrapolated start: %d older states kept; sampled (r,r)_W plain %.3e, taken %.3e; next step: %d states",
            gp.hist_had, hc[9], hc[10], (int)hc[4]);
    for (int j = 0; j < (int)hc[4]; ++j) fprintf(stderr, " z(n-%d)*%.4f", (int)hc[5 + j] + 1, hc[j]);
    fprintf(stderr, ", fit leaves %.3e\n", hc[11]);
  }
  s->hist_cnt = std::min(gp.hist_had + 1, gp.R);
  s->hist_k = gp.R;
  s->hist_de = owner;
}

void do_step(pg_solver* s, int scheme, const pg_krylov_opts* opts, SolveStats& st) {
  PG_REQUIRE(s->scheme_ctor != PG_SCHEME_STEADY, "a steady solver has no time loop");
  PG_REQUIRE(!s->moving, "a moving-body solver is one space-time step: create the next one from the next time slab");
  PG_REQUIRE(s->initial_done, "Solver is not initialized. Call pg_solver_initial_solve first.");
  const pg_krylov_opts o = opts ? *opts : default_opts();
  hipStream_t stream = ctx().stream;
  fit_join(s);
  ensure_run_matrix(s, scheme);
  const CsrMatrix& A = run_matrix(s);
  const i64 n = s->nb.n_own;
  const bool warm = o.warm_start != 0 && o.method == PG_METHOD_BICGSTAB && n > 0;
  const DiagElim* used_de = nullptr;   // the compact path taken in this step (its "quiet" bookkeeping stays valid)
  struct QuietGuard {                  // any other path moves z behind the compact path's back
    pg_solver* s; const DiagElim*& used;
    ~QuietGuard() {
      if (used != &s->diag_ctor) s->diag_ctor.snapped_version = -1;
      if (used != &s->diag_run) s->diag_run.snapped_version = -1;
    }
  } quiet_guard{s, used_de};
  s->t += s->dt;                     // diffusion.jl:287
  ensure_bconst(s, scheme);
  if (warm) {
    // loop form on the scaled state: z is the previous Krylov solution itself
    if (s->z_matrix != &A) {         // the matrix (hence S) changed since z was computed: BE ctor system -> CN run system
      hipLaunchKernelGGL(k_rescale_state, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, stream, n, s->z_matrix->ds.p, A.ds.p, s->z.p);
      s->z_matrix = &A;
    }
    // ŷ = Â z = B⁻¹S A x: CN right-hand side and warm-start residual -- unless the previous step has queued it already
    const bool have_y = s->spec_y_valid && s->spec_matrix == &A;
    s->spec_y_valid = false;
    if (!have_y) spmv_halo(A, s->nb, s->slab, s->z.p, s->y.p, stream);
    const bool blk_mf = s->nphase == 2 && scheme == PG_SCHEME_CN;   // (see k_rhs_block_mf)
    if (A.n_blk > 0 && blk_mf) {
      hipLaunchKernelGGL(k_rhs_block_mf, dim3(grid_for(A.n_blk, BLOCK)), dim3(BLOCK), 0, stream, make_params(s, scheme), make_segs(s->nb),
                         s->Mloc, A.n_blk, A.blk_rows.p, A.blk_idx.p, A.blk_coef.p, s->nb.row_cell.p, s->nb.red.p, A.ds.p,
                         (const double*)s->z.p, s->mass.p, s->bconst.p, s->fixed.p, s->b.p);
    } else if (A.n_blk > 0) {
      if (s->blk_cache_matrix != &A || s->blk_cache_scheme != scheme || s->blk_cache_version != s->bconst_version) {
        if (s->blk_wz.n != A.n_blk * MAX_KINDS) s->blk_wz.alloc(A.n_blk * MAX_KINDS);   // (time-dependent data: rebuilt every step)
        if (s->blk_c0.n != A.n_blk) s->blk_c0.alloc(A.n_blk);
        hipLaunchKernelGGL(k_blk_cache, dim3(grid_for(A.n_blk, BLOCK)), dim3(BLOCK), 0, stream, A.n_blk, scheme, A.blk_idx.p,
                           A.blk_coef.p, A.ds.p, s->mass.p, s->bconst.p, s->fixed.p, s->blk_wz.p, s->blk_c0.p);
        s->blk_cache_matrix = &A; s->blk_cache_scheme = scheme; s->blk_cache_version = s->bconst_version;
      }
      hipLaunchKernelGGL(k_rhs_block_c, dim3(grid_for(A.n_blk, BLOCK)), dim3(BLOCK), 0, stream, A.n_blk, scheme, A.blk_rows.p,
                         A.blk_idx.p, (const double*)s->blk_wz.p, (const double*)s->blk_c0.p, A.blk_cn.p, (const double*)s->z.p,
                         (const double*)s->y.p, s->b.p);
    }
    KrylovWork& w = s->work;
    // rows alone on their diagonal (interface AND border identity rows) left out, compact vectors (pg_reduce.hip, DiagElim)
    DiagElim& DE = (&A == &s->A_ctor) ? s->diag_ctor : s->diag_run;
    if (!DE.tried) build_diag_elim(A, s->nb, s->slab, DE);
    if (DE.active && krylov_uses_polynomial(DE.A, o)) {
      // r = r̂ = p of the remaining rows go straight to the compact vectors; the other rows are solved in the same pass
      const int stamp = (int)((s->steps_done % 2000000000) + 1);   // marks E.flag when a diagonal row moved in THIS step
      // quiet: the previous step ran this path with the same constant data, so every diagonal row already holds its solution
      // and none can move (beyond rounding) -- the coupling / renorm launches are not queued and the solver's start phase is
      // folded into the right-hand-side kernel (3 launches less)
      // Several ranks: the host-side knowledge "same constant data as the step before" is the same on every rank, so the
      // coupling / renorm launches (and the exchange of the deltas they need there) are skipped collectively; only the folded
      // start is a one-rank thing (the scalar phase of several ranks goes through an all-reduce).
      const bool single = ctx().nranks == 1 && !ctx().comm;
      const bool same_data = DE.snapped_version == s->bconst_version;
      const bool quiet = same_data && single;
      // extrapolated start (GuessArgs): steps with unchanged data only -- the rows alone on their diagonal rest, so the
      // differences of the kept states are differences of the loop's rows alone and the products ŷ^{n-o} combine linearly.
      // Several ranks: "unchanged data" is the same verdict everywhere, the fit's sums go through an all-reduce and every rank
      // takes the same decision.
      guess_policy(DE.guess_on, DE.last_products, DE.bytes_per_rank);
      GuessPlan gp = guess_prepare(s, &DE, same_data && DE.guess_on, DE.last_products, stream);
      const int KH = gp.KH;
      // the extrapolated state itself is formed by the solve's first update of x (KrylovWork::xguess): the older states are
      // then read there through the compact system's index map, once, and this kernel neither reads them nor writes the state
      const Config& cfg = config();
      // (not after a solve that met the tolerance at its start: the next one probably will too, no update of x runs then, and
      //  writing the state from the first s kernel's early exit costs more than writing it here -- 64^3 BE near steady state)
      const bool defer = KH > 0 && cfg.guess_defer && cfg.poly_xspace && cfg.fuse_half_update && DE.last_products > 0.0;
      gp.ga.defer = defer ? 1 : 0;
      const GuessArgs& ga = gp.ga;
      const double* zprev = s->z.p;      // z^n (the buffers trade places below)
#define PG_RHS_INIT_C(KHV)                                                                                                        \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rhs_init_c<KHV>), dim3(w.grid), dim3(BLOCK), 0, stream, n, scheme, s->z.p, s->y.p, A.ds.p,   \
                     s->mass.p, s->bconst.p, s->fixed.p, A.isblk.p, DE.cmap.p, s->b.p, (const double*)DE.gdiag.p, DE.delta.p,     \
                     DE.flag.p, stamp, w.rhat.p, w.partials.p, quiet ? w.ticket.p : nullptr, w.sc.p, o.reltol * o.reltol,          \
                     o.abstol * o.abstol, ga)
      switch (KH) {
        case 1: PG_RHS_INIT_C(1); break;
        case 2: PG_RHS_INIT_C(2); break;
        case 3: PG_RHS_INIT_C(3); break;
        case 4: PG_RHS_INIT_C(4); break;
        default: PG_RHS_INIT_C(0); break;
      }
#undef PG_RHS_INIT_C
      PG_HIP(hipGetLastError());
      guess_after_rhs(s, gp, n, (const int*)DE.cmap.p, (const double*)A.ds.p, (const double*)w.rhat.p, single, w, stream);
      if (!same_data) diag_fix(DE, s->nb, s->slab, stamp, !single, w.rhat.p, w.partials.p, w.grid, stream);
      w.start_folded = quiet;
      DE.snapped_version = s->bconst_version;
      used_de = &DE;
      w.scatter = DE.rlist.p;
      w.p_in_rhat = true;
      bool solved = false;
      s->spec_pending = false;
      if (quiet && s->hint_more_steps && config().speculate_product)
        w.after_first_batch = [s, &A, stream]() {     // the next step's ŷ = Â z, behind this solve's (expected) last update of z
          spmv_halo(A, s->nb, s->slab, s->z.p, s->y.p, stream);
          s->spec_pending = true;
          s->spec_matrix = &A;
        };
      XGuess xg;
      if (defer) {
        xg.zbase = zprev;
        for (int j = 0; j < 8; ++j) xg.zr[j] = ga.zr[j];
        xg.coef = s->guess_coef.p + GUESS_USED;
        w.xguess = xg;
      }
      try {
        krylov_solve(DE.A, DE.nb, s->slab, nullptr, s->z.p, w, o, st, nullptr, nullptr, true);
      } catch (...) {
        w.scatter = nullptr;
        w.after_first_batch = nullptr;
        throw;
      }
      w.scatter = nullptr;
      w.after_first_batch = nullptr;
      // the speculative product stands when the solve ended inside its first batch (nothing has touched z since)
      s->spec_y_valid = s->spec_pending && st.polls == 1 && st.converged && st.poly_degree >= 0;
      s->spec_pending = false;
      solved = st.poly_degree >= 0;            // -1: the polynomial stagnated on the compact system -> the full system below
      DE.last_products = (double)st.products;

      // A quiet step rests on "no row alone on its diagonal moves while the data are unchanged"; k_rhs_init_c checks it
      // anyway and the start phase reports it (S_MOVED): the residual the iteration started from then lacked the coupling
      // term, and the step is finished on the full system from the state reached.
      const bool moved_unseen = solved && quiet && w.h_sc[S_MOVED] != 0.0;
      if (!solved || moved_unseen) {
        s->spec_y_valid = false;
        // z has moved (the rows left out hold their solution, the x-space iteration has updated the rest): the right-hand
        // side b̂ of this step -- written for every row by k_rhs_init_c -- stands, the iteration continues on the full
        // system from the state reached, r = b̂ - Âz
        if (!solved) DE.active = false;
        used_de = nullptr;
        const SolveStats first = st;
        st = SolveStats();
        spmv_halo(A, s->nb, s->slab, s->z.p, s->y.p, stream);
        krylov_solve(A, s->nb, s->slab, s->b.p, s->ysol.p, w, o, st, s->z.p, s->y.p, false);
        std::swap(s->z.p, s->ysol.p);
        st.iters += first.iters;
        st.products += first.products;
      }
      else guess_commit(s, gp, &DE);
      s->x_valid = false;
      s->steps_done += 1;
      return;
    }
    // Dirichlet interface: the γ rows are rows of the identity -- solved here, the iteration runs on Â_ωω (pg_reduce.hip)
    GammaElim& E = (&A == &s->A_ctor) ? s->elim_ctor : s->elim_run;
    if (!E.tried) build_gamma_elim(A, s->nb, E);
    // The plain warm path (no row left out: diphasic systems, interfaces with Robin / Neumann conditions and no border rows)
    // takes the extrapolated start as well, on one rank: the kept products are products of whole states whatever the data do,
    // so steps with changing data qualify too.
    const bool single = ctx().nranks == 1 && !ctx().comm;
    const int which = (&A == &s->A_ctor) ? 0 : 1;
    if (single && !E.active) guess_policy(s->plain_guess_on[which], s->plain_last_products[which], (double)A.spmv_bytes);
    const GuessPlan gp = guess_prepare(s, &A, single && !E.active && s->plain_guess_on[which], s->plain_last_products[which], stream);
#define PG_RHS_INIT(KHV)                                                                                                          \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rhs_init<KHV>), dim3(w.grid), dim3(BLOCK), 0, stream, n, s->nb.n_vec(), scheme, s->z.p,     \
                     s->y.p, A.ds.p, s->mass.p, s->bconst.p, s->fixed.p, A.isblk.p, s->b.p, w.r.p, w.rhat.p, w.p.p, w.partials.p, \
                     gp.ga)
    switch (gp.KH) {
      case 1: PG_RHS_INIT(1); break;
      case 2: PG_RHS_INIT(2); break;
      case 3: PG_RHS_INIT(3); break;
      case 4: PG_RHS_INIT(4); break;
      default: PG_RHS_INIT(0); break;
    }
#undef PG_RHS_INIT
    PG_HIP(hipGetLastError());
    guess_after_rhs(s, gp, n, nullptr, (const double*)A.ds.p, (const double*)w.rhat.p, true, w, stream);
    if (E.active) {
      gamma_fix(E, s->z.p, w.r.p, w.rhat.p, w.p.p, w.partials.p, w.grid, stream);
      krylov_solve(E.A, E.nb, s->slab, s->b.p, s->z.p, w, o, st, nullptr, nullptr, true);
    } else {
      // y0 = z (in place), r0 = b̂ - ŷ: same solution as a zero start, fewer iterations
      krylov_solve(A, s->nb, s->slab, s->b.p, s->z.p, w, o, st, nullptr, nullptr, true);
      s->plain_last_products[which] = (double)st.products;
      guess_commit(s, gp, &A);
    }
    s->x_valid = false;
  } else {
    // cold start / CG: the reference's zero initial guess, on the unscaled state
    s->hist_cnt = 0;
    materialize_x(s);
    if (n > 0) {
      if (scheme == PG_SCHEME_CN) {
        hipLaunchKernelGGL(k_scale_state, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, stream, n, A.ds.p, s->x.p, s->z.p, 1);
        spmv_halo(A, s->nb, s->slab, s->z.p, s->y.p, stream);
      }
      hipLaunchKernelGGL(k_rhs, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, stream, n, scheme, s->x.p, s->y.p, A.ds.p, s->mass.p,
                         s->bconst.p, s->fixed.p, s->b.p);
      if (A.n_blk > 0 && s->nphase == 2 && scheme == PG_SCHEME_CN)
        hipLaunchKernelGGL(k_rhs_block_mf, dim3(grid_for(A.n_blk, BLOCK)), dim3(BLOCK), 0, stream, make_params(s, scheme), make_segs(s->nb),
                           s->Mloc, A.n_blk, A.blk_rows.p, A.blk_idx.p, A.blk_coef.p, s->nb.row_cell.p, s->nb.red.p, A.ds.p,
                           (const double*)s->z.p, s->mass.p, s->bconst.p, s->fixed.p, s->b.p);
      else if (A.n_blk > 0)
        hipLaunchKernelGGL(k_rhs_block, dim3(grid_for(A.n_blk, BLOCK)), dim3(BLOCK), 0, stream, A.n_blk, scheme, A.blk_rows.p,
                           A.blk_idx.p, A.blk_coef.p, A.blk_cn.p, s->x.p, (const double*)nullptr, s->y.p, s->mass.p,
                           s->bconst.p, s->fixed.p, s->b.p);
      PG_HIP(hipGetLastError());
    }
    krylov_solve(A, s->nb, s->slab, s->b.p, s->ysol.p, s->work, o, st);
    std::swap(s->z.p, s->ysol.p);
    s->z_matrix = &A;
    s->x_valid = false;
  }
  s->steps_done += 1;
}

}  // namespace

extern "C" {

static void create_mono(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface, const pg_border_desc* borders,
                        int32_t nborders, const double* Dcoef, const double* source, double dt, const double* T0,
                        int32_t scheme, pg_solver** out) {
  PG_REQUIRE(c && o && bc_interface && out, "solver constructor: NULL argument");
  PG_REQUIRE(o->cap == c, "operators were built from a different capacity");
  PG_REQUIRE(dt > 0.0, "dt must be positive");
  auto* s = new pg_solver();
  std::unique_ptr<pg_solver> guard(s);
  s->nphase = 1;
  s->cap[0] = c;
  s->ops[0] = o;
  s->slab = c->slab;
  s->dt = dt;
  s->scheme_ctor = scheme;
  s->bc_i = *bc_interface;
  if (Dcoef) upload_local(s->Id[0], Dcoef, s->slab);
  if (source) upload_local(s->f_np1[0], source, s->slab);
  if (bc_interface->value_array) {
    upload_local(s->g_np1, bc_interface->value_array, s->slab);
    upload_local(s->g_n, bc_interface->value_array, s->slab);
  }
  s->bc_i.value_array = nullptr;
  setup_common(s, borders, nborders, T0);
  *out = guard.release();
}

int32_t pg_solver_create_unsteady_mono(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                       const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                       const double* source, double dt, const double* T0, int32_t scheme,
                                       pg_solver** out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(scheme == PG_SCHEME_BE || scheme == PG_SCHEME_CN, "scheme must be BE or CN");
  create_mono(c, o, bc_interface, borders, nborders, Dcoef, source, dt, T0, scheme, out);
  PG_API_END
}

static int32_t create_moving(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface, const pg_border_desc* borders,
                             int32_t nborders, const double* Dcoef, const double* source_n, const double* source_np1,
                             const double* T_prev, pg_solver* prev, int32_t scheme, pg_solver** out);

int32_t pg_solver_create_moving_mono(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                     const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                     const double* source_n, const double* source_np1, const double* T_prev,
                                     int32_t scheme, pg_solver** out) {
  return create_moving(c, o, bc_interface, borders, nborders, Dcoef, source_n, source_np1, T_prev, nullptr, scheme, out);
}

int32_t pg_solver_create_moving_mono_next(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                          const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                          const double* source_n, const double* source_np1, pg_solver* previous,
                                          int32_t scheme, pg_solver** out) {
  if (!previous) {
    pg::set_last_error("pg_solver_create_moving_mono_next: NULL previous solver");
    return 1;
  }
  return create_moving(c, o, bc_interface, borders, nborders, Dcoef, source_n, source_np1, nullptr, previous, scheme, out);
}

static int32_t create_moving(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface, const pg_border_desc* borders,
                             int32_t nborders, const double* Dcoef, const double* source_n, const double* source_np1,
                             const double* T_prev, pg_solver* prev, int32_t scheme, pg_solver** out) {
  PG_API_BEGIN
  require_init();
  if (prev) PG_REQUIRE(prev->initial_done, "pg_solver_create_moving_mono_next: the previous slab has not been solved");
  AsyncAllocScope pool;   // one solver per time slab: the stream-ordered allocator (pg_common.h)
  PG_REQUIRE(scheme == PG_SCHEME_BE || scheme == PG_SCHEME_CN, "scheme must be BE or CN");
  PG_REQUIRE(c && o && bc_interface && out, "solver constructor: NULL argument");
  PG_REQUIRE(o->cap == c, "operators were built from a different capacity");
  PG_REQUIRE(c->spacetime, "pg_solver_create_moving_mono needs a space-time capacity (pg_capacity_create_spacetime)");
  PG_REQUIRE(!o->has_velocity, "the moving diffusion solver takes no convection operators");
  PG_REQUIRE(ctx().nranks == 1 && !ctx().comm, "space-time steps are single-rank");
  auto* s = new pg_solver();
  std::unique_ptr<pg_solver> guard(s);
  s->nphase = 1;
  s->cap[0] = c;
  s->ops[0] = o;
  s->slab = c->slab;
  s->dt = 1.0;            // Δt is inside the space-time capacities
  s->moving = true;
  s->scheme_ctor = scheme;
  s->bc_i = *bc_interface;
  const i64 Ml = s->slab.Mloc();
  s->psi_p[0].alloc(Ml);
  s->psi_m[0].alloc(Ml);
  hipLaunchKernelGGL(k_psi, dim3(grid_for(Ml, BLOCK)), dim3(BLOCK), 0, ctx().stream, Ml, (int)scheme, c->Vt[0].p, c->Vt[1].p,
                     s->psi_p[0].p, s->psi_m[0].p);
  PG_HIP(hipGetLastError());
  if (Dcoef) upload_local(s->Id[0], Dcoef, s->slab);
  if (source_np1) upload_local(s->f_np1[0], source_np1, s->slab);
  if (source_n) upload_local(s->f_n[0], source_n, s->slab);
  if (bc_interface->value_array) {
    upload_local(s->g_np1, bc_interface->value_array, s->slab);
    upload_local(s->g_n, bc_interface->value_array, s->slab);
  }
  s->bc_i.value_array = nullptr;
  s->init_from = prev;
  s->A_ctor.want_units = false;
  setup_common(s, borders, nborders, T_prev);
  *out = guard.release();
  PG_API_END
}

// MovingDiffusionUnsteadyDiph + A_/b_diph_unstead_diff_moving of one slab      prescribedmotionsolver/diffusion.jl:272-498
int32_t pg_solver_create_moving_diph(pg_capacity* c1, pg_diffops* o1, pg_capacity* c2, pg_diffops* o2, const pg_jump_desc* ic,
                                     const pg_border_desc* borders, int32_t nborders, const double* D1, const double* D2,
                                     const double* f1_n, const double* f1_np1, const double* f2_n, const double* f2_np1,
                                     const double* T_prev, pg_solver* previous, int32_t scheme, pg_solver** out) {
  PG_API_BEGIN
  require_init();
  if (previous) PG_REQUIRE(previous->initial_done, "pg_solver_create_moving_diph: the previous slab has not been solved");
  AsyncAllocScope pool;   // one solver per time slab: freed blocks are reused without synchronisation (pg_context.hip)
  PG_REQUIRE(scheme == PG_SCHEME_BE || scheme == PG_SCHEME_CN, "scheme must be BE or CN");
  PG_REQUIRE(c1 && c2 && o1 && o2 && ic && out, "solver constructor: NULL argument");
  PG_REQUIRE(o1->cap == c1 && o2->cap == c2, "operators were built from a different capacity");
  PG_REQUIRE(c1->spacetime && c2->spacetime, "pg_solver_create_moving_diph needs space-time capacities (pg_capacity_create_spacetime)");
  PG_REQUIRE(c1->mesh == c2->mesh, "Phase capacities must share the same mesh.");
  PG_REQUIRE(!o1->has_velocity && !o2->has_velocity, "the moving diffusion solver takes no convection operators");
  PG_REQUIRE(ctx().nranks == 1 && !ctx().comm, "space-time steps are single-rank");
  auto* s = new pg_solver();
  std::unique_ptr<pg_solver> guard(s);
  s->nphase = 2;
  s->cap[0] = c1; s->ops[0] = o1;
  s->cap[1] = c2; s->ops[1] = o2;
  s->slab = c1->slab;
  s->dt = 1.0;            // Δt is inside the space-time capacities
  s->moving = true;
  s->scheme_ctor = scheme;
  s->ic = *ic;
  const i64 Ml = s->slab.Mloc();
  for (int q = 0; q < 2; ++q) {
    s->psi_p[q].alloc(Ml);
    s->psi_m[q].alloc(Ml);
    hipLaunchKernelGGL(k_psi, dim3(grid_for(Ml, BLOCK)), dim3(BLOCK), 0, ctx().stream, Ml, (int)scheme, s->cap[q]->Vt[0].p,
                       s->cap[q]->Vt[1].p, s->psi_p[q].p, s->psi_m[q].p);
  }
  PG_HIP(hipGetLastError());
  if (D1) upload_local(s->Id[0], D1, s->slab);
  if (D2) upload_local(s->Id[1], D2, s->slab);
  if (f1_np1) upload_local(s->f_np1[0], f1_np1, s->slab);
  if (f2_np1) upload_local(s->f_np1[1], f2_np1, s->slab);
  if (f1_n) upload_local(s->f_n[0], f1_n, s->slab);
  if (f2_n) upload_local(s->f_n[1], f2_n, s->slab);
  if (ic->g_array) upload_local(s->g_arr, ic->g_array, s->slab);
  if (ic->h_array) upload_local(s->h_arr, ic->h_array, s->slab);
  s->ic.g_array = s->ic.h_array = nullptr;
  s->init_from = previous;
  s->A_ctor.want_units = false;
  setup_common(s, borders, nborders, T_prev);
  *out = guard.release();
  PG_API_END
}

int32_t pg_solver_create_steady_mono(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                     const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                     const double* source, pg_solver** out) {
  PG_API_BEGIN
  require_init();
  create_mono(c, o, bc_interface, borders, nborders, Dcoef, source, 1.0, nullptr, PG_SCHEME_STEADY, out);
  PG_API_END
}

static void create_diph(pg_capacity* c1, pg_diffops* o1, pg_capacity* c2, pg_diffops* o2, const pg_jump_desc* ic,
                        const pg_border_desc* borders, int32_t nborders, const double* D1, const double* D2,
                        const double* f1, const double* f2, double dt, const double* T0, int32_t scheme, pg_solver** out) {
  PG_REQUIRE(c1 && c2 && o1 && o2 && ic && out, "solver constructor: NULL argument");
  PG_REQUIRE(c1->mesh == c2->mesh, "Phase capacities must share the same mesh.");
  PG_REQUIRE(c1->slab.p0 == c2->slab.p0 && c1->slab.p1 == c2->slab.p1, "phase capacities must share the slab partition");
  PG_REQUIRE(dt > 0.0, "dt must be positive");
  auto* s = new pg_solver();
  std::unique_ptr<pg_solver> guard(s);
  s->nphase = 2;
  s->cap[0] = c1;
  s->ops[0] = o1;
  s->cap[1] = c2;
  s->ops[1] = o2;
  s->slab = c1->slab;
  s->dt = dt;
  s->scheme_ctor = scheme;
  s->ic = *ic;
  if (D1) upload_local(s->Id[0], D1, s->slab);
  if (D2) upload_local(s->Id[1], D2, s->slab);
  if (f1) upload_local(s->f_np1[0], f1, s->slab);
  if (f2) upload_local(s->f_np1[1], f2, s->slab);
  if (ic->g_array) upload_local(s->g_arr, ic->g_array, s->slab);
  if (ic->h_array) upload_local(s->h_arr, ic->h_array, s->slab);
  s->ic.g_array = s->ic.h_array = nullptr;
  setup_common(s, borders, nborders, T0);
  *out = guard.release();
}

int32_t pg_solver_create_unsteady_diph(pg_capacity* c1, pg_diffops* o1, pg_capacity* c2, pg_diffops* o2,
                                       const pg_jump_desc* ic, const pg_border_desc* borders, int32_t nborders,
                                       const double* D1, const double* D2, const double* f1, const double* f2, double dt,
                                       const double* T0, int32_t scheme, pg_solver** out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(scheme == PG_SCHEME_BE || scheme == PG_SCHEME_CN, "scheme must be BE or CN");
  create_diph(c1, o1, c2, o2, ic, borders, nborders, D1, D2, f1, f2, dt, T0, scheme, out);
  PG_API_END
}

int32_t pg_solver_create_steady_diph(pg_capacity* c1, pg_diffops* o1, pg_capacity* c2, pg_diffops* o2,
                                     const pg_jump_desc* ic, const pg_border_desc* borders, int32_t nborders,
                                     const double* D1, const double* D2, const double* f1, const double* f2,
                                     pg_solver** out) {
  PG_API_BEGIN
  require_init();
  create_diph(c1, o1, c2, o2, ic, borders, nborders, D1, D2, f1, f2, 1.0, nullptr, PG_SCHEME_STEADY, out);
  PG_API_END
}

int32_t pg_solver_destroy(pg_solver* s) {
  PG_API_BEGIN
  delete s;
  PG_API_END
}

int32_t pg_solver_set_source(pg_solver* s, int32_t phase, const double* f_n, const double* f_np1) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && phase >= 0 && phase < s->nphase, "pg_solver_set_source: bad arguments");
  if (f_n) upload_local(s->f_n[phase], f_n, s->slab);
  if (f_np1) upload_local(s->f_np1[phase], f_np1, s->slab);
  s->bconst_dirty = true;
  if (!s->initial_done) build_first_rhs(s);
  PG_API_END
}

int32_t pg_solver_set_interface_value(pg_solver* s, const double* g_n, const double* g_np1) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && s->nphase == 1, "pg_solver_set_interface_value: monophasic solver expected");
  if (g_n) upload_local(s->g_n, g_n, s->slab);
  if (g_np1) upload_local(s->g_np1, g_np1, s->slab);
  s->bconst_dirty = true;
  if (!s->initial_done) build_first_rhs(s);
  PG_API_END
}

int32_t pg_solver_set_border_values(pg_solver* s, const double* values) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && values, "pg_solver_set_border_values: NULL argument");
  pg_mesh* m = s->cap[0]->mesh;
  mesh_build_border(m);
  const i64 nb = (i64)m->border_key.size();
  if (!s->border_cells_lin.p) {
    std::vector<i64> lin(nb);
    for (i64 q = 0; q < nb; ++q) {
      i64 li = 0;
      for (int d = 0; d < m->N; ++d) li += (m->border_idx[q * m->N + d] - 1) * s->slab.stride[d];
      lin[q] = li;
    }
    s->border_cells_lin.alloc(nb);
    s->border_cells_lin.upload(lin.data(), nb);
  }
  DevBuf<double> dv(nb);
  dv.upload(values, nb);
  hipLaunchKernelGGL(k_set_border_values, dim3(grid_for(nb, BLOCK)), dim3(BLOCK), 0, ctx().stream, nb, s->border_cells_lin.p,
                     dv.p, s->slab.first_cell(), s->Mloc, s->nphase, s->nb.red.p, s->nb.n_own, s->bcv.p);
  PG_HIP(hipGetLastError());
  PG_HIP(hipStreamSynchronize(ctx().stream));
  s->bconst_dirty = true;
  if (!s->initial_done) build_first_rhs(s);
  PG_API_END
}

int32_t pg_solver_initial_solve(pg_solver* s, const pg_krylov_opts* opts, pg_step_info* info) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s, "Solver is not initialized. Call a solver constructor first.");
  SolveStats st;
  std::unique_ptr<AsyncAllocScope> pool(s->moving ? new AsyncAllocScope() : nullptr);   // (work vectors allocated on first use)
  do_initial(s, opts, st);
  fill_info(s, st, info);
  PG_API_END
}

int32_t pg_solver_step(pg_solver* s, int32_t scheme, const pg_krylov_opts* opts, pg_step_info* info) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s, "Solver is not initialized. Call a solver constructor first.");
  PG_REQUIRE(scheme == PG_SCHEME_BE || scheme == PG_SCHEME_CN, "scheme must be BE or CN");
  SolveStats st;
  do_step(s, scheme, opts, st);
  fill_info(s, st, info);
  PG_API_END
}

int32_t pg_solver_run(pg_solver* s, double Tend, int32_t scheme, const pg_krylov_opts* opts, int32_t do_initial_flag,
                      int64_t max_steps, int32_t save_every, pg_run_info* info) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s, "Solver is not initialized. Call a solver constructor first.");
  PG_REQUIRE(scheme == PG_SCHEME_BE || scheme == PG_SCHEME_CN, "scheme must be BE or CN");
  hipStream_t stream = ctx().stream;
  EventPair ev;                     // destroyed on every exit path (do_step may throw)
  const auto t_run0 = std::chrono::steady_clock::now();
  const double wait0 = pg::g_host_wait_us;
  PG_HIP(hipEventRecord(ev.e0, stream));
  double used0 = 0.0;               // older states read by the extrapolated starts so far (device counter, GuessArgs)
  if (s->guess_coef.n >= 16) { fit_join(s, true); s->guess_coef.download(&used0, 1, 13); }
  SolveStats tot;
  i64 steps = 0, iters = 0, unconverged = 0;
  double worst = 0.0;
  auto account = [&](const SolveStats& st) {
    iters += st.iters; tot.spmv_ms += st.spmv_ms; tot.spmv_launches += st.spmv_launches;
    tot.spmv_lean_ms += st.spmv_lean_ms; tot.spmv_lean_launches += st.spmv_lean_launches;
    tot.poly_degree = st.poly_degree;
    tot.poly_xspace = st.poly_xspace;
    tot.half_exit += st.half_exit;
    tot.products += st.products;
    if (!st.converged) ++unconverged;
    if (st.bnorm > 0.0) worst = std::max(worst, st.resnorm / st.bnorm);
  };
  if (do_initial_flag) {
    SolveStats st;
    do_initial(s, opts, st);
    account(st);
    if (save_every > 0) keep_state(s);
  }
  while (s->t < Tend) {             // diffusion.jl:286
    if (max_steps >= 0 && steps >= max_steps) break;
    SolveStats st;
    s->hint_more_steps = (s->t + s->dt < Tend) && (max_steps < 0 || steps + 1 < max_steps);   // (another step follows this one)
    struct HintReset { pg_solver* s; ~HintReset() { s->hint_more_steps = false; } } hint_reset{s};
    do_step(s, scheme, opts, st);
    account(st);
    ++steps;
    if (save_every > 0 && steps % save_every == 0) keep_state(s);
  }
  PG_HIP(hipEventRecord(ev.e1, stream));
  PG_HIP(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  PG_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  if (config().debug && steps > 0) {
    const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_run0).count();
    fprintf(stderr, "[pg_solver_run] %lld steps: wall %.1f us per step, of which the host waited %.1f us for the device (the rest: queueing "
            "launches and bookkeeping); device time by events %.1f us per step\n", (long long)steps, wall_us / steps,
            (pg::g_host_wait_us - wait0) / steps, ms * 1e3 / steps);
  }
  if (info) {
    info->steps = steps;
    info->total_iters = iters;
    info->t_final = s->t;
    info->extremum = max_abs(s);
    info->solve_ms = ms;
    info->spmv_ms_total = tot.spmv_ms;
    info->spmv_launches = tot.spmv_launches;
    info->unconverged_steps = unconverged;
    info->worst_relres = worst;
    info->spmv_lean_ms_total = tot.spmv_lean_ms;
    info->spmv_lean_launches = tot.spmv_lean_launches;
    info->poly_degree = tot.poly_degree;
    info->poly_xspace = tot.poly_xspace;
    info->half_exits = tot.half_exit;
    info->products = tot.products;
    info->guess_states_read = 0;
    if (s->guess_coef.n >= 16) {
      double hc[16];
      fit_join(s, true);
      s->guess_coef.download(hc, 16);
      info->guess_states_read = (int64_t)(hc[13] - used0);
    }
  }
  PG_API_END
}

int32_t pg_solver_num_states(const pg_solver* s, int64_t* out) {
  PG_API_BEGIN
  PG_REQUIRE(s && out, "pg_solver_num_states: NULL argument");
  *out = (int64_t)s->states.size();
  PG_API_END
}

int32_t pg_solver_get_state(const pg_solver* s, int64_t state_index, double* x, int64_t len) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && x, "pg_solver_get_state: NULL argument");
  PG_REQUIRE(len == (i64)s->K * s->M, "pg_solver_get_state: len must be 2M (mono) or 4M (diph)");
  if (state_index < 0) materialize_x(const_cast<pg_solver*>(s));
  const double* src = s->x.p;
  if (state_index >= 0) {
    PG_REQUIRE(state_index < (i64)s->states.size(), "pg_solver_get_state: state index out of range");
    src = s->states[state_index].p;
  }
  hipStream_t st = ctx().stream;
  DevBuf<double> pad((i64)s->K * s->Mloc);
  pad.zero();                                            // s.x = zeros(n)          solver.jl:186
  if (s->nb.n_own > 0) {
    hipLaunchKernelGGL(k_scatter_active, dim3(grid_for(s->nb.n_own, BLOCK)), dim3(BLOCK), 0, st, make_segs(s->nb),
                       s->nb.n_own, s->Mloc, s->nb.row_cell.p, src, pad.p);   // s.x[cols_idx] = x_reduced   :187
    PG_HIP(hipGetLastError());
  }
  // only the owned planes are authoritative on this rank
  const i64 own_lo = (s->slab.p0 - s->slab.s0) * s->slab.plane, own_n = (s->slab.p1 - s->slab.p0) * s->slab.plane;
  for (int k = 0; k < s->K; ++k)
    pad.download(x + (i64)k * s->M + s->slab.p0 * s->slab.plane, own_n, (i64)k * s->Mloc + own_lo);
  PG_API_END
}

int32_t pg_solver_system_info(const pg_solver* s, int32_t which, pg_system_info* out) {
  PG_API_BEGIN
  PG_REQUIRE(s && out, "pg_solver_system_info: NULL argument");
  const CsrMatrix& Af = ((which & 1) == 0 || !s->have_run) ? s->A_ctor : run_matrix(s);
  // which & 4: the matrix the warm time loop iterates on -- Â without the Dirichlet interface unknowns when that reduction
  // is active for this system (pg_reduce.hip), else Â itself; n_own / n_omega / n_gamma always describe the full system
  const GammaElim& E = (&Af == &s->A_ctor) ? s->elim_ctor : s->elim_run;
  const DiagElim& DE = (&Af == &s->A_ctor) ? s->diag_ctor : s->diag_run;
  const CsrMatrix& A = ((which & 4) && DE.active) ? DE.A : (((which & 4) && E.active) ? E.A : Af);
  out->n_own = s->nb.n_own;
  out->nnz = (which & 6) ? A.nnz : A.nnz_raw;
  out->n_ghost = s->nb.n_ghost;
  i64 nw = 0, ng = 0;
  for (int k = 0; k < s->K; ++k) ((k & 1) ? ng : nw) += s->nb.cnt_own[k];
  out->n_omega = nw;
  out->n_gamma = ng;
  out->M_global = s->M;
  out->spmv_bytes = A.spmv_bytes;
  out->spmv_slices = A.nslices;
  out->rows_uniform = A.rows_u;
  out->rows_pattern = A.rows_p;
  out->rows_irregular = A.rows_g;
  out->neumann_ok = A.poly_ok ? 1 : 0;
  out->gershgorin = A.gersh;
  out->spmv_units = A.nunits;
  out->rows_marched = A.rows_m;
  out->rows_matrix = A.n;
  out->n_ghost_loop = ((which & 4) && DE.active) ? DE.nb.n_ghost : s->nb.n_ghost;
  out->loop_is_compact = DE.active ? 1 : 0;
  PG_API_END
}

int32_t pg_solver_get_system_csr(const pg_solver* s, int32_t which, int64_t* rowptr, int64_t* col, double* val, double* b,
                                 int64_t* idx) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s, "pg_solver_get_system_csr: NULL argument");
  const bool precond = (which & 2) != 0;   // 2/3: the preconditioned system (Â, b̂) exactly as the Krylov solver sees it
  const int sel = which & 1;
  if (sel == 1) PG_REQUIRE(s->have_run, "run matrix not assembled yet");
  const CsrMatrix& Ah = sel == 0 ? s->A_ctor : run_matrix(s);
  const i64 n = s->nb.n_own;
  // 0/1: the reference's reduced system A x = b: the matrix is re-assembled without scaling (export is a test /
  // debugging path), b is recovered from b̂ = B⁻¹ S b
  CsrMatrix raw;
  if (!precond && (rowptr || col || val))
    assemble_csr(make_params(s, sel == 0 ? s->scheme_ctor : s->scheme_run), s->slab, s->nb, raw, false);
  const CsrMatrix& A = precond ? Ah : raw;
  if (rowptr) {
    std::vector<int> h(n + 1);
    A.rowptr.download(h.data(), n + 1);
    for (i64 i = 0; i <= n; ++i) rowptr[i] = h[i];
  }
  if (col && A.nnz > 0) {
    std::vector<int> h(A.nnz);
    A.col.download(h.data(), A.nnz);
    for (i64 i = 0; i < A.nnz; ++i) col[i] = h[i];
  }
  if (val && A.nnz > 0) A.val.download(val, A.nnz);
  if (b && n > 0) {
    s->b.download(b, n);
    if (!precond) {
      std::vector<double> hb(b, b + n), hds(n);
      Ah.ds.download(hds.data(), n);
      for (i64 r = 0; r < n; ++r) b[r] = hb[r] / hds[r];
      if (Ah.n_blk > 0) {
        const i64 nq = Ah.n_blk;
        std::vector<int> rows(nq), bi(nq * MAX_KINDS);
        std::vector<double> fw(nq * MAX_KINDS);
        Ah.blk_rows.download(rows.data(), nq);
        Ah.blk_idx.download(bi.data(), nq * MAX_KINDS);
        Ah.blk_fw.download(fw.data(), nq * MAX_KINDS);
        for (i64 q = 0; q < nq; ++q) {
          double v = 0.0;
          for (int a = 0; a < MAX_KINDS; ++a)
            if (bi[q * MAX_KINDS + a] >= 0) v += fw[q * MAX_KINDS + a] * hb[bi[q * MAX_KINDS + a]];
          b[rows[q]] = v / hds[rows[q]];
        }
      }
    }
  }
  if (idx && n > 0) {
    std::vector<int> h(n);
    s->nb.row_cell.download(h.data(), n);
    for (i64 r = 0; r < n; ++r) {
      const int k = s->nb.kind_of_row(r);
      idx[r] = (i64)k * s->M + s->slab.first_cell() + h[r];
    }
  }
  PG_API_END
}

int32_t pg_solver_get_row_scaling(const pg_solver* s, int32_t which, double* ds) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && ds, "pg_solver_get_row_scaling: NULL argument");
  if ((which & 1) == 1) PG_REQUIRE(s->have_run, "run matrix not assembled yet");
  const CsrMatrix& A = (which & 1) == 0 ? s->A_ctor : run_matrix(s);
  if (s->nb.n_own > 0) A.ds.download(ds, s->nb.n_own);
  PG_API_END
}

// max |y_a - y_b| between two SpMV kernel variants applied to the same deterministic test vector
__global__ void k_test_vector(i64 n, double* x) {
  for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    x[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

int32_t pg_debug_spmv_compare(pg_solver* s, int32_t which, int32_t variant_a, int32_t variant_b, double* max_abs_diff,
                              double* max_abs) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && max_abs_diff && max_abs, "pg_debug_spmv_compare: NULL argument");
  if (which == 1) PG_REQUIRE(s->have_run, "run matrix not assembled yet");
  const CsrMatrix& A = which == 0 ? s->A_ctor : run_matrix(s);
  hipStream_t st = ctx().stream;
  const i64 n = s->nb.n_own, nv = s->nb.n_vec();
  *max_abs_diff = 0.0;
  *max_abs = 0.0;
  if (n > 0) {
    DevBuf<double> xv(nv), ya(n), yb(n);
    hipLaunchKernelGGL(k_test_vector, dim3(grid_for(nv, BLOCK)), dim3(BLOCK), 0, st, nv, xv.p);
    launch_spmv_variant(variant_a, A, xv.p, ya.p, st);
    launch_spmv_variant(variant_b, A, xv.p, yb.p, st);
    PG_HIP(hipGetLastError());
    std::vector<double> ha(n), hb(n);
    ya.download(ha.data(), n);
    yb.download(hb.data(), n);
    for (i64 i = 0; i < n; ++i) {
      *max_abs_diff = std::max(*max_abs_diff, std::fabs(ha[i] - hb[i]));
      *max_abs = std::max(*max_abs, std::fabs(ha[i]));
    }
  }
  PG_API_END
}

__global__ void k_scale_rows(i64 n_e, const int* __restrict__ elist, double factor, double* __restrict__ z) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < n_e; q += (i64)gridDim.x * blockDim.x) z[elist[q]] *= factor;
}

int32_t pg_debug_scale_diagonal_rows(pg_solver* s, double factor) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && s->have_run, "pg_debug_scale_diagonal_rows: no run matrix yet");
  const CsrMatrix& A = run_matrix(s);
  DiagElim& DE = (&A == &s->A_ctor) ? s->diag_ctor : s->diag_run;
  PG_REQUIRE(DE.active && DE.n_e > 0, "pg_debug_scale_diagonal_rows: the loop of this solver is not the compact one");
  hipLaunchKernelGGL(k_scale_rows, dim3(grid_for(DE.n_e, BLOCK)), dim3(BLOCK), 0, ctx().stream, DE.n_e, (const int*)DE.elist.p, factor,
                     s->z.p);
  PG_HIP(hipGetLastError());
  PG_HIP(hipStreamSynchronize(ctx().stream));
  s->x_valid = false;
  s->spec_y_valid = false;
  PG_API_END
}

int32_t pg_solver_guess_info(pg_solver* s, int32_t* kept, int32_t* nstates, int32_t* offsets, double* coef, double* rr_plain,
                             double* rr_taken) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && kept && nstates && offsets && coef && rr_plain && rr_taken, "pg_solver_guess_info: NULL argument");
  *kept = s->hist_cnt;
  *nstates = 0;
  *rr_plain = *rr_taken = 0.0;
  for (int j = 0; j < 4; ++j) { offsets[j] = 0; coef[j] = 0.0; }
  if (s->guess_coef.n >= 16 && s->hist_cnt > 0) {
    double hc[16];
    fit_join(s, true);
    s->guess_coef.download(hc, 16);
    *nstates = (int32_t)hc[4];
    for (int j = 0; j < *nstates && j < 4; ++j) { offsets[j] = (int32_t)hc[5 + j] + 1; coef[j] = hc[j]; }
    *rr_plain = hc[9];
    *rr_taken = hc[10];
  }
  PG_API_END
}

int32_t pg_solver_time_spmv(pg_solver* s, int32_t which, int32_t reps, double* avg_ms) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(s && avg_ms && reps > 0, "pg_solver_time_spmv: bad arguments");
  // bits 4-6: mode of the launch (0 plain, 1, 2, 3 fused dots, 4: 2x - Ax); bit 11: the Horner step of the x-space loop
  // (mode 8: out = pc2 base + pc0 x + pc1 A x, three vector streams)
  const int sel = which & 1, mode = (which & 2048) ? 8 : ((which >> 4) & 7);
  PG_REQUIRE(mode <= 4 || mode == 8, "pg_solver_time_spmv: unknown launch mode");
  if (sel == 1) PG_REQUIRE(s->have_run, "run matrix not assembled yet");
  const CsrMatrix& Afull = sel == 0 ? s->A_ctor : run_matrix(s);
  // bit 9: the matrix the warm loop iterates on (without the Dirichlet interface rows, pg_reduce.hip), built here if need be
  GammaElim& E = (&Afull == &s->A_ctor) ? s->elim_ctor : s->elim_run;
  if ((which & 512) && !E.tried) build_gamma_elim(Afull, s->nb, E);
  // bit 12: the compact loop matrix (every row alone on its diagonal left out, DiagElim): what the benchmark's loop streams
  DiagElim& DE = (&Afull == &s->A_ctor) ? s->diag_ctor : s->diag_run;
  if ((which & 4096) && !DE.tried) build_diag_elim(Afull, s->nb, s->slab, DE);
  const CsrMatrix& A = ((which & 4096) && DE.active) ? DE.A : (((which & 512) && E.active) ? E.A : Afull);
  // bit 10: chained launches, each reading what the one before wrote (two vectors in turn): the access pattern of a chain
  // of lean launches in the loop
  const bool chained = (which & 1024) != 0;
  hipStream_t st = ctx().stream;
  EventPair ev;
  KrylovWork& w = s->work;
  // bit 8: "cold" launches -- every launch gets another input / output / dot-operand vector out of a ring of 7 (1.7 GB at
  // 512^3: nothing of the previous launch's vectors survives in the 256 MB Infinity Cache), which is how the launches of
  // the Krylov loop see memory; without it the same x and y stay cache resident from launch to launch
  const bool cold = (which & 256) != 0;
  constexpr int RING = 7;
  std::vector<DevBuf<double>> ring;
  const i64 nv = s->nb.n_vec();
  if (cold) {
    ring.resize(RING);
    for (auto& b : ring) {
      b.alloc(nv > 0 ? nv : 1);
      if (nv > 0) PG_HIP(hipMemcpyAsync(b.p, s->z.p, sizeof(double) * (size_t)nv, hipMemcpyDeviceToDevice, st));
    }
  }
  int it = 0;
  auto one = [&]() {
    const double* xin = cold ? ring[it % RING].p : (chained && (it & 1) ? s->y.p : s->z.p);
    double* yout = cold ? ring[(it + 3) % RING].p : (chained && (it & 1) ? s->z.p : s->y.p);
    const double* ax = cold ? ring[(it + 5) % RING].p : w.rhat.p;
    ++it;
    if (mode == 0) spmv(A, xin, yout, st);
    else if (mode == 8) {
      FinArgs f{nullptr, nullptr, PH_NONE, 0, 0, nullptr};
      f.pc0 = 1.0; f.pc1 = -0.5; f.pc2 = 0.5; f.base = s->b.p;   // (the chain's input: one fixed vector, re-read by every launch)
      launch_spmv(8, A, xin, yout, nullptr, nullptr, nullptr, w.grid, st, &f);
    }
    else launch_spmv(mode, A, xin, yout, ax, w.partials.p, nullptr, w.grid, st);
  };
  for (int i = 0; i < 3; ++i) one();
  PG_HIP(hipEventRecord(ev.e0, st));
  for (int i = 0; i < reps; ++i) one();
  PG_HIP(hipEventRecord(ev.e1, st));
  PG_HIP(hipEventSynchronize(ev.e1));
  PG_HIP(hipGetLastError());
  float ms = 0.f;
  PG_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *avg_ms = ms / reps;
  PG_API_END
}

}  // extern "C"
