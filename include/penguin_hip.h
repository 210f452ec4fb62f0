/*
 * penguin_hip.h -- C ABI of libpenguin_hip.so: the MI355X (gfx950) implementation of
 * Penguin.jl's hot path  Mesh -> Capacity -> DiffusionOps -> DiffusionUnsteadyMono/Diph ->
 * solve_DiffusionUnsteady*!.
 *
 * The reference (100 % Julia) has no FFI for this path; its boundary is the exported Julia API
 * (src/Penguin.jl:25-75).  Each entry point below names the reference item it replaces
 * (paths relative to the reference tree).  julia/PenguinHIP.jl ccall's exactly these symbols;
 * penguin/jl_amd/_lib.py binds the same symbols with ctypes.  See INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int32 status, 0 = ok; on failure pg_last_error() holds a
 *     thread-local message (the Julia wrapper turns it into error(msg)).
 *   - handles are opaque and owned by the library; every array argument is a HOST pointer
 *     owned by the caller and borrowed for the duration of the call only.
 *   - all fields live on the padded node grid  M = prod(n_d + 1), linear index
 *     i + j*(nx+1) + k*(nx+1)*(ny+1) (0-based here; src/solver.jl:362-372, src/utils.jl:21).
 *   - float64 / int64 across the ABI (SparseMatrixCSC{Float64,Int}).  No torch types.
 *   - one process drives one GPU; with nranks > 1 every rank owns a slab of planes of the
 *     slowest dimension and the library exchanges halos / dot products over RCCL itself.
 *   - handles are not thread-safe; one call at a time per handle.
 */
#ifndef PENGUIN_HIP_H
#define PENGUIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pg_mesh pg_mesh;         /* Penguin.Mesh{N}            src/mesh.jl:41-79        */
typedef struct pg_capacity pg_capacity; /* Penguin.Capacity{N}        src/capacity.jl:25-36    */
typedef struct pg_diffops pg_diffops;   /* Penguin.DiffusionOps{N}    src/operators.jl:49-55   */
typedef struct pg_solver pg_solver;     /* Penguin.Solver             src/solver.jl:33-42      */

/* ---- enums --------------------------------------------------------------------------- */
enum { PG_BODY_BALL = 1, PG_BODY_MULTIBALL = 2, PG_BODY_HALFSPACE = 3, PG_BODY_ELLIPSOID = 4 }; /* closed-form level sets evaluated in-kernel */
enum { PG_FLAG_COMPLEMENT = 1, PG_FLAG_NO_CENTROIDS = 2 };
/* capacity fields for pg_capacity_get (d = dimension for A,B,W,C_omega,C_gamma) */
enum { PG_CAP_V = 0, PG_CAP_GAMMA = 1, PG_CAP_CELL_TYPES = 2, PG_CAP_A = 3, PG_CAP_B = 4,
       PG_CAP_W = 5, PG_CAP_C_OMEGA = 6, PG_CAP_C_GAMMA = 7,
       /* space-time capacities only: A_(N+1) at the lower / upper time face ("Vn_1", "Vn"), time component of C_ω / C_γ */
       PG_CAP_ST_V0 = 8, PG_CAP_ST_V1 = 9, PG_CAP_ST_CT_OMEGA = 10, PG_CAP_ST_CT_GAMMA = 11 };
enum { PG_OP_G = 0, PG_OP_H = 1, PG_OP_WINV = 2,
       PG_OP_C0 = 3 /* C_d: PG_OP_C0 + d */, PG_OP_K0 = 6 /* K_d: PG_OP_K0 + d */ };   /* ConvectionOps */
/* interface / border condition kinds                         src/boundary.jl:12-50 */
enum { PG_BC_NONE = 0, PG_BC_DIRICHLET = 1, PG_BC_NEUMANN = 2, PG_BC_ROBIN = 3, PG_BC_PERIODIC = 4 };
/* border keys, in the reference's (unusual) naming           src/solver.jl:379-409 */
enum { PG_KEY_LEFT = 0 /* dim2 = 1 */, PG_KEY_RIGHT = 1 /* dim2 = n2 */, PG_KEY_BOTTOM = 2 /* dim1 = 1 */,
       PG_KEY_TOP = 3 /* dim1 = n1 */, PG_KEY_BACKWARD = 4 /* dim3 = 1 */, PG_KEY_FORWARD = 5 /* dim3 = n3 */ };
enum { PG_SCHEME_BE = 0, PG_SCHEME_CN = 1,                   /* "BE" / "CN" strings of the reference */
       PG_SCHEME_STEADY = 2 };                                /* internal: the steady constructors */
enum { PG_METHOD_BICGSTAB = 0, PG_METHOD_CG = 1, PG_METHOD_GMRES = 2 };   /* IterativeSolvers methods kept:
   bicgstab (also serves `\` and bicgstabl), cg, gmres (solve_system!'s default, src/solver.jl:158) */

/* ---- plain structs ------------------------------------------------------------------- */
typedef struct {
  int32_t kind;   /* PG_BC_DIRICHLET / NEUMANN / ROBIN   (build_I_bc, src/solver.jl:203-223) */
  double alpha;   /* Robin only */
  double beta;    /* Robin only */
  double value;   /* constant value, used when value_array == NULL */
  const double* value_array; /* M host values g(C_gamma, t) or NULL (build_g_g, src/solver.jl:293-323) */
} pg_bc_desc;

typedef struct {
  int32_t key;    /* PG_KEY_*  */
  int32_t kind;   /* PG_BC_DIRICHLET / PERIODIC / NEUMANN(1-D only)  (src/solver.jl:450-499) */
  double value;   /* constant; per-cell values may be set later with pg_solver_set_border_values */
} pg_border_desc;

typedef struct {
  double alpha1, alpha2, g;  /* ScalarJump  src/boundary.jl:96-100 ; row 2 of the 4-block system */
  double beta1, beta2, h;    /* FluxJump    src/boundary.jl:111-115; row 4 */
  const double* g_array;     /* M values or NULL */
  const double* h_array;     /* M values or NULL */
} pg_jump_desc;

typedef struct {
  int32_t method;   /* PG_METHOD_*                                  */
  double reltol;    /* ||r|| <= max(reltol*||b||, abstol)            */
  double abstol;
  int32_t maxiter;  /* <= 0: size of the reduced system              */
  int32_t check_every; /* host convergence poll period in iterations (<=0: default 4) */
  int32_t warm_start;  /* != 0: pg_solver_step starts BiCGStab from the previous time level instead of zero
                          (IterativeSolvers starts from zero; same solution to reltol, fewer iterations) */
  int32_t restart;     /* GMRES only: Krylov vectors per restart cycle (<= 0: 20, IterativeSolvers' default) */
  int32_t precond;     /* BiCGStab only: polynomial right preconditioner in Â (DESIGN.md "Krylov driver").  0: automatic
                          (on where Gershgorin bounds the spectrum; degree 6 for a first solve, then chosen per solve of the
                          warm time loop from the previous solve's convergence rate, 4 .. 32); -1: off (the plain
                          iteration IterativeSolvers runs); m >= 1: m products with Â per application where admissible (<= 40) */
} pg_krylov_opts;

typedef struct {
  int32_t iters;      /* Krylov iterations of this step                        */
  int32_t converged;
  double resnorm;     /* final ||r||                                           */
  double bnorm;
  double extremum;    /* maximum(abs.(s.x)) -- the "Solver Extremum" line, src/solver/diffusion.jl:279 */
  double time;        /* t after the step (fp64 accumulation t += dt, :287)     */
} pg_step_info;

typedef struct {
  int64_t steps;            /* number of loop iterations executed (states = steps + 1)         */
  int64_t total_iters;
  double t_final;
  double extremum;
  double solve_ms;          /* wall time of the whole run on this rank                          */
  /* profiling (pg_set_profiling(1)): HIP-event time of the sampled SpMV launches that carry fused dots (the plain
     iteration's only kind; with the polynomial preconditioner the launch that closes each chain of products)      */
  double spmv_ms_total;
  int64_t spmv_launches;
  int64_t unconverged_steps; /* solves of this run that ended without meeting the tolerance (maxiter / breakdown) */
  double worst_relres;       /* max over the run's solves of ||r|| / ||b|| at exit                                */
  double spmv_lean_ms_total; /* profiling: the sampled LEAN launches -- factors of the preconditioner polynomial, one   */
  int64_t spmv_lean_launches;/* vector in, one out, no dots (DESIGN.md "Krylov driver")                                 */
  int64_t poly_degree;       /* products with Â per application of the preconditioned operator in the last solve (0: plain) */
  int64_t half_exits;        /* solves that met the tolerance after the first half of their last iteration (such an
                                iteration counts as one in total_iters but runs only one application of the operator)  */
  int64_t poly_xspace;       /* 1: the last solve ran the x-space form of the preconditioned loop: the m - 1 launches of a chain
                                are Horner steps (x, the chain's input and the matrix in, one vector out; the first of a chain
                                reads one vector), no recovery; 0: y-space form (lean chains + recovery) or plain iteration       */
  int64_t products;          /* products with the system matrix inside the run's solves (the one product per step that builds the
                                right-hand side and the start residual not counted)                                             */
  int64_t guess_states_read; /* older states read by the extrapolated starts of the run's quiet steps (pg_solver_guess_info):
                                each is two more vector reads of the step's first kernel                                        */
} pg_run_info;

typedef struct {
  int64_t n_own;      /* rows of the reduced system owned by this rank (n of BASELINE.md)       */
  int64_t nnz;        /* stored entries of those rows                                           */
  int64_t n_ghost;    /* ghost unknowns received from neighbours per SpMV                       */
  int64_t n_omega;    /* owned active bulk unknowns                                             */
  int64_t n_gamma;    /* owned active interface unknowns                                        */
  int64_t M_global;   /* prod(n_d+1)                                                            */
  /* stencil-sliced SpMV image of the matrix (DESIGN.md "SpMV"):                                */
  int64_t spmv_bytes;     /* bytes one y = A x launch has to move with this format               */
  int64_t spmv_slices;    /* U + P slices + G chunks                                             */
  int64_t rows_uniform;   /* rows in U slices (shared offsets and values, nothing streamed)      */
  int64_t rows_pattern;   /* rows in P slices (shared offsets, 8 B/entry value stream)           */
  int64_t rows_irregular; /* rows in G chunks (packed CSR, 12 B/entry)                           */
  int64_t neumann_ok;     /* 1: the spectrum of the preconditioned matrix is provably inside |z - 1| < 0.95: BiCGStab
                             runs right-preconditioned with a Chebyshev polynomial in Â (pg_krylov_opts.precond)   */
  double gershgorin;      /* the largest Gershgorin radius that decision rests on (all ranks)     */
  int64_t spmv_units;     /* marching units (DESIGN.md "SpMV"): <= 11 planes x 126 uniform stencil rows each      */
  int64_t rows_marched;   /* rows covered by them (counted in rows_uniform too; they are in no slice)             */
  int64_t rows_matrix;    /* rows of the matrix the other SpMV figures describe: n_own, or fewer for which = 6 / 7 (the
                             warm loop's matrix without the rows that are alone on their diagonal)                   */
  int64_t n_ghost_loop;   /* ghost unknowns of the vectors that matrix works on (which = 6 / 7: the compact loop system
                             keeps the remaining ghosts only; otherwise = n_ghost)                                      */
  int64_t loop_is_compact;/* 1: the warm loop of this system iterates on the compact system (rows alone on their diagonal
                             solved in the right-hand-side pass, pg_reduce.hip) -- on several ranks with halos too     */
} pg_system_info;

/* ---- library / device -------------------------------------------------------------------- */
int32_t pg_last_error(char* buf, size_t n);
/* single-GPU: pg_init(device).  multi-GPU: rank 0 calls pg_get_unique_id, the launcher broadcasts
   the 128 bytes (torch.distributed in bench.py), every rank calls pg_init_distributed. */
int32_t pg_init(int32_t device_id);
int32_t pg_get_unique_id(void* out128);
int32_t pg_init_distributed(int32_t device_id, int32_t rank, int32_t nranks, const void* unique_id128);
int32_t pg_finalize(void);
int32_t pg_device_synchronize(void);
int32_t pg_set_profiling(int32_t on);
int32_t pg_device_name(char* buf, size_t n);
/* the tuning / variant selectors this process runs with (read once from the PG_* environment, pg_common.h Config), as
   "name=value ..." -- so that a measurement can say exactly which variant of the path it timed */
int32_t pg_config_string(char* buf, size_t n);

/* ---- Mesh                                       replaces Mesh(n, L, x0), src/mesh.jl:47-78 ---- */
int32_t pg_mesh_create(int32_t N, const int64_t* n, const double* L, const double* x0, pg_mesh** out);
int32_t pg_mesh_destroy(pg_mesh* m);
int32_t pg_mesh_get_centers(const pg_mesh* m, int32_t d, double* out, int64_t len);   /* len = n_d     */
int32_t pg_mesh_get_nodes(const pg_mesh* m, int32_t d, double* out, int64_t len);     /* len = n_d + 1 */
int32_t pg_mesh_num_border_cells(const pg_mesh* m, int64_t* out);
/* idx: nb*N 1-based Cartesian indices, pos: nb*N centre coordinates, key: nb PG_KEY_*, in the
   reference's order (src/mesh.jl:57-74 + unique!) */
int32_t pg_mesh_get_border_cells(const pg_mesh* m, int64_t* idx, double* pos, int32_t* key);

/* ---- Capacity                      replaces Capacity(body, mesh; method="VOFI"), capacity.jl:51-123 */
/* BALL: params = {c_1..c_N, r}.  MULTIBALL: params = {r, nballs, c^1_1..c^1_N, c^2_1, ...}.
   HALFSPACE: params = {axis (0-based), position, sign}: f(x) = sign (x_axis - position), fluid where f < 0 -- the
   reference's 1-D diphasic bodies `(x,_=0) -> x - xint` (test/convergence_test.jl:111,230) and their N-D extrusions.
   ELLIPSOID: params = {c_1..c_N, a_1..a_N}: f(x) = sqrt(sum ((x_d - c_d)/a_d)^2) - 1, axis-aligned semi-axes a_d > 0
   (volumes, faces, sections exact through the unit ball of the scaled coordinates; the interface measure by
   Gauss-Legendre along the arcs with the weight of the affine map). */
int32_t pg_capacity_create_levelset(pg_mesh* m, int32_t body_kind, const double* params, int32_t nparams,
                                    int32_t flags, pg_capacity** out);
/* fallback for arbitrary Julia bodies: arrays computed by the caller (single rank only).
   A,B,W,C_omega,C_gamma: N pointers each to M doubles; C_gamma may be NULL. */
int32_t pg_capacity_create_from_arrays(pg_mesh* m, const double* V, const double* const* A,
                                       const double* const* B, const double* const* W, const double* Gamma,
                                       const double* const* C_omega, const double* const* C_gamma,
                                       const double* cell_types, pg_capacity** out);
/* Capacity(body, SpaceTimeMesh(mesh, [t0, t1])), prescribedmotionsolver/diffusion.jl:251-252 (mesh.jl:129-146): the first
   time layer of the (N+1)-D capacity of a MOVING ball / half space, N = 1 or 2.  The host evaluates the motion at the
   time-quadrature nodes it chooses (composite Gauss-Legendre inside (t0, t1), weights adding up to t1 - t0) and at the
   two time faces; space is integrated exactly as by pg_capacity_create_levelset, time by the given rule.
   nodes: nq x 10 doubles {tau, weight, c_1, c_2, c_3, r, dc_1/dt, dc_2/dt, dc_3/dt, dr/dt}; half space: c_1 = position,
   dc_1/dt = its speed, r ignored.  body0 / body1 = {c_1, c_2, c_3, r} at t0 / t1.
   pg_capacity_get: V, A_d, B_d, W_d, Γ are the first-layer space-time measures, C_ω / C_γ the spatial centroid components;
   PG_CAP_ST_* give A_(N+1) at the two time faces and the time components of the centroids. */
typedef struct {
  int32_t body_kind;   /* PG_BODY_BALL | PG_BODY_HALFSPACE */
  int32_t flags;       /* PG_FLAG_COMPLEMENT | PG_FLAG_NO_CENTROIDS */
  int32_t axis;        /* half space: 0-based axis */
  int32_t nq;
  double sign;         /* half space: f = sign (x_axis - position(t)) */
  double t0, t1;
  const double* nodes; /* nq x 10 */
  double body0[4], body1[4];
} pg_motion_desc;
int32_t pg_capacity_create_spacetime(pg_mesh* m, const pg_motion_desc* motion, pg_capacity** out);
/* marks a capacity built by pg_capacity_create_from_arrays as the first layer of a space-time capacity (arbitrary moving
   bodies integrated by the caller): V_t0 / V_t1 = A_(N+1) at the time faces (M doubles each), Ct_* may be NULL */
int32_t pg_capacity_set_spacetime(pg_capacity* c, double t0, double t1, const double* V_t0, const double* V_t1,
                                  const double* Ct_omega, const double* Ct_gamma);
int32_t pg_capacity_destroy(pg_capacity* c);
/* out: M doubles (global padded layout; with nranks>1 only the planes this rank stores are filled,
   the rest are left untouched) */
int32_t pg_capacity_get(const pg_capacity* c, int32_t field, int32_t d, double* out, int64_t len);
int32_t pg_capacity_kernel_ms(const pg_capacity* c, double* ms); /* device time of K1-K5 for the bench */

/* ---- DiffusionOps                         replaces DiffusionOps(capacity), operators.jl:127-178 */
int32_t pg_diffops_create(pg_capacity* c, pg_diffops** out);
int32_t pg_diffops_destroy(pg_diffops* o);
/* two-call pattern: nzval == NULL -> only *nnz is written.  CSC of G, H (N*M x M) or Winv (N*M x N*M),
   0-based colptr (ncols+1) / rowval, single rank only. */
int32_t pg_diffops_export_csc(const pg_diffops* o, int32_t which, int64_t* colptr, int64_t* rowval,
                              double* nzval, int64_t* nnz);
/* ConvectionOps(capacity, uₒ, uᵧ), operators.jl:194-210: C_d = δ_p[d]·diag(Σ_m[d] A_d uₒ_d)·Σ_m[d], K_d = diag(Σ_p[d] Hᵀuᵧ).
   u_omega: N pointers to M doubles, u_gamma: N*M doubles (block d = component d).  A solver created from an operator
   with a velocity assembles the advection-diffusion blocks (advectiondiffusion.jl:29-44,180-213: conv_bulk = ΣC_d and
   conv_iface = ½ΣK_d on the bulk rows); pg_diffops_export_csc(PG_OP_C0 + d / PG_OP_K0 + d) returns C_d / K_d. */
int32_t pg_diffops_set_velocity(pg_diffops* o, const double* const* u_omega, const double* u_gamma);
int32_t pg_diffops_grad(const pg_diffops* o, const double* p /*2M*/, double* out /*N*M*/);      /* ∇  :20-23 */
int32_t pg_diffops_div(const pg_diffops* o, const double* qw /*N*M*/, const double* qg /*N*M*/,
                       double* out /*M*/);                                                     /* ∇₋ :30-34 */

/* ---- Solver          replaces DiffusionUnsteadyMono(phase, bc_b, bc_i, Δt, Tᵢ, scheme), diffusion.jl:192-210
   Dcoef: M values D(C_omega) or NULL (=1).  source: M values f(C_omega, Δt) for the first step or NULL (=0).
   T0: 2M, or NULL for zeros.  Builds A (scheme), b(t=0) and applies the border rows with t = 0 exactly as the ctor does. */
int32_t pg_solver_create_unsteady_mono(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                       const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                       const double* source, double dt, const double* T0, int32_t scheme,
                                       pg_solver** out);
/* DiffusionUnsteadyDiph(phase1, phase2, bc_b, ic, Δt, Tᵢ, scheme), diffusion.jl:319-332.  T0: 4M. */
int32_t pg_solver_create_unsteady_diph(pg_capacity* c1, pg_diffops* o1, pg_capacity* c2, pg_diffops* o2,
                                       const pg_jump_desc* ic, const pg_border_desc* borders, int32_t nborders,
                                       const double* D1, const double* D2, const double* f1, const double* f2,
                                       double dt, const double* T0, int32_t scheme, pg_solver** out);
/* DiffusionSteadyMono(phase, bc_b, bc_i), diffusion.jl:14-28 (A_mono_stead_diff :30-43, b_mono_stead_diff :45-58):
   the same blocks without V and Δt; source = f(C_ω) (no time).  solve_DiffusionSteadyMono! (:60-71) =
   pg_solver_initial_solve; pg_solver_step / _run refuse a steady solver. */
/* MovingDiffusionUnsteadyDiph + A_/b_diph_unstead_diff_moving (prescribedmotionsolver/diffusion.jl:272-498): ONE space-time
   slab of the two-phase moving problem.  c1 / c2: space-time capacities of the body and of its complement on the same slab
   (pg_capacity_create_spacetime); T_prev: the previous state [Tω¹; Tγ¹; Tω²; Tγ²] (4M, host) or NULL with `previous` = the
   previous slab's solver (state handed over on the device); f*_n / f*_np1: sources at t and t + Δt evaluated at the
   space-time centroids (NULL = 0).  Solve with pg_solver_initial_solve; the next slab is a new solver. */
int32_t pg_solver_create_moving_diph(pg_capacity* c1, pg_diffops* o1, pg_capacity* c2, pg_diffops* o2, const pg_jump_desc* ic,
                                     const pg_border_desc* borders, int32_t nborders, const double* D1, const double* D2,
                                     const double* f1_n, const double* f1_np1, const double* f2_n, const double* f2_np1,
                                     const double* T_prev, pg_solver* previous, int32_t scheme, pg_solver** out);
int32_t pg_solver_create_steady_mono(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                     const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                     const double* source, pg_solver** out);
/* DiffusionSteadyDiph(phase1, phase2, bc_b, ic), diffusion.jl:88-101 (A_diph_stead_diff :103-144,
   b_diph_stead_diff :146-162); solve_DiffusionSteadyDiph! (:164-175) = pg_solver_initial_solve. */
int32_t pg_solver_create_steady_diph(pg_capacity* c1, pg_diffops* o1, pg_capacity* c2, pg_diffops* o2,
                                     const pg_jump_desc* ic, const pg_border_desc* borders, int32_t nborders,
                                     const double* D1, const double* D2, const double* f1, const double* f2,
                                     pg_solver** out);
/* One space-time step of solve_MovingDiffusionUnsteadyMono! (prescribedmotionsolver/diffusion.jl:16-35, 100-160, 163-225,
   249-258): A = [Vn_1 + Id GᵀWꜝG Ψ, -(Vn_1 - Vn) + Id GᵀWꜝH Ψ; Iᵦ HᵀWꜝG, Iᵦ HᵀWꜝH + Iₐ Γ] on the space-time capacity c
   (Δt is inside it), b from T_prev (2M, or NULL for zeros), border rows as BC_border_mono!.  source_n / source_np1 =
   f(C_ω.., t) and f(C_ω.., t+Δt) (M doubles or NULL = 0; the first is read by "CN" only).  pg_solver_initial_solve solves
   it; pg_solver_get_state(-1) is the new state, the T_prev of the next slab's solver.  pg_solver_step refuses. */
int32_t pg_solver_create_moving_mono(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                     const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                     const double* source_n, const double* source_np1, const double* T_prev,
                                     int32_t scheme, pg_solver** out);
/* the same, with the state of the previous slab's solver (solved, same mesh) as T_prev -- device to device: the time loop
   then moves no state over PCIe unless the caller asks for one (pg_solver_get_state) */
int32_t pg_solver_create_moving_mono_next(pg_capacity* c, pg_diffops* o, const pg_bc_desc* bc_interface,
                                          const pg_border_desc* borders, int32_t nborders, const double* Dcoef,
                                          const double* source_n, const double* source_np1, pg_solver* previous,
                                          int32_t scheme, pg_solver** out);
int32_t pg_solver_destroy(pg_solver* s);

/* per-step data for time-dependent closures; the host evaluates them at the reference's points and
   times (C_omega / C_gamma / mesh.centers; t+Δt -- diffusion.jl:248-249) and passes arrays.
   f_n, g_n (values at t) are only read by the CN scheme; NULL keeps the previous / constant data. */
int32_t pg_solver_set_source(pg_solver* s, int32_t phase, const double* f_n /*M*/, const double* f_np1 /*M*/);
int32_t pg_solver_set_interface_value(pg_solver* s, const double* g_n /*M*/, const double* g_np1 /*M*/);
int32_t pg_solver_set_border_values(pg_solver* s, const double* values /*nb, mesh border order*/);

/* first solve of solve_DiffusionUnsteadyMono! (diffusion.jl:275): uses the ctor's A, b. */
int32_t pg_solver_initial_solve(pg_solver* s, const pg_krylov_opts* opts, pg_step_info* info);
/* one loop iteration (diffusion.jl:286-300): t += Δt; b; border rows; solve. scheme = run scheme. */
int32_t pg_solver_step(pg_solver* s, int32_t scheme, const pg_krylov_opts* opts, pg_step_info* info);
/* whole loop `while t < Tend` with constant-in-time data; max_steps < 0: unbounded.
   do_initial != 0 runs the initial solve first.  save_every > 0 keeps every k-th state on the device. */
int32_t pg_solver_run(pg_solver* s, double Tend, int32_t scheme, const pg_krylov_opts* opts, int32_t do_initial,
                      int64_t max_steps, int32_t save_every, pg_run_info* info);
int32_t pg_solver_num_states(const pg_solver* s, int64_t* out);
/* x: full unknown vector (2M mono / 4M diph) with zeros at eliminated unknowns (solver.jl:186-187).
   state_index < 0: current s.x.  With nranks>1 only owned entries are written (others untouched). */
int32_t pg_solver_get_state(const pg_solver* s, int64_t state_index, double* x, int64_t len);
/* which: 0 = constructor system, 1 = run (loop) system: the reference's reduced A x = b;
          2 / 3 = the same two systems as the Krylov solver iterates on them, left-preconditioned and
          equilibrated: (B^-1 S A S) y = B^-1 S b, x = S y (DESIGN.md "Preconditioner"); nnz differs.
          pg_solver_system_info only: 6 / 7 = what the warm time loop iterates on: the same matrix without the interface
          unknowns of a Dirichlet problem (rows of the identity, solved before the iteration -- DESIGN.md "Dirichlet
          interface rows") once a loop step has built it; identical to 2 / 3 otherwise. */
int32_t pg_solver_system_info(const pg_solver* s, int32_t which, pg_system_info* out);
/* reduced system of this rank as CSR (rowptr n_own+1, col nnz (local numbering: owned then ghosts), val nnz)
   plus b (n_own) and the map idx[n_own] -> index in the full 2M/4M vector (the reference's common_idx). */
int32_t pg_solver_get_system_csr(const pg_solver* s, int32_t which, int64_t* rowptr, int64_t* col, double* val,
                                 double* b, int64_t* idx);
/* S = diag(|a_ii|^-1/2) of system `which` (0 constructor, 1 run): the weights of the convergence test
   ||S r̂|| <= reltol ||S b̂|| (DESIGN.md "Krylov driver") and the map x = S y of the preconditioned system.  ds: n_own. */
int32_t pg_solver_get_row_scaling(const pg_solver* s, int32_t which, double* ds);
/* bench helper: `reps` launches of y = A x on the run matrix, HIP-event timed on the library stream. */
int32_t pg_solver_time_spmv(pg_solver* s, int32_t which, int32_t reps, double* avg_ms);

/* host-logic helper (no GPU needed): slab partition of `nplanes` planes with per-plane weights
   (active rows) into nranks contiguous ranges; bounds has nranks+1 entries. */
int32_t pg_partition_planes(const int64_t* weight, int64_t nplanes, int32_t nranks, int64_t* bounds);
/* host-logic helper (no GPU needed): the local numbering of the slab that owns planes [p0, p1) of `nplanes` planes of
   `plane` cells -- segment offsets, ghost segments and the contiguous chunks exchanged with the neighbours (SURVEY 8e;
   the code path build_numbering on the device ends in).  active: K * Mloc flags, kind-major, over the slab's STORED planes
   [max(p0 - 3, 0), min(p1 + 3, nplanes)), Mloc = stored planes * plane.  out (2 + 10 K int64): n_own, n_ghost, then K
   entries each of cnt_own, off_own, cntL, offL, cntU, offU, sendL_off, sendL_cnt, sendU_off, sendU_cnt (offsets into the
   local vector [owned kinds | lower ghosts | upper ghosts]). */
int32_t pg_slab_numbering_host(int32_t K, int64_t plane, int64_t nplanes, int64_t p0, int64_t p1, const uint8_t* active,
                               int64_t* out);

/* ---- diagnostics (not part of the reference-facing boundary) ------------------------------------------------ */
/* y = A x with two kernel variants (PG_SPMV_VARIANT numbering: 70 stencil slices, 38 chunked CSR, 1 first CSR kernel)
   on the same deterministic vector: max |y_a - y_b| and max |y_a| */
int32_t pg_debug_spmv_compare(pg_solver* s, int32_t which, int32_t variant_a, int32_t variant_b, double* max_abs_diff,
                              double* max_abs);
/* read-only streaming probe: `bytes` read `reps` times with elem_bytes (4|8) per lane, optional non-temporal hint */
int32_t pg_debug_read_probe(int64_t bytes, int32_t elem_bytes, int32_t nt, int32_t blocks, int32_t reps, double* gbs);
/* run the slab-decomposed monophasic path with `nranks` VIRTUAL ranks (host threads sharing this GPU, in-process
   halo exchange / all-reduce standing in for RCCL): Dirichlet(interface_value) on the body, Dirichlet(border_value)
   on `keys`, T0 = 0, initial solve with scheme_ctor then `steps` steps with scheme_run.  x_out: 2M, or NULL (full-size
   rehearsals: sizes and iteration counts only). */
int32_t pg_debug_run_virtual_ranks(int32_t nranks, int32_t N, const int64_t* n, const double* L, int32_t body_kind,
                                   const double* params, int32_t nparams, double interface_value, double border_value,
                                   int32_t nkeys, const int32_t* keys, double dt, int32_t scheme_ctor, int32_t scheme_run,
                                   int64_t steps, double* x_out, int64_t* n_own_out, int64_t* nnz_out,
                                   int64_t* n_ghost_out, int64_t* iters_out);
/* Krylov method (PG_METHOD_*) and GMRES restart length of the virtual-rank runs that follow (default BiCGStab) */
int32_t pg_debug_set_virtual_rank_method(int32_t method, int32_t restart);
/* ramp != 0: the virtual-rank runs that follow change the interface value every step, g = interface_value (1 + ramp step)
   (host-driven steps): rows alone on their diagonal move in every step, on every rank */
int32_t pg_debug_set_virtual_rank_ramp(double ramp);
/* per virtual rank of the last run: rows of the full system, rows / bytes per launch / ghost entries of the system the
   warm loop iterated on (the compact one when loop_rows < full_rows) */
int32_t pg_debug_virtual_rank_info(int32_t rank, int64_t* full_rows, int64_t* loop_rows, int64_t* loop_bytes,
                                   int64_t* loop_ghosts);
/* test hook: multiply the state at every row that is alone on its diagonal by `factor` WITHOUT telling the time loop --
   the next quiet step must notice (S_MOVED) and still end at the right state */
int32_t pg_debug_scale_diagonal_rows(pg_solver* s, double factor);

/* The extrapolated start of the time loop's quiet steps (constant data, one rank; no reference counterpart: the reference
   starts every Krylov solve from zero, solver.jl:158-181): `kept` older states are held; the next step starts from
   z^n + sum_j coef[j] (z^(n-offsets[j]) - z^n), j < nstates <= 4, chosen by a least-squares fit of this step's plain start
   residual (sampled weighted squares: rr_plain without, rr_taken with this step's extrapolation).  PG_GUESS_STATES=0: off. */
int32_t pg_solver_guess_info(pg_solver* s, int32_t* kept, int32_t* nstates, int32_t* offsets /* [4] */, double* coef /* [4] */,
                             double* rr_plain, double* rr_taken);

#ifdef __cplusplus
}
#endif
#endif /* PENGUIN_HIP_H */
