// pg_stencil.h -- the rows of the reference's block systems, written out as stencils.
//
// The reference never forms a stencil: it multiplies Kronecker-lifted sparse matrices
//   G = vcat_d(D⁻_d B_d), H = vcat_d(A_d D⁻_d − D⁻_d B_d), Wꜝ = diag(1/W or 1)      src/operators.jl:138-152
//   mono   [V+θ Id GᵀWꜝG, θ Id GᵀWꜝH ; Iᵦ HᵀWꜝG, Iᵦ HᵀWꜝH + Iₐ Γ]                   src/solver/diffusion.jl:212-241
//   diph   4-block system ordered [Tω¹;Tγ¹;Tω²;Tγ²]                                src/solver/diffusion.jl:334-389
// and then overwrites the border rows (src/solver.jl:417-499, 540-580).  eval_row() produces the
// same row entry by entry from the capacities, including the edge rule
// D⁻[m,m] = 0 (src/operators.jl:9) and the border-row surgery, so that K7/K9/K10 need no sparse
// algebra on the device.
#pragma once
#include "pg_capacity.h"

namespace pg {

constexpr int MAX_KINDS = 4;  // mono: ω,γ   diph: ω1,γ1,ω2,γ2

struct SysParams {
  int nphase;            // 1 = monophasic, 2 = diphasic
  CapView cap[2];
  const double* ct[2];   // cell types (diph border rows are skipped where the phase is absent)
  const double* Id[2];   // D(C_ω) per local cell, or nullptr (= 1)
  double Ia, Ib;         // mono interface condition (build_I_bc)
  double a1, a2, b1, b2; // diph jump coefficients α₁ α₂ β₁ β₂
  double theta;          // Δt (BE) or Δt/2 (CN): multiplies the diffusion part of bulk rows
  double gscale;         // mono interface rows: 1 (BE) or Δt/2 (CN)
  // convection (advection-diffusion, src/solver/advectiondiffusion.jl:180-213): per phase, or nullptr
  const double* conv_a[2][3];   // a_d of C_d = δ_p diag(a_d) Σ_m
  const double* conv_k[2];      // diagonal of 0.5 * sum(K): added to the ω-ω and ω-γ diagonals of bulk rows
  double mass;           // coefficient of V in the bulk rows: 1 (unsteady), 0 (steady: A_mono_stead_diff, diffusion.jl:30-43)
  int border_kind[6];    // per PG_KEY_*
  int border_both_phases; // diph: 1 = border rows in both phases whatever the cell type (moving driver), 0 = skipped where the phase is absent
  double inv_dx;         // 1/Δx for the 1-D Neumann border row
  // moving body, mono (prescribedmotionsolver/diffusion.jl:100-160): cap[0] holds SPACE-TIME capacities (Δt is inside
  // them: theta = gscale = 1) and the bulk rows become
  //   [ Vn_1 + Id GᵀWꜝG Ψ ,  -(Vn_1 - Vn) + Id GᵀWꜝH Ψ ]      Ψ = diag(psip.(Vn, Vn_1)) scales COLUMNS
  // moving body, diph (prescribedmotionsolver/diffusion.jl:292-398): the same per phase, and the FLUX row carries the
  // column scaling and the swept-volume term too:
  //   [ β₁ H₁ᵀWꜝG₁Ψ₁ , β₁ H₁ᵀWꜝH₁Ψ₁ - (Vn1_1 - Vn1) , β₂ H₂ᵀWꜝG₂Ψ₂ , β₂ H₂ᵀWꜝH₂Ψ₂ - (Vn2_1 - Vn2) ]      (:378-381)
  // mv_psi_w[0] == nullptr: static problem (everything below unused); index = phase
  const double* mv_v0[2];     // "Vn_1" = A_t at the lower time face, V(t_n)
  const double* mv_v1[2];     // "Vn"   = A_t at the upper time face, V(t_n + Δt)
  const double* mv_psi_w[2];  // per cell: scaling of the ω columns
  const double* mv_psi_g[2];  // per cell: scaling of the γ columns; nullptr: the constant mv_gconst
  double mv_gconst;
  int mv_explicit;         // 1: the explicit Crank-Nicolson operator [Id GᵀWꜝG Ψn, ½ Id GᵀWꜝH] of :214 -- bulk rows only, no
                           // volume terms; interface rows are empty (b2 = Γ g carries no T term there)
};

__host__ __device__ inline int nkinds(const SysParams& P) { return P.nphase == 1 ? 2 : 4; }

// 1-D pieces of dimension d around local cell lc (index j = idx[d]):
//   gd_k = B_k [k<m], gl_k = −B_{k−1} [k≥1], hd_k = (A_k − B_k)[k<m], hl_k = −(A_k − B_{k−1})[k≥1]
// (0-based k, m = ext_d − 1 is the padding index).
struct Line {
  double gd_j, gl_j, hd_j, hl_j, w_j;        // row k = j
  double gd_p, gl_p, hd_p, hl_p, w_p;        // row k = j+1 (zeros if j+1 > m)
  bool has_m, has_p;                         // neighbours j−1 / j+1 exist on the grid
};

__device__ inline double winv(double w) { return w != 0.0 ? 1.0 / w : 1.0; }  // operators.jl:149-151

__device__ inline Line load_line(const CapView& c, int d, i64 lc, i64 j) {
  Line L;
  const i64 m = c.ext[d] - 1;
  const i64 st = c.stride[d];
  const double* A = c.A[d];
  const double* B = c.B[d];
  const double* W = c.W[d];
  L.has_m = j >= 1;
  L.has_p = j + 1 <= m;
  const double Bj = B[lc], Aj = A[lc];
  const double Bm = L.has_m ? B[lc - st] : 0.0;
  L.w_j = winv(W[lc]);
  L.gd_j = j < m ? Bj : 0.0;
  L.hd_j = j < m ? (Aj - Bj) : 0.0;
  L.gl_j = L.has_m ? -Bm : 0.0;
  L.hl_j = L.has_m ? -(Aj - Bm) : 0.0;
  if (L.has_p) {
    const double Bp = B[lc + st], Ap = A[lc + st];
    L.w_p = winv(W[lc + st]);
    L.gd_p = (j + 1) < m ? Bp : 0.0;
    L.hd_p = (j + 1) < m ? (Ap - Bp) : 0.0;
    L.gl_p = -Bj;
    L.hl_p = -(Ap - Bj);
  } else {
    L.w_p = 0.0; L.gd_p = L.gl_p = L.hd_p = L.hl_p = 0.0;
  }
  return L;
}

// is (phase ph) bulk row of this cell overwritten by a border condition?  returns the kind
__device__ inline int border_row_kind(const SysParams& P, int ph, i64 lc, const i64* idx, int* key_out) {
  const CapView& c = P.cap[0];
  for (int d = 0; d < c.N; ++d)
    if (idx[d] >= c.n[d]) return PG_BC_NONE;  // padding cells are never border cells
  const int key = border_key_of(c.N, c.n, idx[0], idx[1], idx[2]);
  if (key < 0) return PG_BC_NONE;
  const int kind = P.border_kind[key];
  if (kind == PG_BC_NONE) return PG_BC_NONE;
  // solver.jl:574-575 -- the capacity-aware method the static drivers call; the moving diphasic driver calls
  // BC_border_diph!(A, b, bc_b, mesh) (prescribedmotionsolver/diffusion.jl:288,523), the method WITHOUT capacities
  // (solver.jl:540-543): both phases get their border rows wherever the mesh has a border cell
  if (P.nphase == 2 && !P.border_both_phases && P.ct[ph][lc] == 0.0) return PG_BC_NONE;
  if (kind == PG_BC_PERIODIC) {
    // needs the opposite key present (solver.jl:461) and is 2-D only (:512-519)
    const int opp = key ^ 1;
    if (P.border_kind[opp] == PG_BC_NONE || c.N != 2) return PG_BC_NONE;
  }
  if (kind == PG_BC_NEUMANN && c.N != 1) return PG_BC_NONE;  // solver.jl:474-496
  *key_out = key;
  return kind;
}

// emit(colkind, col_local_cell, value) for every structurally possible entry of row (kind, lc).
template <class F>
__device__ inline void eval_row(const SysParams& P, int kind, i64 lc, const i64* idx, F&& emit) {
  const int ph = kind >> 1;           // phase of the row's own unknown (mono: 0)
  const bool bulk = (kind & 1) == 0;  // ω row / γ row
  const CapView& c0 = P.cap[0];
  const int N = c0.N;
  const bool mv = P.mv_psi_w[0] != nullptr;
  if (mv && P.mv_explicit && !bulk) return;

  if (bulk) {
    int key = -1;
    const int bk = border_row_kind(P, ph, lc, idx, &key);
    if (bk == PG_BC_DIRICHLET) {      // A[row,:] = 0; A[row,row] = 1        solver.jl:452-456
      emit(kind, lc, 1.0);
      return;
    }
    if (bk == PG_BC_PERIODIC) {       // x_row − x_partner = 0                solver.jl:458-469
      // partner index is built on the PADDED extents (solver.jl:508-519): left/bottom -> ext-1, right/top -> 0
      i64 partner = lc;
      if (key == PG_KEY_LEFT) partner = lc + (c0.ext[1] - 1 - idx[1]) * c0.stride[1];
      else if (key == PG_KEY_RIGHT) partner = lc - idx[1] * c0.stride[1];
      else if (key == PG_KEY_BOTTOM) partner = lc + (c0.ext[0] - 1 - idx[0]) * c0.stride[0];
      else if (key == PG_KEY_TOP) partner = lc - idx[0] * c0.stride[0];
      emit(kind, lc, 1.0);
      emit(kind, partner, -1.0);
      return;
    }
    if (bk == PG_BC_NEUMANN) {        // (u_b − u_adj)/Δx = g, 1-D only        solver.jl:471-493
      const i64 last = c0.ext[0] - 1;
      i64 adj = (key == PG_KEY_BOTTOM) ? (idx[0] + 1 > last ? last : idx[0] + 1) : (idx[0] - 1 < 0 ? 0 : idx[0] - 1);
      emit(kind, lc, P.inv_dx);
      emit(kind, lc + (adj - idx[0]), -P.inv_dx);
      return;
    }
  }

  if (P.nphase == 2 && kind == 1) {   // scalar jump row: α₁Tγ¹ − α₂Tγ² = g   diffusion.jl:374-376
    emit(1, lc, P.a1);
    emit(3, lc, -P.a2);
    return;
  }

  // phases contributing to this row: bulk rows and mono γ rows use their own phase; the diph flux row
  // (kind 3) sums both phases.
  const int ph_lo = (P.nphase == 2 && kind == 3) ? 0 : ph;
  const int ph_hi = (P.nphase == 2 && kind == 3) ? 1 : ph;
  for (int q = ph_lo; q <= ph_hi; ++q) {
    const CapView& c = P.cap[q];
    const int kw = 2 * q, kg = 2 * q + 1;   // column kinds ω_q, γ_q
    double dW = 0.0, dG = 0.0;              // diagonal sums over dimensions
    double dC = 0.0;                        // diagonal of sum(C) (convection)
    double scale;                           // what multiplies the stencil sums
    if (bulk) scale = P.theta * (P.Id[q] ? P.Id[q][lc] : 1.0);     // θ·Id   (row scaling)
    else if (P.nphase == 1) scale = P.gscale * P.Ib;               // (Δt/2)·Iᵦ
    else scale = q == 0 ? P.b1 : P.b2;                             // β_q
    const bool need = bulk || scale != 0.0;
    if (need) {
      for (int d = 0; d < N; ++d) {
        const Line L = load_line(c, d, lc, idx[d]);
        const i64 st = c.stride[d];
        if (bulk) {
          // GᵀWꜝG and GᵀWꜝH rows
          dW += L.gd_j * L.gd_j * L.w_j + L.gl_p * L.gl_p * L.w_p;
          dG += L.gd_j * L.w_j * L.hd_j + L.gl_p * L.w_p * L.hl_p;
          // row k of C_d = δ_p diag(a) Σ_m (k < m; row m is empty: δ_p[m,m] = 0):
          //   T[k+1]: ½a[k+1] (if k+1 < m: (Σ_m T)[m] = ½T[m-1] has no T[m]),  T[k]: ½(a[k+1] - a[k]),  T[k-1]: -½a[k]
          double cp = 0.0, cm = 0.0;
          const double* ca = P.conv_a[q][d];
          if (ca && idx[d] < c.ext[d] - 1) {
            const double a0 = ca[lc], a1 = ca[lc + st];
            if (idx[d] + 1 < c.ext[d] - 1) cp = 0.5 * a1;
            cm = -0.5 * a0;
            dC += 0.5 * (a1 - a0);
          }
          if (mv) {       // column scaling Ψ (no convection in the moving solver)
            if (L.has_p) {
              emit(kw, lc + st, scale * (L.gl_p * L.w_p * L.gd_p) * P.mv_psi_w[q][lc + st]);
              emit(kg, lc + st, scale * (L.gl_p * L.w_p * L.hd_p) * (P.mv_psi_g[q] ? P.mv_psi_g[q][lc + st] : P.mv_gconst));
            }
            if (L.has_m) {
              emit(kw, lc - st, scale * (L.gd_j * L.w_j * L.gl_j) * P.mv_psi_w[q][lc - st]);
              emit(kg, lc - st, scale * (L.gd_j * L.w_j * L.hl_j) * (P.mv_psi_g[q] ? P.mv_psi_g[q][lc - st] : P.mv_gconst));
            }
            continue;
          }
          if (L.has_p) {
            emit(kw, lc + st, scale * (L.gl_p * L.w_p * L.gd_p) + P.theta * cp);
            emit(kg, lc + st, scale * (L.gl_p * L.w_p * L.hd_p));
          }
          if (L.has_m) {
            emit(kw, lc - st, scale * (L.gd_j * L.w_j * L.gl_j) + P.theta * cm);
            emit(kg, lc - st, scale * (L.gd_j * L.w_j * L.hl_j));
          }
        } else {
          // HᵀWꜝG and HᵀWꜝH rows
          dW += L.hd_j * L.w_j * L.gd_j + L.hl_p * L.w_p * L.gl_p;
          dG += L.hd_j * L.hd_j * L.w_j + L.hl_p * L.hl_p * L.w_p;
          if (mv && P.nphase == 2) {      // moving flux row: columns scaled by Ψ_q (:378-381)
            const double* pw = P.mv_psi_w[q];
            if (L.has_p) {
              emit(kw, lc + st, scale * (L.hl_p * L.w_p * L.gd_p) * pw[lc + st]);
              emit(kg, lc + st, scale * (L.hl_p * L.w_p * L.hd_p) * pw[lc + st]);
            }
            if (L.has_m) {
              emit(kw, lc - st, scale * (L.hd_j * L.w_j * L.gl_j) * pw[lc - st]);
              emit(kg, lc - st, scale * (L.hd_j * L.w_j * L.hl_j) * pw[lc - st]);
            }
            continue;
          }
          if (L.has_p) {
            emit(kw, lc + st, scale * (L.hl_p * L.w_p * L.gd_p));
            emit(kg, lc + st, scale * (L.hl_p * L.w_p * L.hd_p));
          }
          if (L.has_m) {
            emit(kw, lc - st, scale * (L.hd_j * L.w_j * L.gl_j));
            emit(kg, lc - st, scale * (L.hd_j * L.w_j * L.hl_j));
          }
        }
      }
    }
    if (bulk && mv) {
      const double pw = P.mv_psi_w[q][lc], pg = P.mv_psi_g[q] ? P.mv_psi_g[q][lc] : P.mv_gconst;
      const double v0 = P.mv_explicit ? 0.0 : P.mv_v0[q][lc], v1 = P.mv_explicit ? 0.0 : P.mv_v1[q][lc];
      emit(kw, lc, v0 + scale * dW * pw);               // Vn_1 + (Id GᵀWꜝG Ψ)_jj
      emit(kg, lc, -(v0 - v1) + scale * dG * pg);       // -(Vn_1 - Vn) + (Id GᵀWꜝH Ψ)_jj
    } else if (bulk) {
      const double ck = P.conv_k[q] ? P.conv_k[q][lc] : 0.0;          // 0.5 sum(K): on both diagonals (A11 and A12)
      emit(kw, lc, P.mass * c.V[lc] + scale * dW + P.theta * (dC + ck));   // V + θ(Id·GᵀWꜝG + ΣC + ½ΣK)_jj  (steady: no V)
      emit(kg, lc, scale * dG + P.theta * ck);
    } else if (P.nphase == 1) {
      emit(kw, lc, scale * dW);
      emit(kg, lc, scale * dG + P.gscale * (P.Ia * c.G[lc]));   // + (Δt/2)·Iₐ·Γ
    } else if (mv) {                    // moving flux row: β_q (HᵀWꜝG Ψ)_jj, β_q (HᵀWꜝH Ψ)_jj - (Vn_1 - Vn)   (:378-381)
      const double pw = P.mv_psi_w[q][lc];
      emit(kw, lc, scale * dW * pw);
      emit(kg, lc, scale * dG * pw - (P.mv_v0[q][lc] - P.mv_v1[q][lc]));
    } else {
      emit(kw, lc, scale * dW);
      emit(kg, lc, scale * dG);
    }
  }
}

}  // namespace pg
