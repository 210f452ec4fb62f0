// pg_reduce.h -- Krylov iteration without the interface unknowns of a Dirichlet problem (pg_reduce.hip)
#pragma once
#include "pg_system.h"

namespace pg {

struct GammaElim {
  bool tried = false, active = false;
  CsrMatrix A;            // Â_ωω with its own slices / marching units; vectors are prefixes of the full ones
  Numbering nb;           // one kind, no ghosts
  i64 n_w = 0, n_g = 0;
  // coupling block Â_ωγ, rows that have one only: wg_rows[q] is the ω row, entries wg_ptr[q] .. wg_ptr[q+1], columns
  // relative to the γ block
  i64 n_wg = 0;
  DevBuf<int> wg_rows, wg_ptr, wg_col;
  DevBuf<double> wg_val;
  DevBuf<double> gdiag;   // the diagonal of the γ rows (1 to rounding)
  DevBuf<double> delta;   // the change of x_γ in the current step
  DevBuf<int> flag;       // 1: delta is more than rounding noise (the start sums are recomputed)
};

// active only for a two-kind (monophasic) system whose γ rows are rows of the identity and whose ω rows reference no ghost
void build_gamma_elim(const CsrMatrix& A, const Numbering& nb, GammaElim& E);
// after k_rhs_init: x_γ = b̂_γ, r_γ = 0, r_ω -= Â_ωγ (b̂_γ - x_γ) (r̂ and p alike), start sums refreshed when needed
void gamma_fix(const GammaElim& E, double* x, double* r, double* rhat, double* p, double* partials, int grid, hipStream_t st);

}  // namespace pg
