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

// The same idea for EVERY row that is alone on its diagonal -- the Dirichlet interface rows and the identity rows a border
// condition creates (1.05 M of the 10.3 M rows at 512^3: empty cells on the lateral faces).  Those rows are scattered through
// the numbering, so the loop's vectors are COMPACT: cmap[row] = index among the remaining rows (or -1), rlist the inverse.
struct DiagElim {
  long long snapped_version = -1;   // pg_solver's bconst version at the last step that ran this path (-1: none / z moved since)
  bool tried = false, active = false;
  CsrMatrix A;              // the remaining rows and columns, compact numbering, own slices / marching units
  // compact numbering: the full one restricted to the remaining unknowns -- [owned kinds | lower ghosts | upper ghosts], the
  // send chunks still contiguous; the halo exchange of the loop's vectors runs on it unchanged (pg_comm.hip)
  Numbering nb;
  bool halo = false;        // the slabs exchange a halo: ghost entries are part of the compact vectors
  i64 n = 0, n_c = 0, n_e = 0;
  DevBuf<int> cmap;         // n_vec of the full numbering: compact index, or -1 - q for an eliminated unknown (k_maps)
  DevBuf<int> rlist, elist;
  DevBuf<double> gdiag, delta;
  DevBuf<double> dx;        // halo only: full-layout vector the owned deltas are exchanged in (ghost deltas at [n, n_vec))
  DevBuf<int> flag;
  i64 n_wg = 0;             // coupling block: full row wg_rows[q], entries wg_ptr[q] .. , columns = position in elist (owned) or n_e + ghost number
  DevBuf<int> wg_rows, wg_ptr, wg_col;
  DevBuf<double> wg_val;
  // the extrapolated start of the time loop on this system (pg_solver.hip): switched on once a solve has used enough products
  // for the fit's launch to pay (bytes_per_rank: a product's streamed bytes, averaged over the ranks -- the same everywhere)
  double bytes_per_rank = 0.0, last_products = 0.0;
  bool guess_on = false;
};
// Collective when several ranks run (every rank calls it at the same point: all-reduced activation test, one halo exchange).
void build_diag_elim(const CsrMatrix& A, const Numbering& nb, const Slab& slab, DiagElim& E);
// r_full: the start residuals of the rows that are left out (k_rhs_init_c; entries of the other rows are not read), r / r̂ / p:
// the compact start residual.  x_E += r_E / d; the change goes through the coupling block into r, r̂, p; the start sums (slots
// 0 and 2 of partials) are recomputed when the change is more than rounding noise.
// force: redo coupling and start sums unconditionally, with the neighbours' deltas (several ranks, data changed)
void diag_fix(const DiagElim& E, const Numbering& nb_full, const Slab& slab, int stamp, bool force, double* rhat, double* partials,
              int grid, hipStream_t st);

// active only for a two-kind (monophasic) system whose γ rows are rows of the identity and whose ω rows reference no ghost
void build_gamma_elim(const CsrMatrix& A, const Numbering& nb, GammaElim& E);
// after k_rhs_init: x_γ = b̂_γ, r_γ = 0, r_ω -= Â_ωγ (b̂_γ - x_γ) (r̂ and p alike), start sums refreshed when needed
void gamma_fix(const GammaElim& E, double* x, double* r, double* rhat, double* p, double* partials, int grid, hipStream_t st);

}  // namespace pg
