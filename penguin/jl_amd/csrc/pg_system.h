// pg_system.h -- reduced (active-set) linear system of one slab: numbering (K10) + CSR matrix (K7/K9).
#pragma once
#include "pg_stencil.h"

namespace pg {

// Local vector layout (length n_own + n_ghost):
//   [ kind0 owned | kind1 owned | ... | lower ghost: kind0.. | upper ghost: kind0.. ]
// Owned unknowns of a kind are ordered by padded cell index (the reference's common_idx order,
// src/solver.jl:65-71), so the actives of the first / last owned plane are contiguous chunks at the
// start / end of each kind segment: halo exchange needs no pack/unpack kernels.
struct Numbering {
  int K = 2;
  i64 Mloc = 0;
  i64 n_own = 0, n_ghost = 0;
  i64 cnt_own[MAX_KINDS] = {0, 0, 0, 0}, off_own[MAX_KINDS] = {0, 0, 0, 0};
  i64 cntL[MAX_KINDS] = {0, 0, 0, 0}, offL[MAX_KINDS] = {0, 0, 0, 0};   // lower ghost segments (absolute offsets)
  i64 cntU[MAX_KINDS] = {0, 0, 0, 0}, offU[MAX_KINDS] = {0, 0, 0, 0};   // upper ghost segments
  // chunks of OWNED unknowns sent to the lower / upper neighbour (absolute offsets into the owned part)
  i64 sendL_off[MAX_KINDS] = {0, 0, 0, 0}, sendL_cnt[MAX_KINDS] = {0, 0, 0, 0};
  i64 sendU_off[MAX_KINDS] = {0, 0, 0, 0}, sendU_cnt[MAX_KINDS] = {0, 0, 0, 0};
  // rows that touch no ghost column: [interior_lo, interior_hi) per kind is NOT contiguous across kinds,
  // so the overlap split is expressed per kind
  DevBuf<int> red;        // K*Mloc: local vector index of (kind, local cell) or -1
  DevBuf<int> row_cell;   // n_own: local cell of each owned row
  i64 n_vec() const { return n_own + n_ghost; }
  int kind_of_row(i64 r) const {
    int k = 0;
    while (k + 1 < K && r >= off_own[k + 1]) ++k;
    return k;
  }
};

// The stored matrix is the symmetrically equilibrated  Â = S A S,  S = diag(|a_ii|^-1/2)  (S = 1 where
// a_ii = 0).  The reference solves A x = b with a direct solver (or an unpreconditioned Krylov method that
// needs O(10^3) iterations on cut-cell systems: tiny staggered volumes give rows 10^5 apart in scale).
// Solving  Â y = S b,  x = S y  is the same linear system; folding S into the CSR values costs nothing per
// SpMV and brings BiCGStab to ~20 iterations per step.  `ds` holds S for owned AND ghost unknowns.
struct CsrMatrix {
  i64 n = 0, nnz = 0;
  int scheme = -1;
  DevBuf<int> rowptr, col;
  DevBuf<double> val;
  DevBuf<double> ds;   // n_vec
  // SpMV row blocks: chunk c = {first row, first entry} = chunk_desc[2c, 2c+1]; <= 64 rows, <= SPMV_CHUNK_ENTRIES
  // entries; closed by {n, nnz}
  DevBuf<int> chunk_desc;
  i64 nchunks = 0;
  // left cell-block preconditioner (pg_precond.hip): rows of cells with B != I, the local indices of the cell's
  // unknowns (MAX_KINDS per row, -1 padded) and the coefficient rows (B⁻¹S)[row, idx], (B⁻¹MB)[row, idx]
  // (Crank-Nicolson right-hand side, see k_blk_table) and B[row, idx] (export of the un-preconditioned b)
  // stencil-sliced image of the same matrix, the default SpMV path (pg_spmv.hip "Stencil slices"):
  //   U slice: <= 128 consecutive rows with identical (col - row) offsets AND bitwise identical values
  //   P slice: <= 128 consecutive rows with identical offsets, per-row values stored slot-major in pval
  //   G chunk: <= 64 of the remaining (irregular) rows, packed contiguously into their own CSR (g_*)
  // one 128-byte record per slice: {row0 | first packed row, rows | type<<8 | cnt<<16, base, end} + 8 offsets + 8 values
  DevBuf<int> srec;
  DevBuf<double> pval;
  DevBuf<int> g_rowid, g_rowptr, g_col;
  DevBuf<double> g_val;
  // marching units (pg_spmv.hip "marching units"): one 256-byte record per <= 11 planes x 126 rows of uniform 2-D / 3-D
  // stencil rows; these rows are in no slice
  DevBuf<int> mrec;
  i64 nunits = 0, rows_m = 0;
  // tile table of the slice kernel (pg_host_algos.h plan_tiles): 8 x (tiles_per_xcd + 1) x {first unit, first slice}; covers
  // the units and the slices [0, tile_ns) = [0, nslices_int)
  DevBuf<int> tiles;
  int tiles_per_xcd = 0;
  i64 tile_ns = 0;
  // grid position of the rows, for the order of the units only (non-owning, read while the units are planned): local cell
  // of row geo_map[r] (or r), cells per grid line, lines per plane
  const int* geo_cell = nullptr;
  const int* geo_map = nullptr;
  i64 geo_ext0 = 0, geo_lines = 0;
  bool want_units = true;   // false: a system solved once (a slab of the moving solver): planning the units costs more than they save
  i64 nslices = 0;
  i64 nslices_int = 0;   // slices [0, nslices_int) reference no ghost column (computable before the halo of x has landed)
  DevBuf<unsigned char> rowflags;   // k_row_same flags the slices were cut from (structure reuse, assemble_csr_like)
  i64 rows_u = 0, rows_p = 0, rows_g = 0, nnz_p = 0, nnz_g = 0;
  i64 spmv_bytes = 0;        // bytes one launch has to move with this format: records + P/G streams + 16 n (x, y)
  // false when NO rank's rows reference a ghost column (e.g. one body per slab with fluid away from the slab faces):
  // the per-SpMV halo exchange is then skipped on every rank (decided collectively at assembly)
  bool halo_needed = true;
  // true when every eigenvalue of Â provably lies in the disc |λ - 1| < 0.95 (Gershgorin on S Â S⁻¹ = D⁻¹A-like rows,
  // columns of identity rows left out: the matrix is block triangular in them): the truncated Neumann series
  // M⁻¹ = 2I - Â = Â⁻¹(I - (I - Â)²) is then a safe right preconditioner for BiCGStab (pg_krylov.hip).  Collective.
  bool poly_ok = false;
  double gersh = 0.0;   // the largest disc radius found (all ranks)
  i64 n_blk = 0;
  i64 nnz_raw = 0;   // entries of the un-preconditioned reduced matrix (what pg_solver_get_system_csr(0/1) returns)
  DevBuf<int> blk_rows, blk_idx;
  DevBuf<double> blk_coef, blk_cn, blk_fw;
  DevBuf<unsigned char> isblk;   // n: 1 for the rows listed in blk_rows
};
constexpr int SPMV_CHUNK_ENTRIES = 508;   // + alignment shift (<= 3) fits the 512-slot LDS slice of a wave
void build_spmv_chunks(CsrMatrix& A);   // pg_spmv.hip

void build_numbering(const SysParams& P, const Slab& slab, Numbering& nb);
// raw reduced matrix A (scale = false, for export) or the point-equilibrated S A S (scale = true)
void assemble_csr(const SysParams& P, const Slab& slab, const Numbering& nb, CsrMatrix& A, bool scale = true);
// the matrix the Krylov solver iterates on:  Â = B⁻¹ S A S  (pg_precond.hip)
// inherit_halo != nullptr: take halo_needed from there instead of deciding it collectively (no all-reduce)
void assemble_csr_preconditioned(const SysParams& P, const Slab& slab, const Numbering& nb, CsrMatrix& A,
                                 const bool* inherit_halo = nullptr);
// Same operator with other coefficients (BE -> CN: another θ): if the new matrix has T's sparsity pattern, blocked rows
// and row-repetition flags -- it always has, unless a coefficient vanishes under one scheme only -- reuse T's structure
// (row pointers, slices, packed irregular rows) and only refill values: ~10 ms instead of ~70 ms at 512^3.
// Falls back to the full assembly otherwise.  No collectives either way.
void assemble_csr_like(const SysParams& P, const Slab& slab, const Numbering& nb, const CsrMatrix& T, CsrMatrix& A);
// (pg_spmv.hip) slices of A from T's when the new values still support every one of them; false: build from scratch
bool build_slices_like(const CsrMatrix& T, CsrMatrix& A);
// sets A.poly_ok / A.gersh (collective: every rank calls it once per assembled matrix, rowless ranks included)
void decide_poly(CsrMatrix& A);
// out = B⁻¹ S in over the owned rows (in != out)
void apply_left(const CsrMatrix& A, const double* in, double* out, hipStream_t st);
// y (padded, K*Mloc) = K_full * x (padded, K*Mloc): matrix-free application of the un-reduced operator rows
// `rows` of the planes owned by this rank (used for the constructor's first right-hand side under CN).
void apply_rows_padded(const SysParams& P, const Slab& slab, const double* x, double* y);

}  // namespace pg
