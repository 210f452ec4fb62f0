// pg_precond.hip -- assembly of the PRECONDITIONED reduced system the Krylov solver works on:
//
//        Â = B⁻¹ (S A S),     b̂ = B⁻¹ S b,     x = S y
//
//   S = diag(|a_ii|^-1/2)          symmetric point equilibration (cut cells give rows 10^5 apart in scale)
//   B = per-CELL diagonal blocks of S A S over the unknowns that live in one cell (ω,γ monophasic; ω¹,γ¹,ω²,γ²
//       diphasic), identity for cells with a single active unknown or an overwritten border row.
//
// Why: the reference solves these systems with a direct solver (`\`, UMFPACK).  A Krylov method sees the strong
// LOCAL coupling between a cell's bulk and interface unknowns (Robin / Neumann interface rows, the diphasic jump
// and flux rows form a small saddle point per cut cell): measured on the oracle, BiCGStab on S A S needs
// 192 (Robin 20^3), 599 (Neumann 20^3), 661 / >3000 (diphasic 32^2 / 64^2) iterations and fails at 128^2, whereas
// with the cell blocks inverted it needs 13 / 26 / 20 / 28 -- and 14 instead of 18 on the Dirichlet benchmark
// problem.  Same linear system, same solution; everything is folded into the CSR values at assembly, so the SpMV
// is unchanged (rows of cut cells carry the union pattern of their cell: +3 % nnz at 512^3).
#include "pg_scan.h"
#include "pg_system.h"

using namespace pg;

namespace {

struct RowSegs {
  int K;
  i64 off_own[MAX_KINDS];
};

__device__ inline int row_kind_of(const RowSegs& s, i64 r) {
  int k = 0;
  while (k + 1 < s.K && r >= s.off_own[k + 1]) ++k;
  return k;
}

constexpr int NSLOT = 7;   // self + 2N neighbours (N <= 3)

// neighbour slot of column cell `cl` seen from cell `lc`: 0 self, 1+2d = +e_d, 2+2d = -e_d, -1 foreign
__device__ inline int slot_of(const CapView& c, i64 lc, i64 cl) {
  if (cl == lc) return 0;
  for (int d = 0; d < c.N; ++d) {
    if (cl == lc + c.stride[d]) return 1 + 2 * d;
    if (cl == lc - c.stride[d]) return 2 + 2 * d;
  }
  return -1;
}

__device__ inline i64 cell_of_slot(const CapView& c, i64 lc, int slot) {
  if (slot == 0) return lc;
  const int d = (slot - 1) >> 1;
  return (slot & 1) ? lc + c.stride[d] : lc - c.stride[d];
}

struct CellBlock {
  int nk;                               // active unknowns of the cell
  int kinds[MAX_KINDS];
  int rows[MAX_KINDS];                  // their reduced (local vector) indices
  double dsk[MAX_KINDS];                // their point scales
  bool blocked;                         // B != I for this cell
  double binv[MAX_KINDS][MAX_KINDS];    // inverse of the scaled diagonal block, indexed by position in kinds[]
  double bfw[MAX_KINDS][MAX_KINDS];     // the scaled diagonal block itself
};

__device__ inline void cell_block(const SysParams& P, int K, i64 Mloc, i64 lc, const i64* idx, const int* __restrict__ red,
                                  const double* __restrict__ ds, CellBlock& cb) {
  cb.nk = 0;
  for (int k = 0; k < K; ++k) {
    const int r = red[(i64)k * Mloc + lc];
    if (r >= 0) {
      cb.kinds[cb.nk] = k;
      cb.rows[cb.nk] = r;
      cb.dsk[cb.nk] = ds[r];
      ++cb.nk;
    }
  }
  cb.blocked = cb.nk >= 2;
  if (!cb.blocked) return;
  // cells whose bulk row is overwritten by a border condition keep B = I (their rows may reference foreign columns)
  for (int a = 0; a < cb.nk; ++a) {
    const int k = cb.kinds[a];
    if ((k & 1) == 0) {
      int key;
      if (border_row_kind(P, k >> 1, lc, idx, &key) != PG_BC_NONE) { cb.blocked = false; return; }
    }
  }
  double m[MAX_KINDS][MAX_KINDS];
  for (int a = 0; a < MAX_KINDS; ++a)
    for (int b = 0; b < MAX_KINDS; ++b) { m[a][b] = 0.0; cb.binv[a][b] = a == b ? 1.0 : 0.0; }
  for (int a = 0; a < cb.nk; ++a) {
    const int k = cb.kinds[a];
    eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
      if (cl != lc) return;
      for (int b = 0; b < cb.nk; ++b)
        if (cb.kinds[b] == ck) m[a][b] += cb.dsk[a] * v * cb.dsk[b];
    });
  }
  for (int a = 0; a < MAX_KINDS; ++a)
    for (int b = 0; b < MAX_KINDS; ++b) cb.bfw[a][b] = m[a][b];
  // Gauss-Jordan with partial pivoting on the nk x nk block
  const int n = cb.nk;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    double best = fabs(m[c][c]);
    for (int r = c + 1; r < n; ++r)
      if (fabs(m[r][c]) > best) { best = fabs(m[r][c]); piv = r; }
    if (!(best > 1e-300)) { cb.blocked = false; return; }   // singular block: leave the cell un-blocked
    if (piv != c)
      for (int q = 0; q < n; ++q) {
        double t = m[c][q]; m[c][q] = m[piv][q]; m[piv][q] = t;
        t = cb.binv[c][q]; cb.binv[c][q] = cb.binv[piv][q]; cb.binv[piv][q] = t;
      }
    const double inv = 1.0 / m[c][c];
    for (int q = 0; q < n; ++q) { m[c][q] *= inv; cb.binv[c][q] *= inv; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = m[r][c];
      if (f == 0.0) continue;
      for (int q = 0; q < n; ++q) { m[r][q] -= f * m[c][q]; cb.binv[r][q] -= f * cb.binv[c][q]; }
    }
  }
}

// count (FILL = false) or write (FILL = true) the preconditioned row of every owned unknown
template <bool FILL>
__global__ void k_asm_p(SysParams P, RowSegs seg, i64 Mloc, i64 n_own, const int* __restrict__ row_cell,
                        const int* __restrict__ red, const double* __restrict__ ds, int* __restrict__ cnt,
                        unsigned char* __restrict__ isblk, unsigned long long* __restrict__ nnz_raw,
                        const int* __restrict__ rowptr, int* __restrict__ col, double* __restrict__ val) {
  const CapView& c = P.cap[0];
  const int K = seg.K;
  unsigned long long raw = 0;
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_own; r += (i64)gridDim.x * blockDim.x) {
    const int k = row_kind_of(seg, r);
    const i64 lc = row_cell[r];
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    CellBlock cb;
    cell_block(P, K, Mloc, lc, idx, red, ds, cb);
    int n = 0;
    int at = FILL ? rowptr[r] : 0;
    if (!cb.blocked) {
      const double sr = ds[r];
      eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
        if (v == 0.0) return;
        const int cc = red[(i64)ck * Mloc + cl];
        if (cc < 0) return;
        if (FILL) { col[at] = cc; val[at] = sr * v * ds[cc]; ++at; }
        ++n;
      });
      raw += n;
    } else {
      int me = 0;
      for (int a = 0; a < cb.nk; ++a)
        if (cb.kinds[a] == k) me = a;
      if (!FILL)
        eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
          if (v != 0.0 && red[(i64)ck * Mloc + cl] >= 0) ++raw;
        });
      double acc[MAX_KINDS][NSLOT];
      for (int a = 0; a < MAX_KINDS; ++a)
        for (int s = 0; s < NSLOT; ++s) acc[a][s] = 0.0;
      for (int a = 0; a < cb.nk; ++a) {
        const double coef = cb.binv[me][a] * cb.dsk[a];
        if (coef == 0.0) continue;
        eval_row(P, cb.kinds[a], lc, idx, [&](int ck, i64 cl, double v) {
          if (v == 0.0) return;
          const int s = slot_of(c, lc, cl);
          if (s < 0) return;   // cannot happen in a blocked cell (no border rows)
          acc[ck][s] += coef * v;
        });
      }
      for (int ck = 0; ck < K; ++ck)
        for (int s = 0; s < NSLOT; ++s) {
          const double v = acc[ck][s];
          if (v == 0.0) continue;
          const i64 cl = cell_of_slot(c, lc, s);
          const int cc = red[(i64)ck * Mloc + cl];
          if (cc < 0) continue;
          if (FILL) { col[at] = cc; val[at] = v * ds[cc]; ++at; }
          ++n;
        }
    }
    if (!FILL) {
      cnt[r] = n;
      isblk[r] = cb.blocked ? 1 : 0;
    }
  }
  if (!FILL && raw) atomicAdd(nnz_raw, raw);
}

__global__ void k_blk_table(SysParams P, RowSegs seg, i64 Mloc, i64 n_own, const int* __restrict__ row_cell,
                            const int* __restrict__ red, const double* __restrict__ ds,
                            const unsigned char* __restrict__ isblk, const int* __restrict__ pos,
                            int* __restrict__ blk_rows, int* __restrict__ blk_idx, double* __restrict__ blk_coef,
                            double* __restrict__ blk_cn, double* __restrict__ blk_fw) {
  const CapView& c = P.cap[0];
  const int K = seg.K;
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_own; r += (i64)gridDim.x * blockDim.x) {
    if (!isblk[r]) continue;
    const int k = row_kind_of(seg, r);
    const i64 lc = row_cell[r];
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    CellBlock cb;
    cell_block(P, K, Mloc, lc, idx, red, ds, cb);
    int me = 0;
    for (int a = 0; a < cb.nk; ++a)
      if (cb.kinds[a] == k) me = a;
    const int q = pos[r];
    blk_rows[q] = (int)r;
    for (int a = 0; a < MAX_KINDS; ++a) {
      blk_idx[q * MAX_KINDS + a] = a < cb.nk ? cb.rows[a] : -1;
      blk_coef[q * MAX_KINDS + a] = a < cb.nk ? cb.binv[me][a] * cb.dsk[a] : 0.0;
      blk_fw[q * MAX_KINDS + a] = a < cb.nk ? cb.bfw[me][a] : 0.0;
      // (B⁻¹ M B)[me][a], M = 1 on the rows whose Crank-Nicolson right-hand side subtracts A x (all rows of a
      // monophasic system, the bulk rows of a diphasic one: diffusion.jl:257-258, 409-416)
      // Monophasic: M = I, so B⁻¹MB is the identity EXACTLY (the computed product B⁻¹·B is only I + O(eps cond B), and that
      // error would multiply ŷ).  Diphasic: the table is kept for inspection only -- with M != I the product has entries of
      // the size of cond(B) and b̂ = (B⁻¹S)c - (B⁻¹MB)ŷ cancels catastrophically in cells with a tiny cut volume (3-D 16^3:
      // right-hand side wrong by 1e-8, T by 6e-7); the diphasic Crank-Nicolson right-hand side of the block rows is
      // evaluated matrix-free instead (k_rhs_block_mf, pg_solver.hip).
      double cn = 0.0;
      if (a < cb.nk) {
        if (P.nphase == 1) cn = a == me ? 1.0 : 0.0;
        else
          for (int j = 0; j < cb.nk; ++j)
            if ((cb.kinds[j] & 1) == 0) cn += cb.binv[me][j] * cb.bfw[j][a];
      }
      blk_cn[q * MAX_KINDS + a] = cn;
    }
  }
}

// number of stored entries that reference a ghost column
__global__ void k_count_ghost_refs(i64 nnz, i64 n_own, const int* __restrict__ col, unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (i64 k = blockIdx.x * (i64)blockDim.x + threadIdx.x; k < nnz; k += (i64)gridDim.x * blockDim.x) c += col[k] >= n_own ? 1 : 0;
  if (c) atomicAdd(out, c);
}

// largest Gershgorin radius |1 - â_ii| + Σ_{j != i} |â_ij| ds_i / ds_j over the rows with more than one entry, the
// columns j that are identity rows (one entry, their own) left out; ghost columns count (their rows are not here)
__global__ void k_gershgorin(i64 n, const int* __restrict__ rowptr, const int* __restrict__ col,
                             const double* __restrict__ val, const double* __restrict__ ds, double* __restrict__ out) {
  double worst = 0.0;
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    const int a = rowptr[r], b = rowptr[r + 1];
    if (b - a <= 1) {
      // identity-like row: eigenvalue = its diagonal (1 after equilibration)
      const double d = b > a && col[a] == r ? fabs(1.0 - val[a]) : 1.0;
      worst = d > worst ? d : worst;
      continue;
    }
    double diag = 0.0, off = 0.0;
    for (int k = a; k < b; ++k) {
      const int c = col[k];
      if (c == r) { diag = val[k]; continue; }
      if (c < n && rowptr[c + 1] - rowptr[c] == 1 && col[rowptr[c]] == c) continue;   // column of an identity row
      off += fabs(val[k]) * (ds[r] / ds[c]);
    }
    const double rad = fabs(1.0 - diag) + off;
    worst = rad > worst ? rad : worst;
  }
  // max over the grid: values are >= 0, so the bit pattern orders like the number
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_down(worst, o, 64);
    worst = other > worst ? other : worst;
  }
  if ((threadIdx.x & 63) == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(worst));
}

__global__ void k_pl_simple(i64 n, const double* __restrict__ ds, const double* __restrict__ in, double* __restrict__ out) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) out[r] = ds[r] * in[r];
}

__global__ void k_pl_block(i64 nblk, const int* __restrict__ blk_rows, const int* __restrict__ blk_idx,
                           const double* __restrict__ blk_coef, const double* __restrict__ in, double* __restrict__ out) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < nblk; q += (i64)gridDim.x * blockDim.x) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < MAX_KINDS; ++a) {
      const int j = blk_idx[q * MAX_KINDS + a];
      if (j >= 0) s += blk_coef[q * MAX_KINDS + a] * in[j];
    }
    out[blk_rows[q]] = s;
  }
}

// ds[red] = |a_ii|^-1/2 for every numbered unknown (owned and ghost: ghost rows are evaluable locally)
__global__ void k_point_scale(SysParams P, int K, i64 Mloc, const int* red, double* ds) {
  const CapView& c = P.cap[0];
  const i64 total = (i64)K * Mloc;
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < total; q += (i64)gridDim.x * blockDim.x) {
    const int r = red[q];
    if (r < 0) continue;
    const int k = (int)(q / Mloc);
    const i64 lc = q % Mloc;
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    double diag = 0.0, rowmax = 0.0;
    eval_row(P, k, lc, idx, [&](int ck, i64 cl, double v) {
      if (ck == k && cl == lc) diag = v;
      rowmax = fmax(rowmax, fabs(v));
    });
    // A diagonal that is zero -- or zero but for rounding: 1e-33 where time-integrated face capacities of a full cell cancel
    // to 1e-17 instead of exactly (space-time slabs) -- gives no scale: such rows (flux rows of cells one phase is absent
    // from, coupled through the jump row) keep S = 1 and are handled by the cell block.  |a_ii|^-1/2 = 2e16 there made the
    // preconditioned system numerically singular.
    const double a = fabs(diag);
    ds[r] = (a > 1e-13 * rowmax && a < 1e300) ? 1.0 / sqrt(a) : 1.0;
  }
}

}  // namespace

namespace pg {

// grid position of the rows for the order of the marching units (3-D only: pg_host_algos.h plan_march_units)
static void set_geo(CsrMatrix& A, const Slab& slab, const Numbering& nb) {
  A.geo_cell = nb.row_cell.p;
  A.geo_map = nullptr;
  A.geo_ext0 = slab.N == 3 ? slab.ext[0] : 0;
  A.geo_lines = slab.N == 3 ? slab.ext[1] : 0;
}

void assemble_csr_preconditioned(const SysParams& P, const Slab& s, const Numbering& nb, CsrMatrix& A,
                                 const bool* inherit_halo) {
  hipStream_t st = ctx().stream;
  Laps laps;
  const i64 n = nb.n_own;
  A.n = n;
  A.rowptr.alloc(n + 1);
  A.ds.alloc(nb.n_vec() > 0 ? nb.n_vec() : 1);
  A.n_blk = 0;
  A.halo_needed = true;
  RowSegs seg;
  seg.K = nb.K;
  for (int k = 0; k < MAX_KINDS; ++k) seg.off_own[k] = nb.off_own[k];
  if (n == 0) {
    A.rowptr.zero();
    A.nnz = 0;
    if (inherit_halo) A.halo_needed = *inherit_halo;
    else if (ctx().nranks > 1) {   // take part in the collective decision below
      DevBuf<unsigned long long> nref(1);
      nref.zero();
      comm_allreduce_sum_u64(nref.p, 1, st);
      unsigned long long h = 0;
      nref.download(&h, 1);
      A.halo_needed = h != 0;
    }
    decide_poly(A);
    return;
  }
  hipLaunchKernelGGL(k_point_scale, dim3(grid_for((i64)nb.K * nb.Mloc, 256, 256 * 16)), dim3(256), 0, st, P, nb.K, nb.Mloc,
                     nb.red.p, A.ds.p);
  PG_HIP(hipGetLastError());
  DevBuf<int> cnt(n);
  A.isblk.alloc(n);
  DevBuf<unsigned char>& isblk = A.isblk;
  DevBuf<unsigned long long> nraw(1);
  nraw.zero();
  const int gr = grid_for(n, 256, 256 * 16);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_asm_p<false>), dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p,
                     A.ds.p, cnt.p, isblk.p, nraw.p, (const int*)nullptr, (int*)nullptr, (double*)nullptr);
  PG_HIP(hipGetLastError());
  scan_exclusive<int>(cnt.p, A.rowptr.p, n, A.rowptr.p + n, st);
  int nnz = 0;
  A.rowptr.download(&nnz, 1, n);
  PG_REQUIRE(nnz >= 0, "nnz overflow");
  A.nnz = nnz;
  unsigned long long hraw = 0;
  nraw.download(&hraw, 1);
  A.nnz_raw = (i64)hraw;
  A.col.alloc(nnz + 8);   // +8: the SpMV streams 16-byte-aligned pairs/quads and may touch a few entries past nnz
  A.val.alloc(nnz + 8);
  A.col.zero();
  A.val.zero();
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_asm_p<true>), dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p,
                     A.ds.p, (int*)nullptr, (unsigned char*)nullptr, (unsigned long long*)nullptr, (const int*)A.rowptr.p, A.col.p, A.val.p);
  PG_HIP(hipGetLastError());
  // compact table of the rows whose cell has B != I (right-hand sides need B⁻¹S applied too)
  DevBuf<int> pos(n), total(1);
  scan_exclusive<unsigned char>(isblk.p, pos.p, n, total.p, st);
  int nblk = 0;
  total.download(&nblk, 1);
  A.n_blk = nblk;
  A.blk_rows.alloc(nblk > 0 ? nblk : 1);
  A.blk_idx.alloc((i64)MAX_KINDS * (nblk > 0 ? nblk : 1));
  A.blk_coef.alloc((i64)MAX_KINDS * (nblk > 0 ? nblk : 1));
  A.blk_cn.alloc((i64)MAX_KINDS * (nblk > 0 ? nblk : 1));
  A.blk_fw.alloc((i64)MAX_KINDS * (nblk > 0 ? nblk : 1));
  if (nblk > 0) {
    hipLaunchKernelGGL(k_blk_table, dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p, A.ds.p, isblk.p,
                       pos.p, A.blk_rows.p, A.blk_idx.p, A.blk_coef.p, A.blk_cn.p, A.blk_fw.p);
    PG_HIP(hipGetLastError());
  }
  PG_HIP(hipStreamSynchronize(st));
  if (inherit_halo) A.halo_needed = *inherit_halo;
  else if (ctx().nranks > 1) {
    DevBuf<unsigned long long> nref(1);
    nref.zero();
    hipLaunchKernelGGL(k_count_ghost_refs, dim3(grid_for(nnz, 256, 4096)), dim3(256), 0, st, (i64)nnz, n, A.col.p, nref.p);
    PG_HIP(hipGetLastError());
    comm_allreduce_sum_u64(nref.p, 1, st);
    unsigned long long h = 0;
    nref.download(&h, 1);
    A.halo_needed = h != 0;
    if (config().debug)
      fprintf(stderr, "[pg_precond] rank %d: %llu ghost-column references over all ranks => halo %s\n", ctx().rank, h,
              A.halo_needed ? "exchanged" : "skipped");
  }
  laps.lap("  asm: count/fill/table kernels");
  set_geo(A, s, nb);
  build_spmv_chunks(A);
  laps.lap("  asm: SpMV chunks + slices");
  decide_poly(A);
}

__global__ void k_structure_differs(i64 n, const int* __restrict__ cnt, const int* __restrict__ rowptr,
                                    const unsigned char* __restrict__ blk_a, const unsigned char* __restrict__ blk_b,
                                    unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x)
    c += (cnt[r] != rowptr[r + 1] - rowptr[r] || blk_a[r] != blk_b[r]) ? 1 : 0;
  if (c) atomicAdd(out, c);
}

void assemble_csr_like(const SysParams& P, const Slab& s, const Numbering& nb, const CsrMatrix& T, CsrMatrix& A) {
  hipStream_t st = ctx().stream;
  Laps laps;
  const i64 n = nb.n_own;
  const bool halo = T.halo_needed;
  if (n == 0 || T.n != n || !T.rowptr.p) {
    assemble_csr_preconditioned(P, s, nb, A, &halo);
    return;
  }
  RowSegs seg;
  seg.K = nb.K;
  for (int k = 0; k < MAX_KINDS; ++k) seg.off_own[k] = nb.off_own[k];
  A.n = n;
  A.ds.alloc(nb.n_vec() > 0 ? nb.n_vec() : 1);
  hipLaunchKernelGGL(k_point_scale, dim3(grid_for((i64)nb.K * nb.Mloc, 256, 256 * 16)), dim3(256), 0, st, P, nb.K, nb.Mloc,
                     nb.red.p, A.ds.p);
  PG_HIP(hipGetLastError());
  DevBuf<int> cnt(n);
  A.isblk.alloc(n);
  DevBuf<unsigned long long> nraw(2);
  nraw.zero();
  const int gr = grid_for(n, 256, 256 * 16);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_asm_p<false>), dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p,
                     A.ds.p, cnt.p, A.isblk.p, nraw.p, (const int*)nullptr, (int*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL(k_structure_differs, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, n, cnt.p, T.rowptr.p, A.isblk.p,
                     T.isblk.p, nraw.p + 1);
  PG_HIP(hipGetLastError());
  unsigned long long h[2] = {0, 0};
  nraw.download(h, 2);
  if (h[1] != 0) {   // a coefficient vanishes under one scheme only: different pattern
    assemble_csr_preconditioned(P, s, nb, A, &halo);
    return;
  }
  A.nnz = T.nnz;
  A.nnz_raw = (i64)h[0];
  A.halo_needed = halo;
  A.rowptr.alloc(n + 1);
  PG_HIP(hipMemcpyAsync(A.rowptr.p, T.rowptr.p, sizeof(int) * (n + 1), hipMemcpyDeviceToDevice, st));
  A.col.alloc(A.nnz + 8);
  A.val.alloc(A.nnz + 8);
  A.col.zero();
  A.val.zero();
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_asm_p<true>), dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p,
                     A.ds.p, (int*)nullptr, (unsigned char*)nullptr, (unsigned long long*)nullptr, (const int*)A.rowptr.p, A.col.p, A.val.p);
  PG_HIP(hipGetLastError());
  const i64 nblk = T.n_blk, nb1 = nblk > 0 ? nblk : 1;
  A.n_blk = nblk;
  A.blk_rows.alloc(nb1);
  A.blk_idx.alloc((i64)MAX_KINDS * nb1);
  A.blk_coef.alloc((i64)MAX_KINDS * nb1);
  A.blk_cn.alloc((i64)MAX_KINDS * nb1);
  A.blk_fw.alloc((i64)MAX_KINDS * nb1);
  if (nblk > 0) {
    DevBuf<int> pos(n), total(1);
    scan_exclusive<unsigned char>(A.isblk.p, pos.p, n, total.p, st);
    hipLaunchKernelGGL(k_blk_table, dim3(gr), dim3(256), 0, st, P, seg, nb.Mloc, n, nb.row_cell.p, nb.red.p, A.ds.p, A.isblk.p,
                       pos.p, A.blk_rows.p, A.blk_idx.p, A.blk_coef.p, A.blk_cn.p, A.blk_fw.p);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(st));
  }
  laps.lap("  asm-like: values on the ctor matrix' pattern");
  set_geo(A, s, nb);
  if (!build_slices_like(T, A)) build_spmv_chunks(A);
  laps.lap("  asm-like: slices");
  decide_poly(A);
}

void decide_poly(CsrMatrix& A) {
  hipStream_t st = ctx().stream;
  DevBuf<double> worst(1);
  worst.zero();
  if (A.n > 0) {
    hipLaunchKernelGGL(k_gershgorin, dim3(grid_for(A.n, 256, 4096)), dim3(256), 0, st, A.n, A.rowptr.p, A.col.p, A.val.p, A.ds.p,
                       worst.p);
    PG_HIP(hipGetLastError());
  }
  comm_allreduce_max_f64(worst.p, 1, st);
  double h = 0.0;
  worst.download(&h, 1);
  A.gersh = h;
  A.poly_ok = h < 0.95;
  if (config().debug && !(h < 1e300) && A.n > 0) {   // developer aid: the first rows whose disc is not finite
    std::vector<int> rp(A.n + 1), cl(A.nnz);
    std::vector<double> vl(A.nnz), dsv(A.ds.n);
    A.rowptr.download(rp.data(), A.n + 1); A.col.download(cl.data(), A.nnz); A.val.download(vl.data(), A.nnz);
    A.ds.download(dsv.data(), A.ds.n);
    int shown = 0;
    for (i64 r = 0; r < A.n && shown < 4; ++r) {
      bool bad = !(dsv[r] > 0.0) || !(dsv[r] < 1e300);
      for (int k = rp[r]; k < rp[r + 1]; ++k) bad = bad || !(std::fabs(vl[k]) < 1e300);
      if (!bad) continue;
      ++shown;
      fprintf(stderr, "[pg_precond] row %lld: ds %.3e entries", (long long)r, dsv[r]);
      for (int k = rp[r]; k < rp[r + 1]; ++k) fprintf(stderr, " (%d: %.3e, ds %.3e)", cl[k], vl[k], dsv[cl[k]]);
      fprintf(stderr, "\n");
    }
  }
  if (config().debug)
    fprintf(stderr, "[pg_precond] rank %d: largest Gershgorin radius %.4f => Neumann preconditioner %s\n", ctx().rank, h,
            A.poly_ok ? "admissible" : "not used");
}

// out = B⁻¹ S in   (in and out must not alias)
void apply_left(const CsrMatrix& A, const double* in, double* out, hipStream_t st) {
  if (A.n == 0) return;
  hipLaunchKernelGGL(k_pl_simple, dim3(grid_for(A.n, 256)), dim3(256), 0, st, A.n, A.ds.p, in, out);
  if (A.n_blk > 0)
    hipLaunchKernelGGL(k_pl_block, dim3(grid_for(A.n_blk, 256)), dim3(256), 0, st, A.n_blk, A.blk_rows.p, A.blk_idx.p,
                       A.blk_coef.p, in, out);
  PG_HIP(hipGetLastError());
}

}  // namespace pg
