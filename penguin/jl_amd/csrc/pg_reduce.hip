// pg_reduce.hip -- the Krylov iteration without the interface unknowns of a Dirichlet problem.
//
// With a Dirichlet interface condition (Iᵦ = 0, src/solver.jl:203-223) the γ rows of the block system are  Γ Tγ = Γ g
// (src/solver/diffusion.jl:229-232: block3 = 0, block4 = IₐΓ), and after the cell-block preconditioner they are rows of
// the identity (to rounding: s_j a_jj s_j):  Â = [Â_ωω Â_ωγ; 0 D], D ≈ I.  Their solution is the right-hand side itself.  The reference solves them with
// everything else; the time loop here fixes  x_γ = b̂_γ  before the iteration, moves the change through the coupling
// block into the residual,  r_ω ← r_ω − Â_ωγ (b̂_γ − x_γ),  and iterates on  Â_ωω  alone.  Same solution (the γ rows are
// solved exactly instead of to the tolerance); the SpMV loses the γ rows and -- what matters -- the γ COLUMNS of the
// cut cells' and their neighbours' rows, which are most of the bytes of the packed irregular rows (G chunks): many of
// those rows become plain stencil rows with their own values (P slices).
//
// Vectors: the reduced unknowns are the first n_ω entries of the full ones (kind 0 comes first in the numbering), so the
// reduced system works on prefixes of the solver's vectors -- no gather, no copy.  That needs a system without ghost
// columns in the ω rows: one rank, or slabs whose rows reference no ghost (A.halo_needed == false).
#include "pg_krylov.h"
#include "pg_scan.h"
#include "pg_spmv.h"
#include "pg_reduce.h"

namespace pg {
namespace {

using pg::BLOCK;

// are rows [n_w, n) diagonal rows  d_j x_j = b_j ?  (d_j = 1 up to the rounding of the equilibration s_j a_jj s_j)  count
// the offenders, keep the diagonal
__global__ void k_identity_rows(i64 n_w, i64 n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                const double* __restrict__ val, double* __restrict__ gdiag, unsigned long long* bad) {
  unsigned long long c = 0;
  for (i64 r = n_w + blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    const int a = rowptr[r], b = rowptr[r + 1];
    const bool ok = b - a == 1 && col[a] == (int)r && fabs(val[a] - 1.0) <= 1e-12;
    gdiag[r - n_w] = ok ? val[a] : 1.0;
    if (!ok) ++c;
  }
  if (c) atomicAdd(bad, c);
}

// per ω row: entries kept (col < n_w), entries of the coupling block (n_w <= col < n), ghost references (col >= n)
__global__ void k_split_count(i64 n_w, i64 n, const int* __restrict__ rowptr, const int* __restrict__ col, int* cnt_w,
                              int* has_g, int* cnt_g, unsigned long long* ghosts) {
  unsigned long long gh = 0;
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_w; r += (i64)gridDim.x * blockDim.x) {
    int cw = 0, cg = 0;
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int c = col[e];
      if (c < n_w) ++cw;
      else if (c < n) ++cg;
      else ++gh;
    }
    cnt_w[r] = cw;
    cnt_g[r] = cg;
    has_g[r] = cg > 0 ? 1 : 0;
  }
  if (gh) atomicAdd(ghosts, gh);
}

__global__ void k_split_fill(i64 n_w, const int* __restrict__ rowptr, const int* __restrict__ col,
                             const double* __restrict__ val, const int* __restrict__ rp_w, int* col_w, double* val_w,
                             const int* __restrict__ has_g, const int* __restrict__ pos_g, const int* __restrict__ cnt_g,
                             int* wg_rows, int* wg_cnt) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n_w; r += (i64)gridDim.x * blockDim.x) {
    int at = rp_w[r];
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e)
      if (col[e] < n_w) { col_w[at] = col[e]; val_w[at] = val[e]; ++at; }
    if (has_g[r]) {
      wg_rows[pos_g[r]] = (int)r;
      wg_cnt[pos_g[r]] = cnt_g[r];
    }
  }
}

__global__ void k_coupling_fill(i64 n_wg, i64 n_w, i64 n, const int* __restrict__ wg_rows, const int* __restrict__ wg_ptr,
                                const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                int* wg_col, double* wg_val) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < n_wg; q += (i64)gridDim.x * blockDim.x) {
    const int r = wg_rows[q];
    int at = wg_ptr[q];
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e)
      if (col[e] >= n_w && col[e] < n) { wg_col[at] = col[e] - (int)n_w; wg_val[at] = val[e]; ++at; }
  }
}

// γ rows  d_j x_j = b̂_j  with the residual r_j = b̂_j - d_j x_j at hand: δ = r_j / d_j, x_j += δ, r_j = 0; flag: some δ is
// more than rounding noise
__global__ void k_gamma_snap(i64 n_w, i64 n, double* __restrict__ x, double* __restrict__ r, double* __restrict__ rhat,
                             double* __restrict__ p, const double* __restrict__ gdiag, double* __restrict__ delta, int* flag) {
  int f = 0;
  for (i64 j = n_w + blockIdx.x * (i64)blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    const double d = r[j] / gdiag[j - n_w], xo = x[j];
    delta[j - n_w] = d;
    x[j] = xo + d;
    r[j] = 0.0; rhat[j] = 0.0; p[j] = 0.0;
    if (fabs(d) > 1e-12 * fabs(xo)) f = 1;
  }
  if (f) atomicOr(flag, 1);
}

// coupled ω rows: r -= Â_ωγ δ
__global__ void k_gamma_couple(i64 n_wg, const int* __restrict__ wg_rows, const int* __restrict__ wg_ptr,
                               const int* __restrict__ wg_col, const double* __restrict__ wg_val,
                               const double* __restrict__ delta, double* __restrict__ r, double* __restrict__ rhat,
                               double* __restrict__ p) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < n_wg; q += (i64)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int e = wg_ptr[q]; e < wg_ptr[q + 1]; ++e) s += wg_val[e] * delta[wg_col[e]];
    if (s != 0.0) {
      const int i = wg_rows[q];
      const double v = r[i] - s;
      r[i] = v; rhat[i] = v; p[i] = v;
    }
  }
}

// the start sums of k_rhs_init, slots 0 = (r,r) and 2 = (r,r)_W, again -- only if the residual really changed
// (flag == stamp: some row moved in this step; the Dirichlet-interface variant uses 0 / 1 and resets the flag per step, the
// compact variant stamps it with the step number -- k_rhs_init_c, pg_solver.hip -- and never resets it)
__global__ __launch_bounds__(BLOCK) void k_renorm(i64 n_w, const int* __restrict__ flag, int stamp, const double* __restrict__ r,
                                                  const double* __restrict__ ds, double* __restrict__ partials, int force) {
  __shared__ double s_red[BLOCK / 64];
  if (!force && *flag != stamp) return;
  double acc = 0.0, accw = 0.0;
  for (i64 i = blockIdx.x * (i64)BLOCK + threadIdx.x; i < n_w; i += (i64)gridDim.x * BLOCK) {
    const double v = r[i], d = ds[i];
    acc += v * v;
    accw += (d * v) * (d * v);
  }
  const double t = block_sum(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  const double tw = block_sum(accw, s_red);
  if (threadIdx.x == 0) partials[2 * (size_t)gridDim.x + blockIdx.x] = tw;
}


// ---- compact variant: every row that is alone on its diagonal ------------------------------------------------------
// Slabs that exchange a halo: the ghost entries of the vectors take part.  A ghost unknown is eliminated exactly when its
// OWNER eliminates it, so the owners' verdicts travel once through the ordinary halo exchange (as doubles: k_flags_f64 ->
// halo_exchange -> k_ghost_flags); both sides then derive the same compact ghost segments, and the send chunks -- the
// remaining rows of the first / last owned plane -- stay contiguous because the compact order is the full order restricted.
__global__ void k_diag_rows(i64 n, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                            int* __restrict__ is_e, int* __restrict__ is_r) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    const int a = rowptr[r], b = rowptr[r + 1];
    const int e = (b - a == 1 && col[a] == (int)r && fabs(val[a] - 1.0) <= 1e-12) ? 1 : 0;
    is_e[r] = e;
    is_r[r] = 1 - e;
  }
}

__global__ void k_flags_f64(i64 n, const int* __restrict__ is_e, double* __restrict__ f) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) f[r] = (double)is_e[r];
}

// ghost entries [n, nvec): exchanged != 0: f holds the owners' verdicts; else no row references a ghost column and the
// ghosts stay out of the compact vectors altogether (neither eliminated nor remaining)
__global__ void k_ghost_flags(i64 n, i64 nvec, const double* __restrict__ f, int exchanged, int* __restrict__ is_e,
                              int* __restrict__ is_r) {
  for (i64 r = n + blockIdx.x * (i64)blockDim.x + threadIdx.x; r < nvec; r += (i64)gridDim.x * blockDim.x) {
    const int e = exchanged ? (f[r] != 0.0 ? 1 : 0) : 0;
    is_e[r] = e;
    is_r[r] = exchanged ? 1 - e : 0;
  }
}

// cmap[r] >= 0: compact index (owned rows first, then the remaining ghosts); cmap[r] = -1 - q: eliminated, q = its entry of
// gdiag / delta (owned, q < n_e) or n_e + (r - n) (ghost: entry r - n of the exchanged ghost deltas)
__global__ void k_maps(i64 n, i64 nvec, i64 n_e, const int* __restrict__ is_e, const int* __restrict__ is_r,
                       const int* __restrict__ pos_e, const int* __restrict__ pos_r, const int* __restrict__ rowptr,
                       const double* __restrict__ val, const double* __restrict__ ds, int* cmap, int* rlist, int* elist,
                       double* gdiag, double* ds_c) {
  for (i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x; r < nvec; r += (i64)gridDim.x * blockDim.x) {
    if (r < n) {
      if (is_e[r]) {
        cmap[r] = -1 - pos_e[r];   // (k_rhs_init_c finds the row's entry of gdiag / delta through it)
        elist[pos_e[r]] = (int)r;
        gdiag[pos_e[r]] = val[rowptr[r]];
      } else {
        cmap[r] = pos_r[r];
        rlist[pos_r[r]] = (int)r;
        ds_c[pos_r[r]] = ds[r];
      }
    } else if (is_r[r]) {
      cmap[r] = pos_r[r];
      ds_c[pos_r[r]] = ds[r];
    } else {
      cmap[r] = -1 - (int)(n_e + (r - n));
    }
  }
}

__global__ void k_gather_int(int cnt, const int* __restrict__ idx, const int* __restrict__ src, int* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < cnt) out[q] = src[idx[q]];
}

// per remaining row: entries kept (the column remains: owned or ghost), entries of the coupling block (the column is
// eliminated, here or on the neighbour)
__global__ void k_c_count(i64 n_c, const int* __restrict__ rlist, const int* __restrict__ cmap, const int* __restrict__ rowptr,
                          const int* __restrict__ col, int* cnt_c, int* has_g, int* cnt_g) {
  for (i64 c = blockIdx.x * (i64)blockDim.x + threadIdx.x; c < n_c; c += (i64)gridDim.x * blockDim.x) {
    const int r = rlist[c];
    int cw = 0, cg = 0;
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      if (cmap[col[e]] >= 0) ++cw;
      else ++cg;
    }
    cnt_c[c] = cw; cnt_g[c] = cg; has_g[c] = cg > 0 ? 1 : 0;
  }
}

__global__ void k_c_fill(i64 n_c, const int* __restrict__ rlist, const int* __restrict__ cmap, const int* __restrict__ rowptr,
                         const int* __restrict__ col, const double* __restrict__ val, const int* __restrict__ rp_c, int* col_c,
                         double* val_c, const int* __restrict__ has_g, const int* __restrict__ pos_g, const int* __restrict__ cnt_g,
                         int* wg_rows, int* wg_cnt) {
  for (i64 c = blockIdx.x * (i64)blockDim.x + threadIdx.x; c < n_c; c += (i64)gridDim.x * blockDim.x) {
    const int r = rlist[c];
    int at = rp_c[c];
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int j = cmap[col[e]];
      if (j >= 0) { col_c[at] = j; val_c[at] = val[e]; ++at; }
    }
    if (has_g[c]) { wg_rows[pos_g[c]] = r; wg_cnt[pos_g[c]] = cnt_g[c]; }
  }
}

__global__ void k_c_coupling(i64 n_wg, const int* __restrict__ wg_rows, const int* __restrict__ wg_ptr, const int* __restrict__ cmap,
                             const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                             int* wg_col, double* wg_val) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < n_wg; q += (i64)gridDim.x * blockDim.x) {
    const int r = wg_rows[q];
    int at = wg_ptr[q];
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int j = cmap[col[e]];
      if (j < 0) { wg_col[at] = -1 - j; wg_val[at] = val[e]; ++at; }
    }
  }
}

// the owned deltas at their places in a full-layout vector (what the neighbours receive as ghost deltas)
__global__ void k_delta_scatter(i64 n_e, const int* __restrict__ elist, const double* __restrict__ delta, double* __restrict__ dx) {
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < n_e; q += (i64)gridDim.x * blockDim.x) dx[elist[q]] = delta[q];
}

// coupled remaining rows (compact index cmap[wg_rows[q]]): r -= Â_RE δ, r̂ and p alike.  δ of an owned column: delta[j],
// j < n_e; of a ghost column: dghost[j - n_e] (the neighbour's delta, exchanged).  force == 0: only when some row of THIS
// rank moved in this step (one rank); force != 0: the caller knows the data changed (several ranks: a neighbour's row may
// have moved)
__global__ void k_c_couple(i64 n_wg, const int* __restrict__ wg_rows, const int* __restrict__ wg_ptr, const int* __restrict__ wg_col,
                           const double* __restrict__ wg_val, const double* __restrict__ delta, i64 n_e,
                           const double* __restrict__ dghost, const int* __restrict__ cmap, double* __restrict__ rhat,
                           const int* __restrict__ flag, int stamp, int force) {
  if (!force && *flag != stamp) return;   // no diagonal row moved in this step (constant Dirichlet data after the first step)
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < n_wg; q += (i64)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int e = wg_ptr[q]; e < wg_ptr[q + 1]; ++e) {
      const int j = wg_col[e];
      s += wg_val[e] * (j < n_e ? delta[j] : dghost[j - n_e]);
    }
    if (s != 0.0) {
      const int c = cmap[wg_rows[q]];
      rhat[c] -= s;   // (r and p are not stored at the start: KrylovWork::p_in_rhat)
    }
  }
}

}  // namespace

void build_gamma_elim(const CsrMatrix& A, const Numbering& nb, GammaElim& E) {
  E.tried = true;
  E.active = false;
  Context& cx = ctx();
  hipStream_t st = cx.stream;
  if (!config().gamma_elim || nb.K != 2 || A.n <= 0 || !A.rowptr.p) return;
  const i64 n = A.n, n_w = nb.cnt_own[0];
  if (n_w <= 0 || n_w >= n) return;
  // several ranks that exchange a halo: ω rows reference ghosts, prefixes would not do -- and the decision has to be the same
  // on every rank (A.halo_needed is collective); one rank never has ghosts, with or without a communicator
  if (cx.nranks > 1 && A.halo_needed) return;
  DevBuf<unsigned long long> bad(2);
  bad.zero();
  E.gdiag.alloc(n - n_w);
  hipLaunchKernelGGL(k_identity_rows, dim3(grid_for(n - n_w, 256)), dim3(256), 0, st, n_w, n, A.rowptr.p, A.col.p, A.val.p, E.gdiag.p,
                     bad.p);
  DevBuf<int> cnt_w(n_w + 1), has_g(n_w + 1), cnt_g(n_w), pos_g(n_w + 1), tot(2);
  hipLaunchKernelGGL(k_split_count, dim3(grid_for(n_w, 256, 256 * 16)), dim3(256), 0, st, n_w, n, A.rowptr.p, A.col.p, cnt_w.p,
                     has_g.p, cnt_g.p, bad.p + 1);
  PG_HIP(hipGetLastError());
  unsigned long long hb[2];
  bad.download(hb, 2);
  if (hb[0] != 0 || hb[1] != 0) {                                // not a Dirichlet block structure / ghost references
    if (config().debug)
      fprintf(stderr, "[pg_reduce] no reduction: %llu of %lld interface rows are not rows of the identity, %llu ghost references\n",
              hb[0], (long long)(n - n_w), hb[1]);
    return;
  }
  CsrMatrix& R = E.A;
  R.n = n_w;
  R.scheme = A.scheme;
  R.rowptr.alloc(n_w + 1);
  scan_exclusive<int>(cnt_w.p, R.rowptr.p, n_w, R.rowptr.p + n_w, st);
  scan_exclusive<int>(has_g.p, pos_g.p, n_w, tot.p, st);
  int nnz = 0, n_wg = 0;
  R.rowptr.download(&nnz, 1, n_w);
  tot.download(&n_wg, 1);
  R.nnz = nnz;
  R.col.alloc(nnz + 8); R.val.alloc(nnz + 8);
  R.col.zero(); R.val.zero();
  E.n_wg = n_wg;
  E.wg_rows.alloc(n_wg + 1);
  DevBuf<int> wg_cnt(n_wg + 1);
  hipLaunchKernelGGL(k_split_fill, dim3(grid_for(n_w, 256, 256 * 16)), dim3(256), 0, st, n_w, A.rowptr.p, A.col.p, A.val.p,
                     R.rowptr.p, R.col.p, R.val.p, has_g.p, pos_g.p, cnt_g.p, E.wg_rows.p, wg_cnt.p);
  PG_HIP(hipGetLastError());
  E.wg_ptr.alloc(n_wg + 1);
  int nnz_g = 0;
  if (n_wg > 0) {
    scan_exclusive<int>(wg_cnt.p, E.wg_ptr.p, n_wg, E.wg_ptr.p + n_wg, st);
    E.wg_ptr.download(&nnz_g, 1, n_wg);
  } else {
    E.wg_ptr.zero();
  }
  E.wg_col.alloc(nnz_g + 1); E.wg_val.alloc(nnz_g + 1);
  if (n_wg > 0)
    hipLaunchKernelGGL(k_coupling_fill, dim3(grid_for(n_wg, 256)), dim3(256), 0, st, (i64)n_wg, n_w, n, E.wg_rows.p, E.wg_ptr.p,
                       A.rowptr.p, A.col.p, A.val.p, E.wg_col.p, E.wg_val.p);
  PG_HIP(hipGetLastError());
  R.ds.alloc(n_w);
  PG_HIP(hipMemcpyAsync(R.ds.p, A.ds.p, sizeof(double) * (size_t)n_w, hipMemcpyDeviceToDevice, st));
  PG_HIP(hipStreamSynchronize(st));
  R.halo_needed = false;
  R.poly_ok = A.poly_ok;        // Gershgorin already left the identity rows' columns out (decide_poly)
  R.gersh = A.gersh;
  R.nnz_raw = nnz;
  R.geo_cell = A.geo_cell; R.geo_map = nullptr; R.geo_ext0 = A.geo_ext0; R.geo_lines = A.geo_lines;   // rows [0, n_w) keep their numbers
  build_spmv_chunks(R);
  E.nb = Numbering();
  E.nb.K = 1;
  E.nb.Mloc = nb.Mloc;
  E.nb.n_own = n_w;
  E.nb.n_ghost = 0;
  E.nb.cnt_own[0] = n_w;
  E.n_w = n_w;
  E.n_g = n - n_w;
  E.delta.alloc(E.n_g);
  E.flag.alloc(1);
  E.active = true;
  if (config().debug)
    fprintf(stderr, "[pg_reduce] Dirichlet interface rows left out of the iteration: %lld of %lld rows, %lld of %lld entries; coupling block %d rows, %d entries; "
            "irregular rows %lld -> %lld\n", (long long)(n - n_w), (long long)n, (long long)(A.nnz - nnz), (long long)A.nnz, n_wg, nnz_g,
            (long long)A.rows_g, (long long)R.rows_g);
}

void gamma_fix(const GammaElim& E, double* x, double* r, double* rhat, double* p, double* partials, int grid, hipStream_t st) {
  PG_HIP(hipMemsetAsync(E.flag.p, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_gamma_snap, dim3(grid_for(E.n_g, 256)), dim3(256), 0, st, E.n_w, E.n_w + E.n_g, x, r, rhat, p, E.gdiag.p,
                     E.delta.p, E.flag.p);
  if (E.n_wg > 0)
    hipLaunchKernelGGL(k_gamma_couple, dim3(grid_for(E.n_wg, 256)), dim3(256), 0, st, E.n_wg, E.wg_rows.p, E.wg_ptr.p, E.wg_col.p,
                       E.wg_val.p, E.delta.p, r, rhat, p);
  hipLaunchKernelGGL(k_renorm, dim3(grid), dim3(BLOCK), 0, st, E.n_w, (const int*)E.flag.p, 1, (const double*)r, E.A.ds.p, partials, 0);
  PG_HIP(hipGetLastError());
}


void build_diag_elim(const CsrMatrix& A, const Numbering& nb, const Slab& slab, DiagElim& E) {
  E.tried = true;
  E.active = false;
  Context& cx = ctx();
  hipStream_t st = cx.stream;
  const bool multi = cx.nranks > 1 || cx.comm;
  if (!config().diag_elim || !A.poly_ok) return;   // (both the same on every rank: poly_ok is decided collectively)
  if (!multi && (A.n <= 0 || !A.rowptr.p)) return;
  const i64 n = A.n, nvec = nb.n_vec();
  const bool halo = multi && A.halo_needed;              // ghost columns are referenced somewhere: the ghosts take part
  DevBuf<int> is_e(nvec + 1), is_r(nvec + 1), pos_e(n + 1), pos_r(nvec + 2), tot(2);
  if (n > 0)
    hipLaunchKernelGGL(k_diag_rows, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, n, A.rowptr.p, A.col.p, A.val.p, is_e.p, is_r.p);
  scan_exclusive<int>(is_e.p, pos_e.p, n, tot.p, st);
  int ht0 = 0;
  tot.download(&ht0, 1);
  const i64 n_e = ht0, n_c = n - n_e;
  // worth a second matrix?  2 % of the rows -- of ALL ranks' rows: the decision (and the exchange below) is collective
  {
    unsigned long long h[4] = {(unsigned long long)n_e, (unsigned long long)n, n_c <= 0 ? 1ull : 0ull, (unsigned long long)A.spmv_bytes};
    if (multi) {
      DevBuf<unsigned long long> d(4);
      d.upload(h, 4);
      comm_allreduce_sum_u64(d.p, 4, st);
      d.download(h, 4);
    }
    E.bytes_per_rank = (double)h[3] / std::max(1, cx.nranks);   // (the same figure on every rank: pg_solver.hip's extrapolated start)
    if ((double)h[0] < config().diag_elim_frac * (double)h[1] || h[2] != 0) return;   // (a rank without remaining rows: every rank takes the full system)
  }
  // the neighbours' verdicts on my ghost entries
  DevBuf<double> fl(nvec > 0 ? nvec : 1);
  fl.zero();
  if (halo) {
    if (n > 0) hipLaunchKernelGGL(k_flags_f64, dim3(grid_for(n, 256)), dim3(256), 0, st, n, (const int*)is_e.p, fl.p);
    halo_exchange(nb, slab, fl.p, st);
  }
  if (nvec > n)
    hipLaunchKernelGGL(k_ghost_flags, dim3(grid_for(nvec - n, 256)), dim3(256), 0, st, n, nvec, (const double*)fl.p, halo ? 1 : 0, is_e.p,
                       is_r.p);
  scan_exclusive<int>(is_r.p, pos_r.p, nvec, pos_r.p + nvec, st);   // pos_r[nvec] = remaining entries, owned and ghost
  int hr = 0;
  pos_r.download(&hr, 1, nvec);
  const i64 nvec_c = hr;
  PG_REQUIRE(nvec_c >= n_c, "compact numbering: fewer entries than owned rows");
  E.n = n; E.n_c = n_c; E.n_e = n_e; E.halo = halo;
  E.cmap.alloc(nvec > 0 ? nvec : 1); E.rlist.alloc(n_c); E.elist.alloc(n_e > 0 ? n_e : 1); E.gdiag.alloc(n_e > 0 ? n_e : 1);
  E.delta.alloc(n_e > 0 ? n_e : 1); E.flag.alloc(1);
  E.flag.zero();   // (stamped with the step number when a diagonal row moves: k_rhs_init_c; never reset)
  E.delta.zero();
  if (halo) { E.dx.alloc(nvec); E.dx.zero(); }
  CsrMatrix& R = E.A;
  R.n = n_c;
  R.scheme = A.scheme;
  R.ds.alloc(nvec_c > 0 ? nvec_c : 1);
  if (nvec > 0)
    hipLaunchKernelGGL(k_maps, dim3(grid_for(nvec, 256, 256 * 16)), dim3(256), 0, st, n, nvec, n_e, (const int*)is_e.p, (const int*)is_r.p,
                       (const int*)pos_e.p, (const int*)pos_r.p, A.rowptr.p, A.val.p, A.ds.p, E.cmap.p, E.rlist.p, E.elist.p, E.gdiag.p, R.ds.p);
  // the compact numbering: every segment of the full one, restricted (pos_r at the segment ends)
  Numbering& cn = E.nb;
  cn = Numbering();
  cn.K = nb.K;
  cn.Mloc = nb.Mloc;
  {
    std::vector<int> idx;
    auto want = [&](i64 full) { idx.push_back((int)full); return (int)idx.size() - 1; };
    int q_own[MAX_KINDS][2], q_L[MAX_KINDS][2], q_U[MAX_KINDS][2], q_sL[MAX_KINDS][2], q_sU[MAX_KINDS][2];
    for (int k = 0; k < nb.K; ++k) {
      q_own[k][0] = want(nb.off_own[k]); q_own[k][1] = want(nb.off_own[k] + nb.cnt_own[k]);
      q_L[k][0] = want(nb.offL[k]); q_L[k][1] = want(nb.offL[k] + nb.cntL[k]);
      q_U[k][0] = want(nb.offU[k]); q_U[k][1] = want(nb.offU[k] + nb.cntU[k]);
      q_sL[k][0] = want(nb.sendL_off[k]); q_sL[k][1] = want(nb.sendL_off[k] + nb.sendL_cnt[k]);
      q_sU[k][0] = want(nb.sendU_off[k]); q_sU[k][1] = want(nb.sendU_off[k] + nb.sendU_cnt[k]);
    }
    const int cnt = (int)idx.size();
    for (int v : idx) PG_REQUIRE(v >= 0 && v <= nvec, "compact numbering: segment end outside the vector");
    DevBuf<int> d_idx(cnt), d_out(cnt);
    d_idx.upload(idx.data(), cnt);
    hipLaunchKernelGGL(k_gather_int, dim3((cnt + 63) / 64), dim3(64), 0, st, cnt, (const int*)d_idx.p, (const int*)pos_r.p, d_out.p);
    std::vector<int> P(cnt);
    d_out.download(P.data(), cnt);
    for (int k = 0; k < nb.K; ++k) {
      cn.off_own[k] = P[q_own[k][0]]; cn.cnt_own[k] = P[q_own[k][1]] - P[q_own[k][0]];
      cn.offL[k] = P[q_L[k][0]]; cn.cntL[k] = P[q_L[k][1]] - P[q_L[k][0]];
      cn.offU[k] = P[q_U[k][0]]; cn.cntU[k] = P[q_U[k][1]] - P[q_U[k][0]];
      cn.sendL_off[k] = P[q_sL[k][0]]; cn.sendL_cnt[k] = P[q_sL[k][1]] - P[q_sL[k][0]];
      cn.sendU_off[k] = P[q_sU[k][0]]; cn.sendU_cnt[k] = P[q_sU[k][1]] - P[q_sU[k][0]];
    }
    cn.n_own = n_c;
    cn.n_ghost = nvec_c - n_c;
  }
  DevBuf<int> cnt_c(n_c + 1), has_g(n_c + 1), cnt_g(n_c > 0 ? n_c : 1), pos_g(n_c + 1);
  hipLaunchKernelGGL(k_c_count, dim3(grid_for(n_c, 256, 256 * 16)), dim3(256), 0, st, n_c, E.rlist.p, E.cmap.p, A.rowptr.p, A.col.p,
                     cnt_c.p, has_g.p, cnt_g.p);
  PG_HIP(hipGetLastError());
  R.rowptr.alloc(n_c + 1);
  scan_exclusive<int>(cnt_c.p, R.rowptr.p, n_c, R.rowptr.p + n_c, st);
  scan_exclusive<int>(has_g.p, pos_g.p, n_c, tot.p, st);
  int nnz = 0, n_wg = 0;
  R.rowptr.download(&nnz, 1, n_c);
  tot.download(&n_wg, 1);
  R.nnz = nnz;
  R.col.alloc(nnz + 8); R.val.alloc(nnz + 8);
  R.col.zero(); R.val.zero();
  E.n_wg = n_wg;
  E.wg_rows.alloc(n_wg + 1);
  DevBuf<int> wg_cnt(n_wg + 1);
  hipLaunchKernelGGL(k_c_fill, dim3(grid_for(n_c, 256, 256 * 16)), dim3(256), 0, st, n_c, E.rlist.p, E.cmap.p, A.rowptr.p, A.col.p, A.val.p,
                     R.rowptr.p, R.col.p, R.val.p, has_g.p, pos_g.p, cnt_g.p, E.wg_rows.p, wg_cnt.p);
  PG_HIP(hipGetLastError());
  E.wg_ptr.alloc(n_wg + 1);
  int nnz_g = 0;
  if (n_wg > 0) {
    scan_exclusive<int>(wg_cnt.p, E.wg_ptr.p, n_wg, E.wg_ptr.p + n_wg, st);
    E.wg_ptr.download(&nnz_g, 1, n_wg);
  } else {
    E.wg_ptr.zero();
  }
  E.wg_col.alloc(nnz_g + 1); E.wg_val.alloc(nnz_g + 1);
  if (n_wg > 0)
    hipLaunchKernelGGL(k_c_coupling, dim3(grid_for(n_wg, 256)), dim3(256), 0, st, (i64)n_wg, E.wg_rows.p, E.wg_ptr.p, E.cmap.p,
                       A.rowptr.p, A.col.p, A.val.p, E.wg_col.p, E.wg_val.p);
  PG_HIP(hipGetLastError());
  PG_HIP(hipStreamSynchronize(st));
  R.halo_needed = halo;
  R.poly_ok = A.poly_ok;
  R.gersh = A.gersh;
  R.nnz_raw = nnz;
  R.geo_cell = A.geo_cell; R.geo_map = E.rlist.p; R.geo_ext0 = A.geo_ext0; R.geo_lines = A.geo_lines;
  build_spmv_chunks(R);
  E.active = true;
  if (config().debug)
    fprintf(stderr, "[pg_reduce] rank %d: rows alone on their diagonal left out of the iteration: %lld of %lld rows (compact vectors, "
            "%lld of %lld ghost entries remain), %lld of %lld entries; coupling block %d rows, %d entries; irregular rows %lld -> %lld\n",
            cx.rank, (long long)n_e, (long long)n, (long long)(nvec_c - n_c), (long long)(nvec - n), (long long)(A.nnz - nnz),
            (long long)A.nnz, n_wg, nnz_g, (long long)A.rows_g, (long long)R.rows_g);
}

// (the rows themselves were solved by k_rhs_init_c: x += δ, δ kept in E.delta, E.flag = stamp if any δ != 0)
// force: the data of the right-hand side changed since the last step, so rows may have moved on ANY rank: the owners'
// deltas travel to the neighbours (one halo exchange of a full-layout vector), coupling and start sums are redone
// unconditionally.  One rank never forces: its own flag says whether anything moved.
void diag_fix(const DiagElim& E, const Numbering& nb_full, const Slab& slab, int stamp, bool force, double* rhat, double* partials,
              int grid, hipStream_t st) {
  const double* dghost = nullptr;
  if (E.halo) {
    if (E.n_e > 0)
      hipLaunchKernelGGL(k_delta_scatter, dim3(grid_for(E.n_e, 256)), dim3(256), 0, st, E.n_e, (const int*)E.elist.p, (const double*)E.delta.p, E.dx.p);
    halo_exchange(nb_full, slab, E.dx.p, st);
    dghost = E.dx.p + E.n;
  }
  if (E.n_wg > 0)
    hipLaunchKernelGGL(k_c_couple, dim3(grid_for(E.n_wg, 256)), dim3(256), 0, st, E.n_wg, E.wg_rows.p, E.wg_ptr.p, E.wg_col.p, E.wg_val.p,
                       (const double*)E.delta.p, E.n_e, dghost, E.cmap.p, rhat, (const int*)E.flag.p, stamp, force ? 1 : 0);
  hipLaunchKernelGGL(k_renorm, dim3(grid), dim3(BLOCK), 0, st, E.n_c, (const int*)E.flag.p, stamp, (const double*)rhat, E.A.ds.p, partials,
                     force ? 1 : 0);
  PG_HIP(hipGetLastError());
}

}  // namespace pg
