// pg_diffops.hip -- API-only pieces of DiffusionOps (src/operators.jl:20-34,127-178): ∇, ∇₋ as stencil kernels
// and an on-demand CSC export of G, H, Wꜝ so that the Julia wrapper can populate operator.G/H/Wꜝ.
// (The time loop never forms these matrices: see pg_stencil.h.)
#include "pg_stencil.h"

using namespace pg;

namespace {

// ∇(op, p) = Wꜝ (G pω + H pγ)                                            operators.jl:20-23
__global__ void k_grad(CapView c, i64 M, const double* p, double* out) {
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < M; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    for (int d = 0; d < c.N; ++d) {
      const Line L = load_line(c, d, lc, idx[d]);
      const i64 st = c.stride[d];
      double s = L.gd_j * p[lc] + L.hd_j * p[M + lc];
      if (L.has_m) s += L.gl_j * p[lc - st] + L.hl_j * p[M + lc - st];
      out[(i64)d * M + lc] = L.w_j * s;
    }
  }
}

// ∇₋(op, qω, qγ) = −(Gᵀ + Hᵀ) qω + Hᵀ qγ                                  operators.jl:30-34
__global__ void k_div(CapView c, i64 M, const double* qw, const double* qg, double* out) {
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < M; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    double s = 0.0;
    for (int d = 0; d < c.N; ++d) {
      const Line L = load_line(c, d, lc, idx[d]);
      const i64 st = c.stride[d];
      const i64 o = (i64)d * M;
      double a = (L.gd_j + L.hd_j) * qw[o + lc];
      double b = L.hd_j * qg[o + lc];
      if (L.has_p) {
        a += (L.gl_p + L.hl_p) * qw[o + lc + st];
        b += L.hl_p * qg[o + lc + st];
      }
      s += -a + b;
    }
    out[lc] = s;
  }
}

// ---- ConvectionOps (src/operators.jl:194-210) -------------------------------------------------------------------
// a_d = Σ_m[d] (A_d ∘ uω_d):  a[k] = ½((Au)[k] + (Au)[k-1]); first row ½(Au)[0]; last row (diagonal zeroed, :12) ½(Au)[m-1]
__global__ void k_conv_a(CapView c, i64 Mloc, int d, const double* __restrict__ u, double* __restrict__ a) {
  const i64 st = c.stride[d], m = c.ext[d] - 1;
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    const i64 k = idx[d];
    double v = 0.0;
    if (k < m) v += c.A[d][lc] * u[lc];
    if (k >= 1 && lc - st >= 0) v += c.A[d][lc - st] * u[lc - st];
    a[lc] = 0.5 * v;
  }
}

// h = Hᵀ uγ = Σ_d H_dᵀ uγ_d:  (H_dᵀ y)[i] = hd_i y_i + hl_{i+1} y_{i+1},  hd_k = A_k - B_k (k < m),  hl_k = -(A_k - B_{k-1})
__global__ void k_conv_h(CapView c, i64 Mloc, const double* __restrict__ ug /*N x Mloc*/, double* __restrict__ h) {
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    double s = 0.0;
    for (int d = 0; d < c.N; ++d) {
      const i64 st = c.stride[d], m = c.ext[d] - 1, k = idx[d];
      const double* y = ug + (i64)d * Mloc;
      if (k < m) s += (c.A[d][lc] - c.B[d][lc]) * y[lc];
      if (k + 1 <= m && lc + st < Mloc) s += -(c.A[d][lc + st] - c.B[d][lc]) * y[lc + st];
    }
    h[lc] = s;
  }
}

// diagonal of 0.5 * sum_d K_d,  K_d = diag(Σ_p[d] h):  (Σ_p h)[k] = ½(h[k] + h[k+1]) for k < m, 0 in the last row
__global__ void k_conv_kappa(CapView c, i64 Mloc, const double* __restrict__ h, double* __restrict__ kap) {
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(c.N, c.ext, c.plane, c.s0, lc, idx);
    double s = 0.0;
    for (int d = 0; d < c.N; ++d) {
      const i64 st = c.stride[d], m = c.ext[d] - 1;
      if (idx[d] < m && lc + st < Mloc) s += 0.5 * (h[lc] + h[lc + st]);
    }
    kap[lc] = 0.5 * s;
  }
}

}  // namespace

extern "C" {

int32_t pg_diffops_set_velocity(pg_diffops* o, const double* const* u_omega, const double* u_gamma) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(o && u_omega && u_gamma, "pg_diffops_set_velocity: NULL argument");
  pg_capacity* c = o->cap;
  const Slab& s = c->slab;
  const i64 Ml = s.Mloc(), M = s.M;
  hipStream_t st = ctx().stream;
  const CapView cv = cap_view(c);
  const int gr = grid_for(Ml, 256, 256 * 16);
  DevBuf<double> u(Ml), ug((i64)c->N * Ml);
  for (int d = 0; d < c->N; ++d) {
    PG_REQUIRE(u_omega[d], "pg_diffops_set_velocity: NULL velocity component");
    u.upload(u_omega[d] + s.first_cell(), Ml);
    o->conv_a[d].alloc(Ml);
    hipLaunchKernelGGL(k_conv_a, dim3(gr), dim3(256), 0, st, cv, Ml, d, u.p, o->conv_a[d].p);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(st));
    ug.upload(u_gamma + (i64)d * M + s.first_cell(), Ml, (i64)d * Ml);
  }
  o->conv_h.alloc(Ml);
  o->conv_k.alloc(Ml);
  hipLaunchKernelGGL(k_conv_h, dim3(gr), dim3(256), 0, st, cv, Ml, ug.p, o->conv_h.p);
  hipLaunchKernelGGL(k_conv_kappa, dim3(gr), dim3(256), 0, st, cv, Ml, o->conv_h.p, o->conv_k.p);
  PG_HIP(hipGetLastError());
  PG_HIP(hipStreamSynchronize(st));
  o->has_velocity = true;
  PG_API_END
}

int32_t pg_diffops_grad(const pg_diffops* o, const double* p, double* out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(o && p && out, "pg_diffops_grad: NULL argument");
  PG_REQUIRE(ctx().nranks == 1, "pg_diffops_grad: single rank only");
  const pg_capacity* c = o->cap;
  const i64 M = c->slab.M;
  DevBuf<double> dp(2 * M), dout((i64)c->N * M);
  dp.upload(p, 2 * M);
  hipLaunchKernelGGL(k_grad, dim3(grid_for(M, 256)), dim3(256), 0, ctx().stream, cap_view(c), M, dp.p, dout.p);
  PG_HIP(hipGetLastError());
  dout.download(out, (i64)c->N * M);
  PG_API_END
}

int32_t pg_diffops_div(const pg_diffops* o, const double* qw, const double* qg, double* out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(o && qw && qg && out, "pg_diffops_div: NULL argument");
  PG_REQUIRE(ctx().nranks == 1, "pg_diffops_div: single rank only");
  const pg_capacity* c = o->cap;
  const i64 M = c->slab.M, NM = (i64)c->N * M;
  DevBuf<double> dw(NM), dg(NM), dout(M);
  dw.upload(qw, NM);
  dg.upload(qg, NM);
  hipLaunchKernelGGL(k_div, dim3(grid_for(M, 256)), dim3(256), 0, ctx().stream, cap_view(c), M, dw.p, dg.p, dout.p);
  PG_HIP(hipGetLastError());
  dout.download(out, M);
  PG_API_END
}

int32_t pg_diffops_export_csc(const pg_diffops* o, int32_t which, int64_t* colptr, int64_t* rowval, double* nzval,
                              int64_t* nnz) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(o && nnz, "pg_diffops_export_csc: NULL argument");
  PG_REQUIRE(ctx().nranks == 1, "pg_diffops_export_csc: single rank only");
  const pg_capacity* c = o->cap;
  const Slab& s = c->slab;
  const int N = c->N;
  const i64 M = s.M;
  if (which == PG_OP_WINV) {
    *nnz = (i64)N * M;
    if (!nzval) return 0;
    std::vector<double> w(M);
    for (int d = 0; d < N; ++d) {
      c->W[d].download(w.data(), M);
      for (i64 i = 0; i < M; ++i) {
        const i64 q = (i64)d * M + i;
        colptr[q] = q;
        rowval[q] = q;
        nzval[q] = w[i] != 0.0 ? 1.0 / w[i] : 1.0;     // operators.jl:149-151
      }
    }
    colptr[(i64)N * M] = (i64)N * M;
    return 0;
  }
  if (which >= PG_OP_C0 && which < PG_OP_C0 + 3) {
    // C_d = δ_p diag(a_d) Σ_m  (M x M, tridiagonal along dimension d; rows k = m are empty)
    const int d = which - PG_OP_C0;
    PG_REQUIRE(o->has_velocity && d < N, "pg_diffops_export_csc: no velocity set (ConvectionOps) or bad dimension");
    std::vector<double> a(M);
    o->conv_a[d].download(a.data(), M);
    const i64 m = s.ext[d] - 1, st = s.stride[d];
    // column j = (.., k, ..): rows k-1 (coef ½a[k] if k < m... as the T[k+1]-coefficient of row k-1), k, k+1
    i64 at = 0;
    std::vector<i64> cp(M + 1);
    std::vector<i64> rv;
    std::vector<double> nv;
    for (i64 j = 0; j < M; ++j) {
      cp[j] = at;
      i64 idx[3];
      decode_cell(N, s.ext, s.plane, 0, j, idx);
      const i64 k = idx[d];
      // row k-1 (exists if k >= 1; k-1 < m always): coefficient of T[(k-1)+1] = ½a[k] if k < m
      if (k >= 1 && k < m) { rv.push_back(j - st); nv.push_back(0.5 * a[j]); ++at; }
      // row k (k < m): ½(a[k+1] - a[k])
      if (k < m) { rv.push_back(j); nv.push_back(0.5 * (a[j + st] - a[j])); ++at; }
      // row k+1 (k+1 < m): coefficient of T[(k+1)-1] = -½a[k+1]
      if (k + 1 < m) { rv.push_back(j + st); nv.push_back(-0.5 * a[j + st]); ++at; }
    }
    cp[M] = at;
    *nnz = at;
    if (!nzval) return 0;
    for (i64 j = 0; j <= M; ++j) colptr[j] = cp[j];
    for (i64 q = 0; q < at; ++q) { rowval[q] = rv[q]; nzval[q] = nv[q]; }
    return 0;
  }
  if (which >= PG_OP_K0 && which < PG_OP_K0 + 3) {
    // K_d = diag(Σ_p[d] Hᵀuγ)
    const int d = which - PG_OP_K0;
    PG_REQUIRE(o->has_velocity && d < N, "pg_diffops_export_csc: no velocity set (ConvectionOps) or bad dimension");
    *nnz = M;
    if (!nzval) return 0;
    std::vector<double> h(M);
    o->conv_h.download(h.data(), M);
    const i64 m = s.ext[d] - 1, st = s.stride[d];
    for (i64 j = 0; j < M; ++j) {
      i64 idx[3];
      decode_cell(N, s.ext, s.plane, 0, j, idx);
      colptr[j] = j;
      rowval[j] = j;
      nzval[j] = idx[d] < m ? 0.5 * (h[j] + h[j + st]) : 0.0;
    }
    colptr[M] = M;
    return 0;
  }
  PG_REQUIRE(which == PG_OP_G || which == PG_OP_H, "pg_diffops_export_csc: unknown operator");
  // pattern of D⁻_d (diagonal + sub-diagonal along dimension d), stacked over d
  i64 count = 0;
  for (int d = 0; d < N; ++d) count += M + (M - M / s.ext[d]);
  *nnz = count;
  if (!nzval) return 0;
  std::vector<std::vector<double>> A(N, std::vector<double>(M)), B(N, std::vector<double>(M));
  for (int d = 0; d < N; ++d) {
    c->A[d].download(A[d].data(), M);
    c->B[d].download(B[d].data(), M);
  }
  i64 at = 0;
  for (i64 j = 0; j < M; ++j) {
    colptr[j] = at;
    i64 idx[3];
    decode_cell(N, s.ext, s.plane, 0, j, idx);
    for (int d = 0; d < N; ++d) {
      const i64 m = s.ext[d] - 1, st = s.stride[d];
      const bool last = idx[d] == m;
      // diagonal entry (d, j): D⁻[j,j] = 1 (0 at the last index)
      rowval[at] = (i64)d * M + j;
      if (which == PG_OP_G) nzval[at] = last ? 0.0 : B[d][j];
      else nzval[at] = last ? 0.0 : (A[d][j] - B[d][j]);
      ++at;
      if (!last) {   // sub-diagonal entry in row j+st: D⁻[j+st, j] = −1
        rowval[at] = (i64)d * M + j + st;
        if (which == PG_OP_G) nzval[at] = -B[d][j];
        else nzval[at] = -(A[d][j + st] - B[d][j]);
        ++at;
      }
    }
  }
  colptr[M] = at;
  PG_REQUIRE(at == count, "internal: CSC export count mismatch");
  PG_API_END
}

}  // extern "C"
