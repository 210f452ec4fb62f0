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

}  // namespace

extern "C" {

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
