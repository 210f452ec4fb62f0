// pg_capacity.h -- device-resident Capacity (src/capacity.jl:25-36) of one slab.
#pragma once
#include "pg_common.h"
#include "pg_geom.h"

struct pg_capacity {
  pg_mesh* mesh = nullptr;
  pg::Slab slab;
  int N = 0;
  bool from_body = false;
  bool has_cg = true;
  pggeom::BallSet body;
  // all arrays: slab.Mloc() doubles in the local stored layout (plane-major, dim-0 fastest)
  pg::DevBuf<double> V, G, ct;
  pg::DevBuf<double> A[3], B[3], W[3], Cw[3], Cg[3];
  double kernel_ms = 0.0;
  pg::i64 n_cut_local = 0;
  // space-time capacity of ONE time slab [t0, t1] (pg_spacetime.hip): V, A, B, W, Γ above are the time-integrated
  // measures of the first time layer of the reference's (N+1)-D capacity; Vt = A_(N+1) at the two time faces
  // (V(t0), V(t1)), Ctw / Ctg = the time components of the space-time centroids C_ω / C_γ
  bool spacetime = false;
  double t0 = 0.0, t1 = 0.0;
  pg::DevBuf<double> Vt[2], Ctw, Ctg;
};

struct pg_diffops {
  pg_capacity* cap = nullptr;
  // ConvectionOps (src/operators.jl:194-210), set by pg_diffops_set_velocity; local (stored planes) arrays
  bool has_velocity = false;
  pg::DevBuf<double> conv_a[3];   // a_d = Σ_m[d] (A_d ∘ uω_d): C_d = δ_p[d] diag(a_d) Σ_m[d]
  pg::DevBuf<double> conv_h;      // Hᵀ uγ
  pg::DevBuf<double> conv_k;      // ½ Σ_d Σ_p[d] Hᵀuγ = the diagonal of 0.5 * sum(K)
};

namespace pg {
// pointers handed to kernels
struct CapView {
  int N;
  i64 ext[3], n[3], stride[3];
  i64 plane, s0, s1, nplanes;   // stored planes [s0,s1) of the slowest dim
  const double* V;
  const double* G;
  const double* A[3];
  const double* B[3];
  const double* W[3];
};
CapView cap_view(const pg_capacity* c);

// local cell -> 0-based Cartesian index (slowest dim from the plane number)
__host__ __device__ inline void decode_cell(int N, const i64* ext, i64 plane, i64 s0, i64 lc, i64* idx) {
  const i64 p = lc / plane + s0;
  i64 rem = lc % plane;
  idx[0] = idx[1] = idx[2] = 0;
  for (int d = 0; d < N - 1; ++d) {
    idx[d] = rem % ext[d];
    rem /= ext[d];
  }
  idx[N - 1] = p;
}
}  // namespace pg
