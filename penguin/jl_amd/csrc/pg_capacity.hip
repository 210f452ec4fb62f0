// pg_capacity.hip -- K1-K5 of SURVEY.md section 2.3: per-cell cut-cell capacities on the GPU.
//
//   reference                                             here
//   CartesianGeometry.integrate(Tuple{0},...)  capacity.jl:90-92   k_classify + k_cut_cells  (V, C_w, Gamma, type)
//   computeInterfaceCentroids                  capacity.jl:137-197 k_cut_cells               (C_gamma)
//   integrate(Tuple{1},...)                    capacity.jl:103     k_sections                (A_d)
//   integrate(Tuple{1},...,bary)               capacity.jl:105     k_sections                (B_d)
//   integrate(Tuple{0},...,bary)               capacity.jl:104     k_stagger + k_stagger_cut (W_d)
//
// Layout: one thread per padded cell, dim-0 fastest => every store is a coalesced 8-byte-per-lane
// stream.  Cells whose geometry needs quadrature (cut cells, ~n^(N-1) of n^N) are appended to a
// compacted work list and processed by a dense second kernel so that no wave idles on a lone cut
// cell.  Compiled with -ffp-contract=off: classification must match the oracle bit for bit.
#include "pg_capacity.h"

#include <algorithm>
#include <memory>

using namespace pg;
using namespace pggeom;

__constant__ GLTable c_gl;

namespace {

struct GeoView {
  int N;
  i64 ext[3], n[3], stride[3];
  i64 plane, s0, s1;
  const double* nodes[3];
  double h[3];     // mesh spacing L_d / n_d, the number Mesh() builds the nodes from (src/mesh.jl:49-50)
};

// Measures of FULL cells / faces / staggered volumes come from the mesh spacing h, not from differences of node
// coordinates: x0 + (j+½)h carries rounding noise, so nodes[i+1] - nodes[i] varies in the last bits from cell to cell
// unless h is a binary fraction.  With h every full cell of a uniform mesh has bitwise the same capacities, hence the
// same matrix rows, and the SpMV's uniform slices work on ANY mesh Mesh() can build (768^3 over [0,4]^3: h = 1/192).
// The difference to the node-difference value is ~1e-16 relative -- the noise level of the reference's own VOFI sums.
__device__ inline double full_measure(const GeoView& g, int skip) {
  double p = 0.0;
  bool first = true;
  for (int d = 0; d < g.N; ++d) {
    if (d == skip) continue;
    p = first ? g.h[d] : p * g.h[d];
    first = false;
  }
  return first ? 1.0 : p;
}

__device__ inline bool is_real_cell(const GeoView& g, const i64* idx) {
  for (int d = 0; d < g.N; ++d)
    if (idx[d] >= g.n[d]) return false;
  return true;
}

__device__ inline void cell_box(const GeoView& g, const i64* idx, double* lo, double* hi) {
  for (int d = 0; d < g.N; ++d) {
    lo[d] = g.nodes[d][idx[d]];
    hi[d] = g.nodes[d][idx[d] + 1];
  }
}

// K1 (cheap part): classify every stored cell; full/empty cells are finished here.
__global__ void k_classify(GeoView g, BallSet bs, i64 Mloc, double* V, double* G, double* ct, double* Cw0,
                           double* Cw1, double* Cw2, double* Cg0, double* Cg1, double* Cg2, int* cut_list,
                           int* cut_count) {
  double* Cw[3] = {Cw0, Cw1, Cw2};
  double* Cg[3] = {Cg0, Cg1, Cg2};
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
    double v = 0.0, t = 0.0;
    double cw[3] = {0.0, 0.0, 0.0};
    if (is_real_cell(g, idx)) {
      double lo[3], hi[3];
      cell_box(g, idx, lo, hi);
      int type;
      pick_ball(bs, lo, hi, type);
      if (type != PG_CUT && bs.complement) type = 1 - type;
      for (int d = 0; d < g.N; ++d) cw[d] = 0.5 * (lo[d] + hi[d]);
      if (type == PG_FULL) v = full_measure(g, -1);
      if (type == PG_CUT) {
        const int slot = atomicAdd(cut_count, 1);
        cut_list[slot] = (int)lc;
      }
      t = (double)type;
    }
    V[lc] = v;
    G[lc] = 0.0;
    ct[lc] = t;
    for (int d = 0; d < g.N; ++d) {
      Cw[d][lc] = cw[d];
      if (Cg[d]) Cg[d][lc] = 0.0;
    }
  }
}

// The cut-cell quadrature is ~10^5 flops of branchy fp64 per box and there are only O(n^(N-1)) boxes: one thread per box
// leaves ~3 waves per SIMD walking long serial chains (20 + 43 ms at 512^3).  CUT_LANES lanes share a box instead: each
// takes one Gauss-Legendre node of every z piece and the partial moments are summed with shuffles.
constexpr int CUT_LANES = 16;   // = NGL

struct LaneGroup {
  __device__ void operator()(pggeom::Mom& m) const {
#pragma unroll
    for (int off = CUT_LANES / 2; off > 0; off >>= 1) {
      m.vol += __shfl_xor(m.vol, off, CUT_LANES);
      m.gamma += __shfl_xor(m.gamma, off, CUT_LANES);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        m.m[d] += __shfl_xor(m.m[d], off, CUT_LANES);
        m.gm[d] += __shfl_xor(m.gm[d], off, CUT_LANES);
      }
    }
  }
};

// K1/K5 (expensive part): CUT_LANES lanes per cut cell of the compacted list.
__global__ void k_cut_cells(GeoView g, BallSet bs, const int* cut_list, int ncut, double* V, double* G,
                            double* Cw0, double* Cw1, double* Cw2, double* Cg0, double* Cg1, double* Cg2) {
  double* Cw[3] = {Cw0, Cw1, Cw2};
  double* Cg[3] = {Cg0, Cg1, Cg2};
  const i64 gt = blockIdx.x * (i64)blockDim.x + threadIdx.x;
  const int ql = (int)(gt % CUT_LANES);
  i64 k = gt / CUT_LANES;
  const bool live = k < ncut;
  if (!live) k = ncut - 1;          // whole groups stay in the shuffles
  const i64 lc = cut_list[k];
  i64 idx[3];
  decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
  double lo[3], hi[3];
  cell_box(g, idx, lo, hi);
  const BoxMeasure m = box_measure(bs, lo, hi, true, c_gl, ql, CUT_LANES, LaneGroup());
  if (!live || ql != 0) return;
  V[lc] = m.vol;
  G[lc] = m.gamma;
  for (int d = 0; d < g.N; ++d) {
    Cw[d][lc] = m.cen[d];
    if (Cg[d]) Cg[d][lc] = m.cg[d];
  }
}

// K2 + K4: A_d (face x_d = nodes_d[i_d], i_d <= n_d, other dims real) and B_d (section through C_w[d]).
// A face or a centroid section of cell i lies inside the closed cell: if the cell is FULL (EMPTY) the section is FULL
// (EMPTY) by the same monotone far / near tests that classified the cell, with the same value prod_ext(...) (0) the
// geometric path returns -- so only cut cells (and the padding faces next to them) evaluate geometry: 17 -> 3 ms at 512^3.
__global__ void k_sections(GeoView g, BallSet bs, i64 Mloc, const double* ct, const double* Cw0, const double* Cw1,
                           const double* Cw2, double* A0, double* A1, double* A2, double* B0, double* B1,
                           double* B2) {
  const double* Cw[3] = {Cw0, Cw1, Cw2};
  double* A[3] = {A0, A1, A2};
  double* B[3] = {B0, B1, B2};
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
    const bool real = is_real_cell(g, idx);
    for (int d = 0; d < g.N; ++d) {
      // A_d: other dims must be real; the section box uses the cell extents of the other dims
      bool others_real = true;
      for (int k = 0; k < g.N; ++k)
        if (k != d && idx[k] >= g.n[k]) others_real = false;
      double a = 0.0, b = 0.0;
      if (others_real) {
        double lo[3], hi[3];
        for (int k = 0; k < g.N; ++k) {
          const i64 ik = idx[k] < g.n[k] ? idx[k] : g.n[k] - 1;
          lo[k] = g.nodes[k][ik];
          hi[k] = g.nodes[k][ik + 1];
        }
        // type of the real cell that contains the face: cell i, or cell i - e_d for the padding face i_d = n_d
        double t = (double)PG_CUT;
        if (real) t = ct[lc];
        else if (idx[d] >= g.n[d] && lc - g.stride[d] >= 0) t = ct[lc - g.stride[d]];
        // the faces / sections of a full (empty) cell are full (empty) -- except in 1-D, where a face is a POINT and a
        // point exactly on the interface counts as fluid (f <= 0) whichever cell owns it: always evaluated there
        if (g.N > 1 && t == (double)PG_FULL) {
          a = full_measure(g, d);
          b = real ? a : 0.0;
        } else if (g.N > 1 && t == (double)PG_EMPTY) {
          a = 0.0;
          b = 0.0;
        } else {
          a = section_measure(bs, d, g.nodes[d][idx[d]], lo, hi, full_measure(g, d));
          if (real) b = section_measure(bs, d, Cw[d][lc], lo, hi, full_measure(g, d));
        }
      }
      A[d][lc] = a;
      B[d][lc] = b;
    }
  }
}

// K3 (cheap part): W_d between the centroids of i-e_d and i; cut boxes go to the work list.
__global__ void k_stagger(GeoView g, BallSet bs, i64 Mloc, const double* ct, const double* Cw0, const double* Cw1,
                          const double* Cw2, double* W0, double* W1, double* W2, int* wlist, int* wcount, int wcap) {
  const double* Cw[3] = {Cw0, Cw1, Cw2};
  double* W[3] = {W0, W1, W2};
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
    for (int d = 0; d < g.N; ++d) {
      double w = 0.0;
      bool others_real = true;
      for (int k = 0; k < g.N; ++k)
        if (k != d && idx[k] >= g.n[k]) others_real = false;
      // prev = max(i-1, first), next = min(i, last real)      capacity.jl:401-402
      const i64 ip = idx[d] - 1 < 0 ? 0 : idx[d] - 1;
      const i64 in = idx[d] < g.n[d] - 1 ? idx[d] : g.n[d] - 1;
      const i64 lp = lc + (ip - idx[d]) * g.stride[d];
      const i64 ln = lc + (in - idx[d]) * g.stride[d];
      // the neighbour in the slowest dim may lie outside the stored planes: leave 0 (never read)
      const bool have = lp >= 0 && ln >= 0 && lp < Mloc && ln < Mloc;
      if (others_real && have && ip != in) {
        const double tp = ct[lp], tn = ct[ln];
        if (!(tp == 0.0 && tn == 0.0)) {   // capacity.jl:420-427
          double lo[3], hi[3];
          for (int k = 0; k < g.N; ++k) {
            if (k == d) continue;
            lo[k] = g.nodes[k][idx[k]];
            hi[k] = g.nodes[k][idx[k] + 1];
          }
          lo[d] = Cw[d][lp];
          hi[d] = Cw[d][ln];
          int type;
          pick_ball(bs, lo, hi, type);
          if (type != PG_CUT && bs.complement) type = 1 - type;
          bool degenerate = false;
          for (int k = 0; k < g.N; ++k)
            if (!(hi[k] - lo[k] > 0.0)) degenerate = true;
          if (!degenerate) {
            // between the centres of two full cells: exactly one cell volume
            if (type == PG_FULL) w = (tp == 1.0 && tn == 1.0) ? full_measure(g, -1) : prod_ext(lo, hi, g.N, -1);
            else if (type == PG_CUT) {
              const int slot = atomicAdd(wcount, 1);
              if (slot < wcap) {
                wlist[2 * slot] = (int)lc;
                wlist[2 * slot + 1] = d;
              }
            }
          }
        }
      }
      W[d][lc] = w;
    }
  }
}

__global__ void k_stagger_cut(GeoView g, BallSet bs, const int* wlist, int nw, const double* Cw0,
                              const double* Cw1, const double* Cw2, double* W0, double* W1, double* W2) {
  const double* Cw[3] = {Cw0, Cw1, Cw2};
  double* W[3] = {W0, W1, W2};
  const i64 gt = blockIdx.x * (i64)blockDim.x + threadIdx.x;
  const int ql = (int)(gt % CUT_LANES);
  i64 k = gt / CUT_LANES;
  const bool live = k < nw;
  if (!live) k = nw - 1;
  const i64 lc = wlist[2 * k];
  const int d = wlist[2 * k + 1];
  i64 idx[3];
  decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
  const i64 ip = idx[d] - 1 < 0 ? 0 : idx[d] - 1;
  const i64 in = idx[d] < g.n[d] - 1 ? idx[d] : g.n[d] - 1;
  const i64 lp = lc + (ip - idx[d]) * g.stride[d];
  const i64 ln = lc + (in - idx[d]) * g.stride[d];
  double lo[3], hi[3];
  for (int q = 0; q < g.N; ++q) {
    if (q == d) continue;
    lo[q] = g.nodes[q][idx[q]];
    hi[q] = g.nodes[q][idx[q] + 1];
  }
  lo[d] = Cw[d][lp];
  hi[d] = Cw[d][ln];
  const double w = box_measure(bs, lo, hi, false, c_gl, ql, CUT_LANES, LaneGroup()).vol;
  if (live && ql == 0) W[d][lc] = w;
}

// per-plane count of non-empty cells (slab balancing weights)
__global__ void k_plane_weights(GeoView g, BallSet bs, i64 plane_lo, i64 plane_hi, unsigned long long* weight) {
  const i64 total = (plane_hi - plane_lo) * g.plane;
  for (i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x; q < total; q += (i64)gridDim.x * blockDim.x) {
    const i64 p = q / g.plane + plane_lo;
    i64 rem = q % g.plane;
    i64 idx[3] = {0, 0, 0};
    for (int d = 0; d < g.N - 1; ++d) { idx[d] = rem % g.ext[d]; rem /= g.ext[d]; }
    idx[g.N - 1] = p;
    if (!is_real_cell(g, idx)) continue;
    double lo[3], hi[3];
    cell_box(g, idx, lo, hi);
    int type;
    pick_ball(bs, lo, hi, type);
    if (type != PG_CUT && bs.complement) type = 1 - type;
    if (type != PG_EMPTY) atomicAdd(&weight[p], type == PG_CUT ? 2ull : 1ull);
    else {
      // an empty cell on a lateral face of the box still becomes a row when a border condition is set there (identity
      // row of BC_border_mono!): 1.05 M of the 10.3 M rows of the 512^3 benchmark problem, spread over ALL planes --
      // left out, the slabs outside the body end up 20 % larger than the others when one body is cut into 8 slabs
      bool lateral = false;
      for (int d = 0; d < g.N - 1; ++d) lateral = lateral || idx[d] == 0 || idx[d] == g.n[d] - 1;
      if (lateral) atomicAdd(&weight[p], 1ull);
    }
  }
}

// ---- space-time capacities of one time slab [t0, t1] ------------------------------------------------------------
// Capacity(body, SpaceTimeMesh(mesh, [t, t+Δt])) of the reference (prescribedmotionsolver/diffusion.jl:251-252) hands the
// (N+1)-D cell  (space cell) x [t0, t1]  to libvofi.  Here the body is a ball / half space whose parameters move in
// time, and every first-layer capacity is the time integral of the corresponding spatial measure (exact in space, as
// the static kernels; composite Gauss-Legendre in time, nodes supplied by the host which evaluates the motion there):
//   V = ∫V(τ)dτ   C_ω = ∫∫(x, τ) / V   A_d = ∫A_d(τ)dτ   B_d = ∫|{x_d = C_ω,d} ∩ fluid(τ)|dτ   W_d = ∫|box(C_ω) ∩ fluid(τ)|dτ
//   Γ = ∫Γ(τ) sqrt(1 + v_n²) dτ  (v_n: normal speed of the interface at the spatial interface centroid)
//   A_(N+1) at the two time faces = V(t0), V(t1)   ("Vn_1", "Vn" of diffusion.jl:113-114)
struct MotionNode {
  double tau, w, c[3], r, dc[3], dr;   // half space: c[0] = position, dc[0] = its speed
};
struct MotionView {
  int kind, complement, axis, nq;
  double sgn, t0, t1;
  const MotionNode* q;
  MotionNode end[2];
  // the body at mid time and how far any of the bodies the rule sees (nodes and both faces) is from it: |c - c_mid| + |r - r_mid|
  // at most `reach` -- a cell FULL for the mid body shrunk by `reach` (EMPTY for it grown by `reach`) is so at every node
  MotionNode mid;
  double reach;
  // the body at node k (k < nq) and at the two time faces (nq, nq + 1), built by the host: every lane of a wave reads the
  // same one (scalar loads) -- a BallSet per lane would live in scratch
  const BallSet* bodies;
};

// Work is split as in the static kernels: a cheap pass over all cells finishes the ones that need no quadrature and
// lists the others; a second kernel gives every listed item one WAVE whose lanes take the time nodes (a box / section
// measure is a few thousand fp64 instructions with square roots and arc tangents: 64 of them in one lane, for the few
// cells near the interface, left the whole grid waiting -- 6.4 ms per slab at 1024², 0.3 ms this way).
__device__ inline double wave_add(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct StOut {
  double *V, *G, *ct, *Cw[3], *Cg[3], *Vt0, *Vt1, *Ctw, *Ctg;
};

__device__ inline void st_store_plain(const GeoView& g, const MotionView& mv, const StOut& o, i64 lc, const double* lo, const double* hi,
                                      bool real, bool fluid) {
  double cw[3] = {0.0, 0.0, 0.0};
  double v = 0.0, v01 = 0.0, t = 0.0, tw = 0.0;
  if (real) {
    for (int d = 0; d < g.N; ++d) cw[d] = 0.5 * (lo[d] + hi[d]);
    tw = 0.5 * (mv.t0 + mv.t1);
    if (fluid) { v = full_measure(g, -1) * (mv.t1 - mv.t0); v01 = full_measure(g, -1); t = (double)PG_FULL; }
  }
  o.V[lc] = v; o.G[lc] = 0.0; o.ct[lc] = t; o.Vt0[lc] = v01; o.Vt1[lc] = v01; o.Ctw[lc] = tw; o.Ctg[lc] = 0.0;
  for (int d = 0; d < g.N; ++d) {
    o.Cw[d][lc] = cw[d];
    if (o.Cg[d]) o.Cg[d][lc] = 0.0;
  }
}

// pass 1 of V, C_ω, Γ, C_γ, type, V(t0), V(t1): cells far from the interface during the whole slab are finished here
__global__ void k_st_classify(GeoView g, MotionView mv, i64 Mloc, StOut o, int* list, int* count) {
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    const bool real = is_real_cell(g, idx);
    int far = -1;   // 1: inside every body the rule sees, -1: outside every one, 0: near
    if (real) {
      cell_box(g, idx, lo, hi);
      far = 0;
      if (mv.kind == BODY_HALFSPACE) {
        const double pl = mv.mid.c[0] - mv.reach, ph = mv.mid.c[0] + mv.reach;
        if (hi[mv.axis] <= pl) far = mv.sgn > 0.0 ? 1 : -1;
        else if (lo[mv.axis] >= ph) far = mv.sgn > 0.0 ? -1 : 1;
      } else {
        if (mv.mid.r - mv.reach > 0.0 && ball_box_type(mv.mid.c, mv.mid.r - mv.reach, lo, hi, g.N) == PG_FULL) far = 1;
        else if (ball_box_type(mv.mid.c, mv.mid.r + mv.reach, lo, hi, g.N) == PG_EMPTY) far = -1;
      }
    }
    if (far != 0) st_store_plain(g, mv, o, lc, lo, hi, real, (far == 1) != (mv.complement != 0));
    else list[atomicAdd(count, 1)] = (int)lc;
  }
}

// pass 2: one wave per listed cell, lanes over the time nodes (+ lanes 0 / 1: the two time faces)
__global__ void k_st_cells(GeoView g, MotionView mv, const int* __restrict__ list, int nlist, StOut o) {
  const int lane = threadIdx.x & 63;
  const i64 item = (blockIdx.x * (i64)blockDim.x + threadIdx.x) >> 6;
  if (item >= nlist) return;
  const i64 lc = list[item];
  i64 idx[3];
  decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
  double lo[3], hi[3];
  cell_box(g, idx, lo, hi);
  double v = 0.0, gam = 0.0, momt = 0.0, gmt = 0.0, vend = 0.0;
  double mom[3] = {0.0, 0.0, 0.0}, gm[3] = {0.0, 0.0, 0.0};
  bool full = true, empty = true;
  if (lane < 2) {
    const BoxMeasure m = box_measure(mv.bodies[mv.nq + lane], lo, hi, false, c_gl);
    vend = m.vol;
    full = m.type == PG_FULL;
    empty = m.type == PG_EMPTY;
  }
  for (int k = lane; k < mv.nq; k += 64) {
    const MotionNode q = mv.q[k];
    const BoxMeasure m = box_measure(mv.bodies[k], lo, hi, true, c_gl);
    full = full && m.type == PG_FULL;
    empty = empty && m.type == PG_EMPTY;
    const double wv = q.w * m.vol;
    v += wv;
    momt += wv * q.tau;
    for (int d = 0; d < g.N; ++d) mom[d] += wv * m.cen[d];
    if (m.gamma > 0.0) {
      double vn;
      if (mv.kind == BODY_HALFSPACE) vn = q.dc[0];
      else {
        double nn = 0.0, dot = 0.0;
        for (int d = 0; d < g.N; ++d) {
          const double e = m.cg[d] - q.c[d];
          nn += e * e;
          dot += e * q.dc[d];
        }
        vn = q.dr + (nn > 0.0 ? dot / sqrt(nn) : 0.0);
      }
      const double ws = q.w * m.gamma * sqrt(1.0 + vn * vn);
      gam += ws;
      gmt += ws * q.tau;
      for (int d = 0; d < g.N; ++d) gm[d] += ws * m.cg[d];
    }
  }
  const bool all_full = __all(full), all_empty = __all(empty);
  v = wave_add(v); gam = wave_add(gam); momt = wave_add(momt); gmt = wave_add(gmt);
  for (int d = 0; d < g.N; ++d) { mom[d] = wave_add(mom[d]); gm[d] = wave_add(gm[d]); }
  const double v0 = __shfl(vend, 0, 64), v1 = __shfl(vend, 1, 64);
  if (lane != 0) return;
  if (all_full || all_empty) {
    st_store_plain(g, mv, o, lc, lo, hi, true, all_full);
    return;
  }
  double cw[3], cg[3] = {0.0, 0.0, 0.0}, tw = 0.5 * (mv.t0 + mv.t1), tg = 0.0;
  for (int d = 0; d < g.N; ++d) cw[d] = 0.5 * (lo[d] + hi[d]);
  if (v > 0.0) {
    for (int d = 0; d < g.N; ++d) cw[d] = mom[d] / v;
    tw = momt / v;
  }
  if (gam > 0.0) {
    for (int d = 0; d < g.N; ++d) cg[d] = gm[d] / gam;
    tg = gmt / gam;
  }
  o.V[lc] = v; o.G[lc] = gam; o.ct[lc] = (double)PG_CUT; o.Vt0[lc] = v0; o.Vt1[lc] = v1; o.Ctw[lc] = tw; o.Ctg[lc] = tg;
  for (int d = 0; d < g.N; ++d) {
    o.Cw[d][lc] = cw[d];
    if (o.Cg[d]) o.Cg[d][lc] = cg[d];
  }
}

// A_d, B_d: the conventions of k_sections, integrated over the slab.  Pass 1 writes the faces that need no quadrature and
// lists (cell, d) for the others; pass 2: one wave per item.
__global__ void k_st_sections(GeoView g, MotionView mv, i64 Mloc, const double* ct, double* A0, double* A1, double* A2, double* B0,
                              double* B1, double* B2, int* list, int* count, int cap) {
  double* A[3] = {A0, A1, A2};
  double* B[3] = {B0, B1, B2};
  const double dt = mv.t1 - mv.t0;
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
    const bool real = is_real_cell(g, idx);
    for (int d = 0; d < g.N; ++d) {
      bool others_real = true;
      for (int k = 0; k < g.N; ++k)
        if (k != d && idx[k] >= g.n[k]) others_real = false;
      double a = 0.0, b = 0.0;
      if (others_real) {
        double t = (double)PG_CUT;
        if (real) t = ct[lc];
        else if (idx[d] >= g.n[d] && lc - g.stride[d] >= 0) t = ct[lc - g.stride[d]];
        if (g.N > 1 && t == (double)PG_FULL) {
          a = full_measure(g, d) * dt;
          b = real ? a : 0.0;
        } else if (!(g.N > 1 && t == (double)PG_EMPTY)) {
          const int slot = atomicAdd(count, 1);
          if (slot < cap) { list[2 * slot] = (int)lc; list[2 * slot + 1] = d; }
        }
      }
      A[d][lc] = a;
      B[d][lc] = b;
    }
  }
}

__global__ void k_st_sections_cut(GeoView g, MotionView mv, const int* __restrict__ list, int nlist, const double* Cw0,
                                  const double* Cw1, const double* Cw2, double* A0, double* A1, double* A2, double* B0,
                                  double* B1, double* B2) {
  const double* Cw[3] = {Cw0, Cw1, Cw2};
  double* A[3] = {A0, A1, A2};
  double* B[3] = {B0, B1, B2};
  const int lane = threadIdx.x & 63;
  const i64 item = (blockIdx.x * (i64)blockDim.x + threadIdx.x) >> 6;
  if (item >= nlist) return;
  const i64 lc = list[2 * item];
  const int d = list[2 * item + 1];
  i64 idx[3];
  decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
  const bool real = is_real_cell(g, idx);
  double lo[3], hi[3];
  for (int k = 0; k < g.N; ++k) {
    const i64 ik = idx[k] < g.n[k] ? idx[k] : g.n[k] - 1;
    lo[k] = g.nodes[k][ik];
    hi[k] = g.nodes[k][ik + 1];
  }
  double a = 0.0, b = 0.0;
  const double fm = full_measure(g, d);
  bool a_full = true, b_full = real;
  for (int k = lane; k < mv.nq; k += 64) {
    const double wq = mv.q[k].w;
    const double sa = section_measure(mv.bodies[k], d, g.nodes[d][idx[d]], lo, hi, fm);
    a += wq * sa;
    a_full = a_full && sa == fm;
    if (real) {
      const double sb = section_measure(mv.bodies[k], d, Cw[d][lc], lo, hi, fm);
      b += wq * sb;
      b_full = b_full && sb == fm;
    }
  }
  a = wave_add(a);
  b = wave_add(b);
  // a section that is full at every time node gets the measure the full cells get, full x Δt, and not the quadrature sum
  // Σ w_k full, which differs from it in the last bits: A_d - B_d between such a face and a full neighbour is then an EXACT
  // zero (as in the static kernels) instead of 1e-17 -- a residue that switches the γ unknowns of full cells on with
  // 1e-34 diagonals and makes their cell blocks singular (the moving two-phase systems met it)
  if (__all(a_full)) a = fm * (mv.t1 - mv.t0);
  if (__all(b_full)) b = fm * (mv.t1 - mv.t0);
  if (lane == 0) { A[d][lc] = a; B[d][lc] = b; }
}

// W_d: the conventions of k_stagger, integrated over the slab; same two passes
__global__ void k_st_stagger(GeoView g, MotionView mv, i64 Mloc, const double* ct, const double* Cw0, const double* Cw1,
                             const double* Cw2, double* W0, double* W1, double* W2, int* list, int* count, int cap) {
  const double* Cw[3] = {Cw0, Cw1, Cw2};
  double* W[3] = {W0, W1, W2};
  const double dt = mv.t1 - mv.t0;
  for (i64 lc = blockIdx.x * (i64)blockDim.x + threadIdx.x; lc < Mloc; lc += (i64)gridDim.x * blockDim.x) {
    i64 idx[3];
    decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
    for (int d = 0; d < g.N; ++d) {
      double w = 0.0;
      bool others_real = true;
      for (int k = 0; k < g.N; ++k)
        if (k != d && idx[k] >= g.n[k]) others_real = false;
      const i64 ip = idx[d] - 1 < 0 ? 0 : idx[d] - 1;
      const i64 in = idx[d] < g.n[d] - 1 ? idx[d] : g.n[d] - 1;
      const i64 lp = lc + (ip - idx[d]) * g.stride[d];
      const i64 ln = lc + (in - idx[d]) * g.stride[d];
      const bool have = lp >= 0 && ln >= 0 && lp < Mloc && ln < Mloc;
      if (others_real && have && ip != in) {
        const double tp = ct[lp], tn = ct[ln];
        if (!(tp == 0.0 && tn == 0.0)) {
          bool degenerate = !(Cw[d][ln] - Cw[d][lp] > 0.0);
          for (int k = 0; k < g.N; ++k)
            if (k != d && !(g.nodes[k][idx[k] + 1] - g.nodes[k][idx[k]] > 0.0)) degenerate = true;
          if (!degenerate) {
            if (tp == 1.0 && tn == 1.0) w = full_measure(g, -1) * dt;
            else {
              const int slot = atomicAdd(count, 1);
              if (slot < cap) { list[2 * slot] = (int)lc; list[2 * slot + 1] = d; }
            }
          }
        }
      }
      W[d][lc] = w;
    }
  }
}

__global__ void k_st_stagger_cut(GeoView g, MotionView mv, const int* __restrict__ list, int nlist, const double* Cw0,
                                 const double* Cw1, const double* Cw2, double* W0, double* W1, double* W2) {
  const double* Cw[3] = {Cw0, Cw1, Cw2};
  double* W[3] = {W0, W1, W2};
  const int lane = threadIdx.x & 63;
  const i64 item = (blockIdx.x * (i64)blockDim.x + threadIdx.x) >> 6;
  if (item >= nlist) return;
  const i64 lc = list[2 * item];
  const int d = list[2 * item + 1];
  i64 idx[3];
  decode_cell(g.N, g.ext, g.plane, g.s0, lc, idx);
  const i64 ip = idx[d] - 1 < 0 ? 0 : idx[d] - 1;
  const i64 in = idx[d] < g.n[d] - 1 ? idx[d] : g.n[d] - 1;
  const i64 lp = lc + (ip - idx[d]) * g.stride[d];
  const i64 ln = lc + (in - idx[d]) * g.stride[d];
  double lo[3], hi[3];
  for (int k = 0; k < g.N; ++k) {
    if (k == d) continue;
    lo[k] = g.nodes[k][idx[k]];
    hi[k] = g.nodes[k][idx[k] + 1];
  }
  lo[d] = Cw[d][lp];
  hi[d] = Cw[d][ln];
  double w = 0.0;
  for (int k = lane; k < mv.nq; k += 64) w += mv.q[k].w * box_measure(mv.bodies[k], lo, hi, false, c_gl).vol;
  w = wave_add(w);
  if (lane == 0) W[d][lc] = w;
}

GeoView geo_view(pg_mesh* m, const Slab& s) {
  GeoView g;
  g.N = s.N;
  for (int d = 0; d < 3; ++d) {
    g.ext[d] = s.ext[d];
    g.n[d] = s.n[d];
    g.stride[d] = s.stride[d];
    g.nodes[d] = nullptr;
  }
  g.plane = s.plane;
  g.s0 = s.s0;
  g.s1 = s.s1;
  for (int d = 0; d < s.N; ++d) {
    if (!m->d_nodes[d].p) {
      m->d_nodes[d].alloc(m->n[d] + 1);
      m->d_nodes[d].upload(m->nodes[d].data(), m->n[d] + 1);
    }
    g.nodes[d] = m->d_nodes[d].p;
  }
  for (int d = 0; d < 3; ++d) g.h[d] = d < s.N ? m->L[d] / (double)m->n[d] : 1.0;
  return g;
}

void alloc_fields(pg_capacity* c) {
  const i64 Ml = c->slab.Mloc();
  PG_REQUIRE(Ml < (i64)2147483647, "capacity: local slab exceeds 2^31 cells");
  c->V.alloc(Ml); c->G.alloc(Ml); c->ct.alloc(Ml);
  for (int d = 0; d < c->N; ++d) {
    c->A[d].alloc(Ml); c->B[d].alloc(Ml); c->W[d].alloc(Ml); c->Cw[d].alloc(Ml);
    if (c->has_cg) c->Cg[d].alloc(Ml);
  }
}

bool g_gl_uploaded = false;
void ensure_gl() {
  if (g_gl_uploaded) return;
  GLTable t;
  gl_init(t);
  PG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_gl), &t, sizeof(t)));
  g_gl_uploaded = true;
}

}  // namespace

namespace pg {
CapView cap_view(const pg_capacity* c) {
  CapView v;
  v.N = c->N;
  for (int d = 0; d < 3; ++d) {
    v.ext[d] = c->slab.ext[d];
    v.n[d] = c->slab.n[d];
    v.stride[d] = c->slab.stride[d];
    v.A[d] = c->A[d].p;
    v.B[d] = c->B[d].p;
    v.W[d] = c->W[d].p;
  }
  v.plane = c->slab.plane;
  v.s0 = c->slab.s0;
  v.s1 = c->slab.s1;
  v.nplanes = c->slab.nplanes;
  v.V = c->V.p;
  v.G = c->G.p;
  return v;
}
}  // namespace pg

extern "C" {

int32_t pg_capacity_create_levelset(pg_mesh* m, int32_t body_kind, const double* params, int32_t nparams,
                                    int32_t flags, pg_capacity** out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(m && params && out, "pg_capacity_create_levelset: NULL argument");
  Context& cx = ctx();
  const int N = m->N;
  BallSet bs;
  std::memset(&bs, 0, sizeof(bs));
  bs.N = N;
  bs.complement = (flags & PG_FLAG_COMPLEMENT) ? 1 : 0;
  for (int d = 0; d < 3; ++d) bs.ax[d] = 1.0;
  if (body_kind == PG_BODY_BALL) {
    PG_REQUIRE(nparams == N + 1, "PG_BODY_BALL expects params = {c_1..c_N, r}");
    bs.nballs = 1;
    for (int d = 0; d < N; ++d) bs.c[0][d] = params[d];
    bs.r = params[N];
  } else if (body_kind == PG_BODY_MULTIBALL) {
    PG_REQUIRE(nparams >= 2, "PG_BODY_MULTIBALL expects params = {r, nballs, centres...}");
    bs.r = params[0];
    bs.nballs = (int)params[1];
    PG_REQUIRE(bs.nballs >= 1 && bs.nballs <= MAX_BALLS, "PG_BODY_MULTIBALL: 1..16 balls");
    PG_REQUIRE(nparams == 2 + bs.nballs * N, "PG_BODY_MULTIBALL: wrong parameter count");
    PG_REQUIRE(!bs.complement, "PG_BODY_MULTIBALL: complement not supported");
    for (int s = 0; s < bs.nballs; ++s)
      for (int d = 0; d < N; ++d) bs.c[s][d] = params[2 + s * N + d];
  } else if (body_kind == PG_BODY_HALFSPACE) {
    PG_REQUIRE(nparams == 3, "PG_BODY_HALFSPACE expects params = {axis (0-based), position, sign}");
    bs.kind = BODY_HALFSPACE;
    bs.axis = (int)params[0];
    bs.pos = params[1];
    bs.sgn = params[2] < 0.0 ? -1.0 : 1.0;
    bs.nballs = 1;
    bs.r = 1.0;
    PG_REQUIRE(bs.axis >= 0 && bs.axis < N && (double)bs.axis == params[0], "PG_BODY_HALFSPACE: axis must be 0 .. N-1");
  } else if (body_kind == PG_BODY_ELLIPSOID) {
    PG_REQUIRE(nparams == 2 * N, "PG_BODY_ELLIPSOID expects params = {c_1..c_N, a_1..a_N}");
    bs.nballs = 1;
    bs.r = 1.0;
    for (int d = 0; d < N; ++d) {
      bs.c[0][d] = params[d];
      bs.ax[d] = params[N + d];
      PG_REQUIRE(bs.ax[d] > 0.0, "PG_BODY_ELLIPSOID: semi-axes must be positive");
    }
    if (N == 1) bs.r = bs.ax[0];                 // an interval: the 1-D ball
    else bs.kind = BODY_ELLIPSOID;
  } else {
    throw Error("pg_capacity_create_levelset: unknown body kind (arbitrary bodies: use pg_capacity_create_from_arrays)");
  }
  PG_REQUIRE(bs.r > 0.0, "ball radius must be positive");

  auto* c = new pg_capacity();
  std::unique_ptr<pg_capacity> guard(c);
  c->mesh = m;
  c->N = N;
  c->from_body = true;
  c->has_cg = !(flags & PG_FLAG_NO_CENTROIDS);
  c->body = bs;
  c->slab = m->base_slab();
  ensure_gl();
  hipStream_t st = cx.stream;

  // ---- slab partition balanced by non-empty cells per plane ----------------
  if (cx.nranks > 1 || cx.comm) {
    const Slab& s = c->slab;
    PG_REQUIRE(s.nplanes >= cx.nranks, "fewer planes than ranks");
    GeoView g = geo_view(m, s);
    DevBuf<unsigned long long> w(s.nplanes);
    w.zero();
    const i64 lo = s.nplanes * cx.rank / cx.nranks, hi = s.nplanes * (cx.rank + 1) / cx.nranks;
    if (hi > lo)
      hipLaunchKernelGGL(k_plane_weights, dim3(grid_for((hi - lo) * s.plane, 256)), dim3(256), 0, st, g, bs, lo, hi, w.p);
    PG_HIP(hipGetLastError());
    comm_allreduce_sum_u64(w.p, s.nplanes, st);
    std::vector<unsigned long long> hw(s.nplanes);
    w.download(hw.data(), s.nplanes);
    std::vector<i64> wt(hw.begin(), hw.end()), bounds(cx.nranks + 1);
    PG_REQUIRE(pg_partition_planes(wt.data(), s.nplanes, cx.nranks, bounds.data()) == 0, "partition failed");
    c->slab.set_own(bounds[cx.rank], bounds[cx.rank + 1]);
  }

  alloc_fields(c);
  const Slab& s = c->slab;
  const i64 Ml = s.Mloc();
  GeoView g = geo_view(m, s);
  DevBuf<int> cut_list(Ml), counters(2);
  DevBuf<int> wlist;
  counters.zero();
  hipEvent_t e0, e1;
  PG_HIP(hipEventCreate(&e0));
  PG_HIP(hipEventCreate(&e1));
  PG_HIP(hipEventRecord(e0, st));
  const int gr = grid_for(Ml, 256, 256 * 16);
  hipLaunchKernelGGL(k_classify, dim3(gr), dim3(256), 0, st, g, bs, Ml, c->V.p, c->G.p, c->ct.p, c->Cw[0].p,
                     c->Cw[1].p, c->Cw[2].p, c->Cg[0].p, c->Cg[1].p, c->Cg[2].p, cut_list.p, counters.p);
  PG_HIP(hipGetLastError());
  int hc[2];
  counters.download(hc, 2);
  const int ncut = hc[0];
  c->n_cut_local = ncut;
  if (ncut > 0) {
    hipLaunchKernelGGL(k_cut_cells, dim3((unsigned)(((i64)ncut * CUT_LANES + 255) / 256)), dim3(256), 0, st, g, bs, cut_list.p, ncut, c->V.p, c->G.p,
                       c->Cw[0].p, c->Cw[1].p, c->Cw[2].p, c->Cg[0].p, c->Cg[1].p, c->Cg[2].p);
    PG_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_sections, dim3(gr), dim3(256), 0, st, g, bs, Ml, c->ct.p, c->Cw[0].p, c->Cw[1].p, c->Cw[2].p, c->A[0].p,
                     c->A[1].p, c->A[2].p, c->B[0].p, c->B[1].p, c->B[2].p);
  PG_HIP(hipGetLastError());
  // W work list: at most N entries per cut-adjacent cell; bound by 2*N*ncut + slack
  const i64 wcap = std::min<i64>((i64)N * Ml, (i64)4 * N * (i64)ncut + 1024);
  wlist.alloc(2 * wcap);
  hipLaunchKernelGGL(k_stagger, dim3(gr), dim3(256), 0, st, g, bs, Ml, c->ct.p, c->Cw[0].p, c->Cw[1].p, c->Cw[2].p,
                     c->W[0].p, c->W[1].p, c->W[2].p, wlist.p, counters.p + 1, (int)wcap);
  PG_HIP(hipGetLastError());
  counters.download(hc, 2);
  const int nw = hc[1];
  PG_REQUIRE(nw <= wcap, "internal: staggered-volume work list overflow");
  if (nw > 0) {
    hipLaunchKernelGGL(k_stagger_cut, dim3((unsigned)(((i64)nw * CUT_LANES + 255) / 256)), dim3(256), 0, st, g, bs, wlist.p, nw, c->Cw[0].p, c->Cw[1].p,
                       c->Cw[2].p, c->W[0].p, c->W[1].p, c->W[2].p);
    PG_HIP(hipGetLastError());
  }
  PG_HIP(hipEventRecord(e1, st));
  PG_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PG_HIP(hipEventElapsedTime(&ms, e0, e1));
  c->kernel_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *out = guard.release();
  PG_API_END
}

int32_t pg_capacity_create_from_arrays(pg_mesh* m, const double* V, const double* const* A, const double* const* B,
                                       const double* const* W, const double* Gamma, const double* const* C_omega,
                                       const double* const* C_gamma, const double* cell_types, pg_capacity** out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(m && V && A && B && W && Gamma && C_omega && cell_types && out, "pg_capacity_create_from_arrays: NULL argument");
  PG_REQUIRE(ctx().nranks == 1, "pg_capacity_create_from_arrays: single rank only");
  auto* c = new pg_capacity();
  std::unique_ptr<pg_capacity> guard(c);
  c->mesh = m;
  c->N = m->N;
  c->from_body = false;
  c->has_cg = C_gamma != nullptr;
  c->slab = m->base_slab();
  alloc_fields(c);
  const i64 M = c->slab.M;
  c->V.upload(V, M);
  c->G.upload(Gamma, M);
  c->ct.upload(cell_types, M);
  for (int d = 0; d < c->N; ++d) {
    c->A[d].upload(A[d], M);
    c->B[d].upload(B[d], M);
    c->W[d].upload(W[d], M);
    c->Cw[d].upload(C_omega[d], M);
    if (c->has_cg) c->Cg[d].upload(C_gamma[d], M);
  }
  *out = guard.release();
  PG_API_END
}

int32_t pg_capacity_create_spacetime(pg_mesh* m, const pg_motion_desc* mo, pg_capacity** out) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(m && mo && out && mo->nodes, "pg_capacity_create_spacetime: NULL argument");
  AsyncAllocScope pool;   // one capacity per time slab: the stream-ordered allocator (pg_common.h)
  PG_REQUIRE(ctx().nranks == 1 && !ctx().comm, "pg_capacity_create_spacetime: single rank only");
  const int N = m->N;
  PG_REQUIRE(N == 1 || N == 2, "space-time capacities: 1-D+t and 2-D+t (the reference's 3-D+t blocks drop the z direction)");
  PG_REQUIRE(mo->body_kind == PG_BODY_BALL || mo->body_kind == PG_BODY_HALFSPACE, "moving body: PG_BODY_BALL or PG_BODY_HALFSPACE");
  PG_REQUIRE(mo->nq >= 1 && mo->nq <= 4096, "pg_capacity_create_spacetime: 1 .. 4096 time nodes");
  PG_REQUIRE(mo->t1 > mo->t0, "pg_capacity_create_spacetime: empty time slab");
  MotionView mv;
  std::memset(&mv, 0, sizeof(mv));
  mv.kind = mo->body_kind == PG_BODY_HALFSPACE ? BODY_HALFSPACE : BODY_BALLS;
  mv.complement = (mo->flags & PG_FLAG_COMPLEMENT) ? 1 : 0;
  mv.axis = mo->axis;
  mv.sgn = mo->sign < 0.0 ? -1.0 : 1.0;
  mv.nq = mo->nq;
  mv.t0 = mo->t0;
  mv.t1 = mo->t1;
  if (mv.kind == BODY_HALFSPACE) PG_REQUIRE(mv.axis >= 0 && mv.axis < N, "moving half space: axis must be 0 .. N-1");
  std::vector<MotionNode> hq(mo->nq);
  double wsum = 0.0;
  for (int k = 0; k < mo->nq; ++k) {
    const double* r = mo->nodes + (size_t)k * 10;
    MotionNode& q = hq[k];
    q.tau = r[0]; q.w = r[1];
    for (int d = 0; d < 3; ++d) { q.c[d] = r[2 + d]; q.dc[d] = r[6 + d]; }
    q.r = r[5]; q.dr = r[9];
    PG_REQUIRE(q.tau > mo->t0 && q.tau < mo->t1, "time nodes must lie inside (t0, t1)");
    if (mv.kind == BODY_BALLS) PG_REQUIRE(q.r > 0.0, "ball radius must be positive");
    wsum += q.w;
  }
  PG_REQUIRE(fabs(wsum - (mo->t1 - mo->t0)) <= 1e-12 * (mo->t1 - mo->t0), "time weights must add up to t1 - t0");
  for (int e = 0; e < 2; ++e) {
    const double* b = e == 0 ? mo->body0 : mo->body1;
    std::memset(&mv.end[e], 0, sizeof(MotionNode));
    mv.end[e].tau = e == 0 ? mo->t0 : mo->t1;
    for (int d = 0; d < 3; ++d) mv.end[e].c[d] = b[d];
    mv.end[e].r = b[3];
    if (mv.kind == BODY_BALLS) PG_REQUIRE(b[3] > 0.0, "ball radius must be positive");
  }
  DevBuf<MotionNode> dq(mo->nq);
  dq.upload(hq.data(), mo->nq);
  mv.q = dq.p;
  {
    const double tm = 0.5 * (mo->t0 + mo->t1);
    int km = 0;
    for (int k = 1; k < mo->nq; ++k)
      if (fabs(hq[k].tau - tm) < fabs(hq[km].tau - tm)) km = k;
    mv.mid = hq[km];
    double reach = 0.0;
    auto dist = [&](const MotionNode& q) {
      double dc = 0.0;
      if (mv.kind == BODY_HALFSPACE) return fabs(q.c[0] - mv.mid.c[0]);
      for (int d = 0; d < N; ++d) dc += (q.c[d] - mv.mid.c[d]) * (q.c[d] - mv.mid.c[d]);
      return std::sqrt(dc) + fabs(q.r - mv.mid.r);
    };
    for (int k = 0; k < mo->nq; ++k) reach = std::max(reach, dist(hq[k]));
    reach = std::max(reach, std::max(dist(mv.end[0]), dist(mv.end[1])));
    mv.reach = reach * (1.0 + 1e-12) + 1e-300;
  }
  std::vector<BallSet> hb(mo->nq + 2);
  for (int k = 0; k < mo->nq + 2; ++k) {
    const MotionNode& q = k < mo->nq ? hq[k] : mv.end[k - mo->nq];
    BallSet& b = hb[k];
    std::memset(&b, 0, sizeof(b));
    b.N = N; b.nballs = 1; b.complement = mv.complement; b.kind = mv.kind; b.axis = mv.axis; b.sgn = mv.sgn;
    b.r = mv.kind == BODY_HALFSPACE ? 1.0 : q.r;
    b.pos = mv.kind == BODY_HALFSPACE ? q.c[0] : 0.0;
    for (int d = 0; d < 3; ++d) { b.c[0][d] = mv.kind == BODY_HALFSPACE ? 0.0 : q.c[d]; b.ax[d] = 1.0; }
  }
  DevBuf<BallSet> dbodies(mo->nq + 2);
  dbodies.upload(hb.data(), mo->nq + 2);
  mv.bodies = dbodies.p;

  auto* c = new pg_capacity();
  std::unique_ptr<pg_capacity> guard(c);
  c->mesh = m;
  c->N = N;
  c->from_body = false;        // (the stored body is not the geometry: nothing may re-evaluate it)
  c->has_cg = !(mo->flags & PG_FLAG_NO_CENTROIDS);
  c->spacetime = true;
  c->t0 = mo->t0;
  c->t1 = mo->t1;
  c->slab = m->base_slab();
  ensure_gl();
  alloc_fields(c);
  const i64 Ml = c->slab.Mloc();
  c->Vt[0].alloc(Ml); c->Vt[1].alloc(Ml); c->Ctw.alloc(Ml); c->Ctg.alloc(Ml);
  hipStream_t st = ctx().stream;
  GeoView g = geo_view(m, c->slab);
  EventPair ev;
  PG_HIP(hipEventRecord(ev.e0, st));
  const int gr = grid_for(Ml, 256, 256 * 16);
  StOut so;
  so.V = c->V.p; so.G = c->G.p; so.ct = c->ct.p; so.Vt0 = c->Vt[0].p; so.Vt1 = c->Vt[1].p; so.Ctw = c->Ctw.p; so.Ctg = c->Ctg.p;
  for (int d = 0; d < 3; ++d) { so.Cw[d] = c->Cw[d].p; so.Cg[d] = c->Cg[d].p; }
  DevBuf<int> near_list(Ml), counters(3);
  counters.zero();
  hipLaunchKernelGGL(k_st_classify, dim3(gr), dim3(256), 0, st, g, mv, Ml, so, near_list.p, counters.p);
  PG_HIP(hipGetLastError());
  int hc[3];
  counters.download(hc, 3);
  const int nnear = hc[0];
  c->n_cut_local = nnear;
  auto waves = [](i64 items) { return dim3((unsigned)((items * 64 + 255) / 256)); };
  if (nnear > 0) {
    hipLaunchKernelGGL(k_st_cells, waves(nnear), dim3(256), 0, st, g, mv, near_list.p, nnear, so);
    PG_HIP(hipGetLastError());
  }
  // at most N faces / staggered volumes per near cell and per neighbour of one (1-D: every face is evaluated)
  const i64 cap = std::min<i64>((i64)N * Ml, (N == 1 ? (i64)Ml : (i64)6 * N * (i64)nnear) + 1024);
  DevBuf<int> list2(2 * cap);
  hipLaunchKernelGGL(k_st_sections, dim3(gr), dim3(256), 0, st, g, mv, Ml, c->ct.p, c->A[0].p, c->A[1].p, c->A[2].p, c->B[0].p,
                     c->B[1].p, c->B[2].p, list2.p, counters.p + 1, (int)cap);
  PG_HIP(hipGetLastError());
  counters.download(hc, 3);
  PG_REQUIRE(hc[1] <= cap, "internal: space-time section work list overflow");
  if (hc[1] > 0) {
    hipLaunchKernelGGL(k_st_sections_cut, waves(hc[1]), dim3(256), 0, st, g, mv, list2.p, hc[1], c->Cw[0].p, c->Cw[1].p, c->Cw[2].p,
                       c->A[0].p, c->A[1].p, c->A[2].p, c->B[0].p, c->B[1].p, c->B[2].p);
    PG_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_st_stagger, dim3(gr), dim3(256), 0, st, g, mv, Ml, c->ct.p, c->Cw[0].p, c->Cw[1].p, c->Cw[2].p, c->W[0].p,
                     c->W[1].p, c->W[2].p, list2.p, counters.p + 2, (int)cap);
  PG_HIP(hipGetLastError());
  counters.download(hc, 3);
  PG_REQUIRE(hc[2] <= cap, "internal: space-time staggered-volume work list overflow");
  if (hc[2] > 0) {
    hipLaunchKernelGGL(k_st_stagger_cut, waves(hc[2]), dim3(256), 0, st, g, mv, list2.p, hc[2], c->Cw[0].p, c->Cw[1].p, c->Cw[2].p,
                       c->W[0].p, c->W[1].p, c->W[2].p);
    PG_HIP(hipGetLastError());
  }
  PG_HIP(hipEventRecord(ev.e1, st));
  PG_HIP(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  PG_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  c->kernel_ms = ms;
  *out = guard.release();
  PG_API_END
}

int32_t pg_capacity_set_spacetime(pg_capacity* c, double t0, double t1, const double* V_t0, const double* V_t1,
                                  const double* Ct_omega, const double* Ct_gamma) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(c && V_t0 && V_t1, "pg_capacity_set_spacetime: NULL argument");
  PG_REQUIRE(!c->from_body && ctx().nranks == 1, "pg_capacity_set_spacetime: for capacities built from arrays, single rank");
  const i64 M = c->slab.M;
  c->Vt[0].alloc(M); c->Vt[1].alloc(M); c->Ctw.alloc(M); c->Ctg.alloc(M);
  c->Vt[0].upload(V_t0, M);
  c->Vt[1].upload(V_t1, M);
  c->Ctw.zero(); c->Ctg.zero();
  if (Ct_omega) c->Ctw.upload(Ct_omega, M);
  if (Ct_gamma) c->Ctg.upload(Ct_gamma, M);
  c->spacetime = true;
  c->t0 = t0; c->t1 = t1;
  PG_API_END
}

int32_t pg_capacity_destroy(pg_capacity* c) {
  PG_API_BEGIN
  delete c;
  PG_API_END
}

int32_t pg_capacity_get(const pg_capacity* c, int32_t field, int32_t d, double* out, int64_t len) {
  PG_API_BEGIN
  require_init();
  PG_REQUIRE(c && out, "pg_capacity_get: NULL argument");
  PG_REQUIRE(len == c->slab.M, "pg_capacity_get: len must be prod(n_d+1)");
  const DevBuf<double>* src = nullptr;
  switch (field) {
    case PG_CAP_V: src = &c->V; break;
    case PG_CAP_GAMMA: src = &c->G; break;
    case PG_CAP_CELL_TYPES: src = &c->ct; break;
    case PG_CAP_A: case PG_CAP_B: case PG_CAP_W: case PG_CAP_C_OMEGA: case PG_CAP_C_GAMMA:
      PG_REQUIRE(d >= 0 && d < c->N, "pg_capacity_get: bad dimension");
      src = field == PG_CAP_A ? &c->A[d] : field == PG_CAP_B ? &c->B[d] : field == PG_CAP_W ? &c->W[d]
          : field == PG_CAP_C_OMEGA ? &c->Cw[d] : &c->Cg[d];
      break;
    case PG_CAP_ST_V0: case PG_CAP_ST_V1: case PG_CAP_ST_CT_OMEGA: case PG_CAP_ST_CT_GAMMA:
      PG_REQUIRE(c->spacetime, "pg_capacity_get: not a space-time capacity");
      src = field == PG_CAP_ST_V0 ? &c->Vt[0] : field == PG_CAP_ST_V1 ? &c->Vt[1] : field == PG_CAP_ST_CT_OMEGA ? &c->Ctw : &c->Ctg;
      break;
    default: throw Error("pg_capacity_get: unknown field");
  }
  PG_REQUIRE(src->p != nullptr, "pg_capacity_get: field not available (centroids disabled?)");
  src->download(out + c->slab.first_cell(), c->slab.Mloc());
  PG_API_END
}

int32_t pg_capacity_kernel_ms(const pg_capacity* c, double* ms) {
  PG_API_BEGIN
  PG_REQUIRE(c && ms, "pg_capacity_kernel_ms: NULL argument");
  *ms = c->kernel_ms;
  PG_API_END
}

int32_t pg_diffops_create(pg_capacity* c, pg_diffops** out) {
  PG_API_BEGIN
  PG_REQUIRE(c && out, "pg_diffops_create: NULL argument");
  auto* o = new pg_diffops();
  o->cap = c;
  *out = o;
  PG_API_END
}

int32_t pg_diffops_destroy(pg_diffops* o) {
  PG_API_BEGIN
  delete o;
  PG_API_END
}

}  // extern "C"
