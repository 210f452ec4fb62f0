"""CPU restatement of the reference's prescribed-motion diffusion path -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

  SpaceTimeMesh                               /root/reference/src/mesh.jl:129-146
  Capacity(body, STmesh)                      /root/reference/src/prescribedmotionsolver/diffusion.jl:251-252
  psip_* / psim_*                             .../diffusion.jl:55-98   (the live definitions; the block at :38-54 is a string)
  A_mono_unstead_diff_moving                  .../diffusion.jl:100-160
  b_mono_unstead_diff_moving                  .../diffusion.jl:163-225
  MovingDiffusionUnsteadyMono                 .../diffusion.jl:16-35
  solve_MovingDiffusionUnsteadyMono!          .../diffusion.jl:227-268
  MovingDiffusionUnsteadyDiph                 .../diffusion.jl:272-290
  A_diph_unstead_diff_moving                  .../diffusion.jl:292-398
  b_diph_unstead_diff_moving                  .../diffusion.jl:400-498
  solve_MovingDiffusionUnsteadyDiph!          .../diffusion.jl:501-535

Parity unpinned for the capacities: the reference gets them from libvofi on the (N+1)-D cells (absent here);
`make_spacetime_capacity` restates their DEFINITION (time integrals of the spatial measures of `oracle/geometry.py`) with
its own, finer time rule.  The algebra (the blocks, the `[1:end÷2]` selections, Ψ, the border rows, the time loop) is
restated literally on the full (N+1)-D Kronecker operators of `penguin_oracle.make_diffusion_ops`.

The reference's names are kept although they read backwards: `Vn_1 = A[N+1][1:end÷2]` is the time-face capacity at the
FIRST time index (the lower face, V(t)), `Vn` the one at the second (V(t+Δt)).
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import geometry as og
from . import penguin_oracle as po


# ---------------------------------------------------------------------------------------------------------------
# moving bodies: a ball |x - c(t)| - r(t) (complement: the reference's  -(|x - c| - r(t)),  examples/2D/SolidMoving/
# MovingHeat.jl:19) and an axis-aligned half space  sign (x_axis - s(t))  (examples/1D/SolidMoving/MovingHeat.jl:18)
# ---------------------------------------------------------------------------------------------------------------
def _ddt(fn: Callable, t: float, h: float):
    return (np.asarray(fn(t + h), dtype=np.float64) - np.asarray(fn(t - h), dtype=np.float64)) / (2.0 * h)


class MovingBall:
    def __init__(self, center: Callable, radius: Callable, complement: bool = False, dcenter: Optional[Callable] = None,
                 dradius: Optional[Callable] = None):
        self.center, self.radius, self.complement = center, radius, bool(complement)
        self.dcenter, self.dradius = dcenter, dradius

    def at(self, t: float) -> og.Ball:
        return og.Ball(tuple(float(v) for v in np.atleast_1d(self.center(t))), float(self.radius(t)), self.complement)

    def normal_speed(self, t: float, cg: Sequence[float], h: float) -> float:
        c = np.atleast_1d(np.asarray(self.center(t), dtype=np.float64))
        dc = np.atleast_1d(self.dcenter(t) if self.dcenter else _ddt(self.center, t, h))
        dr = float(self.dradius(t) if self.dradius else _ddt(self.radius, t, h))
        e = np.asarray(cg, dtype=np.float64)[: len(c)] - c
        nn = float(np.dot(e, e))
        return dr + (float(np.dot(e, dc)) / math.sqrt(nn) if nn > 0.0 else 0.0)


class MovingHalfSpace:
    def __init__(self, axis: int, position: Callable, sign: float = 1.0, complement: bool = False, N: int = 1,
                 dposition: Optional[Callable] = None):
        self.axis, self.position, self.sign, self.complement, self.N = int(axis), position, sign, bool(complement), N
        self.dposition = dposition

    def at(self, t: float) -> og.HalfSpace:
        return og.HalfSpace(self.axis, float(self.position(t)), self.sign, self.complement, self.N)

    def normal_speed(self, t: float, cg, h: float) -> float:
        return float(self.dposition(t) if self.dposition else _ddt(self.position, t, h))


def composite_gauss(t0: float, t1: float, panels: int, order: int) -> Tuple[np.ndarray, np.ndarray]:
    """nodes / weights of `panels` Gauss-Legendre rules of `order` points on [t0, t1]."""
    x, w = np.polynomial.legendre.leggauss(order)
    edges = np.linspace(t0, t1, panels + 1)
    tau, wt = [], []
    for a, b in zip(edges[:-1], edges[1:]):
        tau.append(0.5 * (a + b) + 0.5 * (b - a) * x)
        wt.append(0.5 * (b - a) * w)
    return np.concatenate(tau), np.concatenate(wt)


def SpaceTimeMesh(mesh: po.Mesh, time: Sequence[float]) -> po.Mesh:
    """mesh.jl:135-145: nodes = (space nodes..., time), centres = (space centres..., midpoints).  One time cell here
    (every call site of the path passes [t, t+Δt])."""
    assert len(time) == 2
    t0, t1 = float(time[0]), float(time[1])
    st = po.Mesh(tuple(mesh.dims) + (1,), tuple(float(mesh.nodes[d][-1] - mesh.nodes[d][0]) for d in range(mesh.N)) + (t1 - t0,),
                 tuple(float(mesh.nodes[d][0]) for d in range(mesh.N)) + (t0,))
    # keep the space nodes / centres bit for bit, and the time nodes exactly as given
    st.nodes = tuple(mesh.nodes) + (np.array([t0, t1]),)
    st.centers = tuple(mesh.centers) + (np.array([(t1 + t0) / 2]),)
    return st


def make_spacetime_capacity(body, mesh: po.Mesh, t0: float, t1: float, panels: int = 64, order: int = 4,
                            compute_centroids: bool = True) -> po.Capacity:
    """The (N+1)-D capacity of the slab [t0, t1] on the padded grid (n_1+1, .., n_N+1, 2), dim 1 fastest, time slowest.
    First time layer: time integrals of the spatial measures (the padding conventions of po.make_capacity in space).
    Second layer (the time padding): only A_(N+1) = V(t1) is non-zero.  B_(N+1), W_(N+1) are left zero: no block of the
    moving solver reads them (the `[1:end÷2]` selections keep the space directions of the first layer only)."""
    N = mesh.N
    assert N in (1, 2)
    n, ext, nodes = mesh.dims, mesh.ext, mesh.nodes
    M = int(np.prod(ext))
    tau, wt = composite_gauss(t0, t1, panels, order)
    bodies = [body.at(float(t)) for t in tau]
    b0, b1 = body.at(t0), body.at(t1)
    hdiff = 1e-6 * (t1 - t0)
    st = SpaceTimeMesh(mesh, [t0, t1])
    M2 = 2 * M
    V, G, ct = np.zeros(M2), np.zeros(M2), np.zeros(M2)
    C_w, C_g = np.zeros((M2, N + 1)), np.zeros((M2, N + 1))
    A = tuple(np.zeros(M2) for _ in range(N + 1))
    B = tuple(np.zeros(M2) for _ in range(N + 1))
    W = tuple(np.zeros(M2) for _ in range(N + 1))

    def cells(rng=None):
        rng = rng or [range(1, n[d] + 1) for d in range(N)]
        import itertools
        for rev in itertools.product(*reversed(rng)):
            yield tuple(reversed(rev))

    dt = t1 - t0
    for I in cells():
        li = po.lin_index(ext, I)
        lo = [float(nodes[d][I[d] - 1]) for d in range(N)]
        hi = [float(nodes[d][I[d]]) for d in range(N)]
        m0, m1 = b0.box(lo, hi, want_surface=False), b1.box(lo, hi, want_surface=False)
        types = {m0.type, m1.type}
        v = gam = momt = gmt = 0.0
        mom, gm = np.zeros(N), np.zeros(N)
        for k, bk in enumerate(bodies):
            m = bk.box(lo, hi)
            types.add(m.type)
            v += wt[k] * m.vol
            momt += wt[k] * m.vol * tau[k]
            mom += wt[k] * m.vol * np.asarray(m.centroid)
            if m.gamma > 0.0:
                vn = body.normal_speed(float(tau[k]), m.cgamma, hdiff)
                ws = wt[k] * m.gamma * math.sqrt(1.0 + vn * vn)
                gam += ws
                gmt += ws * tau[k]
                gm += ws * np.asarray(m.cgamma)
        ctr = [0.5 * (lo[d] + hi[d]) for d in range(N)]
        if types == {og.FULL}:
            vol = float(np.prod([hi[d] - lo[d] for d in range(N)]))
            V[li], A[N][li], A[N][M + li], ct[li] = vol * dt, vol, vol, og.FULL
            C_w[li, :N], C_w[li, N] = ctr, 0.5 * (t0 + t1)
        elif types == {og.EMPTY}:
            ct[li] = og.EMPTY
            C_w[li, :N], C_w[li, N] = ctr, 0.5 * (t0 + t1)
        else:
            ct[li] = og.CUT
            V[li], A[N][li], A[N][M + li], G[li] = v, m0.vol, m1.vol, gam
            C_w[li, :N], C_w[li, N] = (mom / v, momt / v) if v > 0.0 else (ctr, 0.5 * (t0 + t1))
            if gam > 0.0:
                C_g[li, :N], C_g[li, N] = gm / gam, gmt / gam

    for d in range(N):
        rng = [range(1, (n[k] + 2) if k == d else (n[k] + 1)) for k in range(N)]
        for I in cells(rng):
            li = po.lin_index(ext, I)
            lo = [float(nodes[k][min(I[k], n[k]) - 1]) for k in range(N)]
            hi = [float(nodes[k][min(I[k], n[k])]) for k in range(N)]
            s = float(nodes[d][I[d] - 1])
            A[d][li] = sum(wt[k] * bodies[k].section(d, s, lo, hi) for k in range(len(bodies)))
    for I in cells():
        li = po.lin_index(ext, I)
        lo = [float(nodes[d][I[d] - 1]) for d in range(N)]
        hi = [float(nodes[d][I[d]]) for d in range(N)]
        for d in range(N):
            B[d][li] = sum(wt[k] * bodies[k].section(d, float(C_w[li, d]), lo, hi) for k in range(len(bodies)))
    for d in range(N):
        rng = [range(1, (n[k] + 2) if k == d else (n[k] + 1)) for k in range(N)]
        for I in cells(rng):
            li = po.lin_index(ext, I)
            prev_i, next_i = max(I[d] - 1, 1), min(I[d], n[d])            # capacity.jl:401-402
            lp = po.lin_index(ext, tuple(prev_i if k == d else I[k] for k in range(N)))
            ln = po.lin_index(ext, tuple(next_i if k == d else I[k] for k in range(N)))
            lo = [float(C_w[lp, k]) if k == d else float(nodes[k][I[k] - 1]) for k in range(N)]
            hi = [float(C_w[ln, k]) if k == d else float(nodes[k][I[k]]) for k in range(N)]
            if ct[lp] == og.EMPTY and ct[ln] == og.EMPTY:
                continue
            W[d][li] = sum(wt[k] * bodies[k].box(lo, hi, want_surface=False).vol for k in range(len(bodies)))
    if not compute_centroids:
        C_g = np.zeros((0, N + 1))
    return po.Capacity(A, B, V, W, C_w, C_g, G, ct, st, body)


# ---------------------------------------------------------------------------------------------------------------
# Ψ                                                                                      diffusion.jl:55-98
# ---------------------------------------------------------------------------------------------------------------
def psip_cn(a, b):
    if a == 0 and b == 0:
        return 0.0
    if a != 0 and b != 0:
        return 0.5
    if a == 0 and b != 0:
        return 0.5
    return 1.0


def psim_cn(a, b):
    if a == 0 and b == 0:
        return 0.0
    if a != 0 and b != 0:
        return 0.5
    if a == 0 and b != 0:      # "Fresh"
        return 0.5
    return 0.0                 # "Dead"


def psip_be(a, b):
    return 0.0 if (a == 0 and b == 0) else 1.0


def psim_be(a, b):
    return 0.0


def _half(x):
    """`x[1:end÷2, 1:end÷2]` of a matrix, `x[1:end÷2]` of a vector."""
    if sp.issparse(x):
        r, c = x.shape
        return x.tocsr()[: r // 2, : c // 2]
    return x[: len(x) // 2]


def A_mono_unstead_diff_moving(op: po.DiffusionOps, cap: po.Capacity, D, bc, scheme: str) -> sp.csr_matrix:
    """diffusion.jl:100-160."""
    cap_index = len(op.size) - 1                                   # :108 (0-based here)
    At = cap.A[cap_index]
    Vn_1, Vn = At[: len(At) // 2], At[len(At) // 2:]               # :113-114
    psip = psip_cn if scheme == "CN" else psip_be
    Psi = sp.diags(np.array([psip(a, b) for a, b in zip(Vn, Vn_1)]))   # :122
    Ia, Ib = po.build_I_bc(op, bc)
    Id_full = sp.diags(po.build_I_D(op, D, cap))
    Wi, G, H = _half(op.Winv), _half(op.G), _half(op.H)            # :145-147
    Ig, Id = _half(sp.diags(cap.G)), _half(Id_full)
    GT, HT = G.T.tocsr(), H.T.tocsr()
    b1 = sp.diags(Vn_1) + Id @ GT @ Wi @ G @ Psi                   # :154
    b2 = -(sp.diags(Vn_1) - sp.diags(Vn)) + Id @ GT @ Wi @ H @ Psi   # :155
    b3 = Ib * (HT @ Wi @ G)                                        # :156
    b4 = Ib * (HT @ Wi @ H) + Ia * Ig                              # :157
    return sp.bmat([[b1, b2], [b3, b4]], format="csr")


def b_mono_unstead_diff_moving(op, cap, D, f, bc, Ti, dt, t, scheme) -> np.ndarray:
    """diffusion.jl:163-225."""
    cap_index = len(op.size) - 1
    fn = po.build_source(op, f, t, cap)                            # :170-171: f(C_ω..., t): C_ω has N+1 components
    fn1 = po.build_source(op, f, t + dt, cap)
    gg = po.build_g_g(op, bc, cap)                                 # :172 (no time argument: value(C_γ...))
    Id_full = sp.diags(po.build_I_D(op, D, cap))
    At = cap.A[cap_index]
    Vn_1, Vn = At[: len(At) // 2], At[len(At) // 2:]
    psim = psim_cn if scheme == "CN" else psim_be
    Psi = sp.diags(np.array([psim(a, b) for a, b in zip(Vn, Vn_1)]))   # :185
    Wi, G, H, V = _half(op.Winv), _half(op.G), _half(op.H), _half(op.V)
    Ig, Id = _half(sp.diags(cap.G)), _half(Id_full)
    To, Tg = Ti[: len(Ti) // 2], Ti[len(Ti) // 2:]
    fn, fn1, gg = _half(fn), _half(fn1), _half(gg)
    GT = G.T.tocsr()
    if scheme == "CN":                                             # :214
        b1 = (sp.diags(Vn) - Id @ GT @ Wi @ G @ Psi) @ To - 0.5 * (Id @ GT @ Wi @ H @ Tg) + 0.5 * (V @ (fn + fn1))
    else:                                                          # :216
        b1 = Vn * To + V @ fn1
    b2 = Ig @ gg                                                   # :218
    return np.concatenate([b1, b2])


def MovingDiffusionUnsteadyMono(phase: po.Phase, bc_b, bc_i, dt: float, Ti: np.ndarray, mesh: po.Mesh, scheme: str) -> po.Solver:
    """diffusion.jl:16-35 (t = 0.0 in b and in the border rows)."""
    s = po.Solver("Unsteady", "Monophasic", "Diffusion")
    sch = "CN" if scheme == "CN" else "BE"
    s.A = A_mono_unstead_diff_moving(phase.operator, phase.capacity, phase.Diffusion_coeff, bc_i, sch)
    s.b = b_mono_unstead_diff_moving(phase.operator, phase.capacity, phase.Diffusion_coeff, phase.source, bc_i, Ti, dt, 0.0, sch)
    s.A, s.b = po.BC_border_mono(s.A, s.b, bc_b, mesh, t=0.0)
    return s


def solve_MovingDiffusionUnsteadyMono(s: po.Solver, phase: po.Phase, body, dt: float, Ts: float, Te: float, bc_b, bc,
                                      mesh: po.Mesh, scheme: str, method: str = "\\", capacity_fn: Optional[Callable] = None,
                                      max_steps: Optional[int] = None, **kwargs):
    """diffusion.jl:227-268.  `capacity_fn(t0, t1)` replaces `Capacity(body, STmesh)` (default: make_spacetime_capacity);
    the parity tests pass the capacities the HIP path computed, so that the algebra is compared on identical inputs."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")
    t = Ts
    po.solve_system(s, method=method, **kwargs)                    # :240
    s.states.append(s.x)
    Ti = s.x
    steps = 0
    while t < Te:                                                  # :247
        if max_steps is not None and steps >= max_steps:
            break
        t += dt
        cap = capacity_fn(t, t + dt) if capacity_fn else make_spacetime_capacity(body, mesh, t, t + dt)   # :251-252
        op = po.make_diffusion_ops(cap)
        s.A = A_mono_unstead_diff_moving(op, cap, phase.Diffusion_coeff, bc, scheme)
        s.b = b_mono_unstead_diff_moving(op, cap, phase.Diffusion_coeff, phase.source, bc, Ti, dt, t, scheme)
        s.A, s.b = po.BC_border_mono(s.A, s.b, bc_b, mesh, t=t)   # :258
        po.solve_system(s, method=method, **kwargs)
        s.states.append(s.x)
        Ti = s.x
        steps += 1
    return s


# ---------------------------------------------------------------------------------------------------------------
# two phases                                                                             diffusion.jl:272-535
# ---------------------------------------------------------------------------------------------------------------
def _time_faces(op: po.DiffusionOps, cap: po.Capacity):
    At = cap.A[len(op.size) - 1]                                   # capacite.A[cap_index] (:337-340)
    return At[: len(At) // 2], At[len(At) // 2:]                   # Vn_1, Vn


def A_diph_unstead_diff_moving(op1, op2, cap1, cap2, D1, D2, ic: po.InterfaceConditions, scheme: str) -> sp.csr_matrix:
    """diffusion.jl:292-398."""
    jump, flux = ic.scalar, ic.flux
    Vn1_1, Vn1 = _time_faces(op1, cap1)
    Vn2_1, Vn2 = _time_faces(op2, cap2)
    psip = psip_cn if scheme == "CN" else psip_be
    Psi1 = sp.diags(np.array([psip(a, b) for a, b in zip(Vn1, Vn1_1)]))     # :350
    Psi2 = sp.diags(np.array([psip(a, b) for a, b in zip(Vn2, Vn2_1)]))
    n = len(Vn1)
    Ia1, Ia2 = jump.alpha1 * sp.identity(n), jump.alpha2 * sp.identity(n)
    Ib1, Ib2 = flux.beta1, flux.beta2
    W1, G1, H1 = _half(op1.Winv), _half(op1.G), _half(op1.H)                 # :358-364
    W2, G2, H2 = _half(op2.Winv), _half(op2.G), _half(op2.H)
    Id1, Id2 = _half(sp.diags(po.build_I_D(op1, D1, cap1))), _half(sp.diags(po.build_I_D(op2, D2, cap2)))
    G1T, H1T, G2T, H2T = G1.T.tocsr(), H1.T.tocsr(), G2.T.tocsr(), H2.T.tocsr()
    dV1, dV2 = sp.diags(Vn1_1) - sp.diags(Vn1), sp.diags(Vn2_1) - sp.diags(Vn2)
    block1 = sp.diags(Vn1_1) + Id1 @ G1T @ W1 @ G1 @ Psi1                    # :374
    block2 = -dV1 + Id1 @ G1T @ W1 @ H1 @ Psi1
    block3 = sp.diags(Vn2_1) + Id2 @ G2T @ W2 @ G2 @ Psi2
    block4 = -dV2 + Id2 @ G2T @ W2 @ H2 @ Psi2
    block5 = Ib1 * (H1T @ W1 @ G1 @ Psi1)                                    # :379
    block6 = Ib1 * (H1T @ W1 @ H1 @ Psi1) - dV1
    block7 = Ib2 * (H2T @ W2 @ G2 @ Psi2)
    block8 = Ib2 * (H2T @ W2 @ H2 @ Psi2) - dV2
    Z = sp.csr_matrix((n, n))
    return sp.bmat([[block1, block2, Z, Z], [Z, Ia1, Z, -Ia2], [Z, Z, block3, block4], [block5, block6, block7, block8]], format="csr")


def b_diph_unstead_diff_moving(op1, op2, cap1, cap2, D1, D2, f1, f2, ic: po.InterfaceConditions, Ti, dt, t, scheme) -> np.ndarray:
    """diffusion.jl:400-498."""
    jump, flux = ic.scalar, ic.flux
    f1n, f1n1 = po.build_source(op1, f1, t, cap1), po.build_source(op1, f1, t + dt, cap1)     # :415-418
    f2n, f2n1 = po.build_source(op2, f2, t, cap2), po.build_source(op2, f2, t + dt, cap2)
    gg = po.build_g_g(op1, jump, cap1)                                                          # :423-424 (no time argument)
    hh = po.build_g_g(op2, flux, cap2)
    Vn1_1, Vn1 = _time_faces(op1, cap1)
    Vn2_1, Vn2 = _time_faces(op2, cap2)
    psim = psim_cn if scheme == "CN" else psim_be
    Psi1 = sp.diags(np.array([psim(a, b) for a, b in zip(Vn1, Vn1_1)]))                         # :439-440
    Psi2 = sp.diags(np.array([psim(a, b) for a, b in zip(Vn2, Vn2_1)]))
    q = len(Ti) // 4
    To1, Tg1, To2, Tg2 = Ti[:q], Ti[q:2 * q], Ti[2 * q:3 * q], Ti[3 * q:]                       # :459-463
    f1n, f1n1, f2n, f2n1 = _half(f1n), _half(f1n1), _half(f2n), _half(f2n1)
    gg, hh = _half(gg), _half(hh)
    Ig2 = _half(sp.diags(cap2.G))
    Id1, Id2 = _half(sp.diags(po.build_I_D(op1, D1, cap1))), _half(sp.diags(po.build_I_D(op2, D2, cap2)))
    W1, G1, H1, V1 = _half(op1.Winv), _half(op1.G), _half(op1.H), _half(op1.V)
    W2, G2, H2, V2 = _half(op2.Winv), _half(op2.G), _half(op2.H), _half(op2.V)
    G1T, G2T = G1.T.tocsr(), G2.T.tocsr()
    if scheme == "CN":                                                                          # :487-488
        b1 = (sp.diags(Vn1) - Id1 @ G1T @ W1 @ G1 @ Psi1) @ To1 - Id1 @ G1T @ W1 @ H1 @ Psi1 @ Tg1 + 0.5 * (V1 @ (f1n + f1n1))
        b3 = (sp.diags(Vn2) - Id2 @ G2T @ W2 @ G2 @ Psi2) @ To2 - Id2 @ G2T @ W2 @ H2 @ Psi2 @ Tg2 + 0.5 * (V2 @ (f2n + f2n1))
    else:                                                                                       # :490-491
        b1 = (sp.diags(Vn1) - Id1 @ G1T @ W1 @ G1 @ Psi1) @ To1 - Id1 @ G1T @ W1 @ H1 @ Psi1 @ Tg1 + V1 @ f1n1
        b3 = (sp.diags(Vn2) - Id2 @ G2T @ W2 @ G2 @ Psi2) @ To2 - Id2 @ G2T @ W2 @ H2 @ Psi2 @ Tg2 + V2 @ f2n1
    b2 = gg                                                                                     # :495
    b4 = Ig2 @ hh                                                                               # :496
    return np.concatenate([b1, b2, b3, b4])


def _border_diph(A, b, bc_b, cap1, cap2, mesh: po.Mesh, t):
    """BC_border_diph!(s.A, s.b, bc_b, mesh) (:288, :523): the method WITHOUT capacities (src/solver.jl:540-543), so both
    phases get their border rows whatever the cell types say (the static drivers call the capacity-aware method)."""
    ps = A.shape[0] // 4
    return po._apply_border(A, b, bc_b, mesh, t, offsets=(0, 2 * ps), skip=None)


def MovingDiffusionUnsteadyDiph(phase1: po.Phase, phase2: po.Phase, bc_b, ic, dt: float, Ti: np.ndarray, mesh: po.Mesh,
                                scheme: str) -> po.Solver:
    """diffusion.jl:272-290 (t = 0.0 in b)."""
    s = po.Solver("Unsteady", "Diphasic", "Diffusion")
    sch = "CN" if scheme == "CN" else "BE"
    s.A = A_diph_unstead_diff_moving(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity, phase1.Diffusion_coeff,
                                     phase2.Diffusion_coeff, ic, sch)
    s.b = b_diph_unstead_diff_moving(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity, phase1.Diffusion_coeff,
                                     phase2.Diffusion_coeff, phase1.source, phase2.source, ic, Ti, dt, 0.0, sch)
    s.A, s.b = _border_diph(s.A, s.b, bc_b, phase1.capacity, phase2.capacity, mesh, None)
    return s


def solve_MovingDiffusionUnsteadyDiph(s: po.Solver, phase1: po.Phase, phase2: po.Phase, body, body_c, dt: float, Te: float, bc_b, ic,
                                      mesh: po.Mesh, scheme: str, method: str = "\\", capacity_fn: Optional[Callable] = None,
                                      max_steps: Optional[int] = None, **kwargs):
    """diffusion.jl:501-535 (the loop starts at t = 0.0, :513).  `capacity_fn(t0, t1)` -> (capacity1, capacity2) replaces the
    two `Capacity(body, STmesh)` calls (default: make_spacetime_capacity of body and body_c)."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")
    t = 0.0
    po.solve_system(s, method=method, **kwargs)
    s.states.append(s.x)
    Ti = s.x
    steps = 0
    while t < Te:
        if max_steps is not None and steps >= max_steps:
            break
        t += dt
        if capacity_fn:
            cap1, cap2 = capacity_fn(t, t + dt)
        else:
            cap1, cap2 = make_spacetime_capacity(body, mesh, t, t + dt), make_spacetime_capacity(body_c, mesh, t, t + dt)
        op1, op2 = po.make_diffusion_ops(cap1), po.make_diffusion_ops(cap2)
        s.A = A_diph_unstead_diff_moving(op1, op2, cap1, cap2, phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, scheme)
        s.b = b_diph_unstead_diff_moving(op1, op2, cap1, cap2, phase1.Diffusion_coeff, phase2.Diffusion_coeff, phase1.source,
                                         phase2.source, ic, Ti, dt, t, scheme)
        s.A, s.b = _border_diph(s.A, s.b, bc_b, cap1, cap2, mesh, None)
        po.solve_system(s, method=method, **kwargs)
        s.states.append(s.x)
        Ti = s.x
        steps += 1
    return s


def spatial_layer(cap: po.Capacity, mesh: po.Mesh) -> po.Capacity:
    """The first time layer of a space-time capacity as an N-D capacity on `mesh` (what the HIP path stores)."""
    N = mesh.N
    M = int(np.prod(mesh.ext))
    cg = cap.C_g[:M, :N] if cap.C_g.shape[0] else cap.C_g
    return po.Capacity(tuple(a[:M] for a in cap.A[:N]), tuple(b[:M] for b in cap.B[:N]), cap.V[:M], tuple(w[:M] for w in cap.W[:N]),
                       cap.C_w[:M, :N], cg, cap.G[:M], cap.cell_types[:M], mesh, cap.body)
