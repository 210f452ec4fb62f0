"""ORACLE (test infrastructure) -- CPU restatement of Penguin.jl's hot path
Mesh -> Capacity -> DiffusionOps -> DiffusionUnsteadyMono/Diph -> solve_*!

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (penguin/jl_amd + libpenguin_hip.so) never does, and has no CPU
fallback.

Every function cites the reference lines it follows (paths relative to /root/reference).
The restatement is deliberately literal: operators are built with Kronecker products and
sparse-sparse products exactly as the reference does, rows are overwritten after
assembly, the zero rows/cols are eliminated with sum(abs(A)), and the time loop keeps all
of the reference's quirks.  Everything is float64 / int64.

Pinning status (see tests/test_oracle_*.py and DESIGN.md):
  * mesh vectors, border lists, operator sizes, constant-field gradient/divergence,
    cut-set == {Gamma>0}: pinned bit-exactly on the reference's own test values
    (test/mesh_test.jl, test/operators_test.jl, test/capacity_test.jl:255-257).
  * solver path: pinned on the reference's known answers (test/solver/diffusion_test.jl:57-80,
    test/convergence_test.jl:72-98, analytic Bessel series of examples/2D/Diffusion/Heat.jl).
  * per-cell capacity values: libvofi is un-vendored and Julia is absent => parity unpinned
    (oracle/geometry.py header).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .geometry import CUT, EMPTY, FULL, Ball, MultiBall

# =============================================================================
# Mesh                                                       src/mesh.jl:41-79
# =============================================================================


class Mesh:
    """src/mesh.jl:47-78.  centers_d[j] = x0 + j*(L/n), nodes_d[j] = x0 + (j+0.5)*(L/n)."""

    def __init__(self, n: Sequence[int], domain_size: Sequence[float], x0: Optional[Sequence[float]] = None):
        N = len(n)
        if x0 is None:
            x0 = tuple(0.0 for _ in range(N))
        self.N = N
        self.centers = tuple(
            np.array([x0[i] + j * (domain_size[i] / n[i]) for j in range(n[i])], dtype=np.float64) for i in range(N)
        )  # :49
        self.nodes = tuple(
            np.array([x0[i] + (j + 0.5) * (domain_size[i] / n[i]) for j in range(n[i] + 1)], dtype=np.float64)
            for i in range(N)
        )  # :50
        self.dims = tuple(len(c) for c in self.centers)  # :53
        self.border_cells = self._border_cells()  # :54-74

    def _border_cells(self):
        """:57-74  for d, for face in (1, dims[d]), Iterators.product (dim 1 fastest), unique!"""
        N, dims = self.N, self.dims
        out = []
        seen = set()
        for d in range(N):
            for face in (1, dims[d]):
                ranges = [range(1, dims[i] + 1) for i in range(N)]
                ranges[d] = range(face, face + 1)
                # Iterators.product: first iterator varies fastest
                for rev in itertools.product(*reversed(ranges)):
                    idx = tuple(reversed(rev))
                    if idx in seen:
                        continue
                    seen.add(idx)
                    pos = tuple(float(self.centers[i][idx[i] - 1]) for i in range(N))
                    out.append((idx, pos))
        return out

    def nC(self) -> int:  # :86
        return int(np.prod(self.dims))

    @property
    def ext(self) -> Tuple[int, ...]:
        """(n_d + 1): the padded node counts all fields live on."""
        return tuple(d + 1 for d in self.dims)


def lin_index(ext: Sequence[int], idx1: Sequence[int]) -> int:
    """0-based linear index of a 1-based Cartesian index on the padded grid (dim 1 fastest),
    src/solver.jl:362-372."""
    li = 0
    stride = 1
    for d, i in enumerate(idx1):
        li += (i - 1) * stride
        stride *= ext[d]
    return li


# =============================================================================
# Capacity                                         src/capacity.jl:25-36,81-123
# =============================================================================


@dataclass
class Capacity:
    """src/capacity.jl:25-36.  All diagonal matrices are stored as their diagonals
    (length M = prod(n_d+1)); C_w / C_g are (M,N) arrays."""

    A: Tuple[np.ndarray, ...]
    B: Tuple[np.ndarray, ...]
    V: np.ndarray
    W: Tuple[np.ndarray, ...]
    C_w: np.ndarray
    C_g: np.ndarray
    G: np.ndarray  # Gamma
    cell_types: np.ndarray
    mesh: Mesh
    body: object

    @property
    def N(self):
        return self.mesh.N


def make_capacity(body, mesh: Mesh, compute_centroids: bool = True) -> Capacity:
    """Restates VOFI() src/capacity.jl:81-123 with the capacity definitions of
    GeometricMoments (:264-430).  Padding layer (any i_d = n_d+1) is `zero` (:91) except
    A_d / W_d at i_d = n_d+1 (other indices real), which exist geometrically: A_d there is the
    upper face of the last cell (front_tracking.jl:925-934 loops i in 1:nx+1), W_d follows
    capacity.jl:399-403 (prev = next = n_d => zero width => 0)."""
    N = mesh.N
    n = mesh.dims
    ext = mesh.ext
    M = int(np.prod(ext))
    nodes = mesh.nodes
    V = np.zeros(M)
    G = np.zeros(M)
    ct = np.zeros(M)
    C_w = np.zeros((M, N))
    C_g = np.zeros((M, N))
    A = tuple(np.zeros(M) for _ in range(N))
    B = tuple(np.zeros(M) for _ in range(N))
    W = tuple(np.zeros(M) for _ in range(N))

    def cells():
        for rev in itertools.product(*[range(1, n[d] + 1) for d in reversed(range(N))]):
            yield tuple(reversed(rev))

    # ---- K1/K5: V, centroid, Gamma, type, C_gamma         capacity.jl:90-92,137-197
    for I in cells():
        li = lin_index(ext, I)
        lo = [float(nodes[d][I[d] - 1]) for d in range(N)]
        hi = [float(nodes[d][I[d]]) for d in range(N)]
        m = body.box(lo, hi)
        V[li] = m.vol
        ct[li] = m.type
        C_w[li, :] = m.centroid
        G[li] = m.gamma
        C_g[li, :] = m.cgamma

    # ---- K2: A_d at x_d = nodes_d[i_d], i_d = 1..n_d+1       capacity.jl:103, :355-371
    for d in range(N):
        rng = [range(1, (n[k] + 2) if k == d else (n[k] + 1)) for k in range(N)]
        for rev in itertools.product(*reversed(rng)):
            I = tuple(reversed(rev))
            li = lin_index(ext, I)
            lo = [float(nodes[k][min(I[k], n[k]) - 1]) for k in range(N)]
            hi = [float(nodes[k][min(I[k], n[k])]) for k in range(N)]
            A[d][li] = body.section(d, float(nodes[d][I[d] - 1]), lo, hi)

    # ---- K4: B_d = section through the cell centroid          capacity.jl:105, :373-391
    for I in cells():
        li = lin_index(ext, I)
        lo = [float(nodes[d][I[d] - 1]) for d in range(N)]
        hi = [float(nodes[d][I[d]]) for d in range(N)]
        for d in range(N):
            B[d][li] = body.section(d, float(C_w[li, d]), lo, hi)

    # ---- K3: W_d between centroids of i-e_d and i            capacity.jl:104, :396-429
    for d in range(N):
        rng = [range(1, (n[k] + 2) if k == d else (n[k] + 1)) for k in range(N)]
        for rev in itertools.product(*reversed(rng)):
            I = tuple(reversed(rev))
            li = lin_index(ext, I)
            prev_i = max(I[d] - 1, 1)  # :401
            next_i = min(I[d], n[d])  # :402
            Ip = tuple(prev_i if k == d else I[k] for k in range(N))
            In = tuple(next_i if k == d else I[k] for k in range(N))
            lp = lin_index(ext, Ip)
            ln = lin_index(ext, In)
            lo = [float(C_w[lp, k]) if k == d else float(nodes[k][I[k] - 1]) for k in range(N)]
            hi = [float(C_w[ln, k]) if k == d else float(nodes[k][I[k]]) for k in range(N)]
            tp, tn = ct[lp], ct[ln]
            if tp == EMPTY and tn == EMPTY:
                W[d][li] = 0.0  # :424-426
            else:
                W[d][li] = body.box(lo, hi, want_surface=False).vol

    if not compute_centroids:
        C_g = np.zeros((0, N))  # capacity.jl:119
    return Capacity(A, B, V, W, C_w, C_g, G, ct, mesh, body)


# =============================================================================
# Operators                                              src/operators.jl:9-178
# =============================================================================


def delta_m(n: int) -> sp.csr_matrix:
    """ẟ_m, src/operators.jl:9: backward difference, D[n,n] = 0."""
    main = np.ones(n)
    main[-1] = 0.0
    return sp.diags([main, -np.ones(n - 1)], [0, -1], shape=(n, n), format="csr")


def delta_p(n: int) -> sp.csr_matrix:
    """δ_p, src/operators.jl:10."""
    main = -np.ones(n)
    main[-1] = 0.0
    return sp.diags([main, np.ones(n - 1)], [0, 1], shape=(n, n), format="csr")


def sigma_m(n: int) -> sp.csr_matrix:
    """Σ_m, src/operators.jl:11."""
    main = 0.5 * np.ones(n)
    main[-1] = 0.0
    return sp.diags([main, 0.5 * np.ones(n - 1)], [0, -1], shape=(n, n), format="csr")


def sigma_p(n: int) -> sp.csr_matrix:
    """Σ_p, src/operators.jl:12."""
    main = 0.5 * np.ones(n)
    main[-1] = 0.0
    return sp.diags([main, 0.5 * np.ones(n - 1)], [0, 1], shape=(n, n), format="csr")


def build_differential_operator(op_fn, mesh: Mesh, dim: int) -> sp.csr_matrix:
    """src/operators.jl:92-113: kron(op[N], ..., op[1]) -- dim 1 fastest.  `dim` is 0-based."""
    N = mesh.N
    counts = mesh.ext
    if N == 1:
        return op_fn(counts[0])
    ops = [op_fn(counts[i]) if i == dim else sp.identity(counts[i], format="csr") for i in range(N)]
    res = ops[N - 1]
    for i in range(N - 2, -1, -1):
        res = sp.kron(res, ops[i], format="csr")
    return res.tocsr()


@dataclass
class DiffusionOps:
    """src/operators.jl:49-55."""

    G: sp.csr_matrix
    H: sp.csr_matrix
    Winv: sp.csr_matrix  # Wꜝ
    V: sp.csr_matrix
    size: Tuple[int, ...]


def make_diffusion_ops(cap: Capacity) -> DiffusionOps:
    """compute_base_operators + DiffusionOps, src/operators.jl:127-178."""
    mesh = cap.mesh
    N = mesh.N
    D_m = [build_differential_operator(delta_m, mesh, d) for d in range(N)]
    Gp = [D_m[d] @ sp.diags(cap.B[d]) for d in range(N)]  # :138
    G = sp.vstack(Gp, format="csr")
    Hp = [sp.diags(cap.A[d]) @ D_m[d] - D_m[d] @ sp.diags(cap.B[d]) for d in range(N)]  # :141
    H = sp.vstack(Hp, format="csr")
    diagW = np.concatenate([cap.W[d] for d in range(N)])  # :145-146
    winv = np.where(diagW != 0.0, 1.0 / np.where(diagW != 0.0, diagW, 1.0), 1.0)  # :149-151
    Winv = sp.diags(winv, format="csr")
    return DiffusionOps(G, H, Winv, sp.diags(cap.V, format="csr"), mesh.ext)


@dataclass
class ConvectionOps:
    """src/operators.jl:69-77."""

    C: Tuple[sp.csr_matrix, ...]
    K: Tuple[sp.csr_matrix, ...]
    G: sp.csr_matrix
    H: sp.csr_matrix
    Winv: sp.csr_matrix
    V: sp.csr_matrix
    size: Tuple[int, ...]


def make_convection_ops(cap: Capacity, u_omega: Sequence[np.ndarray], u_gamma: np.ndarray) -> ConvectionOps:
    """ConvectionOps(Capacity, uₒ, uᵧ), src/operators.jl:194-210:
    C_d = D_p[d] * spdiagm(S_m[d] * A[d] * uₒ[d]) * S_m[d],  K_d = spdiagm(S_p[d] * H' * uᵧ)."""
    mesh = cap.mesh
    N = mesh.N
    base = make_diffusion_ops(cap)
    D_p = [build_differential_operator(delta_p, mesh, d) for d in range(N)]
    S_m = [build_differential_operator(sigma_m, mesh, d) for d in range(N)]
    S_p = [build_differential_operator(sigma_p, mesh, d) for d in range(N)]
    C = tuple((D_p[d] @ sp.diags(S_m[d] @ (cap.A[d] * np.asarray(u_omega[d], dtype=float))) @ S_m[d]).tocsr() for d in range(N))
    h = base.H.T @ np.asarray(u_gamma, dtype=float)
    K = tuple(sp.diags(S_p[d] @ h, format="csr") for d in range(N))
    return ConvectionOps(C, K, base.G, base.H, base.Winv, base.V, mesh.ext)


def grad(op: DiffusionOps, p: np.ndarray) -> np.ndarray:
    """∇, src/operators.jl:20-23."""
    h = len(p) // 2
    return op.Winv @ (op.G @ p[:h] + op.H @ p[h:])


def div(op: DiffusionOps, qw: np.ndarray, qg: np.ndarray) -> np.ndarray:
    """∇₋, src/operators.jl:30-34."""
    GT = op.G.T
    HT = op.H.T
    return -((GT + HT) @ qw) + HT @ qg


# =============================================================================
# Boundary / interface conditions                          src/boundary.jl:12-137
# =============================================================================


@dataclass
class Dirichlet:
    value: Union[float, Callable]


@dataclass
class Neumann:
    value: Union[float, Callable]


@dataclass
class Robin:
    alpha: float
    beta: float
    value: Union[float, Callable]


@dataclass
class Periodic:
    pass


@dataclass
class ScalarJump:
    alpha1: float
    alpha2: float
    value: Union[float, Callable]


@dataclass
class FluxJump:
    beta1: float
    beta2: float
    value: Union[float, Callable]


@dataclass
class BorderConditions:
    borders: Dict[str, object]


@dataclass
class InterfaceConditions:
    scalar: ScalarJump
    flux: FluxJump


@dataclass
class Phase:
    """src/phase.jl:12-17."""

    capacity: Capacity
    operator: DiffusionOps
    source: Callable
    Diffusion_coeff: Callable


# =============================================================================
# Solver core                                               src/solver.jl:33-580
# =============================================================================


@dataclass
class Solver:
    """src/solver.jl:33-42."""

    time_type: str
    phase_type: str
    equation_type: str
    A: Optional[sp.csr_matrix] = None
    b: Optional[np.ndarray] = None
    x: Optional[np.ndarray] = None
    ch: list = field(default_factory=list)
    states: list = field(default_factory=list)
    # bookkeeping the reference does not keep (used by the parity tests only)
    last_idx: Optional[np.ndarray] = None
    last_A_reduced: Optional[sp.csr_matrix] = None
    last_b_reduced: Optional[np.ndarray] = None
    iters: list = field(default_factory=list)


def remove_zero_rows_cols(A: sp.csr_matrix, b: np.ndarray):
    """src/solver.jl:59-78."""
    absA = abs(A)
    row_sums = np.asarray(absA.sum(axis=1)).ravel()
    col_sums = np.asarray(absA.sum(axis=0)).ravel()
    rows_idx = np.flatnonzero(row_sums != 0.0)
    cols_idx = np.flatnonzero(col_sums != 0.0)
    common = np.intersect1d(rows_idx, cols_idx)
    Ar = A.tocsr()[common, :][:, common]
    return Ar.tocsr(), b[common], common


def bicgstab_ref(A: sp.csr_matrix, b: np.ndarray, reltol: float = 1e-12, abstol: float = 0.0,
                 maxiter: int = 10000, x0: Optional[np.ndarray] = None):
    """Unpreconditioned BiCGStab (van der Vorst 1992), zero initial guess as
    IterativeSolvers' default; convergence when ||r|| <= max(reltol*||b||, abstol), tested
    after the half step (s) and the full step (r).  This is the iteration the HIP driver
    runs (penguin/jl_amd/csrc/pg_krylov.hip) so iteration counts can be compared."""
    n = A.shape[0]
    x = np.zeros(n) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    rhat = r.copy()
    bnorm = np.linalg.norm(b)
    tol = max(reltol * bnorm, abstol)
    rnorm = np.linalg.norm(r)
    if rnorm <= tol:
        return x, 0, rnorm
    rho_old = alpha = omega = 1.0
    v = np.zeros(n)
    p = np.zeros(n)
    it = 0
    rho = rhat @ r
    rhat2 = rhat @ rhat
    restart = False
    while it < maxiter:
        it += 1
        if restart:
            p = r.copy()
            restart = False
        else:
            beta = (rho / rho_old) * (alpha / omega)
            p = r + beta * (p - omega * v)
        v = A @ p
        den = rhat @ v
        force = den == 0.0
        alpha = 0.0 if force else rho / den      # (r̂,Ap) == 0: minimal-residual half step, then restart
        s = r - alpha * v
        t = A @ s
        tt = t @ t
        omega = (t @ s) / tt if tt != 0.0 else 0.0
        x = x + alpha * p + omega * s
        r = s - omega * t
        rho_old = rho
        rho = rhat @ r
        rr = r @ r
        rnorm = math.sqrt(rr)
        if rnorm <= tol:
            break
        if omega == 0.0 or force or rho * rho < 1e-20 * rhat2 * rr:
            # (r̂,r) collapsed (cos < 1e-10): restart with r̂ := r -- same rule as pg_krylov.hip
            rhat = r.copy()
            rho = rhat2 = rr
            alpha = omega = 1.0
            restart = True
    return x, it, rnorm


def cg_ref(A: sp.csr_matrix, b: np.ndarray, reltol: float = 1e-12, abstol: float = 0.0, maxiter: int = 10000):
    """Plain CG (Hestenes-Stiefel), zero initial guess -- IterativeSolvers.cg semantics."""
    n = A.shape[0]
    x = np.zeros(n)
    r = b.copy()
    p = r.copy()
    rr = r @ r
    tol = max(reltol * math.sqrt(b @ b), abstol)
    it = 0
    if math.sqrt(rr) <= tol:
        return x, 0, math.sqrt(rr)
    while it < maxiter:
        it += 1
        q = A @ p
        alpha = rr / (p @ q)
        x = x + alpha * p
        r = r - alpha * q
        rr_new = r @ r
        if math.sqrt(rr_new) <= tol:
            rr = rr_new
            break
        p = r + (rr_new / rr) * p
        rr = rr_new
    return x, it, math.sqrt(rr)


def gmres_ref(A: sp.csr_matrix, b: np.ndarray, reltol: float = 1e-12, abstol: float = 0.0, restart: int = 20,
              maxiter: int = 10000, weights: Optional[np.ndarray] = None):
    """Restarted GMRES as IterativeSolvers 0.9.4 `gmres` runs it (the default `method` of solve_system!,
    src/solver.jl:158): zero initial guess, Arnoldi with modified Gram-Schmidt, Givens rotations, convergence on the
    rotated residual estimate |g_{j+1}| <= max(reltol*||r0||, abstol), true residual recomputed at every restart.
    IterativeSolvers is not vendored: this follows the published algorithm (Saad & Schultz 1986); the solution is
    checked against scipy's gmres and the direct solve in tests/test_oracle_pins.py.  pg_gmres.hip orthogonalises with
    classical Gram-Schmidt applied twice instead (one fused multi-dot per pass); counts agree to +-1.

    weights (the HIP path's acceptance rule, pg_gmres.hip): the Givens estimate only ends a cycle; the solve is accepted at a
    restart on the true residual in the weighted norm, ||w r|| <= max(reltol ||w b||, abstol), and a cycle that met the
    estimate but not the weighted test is followed by one with the inner tolerance lowered by the factor still missing
    (x 1/4).  The third return value is then the weighted residual norm."""
    n = A.shape[0]
    m = max(1, min(restart, n))
    x = np.zeros(n)
    r = b.copy()
    beta = np.linalg.norm(r)
    tol = max(reltol * beta, abstol)
    it = 0
    res = beta
    if weights is not None:
        tolw = max(reltol * np.linalg.norm(weights * b), abstol)
        resw = np.linalg.norm(weights * r)
        while it < maxiter and resw > tolw and beta > 0.0:
            V = np.zeros((m + 1, n))
            H = np.zeros((m + 1, m))
            cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m + 1)
            V[0] = r / beta
            g[0] = beta
            J = 0
            for j in range(m):
                if it >= maxiter:
                    break
                w = A @ V[j]
                for i in range(j + 1):
                    H[i, j] = V[i] @ w
                    w = w - H[i, j] * V[i]
                hn = np.linalg.norm(w)
                H[j + 1, j] = hn
                for i in range(j):
                    a, c = H[i, j], H[i + 1, j]
                    H[i, j], H[i + 1, j] = cs[i] * a + sn[i] * c, -sn[i] * a + cs[i] * c
                d = math.hypot(H[j, j], hn)
                it += 1
                if d == 0.0:
                    break
                cs[j], sn[j] = H[j, j] / d, hn / d
                H[j, j], H[j + 1, j] = d, 0.0
                g[j + 1] = -sn[j] * g[j]
                g[j] = cs[j] * g[j]
                J = j + 1
                if abs(g[j + 1]) <= tol or hn == 0.0:
                    break
                V[j + 1] = w / hn
            if J == 0:
                break
            y = np.linalg.solve(np.triu(H[:J, :J]), g[:J])
            x = x + V[:J].T @ y
            r = b - A @ x
            beta = np.linalg.norm(r)
            resw = np.linalg.norm(weights * r)
            if resw > tolw and resw > 0.0:
                tol = min(tol, 0.5 * beta * (tolw / resw))     # (0.25 on the squares, as the device code has it)
        return x, it, resw
    while it < maxiter:
        if res <= tol or beta == 0.0:
            break
        V = np.zeros((m + 1, n))
        H = np.zeros((m + 1, m))
        cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m + 1)
        V[0] = r / beta
        g[0] = beta
        J = 0
        for j in range(m):
            if it >= maxiter:
                break
            w = A @ V[j]
            for i in range(j + 1):
                H[i, j] = V[i] @ w
                w = w - H[i, j] * V[i]
            hn = np.linalg.norm(w)
            H[j + 1, j] = hn
            for i in range(j):
                a, c = H[i, j], H[i + 1, j]
                H[i, j], H[i + 1, j] = cs[i] * a + sn[i] * c, -sn[i] * a + cs[i] * c
            d = math.hypot(H[j, j], hn)
            it += 1
            if d == 0.0:
                break
            cs[j], sn[j] = H[j, j] / d, hn / d
            H[j, j], H[j + 1, j] = d, 0.0
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            J = j + 1
            res = abs(g[j + 1])
            if res <= tol or hn == 0.0:
                break
            V[j + 1] = w / hn
        if J > 0:
            y = np.linalg.solve(np.triu(H[:J, :J]), g[:J])
            x = x + V[:J].T @ y
        if res <= tol or J == 0:
            break
        r = b - A @ x
        beta = res = np.linalg.norm(r)
    return x, it, res


def solve_system(s: Solver, method: str = "\\", **kwargs):
    """src/solver.jl:158-188.  method "\\" = direct LU (UMFPACK in the reference, SuperLU
    here), "bicgstab" / "cg" / "gmres" = the Krylov restatements above."""
    n = s.A.shape[0]
    Ar, br, idx = remove_zero_rows_cols(s.A, s.b)  # :163
    if method == "\\":
        xr = spla.spsolve(Ar.tocsc(), br) if Ar.shape[0] > 0 else np.zeros(0)  # :171
        its = 0
    elif method == "bicgstab":
        xr, its, _ = bicgstab_ref(Ar, br, **kwargs)
    elif method == "cg":
        xr, its, _ = cg_ref(Ar, br, **kwargs)
    elif method == "gmres":
        xr, its, _ = gmres_ref(Ar, br, **kwargs)
    else:
        raise ValueError(f"unknown method {method}")
    s.x = np.zeros(n)  # :186
    s.x[idx] = xr  # :187
    s.last_idx, s.last_A_reduced, s.last_b_reduced = idx, Ar, br
    s.iters.append(its)


def build_I_bc(op: DiffusionOps, bc):
    """src/solver.jl:203-223 -> (Ia, Ib) as scalars multiplying I(n)."""
    if isinstance(bc, Dirichlet):
        return 1.0, 0.0
    if isinstance(bc, Neumann):
        return 0.0, 1.0
    if isinstance(bc, Robin):
        return float(bc.alpha), float(bc.beta)
    return 0.0, 0.0


def get_all_coordinates(C: np.ndarray):
    """src/solver.jl:230-248: pad to (x,y,z) with zeros."""
    n, N = C.shape
    out = np.zeros((n, 3))
    out[:, :N] = C
    return out


def build_I_D(op: DiffusionOps, D, cap: Capacity) -> np.ndarray:
    """src/solver.jl:255-266 -> diagonal of Id."""
    n = int(np.prod(op.size))
    if callable(D):
        co = get_all_coordinates(cap.C_w)
        return np.array([D(*c) for c in co], dtype=np.float64)
    return np.full(n, float(D))


def build_source(op: DiffusionOps, f: Callable, t: Optional[float], cap: Capacity) -> np.ndarray:
    """src/solver.jl:273-286."""
    co = get_all_coordinates(cap.C_w)
    if t is None:
        return np.array([f(*c) for c in co], dtype=np.float64)
    return np.array([f(*c, t) for c in co], dtype=np.float64)


def build_g_g(op: DiffusionOps, bc, cap: Capacity, t: Optional[float] = None) -> np.ndarray:
    """src/solver.jl:293-323: value(C_g..., t), falling back to value(C_g...)."""
    n = int(np.prod(op.size))
    if callable(bc.value):
        co = get_all_coordinates(cap.C_g)
        if t is not None:
            try:
                return np.array([bc.value(*c, t) for c in co], dtype=np.float64)
            except TypeError:
                pass
        return np.array([bc.value(*c) for c in co], dtype=np.float64)
    return np.full(n, float(bc.value))


def classify_boundary_cell_fast(ci: Sequence[int], mesh: Mesh) -> str:
    """src/solver.jl:379-409 -- NOTE the naming: left/right = dim 2, bottom/top = dim 1."""
    nd = mesh.N
    if nd >= 2:
        if ci[1] == 1:
            return "left"
        if ci[1] == mesh.dims[1]:
            return "right"
    if ci[0] == 1:
        return "bottom"
    if ci[0] == mesh.dims[0]:
        return "top"
    if nd >= 3:
        if ci[2] == 1:
            return "backward"
        if ci[2] == mesh.dims[2]:
            return "forward"
    raise RuntimeError(f"Cell {ci} is not on any boundary")


_OPPOSITE = {"left": "right", "right": "left", "bottom": "top", "top": "bottom",
             "backward": "forward", "forward": "backward"}


def eval_bc_value(value, pos, t):
    """src/solver.jl:441-448: border values get the N *unpadded* coordinates."""
    if callable(value):
        if t is None:
            return float(value(*pos))
        try:
            return float(value(*pos, t))
        except TypeError:
            return float(value(*pos))
    return float(value)


def find_corresponding_cell_optimized(li0: int, key: str, mesh: Mesh) -> int:
    """src/solver.jl:506-530 (0-based linear indices in and out)."""
    ext = mesh.ext
    ci = []
    rem = li0
    for d in range(mesh.N):
        ci.append(rem % ext[d] + 1)
        rem //= ext[d]
    if key == "left":
        new = (ci[0], ext[1])
    elif key == "right":
        new = (ci[0], 1)
    elif key == "bottom":
        new = (ext[0], ci[1])
    elif key == "top":
        new = (1, ci[1])
    elif key == "backward" and mesh.N >= 3:
        new = (ci[0], ci[1], ext[2])
    elif key == "forward" and mesh.N >= 3:
        new = (ci[0], ci[1], 1)
    else:
        raise RuntimeError(f"Unknown boundary key: {key}")
    if len(new) != mesh.N:
        raise RuntimeError("Periodic partner lookup builds 2-index CartesianIndex: 2-D only (solver.jl:512-519)")
    return lin_index(ext, new)


def _apply_border(A: sp.csr_matrix, b: np.ndarray, bc_b: BorderConditions, mesh: Mesh, t,
                  offsets: Sequence[int], skip=None):
    """Shared body of BC_border_mono!/BC_border_diph! (src/solver.jl:417-434,450-499,552-580).
    Row overwrite `A[row,:] .= 0; A[row,row] = 1; b[row] = value` is applied as one masked
    update (same values)."""
    n = A.shape[0]
    zero_rows = np.zeros(n, dtype=bool)
    add_r, add_c, add_v = [], [], []
    b = b.copy()
    for (ci, pos) in mesh.border_cells:
        key = classify_boundary_cell_fast(ci, mesh)
        cond = bc_b.borders.get(key)
        if cond is None:
            continue
        li = lin_index(mesh.ext, ci)
        for k, off in enumerate(offsets):
            if skip is not None and skip[k][li] == 0:
                continue  # solver.jl:574-575
            row = li + off
            if isinstance(cond, Dirichlet):  # :452-456
                zero_rows[row] = True
                add_r.append(row); add_c.append(row); add_v.append(1.0)
                b[row] = eval_bc_value(cond.value, pos, t)
            elif isinstance(cond, Periodic):  # :458-469
                if _OPPOSITE[key] in bc_b.borders:
                    cor = find_corresponding_cell_optimized(li, key, mesh) + off
                    zero_rows[row] = True
                    add_r += [row, row]; add_c += [row, cor]; add_v += [1.0, -1.0]
                    b[row] = 0.0
            elif isinstance(cond, Neumann):  # :471-496 (1-D only)
                if mesh.N == 1:
                    dx = float(np.min(np.diff(mesh.nodes[0])))
                    dims0 = mesh.ext[0]
                    li1 = li + 1
                    if key == "bottom":
                        adj = min(li1 + 1, dims0)
                    elif key == "top":
                        adj = max(li1 - 1, 1)
                    else:
                        adj = min(li1 + 1, dims0) if key == "left" else max(li1 - 1, 1)
                    zero_rows[row] = True
                    add_r += [row, row]; add_c += [row, adj - 1 + off]; add_v += [1.0 / dx, -1.0 / dx]
                    b[row] = eval_bc_value(cond.value, pos, t)
    keep = sp.diags((~zero_rows).astype(np.float64))
    A2 = (keep @ A).tocsr()
    if add_r:
        A2 = A2 + sp.csr_matrix((add_v, (add_r, add_c)), shape=A.shape)
    A2.eliminate_zeros()
    return A2.tocsr(), b


def BC_border_mono(A, b, bc_b: BorderConditions, mesh: Mesh, t=None):
    """src/solver.jl:417-434."""
    return _apply_border(A, b, bc_b, mesh, t, offsets=(0,))


def BC_border_diph(A, b, bc_b: BorderConditions, cap1: Capacity, cap2: Capacity, t=None):
    """src/solver.jl:545-580 (capacity-aware form used by the diffusion drivers)."""
    n4 = A.shape[0]
    ps = n4 // 4
    return _apply_border(A, b, bc_b, cap1.mesh, t, offsets=(0, 2 * ps),
                         skip=(cap1.cell_types, cap2.cell_types))


# =============================================================================
# Diffusion drivers                                  src/solver/diffusion.jl
# =============================================================================


def _blocks(op: DiffusionOps):
    GT, HT, W, G, H = op.G.T.tocsr(), op.H.T.tocsr(), op.Winv, op.G, op.H
    return GT @ W @ G, GT @ W @ H, HT @ W @ G, HT @ W @ H


def A_mono_unstead_diff(op: DiffusionOps, cap: Capacity, D, bc, dt: float, scheme: str) -> sp.csr_matrix:
    """src/solver/diffusion.jl:212-241."""
    n = int(np.prod(op.size))
    Ia, Ib = build_I_bc(op, bc)
    Ig = sp.diags(cap.G)
    Id = sp.diags(build_I_D(op, D, cap))
    L, Mx, P, Q = _blocks(op)
    if scheme == "CN":  # :224-227
        b1 = op.V + dt / 2 * (Id @ L)
        b2 = dt / 2 * (Id @ Mx)
        b3 = dt / 2 * Ib * P
        b4 = dt / 2 * Ib * Q + dt / 2 * (Ia * Ig)
    else:  # :229-232
        b1 = op.V + dt * (Id @ L)
        b2 = dt * (Id @ Mx)
        b3 = Ib * P
        b4 = Ib * Q + (Ia * Ig)
    return sp.bmat([[b1, b2], [b3, b4]], format="csr")


def b_mono_unstead_diff(op, f, D, cap, bc, Ti, dt, t, scheme) -> np.ndarray:
    """src/solver/diffusion.jl:243-265."""
    N = int(np.prod(op.size))
    Ig = cap.G
    fn = build_source(op, f, t, cap)
    fn1 = build_source(op, f, t + dt, cap)
    gn = build_g_g(op, bc, cap, t)
    gn1 = build_g_g(op, bc, cap, t + dt)
    Ia, Ib = build_I_bc(op, bc)
    Id = build_I_D(op, D, cap)
    Tw, Tg = Ti[:N], Ti[N:]
    V = cap.V
    if scheme == "CN":  # :257-258
        L, Mx, P, Q = _blocks(op)
        b1 = (V * Tw - dt / 2 * Id * (L @ Tw)) - dt / 2 * Id * (Mx @ Tg) + dt / 2 * V * (fn + fn1)
        b2 = dt / 2 * Ig * (gn + gn1) - dt / 2 * Ib * (P @ Tw) - dt / 2 * Ib * (Q @ Tg) - dt / 2 * Ia * Ig * Tg
    else:  # :260-261
        b1 = V * Tw + dt * V * fn1
        b2 = Ig * gn1
    return np.concatenate([b1, b2])


def DiffusionUnsteadyMono(phase: Phase, bc_b: BorderConditions, bc_i, dt: float, Ti: np.ndarray, scheme: str) -> Solver:
    """src/solver/diffusion.jl:192-210."""
    s = Solver("Unsteady", "Monophasic", "Diffusion")
    sch = "CN" if scheme == "CN" else "BE"
    s.A = A_mono_unstead_diff(phase.operator, phase.capacity, phase.Diffusion_coeff, bc_i, dt, sch)
    s.b = b_mono_unstead_diff(phase.operator, phase.source, phase.Diffusion_coeff, phase.capacity, bc_i, Ti, dt, 0.0, sch)
    s.A, s.b = BC_border_mono(s.A, s.b, bc_b, phase.capacity.mesh, t=0.0)  # :207
    return s


def solve_DiffusionUnsteadyMono(s: Solver, phase: Phase, dt: float, Tend: float, bc_b: BorderConditions, bc,
                                scheme: str, method: str = "\\", max_steps: Optional[int] = None, **kwargs):
    """src/solver/diffusion.jl:268-301, quirks (i)-(vi) of SURVEY.md a18 kept."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")  # :269-271
    t = 0.0
    solve_system(s, method=method, **kwargs)  # :275
    s.states.append(s.x)
    Ti = s.x
    s.A = A_mono_unstead_diff(phase.operator, phase.capacity, phase.Diffusion_coeff, bc, dt, scheme)  # :283
    steps = 0
    while t < Tend:  # :286
        if max_steps is not None and steps >= max_steps:
            break
        t += dt  # :287
        s.b = b_mono_unstead_diff(phase.operator, phase.source, phase.Diffusion_coeff, phase.capacity, bc, Ti, dt, t, scheme)
        s.A, s.b = BC_border_mono(s.A, s.b, bc_b, phase.capacity.mesh, t=t)  # :292
        solve_system(s, method=method, **kwargs)  # :294
        s.states.append(s.x)
        Ti = s.x
        steps += 1
    return s


# ---- steady mono (next row f.1; used here only to pin the oracle on the reference's
#      Poisson known answers, test/convergence_test.jl:30-70) --------------------------


def A_mono_stead_diff(op, cap, D, bc):
    """src/solver/diffusion.jl:30-43."""
    Ia, Ib = build_I_bc(op, bc)
    Ig = sp.diags(cap.G)
    Id = sp.diags(build_I_D(op, D, cap))
    L, Mx, P, Q = _blocks(op)
    return sp.bmat([[Id @ L, Id @ Mx], [Ib * P, Ib * Q + Ia * Ig]], format="csr")


def b_mono_stead_diff(op, f, cap, bc):
    """src/solver/diffusion.jl:45-58."""
    fo = build_source(op, f, None, cap)
    gg = build_g_g(op, bc, cap)
    return np.concatenate([cap.V * fo, cap.G * gg])


def DiffusionSteadyMono(phase: Phase, bc_b, bc_i) -> Solver:
    """src/solver/diffusion.jl:14-28."""
    s = Solver("Steady", "Monophasic", "Diffusion")
    s.A = A_mono_stead_diff(phase.operator, phase.capacity, phase.Diffusion_coeff, bc_i)
    s.b = b_mono_stead_diff(phase.operator, phase.source, phase.capacity, bc_i)
    s.A, s.b = BC_border_mono(s.A, s.b, bc_b, phase.capacity.mesh)
    return s


def solve_DiffusionSteadyMono(s: Solver, method: str = "\\", **kwargs):
    """src/solver/diffusion.jl:60-71."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")
    solve_system(s, method=method, **kwargs)
    return s


def A_diph_stead_diff(op1, op2, cap1, cap2, D1, D2, ic: InterfaceConditions):
    """src/solver/diffusion.jl:103-144."""
    n = int(np.prod(op1.size))
    jump, flux = ic.scalar, ic.flux
    I_n = sp.identity(n, format="csr")
    Id1 = sp.diags(build_I_D(op1, D1, cap1))
    Id2 = sp.diags(build_I_D(op2, D2, cap2))
    L1, M1, P1, Q1 = _blocks(op1)
    L2, M2, P2, Q2 = _blocks(op2)
    return sp.bmat(
        [[Id1 @ L1, Id1 @ M1, None, None],
         [None, jump.alpha1 * I_n, None, -jump.alpha2 * I_n],
         [None, None, Id2 @ L2, Id2 @ M2],
         [flux.beta1 * P1, flux.beta1 * Q1, flux.beta2 * P2, flux.beta2 * Q2]], format="csr")


def b_diph_stead_diff(op1, op2, f1, f2, cap1, cap2, ic):
    """src/solver/diffusion.jl:146-162."""
    gg = build_g_g(op1, ic.scalar, cap1)
    hh = build_g_g(op2, ic.flux, cap2)
    return np.concatenate([cap1.V * build_source(op1, f1, None, cap1), gg,
                           cap2.V * build_source(op2, f2, None, cap2), cap2.G * hh])


def DiffusionSteadyDiph(phase1: Phase, phase2: Phase, bc_b, ic) -> Solver:
    """src/solver/diffusion.jl:88-101."""
    s = Solver("Steady", "Diphasic", "Diffusion")
    s.A = A_diph_stead_diff(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity,
                            phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic)
    s.b = b_diph_stead_diff(phase1.operator, phase2.operator, phase1.source, phase2.source, phase1.capacity,
                            phase2.capacity, ic)
    s.A, s.b = BC_border_diph(s.A, s.b, bc_b, phase1.capacity, phase2.capacity)
    return s


def solve_DiffusionSteadyDiph(s: Solver, method: str = "\\", **kwargs):
    """src/solver/diffusion.jl:164-175."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")
    solve_system(s, method=method, **kwargs)
    return s


# ---- advection-diffusion, monophasic (next row f.2; src/solver/advectiondiffusion.jl) ----------------
# The reference has no test for these drivers, and the loop of its unsteady monophasic driver calls
# b_mono_unstead_advdiff with 8 arguments for a 9-parameter method (:272 vs :215): it raises a MethodError
# after the first solve.  Restated here is the evident intent (the same loop as solve_DiffusionUnsteadyMono!
# with the advection-diffusion blocks); parity for this row is UNPINNED.


def _conv(op: ConvectionOps):
    n = int(np.prod(op.size))
    conv_bulk = sum(op.C[1:], op.C[0])
    conv_iface = 0.5 * sum(op.K[1:], op.K[0])
    return conv_bulk.tocsr(), conv_iface.tocsr(), n


def A_mono_stead_advdiff(op: ConvectionOps, cap, D, bc):
    """src/solver/advectiondiffusion.jl:29-44."""
    Ia, Ib = build_I_bc(op, bc)
    Ig = sp.diags(cap.G)
    Id = sp.diags(build_I_D(op, D, cap))
    L, Mx, P, Q = _blocks(op)
    cb, ci, _ = _conv(op)
    return sp.bmat([[cb + ci + Id @ L, ci + Id @ Mx], [Ib * P, Ib * Q + Ia * Ig]], format="csr")


def AdvectionDiffusionSteadyMono(phase: Phase, bc_b, bc_i) -> Solver:
    """src/solver/advectiondiffusion.jl:13-27."""
    s = Solver("Steady", "Monophasic", "DiffusionAdvection")
    s.A = A_mono_stead_advdiff(phase.operator, phase.capacity, phase.Diffusion_coeff, bc_i)
    s.b = b_mono_stead_diff(phase.operator, phase.source, phase.capacity, bc_i)      # :46-58 is the diffusion b
    s.A, s.b = BC_border_mono(s.A, s.b, bc_b, phase.capacity.mesh)
    return s


def A_diph_stead_advdiff(op1: ConvectionOps, op2: ConvectionOps, cap1, cap2, D1, D2, ic):
    """src/solver/advectiondiffusion.jl:95-126."""
    n = int(np.prod(op1.size))
    jump, flux = ic.scalar, ic.flux
    I_n = sp.identity(n, format="csr")
    Id1 = sp.diags(build_I_D(op1, D1, cap1))
    Id2 = sp.diags(build_I_D(op2, D2, cap2))
    L1, M1, P1, Q1 = _blocks(op1)
    L2, M2, P2, Q2 = _blocks(op2)
    cb1, ci1, _ = _conv(op1)
    cb2, ci2, _ = _conv(op2)
    return sp.bmat(
        [[Id1 @ L1 + (cb1 + ci1), Id1 @ M1 + ci1, None, None],
         [None, jump.alpha1 * I_n, None, -jump.alpha2 * I_n],
         [None, None, Id2 @ L2 + (cb2 + ci2), Id2 @ M2 + ci2],
         [flux.beta1 * P1, flux.beta1 * Q1, flux.beta2 * P2, flux.beta2 * Q2]], format="csr")


def AdvectionDiffusionSteadyDiph(phase1: Phase, phase2: Phase, bc_b, ic) -> Solver:
    """src/solver/advectiondiffusion.jl:80-93."""
    s = Solver("Steady", "Diphasic", "DiffusionAdvection")
    s.A = A_diph_stead_advdiff(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity,
                               phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic)
    s.b = b_diph_stead_diff(phase1.operator, phase2.operator, phase1.source, phase2.source, phase1.capacity,
                            phase2.capacity, ic)                     # :128-139 is the diffusion b
    s.A, s.b = BC_border_diph(s.A, s.b, bc_b, phase1.capacity, phase2.capacity)
    return s


def A_mono_unstead_advdiff(op: ConvectionOps, cap, D, bc, dt: float, scheme: str):
    """src/solver/advectiondiffusion.jl:180-213."""
    Ia, Ib = build_I_bc(op, bc)
    Ig = sp.diags(cap.G)
    Id = sp.diags(build_I_D(op, D, cap))
    L, Mx, P, Q = _blocks(op)
    cb, ci, _ = _conv(op)
    tie = Ib * Q + Ia * Ig
    if scheme == "CN":
        return sp.bmat([[op.V + dt / 2 * (cb + ci + Id @ L), dt / 2 * (ci + Id @ Mx)],
                        [dt / 2 * Ib * P, dt / 2 * tie]], format="csr")
    return sp.bmat([[op.V + dt * (cb + ci + Id @ L), dt * (ci + Id @ Mx)], [Ib * P, tie]], format="csr")


def b_mono_unstead_advdiff(op: ConvectionOps, f, cap, D, bc, Ti, dt, t, scheme):
    """src/solver/advectiondiffusion.jl:215-254."""
    N = int(np.prod(op.size))
    Ig = cap.G
    Ia, Ib = build_I_bc(op, bc)
    fn, fn1 = build_source(op, f, t, cap), build_source(op, f, t + dt, cap)
    gn, gn1 = build_g_g(op, bc, cap, t), build_g_g(op, bc, cap, t + dt)
    Tw, Tg = Ti[:N], Ti[N:]
    V = cap.V
    if scheme == "CN":
        Id = sp.diags(build_I_D(op, D, cap))
        L, Mx, P, Q = _blocks(op)
        cb, ci, _ = _conv(op)
        b1 = V * Tw - dt / 2 * ((cb + ci + Id @ L) @ Tw) - dt / 2 * ((ci + Id @ Mx) @ Tg) + dt / 2 * V * (fn + fn1)
        b2 = dt / 2 * Ig * (gn + gn1) - dt / 2 * Ib * (P @ Tw) - dt / 2 * ((Ib * Q + Ia * sp.diags(Ig)) @ Tg)
    else:
        b1 = V * Tw + dt * V * fn1
        b2 = Ig * gn1
    return np.concatenate([b1, b2])


def AdvectionDiffusionUnsteadyMono(phase: Phase, bc_b, bc_i, dt, Ti, scheme) -> Solver:
    """src/solver/advectiondiffusion.jl:163-178, with the border rows applied as DiffusionUnsteadyMono does (the
    reference's constructor leaves them out and only its -- broken -- loop applies them)."""
    s = Solver("Unsteady", "Monophasic", "DiffusionAdvection")
    sch = "CN" if scheme == "CN" else "BE"
    s.A = A_mono_unstead_advdiff(phase.operator, phase.capacity, phase.Diffusion_coeff, bc_i, dt, sch)
    s.b = b_mono_unstead_advdiff(phase.operator, phase.source, phase.capacity, phase.Diffusion_coeff, bc_i, Ti, dt, 0.0, sch)
    s.A, s.b = BC_border_mono(s.A, s.b, bc_b, phase.capacity.mesh, t=0.0)
    return s


def solve_AdvectionDiffusionUnsteadyMono(s: Solver, phase: Phase, dt, Tend, bc_b, bc, scheme, method="\\",
                                         max_steps: Optional[int] = None, **kwargs):
    """src/solver/advectiondiffusion.jl:256-282 (intended loop, see the note above)."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")
    t = 0.0
    solve_system(s, method=method, **kwargs)
    s.states.append(s.x)
    Ti = s.x
    steps = 0
    while t < Tend:
        if max_steps is not None and steps >= max_steps:
            break
        t += dt
        s.A = A_mono_unstead_advdiff(phase.operator, phase.capacity, phase.Diffusion_coeff, bc, dt, scheme)
        s.b = b_mono_unstead_advdiff(phase.operator, phase.source, phase.capacity, phase.Diffusion_coeff, bc, Ti, dt, t, scheme)
        s.A, s.b = BC_border_mono(s.A, s.b, bc_b, phase.capacity.mesh, t=t)
        solve_system(s, method=method, **kwargs)
        s.states.append(s.x)
        Ti = s.x
        steps += 1
    return s


# ---- advection-diffusion, unsteady diphasic (src/solver/advectiondiffusion.jl:299-418) ----------------


def A_diph_unstead_advdiff(op1: ConvectionOps, op2: ConvectionOps, cap1, cap2, D1, D2, ic, dt: float, scheme: str):
    """src/solver/advectiondiffusion.jl:313-354."""
    if scheme not in ("BE", "CN"):
        raise ValueError("Unknown scheme.")                                   # :343-345
    n = int(np.prod(op1.size))
    jump, flux = ic.scalar, ic.flux
    I_n = sp.identity(n, format="csr")
    Id1 = sp.diags(build_I_D(op1, D1, cap1))
    Id2 = sp.diags(build_I_D(op2, D2, cap2))
    L1, M1, P1, Q1 = _blocks(op1)
    L2, M2, P2, Q2 = _blocks(op2)
    cb1, ci1, _ = _conv(op1)
    cb2, ci2, _ = _conv(op2)
    th = dt / 2 if scheme == "CN" else dt
    return sp.bmat(
        [[op1.V + th * (cb1 + ci1) + th * (Id1 @ L1), th * ci1 + th * (Id1 @ M1), None, None],
         [None, jump.alpha1 * I_n, None, -jump.alpha2 * I_n],
         [None, None, op2.V + th * (cb2 + ci2) + th * (Id2 @ L2), th * ci2 + th * (Id2 @ M2)],
         [flux.beta1 * P1, flux.beta1 * Q1, flux.beta2 * P2, flux.beta2 * Q2]], format="csr")


def b_diph_unstead_advdiff(op1: ConvectionOps, op2: ConvectionOps, f1, f2, cap1, cap2, D1, D2, ic, Ti, dt, t, scheme):
    """src/solver/advectiondiffusion.jl:356-387.  Restated as written: the CN right-hand side carries the convection
    terms only -- the diffusion part of the explicit half step is absent (:375-377) -- unlike the monophasic one."""
    N = int(np.prod(op1.size))
    jump, flux = ic.scalar, ic.flux
    gg = build_g_g(op1, jump, cap1)
    hh = build_g_g(op2, flux, cap2)
    f1n, f2n = build_source(op1, f1, t, cap1), build_source(op2, f2, t, cap2)
    f1p, f2p = build_source(op1, f1, t + dt, cap1), build_source(op2, f2, t + dt, cap2)
    Tw1, Tg1, Tw2, Tg2 = Ti[:N], Ti[N:2 * N], Ti[2 * N:3 * N], Ti[3 * N:]
    if scheme == "CN":
        cb1, ci1, _ = _conv(op1)
        cb2, ci2, _ = _conv(op2)
        b1 = cap1.V * Tw1 - dt / 2 * ((cb1 + ci1) @ Tw1) - dt / 2 * (ci1 @ Tg1) + dt / 2 * cap1.V * (f1n + f1p)
        b3 = cap2.V * Tw2 - dt / 2 * ((cb2 + ci2) @ Tw2) - dt / 2 * (ci2 @ Tg2) + dt / 2 * cap2.V * (f2n + f2p)
    elif scheme == "BE":
        b1 = cap1.V * Tw1 + dt * cap1.V * f1p
        b3 = cap2.V * Tw2 + dt * cap2.V * f2p
    else:
        raise ValueError("Unknown scheme.")
    return np.concatenate([b1, gg, b3, cap2.G * hh])


def AdvectionDiffusionUnsteadyDiph(phase1: Phase, phase2: Phase, bc_b, ic, dt, Ti, scheme, ctor_borders: bool = False) -> Solver:
    """src/solver/advectiondiffusion.jl:299-311.  The reference's constructor does not apply the border rows (its loop
    does, :404); ctor_borders=True applies them as DiffusionUnsteadyDiph does -- the form the HIP path implements."""
    s = Solver("Unsteady", "Diphasic", "DiffusionAdvection")
    s.A = A_diph_unstead_advdiff(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity,
                                 phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, dt, scheme)
    s.b = b_diph_unstead_advdiff(phase1.operator, phase2.operator, phase1.source, phase2.source, phase1.capacity,
                                 phase2.capacity, phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, Ti, dt, 0.0, scheme)
    if ctor_borders:
        s.A, s.b = BC_border_diph(s.A, s.b, bc_b, phase1.capacity, phase2.capacity)
    return s


def solve_AdvectionDiffusionUnsteadyDiph(s, phase1, phase2, dt, Tend, bc_b, ic, scheme, method="\\",
                                         max_steps: Optional[int] = None, **kwargs):
    """src/solver/advectiondiffusion.jl:389-418."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")
    t = 0.0
    solve_system(s, method=method, **kwargs)
    s.states.append(s.x)
    Ti = s.x
    steps = 0
    while t < Tend:
        if max_steps is not None and steps >= max_steps:
            break
        t += dt
        s.A = A_diph_unstead_advdiff(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity,
                                     phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, dt, scheme)
        s.b = b_diph_unstead_advdiff(phase1.operator, phase2.operator, phase1.source, phase2.source, phase1.capacity,
                                     phase2.capacity, phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, Ti, dt, t, scheme)
        s.A, s.b = BC_border_diph(s.A, s.b, bc_b, phase1.capacity, phase2.capacity)
        solve_system(s, method=method, **kwargs)
        s.states.append(s.x)
        Ti = s.x
        steps += 1
    return s


# ---- Darcy: aliases of the diffusion drivers (src/solver/darcy.jl) --------------------


def DarcyFlow(phase: Phase, bc_b, bc_i) -> Solver:
    """src/solver/darcy.jl:1-15."""
    return DiffusionSteadyMono(phase, bc_b, bc_i)


def solve_DarcyFlow(s: Solver, method: str = "\\", **kwargs):
    """src/solver/darcy.jl:17-24."""
    solve_DiffusionSteadyMono(s, method=method, **kwargs)
    s.states.append(s.x)
    return s


def solve_darcy_velocity(solver: Solver, fluide: Phase, state_i: int = 1) -> np.ndarray:
    """src/solver/darcy.jl:26-41 (state_i 1-based)."""
    ct = fluide.capacity.cell_types
    st = np.array(solver.states[state_i - 1], dtype=float)
    half = st.shape[0] // 2
    pw, pg = st[:half].copy(), st[half:].copy()
    pw[ct == 0] = np.nan
    pg[ct == 0] = np.nan
    pg[ct == 1] = np.nan
    return -grad(fluide.operator, np.concatenate([pw, pg]))


# ---- unsteady diphasic (config 5) -----------------------------------------------------


def A_diph_unstead_diff(op1, op2, cap1, cap2, D1, D2, ic: InterfaceConditions, dt, scheme):
    """src/solver/diffusion.jl:334-389."""
    n = int(np.prod(op1.size))
    jump, flux = ic.scalar, ic.flux
    I_n = sp.identity(n, format="csr")
    Ia1, Ia2 = jump.alpha1 * I_n, jump.alpha2 * I_n
    Id1 = sp.diags(build_I_D(op1, D1, cap1))
    Id2 = sp.diags(build_I_D(op2, D2, cap2))
    L1, M1, P1, Q1 = _blocks(op1)
    L2, M2, P2, Q2 = _blocks(op2)
    th = dt / 2 if scheme == "CN" else dt
    block1 = op1.V + th * (Id1 @ L1)
    block2 = th * (Id1 @ M1)
    block3 = op2.V + th * (Id2 @ L2)
    block4 = th * (Id2 @ M2)
    block5 = flux.beta1 * P1
    block6 = flux.beta1 * Q1
    block7 = flux.beta2 * P2
    block8 = flux.beta2 * Q2
    return sp.bmat(
        [[block1, block2, None, None],
         [None, Ia1, None, -Ia2],
         [None, None, block3, block4],
         [block5, block6, block7, block8]], format="csr")


def b_diph_unstead_diff(op1, op2, f1, f2, cap1, cap2, D1, D2, ic, Ti, dt, t, scheme):
    """src/solver/diffusion.jl:391-420 (g, h built WITHOUT t, :397)."""
    N = int(np.prod(op1.size))
    jump, flux = ic.scalar, ic.flux
    gg = build_g_g(op1, jump, cap1)
    hh = build_g_g(op2, flux, cap2)
    f1n, f2n = build_source(op1, f1, t, cap1), build_source(op2, f2, t, cap2)
    f1p, f2p = build_source(op1, f1, t + dt, cap1), build_source(op2, f2, t + dt, cap2)
    Id1, Id2 = build_I_D(op1, D1, cap1), build_I_D(op2, D2, cap2)
    Tw1, Tg1, Tw2, Tg2 = Ti[:N], Ti[N:2 * N], Ti[2 * N:3 * N], Ti[3 * N:]
    if scheme == "CN":
        L1, M1, _, _ = _blocks(op1)
        L2, M2, _, _ = _blocks(op2)
        b1 = (cap1.V * Tw1 - dt / 2 * Id1 * (L1 @ Tw1)) - dt / 2 * Id1 * (M1 @ Tg1) + dt / 2 * cap1.V * (f1n + f1p)
        b3 = (cap2.V * Tw2 - dt / 2 * Id2 * (L2 @ Tw2)) - dt / 2 * Id2 * (M2 @ Tg2) + dt / 2 * cap2.V * (f2n + f2p)
    else:
        b1 = cap1.V * Tw1 + dt * cap1.V * f1p
        b3 = cap2.V * Tw2 + dt * cap2.V * f2p
    return np.concatenate([b1, gg, b3, cap2.G * hh])


def DiffusionUnsteadyDiph(phase1: Phase, phase2: Phase, bc_b, ic, dt, Ti, scheme) -> Solver:
    """src/solver/diffusion.jl:319-332."""
    s = Solver("Unsteady", "Diphasic", "Diffusion")
    s.A = A_diph_unstead_diff(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity,
                              phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, dt, scheme)
    s.b = b_diph_unstead_diff(phase1.operator, phase2.operator, phase1.source, phase2.source, phase1.capacity,
                              phase2.capacity, phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, Ti, dt, 0.0, scheme)
    s.A, s.b = BC_border_diph(s.A, s.b, bc_b, phase1.capacity, phase2.capacity)
    return s


def solve_DiffusionUnsteadyDiph(s, phase1, phase2, dt, Tend, bc_b, ic, scheme, method="\\",
                                max_steps: Optional[int] = None, **kwargs):
    """src/solver/diffusion.jl:422-454."""
    if s.A is None:
        raise RuntimeError("Solver is not initialized. Call a solver constructor first.")
    t = 0.0
    solve_system(s, method=method, **kwargs)
    s.states.append(s.x)
    Ti = s.x
    s.A = A_diph_unstead_diff(phase1.operator, phase2.operator, phase1.capacity, phase2.capacity,
                              phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, dt, scheme)
    steps = 0
    while t < Tend:
        if max_steps is not None and steps >= max_steps:
            break
        t += dt
        s.b = b_diph_unstead_diff(phase1.operator, phase2.operator, phase1.source, phase2.source, phase1.capacity,
                                  phase2.capacity, phase1.Diffusion_coeff, phase2.Diffusion_coeff, ic, Ti, dt, t, scheme)
        s.A, s.b = BC_border_diph(s.A, s.b, bc_b, phase1.capacity, phase2.capacity)  # :446 (no t)
        solve_system(s, method=method, **kwargs)
        s.states.append(s.x)
        Ti = s.x
        steps += 1
    return s


# =============================================================================
# Convergence metric                                     src/convergence.jl:4-93
# =============================================================================


def lp_norm(errors, indices, pval, cap: Capacity):
    """src/convergence.jl:4-15."""
    if pval == math.inf:
        return float(np.max(np.abs(errors[indices]), initial=0.0))
    part = float(np.sum(np.abs(errors[indices]) ** pval * cap.V[indices]))
    return (part / float(np.sum(cap.V))) ** (1.0 / pval)


def check_convergence(u_analytical: Callable, solver: Solver, cap: Capacity, p=2):
    """src/convergence.jl:45-93 (absolute norms)."""
    u_ana = np.array([u_analytical(*c) for c in cap.C_w], dtype=np.float64)
    u_num = solver.x[: len(solver.x) // 2]
    err = u_ana - u_num
    ct = cap.cell_types
    idx_all = np.flatnonzero((ct == 1) | (ct == -1))
    idx_full = np.flatnonzero(ct == 1)
    idx_cut = np.flatnonzero(ct == -1)
    idx_empty = np.flatnonzero(ct == 0)
    return (u_ana, u_num, lp_norm(err, idx_all, p, cap), lp_norm(err, idx_full, p, cap),
            lp_norm(err, idx_cut, p, cap), lp_norm(err, idx_empty, p, cap))
