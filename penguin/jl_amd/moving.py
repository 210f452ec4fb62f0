"""Prescribed-motion diffusion through the HIP path -- the reference's
src/prescribedmotionsolver/diffusion.jl:16-35 (constructor), :100-225 (blocks), :227-268 (time loop) and
src/mesh.jl:129-146 (SpaceTimeMesh), with the same names and argument order.

    STmesh   = SpaceTimeMesh(mesh, [0.0, Δt])
    capacity = Capacity(body, STmesh)                       # body: MovingSphere / MovingHalfSpace
    operator = DiffusionOps(capacity)
    solver   = MovingDiffusionUnsteadyMono(Phase(capacity, operator, f, K), bc_b, bc, Δt, u0, mesh, "BE")
    solve_MovingDiffusionUnsteadyMono_b(solver, phase, body, Δt, Tstart, Tend, bc_b, bc, mesh, "BE")

Every time slab builds a new space-time capacity on the GPU (pg_capacity_create_spacetime), assembles the moving blocks
there (pg_solver_create_moving_mono) and solves; the host only evaluates the motion at the time-quadrature nodes and the
user's closures.  The level set of the reference is an arbitrary `body(x..., t)`; here it is a tagged moving body, as for
static capacities (an arbitrary one: integrate it yourself and use SpaceTimeCapacity.from_arrays).
1-D and 2-D in space: the reference's 3-D+t selection `[1:end÷2]` drops the z direction."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Sequence

import numpy as np

from . import _lib as L
from . import api
from ._lib import PenguinHipError


class SpaceTimeMesh:
    """SpaceTimeMesh(spaceMesh, [t0, t1]) -- src/mesh.jl:129-146: nodes / centers / dims with the time dimension appended."""

    def __init__(self, spaceMesh: api.Mesh, time: Sequence[float], tag=None):
        if len(time) != 2:
            raise ValueError("SpaceTimeMesh: one time cell [t, t+Δt] (every call site of the moving solvers)")
        self.space = spaceMesh
        self.time = (float(time[0]), float(time[1]))
        self.nodes = tuple(spaceMesh.nodes) + (np.array(self.time),)
        self.centers = tuple(spaceMesh.centers) + (np.array([(self.time[0] + self.time[1]) / 2]),)
        self.dims = tuple(len(c) for c in self.centers)
        self.tag = tag if tag is not None else getattr(spaceMesh, "tag", None)
        self.N = spaceMesh.N + 1


def _ddt(fn: Callable, t: float, h: float) -> np.ndarray:
    return (np.atleast_1d(np.asarray(fn(t + h), dtype=np.float64)) - np.atleast_1d(np.asarray(fn(t - h), dtype=np.float64))) / (2 * h)


class MovingSphere:
    """f(x, t) = |x - center(t)| - radius(t), fluid where f <= 0 (complement: outside, the growing disc of
    examples/2D/SolidMoving/MovingHeat.jl:19).  dcenter / dradius: exact time derivatives (default: central differences);
    they only enter the space-time interface measure Γ."""

    def __init__(self, center: Callable, radius: Callable, complement: bool = False, dcenter: Optional[Callable] = None,
                 dradius: Optional[Callable] = None):
        self.center, self.radius, self.complement = center, radius, bool(complement)
        self.dcenter, self.dradius = dcenter, dradius
        self.kind = L.PG_BODY_BALL
        self.axis, self.sign = 0, 1.0

    def __call__(self, *xt):
        *x, t = xt
        c = np.atleast_1d(self.center(t))
        f = np.sqrt(sum((np.asarray(x[d]) - c[d]) ** 2 for d in range(len(c)))) - self.radius(t)
        return -f if self.complement else f

    def _state(self, t: float, N: int):
        c = np.atleast_1d(np.asarray(self.center(t), dtype=np.float64))
        if len(c) != N:
            raise ValueError("body dimension does not match the mesh")
        out = np.zeros(4)
        out[:N] = c
        out[3] = float(self.radius(t))
        return out

    def _rate(self, t: float, N: int, h: float):
        out = np.zeros(4)
        out[:N] = np.atleast_1d(self.dcenter(t)) if self.dcenter else _ddt(self.center, t, h)
        out[3] = float(self.dradius(t)) if self.dradius else float(_ddt(self.radius, t, h)[0])
        return out


MovingCircle = MovingSphere


class MovingHalfSpace:
    """f(x, t) = sign (x_axis - position(t)), fluid where f < 0 -- examples/1D/SolidMoving/MovingHeat.jl:18
    `body = (x,t,_=0) -> x - xf - c*sqrt(t)`."""

    def __init__(self, axis: int, position: Callable, sign: float = 1.0, complement: bool = False,
                 dposition: Optional[Callable] = None):
        self.axis, self.position, self.sign, self.complement = int(axis), position, (-1.0 if sign < 0 else 1.0), bool(complement)
        self.dposition = dposition
        self.kind = L.PG_BODY_HALFSPACE

    def __call__(self, *xt):
        *x, t = xt
        f = self.sign * (np.asarray(x[self.axis]) - self.position(t))
        return -f if self.complement else f

    def _state(self, t: float, N: int):
        out = np.zeros(4)
        out[0] = float(self.position(t))
        out[3] = 1.0
        return out

    def _rate(self, t: float, N: int, h: float):
        out = np.zeros(4)
        out[0] = float(self.dposition(t)) if self.dposition else float(_ddt(self.position, t, h)[0])
        return out


def time_rule(t0: float, t1: float, panels: int, order: int):
    x, w = np.polynomial.legendre.leggauss(order)
    edges = np.linspace(t0, t1, panels + 1)
    tau = np.concatenate([0.5 * (a + b) + 0.5 * (b - a) * x for a, b in zip(edges[:-1], edges[1:])])
    wt = np.concatenate([0.5 * (b - a) * w for a, b in zip(edges[:-1], edges[1:])])
    return tau, wt * ((t1 - t0) / wt.sum())        # (the weights add up to t1 - t0 to the last bit the ABI checks)


class SpaceTimeCapacity(api.Capacity):
    """Capacity(body, STmesh): first time layer of the (N+1)-D capacity, resident on the GPU.  `V`, `A`, `B`, `W`, `Γ`,
    `C_ω`, `C_γ`, `cell_types` are the layer's N-D fields (length M = prod(n_d+1)); `Vn_1` / `Vn` the time-face capacities
    A_(N+1) at t0 / t1 (the reference's names, diffusion.jl:113-114); `C_ω_st` / `C_γ_st` the (M, N+1) centroids with the
    time component; `A_st` the reference-layout A tuple (N+1 arrays of length 2M, second layer = time padding).
    time_panels x time_order: the composite Gauss-Legendre rule in time (space is exact)."""

    def __init__(self, body, mesh: SpaceTimeMesh, method: str = "VOFI", compute_centroids: bool = True,
                 time_panels: int = 16, time_order: int = 4):
        if not hasattr(body, "_state"):
            raise PenguinHipError("Capacity(body, SpaceTimeMesh): pass a MovingSphere / MovingHalfSpace (an arbitrary moving "
                                  "level set needs SpaceTimeCapacity.from_arrays)")
        L.init()
        self.stmesh, self.mesh, self.body, self.compute_centroids = mesh, mesh.space, body, compute_centroids
        N = self.mesh.N
        t0, t1 = mesh.time
        tau, wt = time_rule(t0, t1, time_panels, time_order)
        h = 1e-6 * (t1 - t0)
        nodes = np.zeros((len(tau), 10))
        for k, (t, w) in enumerate(zip(tau, wt)):
            st, rt = body._state(float(t), N), body._rate(float(t), N, h)
            nodes[k] = [t, w, st[0], st[1], st[2], st[3], rt[0], rt[1], rt[2], rt[3]]
        self._nodes = np.ascontiguousarray(nodes)
        desc = L.pg_motion_desc()
        desc.body_kind = body.kind
        desc.flags = (L.PG_FLAG_COMPLEMENT if body.complement else 0) | (0 if compute_centroids else L.PG_FLAG_NO_CENTROIDS)
        desc.axis, desc.sign = int(body.axis), float(body.sign)
        desc.nq, desc.t0, desc.t1 = len(tau), t0, t1
        desc.nodes = L.dptr(self._nodes)
        b0, b1 = body._state(t0, N), body._state(t1, N)
        for i in range(4):
            desc.body0[i], desc.body1[i] = b0[i], b1[i]
        self._h = C.c_void_p()
        L.check(L.lib().pg_capacity_create_spacetime(self.mesh._h, C.byref(desc), C.byref(self._h)))
        self._cache = {}

    @classmethod
    def from_arrays(cls, stmesh: SpaceTimeMesh, V, A, B, W, Gamma, C_omega, C_gamma, cell_types, Vn_1, Vn, body=None):
        """First-layer fields computed by the caller (arbitrary moving bodies).  C_omega / C_gamma: (M, N+1)."""
        mesh = stmesh.space
        N = mesh.N
        base = api.Capacity.from_arrays(mesh, V, A, B, W, Gamma, np.asarray(C_omega)[:, :N],
                                        np.asarray(C_gamma)[:, :N] if C_gamma is not None and len(C_gamma) else None,
                                        cell_types, body=body)
        self = cls.__new__(cls)
        self.__dict__.update(base.__dict__)
        base._h = None                                   # ownership moved
        self.stmesh = stmesh
        v0, v1 = np.ascontiguousarray(Vn_1, dtype=np.float64), np.ascontiguousarray(Vn, dtype=np.float64)
        ctw = np.ascontiguousarray(np.asarray(C_omega)[:, N], dtype=np.float64)
        ctg = np.ascontiguousarray(np.asarray(C_gamma)[:, N], dtype=np.float64) if C_gamma is not None and len(C_gamma) else None
        L.check(L.lib().pg_capacity_set_spacetime(self._h, C.c_double(stmesh.time[0]), C.c_double(stmesh.time[1]), L.dptr(v0),
                                                  L.dptr(v1), L.dptr(ctw), L.dptr(ctg) if ctg is not None else None))
        return self

    @property
    def Vn_1(self):
        return self._get(L.PG_CAP_ST_V0)

    @property
    def Vn(self):
        return self._get(L.PG_CAP_ST_V1)

    @property
    def C_ω_st(self):
        return np.column_stack([self.C_ω, self._get(L.PG_CAP_ST_CT_OMEGA)])

    @property
    def C_γ_st(self):
        if not self.compute_centroids:
            return np.zeros((0, self.N + 1))
        return np.column_stack([self.C_γ, self._get(L.PG_CAP_ST_CT_GAMMA)])

    @property
    def A_st(self):
        z = np.zeros_like(self.V)
        return tuple(np.concatenate([a, z]) for a in self.A) + (np.concatenate([self.Vn_1, self.Vn]),)

    # closures see the space-time centroids (build_source / build_g_g read capacity.C_ω / C_γ of the (N+1)-D capacity)
    def _cw(self):
        return self.C_ω_st

    def _cg(self):
        return self.C_γ_st


def _capacity_new(cls, body=None, mesh=None, *args, **kwargs):
    """Capacity(body, STmesh) of the reference: the same constructor name serves both mesh kinds."""
    if cls is api.Capacity and isinstance(mesh, SpaceTimeMesh):
        return object.__new__(SpaceTimeCapacity)
    return object.__new__(cls)


api.Capacity.__new__ = staticmethod(_capacity_new)


def _create_step(s: api.Solver, phase: api.Phase, bc_b, bc_i, Δt: float, Tᵢ: Optional[np.ndarray], mesh: api.Mesh, scheme: str,
                 t: float, from_previous: bool = False):
    """A_/b_mono_unstead_diff_moving + BC_border_mono!(A, b, bc_b, mesh; t) of one slab (diffusion.jl:29-33, 254-258)."""
    cap = phase.capacity
    if not isinstance(cap, SpaceTimeCapacity):
        raise PenguinHipError("the moving solver needs a space-time capacity: Capacity(body, SpaceTimeMesh(mesh, [t, t+Δt]))")
    if cap.mesh is not mesh and tuple(cap.mesh.dims) != tuple(mesh.dims):
        raise ValueError("mesh does not match the capacity's space mesh")
    M = int(np.prod(mesh.ext))
    sch = "CN" if scheme == "CN" else "BE"
    desc, g_arr = api._interface_desc(bc_i, cap._cg, None)          # build_g_g(operator, bc, capacity): value(C_γ...)  :172
    D_arr = api._dcoef(phase, M)
    f1 = api._padded_field(api._eval(phase.source, cap._cw, float(t + Δt), 3), M)    # f(C_ω..., t+Δt)   :171
    f0 = api._padded_field(api._eval(phase.source, cap._cw, float(t), 3), M) if sch == "CN" else None
    if sch == "CN" and f1 is not None and f0 is None:
        f0 = np.zeros(M)
    borders, nb, bvals = api._border_descs(bc_b, mesh, float(t))
    old, new = s._h, C.c_void_p()
    common = (cap._h, phase.operator._h, C.byref(desc), borders, C.c_int32(nb), L.dptr(D_arr) if D_arr is not None else None,
              L.dptr(f0) if f0 is not None else None, L.dptr(f1) if f1 is not None else None)
    if from_previous:      # the previous slab's state stays on the device (pg_solver_create_moving_mono_next)
        L.check(L.lib().pg_solver_create_moving_mono_next(*common, old, C.c_int32(L.PG_SCHEME[sch]), C.byref(new)))
    else:
        L.check(L.lib().pg_solver_create_moving_mono(*common, L.dptr(Tᵢ) if Tᵢ is not None else None,
                                                     C.c_int32(L.PG_SCHEME[sch]), C.byref(new)))
    s._h = new
    if old:
        L.check(L.lib().pg_solver_destroy(old))
    if bvals is not None:
        L.check(L.lib().pg_solver_set_border_values(s._h, L.dptr(bvals)))
    s._keep = (cap, phase.operator)      # the device solver reads the capacity: keep it alive as long as the handle
    s._initial_done = False


def MovingDiffusionUnsteadyMono(phase: api.Phase, bc_b, bc_i, Δt: float, Tᵢ: np.ndarray, mesh: api.Mesh, scheme: str,
                                verbose: bool = False) -> api.Solver:
    """MovingDiffusionUnsteadyMono(phase, bc_b, bc_i, Δt, Tᵢ, mesh, scheme) -- prescribedmotionsolver/diffusion.jl:16-35."""
    if verbose:
        print("Solver Creation:\n- Moving problem\n- Monophasic problem\n- Unsteady problem\n- Diffusion problem")
    s = api.Solver("Unsteady", "Monophasic", "Diffusion")
    M = int(np.prod(mesh.ext))
    s._nunk = 2 * M
    if Tᵢ is not None:
        Tᵢ = np.ascontiguousarray(Tᵢ, dtype=np.float64)
        if Tᵢ.shape != (2 * M,):
            raise ValueError(f"Tᵢ must have length 2*prod(n+1) = {2 * M}")
    s._ctx = dict(phase=phase, bc_i=bc_i, dt=float(Δt), M=M)
    _create_step(s, phase, bc_b, bc_i, float(Δt), Tᵢ, mesh, scheme, 0.0)     # t = 0.0 in b and the border rows (:27-33)
    return s


def solve_MovingDiffusionUnsteadyMono_b(s: api.Solver, phase: api.Phase, body, Δt: float, Tₛ: float, Tₑ: float, bc_b, bc,
                                        mesh: api.Mesh, scheme: str, method="gmres", algorithm=None, geometry_method="VOFI",
                                        verbose: bool = False, max_steps: Optional[int] = None, time_panels: int = 16,
                                        time_order: int = 4, save_states: bool = True, **kwargs):
    """solve_MovingDiffusionUnsteadyMono!(s, phase, body, Δt, Tₛ, Tₑ, bc_b, bc, mesh, scheme; method, ...) --
    prescribedmotionsolver/diffusion.jl:227-268: the constructor's system first (states[1]); then `while t < Tₑ`:
    t += Δt, the capacity of the slab [t, t+Δt], new blocks and border rows at t, solve, push.
    `save_states=False` (not in the reference): the state is handed from slab to slab on the device and only the last one is
    fetched (`s.x`, `s.states[-1]`); the reference's `push!(s.states, s.x)` costs a device-to-host copy of 2M doubles per slab."""
    if s is None or not s._h:
        raise PenguinHipError("Solver is not initialized. Call a solver constructor first.")
    opts = api._krylov_opts(method, kwargs)
    sch = "CN" if scheme == "CN" else "BE"

    def solve_current(what):
        info = L.pg_step_info()
        L.check(L.lib().pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
        api._step_info_check(s, info, what)
        s._initial_done = True
        s.ch.append(info)
        if save_states:
            s.x = s._fetch_state(-1)
            s.states.append(s.x)
        if verbose:
            print("Solver Extremum : ", float(info.extremum))

    t = float(Tₛ)
    if verbose:
        print(f"Time : {t}")
    solve_current("the first solve")
    Tᵢ = s.x
    steps = 0
    while t < Tₑ:
        if max_steps is not None and steps >= max_steps:
            break
        t += Δt
        if verbose:
            print(f"Time : {t}")
        cap = api.Capacity(body, SpaceTimeMesh(mesh, [t, t + Δt]), time_panels=time_panels, time_order=time_order,
                           compute_centroids=True, method=geometry_method)
        ph = api.Phase(cap, api.DiffusionOps(cap), phase.source, phase.Diffusion_coeff)
        _create_step(s, ph, bc_b, bc, float(Δt), Tᵢ, mesh, sch, t, from_previous=not save_states)
        solve_current(f"the solve of the slab starting at t = {t}")
        Tᵢ = s.x
        steps += 1
    if not save_states:
        s.x = s._fetch_state(-1)
        s.states.append(s.x)
    return s


# ---------------------------------------------------------------------------------------------------------------------
# two phases                                                          prescribedmotionsolver/diffusion.jl:272-535
# ---------------------------------------------------------------------------------------------------------------------
def _create_step_diph(s: api.Solver, phase1: api.Phase, phase2: api.Phase, bc_b, ic, Δt: float, Tᵢ: Optional[np.ndarray],
                      mesh: api.Mesh, scheme: str, t: float, from_previous: bool = False):
    """A_/b_diph_unstead_diff_moving + BC_border_diph!(A, b, bc_b, mesh) of one slab (diffusion.jl:281-288, 519-523)."""
    cap1, cap2 = phase1.capacity, phase2.capacity
    for cap in (cap1, cap2):
        if not isinstance(cap, SpaceTimeCapacity):
            raise PenguinHipError("the moving solver needs space-time capacities: Capacity(body, SpaceTimeMesh(mesh, [t, t+Δt]))")
        if cap.mesh is not mesh and tuple(cap.mesh.dims) != tuple(mesh.dims):
            raise ValueError("mesh does not match the capacity's space mesh")
    M = int(np.prod(mesh.ext))
    sch = "CN" if scheme == "CN" else "BE"
    jump, flux = ic.scalar, ic.flux
    # build_g_g(operator, jump, capacity): value(C_γ...) at the space-time interface centroids, no time argument (:423-424)
    g = api._eval(jump.value, cap1._cg, None, 3) if callable(jump.value) else float(jump.value)
    h = api._eval(flux.value, cap2._cg, None, 3) if callable(flux.value) else float(flux.value)
    g_arr = None if isinstance(g, float) else api._padded_field(g, M)
    h_arr = None if isinstance(h, float) else api._padded_field(h, M)
    p = lambda a: L.dptr(a) if a is not None else None
    desc = L.pg_jump_desc(float(jump.α1), float(jump.α2), g if isinstance(g, float) else 0.0, float(flux.β1), float(flux.β2),
                          h if isinstance(h, float) else 0.0, p(g_arr), p(h_arr))
    D1, D2 = api._dcoef(phase1, M), api._dcoef(phase2, M)
    fs = []
    for ph in (phase1, phase2):
        f1 = api._padded_field(api._eval(ph.source, ph.capacity._cw, float(t + Δt), 3), M)       # f(C_ω..., t+Δt)   :416,418
        f0 = api._padded_field(api._eval(ph.source, ph.capacity._cw, float(t), 3), M) if sch == "CN" else None
        if sch == "CN" and f1 is not None and f0 is None:
            f0 = np.zeros(M)
        fs.append((f0, f1))
    borders, nb, bvals = api._border_descs(bc_b, mesh, None)       # BC_border_diph!(s.A, s.b, bc_b, mesh): no t (:288, :523)
    old, new = s._h, C.c_void_p()
    L.check(L.lib().pg_solver_create_moving_diph(
        cap1._h, phase1.operator._h, cap2._h, phase2.operator._h, C.byref(desc), borders, C.c_int32(nb), p(D1), p(D2),
        p(fs[0][0]), p(fs[0][1]), p(fs[1][0]), p(fs[1][1]),
        None if from_previous else (L.dptr(Tᵢ) if Tᵢ is not None else None), old if from_previous else None,
        C.c_int32(L.PG_SCHEME[sch]), C.byref(new)))
    s._h = new
    if old:
        L.check(L.lib().pg_solver_destroy(old))
    if bvals is not None:
        L.check(L.lib().pg_solver_set_border_values(s._h, L.dptr(bvals)))
    s._keep = (cap1, cap2, phase1.operator, phase2.operator)
    s._initial_done = False


def MovingDiffusionUnsteadyDiph(phase1: api.Phase, phase2: api.Phase, bc_b, ic, Δt: float, Tᵢ: np.ndarray, mesh: api.Mesh,
                                scheme: str, verbose: bool = False) -> api.Solver:
    """MovingDiffusionUnsteadyDiph(phase1, phase2, bc_b, ic, Δt, Tᵢ, mesh, scheme) -- prescribedmotionsolver/diffusion.jl:272-290."""
    if verbose:
        print("Solver Creation:\n- Moving problem\n- Diphasic problem\n- Unsteady problem\n- Diffusion problem")
    s = api.Solver("Unsteady", "Diphasic", "Diffusion")
    M = int(np.prod(mesh.ext))
    s._nunk = 4 * M
    if Tᵢ is not None:
        Tᵢ = np.ascontiguousarray(Tᵢ, dtype=np.float64)
        if Tᵢ.shape != (4 * M,):
            raise ValueError(f"Tᵢ must have length 4*prod(n+1) = {4 * M}")
    s._ctx = dict(dt=float(Δt), M=M)
    _create_step_diph(s, phase1, phase2, bc_b, ic, float(Δt), Tᵢ, mesh, scheme, 0.0)      # t = 0.0 in b (:282, :285)
    return s


def solve_MovingDiffusionUnsteadyDiph_b(s: api.Solver, phase1: api.Phase, phase2: api.Phase, body, body_c, Δt: float, Tₑ: float,
                                        bc_b, ic, mesh: api.Mesh, scheme: str, method="gmres", algorithm=None, verbose: bool = False,
                                        max_steps: Optional[int] = None, time_panels: int = 16, time_order: int = 4,
                                        save_states: bool = True, **kwargs):
    """solve_MovingDiffusionUnsteadyDiph!(s, phase1, phase2, body, body_c, Δt, Tₑ, bc_b, ic, mesh, scheme; method, ...) --
    prescribedmotionsolver/diffusion.jl:501-535: the constructor's system first (states[1]); then from t = 0.0 `while t < Tₑ`:
    t += Δt, the two capacities of the slab [t, t+Δt], new blocks and border rows, solve, push."""
    if s is None or not s._h:
        raise PenguinHipError("Solver is not initialized. Call a solver constructor first.")
    opts = api._krylov_opts(method, kwargs)
    sch = "CN" if scheme == "CN" else "BE"

    def solve_current(what):
        info = L.pg_step_info()
        L.check(L.lib().pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
        api._step_info_check(s, info, what)
        s._initial_done = True
        s.ch.append(info)
        if save_states:
            s.x = s._fetch_state(-1)
            s.states.append(s.x)
        if verbose:
            print("Solver Extremum : ", float(info.extremum))

    t = 0.0                                        # :513
    if verbose:
        print(f"Time : {t}")
    solve_current("the first solve")
    Tᵢ = s.x
    steps = 0
    while t < Tₑ:
        if max_steps is not None and steps >= max_steps:
            break
        t += Δt
        if verbose:
            print(f"Time : {t}")
        caps = [api.Capacity(b, SpaceTimeMesh(mesh, [t, t + Δt]), time_panels=time_panels, time_order=time_order,
                             compute_centroids=True) for b in (body, body_c)]
        ph1 = api.Phase(caps[0], api.DiffusionOps(caps[0]), phase1.source, phase1.Diffusion_coeff)
        ph2 = api.Phase(caps[1], api.DiffusionOps(caps[1]), phase2.source, phase2.Diffusion_coeff)
        _create_step_diph(s, ph1, ph2, bc_b, ic, float(Δt), Tᵢ, mesh, sch, t, from_previous=not save_states)
        solve_current(f"the solve of the slab starting at t = {t}")
        Tᵢ = s.x
        steps += 1
    if not save_states:
        s.x = s._fetch_state(-1)
        s.states.append(s.x)
    return s
