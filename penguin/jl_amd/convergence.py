"""Mesh-refinement studies of the reference, run through the HIP path, with the CSV files the reference writes.

* `run_mesh_convergence` -- benchmark/Heat3D.jl:12-196 (3-D sphere, BE first solve then CN, Δt = 0.75 h², Tend = 0.1,
  `method = \\`) and its 2-D twin on the disc of examples/2D/Diffusion/Heat.jl (BE, Δt = 0.25 h²): one
  `check_convergence` per mesh, per-mesh `mesh_NNNNxNNNN[xNNNN].csv`, `summary.csv` and `convergence_rates.csv` (the
  least-squares slope of log(err) over log(h), rounded to 2 digits) exactly as benchmark/Heat3D.jl:90-160 lays them out.
* `heat3d_analytical`, `radial_heat_xy` -- the reference's own analytic series (benchmark/Heat3D.jl:207-252,
  examples/2D/Diffusion/Heat.jl:63-100).
* `pairwise_orders` -- BenchPhaseFlow/utils/convergence.jl `compute_pairwise_orders`.

The orders these runs produce are the only reference-held evidence that constrains the capacity conventions libvofi would
otherwise pin: a wrong face / staggered-volume convention still passes a 1e-2 known answer but loses the
second-order slope (BenchPhaseFlow/problems/scalar/Scalar_3D_Diffusion_Heat_Dirichlet.jl:170-178 asserts `orders.all > 1`).
"""
from __future__ import annotations

import csv
import math
import os
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import api


def heat3d_analytical(x, y, z, t, center, radius, D=1.0, T0=0.0, Tb=1.0):
    """benchmark/Heat3D.jl:207-252: a sphere with surface value Tb from the uniform state T0; 20 terms of the series, the
    r -> 0 limit below r = 1e-10, Tb outside.  Scalars or arrays (evaluated for all points at once)."""
    x, y, z = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), np.asarray(z, dtype=np.float64)
    r = np.sqrt((x - center[0]) ** 2 + (y - center[1]) ** 2 + (z - center[2]) ** 2)
    n = np.arange(1, 21, dtype=np.float64).reshape((-1,) + (1,) * r.ndim)
    sign = np.where(n % 2 == 1, 1.0, -1.0)
    decay = np.exp(-D * math.pi ** 2 * n ** 2 * t / radius ** 2)
    rs = np.where(r < 1e-10, 1.0, r)                                  # (the centre takes its own formula below)
    inner = Tb + (T0 - Tb) * (2 * radius / (math.pi * rs)) * np.sum(sign / n * np.sin(math.pi * n * rs / radius) * decay, axis=0)
    centre = Tb + (T0 - Tb) * 2 * np.sum(sign * decay, axis=0)
    out = np.where(r >= radius, Tb, np.where(r < 1e-10, centre, inner))
    return float(out) if out.ndim == 0 else out


_J0_ZEROS = None


def radial_heat_xy(x, y, t, center, R=1.0, nterms=1000):
    """examples/2D/Diffusion/Heat.jl:63-100: disc with surface value 1 from the zero state (1000 zeros of J0); NaN outside
    the disc, as there.  Scalars or arrays."""
    global _J0_ZEROS
    from scipy.special import j0, j1, jn_zeros

    if _J0_ZEROS is None or len(_J0_ZEROS) < nterms:
        _J0_ZEROS = jn_zeros(0, nterms)
    al = _J0_ZEROS[:nterms]
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    r = np.hypot(x - center[0], y - center[1])
    inside = r < R
    out = np.full(r.shape, np.nan)
    if np.any(inside):
        ri = r[inside] if r.ndim else r.reshape(1)
        coef = np.exp(-al ** 2 * t) / (al * j1(al))
        coef = coef[coef != 0.0]                                      # (terms that underflow to zero add nothing)
        a = al[: len(coef)]
        vals = np.empty(ri.shape)
        for q in range(0, len(ri), 4096):                             # bounded temporaries: 4096 points x terms
            vals[q:q + 4096] = 1.0 - 2.0 * (j0(np.outer(ri[q:q + 4096] / R, a)) @ coef)
        if r.ndim:
            out[inside] = vals
        else:
            out = vals[0]
    return float(out) if np.ndim(out) == 0 else out


def pairwise_orders(h: Sequence[float], err: Sequence[float]) -> List[float]:
    """log(e_i / e_(i-1)) / log(h_i / h_(i-1)); NaN for the first entry (BenchPhaseFlow/utils/convergence.jl)."""
    out = [float("nan")]
    for i in range(1, len(h)):
        ok = err[i] > 0 and err[i - 1] > 0 and h[i] != h[i - 1]
        out.append(math.log(err[i] / err[i - 1]) / math.log(h[i] / h[i - 1]) if ok else float("nan"))
    return out


def _fit_order(h: Sequence[float], err: Sequence[float]) -> float:
    """slope of log(err) = p log(h) + c (LsqFit.curve_fit of a straight line, benchmark/Heat3D.jl:108-123)."""
    good = [(math.log(a), math.log(b)) for a, b in zip(h, err) if b > 0 and math.isfinite(b)]
    if len(good) < 2:
        return float("nan")
    lx, ly = np.array([g[0] for g in good]), np.array([g[1] for g in good])
    return float(np.polyfit(lx, ly, 1)[0])


def run_mesh_convergence(n_list: Sequence[int], radius: float, center: Sequence[float], u_analytical: Optional[Callable] = None,
                         L: float = 4.0, norm=2, Tend: float = 0.1, output_dir: Optional[str] = None,
                         verbose: bool = False, **solve_kwargs) -> Dict:
    """One heat problem per mesh size through the HIP path, errors with check_convergence, CSV files as the reference.

    N = len(center): 3 -> benchmark/Heat3D.jl (sphere, Dirichlet(1) on the interface and on :left/:right/:top/:bottom,
    T0 = 0, BE constructor then CN, Δt = 0.75 h²); 2 -> examples/2D/Diffusion/Heat.jl with the constant interface value
    its analytic series assumes (circle, Dirichlet(1) interface, Dirichlet(0) borders, T0 = [0; 1], BE, Δt = 0.25 h²).
    `u_analytical(x.., t_reached)`: default = the reference's series evaluated at the time the loop actually reaches."""
    N = len(center)
    assert N in (2, 3)
    h_vals, errs, full, cut, empty, rows = [], [], [], [], [], []
    run_dir = None
    if output_dir is not None:
        run_dir = os.path.join(output_dir, time.strftime("%Y-%m-%d_%H-%M-%S"))
        os.makedirs(run_dir, exist_ok=True)
    for n in n_list:
        mesh = api.Mesh((n,) * N, (L,) * N, (0.0,) * N)
        cap = api.Capacity(api.Sphere(tuple(center), radius), mesh)
        ph = api.Phase(cap, api.DiffusionOps(cap), 0.0, 1.0)
        M = int(np.prod(mesh.ext))
        hx = L / n
        keys = ("left", "right", "top", "bottom")
        if N == 3:
            bcb = api.BorderConditions({k: api.Dirichlet(1.0) for k in keys})
            dt, u0, ctor, scheme = 0.75 * hx ** 2, np.zeros(2 * M), "BE", "CN"          # Heat3D.jl:55-74
        else:
            bcb = api.BorderConditions({k: api.Dirichlet(0.0) for k in keys})
            dt, u0, ctor, scheme = 0.25 * hx ** 2, np.concatenate([np.zeros(M), np.ones(M)]), "BE", "BE"   # Heat.jl:25-48
        s = api.DiffusionUnsteadyMono(ph, bcb, api.Dirichlet(1.0), dt, u0, ctor)
        api.solve_DiffusionUnsteadyMono_b(s, ph, dt, Tend, bcb, api.Dirichlet(1.0), scheme, save_states=False, **solve_kwargs)
        # time the state belongs to: one first solve + the loop's steps, each an implicit step of Δt
        t_reached = dt * (1 + int(s.last_run.steps))
        if u_analytical is not None:
            ua = lambda *x: u_analytical(*x, t_reached)
        elif N == 3:
            ua = lambda x, y, z: heat3d_analytical(x, y, z, t_reached, center, radius)
        else:
            ua = lambda x, y: radial_heat_xy(x, y, t_reached, center, radius)
        _, _, e_all, e_full, e_cut, e_empty = api.check_convergence(ua, s, cap, norm)
        h = 1.0 / n                                                   # "representative mesh size", Heat3D.jl:81
        h_vals.append(h); errs.append(e_all); full.append(e_full); cut.append(e_cut); empty.append(e_empty)
        dims = {"nx": n, "ny": n} if N == 2 else {"nx": n, "ny": n, "nz": n}
        row = {"mesh_size": h, **dims, "global_error": e_all, "full_error": e_full, "cut_error": e_cut, "empty_error": e_empty}
        rows.append(row)
        if verbose:
            print(f"n = {n}: global {e_all:.3e} full {e_full:.3e} cut {e_cut:.3e} (t = {t_reached:.5f}, "
                  f"{s.last_run.total_iters} Krylov iterations)", flush=True)
        if run_dir is not None:
            name = "mesh_" + "x".join(f"{n:04d}" for _ in range(N)) + ".csv"        # Heat3D.jl:99 (@sprintf "mesh_%04dx%04dx%04d.csv")
            _write_csv(os.path.join(run_dir, name), [row])
    orders = {"all": _fit_order(h_vals, errs), "full": _fit_order(h_vals, full), "cut": _fit_order(h_vals, cut)}
    out = {"h_vals": h_vals, "err_vals": errs, "err_full_vals": full, "err_cut_vals": cut, "err_empty_vals": empty,
           "orders": orders, "pair_order_all": pairwise_orders(h_vals, errs), "run_dir": run_dir, "norm": norm}
    if run_dir is not None:
        _write_csv(os.path.join(run_dir, "summary.csv"), rows)                                               # Heat3D.jl:136-153
        _write_csv(os.path.join(run_dir, "convergence_rates.csv"),
                   [{"parameter": f"p_{k}", "value": round(v, 2)} for k, v in (("global", orders["all"]), ("full", orders["full"]),
                                                                                ("cut", orders["cut"]))])
        out["summary_csv"] = os.path.join(run_dir, "summary.csv")
        out["rates_csv"] = os.path.join(run_dir, "convergence_rates.csv")
    return out


def _write_csv(path: str, rows: List[Dict]) -> None:
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def oscillating_disk(radius_mean=1.0, radius_amp=0.5, period=1.0, center=(2.0, 2.0), D=1.0):
    """benchmark/Heat_2d_moving.jl:463-507: the manufactured solution (1 + ½ sin(2πt/T)) cos(πx) cos(πy) inside a disc whose
    radius oscillates, zero outside, and its source term f(x, y, z, t) (z is the slab's time centroid, unused)."""
    R = lambda t: radius_mean + radius_amp * np.sin(2 * np.pi * t / period)
    dR = lambda t: radius_amp * 2 * np.pi / period * np.cos(2 * np.pi * t / period)

    def phi(x, y, t):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        r = np.hypot(x - center[0], y - center[1])
        return np.where(r > R(t), 0.0, (1 + 0.5 * np.sin(2 * np.pi * t / period)) * np.cos(np.pi * x) * np.cos(np.pi * y))

    def source(x, y, z, t):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        r = np.hypot(x - center[0], y - center[1])
        cc = np.cos(np.pi * x) * np.cos(np.pi * y)
        f = (np.pi / period) * cc * np.cos(2 * np.pi * t / period) + 2 * np.pi ** 2 * D * (1 + 0.5 * np.sin(2 * np.pi * t / period)) * cc
        return np.where(r > R(t), 0.0, f)

    return R, dR, phi, source


def run_mesh_convergence_moving(nx_list: Sequence[int], radius_mean: float = 1.0, radius_amp: float = 0.5, period: float = 1.0,
                                center=(2.0, 2.0), D: float = 1.0, lx: float = 4.0, Tend: float = 0.1, norm=2,
                                output_dir: Optional[str] = None, verbose: bool = False, literal: bool = True,
                                **solve_kwargs) -> Dict:
    """benchmark/Heat_2d_moving.jl:11-247 through the HIP moving path: per mesh Δt = ½ (lx/nx)², Tstart = Δt, the space-time
    capacity of [0, Δt] for the constructor, the initial state Φ_ana(nodes, Tstart), Dirichlet(Φ_ana(x, y, t_c)) on the
    interface, Dirichlet(0) borders, "BE"; errors by check_convergence on the STATIC capacity of the disc at Tend; the CSV
    files it writes (config.csv, mesh_NNNNxNNNN.csv, summary.csv with the dt column, convergence_rates.csv).
    `literal=False` removes two first-order artefacts of the benchmark itself (not of the solver): the initial state is
    sampled at the cell centres instead of the nodes (half a cell away from where the unknowns live), and the error is taken
    at the time the loop really reaches (it overshoots Tend by one to two steps)."""
    from . import moving as mv

    R, dR, phi, source = oscillating_disk(radius_mean, radius_amp, period, center, D)
    body = mv.MovingSphere(lambda t: center, R, dcenter=lambda t: (0.0, 0.0), dradius=dR)
    run_dir = None
    if output_dir is not None:
        run_dir = os.path.join(output_dir, time.strftime("%Y-%m-%d_%H-%M-%S"))
        os.makedirs(run_dir, exist_ok=True)
        _write_csv(os.path.join(run_dir, "config.csv"),
                   [{"parameter": k, "value": v} for k, v in (("radius_mean", radius_mean), ("radius_amp", radius_amp), ("period", period),
                                                              ("center_x", center[0]), ("center_y", center[1]), ("D", D), ("Tend", Tend))])
    rows, h_vals, errs, full, cut = [], [], [], [], []
    for nx in nx_list:
        mesh = api.Mesh((nx, nx), (lx, lx), (0.0, 0.0))
        dt = 0.5 * (lx / nx) ** 2
        Tstart = dt
        cap = api.Capacity(body, mv.SpaceTimeMesh(mesh, [0.0, dt]))
        phase = api.Phase(cap, api.DiffusionOps(cap), source, D)
        bc = api.Dirichlet(0.0)
        bcb = api.BorderConditions({k: bc for k in ("left", "right", "top", "bottom")})
        bci = api.Dirichlet(lambda x, y, t: phi(x, y, t))                  # (x, y, t_c): the interface centroid's time
        xs = [mesh.nodes[d] if literal else mesh.nodes[d] + 0.5 * (lx / nx) for d in range(2)]
        X, Y = np.meshgrid(xs[0], xs[1], indexing="ij")                    # idx = (j-1)(nx+1) + i: x fastest
        u0 = np.concatenate([np.asarray(phi(X, Y, Tstart)).ravel(order="F"), np.zeros((nx + 1) ** 2)])
        s = mv.MovingDiffusionUnsteadyMono(phase, bcb, bci, dt, u0, mesh, "BE")
        mv.solve_MovingDiffusionUnsteadyMono_b(s, phase, body, dt, Tstart, Tend, bcb, bci, mesh, "BE", save_states=False, **solve_kwargs)
        # the loop of diffusion.jl:247-266: t = Tstart; while t < Tend: t += Δt; slab [t, t+Δt] -> the state belongs to t + Δt
        t_reached = Tstart
        while t_reached < Tend:
            t_reached += dt
        t_reached += dt
        t_cmp = Tend if literal else t_reached
        cap_end = api.Capacity(api.Sphere(tuple(center), float(R(t_cmp))), mesh, compute_centroids=False)
        _, _, e_all, e_full, e_cut, e_empty = api.check_convergence(lambda x, y: phi(x, y, t_cmp), s, cap_end, norm)
        h = lx / nx
        row = {"mesh_size": h, "nx": nx, "ny": nx, "dt": dt, "global_error": e_all, "full_error": e_full, "cut_error": e_cut,
               "empty_error": e_empty}
        rows.append(row)
        h_vals.append(h); errs.append(e_all); full.append(e_full); cut.append(e_cut)
        if verbose:
            print(f"nx = {nx}: global {e_all:.3e} full {e_full:.3e} cut {e_cut:.3e} ({len(s.ch)} slabs)", flush=True)
        if run_dir is not None:
            _write_csv(os.path.join(run_dir, f"mesh_{nx:04d}x{nx:04d}.csv"), [row])
    orders = {"all": _fit_order(h_vals, errs), "full": _fit_order(h_vals, full), "cut": _fit_order(h_vals, cut)}
    out = {"h_vals": h_vals, "err_vals": errs, "err_full_vals": full, "err_cut_vals": cut, "orders": orders,
           "pair_order_all": pairwise_orders(h_vals, errs), "run_dir": run_dir}
    if run_dir is not None:
        _write_csv(os.path.join(run_dir, "summary.csv"), rows)
        _write_csv(os.path.join(run_dir, "convergence_rates.csv"),
                   [{"parameter": f"p_{k}", "value": round(v, 2)} for k, v in (("global", orders["all"]), ("full", orders["full"]),
                                                                                ("cut", orders["cut"]))])
    return out
