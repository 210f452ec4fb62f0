"""Measurement of the prescribed-motion path: slabs per second of solve_MovingDiffusionUnsteadyMono! on a 2-D
mesh, with the time of each phase of a slab (space-time capacity kernels, host closures, assembly, Krylov solve).

    python scripts/moving_bench.py [n=1024] [slabs=10] [scheme=BE]
The problem is examples/2D/SolidMoving/MovingHeat.jl's (a disc growing like R0 + c sqrt(t) in a 16 x 16 box, fluid outside,
Dirichlet(1) on the interface, Dirichlet(0) borders), refined."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
from penguin.jl_amd import moving as mv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
slabs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
scheme = sys.argv[3] if len(sys.argv) > 3 else "BE"
pj.init(0)
lx = 16.0
mesh = pj.Mesh((n, n), (lx, lx), (-8.0, -8.0))
c = 1.56
body = pj.MovingSphere(lambda t: (0.0, 0.0), lambda t: 1.0 + c * np.sqrt(t + 0.01), complement=True,
                       dcenter=lambda t: (0.0, 0.0), dradius=lambda t: 0.5 * c / np.sqrt(t + 0.01))
dt = 1.0 * (lx / n) ** 2                     # examples/2D/SolidMoving/MovingHeat.jl:39
M = (n + 1) ** 2
bc = pj.Dirichlet(1.0)
bcb = pj.BorderConditions({k: pj.Dirichlet(0.0) for k in ("left", "right", "top", "bottom")})
T = {"capacity_kernels_ms": 0.0, "capacity_call_s": 0.0, "create_solver_s": 0.0, "solve_s": 0.0, "state_fetch_s": 0.0}


DEVICE_STATE = len(sys.argv) > 4 and sys.argv[4] == "device"     # hand the state over on the device, fetch nothing
prev = None


def slab(t, Ti, first=False):
    global prev
    t0 = time.perf_counter()
    cap = pj.Capacity(body, pj.SpaceTimeMesh(mesh, [t, t + dt]))
    T["capacity_kernels_ms"] += cap.kernel_ms
    t1 = time.perf_counter()
    ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
    s = pj.MovingDiffusionUnsteadyMono(ph, bcb, bc, dt, Ti, mesh, scheme) if first else None
    if not first:
        if DEVICE_STATE:
            s = prev
            mv._create_step(s, ph, bcb, bc, dt, None, mesh, scheme, t, from_previous=True)
        else:
            s = pj.Solver("Unsteady", "Monophasic", "Diffusion")
            s._nunk = 2 * M
            mv._create_step(s, ph, bcb, bc, dt, Ti, mesh, scheme, t)
    L.check(L.lib().pg_device_synchronize())
    t2 = time.perf_counter()
    info = L.pg_step_info()
    opts = pj.api._krylov_opts("bicgstab", {})
    L.check(L.lib().pg_solver_initial_solve(s._h, __import__("ctypes").byref(opts), __import__("ctypes").byref(info)))
    L.check(L.lib().pg_device_synchronize())
    t3 = time.perf_counter()
    x = None if DEVICE_STATE else s._fetch_state(-1)
    prev = s
    t4 = time.perf_counter()
    T["capacity_call_s"] += t1 - t0
    T["create_solver_s"] += t2 - t1
    T["solve_s"] += t3 - t2
    T["state_fetch_s"] += t4 - t3
    assert info.converged
    return x, info.iters, s.system_info(0).n_own


x, _, _ = slab(0.0, np.concatenate([np.zeros(M), np.ones(M)]), True)     # warm-up slab (first-use allocations)
for k in T:
    T[k] = 0.0
t_all = time.perf_counter()
its = []
t = 0.0
for k in range(slabs):
    t += dt
    x, it, rows = slab(t, x)
    its.append(it)
wall = time.perf_counter() - t_all
out = {"what": "prescribed-motion diffusion, 2-D growing disc (fluid outside), one space-time slab per step", "n": n, "cells": M,
       "rows_last_slab": int(rows), "scheme": scheme, "state_between_slabs": "device" if DEVICE_STATE else "host (the reference's push!)", "slabs": slabs, "slabs_per_s": slabs / wall, "ms_per_slab": wall / slabs * 1e3,
       "krylov_iters_per_slab": float(np.mean(its)),
       "per_slab_ms": {k: (v if k.endswith("_ms") else v * 1e3) / slabs for k, v in T.items()},
       "capacity_cells_per_s": M * slabs / (T["capacity_kernels_ms"] * 1e-3), "time_nodes": 64, "device": pj.device_name()}
if x is not None:      # (a checksum of the last state: runs with different allocator settings must agree on it)
    out["state_l2"] = float(np.linalg.norm(x))
    out["state_max"] = float(np.max(np.abs(x)))
    out["state_finite"] = bool(np.all(np.isfinite(x)))
print(json.dumps(out))
