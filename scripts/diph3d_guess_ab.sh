#!/bin/bash
# 3-D diphasic 64^3 (plain BiCGStab loop), 40 CN steps with and without the extrapolated start: same end state?
# (RELTOL=1e-14 bash ...: the reference run; this problem amplifies tolerance-level differences from step to step)
for g in 0 4; do
PG_GUESS_STATES=$g python - <<'PY'
import sys, time, ctypes as C, numpy as np, os
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
pj.init(0); lib = L.lib()
n, Lx, c, r = 64, 4.0, (2.01, 2.01, 2.01), 1.0
M = (n + 1) ** 3
mesh = pj.Mesh((n,) * 3, (Lx,) * 3)
cap1, cap2 = pj.Capacity(pj.Sphere(c, r), mesh), pj.Capacity(pj.Sphere(c, r, complement=True), mesh)
p1, p2 = pj.Phase(cap1, pj.DiffusionOps(cap1), 0.0, 1.0), pj.Phase(cap2, pj.DiffusionOps(cap2), 0.0, 2.0)
ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 0.5, 0.0), pj.FluxJump(1.0, 1.0, 0.0))
bcb = pj.BorderConditions({k: pj.Dirichlet(0.0) for k in ("left", "right", "top", "bottom")})
dt = 0.5 * (Lx / n) ** 2
u0 = np.concatenate([np.ones(M), np.ones(M), np.zeros(M), np.zeros(M)])
s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, u0, "BE")
opts = L.pg_krylov_opts(0, float(os.environ.get('RELTOL', '1e-12')), 0.0, 0, 4, 1); si = L.pg_step_info()
L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
run = L.pg_run_info()
L.check(lib.pg_device_synchronize()); t0 = time.perf_counter()
L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(1), C.byref(opts), 0, C.c_int64(40), 0, C.byref(run)))
L.check(lib.pg_device_synchronize()); el = time.perf_counter() - t0
x = s._fetch_state()
print("PG_GUESS_STATES", os.environ["PG_GUESS_STATES"], "reltol", os.environ.get("RELTOL", "1e-12"), "steps/s %.1f" % (40 / el), "iters/step %.2f" % (run.total_iters / run.steps), "products/step %.2f" % (run.products / run.steps),
      "states/step %.2f" % (run.guess_states_read / run.steps), "compact", s.system_info(1).loop_is_compact, "unconverged", run.unconverged_steps, "guess", s.guess_info()["offsets"],
      "proj %.13e %.13e" % (float(np.sum(x * np.cos(np.arange(x.size) * 1e-3))), float(np.linalg.norm(x))))
PY
done
