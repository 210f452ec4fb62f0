"""Long run to (near) steady state with and without the extrapolated start: every solve converged, same end state.
    python scripts/guess_long_run.py [n=64] [steps=3000] [scheme=CN] [reltol=1e-12]     (run once per setting of PG_GUESS_STATES; prints a checksum)"""
import sys, time, ctypes as C, numpy as np, os
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
scheme = sys.argv[3] if len(sys.argv) > 3 else "CN"
reltol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-12
pj.init(0); lib = L.lib()
mesh = pj.Mesh((n,) * 3, (4.0,) * 3)
cap = pj.Capacity(pj.Sphere((2.01,) * 3, 1.0), mesh)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
dt = 0.75 * (4.0 / n) ** 2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
opts = L.pg_krylov_opts(0, reltol, 0.0, 0, 4, 1); si = L.pg_step_info()
L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
run = L.pg_run_info()
t0 = time.perf_counter()
L.check(lib.pg_solver_run(s._h, C.c_double(1e30), C.c_int32(1 if scheme == "CN" else 0), C.byref(opts), 0, C.c_int64(steps), 0, C.byref(run)))
el = time.perf_counter() - t0
x = s._fetch_state()
w = np.cos(np.arange(x.size) * 1e-3)
w2 = np.sin(np.arange(x.size) * 7.3e-4 + 0.3)
print("projections %.15e %.15e norm %.15e" % (float(x @ w), float(x @ w2), float(np.linalg.norm(x))))
print("PG_GUESS_STATES", os.environ.get("PG_GUESS_STATES", "default"), scheme, "n", n, "steps", run.steps, "steps/s %.0f" % (run.steps / el),
      "products/step %.2f" % (run.products / run.steps), "unconverged", run.unconverged_steps, "worst relres %.2e" % run.worst_relres,
      "finite", bool(np.all(np.isfinite(x))), "max %.15f" % float(np.max(x)), "checksum %.13e" % float(x @ w), "guess", s.guess_info())
