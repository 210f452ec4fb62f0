"""A run that goes on until the field is (nearly) steady: late solves meet the tolerance at their start or need the smallest
polynomial only -- the regime in which the extrapolated state is written by the first s kernel's early exit, the degree
estimate sits at its floor and the fit runs every eighth step.  Writes the end state and the run's counters to OUT.npz.
    python scripts/steady_state_sequence.py OUT [n=32] [steps=900] [scheme=BE]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 900
scheme = sys.argv[4] if len(sys.argv) > 4 else "BE"
pj.init(0)
lib = L.lib()
mesh = pj.Mesh((n,) * 3, (4.0,) * 3)
cap = pj.Capacity(pj.Sphere((2.01,) * 3, 1.0), mesh)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
dt = 0.75 * (4.0 / n) ** 2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, 1)
si = L.pg_step_info()
L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
run = L.pg_run_info()
L.check(lib.pg_solver_run(s._h, C.c_double(1e30), C.c_int32(1 if scheme == "CN" else 0), C.byref(opts), 0, C.c_int64(steps), 0, C.byref(run)))
x = s._fetch_state()
np.savez(out, x=x, products=run.products, steps=run.steps, unconverged=run.unconverged_steps, kept=s.guess_info()["kept"],
         states_read=run.guess_states_read)
print("products/step", run.products / run.steps, "unconverged", run.unconverged_steps, "max", float(np.max(x)))
