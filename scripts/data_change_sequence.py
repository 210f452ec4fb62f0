"""A run whose data change in the middle (border values 1 -> 1.5 after 14 steps, back after 28): quiet steps with the
extrapolated start, a step with new data (the kept states are dropped, the rows alone on their diagonal move), quiet steps
again.  Writes the states after 14, 15, 28, 29 and 42 steps to OUT.npz.
    python scripts/data_change_sequence.py OUT [n=32]          (tests/test_gpu_parity.py runs it with PG_GUESS_STATES=0 and default)"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
from penguin.jl_amd.api import _border_values

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pj.init(0)
lib = L.lib()
mesh = pj.Mesh((n,) * 3, (4.0,) * 3)
cap = pj.Capacity(pj.Sphere((2.01,) * 3, 1.0), mesh)
keys = ("left", "right", "top", "bottom")
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in keys})
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
dt = 0.75 * (4.0 / n) ** 2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
opts = L.pg_krylov_opts(0, 1e-13, 0.0, 0, 4, 1)
si = L.pg_step_info()
L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
run = L.pg_run_info()
states, info = {}, {}


def advance(k, tag):
    L.check(lib.pg_solver_run(s._h, C.c_double(1e30), C.c_int32(1), C.byref(opts), 0, C.c_int64(k), 0, C.byref(run)))
    assert run.unconverged_steps == 0
    states[tag] = s._fetch_state().copy()
    info[tag] = s.guess_info()["kept"]


advance(14, "s14")
L.check(lib.pg_solver_set_border_values(s._h, L.dptr(_border_values(pj.BorderConditions({k: pj.Dirichlet(1.5) for k in keys}), mesh, 0.0))))
advance(1, "s15")
advance(13, "s28")
L.check(lib.pg_solver_set_border_values(s._h, L.dptr(_border_values(bcb, mesh, 0.0))))
advance(1, "s29")
advance(13, "s42")
np.savez(out, **states, kept=np.array([info[k] for k in ("s14", "s15", "s28", "s29", "s42")]))
print("kept", [info[k] for k in ("s14", "s15", "s28", "s29", "s42")])
