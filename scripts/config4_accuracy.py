"""How accurate are the states of the 512^3 run (config 4) for a given Krylov tolerance and iteration?  Pairwise rel-L2
between runs of the same 1 + 3 solves: default loop (warm start + polynomial preconditioner) and IterativeSolvers' plain
cold-start iteration, at reltol 1e-12 and 1e-14 (the tight runs serve as the reference).

    python scripts/config4_accuracy.py [n=512] [steps=3]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
import penguin.jl_amd as pj

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pj.init(0)
mesh = pj.Mesh((n,) * 3, (4.0,) * 3)
cap = pj.Capacity(pj.Sphere((2.01,) * 3, 1.0), mesh)
keys = ("left", "right", "top", "bottom")
dt = 0.75 * (4.0 / n) ** 2


def run(**kw):
    ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
    bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in keys})
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 1e30, bcb, pj.Dirichlet(1.0), "CN", save_states=False, max_steps=steps, **kw)
    return s.x, int(s.last_run.total_iters), float(s.last_run.worst_relres)


runs = {
    "default_1e-12": dict(reltol=1e-12),
    "plain_cold_1e-12": dict(reltol=1e-12, warm_start=False, precond=-1),
    "default_1e-14": dict(reltol=1e-14),
    "plain_cold_1e-14": dict(reltol=1e-14, warm_start=False, precond=-1),
    "plain_warm_1e-12": dict(reltol=1e-12, precond=-1),
    "precond_cold_1e-12": dict(reltol=1e-12, warm_start=False),
}
xs = {}
for k, kw in runs.items():
    x, it, wr = run(**kw)
    xs[k] = x
    print(f"{k:22s} iters {it:4d} worst ||r||/||b|| {wr:.2e}  ||x|| {np.linalg.norm(x):.6e}", flush=True)
ref = xs["default_1e-14"]
nr = np.linalg.norm(ref)
for k, x in xs.items():
    d = x - ref
    print(f"{k:22s} vs default_1e-14: rel-L2 {np.linalg.norm(d) / nr:.3e}  max|d| {np.max(np.abs(d)):.3e} at {int(np.argmax(np.abs(d)))}")
d = xs["plain_cold_1e-14"] - xs["default_1e-14"]
print(f"tight pair: rel-L2 {np.linalg.norm(d) / nr:.3e} max|d| {np.max(np.abs(d)):.3e}")
