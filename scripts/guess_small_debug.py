"""Diagnostic: the extrapolated start on a small problem, step by step (python scripts/guess_small_debug.py [n] [scheme])."""
import sys
import numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj

pj.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
scheme = sys.argv[2] if len(sys.argv) > 2 else "CN"
M = (n + 1) ** 3
dt = 0.75 * (4.0 / n) ** 2
mesh = pj.Mesh((n,) * 3, (4.0,) * 3)
cap = pj.Capacity(pj.Sphere((2.01,) * 3, 1.0), mesh)
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, np.zeros(2 * M), "BE")
for k in range(1, 16):
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, k * dt, bcb, pj.Dirichlet(1.0), scheme, reltol=1e-13, save_states=True) if k == 1 else None
    break
print("after first call:", s.guess_info(), "info1", [(f, getattr(s.system_info(1), f)) for f in ("n_own", "loop_is_compact")])
s2 = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, np.zeros(2 * M), "BE")
pj.solve_DiffusionUnsteadyMono_b(s2, ph, dt, 20 * dt, bcb, pj.Dirichlet(1.0), scheme, reltol=1e-13, save_states=True)
print("20 steps, states saved:", s2.guess_info())
s3 = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, np.zeros(2 * M), "BE")
pj.solve_DiffusionUnsteadyMono_b(s3, ph, dt, 20 * dt, bcb, pj.Dirichlet(1.0), scheme, reltol=1e-13, save_states=False)
print("20 steps, device loop:", s3.guess_info())
print("difference of the two end states:", np.linalg.norm(s2.x - s3.x) / np.linalg.norm(s3.x))
