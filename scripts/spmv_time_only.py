"""Time ONLY the plain SpMV of the 512^3 system (no solves: safe for ablation builds whose results are wrong)."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pj.init(0)
mesh = pj.Mesh((n, n, n), (4., 4., 4.))
cap = pj.Capacity(pj.Sphere((2.01, 2.01, 2.01), 1.0), mesh)
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), 0.75 * (4.0 / n) ** 2, None, "BE")
info = s.system_info(0)
b = 12 * info.nnz + 20 * info.n_own
ms = C.c_double()
import os
for mode in [int(m) for m in os.environ.get("PG_TIME_MODES", "0").split(",")]:
    for rep in range(2):
        L.check(L.lib().pg_solver_time_spmv(s._h, 0 | (mode << 4), 50, C.byref(ms)))
        print(f"spmv mode {mode} {ms.value:.4f} ms = {b/ms.value/1e6:.0f} GB/s ({b/ms.value/1e6/80:.1f}% of 8 TB/s)", flush=True)
