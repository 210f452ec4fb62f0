"""Where do the HIP capacities and the oracle's differ on geometries that pass exactly through mesh nodes?"""
import sys
import numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from oracle import penguin_oracle as po
from oracle.geometry import Ball

pj.init(0)
CASES = [
    (1, (8,), (1.0,), (0.0,), (0.5625,), 0.25, False),
    (2, (16, 16), (4.0, 4.0), (0.0, 0.0), (2.125, 2.125), 1.0, False),
    (2, (16, 16), (4.0, 4.0), (0.0, 0.0), (2.125, 2.0), 0.875, True),
]
for N, n, L, x0, c, r, comp in CASES:
    mesh, omesh = pj.Mesh(n, L, x0), po.Mesh(n, L, x0)
    cap = pj.Capacity(pj.Sphere(c, r, complement=comp), mesh)
    ocap = po.make_capacity(Ball(c, r, complement=comp), omesh)
    ext = tuple(v + 1 for v in n)
    print("case", N, n, c, r, comp)
    fields = [("V", cap.V, ocap.V), ("G", cap.Γ, ocap.G)] + [(f"A{d}", cap.A[d], ocap.A[d]) for d in range(N)] + \
             [(f"B{d}", cap.B[d], ocap.B[d]) for d in range(N)] + [(f"W{d}", cap.W[d], ocap.W[d]) for d in range(N)] + \
             [(f"Cw{d}", cap.C_ω[:, d], ocap.C_w[:, d]) for d in range(N)]
    for name, a, b in fields:
        bad = np.flatnonzero(np.abs(a - b) > 1e-11)
        for i in bad[:6]:
            idx = np.unravel_index(i, ext[::-1])[::-1]
            print(f"  {name}[{idx}] hip={a[i]!r} oracle={b[i]!r} type={cap.cell_types[i]} V={cap.V[i]!r}")
        if len(bad) > 6:
            print(f"  {name}: {len(bad)} cells differ")
