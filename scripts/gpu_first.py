import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import penguin.jl_amd as pj
from oracle import penguin_oracle as po
from oracle.geometry import Ball
pj.init(0); print(pj.device_name())
# ---- config "Heat Monophasic" 20x20 -------------------------------------------
n=20
mesh = pj.Mesh((n,n),(4.,4.),(0.,0.)); omesh = po.Mesh((n,n),(4.,4.),(0.,0.))
body = pj.Sphere((2.01,2.01),1.0); obody = Ball((2.01,2.01),1.0)
cap = pj.Capacity(body, mesh); ocap = po.make_capacity(obody, omesh)
for name,a,b in [("V",cap.V,ocap.V),("G",cap.Γ,ocap.G),("ct",cap.cell_types,ocap.cell_types),("Cw",cap.C_ω,ocap.C_w),("Cg",cap.C_γ,ocap.C_g)]+[(f"A{d}",cap.A[d],ocap.A[d]) for d in range(2)]+[(f"B{d}",cap.B[d],ocap.B[d]) for d in range(2)]+[(f"W{d}",cap.W[d],ocap.W[d]) for d in range(2)]:
    print(name, "max abs diff", np.max(np.abs(np.asarray(a)-np.asarray(b))))
op = pj.DiffusionOps(cap); oop = po.make_diffusion_ops(ocap)
print("G diff", abs(op.G - oop.G.tocsc()).max(), "H diff", abs(op.H-oop.H.tocsc()).max(), "W diff", abs(op.Winv-oop.Winv.tocsc()).max())
M=(n+1)**2
f=lambda x,y,z,t:0.0; D=lambda x,y,z:1.0
ph = pj.Phase(cap, op, f, D); oph = po.Phase(ocap, oop, f, D)
bcb = pj.BorderConditions({k: pj.Dirichlet(0.0) for k in ("left","right","top","bottom")})
obcb = po.BorderConditions({k: po.Dirichlet(0.0) for k in ("left","right","top","bottom")})
g = lambda x,y,z,t: np.sin(np.pi*x)*np.sin(np.pi*y)
u0 = np.concatenate([np.zeros(M), np.ones(M)])
dt = 0.25*(4/n)**2
for scheme in ("BE","CN"):
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(g), dt, u0, "BE")
    A0,b0,idx0 = s.system(0)
    os_ = po.DiffusionUnsteadyMono(oph, obcb, po.Dirichlet(g), dt, u0, "BE")
    Ar, br, oidx = po.remove_zero_rows_cols(os_.A, os_.b)
    print(scheme, "n", len(idx0), len(oidx), "idx equal", np.array_equal(idx0, oidx))
    if np.array_equal(idx0, oidx):
        print("  A diff", abs(A0[:, :len(idx0)]-Ar).max(), "b diff", np.max(np.abs(b0-br)))
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 0.01, bcb, pj.Dirichlet(g), scheme, reltol=1e-13)
    po.solve_DiffusionUnsteadyMono(os_, oph, dt, 0.01, obcb, po.Dirichlet(g), scheme, method="\\")
    print("  states", len(s.states), len(os_.states), "rel L2", np.linalg.norm(s.x-os_.x)/np.linalg.norm(os_.x), "max ug", s.x[M:].max())
