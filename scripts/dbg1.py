import sys, numpy as np
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from oracle import penguin_oracle as po
from tests.common import oracle_capacity_from_product, rel_l2
import scipy.sparse.linalg as spla
pj.init(0)
n=16; M=(n+1)**2
u0=np.zeros(2*M); dt=0.25*(4/n)**2
mesh=pj.Mesh((n,n),(4.,4.)); omesh=po.Mesh((n,n),(4.,4.))
cap=pj.Capacity(pj.Sphere((2.,2.),3.5),mesh); ocap=oracle_capacity_from_product(cap,omesh)
f=lambda x,y,z,t:0.0; D=lambda x,y,z:1.0
ph=pj.Phase(cap,pj.DiffusionOps(cap),f,D); oph=po.Phase(ocap,po.make_diffusion_ops(ocap),f,D)
b1={"left":pj.Dirichlet(1.0),"right":pj.Dirichlet(0.0)}; b2={"left":po.Dirichlet(1.0),"right":po.Dirichlet(0.0)}
s=pj.DiffusionUnsteadyMono(ph,pj.BorderConditions(b1),pj.Dirichlet(0.5),dt,u0,"BE")
so=po.DiffusionUnsteadyMono(oph,po.BorderConditions(b2),po.Dirichlet(0.5),dt,u0,"BE")
A,b,idx=s.system(0)
A=A[:,:len(idx)]
x_lu=spla.spsolve(A.tocsc(),b)
pj.solve_DiffusionUnsteadyMono_b(s,ph,dt,0*dt,pj.BorderConditions(b1),pj.Dirichlet(0.5),"BE",reltol=1e-13,log=True)
print(s.ch)
x0=s.states[0][idx]
print("first solve vs LU of own system:", rel_l2(x0,x_lu), "resid", np.linalg.norm(A@x0-b)/np.linalg.norm(b))
d=A.diagonal(); print("diag min/max", np.abs(d).min(), np.abs(d).max(), "zeros on diag", np.sum(d==0))
import scipy.sparse as sp
Ds=sp.diags(1/np.sqrt(np.abs(np.where(d==0,1,d)))); As=(Ds@A@Ds).tocsr()
y,it,rn=po.bicgstab_ref(As,Ds@b,reltol=1e-13); print("oracle bicgstab on equilibrated:", it, rel_l2(Ds@y,x_lu))
