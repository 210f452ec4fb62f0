"""Host experiment (numpy, 48^3 model problem with the benchmark's spectrum, g = 0.72): right-preconditioned BiCGStab in x-space with
the Chebyshev-Horner chain stored in fp32 (vectors rounded after every Horner step) against fp64 chains.  Result (DESIGN.md
"What comes next"): fp32 chains follow the fp64 history down to ~7 digits per application -- 2 applications of degree 14 reach
1e-12 (28 products) where one fp64 application of degree 26 does."""
import numpy as np, scipy.sparse as sp
np.random.seed(1)
n=48
I=sp.identity(n); e=np.ones(n)
T=sp.diags([-e[:-1],2*e,-e[:-1]],[-1,0,1])
K=sp.kron(sp.kron(T,I),I)+sp.kron(sp.kron(I,T),I)+sp.kron(sp.kron(I,I),T)
# CN: A = I + theta K with spectrum ratio like g=0.72 after scaling: A_hat = A/diag; eigen in [1-g,1+g]
# A = I + c K; diag = 1+6c ; offdiag sum = 6c -> g = 6c/(1+6c) = 0.72 -> c = 0.72/(6*0.28)
c=0.72/(6*0.28)
A=(sp.identity(n**3)+c*K).tocsr(); d=A.diagonal(); A=sp.diags(1/np.sqrt(d))@A@sp.diags(1/np.sqrt(d)); A=A.tocsr()
g=0.72
def taus(m):
    lam=[1+g*np.cos(np.pi*(2*k+1)/(2*m)) for k in range(m)]
    out=[];lo=0;hi=m-1
    for k in range(m):
        if k&1: out.append(1/lam[hi]);hi-=1
        else: out.append(1/lam[lo]);lo+=1
    return out
def Minv(p,m,f32):
    tau=taus(m)
    cast=(lambda v:v.astype(np.float32).astype(np.float64)) if f32 else (lambda v:v)
    base=cast(p)
    u=cast(tau[m-1]*base)
    for k in range(m-2,-1,-1):
        u=tau[k]*base+u-tau[k]*(A@u)
        if k>0: u=cast(u)
    return u
def bicgstab(b,x,m,f32,tol=1e-12,maxapp=10):
    r=b-A@x; rhat=r.copy(); p=r.copy(); rho=rhat@r; bn=np.linalg.norm(b); apps=0; hist=[np.linalg.norm(r)/bn]
    while apps<maxapp:
        ph=Minv(p,m,f32); v=A@ph; apps+=1
        alpha=rho/(rhat@v); s=r-alpha*v
        hist.append(np.linalg.norm(s)/bn)
        if np.linalg.norm(s)<=tol*bn: x=x+alpha*ph; return x,apps,hist
        sh=Minv(s,m,f32); t=A@sh; apps+=1
        omega=(t@s)/(t@t); x=x+alpha*ph+omega*sh; r=s-omega*t
        hist.append(np.linalg.norm(r)/bn)
        if np.linalg.norm(r)<=tol*bn: return x,apps,hist
        rho_new=rhat@r; beta=(rho_new/rho)*(alpha/omega); rho=rho_new; p=r+beta*(p-omega*v)
    return x,apps,hist
xs=np.random.rand(n**3); b=A@xs
# warm start: x0 = solution + smooth-ish error of relative size 1e-2
err=np.random.rand(n**3); x0=xs+1e-2*err*np.linalg.norm(xs)/np.linalg.norm(err)
for m,f32 in [(26,False),(26,True),(13,False),(13,True),(14,True),(16,True),(10,True),(9,True)]:
    x,apps,hist=bicgstab(b,x0.copy(),m,f32)
    print(m,'f32' if f32 else 'f64','apps',apps,'products',apps*m,'hist',['%.1e'%h for h in hist],'true',np.linalg.norm(b-A@x)/np.linalg.norm(b))
