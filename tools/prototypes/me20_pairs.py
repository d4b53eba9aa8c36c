import sys, numpy as np, copy
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import bench
from oracle import oracle as O
from pcgym_amd.config import EnvSpec
_, p_env, _, _, _ = bench.single_workload("me20")
base=EnvSpec(copy.deepcopy(p_env))
rng=np.random.default_rng(3)
B=6000
lo,hi=base.a_low,base.a_high
x0=np.array(base.x0[:base.nx],dtype=float)
x=np.tile(x0[:,None],(1,B))*(1+0.05*rng.uniform(-1,1,(base.nx,B)))
ref=EnvSpec(dict(copy.deepcopy(p_env), integrator="dopri5", rtol=1e-13, atol=1e-13))
X=[];U=[];W=[]
for t in range(3):
    u=lo[:,None]+rng.uniform(0,1,(base.na,B))*(hi-lo)[:,None]
    w,_=O.integrate(ref,x,u); X.append(x);U.append(u);W.append(w); x=w
X=np.concatenate(X,1);U=np.concatenate(U,1);W=np.concatenate(W,1)
for integ in ("dopri5","tsit5"):
    for tol in (1e-8,2e-8,5e-9):
        s=EnvSpec(dict(copy.deepcopy(p_env), integrator=integ, rtol=tol, atol=tol))
        y,ns=O.integrate(s,X,U)
        err=np.max(np.abs(y-W)/(1e-6*np.abs(W)+1e-8))
        rel=np.max(np.abs(y-W)/np.maximum(np.abs(W),1e-4))
        print(f"{integ} tol {tol:.0e}: attempts mean {ns.sum(0).mean():.2f} max {ns.sum(0).max()} rejected {ns[1].mean():.2f}; err in CVODES units {err:.3f}; rel(floor 1e-4) {rel:.2e}", flush=True)
