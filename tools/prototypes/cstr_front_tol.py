"""What tolerance of the adaptive pair keeps a cstr env that ignites inside its step within 3 x the reference's CVODES tolerances
(1e-6 |x| + 1e-8) of a 1e-13 solve, by env step size: the observation box U(0.7,1) x U(310,350) K over episodes and the wide box
of tests/test_erk.py.  CPU only (oracle).  -> config.cstr_default_tol.   python tools/prototypes/cstr_front_tol.py"""
import sys, numpy as np, copy
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import oracle as O
import scenarios as SC
from pcgym_amd.config import EnvSpec
def _spec(name, **kw):
    p=copy.deepcopy(SC.scenarios()[name]["env_params"]); p.update(kw); return EnvSpec(p)
def rule(dt, c): return min(1e-8, max(1e-10, c*(1/60)/dt))
rng=np.random.default_rng(0)
for tsim,steps in ((1.0,100),(5.0,40),(26.0,12),(60.0,8)):
    B=20000
    ref=_spec("cstr_canonical", integrator="dopri5", rtol=1e-13, atol=1e-13, tsim=tsim)
    dt=ref.dt
    x=np.stack([rng.uniform(0.7,1.0,B), rng.uniform(310,350,B)])
    X=[];U=[];W=[]
    for t in range(steps):
        u=rng.uniform(295,302,(1,B)); want,_=O.integrate(ref,x,u); X.append(x);U.append(u);W.append(want); x=want
    X=np.concatenate(X,1);U=np.concatenate(U,1);W=np.concatenate(W,1)
    # wide box too
    Bw=20000
    xw=np.stack([rng.uniform(0.0,1.2,Bw)**2, rng.uniform(290,600,Bw)]); uw=rng.uniform(280,320,(1,Bw)); ww,_=O.integrate(ref,xw,uw)
    for c in (1e-8, 5e-9):
        tol=rule(dt,c)
        plan=_spec("cstr_canonical", integrator="tsit5g", tsim=tsim, rtol=tol, atol=tol)
        got,ns=O.integrate(plan,X,U); a=ns.sum(0); esc=a>0
        sc=np.max(np.abs(got-W)/(1e-6*np.abs(W)+1e-8),axis=0)
        gw,nw=O.integrate(plan,xw,uw); scw=np.max(np.abs(gw-ww)/(1e-6*np.abs(ww)+1e-8),axis=0); ok=np.isfinite(scw)
        print(f"dt {dt:.4f} c {c:.0e} tol {tol:.2e}: box: attempts max {a.max()} scaled max {sc[esc].max():.2f} | wide box: attempts max {nw.sum(0).max()} scaled max {scw[ok].max():.2f}", flush=True)
