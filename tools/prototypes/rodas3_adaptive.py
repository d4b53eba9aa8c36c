"""f-4 prototype (numpy, CPU): adaptive Rodas3 (Sandu et al. 1997: 4 stages, order 3(2), stiffly accurate, L-stable) on the
stiff extraction box of BASELINE configs[2], against the LSODA(1e-13) fixtures.  Result:
tol 1e-5 -> 55 steps, max rel. error 1e-3; 1e-6 -> 117, 1e-4; 1e-7 -> 250, 1e-5; 1e-8 -> 540 steps, 1.2e-6 --
against ~70 steps of DOPRI5 for 1e-8: an order-3 linearly implicit pair is not competitive here."""
import numpy as np
import os
g=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'tight_multistage_extraction.npz'))
X,U,XF,dt=g['x'],g['u'],g['xf'],float(g['dt'])
Vl=5.;Vg=5.;m=1.;Kla=5.;X0=0.6;Y6=0.05
def f(x,u):
    L,G=u[0],u[1]; d=np.zeros(10)
    for s in range(5):
        Xs,Ys=x[2*s],x[2*s+1]
        Q=Kla*(Xs-Ys*Ys/m)*Vl
        Xp=X0 if s==0 else x[2*s-2]; Yn=Y6 if s==4 else x[2*s+3]
        d[2*s]=(L*(Xp-Xs)-Q)/Vl; d[2*s+1]=(G*(Yn-Ys)+Q)/Vg
    return d
def jac(x,u):
    L,G=u[0],u[1]; J=np.zeros((10,10))
    for s in range(5):
        Ys=x[2*s+1]
        dQdX=Kla*Vl; dQdY=-Kla*Vl*2*Ys/m
        J[2*s,2*s]=(-L-dQdX)/Vl; J[2*s,2*s+1]=-dQdY/Vl
        if s>0: J[2*s,2*s-2]=L/Vl
        J[2*s+1,2*s+1]=(-G+dQdY)/Vg; J[2*s+1,2*s]=dQdX/Vg
        if s<4: J[2*s+1,2*s+3]=G/Vg
    return J
gam=0.5
A=np.zeros((4,4)); A[2,0]=2; A[3,0]=2; A[3,2]=1
C=np.zeros((4,4)); C[1,0]=4; C[2,0]=1; C[2,1]=-1; C[3,0]=1; C[3,1]=-1; C[3,2]=-8/3
M=np.array([2,0,1,1.]); E=np.array([0,0,0,1.])
def rodas3(x,u,dt,rtol,atol):
    t=0; h=min(dt,0.01/max(np.linalg.norm(f(x,u)),1e-9)*max(np.linalg.norm(x),1e-3)); nacc=nrej=0
    while t<dt*(1-1e-14):
        if t+h>dt: h=dt-t
        J=jac(x,u); W=np.eye(10)/(h*gam)-J
        K=np.zeros((4,10)); fx=None
        for i in range(4):
            if i==1: y=x; fi=K_f0   # ros_NewF[1]=False -> reuse f of stage 0 argument (a21=0)
            else:
                y=x+A[i,:i]@K[:i]; fi=f(y,u)
            if i==0: K_f0=fi
            rhs=fi+(C[i,:i]@K[:i])/h
            K[i]=np.linalg.solve(W,rhs)
        xn=x+M@K; err=E@K
        sc=atol+rtol*np.maximum(np.abs(x),np.abs(xn)); En=np.sqrt(np.mean((err/sc)**2))
        if En<1:
            t+=h; x=xn; nacc+=1
        else: nrej+=1
        fac=min(6,max(0.2,0.9*En**(-1/3))) if En>0 else 6
        h*=fac
    return x,nacc,nrej
for tol in (1e-5,1e-6,1e-7,1e-8):
    errs=[];steps=[]
    for i in range(24):
        xf,na,nr=rodas3(X[i].copy(),U[i],dt,tol,tol)
        errs.append(np.max(np.abs(xf-XF[i])/np.maximum(np.abs(XF[i]),1e-6))); steps.append(na+nr)
    print("tol %.0e  err max %.2e median %.2e   steps mean %.1f max %d"%(tol,max(errs),np.median(errs),np.mean(steps),max(steps)))
