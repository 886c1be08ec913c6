import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
Phi=lambda x:0.5*(1+erf(x/np.sqrt(2)))
phi=lambda x:np.exp(-x*x/2)/np.sqrt(2*np.pi)
def fit(deg,A,wd=0.3,iters=60):
    a=np.linspace(0,A,8001)
    tgt=a*Phi(-a); tdg=Phi(-a)-a*phi(a)
    aa=np.linspace(0,min(A,5),2000)
    c=np.polyfit(aa,np.log2(Phi(-aa)),deg)
    w1=np.ones_like(a); w2=np.ones_like(a)
    def errs(c):
        Pt=2.0**np.polyval(c,a)
        dP=np.polyval(np.polyder(c),a)
        T=Pt*(1+np.log(2)*a*dP)
        return a*Pt-tgt, T-tdg
    for it in range(iters):   # Lawson-style reweighting towards minimax
        def res(c):
            r1,r2=errs(c); return np.concatenate([w1*r1, wd*w2*r2])
        c=least_squares(res,c,method='lm',xtol=1e-15,ftol=1e-15).x
        r1,r2=errs(c)
        w1=w1*(0.5+np.abs(r1)/np.abs(r1).max()); w1/=w1.mean()
        w2=w2*(0.5+np.abs(r2)/np.abs(r2).max()); w2/=w2.mean()
    r1,r2=errs(c)
    return c,np.abs(r1).max(),np.abs(r2).max()
for deg in (3,4,5):
    for A in (5.0,5.5,6.0):
        for wd in (0.1,0.3):
            c,e1,e2=fit(deg,A,wd)
            # tail beyond clamp: error of holding a=A
            tail=max(A*Phi(-A), abs(Phi(-A)-A*phi(A)))
            print(deg,A,wd,'g %.2e dg %.2e tail %.1e'%(e1,e2,tail), np.array2string(c,precision=9))
