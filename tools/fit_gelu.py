#!/usr/bin/env python
"""Weighted minimax (Lawson) fit of erf(t / sqrt 2) ~ t P(t^2) on [0, c] with q(c) = 1, for the transcendental-free GELU of
csrc/common.h (gelu2p): prints max |gelu error| in fp32 Horner arithmetic and the coefficients per (c, terms)."""
import numpy as np
from scipy.special import erf
def fit(c, ncoef, iters=600):
    t=np.linspace(0,c,40001)[1:]
    s=t/c
    f=erf(t/np.sqrt(2)); f[-1]=1.0
    A=np.stack([s**(2*k+1) for k in range(ncoef)],1)
    wgt=np.maximum(t,0.05)
    w=np.ones_like(t)
    for it in range(iters):
        ww=w*wgt; ww[-1]=1e4*ww.max()
        coef,*_=np.linalg.lstsq(A*ww[:,None],f*ww,rcond=None)
        err=np.abs(A@coef-f)*wgt
        w=w*(1+(err/err.max())); w/=w.mean()
    return coef/np.array([c**(2*k+1) for k in range(ncoef)])
def evalf32(coef,c,x):
    x=x.astype(np.float32)
    t=np.clip(x,-np.float32(c),np.float32(c))
    u=(t*t).astype(np.float32)
    p=np.float32(coef[-1])*np.ones_like(u)
    for k in range(len(coef)-2,-1,-1):
        p=(p*u+np.float32(coef[k])).astype(np.float32)
    q=(t*p).astype(np.float32)
    hx=np.float32(0.5)*x
    return (hx*q+hx).astype(np.float32), q
x=np.linspace(-10,10,500001)
ref=0.5*x*(1+erf(x/np.sqrt(2)))
for c in (3.5,3.6,3.7,3.8,3.9,4.0,4.2):
    for n in (5,6,7):
        coef=fit(c,n)
        g,q=evalf32(coef,c,x)
        e=np.abs(g-ref)
        print(c,n,'max abs err %.2e'%e.max(),'at %.3f'%x[e.argmax()],'qmax=%.7f'%q.max(), 'coef', ','.join('%.9e'%v for v in coef))
