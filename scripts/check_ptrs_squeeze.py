#!/usr/bin/env python3
"""Error of the fp32 squeeze in front of the PTRS log-density comparison (k_prep.h ptrs_squeeze) against the fp64
value, over the undecided trials of the sampler for lam = 10 .. 4e6: prints the largest error, the error relative
to 1 + d^2/lam (the kernel's margin is 2e-4 of that) and the number of wrong decisions (must be 0).

    python scripts/check_ptrs_squeeze.py
"""
import numpy as np
from scipy.special import gammaln
f=np.float32
rng=np.random.default_rng(5)
n=4_000_000
lam=np.exp(rng.uniform(np.log(10),np.log(4e6),n))
slam=np.sqrt(lam); b=0.931+2.53*slam; a=-0.059+0.02483*b
invalpha=1.1239+1.1328/(b-3.4); vr=0.9277-3.6224/(b-2)
U=rng.uniform(0,1,n)-0.5; V=rng.uniform(0,1,n)
us=0.5-np.abs(U)
k=np.floor((2*a/us+b)*U+lam+0.43)
und=~((us>=0.07)&(V<=vr)) & ~((k<0)|((us<0.013)&(V>us)))
lam,b,a,invalpha,us,V,k=[v[und] for v in (lam,b,a,invalpha,us,V,k)]
print("undecided", und.mean())
D64=np.log(V)+np.log(invalpha)-np.log(a/(us*us)+b) - (-lam+k*np.log(lam)-gammaln(k+1))
# fp32 squeeze
d=(k-lam)
ok=(k>=10)&(np.abs(d)<=0.25*lam)
d32=d.astype(f); lam32=lam.astype(f); k32=k.astype(f)
x=(d32*(f(1)/lam32)).astype(f)
# g(x)/x^2 = -(1/2 - x/3 + x^2/4 - ...), 14 terms
N=14
p=np.full(x.shape, f((-1)**(N+2)/(N+1)), dtype=f)  # coefficient of x^(N-1) in series sum_{j>=2} (-1)^(j+1) x^j/j divided by x^2
for j in range(N,1,-1):
    c=f((-1)**(j+1)/j)
    p=(p*x+c).astype(f)
g=(p*x*x).astype(f)          # ln(1+x)-x
kinv=(f(1)/k32)
rhs32=(-(d32*d32)*(f(1)/lam32) - k32*g - f(0.5)*np.log((f(6.283185307179586)*k32).astype(f)).astype(f) - kinv*(f(1/12.)-kinv*kinv*f(1/360.))).astype(f)
den=(a/(us*us)+b)
lhs32=np.log((V*invalpha/den).astype(f)).astype(f)
D32=(lhs32-rhs32).astype(np.float64)
err=np.abs(D32-D64)[ok]
print("usable frac", ok.mean(), "max err", err.max(), "99.99%", np.quantile(err,0.9999))
for lo,hi in [(10,100),(100,1e3),(1e3,1e4),(1e4,1e5),(1e5,1e6),(1e6,4e6)]:
    m=ok&(lam>=lo)&(lam<hi)
    print(lo,hi,m.sum(), np.abs(D32-D64)[m].max())
eps=1e-3
dec=ok&(np.abs(D32)>=eps)
print("decided by squeeze", dec.mean(), "wrong", ((D32[dec]<=0)!=(D64[dec]<=0)).sum())
s=1+ (d*d/lam)
r=(np.abs(D32-D64)/s)[ok]
print("max err/scale", r.max(), np.quantile(r,0.99999))
eps=2e-4*s
dec=ok&(np.abs(D32)>=eps)
print("decided", dec.mean(), "wrong", ((D32[dec]<=0)!=(D64[dec]<=0)).sum(), "min margin ratio", (eps[ok]/np.maximum(np.abs(D32-D64)[ok],1e-30)).min())
