#!/usr/bin/env python
"""The constants of csrc/mhmr_common.h gelu_fast (round 6).  x Phi(x) = max(x, 0) - a Q(a), a = |x|, Q the normal law's upper tail;
log2 Q is smooth and concave, so Q(a) = exp2(P(a)) with a low-degree P.  Weighted least squares, re-weighted towards the minimax of the
ABSOLUTE error of a exp2(P(a)) over [0, 12]; then the float32 evaluation (Horner by fma, one exp2, one fma) is simulated.
Degree 3: 5.5e-5, 4: 6.1e-6 (positive leading coefficient: needs a clamp), 5: 4.4e-7 (shipped: negative leading coefficient)."""
import numpy as np
from scipy.special import erfc
def Q(a): return 0.5*erfc(a/np.sqrt(2))
a = np.linspace(0, 12, 240001)
target = np.log2(Q(a)); h = a*Q(a)
def fit(deg, iters=400):
    ww = h + 1e-12
    A = np.vander(a, deg+1, increasing=True)
    best=None
    for it in range(iters):
        c = np.linalg.lstsq(A*ww[:,None], target*ww, rcond=None)[0]
        P = A@c
        abs_err = np.abs(a*np.exp2(P) - h)
        m = abs_err.max()
        if best is None or m < best[0]: best=(m,c.copy())
        ww = ww*(1+ 2*abs_err/m)**0.5
        ww /= ww.max()
    return best
m,c = fit(5)
print("fp64 fit", m, [repr(float(v)) for v in c])
c32 = c.astype(np.float32)
print("c32", [repr(float(v)) for v in c32])
# fp32 simulation (fma emulated by float64 product+add then round)
def f32(x): return np.asarray(x, dtype=np.float64).astype(np.float32)
def fma(a_,b_,c_): return f32(a_.astype(np.float64)*np.float64(b_) + np.float64(c_)) if np.isscalar(b_) or np.ndim(b_)==0 else f32(a_.astype(np.float64)*b_.astype(np.float64)+c_.astype(np.float64))
x = np.linspace(-12, 12, 2400001).astype(np.float32)
ax = np.abs(x)
p = np.full_like(ax, c32[5])
for k in (4,3,2,1,0):
    p = f32(p.astype(np.float64)*ax.astype(np.float64) + np.float64(c32[k]))
E = f32(np.exp2(p.astype(np.float64)))   # v_exp_f32 ~1 ulp
r = np.maximum(x, np.float32(0))
out = f32(-(ax.astype(np.float64))*E.astype(np.float64) + r.astype(np.float64))
xd = x.astype(np.float64)
ref = 0.5*xd*erfc(-xd/np.sqrt(2))
err = np.abs(out.astype(np.float64)-ref)
print("fp32 sim max abs err", err.max(), "at", x[err.argmax()])
for big in (20., 50., 1e3, 1e6, 1e10, 3e38):
    ab=np.float32(big); p=np.float32(c32[5])
    with np.errstate(all='ignore'):
        for k in (4,3,2,1,0): p = np.float32(np.float64(p)*np.float64(ab)+np.float64(c32[k]))
    print(big, p)
