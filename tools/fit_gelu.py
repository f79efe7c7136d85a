"""Coefficients of the one-exponential GELU of the mixed-precision kernels (csrc/common.h: phi_fast2 / gelu_fast2 /
gelu_fast_grad2).

Phi(x) = 0.5 erfc(-x / sqrt 2).  For z = min(|x|, Zc):  Q(z) = Phi(-z) = 2 ** (-1 + z P(z))  with P a polynomial of degree
`deg`, fitted (Lawson-reweighted least squares = weighted minimax) for the absolute error of x Phi(x); Phi = Q for x < 0,
1 - Q otherwise.  Prints the fit error, the errors of GELU / Phi / GELU' evaluated in fp32 arithmetic against scipy's
erfc in fp64, for a few (deg, Zc), then the coefficients used (deg 6, Zc 6).

    python tools/fit_gelu.py
"""
import numpy as np
from scipy.special import erfc, erf
np.set_printoptions(precision=17)
def Q(z): return 0.5*erfc(z/np.sqrt(2))
def fit(deg, Zc, iters=60):
    z = np.linspace(1e-6, Zc, 20001)
    y = (np.log2(Q(z)) + 1.0)/z          # P(z), S = -1 + z P(z)
    w0 = np.maximum(1, z)*Q(z)*np.log(2)*z   # d(gelu) = w0 * dP
    # scale variable to [-1,1] for conditioning
    t = 2*z/Zc - 1
    V = np.polynomial.chebyshev.chebvander(t, deg)
    lw = np.ones_like(z)
    for it in range(iters):
        W = w0*lw
        c, *_ = np.linalg.lstsq(V*W[:,None], y*W, rcond=None)
        err = np.abs((V@c - y)*w0)
        lw = lw*(err/err.max())**0.5 + 1e-12
        lw /= lw.max()
    # convert to monomial in z
    pc = np.polynomial.chebyshev.cheb2poly(c)   # in t
    # t = 2z/Zc - 1 : compose
    P = np.polynomial.Polynomial(pc)(np.polynomial.Polynomial([-1, 2/Zc]))
    return P.coef, err.max()
def eval32(coef, x, Zc):
    x = x.astype(np.float32)
    z = np.minimum(np.abs(x), np.float32(Zc)).astype(np.float32)
    c = coef.astype(np.float32)
    p = np.full_like(z, c[-1])
    for a in c[-2::-1]:
        p = (p*z + a).astype(np.float32)
    s = (p*z - np.float32(1)).astype(np.float32)
    q = np.exp2(s).astype(np.float32)
    phi = np.where(x < 0, q, np.float32(1) - q).astype(np.float32)
    return phi
for deg in (5,6,7,8):
    for Zc in (5.0, 5.5, 6.0, 7.0):
        coef, e = fit(deg, Zc)
        x = np.linspace(-10, 10, 400001)
        phi = eval32(coef, x, Zc).astype(np.float64)
        ref = 0.5*erfc(-x/np.sqrt(2))
        eg = np.abs(x*phi - x*ref).max()
        ep = np.abs(phi - ref).max()
        print(deg, Zc, "fit err %.2e  fp32: gelu abs err %.2e  phi abs err %.2e" % (e, eg, ep))
print("----")
coef, e = fit(6, 6.0)
for c in coef: print("%.9ef" % np.float32(c), float(c).hex())
# verify gradient accuracy
x = np.linspace(-10, 10, 400001)
phi = eval32(coef, x, 6.0).astype(np.float64)
x32 = x.astype(np.float32)
pdf = (np.exp2((x32*x32*np.float32(-0.72134752)).astype(np.float32)).astype(np.float32)*np.float32(0.39894228)).astype(np.float32)
g = phi + x*pdf.astype(np.float64)
gref = 0.5*erfc(-x/np.sqrt(2)) + x*np.exp(-x*x/2)/np.sqrt(2*np.pi)
print("grad abs err %.2e" % np.abs(g-gref).max())
