#!/usr/bin/env python3
"""Fit the division-free GELU used by the fused token-mixing kernel (16-bit outputs):
     gelu(x) = x * Phi(x),   Phi(x) ~= 0.5 + t * Q(t*t - 1),   t = clamp(x * sqrt(2) / A, -sqrt 2, sqrt 2)
Q = polynomial of degree K-1 in u = t^2 - 1 in [-1, 1] (well conditioned in fp32), weighted minimax fit of the error of Phi.
Prints the coefficients (highest first, Horner order) and the error of the fp32 evaluation over a dense grid."""
import sys
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf

A = float(sys.argv[1]) if len(sys.argv) > 1 else 4.5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 11
R2 = np.sqrt(2.0)


def target(t):            # (Phi(x) - 0.5) / t with x = t * A / sqrt 2
    x = t * A / R2
    return np.where(t > 0, 0.5 * erf(x / R2) / np.maximum(t, 1e-300), A / R2 / np.sqrt(2 * np.pi))


n = 6000
t = (np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 + 0.5) * R2
u = t * t - 1
f = target(t)
V = C.chebvander(u, K - 1)
w = t.copy()
for it in range(200):
    c, *_ = np.linalg.lstsq(V * w[:, None], f * w, rcond=None)
    e = np.abs((V @ c - f) * t)
    w = w * (1 + 2 * e / e.max())
    w /= w.max()
mono = C.cheb2poly(c)                     # ascending powers of u
coef = mono[::-1].astype(np.float32)      # Horner order, fp32


def fma32(a, b, c_):
    return (a.astype(np.float64) * b.astype(np.float64) + c_.astype(np.float64)).astype(np.float32)


def gelu_poly32(x):
    x = x.astype(np.float32)
    tt = np.clip((x * np.float32(R2 / A)).astype(np.float32), np.float32(-R2), np.float32(R2))
    uu = fma32(tt, tt, np.float32(-1.0) * np.ones_like(tt))
    q = np.full_like(tt, coef[0])
    for cc in coef[1:]:
        q = fma32(q, uu, np.full_like(tt, cc))
    ph = fma32(tt, q, np.full_like(tt, np.float32(0.5)))
    return (x * ph).astype(np.float32), ph


xs = np.concatenate([np.linspace(-12, 12, 2000001), np.linspace(-0.01, 0.01, 20001)])
g, ph = gelu_poly32(xs)
ref_phi = 0.5 * (1 + erf(xs / R2))
ref = xs * ref_phi
print("A = %.3f  K = %d   sum|coef| = %.2f" % (A, K, np.abs(mono).sum()))
print("max |Phi err| = %.3e   max |gelu err| = %.3e (at x = %.3f)   max |gelu err| on |x| <= A: %.3e"
      % (np.abs(ph - ref_phi).max(), np.abs(g - ref).max(), xs[np.abs(g - ref).argmax()], np.abs(g - ref)[np.abs(xs) <= A].max()))
print("scale sqrt2/A = %.9g" % (R2 / A))
print("coefficients (Horner order, u^%d first):" % (K - 1))
print(", ".join("%.9gf" % v for v in coef))

if len(sys.argv) > 3 and sys.argv[3] == "folded":
    # the same fit for a kernel whose first product already delivers s = x / A (A a power of two, folded into the weights):
    # Phi(x) ~= 0.5 + s * P(s * s),  s = clamp(x / A, -1, 1):  t = sqrt2 s, u = 2 s^2 - 1  ->  P(w) = sqrt2 * Q(2 w - 1)
    comp = np.zeros(K)
    base = np.ones(1)
    for k in range(K):
        comp[:len(base)] += mono[k] * base
        base = P.polymul(base, [-1.0, 2.0])
    pw = (comp * R2)[::-1].astype(np.float32)          # Horner order in w = s^2

    def gelu_folded32(x):
        x = x.astype(np.float32)
        s = (x * np.float32(1.0 / A)).astype(np.float32)
        sc = np.clip(s, np.float32(-1), np.float32(1))
        w_ = (sc * sc).astype(np.float32)
        q = np.full_like(sc, pw[0])
        for cc in pw[1:]:
            q = fma32(q, w_, np.full_like(sc, cc))
        ph = fma32(sc, q, np.full_like(sc, np.float32(0.5)))
        return (x * ph).astype(np.float32), ph
    g2, ph2 = gelu_folded32(xs)
    print("folded form: max |Phi err| = %.3e   max |gelu err| on |x| <= A: %.3e   max rel beyond: %.3e   sum|coef| = %.1f"
          % (np.abs(ph2 - ref_phi).max(), np.abs(g2 - ref)[np.abs(xs) <= A].max(),
             (np.abs(g2 - ref)[np.abs(xs) > A] / np.abs(xs[np.abs(xs) > A])).max(), np.abs(pw).sum()))
    print("coefficients (Horner order, w^%d first):" % (K - 1))
    print(", ".join("%.9g" % v for v in pw))

if len(sys.argv) > 3 and sys.argv[3] == "raw":
    # the same fit in the variable the kernels have at hand: Phi(x) ~= 0.5 + t * R(t * t), t = clamp(x, -A, A) -- no scaling multiply
    # in front of the clamp: t = (A / sqrt2) t', u' = 2 (t / A)^2 - 1  ->  R(v) = (sqrt2 / A) Q(2 v / A^2 - 1)
    comp = np.zeros(K)
    base = np.ones(1)
    for k in range(K):
        comp[:len(base)] += mono[k] * base
        base = P.polymul(base, [-1.0, 2.0 / (A * A)])
    rv = (comp * R2 / A)[::-1].astype(np.float32)          # Horner order in v = t^2

    def gelu_raw32(x):
        x = x.astype(np.float32)
        tt = np.clip(x, np.float32(-A), np.float32(A))
        v_ = (tt * tt).astype(np.float32)
        q = np.full_like(tt, rv[0])
        for cc in rv[1:]:
            q = fma32(q, v_, np.full_like(tt, cc))
        ph = fma32(tt, q, np.full_like(tt, np.float32(0.5)))
        return (x * ph).astype(np.float32), ph
    g3, ph3 = gelu_raw32(xs)
    print("raw form: max |Phi err| = %.3e   max |gelu err| on |x| <= A: %.3e   max rel beyond: %.3e"
          % (np.abs(ph3 - ref_phi).max(), np.abs(g3 - ref)[np.abs(xs) <= A].max(),
             (np.abs(g3 - ref)[np.abs(xs) > A] / np.abs(xs[np.abs(xs) > A])).max()))
    print("coefficients (Horner order, v^%d first):" % (K - 1))
    print(", ".join("%.9g" % v for v in rv))
