#!/usr/bin/env python3
"""Fit of the packed-f16 GELU of the generated token kernel's bf16 grade (q4gen.GELU_H2; round 5).

    h = f16(x) ; t = h * s ; u = t * t - 1 ; Phi = clamp01(0.5 + t * Q(u)) ; gelu = h * Phi          every operation ONE f16 rounding

s = f16(sqrt(2) / A), Q = n coefficients by Horner.  The coefficients are fitted for the error of x * Phi on [0, A] (Lawson-weighted least
squares toward minimax) and then rounded to f16 ONE AT A TIME, constant term first, refitting the remaining ones after each rounding --
rounding all of them at once costs 3e-4 of slope error in the constant term alone.  n must be ODD: the leading coefficient is then
positive, the polynomial runs off to +-infinity beyond the interval in the direction the clamp wants, and no operand clamp is needed.
What limits the result is f16 arithmetic, not the degree: n = 7 on A = 4 leaves 1.2e-4 .. 3.5e-4 rms (x ~ N(0, 0.5 .. 2)) and 2.4e-3 max
before the output rounding; n = 5 is 4x worse in rms, n = 9 no better.      usage: python tools/fit_gelu_h2.py [n] [A]
"""
import math
import sys

import numpy as np

R2 = math.sqrt(2.0)


def Phi(x):
    return 0.5 * (1.0 + np.vectorize(math.erf)(np.asarray(x, np.float64) / R2))


def f16(v):
    with np.errstate(over="ignore", invalid="ignore"):
        return np.asarray(v, dtype=np.float64).astype(np.float16)


def eval_h2(x32, coefs, A):
    """(gelu as the f16 product, Phi) for fp32 inputs, operation by operation"""
    with np.errstate(over="ignore", invalid="ignore"):
        h = np.asarray(x32, np.float32).astype(np.float16).astype(np.float64)
        s = float(np.float16(R2 / A))
        t = f16(h * s).astype(np.float64)
        u = f16(t * t - 1.0).astype(np.float64)
        c = [float(np.float16(v)) for v in coefs]
        q = f16(c[0] * u + c[1]).astype(np.float64)
        for k in range(2, len(c)):
            q = f16(q * u + c[k]).astype(np.float64)
        p = np.clip(f16(t * q + 0.5).astype(np.float64), 0.0, 1.0)
        return f16(h * p).astype(np.float64), p


def lawson(V, tgt, w, iters=50):
    ww = w.copy()
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V * ww[:, None], tgt * ww, rcond=None)
        e = np.abs((V @ c - tgt) * w)
        ww = ww * (1 + e / e.max()) ** 0.5
    return c


def fit(n, A):
    s16 = float(np.float16(R2 / A))
    x = np.linspace(1e-3, A, 20001)
    t = x * s16
    u = t * t - 1
    target, w = (Phi(x) - 0.5) / t, x * t          # error of gelu = x * t * dQ
    V = np.vander(u, n)
    c = lawson(V, target, w)
    fixed = {}
    for idx in range(n - 1, -1, -1):
        fixed[idx] = float(np.float16(c[idx]))
        free = [i for i in range(n) if i not in fixed]
        if not free:
            break
        cf = lawson(V[:, free], target - sum(V[:, i] * v for i, v in fixed.items()), w, 30)
        for i, v in zip(free, cf):
            c[i] = v
    return [fixed[i] for i in range(n)]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    A = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    c = fit(n, A)
    print("scale %r  coefs %r" % (float(np.float16(R2 / A)), c))
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)
    xa = allh[np.isfinite(allh)].astype(np.float32)
    g, p = eval_h2(xa, c, A)
    ref = xa.astype(np.float64) * Phi(xa)
    err = np.abs(g - ref)
    print("every finite f16 input: max |err| %.3e at %g; max |err| / |x| on |x| >= 0.25: %.3e; max |err| on |x| < 0.25: %.3e; Phi(x > A) >= %.5f, Phi(x < -A) <= %.5f"
          % (err.max(), xa[err.argmax()], (err / np.maximum(np.abs(xa), 1e-30))[np.abs(xa) >= 0.25].max(), err[np.abs(xa) < 0.25].max(),
             p[xa > A].min(), p[xa < -A].max()))
    rng = np.random.default_rng(0)
    for sd in (0.5, 1.0, 2.0):
        x = (rng.standard_normal(1000000) * sd).astype(np.float32)
        e = eval_h2(x, c, A)[0] - x.astype(np.float64) * Phi(x)
        print("x ~ N(0, %.1f): rms error %.3e, max %.3e" % (sd, math.sqrt((e ** 2).mean()), np.abs(e).max()))
