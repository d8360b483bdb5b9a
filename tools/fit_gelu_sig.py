#!/usr/bin/env python3
"""Fit the bf16-grade GELU of round 4 (csrc/mlpk_common.h MLPK_GELUS_*, q4gen.py / t4gen.py):

    gelu(x) = x * Phi(x),   Phi(x) ~= 1 / (1 + 2^(x * (k0 + k1 |x| + k2 x^2)))

SEVEN single-issue instructions per element -- two fma (|x| is a free source modifier), a multiply, v_exp_f32, an add, v_rcp_f32,
a multiply -- against eleven for the clamped polynomial it replaces (0.5 + t R(t^2), 8 coefficients), in kernels whose epilogues are
bound by the number of instructions ONE wave can issue behind its MFMAs (docs/DESIGN_LOG_r1-r4.md 3.1f).  The exponent polynomial is in |x|, not in
x^2: no squaring step, and its leading coefficient keeps the sign of k0, so the form has the right limits (Phi -> 0 / 1, gelu -> -0 / x)
for every finite x -- no clamp.  Fit: minimax of the error of GELU itself, |x| * |Phi~ - Phi|, over x in [-14, 14], with the formula
evaluated in emulated fp32 exactly as the kernels evaluate it.  Prints k (already multiplied by -log2 e) and the error table."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x64 = np.linspace(-14, 14, 56001)
PHI = 0.5 * (1 + erf(x64 / np.sqrt(2)))
x32 = x64.astype(np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)


def phi32(k, x=x32):
    """the kernels' operation sequence in fp32 (v_exp_f32 / v_rcp_f32 are 1-ulp approximations: modelled as correctly rounded)"""
    k = [np.float32(v) for v in k]
    a = np.abs(x)
    q = fma32(a, np.full_like(x, k[2]), k[1])
    q = fma32(a, q, k[0])
    z = (x * q).astype(np.float32)
    with np.errstate(over="ignore"):
        e = np.exp2(z.astype(np.float64)).astype(np.float32)
        r = (np.float32(1.0) / (np.float32(1.0) + e)).astype(np.float32)
    return r


def errs(k):
    d = np.abs(phi32(k).astype(np.float64) - PHI)
    return d.max(), (np.abs(x64) * d).max()


if __name__ == "__main__":
    L2E = np.log2(np.e)
    best = None
    for seed in range(6):
        k = -L2E * np.array([1.5851, 0.0212, 0.0628]) * (1 + 0.03 * np.random.default_rng(seed).standard_normal(3))
        for _ in range(8):
            k = minimize(lambda v: errs(v)[1], k, method="Nelder-Mead", options=dict(xatol=1e-12, fatol=1e-14, maxiter=20000, maxfev=20000)).x
        if best is None or errs(k)[1] < errs(best)[1]:
            best = k
    k = [float(np.float32(v)) for v in best]
    print("k0, k1, k2 (x -log2 e) = %.9g, %.9g, %.9g" % tuple(k))
    print("max |Phi error| = %.3g   max |gelu error| = %.3g" % errs(k))
    for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 14)):
        m = (np.abs(x64) >= lo) & (np.abs(x64) <= hi)
        d = np.abs(phi32(k).astype(np.float64) - PHI)
        print("  %4.1f <= |x| <= %4.1f: |Phi error| <= %.3g   |gelu error| <= %.3g" % (lo, hi, d[m].max(), (np.abs(x64) * d)[m].max()))
    big = np.array([-1e4, -1e3, -50.0, -20.0, 20.0, 50.0, 1e3, 1e4, 1e19, -1e19], np.float32)
    print("tails:", [(float(v), float(v * phi32(k, np.array([v], np.float32))[0])) for v in big])
