"""Fit the odd polynomials behind gelu_poly2 / dgelu_poly2 (nextgen-uia_amd/csrc/uia_common.h).

Phi(x) - 1/2 and gelu'(x) - 1/2 are odd; each is fitted on [-R, R] as y*P(y^2), y = x/R, by Lawson-reweighted least
squares in the odd Chebyshev basis (well conditioned), converted to monomials, rescaled to x units and checked with an
fp32 Horner evaluation over [-8, 8] (the kernel clamps the argument to +-R).
Run: python tools/fit_gelu_poly.py
"""
import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy.special import erf

R, NTERMS = 4.0, 8
Phi = lambda x: 0.5 * (1 + erf(x / np.sqrt(2)))
phi = lambda x: np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def fit_odd(f):
    y = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000)
    y = y[y > 0]
    w = np.ones_like(y)
    basis = np.stack([Ch.chebval(y, [0] * (2 * k + 1) + [1]) for k in range(NTERMS)], 1)
    for _ in range(200):
        sw = np.sqrt(w)
        c, *_ = np.linalg.lstsq(basis * sw[:, None], f(y * R) * sw, rcond=None)
        e = np.abs(basis @ c - f(y * R))
        w = w * (e / e.max() + 1e-2)
        w /= w.sum()
    full = np.zeros(2 * NTERMS)
    full[1::2] = c
    mono = Ch.cheb2poly(full)[1::2]
    return np.array([v / R ** (2 * k + 1) for k, v in enumerate(mono)])


def horner32(c, x):
    x = x.astype(np.float32)
    xc = np.clip(x, -R, R).astype(np.float32)
    u = xc * xc
    a = np.full_like(x, np.float32(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        a = a * u + np.float32(c[k])
    return (xc * a + np.float32(0.5)).astype(np.float64)


if __name__ == "__main__":
    xx = np.linspace(-8, 8, 400001)
    for name, f in (("Phi", Phi), ("dgelu", lambda x: Phi(x) + x * phi(x))):
        c = fit_odd(lambda x: f(x) - 0.5)
        err = np.abs(horner32(c, xx) - f(xx)).max()
        print(f"{name}: max abs err (fp32 Horner, clamp at {R}) = {err:.2e}")
        print("  {" + ", ".join("%.9ef" % v for v in c) + "}")
