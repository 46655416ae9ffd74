#!/usr/bin/env python3
"""Coefficients of the two odd degree-17 polynomials behind the bf16 GELU epilogues (devias_amd/csrc/common.h):
erf(x/sqrt2) and g(x) = erf(x/sqrt2)/2 + x*pdf(x) on t = clamp(x, -4, 4)/4.  Least squares at Chebyshev nodes; prints the
coefficients (lowest power first) and the maximum error of the fp32 Horner evaluation over |x| <= 8."""
from math import erf, pi, sqrt

import numpy as np

L = 4.0


def fit(f, deg, n=8001):
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    y = np.array([f(v * L) for v in x])
    A = np.stack([x ** k for k in range(1, deg + 1, 2)], 1)
    return np.linalg.lstsq(A, y, rcond=None)[0]


def horner32(c, x):
    t = (np.clip(x, -L, L).astype(np.float32) * np.float32(1 / L)).astype(np.float32)
    u = (t * t).astype(np.float32)
    acc = np.full_like(u, np.float32(c[-1]))
    for k in c[-2::-1]:
        acc = (acc * u + np.float32(k)).astype(np.float32)
    return (acc * t).astype(np.float32)


if __name__ == "__main__":
    xx = np.linspace(-8, 8, 400001)
    for name, f in (("erf(x/sqrt2)", lambda x: erf(x / sqrt(2))),
                    ("erf(x/sqrt2)/2 + x*pdf(x)", lambda x: 0.5 * erf(x / sqrt(2)) + x * np.exp(-0.5 * x * x) / sqrt(2 * pi))):
        c = fit(f, 17)
        err = np.abs(horner32(c, xx) - np.array([f(v) for v in xx])).max()
        print(f"{name}: max |error| {err:.2e}\n  " + ", ".join("%.9e" % v for v in c))
