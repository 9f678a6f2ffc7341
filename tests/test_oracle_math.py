"""HJ-NUM-1 transcendental polynomials: accuracy against numpy in float64."""
import ctypes as C

import numpy as np


def ulp_err(got, want):
    want32 = want.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want) / np.maximum(ulp, 1e-45)


def test_exp_accuracy_and_specials(oracle):
    L = oracle.lib()
    x = np.concatenate([np.linspace(-86.9, 87.9, 20001), np.linspace(-10, 0.0, 5001)]).astype(np.float32)
    got = np.array([L.hjo_exp(float(v)) for v in x], np.float32)
    assert ulp_err(got, np.exp(x.astype(np.float64))).max() <= 2.0
    assert L.hjo_exp(0.0) == 1.0 and L.hjo_exp(-0.0) == 1.0
    assert L.hjo_exp(-87.0) == 0.0 and L.hjo_exp(-1e30) == 0.0 and L.hjo_exp(float("-inf")) == 0.0
    assert L.hjo_exp(89.0) == float("inf") and np.isnan(L.hjo_exp(float("nan")))
    # the constant of the reconstruction filter: exp(-8) (c0 = 3.3546e-4)
    assert abs(L.hjo_exp(-8.0) - 3.3546262790e-4) < 1e-10


def test_sincos2pi_accuracy(oracle):
    L = oracle.lib()
    out = (C.c_float * 2)()
    v = np.concatenate([np.linspace(0, 1, 10001), np.random.default_rng(1).random(5000)]).astype(np.float32)
    err = 0.0
    for x in v:
        L.hjo_sincos2pi(float(x), out)
        err = max(err, abs(out[0] - np.sin(2 * np.pi * float(x))), abs(out[1] - np.cos(2 * np.pi * float(x))))
        assert abs(out[0] ** 2 + out[1] ** 2 - 1) < 5e-7
    assert err < 2.5e-7    # far inside the Vulkan GLSL tolerance of 2^-11 the reference runs under
    for x, s, c in ((0.0, 0.0, 1.0), (0.25, 1.0, 0.0), (0.5, 0.0, -1.0), (0.75, -1.0, 0.0), (1.0, 0.0, 1.0)):
        L.hjo_sincos2pi(x, out)
        assert out[0] == s and out[1] == c


def test_atan2_asin_accuracy(oracle):
    L = oracle.lib()
    r = np.random.default_rng(2)
    y, x = r.normal(size=4000).astype(np.float32), r.normal(size=4000).astype(np.float32)
    got = np.array([L.hjo_atan2(float(a), float(b)) for a, b in zip(y, x)])
    assert np.abs(got - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 5e-7
    s = np.linspace(-1, 1, 4001).astype(np.float32)
    got = np.array([L.hjo_asin(float(a)) for a in s])
    assert np.abs(got - np.arcsin(s.astype(np.float64))).max() < 5e-7
    assert np.isnan(L.hjo_atan2(0.0, 0.0))          # -> the reference's isnan guard sets uv.x = 0 (sphere.glsl:49-51)
    assert abs(L.hjo_atan2(1.0, 0.0) - np.pi / 2) < 1e-7 and abs(L.hjo_atan2(0.0, -1.0) - np.pi) < 1e-6
