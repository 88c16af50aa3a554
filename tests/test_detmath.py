"""ccmp_detmath.h (through the det oracle build) against glibc: accuracy in ulps."""
import ctypes as C

import numpy as np


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(b), 1e-300))


def test_sincos_accuracy(oracle_det):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-7, 7, 20000), rng.uniform(-1e5, 1e5, 5000), rng.uniform(-1e-6, 1e-6, 1000),
                         np.array([0.0, np.pi / 2, np.pi, -np.pi / 4, 3 * np.pi / 4, 1e-300])])
    s, c = C.c_double(), C.c_double()
    S, Cc = np.empty_like(xs), np.empty_like(xs)
    for i, x in enumerate(xs):
        oracle_det.lib.orc_sincos(x, C.byref(s), C.byref(c))
        S[i], Cc[i] = s.value, c.value
    assert _ulps(S, np.sin(xs)).max() <= 1.0 and _ulps(Cc, np.cos(xs)).max() <= 1.0
    assert np.abs(S * S + Cc * Cc - 1).max() < 5e-16


def test_sincos_out_of_range_is_nan(oracle_det):
    s, c = C.c_double(), C.c_double()
    for x in (2e6, -2e6, np.inf, np.nan):
        oracle_det.lib.orc_sincos(x, C.byref(s), C.byref(c))
        assert np.isnan(s.value) and np.isnan(c.value)


def test_atan2_accuracy(oracle_det):
    rng = np.random.default_rng(2)
    y = np.abs(np.concatenate([rng.standard_normal(20000), rng.uniform(0, 1e-8, 2000), [0.0, 1.0, 0.0, 1e300]]))
    x = np.abs(np.concatenate([rng.standard_normal(20000), rng.uniform(0, 1, 2000), [0.0, 0.0, 1.0, 1e-300]]))
    got = np.array([oracle_det.lib.orc_atan2_nn(a, b) for a, b in zip(y, x)])
    exp = np.arctan2(y, x)
    assert _ulps(got, exp).max() <= 2.0
    assert got[-4] == 0.0 and got[-3] == np.pi / 2
