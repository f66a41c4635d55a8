"""Loader for tests/golden/reference_kat.json (the reference's own known-answer
vectors, transcribed as data).  Symbolic entries name Rust f32 constants."""
import json
import os

import numpy as np

f32 = np.float32
EPS = f32(1.1920929e-7)  # approx's default epsilon for f32 == f32::EPSILON

# std::f32::consts
CONSTS = {
    "FRAC_1_SQRT_2": f32(0.707106781186547524400844362104849039),
    "SQRT_2": f32(1.41421356237309504880168872420969808),
    "PI": f32(3.14159265358979323846264338327950288),
    "FRAC_PI_2": f32(1.57079632679489661923132169163975144),
    "FRAC_PI_4": f32(0.785398163397448309615660845819875721),
}
CONSTS["INF"] = f32(np.inf)
CONSTS["sqrt14"] = np.sqrt(f32(14.0))
CONSTS["0.1+0.9*FRAC_1_SQRT_2"] = f32(0.1) + f32(0.9) * CONSTS["FRAC_1_SQRT_2"]


def val(x):
    """Resolve a scalar that may be a symbolic f32 constant ('-NAME' negates)."""
    if isinstance(x, str):
        if x.startswith("-"):
            return -CONSTS[x[1:]]
        return CONSTS[x]
    return f32(x)


def vec(xs):
    return np.array([val(x) for x in xs], dtype=f32)


def point(xs):
    xs = list(xs)
    return vec(xs if len(xs) == 4 else xs + [1.0])


def vector(xs):
    xs = list(xs)
    return vec(xs if len(xs) == 4 else xs + [0.0])


def mat(rows):
    return np.array([[val(x) for x in r] for r in rows], dtype=f32)


def load():
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "reference_kat.json")) as f:
        return json.load(f)


def assert_eps(actual, expected, eps=EPS):
    """assert_abs_diff_eq! with approx's default epsilon (absolute, <=)."""
    a = np.asarray(actual, dtype=f32).reshape(-1)
    e = np.asarray(expected, dtype=f32).reshape(-1)
    assert a.shape == e.shape, (a, e)
    d = np.abs(a.astype(np.float64) - e.astype(np.float64))
    assert np.all(d <= np.float64(eps)), (a, e, d)


def assert_exact(actual, expected):
    """assert_eq! on f32 values (0.0 == -0.0, as in Rust)."""
    a = np.asarray(actual, dtype=f32).reshape(-1)
    e = np.asarray(expected, dtype=f32).reshape(-1)
    assert a.shape == e.shape, (a, e)
    assert np.all(a == e), (a, e)
