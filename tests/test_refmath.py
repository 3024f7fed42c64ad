"""The oracle's restated exp / sin / cos / pow(x,2) (oracle/fsq_refmath.c) must equal the libm the
reference ran on (glibc 2.35, same image here and on the GPU box) bit for bit."""
import ctypes

import numpy as np
import pytest

import oracle as O


@pytest.fixture(scope="module", autouse=True)
def _build():
    O.build()


def _fn(L, name, nargs=1):
    f = getattr(L, name)
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_double] * nargs
    return f


def test_refmath_equals_libm():
    L = O.lib()
    libm = ctypes.CDLL("libm.so.6")
    rng = np.random.default_rng(7)
    n = 200000
    pairs = [("fsq_ref_exp", "exp", np.concatenate([rng.uniform(-80, 2, n), rng.uniform(-750, 710, 2000)])),
             ("fsq_ref_sin", "sin", np.concatenate([rng.uniform(0, 6.3, n), rng.uniform(-1e6, 1e6, 20000)])),
             ("fsq_ref_cos", "cos", np.concatenate([rng.uniform(0, 6.3, n), rng.uniform(-1e6, 1e6, 20000)]))]
    for mine, ref, xs in pairs:
        f, g = _fn(L, mine), _fn(libm, ref)
        bad = sum(1 for x in xs if np.float64(f(x)).view(np.uint64) != np.float64(g(x)).view(np.uint64))
        assert bad == 0, (mine, bad)
    f, g = _fn(L, "fsq_ref_pow2"), _fn(libm, "pow", 2)
    xs = np.concatenate([rng.uniform(-2, 2, n), 10.0 ** rng.uniform(-150, 150, 20000),
                         1 + rng.uniform(-1e-3, 1e-3, 20000), [0.0, -0.0, 1.0, 5e-324, 1e-310, 1e200, -1e200]])
    bad = sum(1 for x in xs if np.float64(f(x)).view(np.uint64) != np.float64(g(x, 2.0)).view(np.uint64))
    assert bad == 0
    # and pow(x, 2) is NOT x*x: the reason the restatement exists
    assert sum(1 for x in xs[:n] if f(x) != x * x) > 0
