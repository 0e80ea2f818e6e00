"""open_ludwig_amd/csrc/jl_math.h - the double log2 / exp2 / log that the HIP kernels AND the CPU oracle compile (the one piece
of arithmetic they share, so that wall-model cells are bit-comparable) - checked against glibc through numpy: the shared
source must be an accurate implementation in its own right, or the parity tests of the wall model would compare a bug with itself."""
import ctypes as C

import numpy as np

from oracle import oracle


def _call(which, x):
    lib = oracle.lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib.oracle_jl_math.restype = None
    lib.oracle_jl_math(C.c_int(which), C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int64(x.size))
    return out


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_log2_log_exp2_within_4_ulp_of_glibc():
    rng = np.random.default_rng(1)
    # every magnitude a Float32 can hold, plus values near 1 (where log loses its leading digits)
    x = np.concatenate([np.exp2(rng.uniform(-149, 128, 400_000)), 1.0 + rng.uniform(-0.3, 0.45, 200_000),
                        rng.uniform(0.5, 2.0, 200_000).astype(np.float32).astype(np.float64)])
    near1 = np.abs(x - 1.0) < 0.3
    for which, ref in ((0, np.log2), (2, np.log)):
        got, want = _call(which, x), ref(x)
        u = _ulps(got, want)
        assert u[~near1].max() <= 4.0, (which, u[~near1].max())
        # near 1 the result is tiny: bound the absolute error by 2 ulp of the mantissa part instead
        assert (np.abs(got - want)[near1] <= 4 * np.spacing(0.5)).all()
    e = rng.uniform(-1000, 1000, 400_000)
    e = np.concatenate([e, rng.uniform(-30, 30, 400_000), np.arange(-1022, 1024, dtype=np.float64), np.arange(-50, 50) + 0.5])
    assert _ulps(_call(1, e), np.exp2(e)).max() <= 2.0


def test_special_values():
    x = np.array([0.0, -1.0, np.inf, np.nan, 1.0, 2.0, 0.5, 8.0])
    l2 = _call(0, x)
    assert l2[0] == -np.inf and np.isnan(l2[1]) and l2[2] == np.inf and np.isnan(l2[3])
    assert list(l2[4:]) == [0.0, 1.0, -1.0, 3.0]
    ln = _call(2, x)
    assert ln[0] == -np.inf and np.isnan(ln[1]) and ln[2] == np.inf and ln[4] == 0.0
    e = _call(1, np.array([0.0, 1.0, -1.0, 10.0, 1024.0, -1100.0, np.nan, 1023.75]))
    assert list(e[:4]) == [1.0, 2.0, 0.5, 1024.0] and e[4] == np.inf and e[5] == 0.0 and np.isnan(e[6])
    assert abs(e[7] / np.exp2(1023.75) - 1.0) < 1e-15


def test_float32_pow_agrees_with_glibc_after_rounding():
    """Base.^(::Float32, ::Float32) = Float32(exp2(log2(Float64(x)) * y)): the wall model's x^(1/7) over its input range.
    Two accurate double kernels may differ in the last double bit; after the rounding to Float32 that shows in ~1e-8 of the calls."""
    lib = oracle.lib()
    rng = np.random.default_rng(2)
    n = 2_000_000
    x = np.exp2(rng.uniform(-40, 10, n)).astype(np.float32)
    y = np.full(n, np.float32(1.0) / np.float32(7.0), dtype=np.float32)
    out = np.empty(n, dtype=np.float32)
    lib.oracle_jl_powf.restype = None
    lib.oracle_jl_powf(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int64(n))
    want = np.exp2(np.log2(x.astype(np.float64)) * y.astype(np.float64)).astype(np.float32)
    differ = np.count_nonzero(out != want)
    assert differ <= 2, differ
    assert np.abs(out.astype(np.float64) / want - 1.0).max() < 1.3e-7
    # the constant the reference folds: (2 * 8.3)^(-1/7)
    c = np.array([np.float32(2.0) * np.float32(8.3)], dtype=np.float32)
    yc = np.array([np.float32(-1.0) / np.float32(7.0)], dtype=np.float32)
    oc = np.empty(1, dtype=np.float32)
    lib.oracle_jl_powf(C.c_void_p(c.ctypes.data), C.c_void_p(yc.ctypes.data), C.c_void_p(oc.ctypes.data), C.c_int64(1))
    assert oc[0] == np.float32(np.exp2(np.log2(np.float64(c[0])) * np.float64(yc[0])))
