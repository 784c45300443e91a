"""csrc/csdo_math.h: the ONE sin / cos / tan / atan2 of every build of the device program.  CPU: accuracy of the host build against
mpmath and agreement with the C library (what the oracle and the reference call); GPU: the device build returns the host
build's bits - the property the bit-level chain tests (test_gpu_sets.py) rest on."""
import numpy as np
import pytest

from tests import emu_lib

FNS = {0: "sin", 1: "cos", 2: "tan"}


def _args(rng, n):
    return {"quarter": rng.uniform(-0.8, 0.8, n), "path yaw": rng.uniform(-20.0, 20.0, n), "large": rng.uniform(-1e5, 1e5, n),
            "steer": rng.uniform(-0.7, 0.7, n), "tiny": rng.uniform(-1e-3, 1e-3, n),
            "near multiples of pi/2": np.round(rng.uniform(-200, 200, n)) * (np.pi / 2) * (1 + rng.normal(size=n) * 1e-12)}


def _ulp_error(got, exact_fn, xs, ys=None):
    import mpmath as mp
    mp.mp.prec = 160
    worst = 0.0
    for i in range(len(xs)):
        t = exact_fn(mp.mpf(float(xs[i]))) if ys is None else exact_fn(mp.mpf(float(xs[i])), mp.mpf(float(ys[i])))
        u = np.spacing(abs(float(t)))
        worst = max(worst, float(abs((mp.mpf(float(got[i])) - t) / mp.mpf(float(u)))))
    return worst


def test_sin_cos_tan_are_within_0p6_ulp_and_agree_with_the_c_library():
    import mpmath as mp
    rng = np.random.default_rng(5)
    for name, xs in _args(rng, 1500).items():
        for fn, f in ((0, mp.sin), (1, mp.cos), (2, mp.tan)):
            got, libm = emu_lib.math_eval(fn, xs), emu_lib.math_eval(fn + 10, xs)
            err = _ulp_error(got, f, xs)
            assert err < 0.6, (name, FNS[fn], err)
            assert np.mean(got == libm) >= 0.985, (name, FNS[fn], float(np.mean(got == libm)))


def test_atan2_is_within_0p55_ulp_in_every_octant_and_at_the_breakpoints():
    import mpmath as mp
    rng = np.random.default_rng(6)
    y, x = rng.normal(size=3000) * 10, rng.normal(size=3000) * 10
    got = emu_lib.math_eval(3, y, x)
    assert _ulp_error(got, mp.atan2, y, x) < 0.55
    assert np.mean(got == emu_lib.math_eval(13, y, x)) >= 0.99
    # ratios at the edges of the five intervals of the reduction, and |y| ~ |x|
    y = rng.normal(size=3000) * 10
    x = y * rng.choice([0.125, 0.375, 0.625, 0.875, 1, -1, 8, -8, 8 / 3.0, 1.6], 3000) * (1 + rng.normal(size=3000) * 1e-9)
    assert _ulp_error(emu_lib.math_eval(3, y, x), mp.atan2, y, x) < 0.55
    # axes and signed zeros: the C library's values
    sy = np.array([0.0, 0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 0.0, 5.0, -5.0, 1e-300, 3.0, -3.0])
    sx = np.array([0.0, -0.0, 0.0, -1.0, -1.0, 0.0, 0.0, 1.0, -0.0, -0.0, 1e300, 3.0, -3.0])
    a, b = emu_lib.math_eval(3, sy, sx), emu_lib.math_eval(13, sy, sx)
    assert np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b)), (a, b)
    assert np.isnan(emu_lib.math_eval(3, [np.nan, 1.0], [1.0, np.nan])).all()


def test_special_arguments():
    assert np.array_equal(emu_lib.math_eval(0, [0.0, np.pi]), emu_lib.math_eval(10, [0.0, np.pi]))
    assert np.array_equal(emu_lib.math_eval(1, [0.0, -0.0]), [1.0, 1.0])
    for fn in (0, 1, 2):
        assert np.isnan(emu_lib.math_eval(fn, [np.inf, -np.inf, np.nan])).all()
    # beyond 2^20 pi/2 the reduction loses accuracy gracefully (documented in the header), it does not blow up
    big = np.array([2.0e6, -7.5e6, 1.0e9])
    assert np.all(np.abs(emu_lib.math_eval(0, big) - np.sin(big)) < 1e-6)


@pytest.mark.gpu
def test_device_build_returns_the_host_builds_bits(gpu_handle):
    rng = np.random.default_rng(7)
    for name, xs in _args(rng, 200000).items():
        for fn in (0, 1, 2):
            dev, host = gpu_handle.math_eval(fn, xs), emu_lib.math_eval(fn, xs)
            assert np.array_equal(dev.view(np.uint64), host.view(np.uint64)), (name, FNS[fn], int(np.sum(dev != host)))
    y, x = rng.normal(size=400000) * 10, rng.normal(size=400000) * 10
    x[::7] = y[::7] * rng.choice([0.125, 0.375, 0.625, 0.875, 1, -1, 8, -8], x[::7].size)
    y[::1001] = 0.0
    x[::1003] = -0.0
    assert np.array_equal(gpu_handle.math_eval(3, y, x).view(np.uint64), emu_lib.math_eval(3, y, x).view(np.uint64))
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e300, 2.0e6, np.pi / 4, -np.pi / 4, 0.7853981633974484])
    for fn in (0, 1, 2):
        assert np.array_equal(gpu_handle.math_eval(fn, sp).view(np.uint64), emu_lib.math_eval(fn, sp).view(np.uint64)), fn
