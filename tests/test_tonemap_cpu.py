"""The tone map's short form of pow in binary32 (k_tonemap in raytracer_amd/csrc/rt_kernels.h): the window it tests must
contain the value the reference computes (raytrace.c:433-466: pow in double, times 255, rounded to float, truncated),
whatever the hardware's square root does within its one unit in the last place."""
import numpy as np
import pytest

F = np.float32


def fast_window(x, halves, sqrt_ulps):
    """Both ends of the kernel's window for x ** (halves / 2) * 255, with the square root off by sqrt_ulps ulps."""
    with np.errstate(all="ignore"):
        pf = np.sqrt(x).astype(F) if halves & 1 else np.ones_like(x)
        for _ in range(abs(sqrt_ulps)):
            pf = np.nextafter(pf, F(np.inf) if sqrt_ulps > 0 else F(0)) if halves & 1 else pf
        for _ in range(halves >> 1):
            pf = (pf * x).astype(F)
        t = (pf * F(255.0)).astype(F)
        return (t * F(0.9999992)).astype(F), (t * F(1.0000008)).astype(F)


@pytest.mark.parametrize("halves", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("sqrt_ulps", [-1, 0, 1])
def test_the_binary32_window_holds_the_reference_value(halves, sqrt_ulps):
    rng = np.random.default_rng(halves * 7 + sqrt_ulps)
    edges = (np.arange(1, 256, dtype=np.float64) / 255.0) ** (2.0 / halves)
    near = np.concatenate([(edges * (1.0 + d)).astype(F) for d in (-3e-7, -1e-7, 0.0, 1e-7, 3e-7)])
    x = np.concatenate([rng.random(400000, dtype=F), (rng.random(100000, dtype=F) * F(4.0)).astype(F), near,
                        np.array([0.0, 1.0, 4.0, 1e-30, 1e-10], F)])
    lo, hi = fast_window(x, halves, sqrt_ulps)
    want = ((x.astype(np.float64) ** (halves / 2.0)) * 255.0).astype(F)
    # (products that underflow lose their relative accuracy, and their byte is 0 whatever they are)
    big = want > F(1e-20)
    assert np.all(lo[big] <= want[big]) and np.all(want[big] <= hi[big])
    assert np.all(hi[~big] < 1) and np.all(want[~big] < 1)
    # hence: where both ends truncate to the same integer, that integer is the reference's
    same = np.trunc(lo) == np.trunc(hi)
    assert same.mean() > 0.99
    assert np.array_equal(np.trunc(lo)[same], np.trunc(want)[same])
