"""The time surface's `std::exp` (event_detector.cc:243-259), by sweep instead of by sampling.

The device renders `exp(-(t_sync - m) / decay_sec)` with the OCML fp64 `exp`, the oracle (like the reference) with
glibc's.  A difference of one ulp between the two can only change a byte where `v * 127.5 + 127.5` lies within an ulp
of a rounding boundary — rounds 1-5 argued that from > 10^7 random pixels.  Here every age `dt = k * 2^-22 s`,
k = 0 .. 2^21 - 1 (0 .. 0.5 s in steps of 0.24 us: every value an event stamped with microseconds can have against a
sync time of the same grid, and four times finer), is rendered through k_time_surface4 in both polarities, with and
without `ignore_polarity`, at the shipped decay (20 ms) and at two others, from small and from epoch-sized stamps — and
compared byte for byte with the oracle.  Should a byte ever differ, the device needs a correctly rounded exp.
"""
import numpy as np
import pytest

from esvio_amd import frontend as FE

pytestmark = pytest.mark.gpu

W, H = 2048, 1024  # 2^21 pixels: pixel i of the plane holds the age i * 2^-22 s


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.mark.parametrize("decay_ms,ignore_polarity,t_sync", [
    (20.0, 0, 1.0), (20.0, 0, 1024.0), (20.0, 0, 1.7e9), (20.0, 1, 1.0), (20.0, 1, 1024.0), (20.0, 1, 1.7e9),
    (30.0, 0, 1.0), (7.3, 0, 1.0)])
def test_every_age_on_the_2_pow_minus_22_grid(oracle, decay_ms, ignore_polarity, t_sync):
    k = np.arange(W * H, dtype=np.float64).reshape(H, W)
    age = k * 2.0 ** -22
    m = t_sync - age  # exact: every operand is a multiple of 2^-22 below 2^31
    assert np.array_equal(t_sync - m, age)
    zero = np.zeros((H, W))
    ft = FE.FeatureTracker(FE.make_config(W, H, decay_ms=decay_ms, ignore_polarity=ignore_polarity, max_cnt=10))
    det = oracle.Detector(W, H, decay_ms=decay_ms, ignore_polarity=ignore_polarity)
    # camera 0: the positive polarity is the newer one (S1 > S0: +exp), camera 1: the negative one (-exp)
    for cam, (s0, s1) in enumerate(((zero, m), (m, zero))):
        ft.detector.set_sae(cam, zero, zero, s0, s1)
        det.set_sae(cam, zero, zero, s0, s1)
    n_diff = 0
    for cam in (0, 1):
        g = ft.detector._ts(cam, t_sync)
        o = det.time_surface(cam, t_sync)
        bad = np.flatnonzero(g != o)
        n_diff += len(bad)
        assert len(bad) == 0, ("cam %d: %d of %d bytes differ, first at k = %d (age %.9f s): device %d, oracle %d"
                               % (cam, len(bad), W * H, bad[0], bad[0] * 2.0 ** -22, g.flat[bad[0]], o.flat[bad[0]]))
        # the sweep covers what it claims to: the whole byte range on the decaying side, saturation at age 0
        vals = np.unique(g)
        if ignore_polarity:  # 255 * exp(.): every byte value
            assert g.flat[0] == 255 and vals.min() == 0 and len(vals) == 256
        elif cam == 0:       # 127.5 * (1 + exp(.))
            assert g.flat[0] == 255 and vals.min() == 128 and len(vals) == 128
        else:                # 127.5 * (1 - exp(.))
            assert g.flat[0] == 0 and vals.max() <= 128 and len(vals) >= 128
    ft.close()
    assert n_diff == 0
