"""The synthetic inputs of bench.py and the tests (esvio_amd/synth.py): layout and rates of the event
streams — 16-byte dvs_msgs::Event records, sorted in time inside a batch, batches back to back — and
the Poisson stream's statistics."""
import numpy as np

from esvio_amd.events import EVENT_DTYPE, event_times
from esvio_amd.synth import PoissonStream, SceneStream


def _check_batches(s, W, H, rate, n=4):
    t_prev = None
    counts = []
    for _ in range(n):
        L, R, t_end = s.next_batch()
        for ev in (L, R):
            assert ev.dtype == EVENT_DTYPE and ev.dtype.itemsize == 16
            assert ev["x"].max() < W and ev["y"].max() < H and set(np.unique(ev["polarity"])) <= {0, 1}
            t = event_times(ev)
            assert np.all(np.diff(t) >= 0)
            assert t[-1] <= t_end * 1e-6 + 1e-9
            if t_prev is not None:
                assert t[0] >= t_prev - 1e-9
        t_prev = t_end * 1e-6 - s.dur_us * 1e-6
        counts.append(len(L))
    return np.array(counts)


def test_scene_stream_layout_and_rate():
    W, H, rate = 346, 260, 1e6
    c = _check_batches(SceneStream(W, H, rate=rate, seed=3), W, H, rate)
    assert 0.5 * rate / 30 < c.mean() < 2.5 * rate / 30  # (small sensors overshoot: the edges set a floor)


def test_poisson_stream_is_homogeneous():
    W, H, rate = 320, 240, 3e6
    s = PoissonStream(W, H, rate=rate, seed=5)
    c = _check_batches(s, W, H, rate, n=6)
    lam = rate / 30
    assert abs(c.mean() - lam) < 5 * np.sqrt(lam / len(c))  # Poisson counts per batch
    L, R, _ = s.next_batch()
    hist = np.bincount(L["y"].astype(np.int64) * W + L["x"], minlength=W * H)
    # uniform over the sensor: the per-pixel count has mean = variance (within sampling error)
    assert abs(hist.var() / hist.mean() - 1.0) < 0.05
    assert abs(L["polarity"].mean() - 0.5) < 0.01
