"""The oracle's restatements of OpenCV's image primitives against third-party numerics (scipy.ndimage
/ numpy), on random images — not OpenCV itself (absent here), but code the builder did not write,
driven by the published definition of each operation:

  pyrDown      separable [1 4 6 4 1] with BORDER_REFLECT_101, (sum + 128) >> 8, every second pixel
  Scharr       [3 10 3]^T x [-1 0 1] (and transposed), BORDER_REFLECT_101, int16
  medianBlur   (2k+1)^2 median, BORDER_REPLICATE
  normalize    NORM_MINMAX to 0..255 with cv::saturate_cast<uchar>(cvRound(.))
  cornerMinEigenVal  Sobel/3060 -> 3x3 box of products -> smaller eigenvalue (float64 evaluation, tolerance)

Agreement is exact for the integer operations."""
import numpy as np
import pytest

nd = pytest.importorskip("scipy.ndimage")


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w), dtype=np.uint8)


@pytest.mark.parametrize("h,w", [(48, 64), (61, 97), (130, 173)])
def test_pyr_down_is_the_binomial_filter(oracle, h, w):
    img = _img(h, w, h)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    a = nd.correlate1d(img.astype(np.int64), k, axis=0, mode="mirror")  # 'mirror' = reflect-101
    a = nd.correlate1d(a, k, axis=1, mode="mirror")
    ref = ((a + 128) >> 8)[::2, ::2].astype(np.uint8)
    assert np.array_equal(oracle.pyr_down(img), ref)


@pytest.mark.parametrize("h,w", [(48, 64), (61, 97)])
def test_scharr_is_the_scharr_operator(oracle, h, w):
    img = _img(h, w, 7 * h).astype(np.int64)
    smooth, diff = np.array([3, 10, 3]), np.array([-1, 0, 1])
    ix = nd.correlate1d(nd.correlate1d(img, smooth, axis=0, mode="mirror"), diff, axis=1, mode="mirror")
    iy = nd.correlate1d(nd.correlate1d(img, diff, axis=0, mode="mirror"), smooth, axis=1, mode="mirror")
    got = oracle.scharr(img.astype(np.uint8))
    assert np.array_equal(got[..., 0], ix.astype(np.int16)) and np.array_equal(got[..., 1], iy.astype(np.int16))


@pytest.mark.parametrize("k", [1, 2, 3])
def test_median_blur_is_the_window_median(oracle, k):
    img = _img(50, 70, k)
    ref = nd.median_filter(img, size=2 * k + 1, mode="nearest")  # 'nearest' = BORDER_REPLICATE
    assert np.array_equal(oracle.median_blur(img, 2 * k + 1), ref)


def test_normalize_minmax(oracle):
    img = _img(40, 60, 3)
    img = (img // 3 + 20).astype(np.uint8)  # a sub-range, so that the scaling does something
    lo, hi = float(img.min()), float(img.max())
    ref = np.rint((img.astype(np.float64) - lo) * (255.0 / (hi - lo))).clip(0, 255).astype(np.uint8)
    assert np.array_equal(oracle.normalize_minmax(img), ref)


def test_corner_min_eigen_val_definition(oracle):
    img = _img(60, 80, 11)
    f = img.astype(np.float64)
    sob_s, sob_d = np.array([1, 2, 1.0]), np.array([-1, 0, 1.0])
    dx = nd.correlate1d(nd.correlate1d(f, sob_s, axis=0, mode="mirror"), sob_d, axis=1, mode="mirror") / 3060.0
    dy = nd.correlate1d(nd.correlate1d(f, sob_d, axis=0, mode="mirror"), sob_s, axis=1, mode="mirror") / 3060.0
    box = lambda a: nd.uniform_filter(a, size=3, mode="mirror") * 9.0  # (un-normalised 3x3 box)
    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    ref = (a + c) - np.sqrt((a - c) ** 2 + b * b)
    _, eig = oracle.good_features_to_track(img, 10, quality=0.01, min_distance=5, want_eig=True)
    assert np.allclose(eig, ref, rtol=2e-4, atol=1e-7), np.abs(eig - ref).max()
