"""rejectWithF_event's findFundamentalMat (feature_tracker.cpp:935) against answers that do NOT come
from the oracle: synthetic two-view scenes whose fundamental matrix and inlier set are known
analytically.  Both the product's host stage (esvio_fe_find_fundamental_mat, fe_host.cpp) and the
oracle's restatement are held to them — the two are same-author restatements of OpenCV's RANSAC,
so their agreement with each other (tests/test_abi.py) pins neither.

What these tests can and cannot say: a RANSAC run returns the consensus set of the best 7-point
hypothesis; with noise-free inliers and outliers placed > 3 px off their epipolar lines that set is
the true inlier set for ANY correct implementation (the seed only changes which all-inlier sample
finds it).  With pixel noise the set depends on the hypothesis drawn; then only bounds hold.  (The
7-point solver's null space follows cv::SVDecomp — tests/test_ransac_nullspace.py.)"""
import numpy as np
import pytest

from esvio_amd import frontend as FE


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def two_view(n, n_out, sigma, seed, planar=False):
    """n points seen by two cameras (K = focal 460, centre 320,240 — rejectWithF_event's virtual
    camera), the last n_out of them moved > 3 px off their epipolar line in image 2.
    Returns p1, p2 (float32 pixels), F (x2^T F x1 = 0), truth (1 = inlier)."""
    rng = np.random.default_rng(seed)
    K = np.array([[460.0, 0, 320], [0, 460, 240], [0, 0, 1]])
    R = _rot(*rng.uniform(-0.05, 0.05, 3))
    t = rng.uniform(-0.3, 0.3, 3) + np.array([0.25, 0, 0])
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n),
                  np.full(n, 5.0) if planar else rng.uniform(3, 9, n)], 1)
    x1 = (K @ X.T).T
    x2 = (K @ (R @ X.T + t[:, None])).T
    p1, p2 = x1[:, :2] / x1[:, 2:], x2[:, :2] / x2[:, 2:]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    truth = np.ones(n, np.uint8)
    for i in range(n - n_out, n):  # push along the epipolar line's normal, both directions
        l = F @ np.array([p1[i, 0], p1[i, 1], 1.0])
        nrm = l[:2] / np.linalg.norm(l[:2])
        p2[i] += nrm * rng.uniform(4, 40) * rng.choice([-1, 1])
        truth[i] = 0
    p1 = p1 + rng.normal(0, sigma, p1.shape) if sigma else p1
    p2 = p2 + rng.normal(0, sigma, p2.shape) if sigma else p2
    return p1.astype(np.float32), p2.astype(np.float32), F, truth


def epi_dist(F, p1, p2):
    """max of the two point-to-epipolar-line distances (what findFundamentalMat thresholds, squared)"""
    a = np.c_[p1.astype(np.float64), np.ones(len(p1))]
    b = np.c_[p2.astype(np.float64), np.ones(len(p2))]
    l2, l1 = a @ F.T, b @ F
    s = np.abs((b * l2).sum(1))
    return np.maximum(s / np.hypot(l2[:, 0], l2[:, 1]), s / np.hypot(l1[:, 0], l1[:, 1]))


def _impls(oracle):
    return (("product", lambda a, b: FE.find_fundamental_mat(a, b, 1.0, 0.99)),
            ("product_4_threads", lambda a, b: FE.find_fundamental_mat(a, b, 1.0, 0.99, threads=4)),
            ("oracle", lambda a, b: oracle.find_fundamental(a, b, 1.0, 0.99)[:2]))


@pytest.mark.parametrize("n,n_out", [(60, 0), (120, 24), (300, 90), (40, 12)])
def test_noise_free_scene_gives_exactly_the_true_inlier_set(oracle, n, n_out):
    for seed in range(4):
        p1, p2, F, truth = two_view(n, n_out, 0.0, 100 * n + seed)
        d = epi_dist(F, p1, p2)
        assert d[truth == 1].max() < 1e-2 and (n_out == 0 or d[truth == 0].min() > 3.0)  # no knife-edge points
        for name, f in _impls(oracle):
            cnt, status = f(p1, p2)
            assert np.array_equal(status, truth), (name, seed, int((status != truth).sum()))
            assert cnt == int(truth.sum()), name


@pytest.mark.parametrize("sigma", [0.1, 0.3])
def test_noisy_scene_bounds(oracle, sigma):
    """pixel noise: every gross outlier is rejected, most true inliers are kept, and every kept point
    is within the threshold (plus what a hypothesis fitted to noisy points can be off by) of the
    TRUE epipolar geometry"""
    kept_frac = []
    for seed in range(6):
        p1, p2, F, truth = two_view(200, 50, sigma, 7000 + seed)
        d = epi_dist(F, p1, p2)
        for name, f in _impls(oracle):
            cnt, status = f(p1, p2)
            assert cnt == int(status.sum()), name
            assert not status[truth == 0].any(), (name, seed)
            assert d[status == 1].max() < 1.0 + 6 * sigma, (name, seed, d[status == 1].max())
            kept_frac.append(status[truth == 1].mean())
    # (the consensus set of an unrefined 7-point hypothesis fitted to noisy points: measured 0.78-0.99)
    lo, mean = (0.9, 0.95) if sigma <= 0.1 else (0.7, 0.8)
    assert min(kept_frac) > lo and np.mean(kept_frac) > mean, (min(kept_frac), np.mean(kept_frac))


def test_degenerate_inputs(oracle):
    """fewer than 8 points -> nothing is accepted (feature_tracker.cpp:912 guards with >= 8, the
    function itself needs >= 7); all points on one line -> getSubset's collinearity test never
    yields a sample, OpenCV's run() returns false on the first iteration and the mask stays zero;
    a planar scene still has a valid F for every all-inlier sample (the inliers survive)"""
    p1, p2, F, truth = two_view(30, 0, 0.0, 5)
    for name, f in _impls(oracle):
        assert f(p1[:6], p2[:6])[0] == 0, name
    x = 50.0 + 10.0 * np.arange(40)  # integer coordinates: the collinearity test's cross products are exactly 0
    line1 = np.stack([x, 2 * x + 7], 1).astype(np.float32)
    line2 = np.stack([x + 3, 2 * x + 11], 1).astype(np.float32)
    for name, f in _impls(oracle):
        cnt, status = f(line1, line2)
        assert cnt == 0 and not status.any(), name
    p1, p2, F, truth = two_view(80, 16, 0.0, 11, planar=True)
    for name, f in _impls(oracle):
        cnt, status = f(p1, p2)
        assert status[truth == 1].all(), name  # (outliers of a planar scene may fit the chosen F)
