"""PIN KIT for the oracle's `[OpenCV]`-marked restatements — skipped where OpenCV is not installed.

Every piece of arithmetic the reference path takes from OpenCV (un-vendored; ROS Noetic links 4.2.0) is restated in
oracle/esvio_oracle.cpp "as recalled" and is what keeps this repository's parity at "unpinned" (DESIGN.md section 2).
This image has no OpenCV, so nothing below runs here.  On any machine that has it

    pip install opencv-python==4.2.0.34        # (an x86-64 wheel: the SIMD128 build the reference node links)
    python -m pytest tests/test_oracle_vs_cv2.py -v

compares, on the committed inputs of tests/golden/cv2_pin_inputs.npz (made by tests/golden/make_cv2_pin_inputs.py),
the oracle against OpenCV itself, call shape by call shape as the reference makes them:

    buildOpticalFlowPyramid levels + Scharr   feature_tracker.cpp:410 (inside calcOpticalFlowPyrLK)
    calcOpticalFlowPyrLK, four call shapes    feature_tracker.cpp:410, 417, 490, 495   (bit-equal to oracle accum 2)
    cv::circle(.., r, .., -1), r = 10/20/30   feature_tracker.cpp:30, 148
    convertTo(CV_8U) ties / saturation        event_detector.cc:259
    CLAHE(40, 8x8) + normalize(MINMAX)        feature_tracker.cpp:377-381
    findFundamentalMat(FM_RANSAC, 1.0, 0.99)  feature_tracker.cpp:935
    goodFeaturesToTrack                       feature_tracker.cpp:228
    medianBlur                                event_detector.cc:263

With ESVIO_CV2_PIN_OUT=<file.npz> the OpenCV outputs are also written to that file; committed as
tests/golden/cv2_pin_outputs.npz they turn `test_oracle_against_returned_opencv_outputs` (which needs no OpenCV) into
a standing pin for everybody else.  Other OpenCV versions: the tests run and report, but only 4.2.x is the reference's
(later versions changed findFundamentalMat's RANSAC and parts of imgproc), so a mismatch there is an xfail, not a
verdict.  What no Python binding can reach — the cv::MatExpr fold of `255.0 * (M + 1.0) / 2.0` into one
convertTo(alpha, beta) — has a self-checking C++ snippet: tests/golden/cv_matexpr_pin.cpp.
"""
import os

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="the pin kit needs OpenCV (opencv-python==4.2.0.34 is the reference's version)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUTS = os.path.join(ROOT, "tests", "golden", "cv2_pin_inputs.npz")
IS_42 = cv2.__version__.startswith("4.2.")
other_version = pytest.mark.xfail(not IS_42, strict=False,
                                  reason="OpenCV %s is not the reference's 4.2.x: a difference here decides nothing" % cv2.__version__)

_collected = {}


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="module")
def inp():
    return np.load(INPUTS)


@pytest.fixture(scope="module", autouse=True)
def _dump_outputs():
    yield
    path = os.environ.get("ESVIO_CV2_PIN_OUT")
    if path:
        np.savez_compressed(path, cv_version=np.array(cv2.__version__), **_collected)


def _keep(name, value):
    _collected[name] = np.asarray(value)
    return value


LK_CRIT = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01)


def cv_lk(prev, nxt, pts, init=None, max_level=3):
    p0 = np.ascontiguousarray(pts, np.float32).reshape(-1, 1, 2)
    if init is None:
        p1, st, _ = cv2.calcOpticalFlowPyrLK(prev, nxt, p0, None, winSize=(21, 21), maxLevel=max_level, criteria=LK_CRIT)
    else:
        p1 = np.ascontiguousarray(init, np.float32).reshape(-1, 1, 2).copy()
        p1, st, _ = cv2.calcOpticalFlowPyrLK(prev, nxt, p0, p1, winSize=(21, 21), maxLevel=max_level, criteria=LK_CRIT,
                                             flags=cv2.OPTFLOW_USE_INITIAL_FLOW)
    return p1.reshape(-1, 2), st.reshape(-1)


@other_version
@pytest.mark.parametrize("img", ["ts_cur_left", "tex_a"])
def test_pyramid_levels_and_scharr(oracle, inp, img):
    """cv::buildOpticalFlowPyramid(img, Size(21,21), 3, withDerivatives) — what calcOpticalFlowPyrLK builds per image:
    pyrDown levels (binomial 5x5, BORDER_REFLECT_101, (s + 128) >> 8) and their Scharr derivatives (Ix, Iy int16)"""
    im = inp[img]
    n, pyr = cv2.buildOpticalFlowPyramid(im, (21, 21), 3, withDerivatives=True)
    assert n == oracle.pyr_levels(im.shape[1], im.shape[0], 21, 3)
    cur = im
    for lvl in range(n + 1):
        cv_img, cv_der = np.asarray(pyr[2 * lvl]), np.asarray(pyr[2 * lvl + 1])
        _keep("pyr_%s_%d" % (img, lvl), cv_img)
        _keep("scharr_%s_%d" % (img, lvl), cv_der)
        assert cv_img.shape == cur.shape and np.array_equal(cv_img, cur), ("pyrDown level", lvl)
        assert np.array_equal(cv_der.reshape(cur.shape + (2,)), oracle.scharr(cur)), ("Scharr level", lvl)
        cur = oracle.pyr_down(cur)
    assert np.array_equal(cv2.pyrDown(im), oracle.pyr_down(im))


@other_version
@pytest.mark.parametrize("pair", ["time_surfaces", "texture"])
def test_calc_optical_flow_pyr_lk_in_the_four_call_shapes(oracle, inp, pair):
    """the reference's four calls (feature_tracker.cpp:410,417,490,495): temporal forward (maxLevel 3), temporal
    backward (maxLevel 1, OPTFLOW_USE_INITIAL_FLOW seeded with prev_pts), stereo forward and stereo backward (maxLevel 3).
    On an x86 SIMD128 OpenCV 4.2 positions and status must equal oracle accum 2 BIT FOR BIT."""
    if pair == "time_surfaces":
        prevL, curL, curR, pts = inp["ts_prev_left"], inp["ts_cur_left"], inp["ts_cur_right"], inp["pts_prev"]
    else:
        prevL, curL, curR, pts = inp["tex_a"], inp["tex_b"], inp["tex_a"], inp["pts_tex"]
    # :410
    c_cur, c_st = cv_lk(prevL, curL, pts)
    o_cur, o_st = oracle.lk(prevL, curL, pts, max_level=3, accum=2)
    # :417 (reverse_pts = prev_pts as the initial flow)
    c_rev, c_rst = cv_lk(curL, prevL, c_cur, init=pts, max_level=1)
    o_rev, o_rst = oracle.lk(curL, prevL, o_cur, pts, max_level=1, flags=4, accum=2)
    # :490 / :495
    c_r, c_sr = cv_lk(curL, curR, c_cur)
    o_r, o_sr = oracle.lk(curL, curR, o_cur, max_level=3, accum=2)
    c_b, c_sb = cv_lk(curR, curL, c_r)
    o_b, o_sb = oracle.lk(curR, curL, o_r, max_level=3, accum=2)
    for name, cv_p, cv_s, o_p, o_s in (("fwd", c_cur, c_st, o_cur, o_st), ("rev", c_rev, c_rst, o_rev, o_rst),
                                       ("stereo", c_r, c_sr, o_r, o_sr), ("stereo_rev", c_b, c_sb, o_b, o_sb)):
        _keep("lk_%s_%s_pts" % (pair, name), cv_p)
        _keep("lk_%s_%s_status" % (pair, name), cv_s)
        assert np.array_equal(cv_s, o_s), (name, "status", int((cv_s != o_s).sum()))
        ok = cv_s == 1
        d = np.abs(cv_p[ok] - o_p[ok])
        assert np.array_equal(cv_p[ok].view(np.uint32), o_p[ok].view(np.uint32)), (
            name, "max |d| = %.3g px over %d points, %d differ" % (d.max(), ok.sum(), int((d > 0).any(axis=1).sum())))
    assert (c_st == 1).sum() > len(pts) // 2


@other_version
@pytest.mark.parametrize("r", [10, 20, 30])
def test_filled_circle(oracle, inp, r):
    """cv::circle(mask, Point, MIN_DIST, 255, -1) (feature_tracker.cpp:30,148): the midpoint disc, clipped at the border"""
    H, W = inp["ts_cur_left"].shape
    for k, (cx, cy) in enumerate(inp["circle_centres"]):
        a = np.zeros((H, W), np.uint8)
        cv2.circle(a, (int(cx), int(cy)), r, 255, -1)
        b = np.zeros((H, W), np.uint8)
        oracle.circle_fill(b, int(cx), int(cy), r, 255)
        _keep("circle_%d_%d" % (r, k), np.packbits(a))
        assert np.array_equal(a, b), (r, cx, cy)
    # on a CV_64F mask like the reference's, the same pixels
    m = np.zeros((H, W), np.float64)
    cv2.circle(m, (100, 100), r, 255.0, -1)
    b = np.zeros((H, W), np.uint8)
    oracle.circle_fill(b, 100, 100, r, 255)
    assert np.array_equal(m == 255.0, b == 255)


@other_version
def test_convert_to_u8_rounding_and_saturation(oracle, inp):
    """Mat::convertTo(CV_8U) of doubles = saturate_cast<uchar>(cvRound(v)): ties to even, saturation on both sides.  The
    oracle's side is its time-surface renderer fed with SAE stamps that produce exactly these values is not possible for
    arbitrary v, so the restated rule itself is held against OpenCV here: rint + clip, and — the quirk the oracle
    restates — |v| >= 2^31 goes through the x86 cvtsd2si indefinite value (INT_MIN), i.e. to 0."""
    v = inp["convert_values"]
    got = cv2.add(v.reshape(1, -1), 0.0, dtype=cv2.CV_8U).reshape(-1)  # saturate_cast<uchar>(double) per element
    _keep("convert_u8", got)
    small = np.abs(v) < 2.0 ** 31
    want = np.clip(np.rint(v[small]), 0, 255).astype(np.uint8)
    assert np.array_equal(got[small], want)
    assert np.all(got[~small & (v > 0)] == 0) or not IS_42, "cvRound overflow: the oracle renders these as 0"
    # and end to end on a surface: glibc exp on this machine + the folded MatExpr (alpha = beta = 127.5)
    import math
    W, H = 64, 48
    rng = np.random.default_rng(3)
    age = rng.uniform(0, 0.15, (H, W))
    pol = rng.integers(0, 2, (H, W))
    t = 10.0
    stamp = t - age
    det = oracle.Detector(W, H)
    zero = np.zeros((H, W))
    det.set_sae(0, zero, zero, np.where(pol == 0, stamp, 0.0), np.where(pol == 1, stamp, 0.0))
    e = np.array([[math.exp(-(t - stamp[y, x]) / 0.02) for x in range(W)] for y in range(H)]) * np.where(pol == 1, 1.0, -1.0)
    cv_ts = cv2.add(e * 127.5 + 127.5, 0.0, dtype=cv2.CV_8U)
    assert np.array_equal(cv_ts, det.time_surface(0, t))


@other_version
@pytest.mark.parametrize("img", ["ts_cur_left", "ts_cur_right", "tex_a"])
def test_clahe_and_normalize(oracle, inp, img):
    """cv::createCLAHE()->apply (clipLimit 40, 8x8 tiles) then cv::normalize(., 0, 255, NORM_MINMAX)
    (feature_tracker.cpp:377-381) — the `equalize: 1` branch"""
    im = inp[img]
    c = cv2.createCLAHE().apply(im)
    _keep("clahe_" + img, c)
    assert np.array_equal(c, oracle.clahe(im))
    nrm = cv2.normalize(c, None, 0, 255, cv2.NORM_MINMAX)
    _keep("clahe_norm_" + img, nrm)
    assert np.array_equal(nrm, oracle.normalize_minmax(oracle.clahe(im)))


@other_version
def test_find_fundamental_mat_masks(oracle, inp):
    """cv::findFundamentalMat(un_prev, un_cur, FM_RANSAC, 1.0, 0.99, status) (feature_tracker.cpp:935) on point sets of
    the kind the tracker hands it (sub-pixel motion between consecutive time surfaces): below 8 points the reference
    does not call; 8..14 (OpenCV 4.2 runs LMedS there), >= 15 RANSAC with cv::RNG(-1): the inlier masks"""
    for k in range(int(inp["f_sets"])):
        p1, p2 = inp["f_p1_%d" % k], inp["f_p2_%d" % k]
        F, mask = cv2.findFundamentalMat(p1, p2, cv2.FM_RANSAC, 1.0, 0.99)
        cnt, st, _ = oracle.find_fundamental(p1, p2, 1.0, 0.99)
        cv_mask = np.zeros(len(p1), np.uint8) if mask is None else mask.reshape(-1).astype(np.uint8)
        _keep("fmat_mask_%d" % k, cv_mask)
        if len(p1) < 8:
            continue  # (rejectWithF_event's guard: never called)
        assert np.array_equal(cv_mask, st), (k, len(p1), int(cv_mask.sum()), int(st.sum()))


@other_version
def test_good_features_to_track(oracle, inp):
    """cv::goodFeaturesToTrack(img, n_pts, max, 0.01, MIN_DIST_IMG, mask) (feature_tracker.cpp:228)"""
    for img, md, mask in (("tex_a", 30, None), ("tex_a", 10, inp["gftt_mask"]), ("ts_cur_left", 15, None)):
        im = inp[img]
        c = cv2.goodFeaturesToTrack(im, 100, 0.01, md, mask=mask)
        c = np.zeros((0, 2), np.float32) if c is None else c.reshape(-1, 2)
        _keep("gftt_%s_%d_%d" % (img, md, mask is not None), c)
        o = oracle.good_features_to_track(im, 100, 0.01, md, mask)
        assert c.shape == o.shape and np.array_equal(c, o), (img, md, len(c), len(o))


@other_version
@pytest.mark.parametrize("k", [1, 2, 3])
def test_median_blur(oracle, inp, k):
    """cv::medianBlur(surface, 2k+1) (event_detector.cc:263)"""
    im = inp["ts_cur_left"]
    m = cv2.medianBlur(im, 2 * k + 1)
    _keep("median_%d" % k, m)
    assert np.array_equal(m, oracle.median_blur(im, 2 * k + 1))
