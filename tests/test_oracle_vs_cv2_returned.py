"""The standing half of the OpenCV pin kit: the oracle against OpenCV OUTPUTS that somebody with OpenCV 4.2 has sent back.

tests/test_oracle_vs_cv2.py, run with ESVIO_CV2_PIN_OUT=tests/golden/cv2_pin_outputs.npz on a machine with
opencv-python==4.2.0.34, writes what OpenCV returns for the committed inputs (tests/golden/cv2_pin_inputs.npz).  Once
that file is committed this test — which needs no OpenCV — holds the oracle to it everywhere, and the "parity unpinned"
of DESIGN.md section 2 ends for every restatement it covers.  No such file has been returned yet: the test skips.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUTS = os.path.join(ROOT, "tests", "golden", "cv2_pin_inputs.npz")
OUTPUTS = os.path.join(ROOT, "tests", "golden", "cv2_pin_outputs.npz")

returned = pytest.mark.skipif(not os.path.exists(OUTPUTS),
                              reason="no OpenCV outputs returned yet (tests/golden/cv2_pin_outputs.npz): parity stays unpinned")


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def test_pin_inputs_are_the_committed_ones():
    """(always runs) the inputs an outside OpenCV run is asked to process are in the tree and readable"""
    d = np.load(INPUTS)
    assert d["ts_cur_left"].shape == (260, 346) and d["ts_cur_left"].dtype == np.uint8
    assert int(d["f_sets"]) >= 8 and len(d["pts_prev"]) > 50


@returned
def test_oracle_against_returned_opencv_outputs(oracle):
    inp, out = np.load(INPUTS), np.load(OUTPUTS)
    ver = str(out["cv_version"])
    assert ver.startswith("4.2."), "outputs of OpenCV %s: only 4.2.x is the reference's" % ver
    checked = 0
    # pyramids + Scharr
    for img in ("ts_cur_left", "tex_a"):
        cur, lvl = inp[img], 0
        while "pyr_%s_%d" % (img, lvl) in out:
            assert np.array_equal(out["pyr_%s_%d" % (img, lvl)], cur)
            assert np.array_equal(out["scharr_%s_%d" % (img, lvl)].reshape(cur.shape + (2,)), oracle.scharr(cur))
            cur, lvl, checked = oracle.pyr_down(cur), lvl + 1, checked + 2
    # LK, the four call shapes chained exactly as tests/test_oracle_vs_cv2.py chains them
    for pair, (a, b, c, p) in (("time_surfaces", ("ts_prev_left", "ts_cur_left", "ts_cur_right", "pts_prev")),
                               ("texture", ("tex_a", "tex_b", "tex_a", "pts_tex"))):
        if "lk_%s_fwd_pts" % pair not in out:
            continue
        prevL, curL, curR, pts = inp[a], inp[b], inp[c], inp[p]
        o_cur, o_st = oracle.lk(prevL, curL, pts, max_level=3, accum=2)
        o_rev, o_rst = oracle.lk(curL, prevL, o_cur, pts, max_level=1, flags=4, accum=2)
        o_r, o_sr = oracle.lk(curL, curR, o_cur, max_level=3, accum=2)
        o_b, o_sb = oracle.lk(curR, curL, o_r, max_level=3, accum=2)
        for name, o_p, o_s in (("fwd", o_cur, o_st), ("rev", o_rev, o_rst), ("stereo", o_r, o_sr), ("stereo_rev", o_b, o_sb)):
            cv_p, cv_s = out["lk_%s_%s_pts" % (pair, name)], out["lk_%s_%s_status" % (pair, name)]
            assert np.array_equal(cv_s, o_s), (pair, name)
            ok = cv_s == 1
            assert np.array_equal(cv_p[ok].view(np.uint32), o_p[ok].view(np.uint32)), (pair, name)
            checked += 1
    # discs
    H, W = inp["ts_cur_left"].shape
    for r in (10, 20, 30):
        for k, (cx, cy) in enumerate(inp["circle_centres"]):
            key = "circle_%d_%d" % (r, k)
            if key in out:
                b = np.zeros((H, W), np.uint8)
                oracle.circle_fill(b, int(cx), int(cy), r, 255)
                assert np.array_equal(out[key], np.packbits(b)), key
                checked += 1
    # CLAHE + normalize, median, goodFeaturesToTrack, findFundamentalMat
    for img in ("ts_cur_left", "ts_cur_right", "tex_a"):
        if "clahe_" + img in out:
            assert np.array_equal(out["clahe_" + img], oracle.clahe(inp[img]))
            assert np.array_equal(out["clahe_norm_" + img], oracle.normalize_minmax(oracle.clahe(inp[img])))
            checked += 2
    for k in (1, 2, 3):
        if "median_%d" % k in out:
            assert np.array_equal(out["median_%d" % k], oracle.median_blur(inp["ts_cur_left"], 2 * k + 1))
            checked += 1
    for img, md, mask in (("tex_a", 30, None), ("tex_a", 10, inp["gftt_mask"]), ("ts_cur_left", 15, None)):
        key = "gftt_%s_%d_%d" % (img, md, mask is not None)
        if key in out:
            o = oracle.good_features_to_track(inp[img], 100, 0.01, md, mask)
            assert out[key].shape == o.shape and np.array_equal(out[key], o), key
            checked += 1
    for k in range(int(inp["f_sets"])):
        key = "fmat_mask_%d" % k
        if key in out and len(inp["f_p1_%d" % k]) >= 8:
            _, st, _ = oracle.find_fundamental(inp["f_p1_%d" % k], inp["f_p2_%d" % k], 1.0, 0.99)
            assert np.array_equal(out[key], st), key
            checked += 1
    assert checked >= 30, "the returned file covers too little (%d comparisons)" % checked
