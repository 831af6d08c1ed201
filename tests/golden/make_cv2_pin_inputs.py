#!/usr/bin/env python3
"""Inputs of the OpenCV pin kit (tests/test_oracle_vs_cv2.py) -> tests/golden/cv2_pin_inputs.npz.

    python tests/golden/make_cv2_pin_inputs.py

What is in the file is DATA: images, point lists, point-pair sets.  The time surfaces and the tracked corners come
from this repository's own oracle (oracle/, test infrastructure) run on this repository's synthetic stream — they are
inputs for OpenCV to be run on, not expected outputs; the expected outputs are whatever an OpenCV 4.2 installation
returns for them.  Seeded: the same file comes out again.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esvio_amd.events import event_times  # noqa: E402
from esvio_amd.synth import SceneStream  # noqa: E402
from oracle import oracle as O  # noqa: E402

W, H = 346, 260  # (DAVIS346: four LK levels — 44x33 at level 3 is still larger than the 21 px window)


def cam_dict(c):
    return dict(fx=c.fx, fy=c.fy, cx=c.cx, cy=c.cy, k1=c.k1, k2=c.k2, p1=c.p1, p2=c.p2)


def main():
    O.build()
    rng = np.random.default_rng(20260930)
    out = {}
    # ---- time surfaces of three consecutive frames of the scene stream, and the corners tracked on them
    s = SceneStream(W, H, rate=1.5e6, seed=5, n_rect=12, size=(30.0, 90.0))
    cfg = O.make_config(W, H, max_cnt=150, min_dist=10, f_ransac=0, lk_accum=2)
    tr = O.Tracker(cfg)
    frames = []
    for f in range(9):
        L, R, _ = s.next_batch()
        r = tr.track_event(event_times(L)[-1], L, R, True)
        frames.append(dict(tsL=tr.time_surface(0).copy(), tsR=tr.time_surface(1).copy(), ids=r.ids.copy(),
                           pts=r.cur_pts.copy()))
    a, b = frames[5], frames[6]
    out["ts_prev_left"], out["ts_cur_left"], out["ts_cur_right"] = a["tsL"], b["tsL"], b["tsR"]
    out["pts_prev"] = a["pts"].astype(np.float32)  # corners on ts_prev_left
    out["pts_cur"] = b["pts"].astype(np.float32)   # ... on ts_cur_left
    # ---- a texture pair with a known sub-pixel shift (more gradient than a time surface has)
    base = rng.integers(0, 256, (H + 16, W + 16)).astype(np.float64)
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    for _ in range(3):
        base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, base)
        base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, base)
    base = (base - base.min()) / (base.max() - base.min()) * 255
    out["tex_a"] = np.rint(base[8:8 + H, 8:8 + W]).astype(np.uint8)
    sh = 0.6 * base[8:8 + H, 10:10 + W] + 0.4 * base[8:8 + H, 11:11 + W]  # shift by 2.4 px in x ...
    sh = 0.7 * sh + 0.3 * np.roll(sh, -1, axis=0)                          # ... and 0.3 px in y
    out["tex_b"] = np.rint(sh).astype(np.uint8)
    gx, gy = np.meshgrid(np.linspace(30, W - 30, 12), np.linspace(30, H - 30, 9))
    out["pts_tex"] = (np.stack([gx.ravel(), gy.ravel()], 1) + rng.uniform(-3, 3, (108, 2))).astype(np.float32)
    # ---- point pairs as rejectWithF_event hands them to cv::findFundamentalMat (feature_tracker.cpp:910-935):
    # corners matched by id between consecutive frames, lifted through the left camera model
    cam = cam_dict(cfg.cam[0])

    def lifted(p):
        q = np.empty((len(p), 2), np.float32)
        for i, (u, v) in enumerate(p):
            x, y, z = O.lift_projective(cam, float(u), float(v))
            q[i] = (cfg.focal_length * x / z + W / 2.0, cfg.focal_length * y / z + H / 2.0)
        return q

    sets = []
    for fa, fb in zip(frames[3:8], frames[4:9]):
        ia = {int(i): k for k, i in enumerate(fa["ids"])}
        common = [(ia[int(i)], k) for k, i in enumerate(fb["ids"]) if int(i) in ia]
        pa = lifted(fa["pts"][[c[0] for c in common]])
        pb = lifted(fb["pts"][[c[1] for c in common]])
        sets.append((pa, pb))
    # the sizes OpenCV 4.2 treats differently: < 8 (the reference does not call), 8..14 (LMedS inside FM_RANSAC),
    # >= 15 (RANSAC); a few gross outliers in the larger ones
    sets.sort(key=lambda ab: -len(ab[0]))
    fsets = []
    for k, n in enumerate((7, 8, 11, 14, 15, 16, 40)):
        pa, pb = sets[k % 3]
        idx = rng.permutation(len(pa))[:n]
        fsets.append((pa[idx], pb[idx]))
    fsets.append(sets[0])  # (no gross outliers: consecutive-frame motion only)
    for pa, pb in sets[:3]:
        pb = pb.copy()
        bad = rng.permutation(len(pb))[:max(3, len(pb) // 10)]
        pb[bad] += rng.uniform(-6, 6, (len(bad), 2)).astype(np.float32)
        fsets.append((pa, pb))
    out["f_sets"] = np.array(len(fsets))
    for i, (pa, pb) in enumerate(fsets):
        out["f_p1_%d" % i], out["f_p2_%d" % i] = pa, pb
    # ---- discs of Event_setMask / Event_FeaturesToTrack: centres incl. ones that clip at the border
    out["circle_centres"] = np.array([[100, 100], [0, 0], [5, 250], [345, 259], [173, 3], [340, 130]], np.int32)
    # ---- a mask for goodFeaturesToTrack (255 = allowed, like mask_image)
    m = np.full((H, W), 255, np.uint8)
    m[60:120, 100:220] = 0
    out["gftt_mask"] = m
    # ---- values for convertTo(CV_8U): exact ties, saturation on both sides, the int32 overflow of cvRound
    t = np.concatenate([np.arange(-3, 260) + 0.5, np.arange(-3, 260).astype(np.float64), rng.uniform(-10, 270, 2000),
                        np.array([1e9, -1e9, 2147483647.5, 3e9, -3e9, 4.2e9, 1e300, -1e300])])
    out["convert_values"] = t
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cv2_pin_inputs.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith(("ts", "pts", "tex"))})
    print("F sets:", [len(p[0]) for p in fsets])


if __name__ == "__main__":
    main()
