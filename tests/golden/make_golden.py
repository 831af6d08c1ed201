#!/usr/bin/env python3
"""Regenerates tests/golden/scene_192x144.npz (lk_accum 1: exact LK sums) and scene_192x144_f32.npz (lk_accum 2:
the float sums in the order of the reference's x86 OpenCV build — the default mode).

The reference cannot be built or run in this image (its sources for the path need
Eigen/OpenCV/ROS), and it ships no fixtures, so these are REGRESSION vectors produced by the CPU
oracle (oracle/liboracle.so), not reference outputs: inputs (event batches) + the oracle's
outputs at every stage.  They pin the oracle against drift and give the GPU path a fixed,
seed-independent target.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from esvio_amd.events import event_times  # noqa: E402
from esvio_amd.synth import SceneStream  # noqa: E402
from oracle import oracle as O  # noqa: E402

W, H, NB = 192, 144, 5
KW = dict(max_cnt=60, min_dist=6, f_ransac=1, flow_back=1)


def main():
    for accum, name in ((1, "scene_192x144.npz"), (2, "scene_192x144_f32.npz")):
        make(accum, name)


def make(accum, fname):
    s = SceneStream(W, H, rate=1.5e5, n_rect=6, seed=2024, size=(25.0, 60.0), speed=(120.0, 260.0),
                    disparity=7, t0_us=1_700_000_000_000_000)
    tr = O.Tracker(O.make_config(W, H, lk_accum=accum, **KW))
    out = {"W": W, "H": H, "n_batches": NB, "cfg_lk_accum": accum}
    for k, v in KW.items():
        out["cfg_" + k] = v
    for b in range(NB):
        L, R, _ = s.next_batch()
        pub = b != 2
        t = event_times(L)[-1]
        r = tr.track_event(t, L, R, pub)
        det = tr.detector()
        out["L%d" % b] = L.view(np.uint8).reshape(-1, 16)
        out["R%d" % b] = R.view(np.uint8).reshape(-1, 16)
        out["pub%d" % b] = pub
        out["t%d" % b] = t
        out["tsL%d" % b] = tr.time_surface(0)
        out["tsR%d" % b] = tr.time_surface(1)
        out["flags%d" % b] = det.corner_flags(L)
        for k in ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity", "ids_right",
                  "cur_right_pts", "cur_un_right_pts", "right_pts_velocity"):
            out["%s%d" % (k, b)] = getattr(r, k)
        if b == NB - 1:
            for cam in (0, 1):
                for name, p in zip(("L0", "L1", "S0", "S1"), det.get_sae(cam)):
                    out["sae_cam%d_%s" % (cam, name)] = p
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), fname)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB; tracks per batch:",
          [len(out["ids%d" % b]) for b in range(NB)], [len(out["ids_right%d" % b]) for b in range(NB)],
          "corner flags:", [int(out["flags%d" % b].sum()) for b in range(NB)])


if __name__ == "__main__":
    main()
