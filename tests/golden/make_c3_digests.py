#!/usr/bin/env python3
"""Digests of the oracle's outputs at BASELINE C3's own size (640x480 stereo, 5 Mev/s per camera, the bench stream's
seed and publish pattern) -> tests/golden/c3_640x480_digests.json.  The 192x144 fixtures carry whole vectors; at this
size the batches alone would be 5 MB per frame, so the stream is regenerated from its seed and what is committed is a
SHA-256 of every input array and of every output the oracle produces for it: a tripwire against a drift of the oracle
(or of the synthetic stream) at the size the bench and the live GPU comparisons run at.  Like the other fixtures:
oracle output, not reference output.   python tests/golden/make_c3_digests.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esvio_amd.events import event_times  # noqa: E402
from esvio_amd.node import FreqControl  # noqa: E402
from esvio_amd.synth import SceneStream  # noqa: E402

W, H, NB, SEED = 640, 480, 8, 12345
KW = dict(max_cnt=300, min_dist=10, flow_back=1, f_ransac=1)
KEYS = ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity", "ids_right", "cur_right_pts", "cur_un_right_pts",
        "right_pts_velocity")


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.view(np.uint8).reshape(-1).tobytes()).hexdigest()


def frames():
    """the bench's stream and publish pattern (bench.py: SceneStream(seed), FreqControl(15))"""
    s = SceneStream(W, H, rate=5e6, batch_hz=30.0, seed=SEED)
    fc = FreqControl(15)
    for _ in range(NB):
        L, R, _t = s.next_batch()
        t = event_times(L)[-1]
        pub = fc.pub_this_frame(t)
        if pub:
            fc.published()
        yield L, R, t, pub


def run(track, time_surface, corner_flags):
    """track(t, L, R, pub) -> result with KEYS; returns the digests per frame"""
    out = []
    for L, R, t, pub in frames():
        r = track(t, L, R, pub)
        d = dict(L=digest(L), R=digest(R), t=repr(float(t)), pub=bool(pub), tsL=digest(time_surface(0)),
                 tsR=digest(time_surface(1)), flags=digest(corner_flags(L)), n_left=int(len(r.ids)), n_right=int(len(r.ids_right)))
        for k in KEYS:
            d[k] = digest(getattr(r, k))
        out.append(d)
    return out


def main():
    from oracle import oracle as O
    O.build()
    res = dict(W=W, H=H, n_batches=NB, seed=SEED, cfg=KW, modes={})
    for accum in (2, 1):
        tr = O.Tracker(O.make_config(W, H, lk_accum=accum, **KW))
        res["modes"][str(accum)] = run(lambda t, L, R, pub: tr.track_event(t, L, R, pub), tr.time_surface,
                                       lambda L: tr.detector().corner_flags(L))
        print("lk_accum", accum, "tracks per frame:", [(d["n_left"], d["n_right"]) for d in res["modes"][str(accum)]])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c3_640x480_digests.json")
    json.dump(res, open(path, "w"), indent=1)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
