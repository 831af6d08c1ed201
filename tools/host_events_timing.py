#!/usr/bin/env python3
"""Replay schedule over HOST-resident batches (what INTEGRATION.md's binding passes), timed alone:
    python tools/host_events_timing.py [--steps 60] [--ahead 4] [--pinned] [--direct]
ESVIO_FE_STAGE_THREADS=0/1/2/4 selects the staging (0: the runtime's pageable copy)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esvio_amd import frontend as FE  # noqa: E402
from esvio_amd.events import event_times  # noqa: E402
from esvio_amd.node import FreqControl  # noqa: E402
from esvio_amd.synth import SceneStream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ahead", type=int, default=4)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--rate", type=float, default=5e6)
    ap.add_argument("--pinned", action="store_true", help="batches in hipHostMalloc'ed memory")
    ap.add_argument("--direct", action="store_true", help="no esvio_fe_set_next_batch: one batch at a time")
    ap.add_argument("--host-threads", type=int, default=8)
    a = ap.parse_args()
    W, H = a.width, a.height
    s = SceneStream(W, H, rate=a.rate, seed=12345)
    n = a.warmup + 3 * a.steps  # three timed passes over the continued stream
    batches = []
    for _ in range(n):
        L, R, _ = s.next_batch()
        if a.pinned:  # pinned memory of the library's own HIP runtime
            L, R = FE.EventBuffer(L).array, FE.EventBuffer(R).array
        batches.append((L, R, event_times(L)[-1]))
    fc = FreqControl(15)
    pubs = []
    for b in batches:
        pubs.append(fc.pub_this_frame(b[2]))
        if pubs[-1]:
            fc.published()
    ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=300, min_dist=10, flow_back=1, f_ransac=1))
    if not a.direct:
        ft.set_lazy_new_stereo(True)
    ft.set_host_threads(a.host_threads)
    announced = 0
    t0 = 0.0
    ev = 0
    res = []
    for i, (L, R, t) in enumerate(batches):
        if i >= a.warmup and (i - a.warmup) % a.steps == 0:
            if i > a.warmup:
                ft.finish(copy=False)
                res.append((time.perf_counter() - t0, ev))
            t0 = time.perf_counter()
            ev = 0
        if not a.direct:
            while announced < min(i + a.ahead, n - 1):
                announced += 1
                ft.set_next_batch(batches[announced][2], batches[announced][0], batches[announced][1], pubs[announced])
        ft.trackEvent(t, L, R, pubs[i], copy=False)
        ev += len(L) + len(R)
    ft.finish(copy=False)
    res.append((time.perf_counter() - t0, ev))
    print("host-resident %s%s: %s ms/step, %s GB/s H2D  (ESVIO_FE_STAGE_THREADS=%s)"
          % ("direct" if a.direct else "replay ahead=%d" % a.ahead, " pinned" if a.pinned else "",
             " / ".join("%.4f" % (dt / a.steps * 1e3) for dt, _ in res),
             " / ".join("%.1f" % (e * 16 / dt / 1e9) for dt, e in res), os.environ.get("ESVIO_FE_STAGE_THREADS", "default")))
    ft.close()


if __name__ == "__main__":
    main()
