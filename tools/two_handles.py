#!/usr/bin/env python3
"""Two live handles in one process (the runtime hands out four hardware queues per priority level and PROCESS, a
handle uses four or five streams): the replay schedule of bench.py on handle A alone, on A with an idle handle B
beside it, on B with A idle, and on A again after B is closed.   python tools/two_handles.py [--steps 20]"""
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--passes", type=int, default=4)
    a = ap.parse_args()
    W, H = 640, 480
    n = a.warmup + a.steps * a.passes
    s = SceneStream(W, H, rate=5e6, seed=12345)
    bat, fc, pubs = [], FreqControl(15), []
    for _ in range(n):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        bat.append((FE.EventBuffer(L, FE.DEVICE).arg, FE.EventBuffer(R, FE.DEVICE).arg, t, len(L) + len(R)))
        pubs.append(fc.pub_this_frame(t))
        if pubs[-1]:
            fc.published()

    def make():
        ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=300, min_dist=10, flow_back=1, f_ransac=1))
        ft.set_lazy_new_stereo(True)
        ft.set_host_threads(8)
        ft.set_launch_thread(True)
        ft.reserve(200000, 200000, host_batches=False)
        return ft

    def run(ft):
        ft.reset()
        ann, res, t0 = 0, [], 0.0
        for i in range(n):
            if i >= a.warmup and (i - a.warmup) % a.steps == 0:
                if i > a.warmup:
                    ft.finish(copy=False)
                    res.append((time.perf_counter() - t0) / a.steps * 1e3)
                t0 = time.perf_counter()
            while ann < min(i + 3, n - 1):
                ann += 1
                ft.set_next_batch(bat[ann][2], bat[ann][0], bat[ann][1], pubs[ann])
            ft.trackEvent(bat[i][2], bat[i][0], bat[i][1], pubs[i], copy=False)
        ft.finish(copy=False)
        res.append((time.perf_counter() - t0) / a.steps * 1e3)
        return " ".join("%.4f" % v for v in res)

    A = make()
    print("A alone:                ", run(A))
    print("A alone, again:         ", run(A))
    B = make()
    print("A, idle B beside it:    ", run(A))
    print("B, idle A beside it:    ", run(B))
    print("A again:                ", run(A))
    B.close()
    print("A after B is closed:    ", run(A))
    A.close()


if __name__ == "__main__":
    main()
