#!/usr/bin/env python3
"""Plain calls (one batch in flight, nothing announced) from pageable host memory, variants of the library's
environment switches timed in turns inside one process:
    python tools/plain_call_ab.py [--rounds 4] [--steps 60] - ESVIO_FE_STAGE_PACK=0 ESVIO_FE_STAGE_THREADS=4
Each turn is a fresh handle (the switches are read when its stager is made), alone in the process while it runs.
HT=n in the environment: esvio_fe_set_host_threads(n) instead of 8; --pinned / --device: the batches in hipHostMalloc'ed
memory / already in HBM."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esvio_amd import frontend as FE  # noqa: E402
from esvio_amd.events import event_times  # noqa: E402
from esvio_amd.node import FreqControl  # noqa: E402
from esvio_amd.synth import SceneStream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help="K=V[,K=V...] per variant ('-' = nothing set)")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rate", type=float, default=5e6)
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--device", action="store_true", help="the batches already in HBM (torch tensors)")
    ap.add_argument("--gap-ms", type=float, default=0.0, help="sleep between the calls (a live sensor: 33 ms); the figure is then "
                    "the mean time INSIDE the calls")
    a = ap.parse_args()
    W, H = 640, 480
    s = SceneStream(W, H, rate=a.rate, seed=12345)
    batches, keep = [], []
    for _ in range(a.warmup + a.steps):
        L, R, _ = s.next_batch()
        if a.pinned:
            L, R = FE.EventBuffer(L).array, FE.EventBuffer(R).array
        t_last = event_times(L)[-1]
        if a.device:
            import torch
            keep.append((torch.from_numpy(L.view("u1").reshape(-1)).cuda(), torch.from_numpy(R.view("u1").reshape(-1)).cuda()))
            L, R = (keep[-1][0].data_ptr(), len(L)), (keep[-1][1].data_ptr(), len(R))
        batches.append((L, R, t_last))
    fc = FreqControl(15)
    pubs = []
    for b in batches:
        pubs.append(fc.pub_this_frame(b[2]))
        if pubs[-1]:
            fc.published()
    res = {v: [] for v in a.variants}
    for r in range(a.rounds):
        for v in a.variants:
            kv = [] if v == "-" else [x.split("=") for x in v.split(",")]
            for k, val in kv:
                os.environ[k] = val
            try:
                ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=300, min_dist=10, flow_back=1, f_ransac=1))
                ft.set_host_threads(int(os.environ.get("HT", "8")))
                nmax = lambda k: max(b[k][1] if a.device else len(b[k]) for b in batches)  # noqa: E731
                ft.reserve(nmax(0), nmax(1), host_batches=not a.device)
                inside = 0.0
                for i, (L, R, t) in enumerate(batches):
                    if i == a.warmup:
                        t0 = time.perf_counter()
                        inside = 0.0
                    tc = time.perf_counter()
                    ft.trackEvent(t, L, R, pubs[i], copy=False)
                    inside += time.perf_counter() - tc
                    if a.gap_ms:
                        time.sleep(a.gap_ms * 1e-3)
                ms = (inside if a.gap_ms else time.perf_counter() - t0) / a.steps * 1e3
                res[v].append(ms)
                ft.close()
            finally:
                for k, _ in kv:
                    os.environ.pop(k, None)
    for v in a.variants:
        print("%-40s median %.4f ms/step  (%s)" % (v, statistics.median(res[v]), " ".join("%.4f" % x for x in res[v])))


if __name__ == "__main__":
    main()
