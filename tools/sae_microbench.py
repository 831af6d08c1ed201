#!/usr/bin/env python3
"""The event-proportional chain alone (createSAE_left/right of one stereo batch: k_tile_hist, k_tile_scan,
k_tile_scatter, k_tile_apply) on device-resident events, for rocprofv3:

    rocprofv3 --kernel-trace --stats -d out -o sae -- python tools/sae_microbench.py [--width 1280 --height 720
              --rate 1e8 --iters 20 --stream scene|poisson]

Prints the library's own per-label HIP-event averages as well (hist + scan share a label)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esvio_amd import frontend as FE  # noqa: E402
from esvio_amd.synth import PoissonStream, SceneStream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--rate", type=float, default=1e8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--nbatches", type=int, default=4)
    ap.add_argument("--stream", default="scene")
    a = ap.parse_args()
    W, H = a.width, a.height
    s = (PoissonStream(W, H, rate=a.rate, seed=12345) if a.stream == "poisson" else SceneStream(W, H, rate=a.rate, seed=12345))
    def to_dev(arr):
        return FE.EventBuffer(arr, FE.DEVICE).arg
    batches = []
    for _ in range(a.nbatches):
        L, R, _ = s.next_batch()
        batches.append((to_dev(L), to_dev(R), len(L) + len(R)))
    ft = FE.FeatureTracker(FE.make_config(W, H))
    for b in batches:  # warm-up: buffers grow
        ft.detector.createSAE_stereo(b[0], b[1])
    ft.set_profiling(True)
    ft.reset_kernel_stats()
    t0 = time.perf_counter()
    ev = 0
    for i in range(a.iters):
        b = batches[i % len(batches)]
        ft.detector.createSAE_stereo(b[0], b[1])
        ev += b[2]
    dt = time.perf_counter() - t0
    st = ft.kernel_stats()
    print("%dx%d %s, %.2f M events per batch: %.1f us per batch wall (incl. the call's sync)"
          % (W, H, a.stream, ev / a.iters / 1e6, dt / a.iters * 1e6))
    tot = 0.0
    for k, what in (("k_tile_hist", "hist"), ("k_tile_scan", "scan"), ("k_tile_scatter", "scatter"), ("k_tile_apply", "apply")):
        v = st[k]
        if v["launches"]:
            us = v["ms"] / v["launches"] * 1e3
            tot += us
            print("  %-14s %8.2f us  (%.2f TB/s of its algorithmic bytes)" % (what, us, v["alg_bytes"] / v["launches"] / us / 1e6))
    print("  chain          %8.2f us = %.2f TB/s of 48 B/event = %.1f %% of 8 TB/s"
          % (tot, 48.0 * ev / a.iters / tot / 1e6, 48.0 * ev / a.iters / tot / 1e6 / 8 * 100))
    # SAEtoTimeSurface of one camera on the planes the batches left (esvio_fe_sae_to_time_surface: the render + the
    # image's way back to the host; the kernel's own time is the library's HIP-event pair)
    ft.reset_kernel_stats()
    t_sync = float(max(p.max() for p in ft.detector.get_sae(0)[2:]))  # (the newest event's time)
    for i in range(a.iters):
        ft.detector._ts(i & 1, t_sync)
    st = ft.kernel_stats()
    for k in ("k_time_surface4", "k_time_surface"):
        v = st.get(k)
        if v and v["launches"]:
            us = v["ms"] / v["launches"] * 1e3
            print("  %-14s %8.2f us  (%.2f TB/s of 17 B/pixel = %.1f %% of 8 TB/s), %d launches"
                  % (k, us, 17.0 * W * H / us / 1e6, 17.0 * W * H / us / 1e6 / 8 * 100, v["launches"]))
    ft.close()


if __name__ == "__main__":
    main()
