"""LK kernel micro-benchmark: per-iteration latency of k_lk on noise images (forces maxCount).
    python tools/lk_microbench.py [lk_accum]      1 (default): exact sums; 2: the float order (k_lk_f32)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from esvio_amd import frontend as FE
from oracle import oracle as O

W, H = 640, 480
rng = np.random.default_rng(0)
ACC = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ft = FE.FeatureTracker(FE.make_config(W, H, lk_accum=ACC))
ft.set_profiling(True)
L_ = O.lib(); out = (C.c_ulonglong * 3)()
for name, (a, b) in {
    "noise": (rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W), dtype=np.uint8)),
    "same": (None, None),
}.items():
    if a is None:
        a = rng.integers(0, 256, (H, W), dtype=np.uint8); b = a
    for n in (1, 4, 64, 300):
        pts = np.stack([rng.uniform(40, W - 40, n), rng.uniform(40, H - 40, n)], 1).astype(np.float32)
        for ml in (3, 0):
            ft.reset_kernel_stats()
            for _ in range(5):
                g, st = ft.calcOpticalFlowPyrLK(a, b, pts, maxLevel=ml)
            s = next(v for k, v in ft.kernel_stats().items() if k in ("k_lk", "k_lk_f32") and v["launches"])
            L_.oracle_lk_iter_stats(out, 1)
            c, cs = O.lk(a, b, pts, max_level=ml, accum=ACC)
            L_.oracle_lk_iter_stats(out, 0)
            assert np.array_equal(g.view(np.uint32), c.view(np.uint32)) and np.array_equal(st, cs)
            print("%-6s n=%3d maxLevel=%d  k_lk avg %.2f us   oracle iters=%d visits=%d (%.1f/visit)" % (
                name, n, ml, s["ms"] / s["launches"] * 1e3, out[0], out[1], out[0] / max(out[1], 1)))
