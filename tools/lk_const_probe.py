"""Launch constant, per-level cost and per-iteration time of the LK kernels, timing only (noise images force maxCount
iterations on every level): launches of maxLevel L in {0, 3} and maxCount K in {10, 30} give
    T(L, K) = C + (L + 1) * S + (L + 1) * K * t        (forward call only)
    python tools/lk_const_probe.py [lk_accum]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from esvio_amd import frontend as FE

W, H = 640, 480
rng = np.random.default_rng(0)
ACC = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ft = FE.FeatureTracker(FE.make_config(W, H, lk_accum=ACC))
ft.set_profiling(True)
a = rng.integers(0, 256, (H, W), dtype=np.uint8)
b = rng.integers(0, 256, (H, W), dtype=np.uint8)
for n in (4, 300):
    pts = np.stack([rng.uniform(60, W - 60, n), rng.uniform(60, H - 60, n)], 1).astype(np.float32)
    T = {}
    for ml in (3, 0):
        for mc in (30, 10):
            ft.reset_kernel_stats()
            for _ in range(8):
                ft.calcOpticalFlowPyrLK(a, b, pts, maxLevel=ml, max_count=mc)
            s = next(v for k, v in ft.kernel_stats().items() if k in ("k_lk", "k_lk_f32") and v["launches"])
            T[(ml, mc)] = s["ms"] / s["launches"] * 1e3
    t0 = (T[(0, 30)] - T[(0, 10)]) / 20          # per iteration, level 0 (raw noise)
    t3 = (T[(3, 30)] - T[(3, 10)]) / 80          # per iteration, mean over the four levels
    cs0 = T[(0, 10)] - 10 * t0                   # C + S
    cs3 = T[(3, 10)] - 40 * t3                   # C + 4 S
    S = (cs3 - cs0) / 3
    print("%s acc %d n %3d: T(L3,30) %.1f T(L3,10) %.1f T(L0,30) %.1f T(L0,10) %.1f us -> iteration %.3f (level 0) / %.3f (all levels) us, "
          "per level %.2f us, launch constant %.1f us" % (os.path.basename(FE.lib_path()), ACC, n, T[(3, 30)], T[(3, 10)], T[(0, 30)],
                                                          T[(0, 10)], t0, t3, S, cs0 - S))
