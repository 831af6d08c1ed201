"""Per-iteration time of the LK kernels, timing only (no parity check: usable with experimental builds,
ESVIO_FE_LIB=tools/_bin/libesvio_fe_X.so).  Noise images force maxCount iterations on every level, so a launch
of n points at maxLevel 3 is 4 x 30 dependent iterations per wave.
    python tools/lk_iter_probe.py [lk_accum]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from esvio_amd import frontend as FE

W, H = 640, 480
rng = np.random.default_rng(0)
ACC = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ft = FE.FeatureTracker(FE.make_config(W, H, lk_accum=ACC))
ft.set_profiling(True)
a = rng.integers(0, 256, (H, W), dtype=np.uint8)
b = rng.integers(0, 256, (H, W), dtype=np.uint8)
pts = np.stack([rng.uniform(60, W - 60, 300), rng.uniform(60, H - 60, 300)], 1).astype(np.float32)
for ml in (3, 0):
    ft.reset_kernel_stats()
    for _ in range(8):
        ft.calcOpticalFlowPyrLK(a, b, pts, maxLevel=ml)
    s = next(v for k, v in ft.kernel_stats().items() if k in ("k_lk", "k_lk_f32") and v["launches"])
    us = s["ms"] / s["launches"] * 1e3
    print("lib %s accum %d maxLevel %d: k_lk %.1f us per launch" % (os.path.basename(FE.lib_path()), ACC, ml, us))
    if ml == 3:
        us3 = us
    else:
        print("   => %.3f us per iteration ((L3 - L0) / 90 iterations), %.1f us per level of set-up + tail" % ((us3 - us) / 90.0, us - 30 * (us3 - us) / 90.0))
