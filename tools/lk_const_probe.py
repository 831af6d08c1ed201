import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from esvio_amd import frontend as FE
W, H = 640, 480
rng = np.random.default_rng(0)
ACC = int(sys.argv[1])
ft = FE.FeatureTracker(FE.make_config(W, H, lk_accum=ACC))
ft.set_profiling(True)
a = rng.integers(0, 256, (H, W), dtype=np.uint8)
b = rng.integers(0, 256, (H, W), dtype=np.uint8)
for n in (4, 300):
    pts = np.stack([rng.uniform(60, W - 60, n), rng.uniform(60, H - 60, n)], 1).astype(np.float32)
    r = {}
    for ml in (3, 0):
        ft.reset_kernel_stats()
        for _ in range(8):
            ft.calcOpticalFlowPyrLK(a, b, pts, maxLevel=ml)
        s = ft.kernel_stats()["k_lk"]
        r[ml] = s["ms"] / s["launches"] * 1e3
    print("%s acc %d n %3d: L3 %.1f L0 %.1f us -> %.3f us/iter, const %.1f us" % (os.path.basename(FE.lib_path()), ACC, n, r[3], r[0], (r[3]-r[0])/90, r[0]-(r[3]-r[0])/3))
