"""A/B of the two partition forms (ESVIO_FE_PART2=1: two levels, k_part_*) on the same batches: SAE planes after every batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from esvio_amd import frontend as FE
from esvio_amd.synth import SceneStream, PoissonStream

W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 640, int(sys.argv[2]) if len(sys.argv) > 2 else 480
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 5e6
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 12
s = SceneStream(W, H, rate=rate, seed=1)
os.environ["ESVIO_FE_PART2"] = "1"
new = FE.FeatureTracker(FE.make_config(W, H))
os.environ.pop("ESVIO_FE_PART2", None)
old = FE.FeatureTracker(FE.make_config(W, H))
bad = 0
for f in range(nb):
    L, R, _ = s.next_batch()
    for cam, ev in ((0, L), (1, R)):
        (new.detector.createSAE_right if cam else new.detector.createSAE_left)(ev)
        (old.detector.createSAE_right if cam else old.detector.createSAE_left)(ev)
        pn, po = new.detector.get_sae(cam), old.detector.get_sae(cam)
        for name, a, b in zip(("L0", "L1", "S0", "S1"), pn, po):
            d = np.argwhere(a != b)
            if len(d):
                bad += 1
                print("frame %d cam %d plane %s: %d pixels differ, first %s" % (f, cam, name, len(d), d[:5].tolist()))
                y, x = d[0]
                idx = np.nonzero((ev["x"] == x) & (ev["y"] == y))[0]
                print("   events at that pixel: idx", idx[:10], "of n", len(ev), " blocks", (idx // 2048)[:10], "pol", ev["polarity"][idx][:10])
                print("   new", a[y, x], "old", b[y, x])
print("done, mismatching planes:", bad)
