"""trackImage (the reference's image node, N4) timed at the shipped frame-camera sizes: ms per stereo
frame through esvio_fe_track_image, host images in, results out.  Not part of bench.py (the headline
is the event path); kept as a check that no stage of the image path falls off a cliff with size."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from esvio_amd import frontend as FE
from esvio_amd.synth import ImageStream

for name, (W, H), max_cnt, min_dist in (("(warm-up of the process)", (346, 260), 150, 10),
                                         ("esvio (DAVIS346 frames)", (346, 260), 150, 10),
                                         ("esvio_VECtor", (1224, 1024), 200, 20),
                                         ("esvio_DSEC", (1440, 1080), 175, 40),
                                         ("esvio_ecmd", (1920, 1200), 200, 30)):
    s = ImageStream(W, H, velocity=(4, -2), disparity=12, seed=3)
    frames = [s.next_frame() for _ in range(24)]
    ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=max_cnt, min_dist=min_dist, flow_back=1))
    for L, R, t in frames[:4]:
        ft.trackImage(t, L, R, True)
    t0 = time.perf_counter()
    for k, (L, R, t) in enumerate(frames[4:]):
        ft.trackImage(t, L, R, k % 2 == 0)
    dt = (time.perf_counter() - t0) / (len(frames) - 4)
    print("%-26s %4dx%-4d  %.3f ms per stereo frame (%.0f frames/s), %d tracks" % (name, W, H, dt * 1e3, 1 / dt, len(ft.ids)))
    ft.close()
