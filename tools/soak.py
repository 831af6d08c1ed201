"""soak.py — randomized end-to-end parity soak on the GPU: for `--minutes` minutes, random sensor sizes,
event rates, feature budgets, equalize / lk_accum / median settings, publish patterns and replay
schedules (0..5 batches announced ahead, lazy mode and helper threads switched between calls, finish()
now and then; round 3: the batches as pageable host arrays, in pinned host memory or in device memory,
motion compensation also on announced batches, the speculative / chained LK waits made to expire now
and then); every frame's public result vectors are compared bit for bit with the oracle's.
Stops at the first difference and prints the configuration that produced it.

    python tools/soak.py --minutes 10 [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from esvio_amd import frontend as FE
from esvio_amd.events import event_times
from esvio_amd.synth import ImageStream, PoissonStream, SceneStream
from oracle import oracle as O

FIELDS = ("cur_pts", "cur_un_pts", "pts_velocity", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity")


def same(ft, r, left_only=False):
    if not (np.array_equal(ft.ids, r.ids) and np.array_equal(ft.track_cnt, r.track_cnt)):
        return "ids / track_cnt"
    if not left_only and not np.array_equal(ft.ids_right, r.ids_right):
        return "ids_right"
    for k in FIELDS[:3] if left_only else FIELDS:
        a, b = getattr(ft, k), getattr(r, k)
        if a.shape != b.shape or not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            return k
    return None


def image_case(rng, case):
    """the image front-end: trackImage over a translating texture, a frame without a right image now and then"""
    W, H = [(346, 260), (640, 480), (1224, 1024), (1440, 1080)][int(rng.integers(0, 4))]
    kw = dict(max_cnt=int(rng.integers(30, 250)), min_dist=int(rng.integers(8, 45)), equalize=int(rng.random() < 0.3),
              lk_accum=2 - int(rng.random() < 0.3), flow_back=int(rng.random() < 0.85))
    seed = int(rng.integers(0, 1 << 30))
    s = ImageStream(W, H, velocity=(int(rng.integers(-6, 7)), int(rng.integers(-5, 6))), disparity=int(rng.integers(4, 20)),
                    seed=seed)
    n = int(rng.integers(3, 8))
    desc = dict(case=case, kind="image", W=W, H=H, seed=seed, frames=n, **kw)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = O.Tracker(O.make_config(W, H, **kw))
    try:
        for f in range(n):
            L, R, t = s.next_frame()
            if rng.random() < 0.15:
                R = None
            pub = bool(rng.random() < 0.7)
            ft.trackImage(t, L, R, pub)
            bad = same(ft, tr.track_image(t, L, R, pub))
            if bad:
                return desc, "frame %d: %s differs" % (f, bad)
    finally:
        ft.close()
    return desc, None


def one_case(rng, case):
    if rng.random() < 0.1:
        return image_case(rng, case)
    W, H = [(346, 260), (640, 480), (173, 131), (800, 600), (1280, 720)][int(rng.integers(0, 5))]
    rate = float(rng.choice([3e5, 1e6, 3e6, 8e6])) * (W * H / (640 * 480)) ** 0.5
    kw = dict(max_cnt=int(rng.integers(20, 400)), min_dist=int(rng.integers(3, 41)), f_ransac=1,
              equalize=int(rng.random() < 0.25), lk_accum=2 - int(rng.random() < 0.3), flow_back=int(rng.random() < 0.85),
              median_blur_kernel_size=int(rng.random() < 0.1))
    seed = int(rng.integers(0, 1 << 30))
    stream = (PoissonStream(W, H, rate=rate, seed=seed) if rng.random() < 0.15 else
              SceneStream(W, H, rate=rate, seed=seed, n_rect=int(rng.integers(4, 30))))
    n = int(rng.integers(6, 22))
    batches = [stream.next_batch()[:2] for _ in range(n)]
    p_pub = float(rng.choice([0.3, 0.5, 0.8, 1.0]))
    pubs = [bool(rng.random() < p_pub) for _ in batches]
    desc = dict(case=case, W=W, H=H, rate=rate, seed=seed, frames=n, p_pub=p_pub, **kw)
    one_stream = rng.random() < 0.2  # (plain calls: both cameras' updates on the main stream instead of on two)
    if one_stream:
        os.environ["ESVIO_FE_NO_CAMSPLIT"] = "1"
    try:
        ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    finally:
        os.environ.pop("ESVIO_FE_NO_CAMSPLIT", None)
    tr = O.Tracker(O.make_config(W, H, **kw))
    replay = rng.random() < 0.7
    mc = rng.random() < 0.3  # Do_motion_correction
    where = ["host", "host", "pinned", "device"][int(rng.integers(0, 4))]  # where the caller keeps the batches
    fault = int(rng.choice([0, 0, 0, 4, 8, 12] if replay else [0, 0, 8]))  # expiring speculative / chained waits
    desc.update(replay=bool(replay), mc=bool(mc), where=where, fault=fault, one_stream=bool(one_stream))
    bufs = []

    def arg(a):
        if where == "host":
            return a
        b = FE.EventBuffer(a, FE.HOST if where == "pinned" else FE.DEVICE)
        bufs.append(b)
        return b.array if where == "pinned" else b.arg
    args = [(arg(L), arg(R)) for L, R in batches]
    mvs = []
    for L, _ in batches:
        te = event_times(L)
        mvs.append(dict(t1=te[0] + float(rng.uniform(-0.2, 1.3)) * (te[-1] - te[0]), v=tuple(rng.uniform(-1, 1, 3)),
                        v_pre=tuple(rng.uniform(-1, 1, 3)), accel=tuple(rng.uniform(-6, 6, 3)),
                        omega=tuple(rng.uniform(-3, 3, 3) * (10 if rng.random() < 0.1 else 1)),
                        fx=0.9 * W, fy=0.9 * W, cx=W / 2.0 + 1.5, cy=H / 2.0 - 0.75))
    announced = 0
    try:
        for f, (L, R) in enumerate(batches):
            if replay:
                if rng.random() < 0.2:
                    ft.set_lazy_new_stereo(bool(rng.integers(0, 2)))
                if rng.random() < 0.15:
                    ft.set_host_threads(int(rng.integers(1, 6)))
                if rng.random() < 0.25:  # (round 4: the prefetch launches issued by the handle's launch thread)
                    ft.set_launch_thread(bool(rng.integers(0, 2)))
                if fault and rng.random() < 0.3:
                    ft.debug_inject(fault if rng.random() < 0.5 else 0)
                announced = max(announced, f)
                want = min(f + int(rng.integers(0, 6)), len(batches) - 1)
                while announced < want:
                    announced += 1
                    ft.set_next_batch(event_times(batches[announced][0])[-1], args[announced][0], args[announced][1],
                                      pubs[announced], measurements=FE.make_motion(**mvs[announced]) if mc else None)
            elif fault and rng.random() < 0.3:  # (a plain call's chained stereo LK giving up)
                ft.debug_inject(fault if rng.random() < 0.5 else 0)
            t = event_times(L)[-1]
            if mc:
                ft.trackEvent(t, args[f][0], args[f][1], pubs[f], measurements=FE.make_motion(**mvs[f]))
                r = tr.track_event(t, L, R, pubs[f], motion=O.make_motion(**mvs[f]))
            else:
                ft.trackEvent(t, args[f][0], args[f][1], pubs[f])
                r = tr.track_event(t, L, R, pubs[f])
            left_only = False
            if replay and (rng.random() < 0.5 or f == len(batches) - 1):
                ft.finish()
            elif replay:
                left_only = True  # (a lazy frame's right-camera entries are completed by the next call)
            bad = same(ft, r, left_only)
            if bad:
                return desc, "frame %d: %s differs" % (f, bad)
        if not np.array_equal(ft.gettimesurface(0), tr.time_surface(0)):
            return desc, "time surface differs"
    except Exception:
        print("EXCEPTION at frame %d of" % f, desc, flush=True)
        raise
    finally:
        ft.close()
        for b in bufs:
            b.free()
    return desc, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    cases = frames = 0

    def on_term(signum, frame):  # (a `timeout` around the run: say how far it got instead of dying silently)
        print("soak: killed by signal %d after %d cases, %d frames, %.1f min without a difference (a case was in progress)"
              % (signum, cases, frames, (time.time() - t0) / 60), flush=True)
        sys.exit(2)
    import signal
    signal.signal(signal.SIGTERM, on_term)
    while time.time() - t0 < a.minutes * 60:
        try:
            desc, err = one_case(rng, cases)
        except Exception as e:  # (a library error is a finding as well: say in which case)
            print("EXCEPTION in case %d (seed %d): %r" % (cases, a.seed, e), flush=True)
            raise
        cases += 1
        frames += desc["frames"]
        if err:
            print("MISMATCH", err, desc)
            sys.exit(1)
    print("soak: %d cases, %d frames, %.1f min, no difference" % (cases, frames, (time.time() - t0) / 60))


if __name__ == "__main__":
    main()
