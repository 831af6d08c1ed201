"""LK sub-pixel positions: exact-sum accumulation (the GPU path, oracle accum=1) against the float
accumulation orders cv::calcOpticalFlowPyrLK can have in the reference's build
(feature_tracker.cpp:410,417-418,490,495 -> OpenCV video/lkpyramid.cpp, `typedef float acctype`):

  accum=2  x86 SIMD128 build (Noetic's distro OpenCV 4.2): four float lanes, lane k takes pixels
           x = 4m + k of the first 16 columns of every window row, columns 16..20 go to a scalar
           float, horizontal sum (q0+q2)+(q1+q3) at the end; the b vector pairs (x, x+4) in int32
           first (pmaddwd).  Restated from the published source as recalled — unpinned like
           everything OpenCV-internal.
  accum=0  the plain scalar loop (a build without SIMD).

The exact sums are the values both float orders approximate, so the GPU is closer to either of
them than they are to each other.  `python tests/lk_orders.py` prints the measured table.
"""
import numpy as np

# measured (see the table this file prints), asserted with ~1.5-2x margin.  On time surfaces the
# tail is heavier than on a smooth texture: a point whose last step lands next to the eps^2 = 1e-4
# px^2 stopping rule does one iteration more or fewer under another summation order and ends up to
# ~1e-2 px away (1-3 points of 300 per call); the bulk is unaffected.
BANDS = {
    # (kind, accum): (p90, p99, max, min fraction within 1e-4 px, max status flips)
    ("texture", 2): (5e-5, 2.5e-4, 5e-4, 0.95, 0),
    ("texture", 0): (1.5e-4, 4.5e-4, 6e-4, 0.85, 1),
    ("scene", 2): (5e-5, 6e-4, 2e-2, 0.97, 1),
    ("scene", 0): (2e-4, 6e-3, 2e-2, 0.85, 2),
}


def distribution(a_pts, a_st, b_pts, b_st):
    both = (a_st == 1) & (b_st == 1)
    d = np.abs(a_pts[both] - b_pts[both]).max(axis=1) if both.any() else np.zeros(1)
    p50, p90, p99, mx = np.percentile(d, [50, 90, 99, 100])
    return dict(n=int(both.sum()), flips=int((a_st != b_st).sum()), p50=float(p50), p90=float(p90),
                p99=float(p99), max=float(mx), within_1e4=float((d <= 1e-4).mean()))


def assert_band(dist, accum, kind="texture"):
    p90, p99, mx, frac, flips = BANDS[(kind, accum)]
    assert dist["flips"] <= flips, dist
    assert dist["p90"] <= p90 and dist["p99"] <= p99 and dist["max"] <= mx, dist
    assert dist["within_1e4"] >= frac, dist


def texture(W, H, seed):
    rng = np.random.default_rng(seed)
    base = rng.random((H // 8 + 3, W // 8 + 3))
    img = np.kron(base, np.ones((8, 8)))[:H + 16, :W + 16]
    k = np.ones(5) / 5
    for ax in (0, 1):
        img = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, img)
    return img


def cases():
    """(name, prev, next, pts, init, max_level, flags): the three call shapes of trackEvent on a
    smooth texture, and consecutive time surfaces of the 640x480 scene stream at tracked corners"""
    W, H = 640, 480
    tex = texture(W, H, 2)
    prev = (tex[8:8 + H, 8:8 + W] * 255).astype(np.uint8)
    nxt = (tex[6:6 + H, 11:11 + W] * 255).astype(np.uint8)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-5, W + 5, 300), rng.uniform(-5, H + 5, 300)], 1).astype(np.float32)
    for (ml, flags) in ((3, 0), (1, 4), (0, 0)):
        init = pts + rng.uniform(-2, 2, pts.shape).astype(np.float32)
        yield ("texture maxLevel %d flags %d" % (ml, flags), prev, nxt, pts, init, ml, flags)


def scene_cases(oracle, n_frames=6):
    """time surfaces of the bench's scene stream and the corners the tracker holds on them"""
    from esvio_amd.events import event_times
    from esvio_amd.synth import SceneStream
    W, H = 640, 480
    s = SceneStream(W, H, rate=5e6, seed=1)
    tr = oracle.Tracker(oracle.make_config(W, H, lk_accum=1, max_cnt=300, min_dist=10, f_ransac=1))
    prev_img = None
    for f in range(n_frames):
        L, R, _ = s.next_batch()
        pts = None if prev_img is None else np.array(r.cur_pts, np.float32)
        r = tr.track_event(event_times(L)[-1], L, R, True)
        img = tr.time_surface(0)
        if prev_img is not None and len(pts):
            yield ("scene frame %d temporal fwd" % f, prev_img, img, pts, pts.copy(), 3, 0)
            yield ("scene frame %d stereo" % f, img, tr.time_surface(1), np.array(r.cur_pts, np.float32),
                   np.array(r.cur_pts, np.float32), 3, 0)
        prev_img = img


def table(oracle, with_scene=True):
    rows = []
    cs = list(cases()) + (list(scene_cases(oracle)) if with_scene else [])
    for name, prev, nxt, pts, init, ml, flags in cs:
        e_pts, e_st = oracle.lk(prev, nxt, pts, init, max_level=ml, flags=flags, accum=1)
        f = {a: oracle.lk(prev, nxt, pts, init, max_level=ml, flags=flags, accum=a) for a in (2, 0)}
        rows.append((name, distribution(e_pts, e_st, *f[2]), distribution(e_pts, e_st, *f[0]),
                     distribution(f[2][0], f[2][1], *f[0])))
    return rows


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    fmt = "%-34s %-22s n=%3d flips=%d p50=%.1e p90=%.1e p99=%.1e max=%.1e within1e-4=%.3f"
    for name, d2, d0, d20 in table(O):
        for tag, d in (("exact vs SIMD128 order", d2), ("exact vs scalar order", d0),
                       ("SIMD128 vs scalar order", d20)):
            print(fmt % (name, tag, d["n"], d["flips"], d["p50"], d["p90"], d["p99"], d["max"], d["within_1e4"]))
