"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bars: SAE planes, time-surface bytes, corner flags and selected corner indices bit-exact;
pyramid / Scharr bit-exact; LK positions bit-exact against the oracle's exact-sum mode (both
sides accumulate the normal equations exactly).  Against float accumulation — what the reference's
x86 OpenCV build does — the difference is a measured distribution, asserted as measured
(tests/lk_orders.py, also run on CPU by tests/test_lk_float_orders.py): vs the SIMD128 lane order
of OpenCV 4.2 p90 <= 5e-5 px, p99 <= 2.5e-4, max <= 5e-4, >= 95 % of the points within 1e-4 px, no
status flips; vs the scalar loop order p90 <= 1.5e-4, max <= 6e-4, >= 85 % within 1e-4.
"""
import numpy as np
import pytest

from esvio_amd import frontend as FE
from esvio_amd.events import EVENT_DTYPE, event_times, make_events
from esvio_amd.synth import SceneStream, uniform_batch

import lk_orders

pytestmark = pytest.mark.gpu


def _mk(W, H, **kw):
    return FE.FeatureTracker(FE.make_config(W, H, **kw))


def _planes_equal(a, b):
    for x, y, name in zip(a, b, ("L0", "L1", "S0", "S1")):
        assert np.array_equal(x, y), "plane %s differs at %d pixels" % (name, (x != y).sum())


@pytest.mark.parametrize("W,H,n", [(346, 260, 33000), (640, 480, 167000)])
def test_sae_and_time_surface_uniform(oracle, W, H, n):
    rng = np.random.default_rng(12345)
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    t0 = 1_700_000_000_000_000
    for b in range(3):
        L = uniform_batch(W, H, n, t0 + b * 33333, 33333, rng)
        R = uniform_batch(W, H, n, t0 + b * 33333, 33333, rng)
        assert ft.detector.createSAE_stereo(L, R) == 0
        det.create_sae(0, L)
        det.create_sae(1, R)
        for cam in (0, 1):
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
        t_sync = event_times(L)[-1]
        for cam, f in ((0, ft.detector.SAEtoTimeSurface_left), (1, ft.detector.SAEtoTimeSurface_right)):
            assert np.array_equal(f(t_sync), det.time_surface(cam, t_sync))
    ft.close()


SAE_PATHS = {  # the tiled update (default) and the two forms of the radix-sort one
    "tiled": {}, "sort_walk": {"ESVIO_FE_SAE_SORT": "1"},
    "sort_per_event": {"ESVIO_FE_SAE_SORT": "1", "ESVIO_FE_SAE_EV_MIN": "0"},
}


@pytest.mark.parametrize("path", list(SAE_PATHS))
def test_sae_adversarial_duplicates(oracle, path, monkeypatch):
    """many events on few pixels (segments of ~8000), equal timestamps, alternating polarity, border
    pixels, out-of-sensor events (skipped + counted); through the tiled update (k_tile_*), the
    sort + per-pixel walk and the sort + per-event kernels (k_sae_apply_ev)"""
    for k, v in SAE_PATHS[path].items():
        monkeypatch.setenv(k, v)
    W, H = 346, 260
    rng = np.random.default_rng(7)
    n = 50000
    hot = np.array([[0, 0], [W - 1, H - 1], [5, 5], [100, 100], [101, 100], [W - 1, 0]])
    pick = rng.integers(0, len(hot), n)
    x, y = hot[pick, 0].copy(), hot[pick, 1].copy()
    t = np.sort(rng.integers(0, 2000, n)) * 5 + 5_000_000  # many exact ties
    p = rng.integers(0, 2, n)
    p[::7] = 1 - p[::7]
    x[100] = W  # out of sensor
    y[200] = H
    ev = make_events(x, y, t, p)
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    assert ft.detector.createSAE_left(ev) == 2
    assert det.create_sae(0, ev) == 2
    _planes_equal(ft.detector.get_sae(0), det.get_sae(0))
    # second batch continues from carried state, some events older than the state (time reversal)
    ev2 = make_events(x, y, t - 1000, 1 - p)
    ft.detector.createSAE_left(ev2)
    det.create_sae(0, ev2)
    _planes_equal(ft.detector.get_sae(0), det.get_sae(0))
    ft.close()


@pytest.mark.parametrize("path", list(SAE_PATHS))
def test_sae_segment_lengths_around_the_wave_path(oracle, path, monkeypatch):
    """per-pixel segments of every length around the grouped fetch (8 positions per step) and the
    wave width (63..65, 127..129), up to 1600 events on one pixel: long same-polarity bursts inside
    the refractory window, polarity flips, equal stamps and stamps going BACKWARDS inside the batch;
    two batches so the carried-in state matters.  With ESVIO_FE_SAE_EV_MIN=0 through the per-event
    kernels: runs that start before / end after a wave, backward and forward scans of every depth."""
    for k, v in SAE_PATHS[path].items():
        monkeypatch.setenv(k, v)
    W, H = 346, 260
    rng = np.random.default_rng(11)
    lengths = [1, 2, 15, 16, 17, 18, 31, 63, 64, 65, 66, 127, 128, 129, 200, 511, 512, 513, 514, 575,
               576, 577, 1025, 1600]
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    for batch in range(2):
        xs, ys, ts, ps = [], [], [], []
        for i, n in enumerate(lengths * 3):
            x, y = 3 + 2 * (i % 100), 7 + 5 * (i // 100) + 40 * batch * (i % 2)
            kind = i % 3
            if kind == 0:      # bursts of one polarity, 1 ms apart (mostly filtered), rare flips
                p = (np.cumsum(rng.random(n) < 0.05) + i) % 2
                t = 6_000_000 + np.arange(n) * 1000
            elif kind == 1:    # random polarity, many ties, some 20 ms gaps (pass by time)
                p = rng.integers(0, 2, n)
                t = 6_000_000 + np.cumsum(rng.choice([0, 0, 500, 20_000], n))
            else:              # stamps jump backwards now and then
                p = rng.integers(0, 2, n)
                t = 6_500_000 + np.cumsum(rng.choice([-3000, 0, 2000, 15_000], n))
            xs.append(np.full(n, x)); ys.append(np.full(n, y)); ts.append(t + batch * 40_000); ps.append(p)
        x, y, t, p = (np.concatenate(v) for v in (xs, ys, ts, ps))
        perm = rng.permutation(len(x))  # interleave the pixels; each pixel's own order ...
        order = perm[np.argsort(np.concatenate([np.arange(len(v)) for v in xs])[perm], kind="stable")]
        ev = make_events(x[order], y[order], t[order], p[order])  # ... stays as generated
        ft.detector.createSAE_left(ev)
        det.create_sae(0, ev)
        _planes_equal(ft.detector.get_sae(0), det.get_sae(0))
        ft.detector.createSAE_right(ev[::-1].copy() if batch else ev)  # right camera: reversed stream
        det.create_sae(1, ev[::-1].copy() if batch else ev)
        _planes_equal(ft.detector.get_sae(1), det.get_sae(1))
    ft.close()


@pytest.mark.parametrize("nL,nR", [(0, 5000), (5000, 0), (1, 1), (63, 65), (2047, 2049), (2048, 2048), (4096, 1),
                                   (2049, 6145), (40000, 3), (0, 1)])
def test_sae_partition_block_boundaries(oracle, nL, nR):
    """the partition cuts each camera's array into scatter blocks of 2048 events of its own (round 6): array lengths
    on, just below and just above the block size, one array empty, one event — and a few hundred events outside the
    sensor in either camera — planes and the rejected count as the oracle's, over three batches"""
    W, H = 346, 260
    rng = np.random.default_rng(1000 * nL + nR)
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    t0 = 1_700_000_000_000_000
    for b in range(3):
        L = uniform_batch(W, H, nL, t0 + b * 33333, 33333, rng)
        R = uniform_batch(W, H, nR, t0 + b * 33333, 33333, rng)
        for a in (L, R):  # every 17th event outside the sensor
            if len(a):
                a["x"][::17] = W + (np.arange(len(a["x"][::17])) % 5)
        rej = ft.detector.createSAE_stereo(L, R)
        want = det.create_sae(0, L) + det.create_sae(1, R)
        assert rej == want == len(L["x"][::17]) + len(R["x"][::17])
        for cam in (0, 1):
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
    ft.close()


@pytest.mark.parametrize("case", ["span_below", "span_at", "span_above", "big_nsec", "forced_wide", "reversed"])
def test_sae_partition_record_formats(oracle, case, monkeypatch):
    """the tiled update partitions 8-byte records (tile-local pixel, polarity, seconds relative to the
    batch's smallest, nsec) when the batch's seconds span less than 2^20 and every nsec fits 30 bits,
    else the raw 16-byte records: planes bit-exact on both sides of the rule, with the smallest second
    not at the start of the stream, with nsec words no ros::Time would hold, and with the wide form
    forced (ESVIO_FE_WIDE_RECORDS=1)"""
    if case == "forced_wide":
        monkeypatch.setenv("ESVIO_FE_WIDE_RECORDS", "1")
    W, H = 346, 260
    rng = np.random.default_rng(77)
    n = 60000
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    for b in range(2):
        ev = np.zeros(n, EVENT_DTYPE)
        ev["x"] = rng.integers(0, W, n)
        ev["y"] = rng.integers(0, H, n)
        ev["x"][:4000] = rng.integers(0, 8, 4000)          # hot pixels: many events per (pixel, polarity)
        ev["y"][:4000] = rng.integers(0, 8, 4000)
        ev["polarity"] = rng.integers(0, 2, n)
        base = 5000 + 7 * b
        span = {"span_below": (1 << 20) - 1, "span_at": 1 << 20, "span_above": (1 << 21) + 5}.get(case, 3)
        sec = base + np.sort(rng.integers(0, span + 1, n))
        sec[0], sec[-1] = base, base + span                 # the exact span
        if case == "reversed":
            sec = sec[::-1].copy()                           # the smallest second comes last
        ev["sec"] = sec
        ev["nsec"] = rng.integers(0, 1_000_000_000, n)
        if case == "big_nsec":
            ev["nsec"][::97] = 0xC0000000 + rng.integers(0, 1000, len(ev["nsec"][::97]))
        perm = rng.permutation(n)[:n // 2]
        L, R = ev, ev[np.sort(perm)].copy()
        assert ft.detector.createSAE_stereo(L, R) == 0
        det.create_sae(0, L)
        det.create_sae(1, R)
        for cam in (0, 1):
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
    ft.close()


def test_time_surface_edge_cases(oracle):
    W, H = 346, 260
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    rng = np.random.default_rng(3)
    S0 = np.where(rng.random((H, W)) < 0.5, rng.uniform(10.0, 10.2, (H, W)), 0.0)
    S1 = np.where(rng.random((H, W)) < 0.5, rng.uniform(10.0, 10.2, (H, W)), 0.0)
    S1[0, :50] = S0[0, :50]  # ties: polarity -1
    Z = np.zeros((H, W))
    ft.detector.set_sae(1, Z, Z, S0, S1)
    det.set_sae(1, Z, Z, S0, S1)
    for t_sync in (10.2, 10.1, 10.0, 9.5, 8.0, 11.0, 10.2 + 1e-9):
        # t_sync earlier than stamps: exp(+x) saturates; far earlier: cvRound overflow -> 0
        assert np.array_equal(ft.detector.SAEtoTimeSurface_right(t_sync), det.time_surface(1, t_sync)), t_sync
    ft.close()
    ft = FE.FeatureTracker(FE.make_config(W, H, ignore_polarity=1))
    det = oracle.Detector(W, H, ignore_polarity=1)
    ft.detector.set_sae(0, Z, Z, S0, S1)
    det.set_sae(0, Z, Z, S0, S1)
    for t_sync in (10.2, 10.0, 9.9):
        assert np.array_equal(ft.detector.SAEtoTimeSurface_left(t_sync), det.time_surface(0, t_sync))
    ft.close()


@pytest.mark.parametrize("W,H,rate", [(346, 260, 1e6), (640, 480, 5e6)])
def test_corner_flags_and_selection_scene(oracle, W, H, rate):
    s = SceneStream(W, H, rate=rate, seed=5, n_rect=12 if W < 400 else 28)
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    total_corners = 0
    for b in range(3):
        L, R, _ = s.next_batch()
        ft.detector.createSAE_stereo(L, R)
        det.create_sae(0, L)
        det.create_sae(1, R)
        t_sync = event_times(L)[-1]
        ts_gpu = ft.detector.SAEtoTimeSurface_left(t_sync)
        ts_cpu = det.time_surface(0, t_sync)
        assert np.array_equal(ts_gpu, ts_cpu)
        fg = ft.detector.isCorner(L)
        fc = det.corner_flags(L)
        assert np.array_equal(fg, fc), "%d flag mismatches" % (fg != fc).sum()
        total_corners += int(fc.sum())
        # greedy selection with a random pre-blocked mask
        mask = np.zeros((H, W), np.uint8)
        rng = np.random.default_rng(b)
        for _ in range(20):
            oracle.circle_fill(mask, int(rng.integers(0, W)), int(rng.integers(0, H)), 10)
        for maxc in (1, 37, 300):
            xy_g, idx_g = ft.Event_FeaturesToTrack(L, maxc, mask)
            xy_c, idx_c = det.features_to_track(L, maxc, 10, mask, ts_cpu)
            assert np.array_equal(idx_g, idx_c)
            assert np.array_equal(xy_g, xy_c)
    assert total_corners > 100  # the scene stream must actually exercise Arc*
    ft.close()


@pytest.mark.parametrize("W,H", [(640, 480), (346, 260), (173, 131)])
def test_pyramid_and_scharr(oracle, W, H):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    ft = _mk(640, 480)
    levels = ft.build_pyramid(img, 3)
    assert len(levels) == oracle.pyr_levels(W, H) + 1
    cur = img
    for l, (im, dv) in enumerate(levels):
        if l > 0:
            cur = oracle.pyr_down(cur)
        assert np.array_equal(im, cur), "level %d image" % l
        assert np.array_equal(dv, oracle.scharr(cur)), "level %d scharr" % l
    ft.close()


def _texture(W, H, seed):
    rng = np.random.default_rng(seed)
    base = rng.random((H // 8 + 3, W // 8 + 3))
    img = np.kron(base, np.ones((8, 8)))[:H + 16, :W + 16]
    k = np.ones(5) / 5
    for ax in (0, 1):
        img = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, img)
    return img


@pytest.mark.parametrize("lk_accum", [2, 1])
def test_lk_parity(oracle, lk_accum):
    """calcOpticalFlowPyrLK alone, both modes of the sums: lk_accum 2 (k_lk_f32: float, in the order of the
    reference's x86 OpenCV build) and lk_accum 1 (k_lk: exact integers) — each bit-identical to the oracle's
    same mode; the exact mode also inside its measured band around the float orders"""
    W, H = 640, 480
    tex = _texture(W, H, 2)
    prev = (tex[8:8 + H, 8:8 + W] * 255).astype(np.uint8)
    nxt = (tex[6:6 + H, 11:11 + W] * 255).astype(np.uint8)  # shift (-3, +2)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-5, W + 5, 300), rng.uniform(-5, H + 5, 300)], 1).astype(np.float32)
    ft = _mk(W, H, lk_accum=lk_accum)
    for (ml, flags) in ((3, 0), (1, FE.LK_USE_INITIAL_FLOW), (0, 0)):
        init = pts + rng.uniform(-2, 2, pts.shape).astype(np.float32)
        g_pts, g_st = ft.calcOpticalFlowPyrLK(prev, nxt, pts, init, maxLevel=ml, flags=flags)
        c_pts, c_st = oracle.lk(prev, nxt, pts, init, max_level=ml, flags=flags, accum=lk_accum)
        assert np.array_equal(g_st, c_st)
        assert np.array_equal(g_pts.view(np.uint32), c_pts.view(np.uint32)), \
            "max |d| = %g" % np.abs(g_pts - c_pts).max()
        if lk_accum == 1:
            # against float accumulation (what the reference's OpenCV does; order- and SIMD-width-dependent)
            # the exact sums sit in a measured band, not an identity — see tests/lk_orders.py for the numbers
            for accum in (2, 0):
                f_pts, f_st = oracle.lk(prev, nxt, pts, init, max_level=ml, flags=flags, accum=accum)
                dist = lk_orders.distribution(g_pts, g_st, f_pts, f_st)
                print("LK GPU vs float order %d (maxLevel %d flags %d): %s" % (accum, ml, flags, dist))
                lk_orders.assert_band(dist, accum)
    ok = g_st == 1
    assert ok.sum() > 150
    ft.close()


@pytest.mark.parametrize("W,H", [(346, 260), (1280, 720), (352, 264)])
def test_render_then_pyramid(oracle, W, H):
    """the plain configuration's time surface + pyramid (k_time_surface4 — k_time_surface, one pixel per thread, where
    the rows are no multiple of 4 — then k_pyr3) at sizes other than the bench's: images and tracks as the oracle's"""
    s = SceneStream(W, H, rate=2e6, seed=21)
    kw = dict(max_cnt=150, min_dist=15)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(3):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, True)
        r = tr.track_event(t, L, R, True)
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
        assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1))
        assert np.array_equal(ft.ids, r.ids) and np.array_equal(ft.cur_pts, r.cur_pts)
        assert np.array_equal(ft.ids_right, r.ids_right) and np.array_equal(ft.cur_right_pts, r.cur_right_pts)
    ft.close()


@pytest.mark.parametrize("lk_accum", [2, 1])
def test_track_event_end_to_end(oracle, lk_accum):
    """12 frames of the stereo scene stream through trackEvent; every public result vector of
    FeatureTracker must equal the oracle's (ids, track_cnt exact; float vectors bit-exact: the LK sums
    are accumulated the same way on both sides, in either mode)."""
    W, H = 640, 480
    s = SceneStream(W, H, rate=5e6, seed=1)
    kw = dict(f_ransac=1, lk_accum=lk_accum)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(12):
        L, R, _ = s.next_batch()
        pub = (f % 3) != 2  # mix published and non-published frames
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pub)
        r = tr.track_event(t, L, R, pub)
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
        assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1))
        assert np.array_equal(ft.ids, r.ids), f
        assert np.array_equal(ft.track_cnt, r.track_cnt)
        assert np.array_equal(ft.ids_right, r.ids_right)
        for k in ("cur_pts", "cur_un_pts", "pts_velocity", "cur_right_pts", "cur_un_right_pts",
                  "right_pts_velocity"):
            a, b = getattr(ft, k), getattr(r, k)
            assert a.shape == b.shape, (f, k)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (f, k, np.abs(a - b).max())
    assert len(ft.ids) > 100 and len(ft.ids_right) > 50 and ft.track_cnt.max() >= 4
    ft.close()


@pytest.mark.parametrize("flow_back,f_ransac", [(0, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("replay", [False, True])
def test_track_event_without_flow_back_or_ransac(oracle, flow_back, f_ransac, replay):
    """`flow_back: 0` (no backward LK launches, no 0.5 px round-trip filter: feature_tracker.cpp:412-431,:492-510) and
    `F-RANSAC off` (rejectWithF_event skipped, :442-447) — every shipped config sets both to 1, so only these cases draw
    the other branches: 10 frames at 640x480 as plain calls and as a replay schedule (batches announced ahead, lazy
    right-camera tails, launch thread), every public vector against the oracle's"""
    W, H = 640, 480
    s = SceneStream(W, H, rate=4e6, seed=31 + 2 * flow_back + f_ransac)
    kw = dict(flow_back=flow_back, f_ransac=f_ransac, max_cnt=200)
    batches = [s.next_batch()[:2] for _ in range(10)]
    pubs = [(f % 3) != 1 for f in range(len(batches))]
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    if replay:
        ft.set_lazy_new_stereo(True)
        ft.set_launch_thread(True)
    announced = 0
    for f, (L, R) in enumerate(batches):
        t = event_times(L)[-1]
        if replay:
            while announced < min(f + 2, len(batches) - 1):
                announced += 1
                La, Ra = batches[announced]
                ft.set_next_batch(event_times(La)[-1], La, Ra, pubs[announced])
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        if replay:  # (lazy mode: the right-camera vectors of a frame are complete after the next call / finish)
            assert np.array_equal(ft.ids, r.ids) and np.array_equal(ft.track_cnt, r.track_cnt), f
            assert np.array_equal(ft.cur_pts.view(np.uint32), r.cur_pts.view(np.uint32)), f
        else:
            _compare_tracks(ft, r, ("plain", flow_back, f_ransac, f))
    if replay:
        ft.finish()
        _compare_tracks(ft, r, ("replay", flow_back, f_ransac, "end"))
    assert len(ft.ids) > 60 and len(ft.ids_right) > 20 and ft.track_cnt.max() >= 3
    ft.close()


def test_track_event_at_epoch_timestamps(oracle):
    """ros::Time stamps of a live system: seconds since 1970 (1.7e9), where a double resolves 2.4e-7 s.
    Everything that subtracts times (the SAE rule's 10 ms filter, the decay, ptsVelocity's dt) does it
    in double on both sides: six frames of trackEvent equal the oracle's, velocities included."""
    W, H = 640, 480
    s = SceneStream(W, H, rate=4e6, seed=9, t0_us=1_700_000_000_000_000)
    kw = dict(f_ransac=1, max_cnt=200)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(6):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        assert t > 1.7e9
        ft.trackEvent(t, L, R, f % 2 == 0)
        _compare_tracks(ft, tr.track_event(t, L, R, f % 2 == 0), ("epoch", f))
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
    assert len(ft.ids) > 60 and np.abs(ft.pts_velocity).max() > 0
    ft.close()


@pytest.mark.parametrize("W,H", [(640, 480), (346, 260)])
def test_equalize_clahe_normalize(oracle, W, H):
    """equalize: 1 (config/esio_DSEC/esio.yaml:90): CLAHE(40, 8x8) + normalize(0,255,MINMAX) of the
    time surface is what LK sees (feature_tracker.cpp:375-382); 346x260 exercises the
    BORDER_REFLECT_101 extension for sizes not divisible by the 8x8 grid."""
    s = SceneStream(W, H, rate=2e6 if W > 400 else 6e5, seed=4, n_rect=14)
    ft = FE.FeatureTracker(FE.make_config(W, H, equalize=1))
    det = oracle.Detector(W, H)
    for b in range(2):
        L, R, _ = s.next_batch()
        ft.detector.createSAE_stereo(L, R)
        det.create_sae(0, L)
        det.create_sae(1, R)
        t = event_times(L)[-1]
        for cam, f in ((0, ft.detector.SAEtoTimeSurface_left), (1, ft.detector.SAEtoTimeSurface_right)):
            raw = f(t)
            assert np.array_equal(raw, det.time_surface(cam, t))      # the function returns the RAW surface
            eq = ft.export_image(cam)                                  # what LK will see
            assert np.array_equal(eq, oracle.normalize_minmax(oracle.clahe(raw))), (b, cam)
    ft.close()


def test_track_event_end_to_end_equalize(oracle):
    W, H = 640, 480
    s = SceneStream(W, H, rate=5e6, seed=2)
    kw = dict(f_ransac=1, equalize=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(8):
        L, R, _ = s.next_batch()
        pub = (f % 3) != 2
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pub)
        r = tr.track_event(t, L, R, pub)
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))  # raw surfaces
        assert np.array_equal(ft.ids, r.ids), f
        assert np.array_equal(ft.track_cnt, r.track_cnt) and np.array_equal(ft.ids_right, r.ids_right)
        for k in ("cur_pts", "cur_un_pts", "pts_velocity", "cur_right_pts", "cur_un_right_pts",
                  "right_pts_velocity"):
            a, b = getattr(ft, k), getattr(r, k)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (f, k)
    assert len(ft.ids) > 100 and len(ft.ids_right) > 50
    ft.close()


def _compare_tracks(ft, r, tag, id_offset=0):
    assert np.array_equal(ft.ids, r.ids + id_offset), tag
    assert np.array_equal(ft.track_cnt, r.track_cnt) and np.array_equal(ft.ids_right, r.ids_right + id_offset), tag
    for k in ("cur_pts", "cur_un_pts", "pts_velocity", "cur_right_pts", "cur_un_right_pts",
              "right_pts_velocity"):
        a, b = getattr(ft, k), getattr(r, k)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (tag, k)


def test_empty_right_batches_reset_and_capacity_growth(oracle):
    """right batch may be empty (node:150 only guards the left one); buffers grow when a later batch
    is larger; esvio_fe_reset behaves like a fresh tracker except that ids keep
    counting (n_id is a static in the reference, feature_tracker.cpp:9)."""
    W, H = 346, 260
    kw = dict(max_cnt=80, min_dist=10, f_ransac=1)
    small = SceneStream(W, H, rate=2e5, seed=21, n_rect=8, size=(30.0, 80.0))
    big = SceneStream(W, H, rate=3e6, seed=22, n_rect=12, size=(30.0, 80.0), t0_us=2_000_000_000)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    empty = np.zeros(0, EVENT_DTYPE)
    for f in range(4):  # tiny batches, right camera silent on odd frames
        L, R, _ = small.next_batch()
        if f % 2:
            R = empty
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, True)
        _compare_tracks(ft, tr.track_event(t, L, R, True), ("small", f))
    ft.reset()
    n_before = int(ft.ids.max()) + 1 if len(ft.ids) else 0
    tr2 = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(4):  # 15x larger batches after the reset: every device buffer is re-grown
        L, R, _ = big.next_batch()
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, f != 2)
        r = tr2.track_event(t, L, R, f != 2)
        off = int(ft.ids.min() - r.ids.min()) if len(r.ids) else 0
        assert off >= n_before or f > 0
        _compare_tracks(ft, r, ("big", f), id_offset=off)
        assert np.array_equal(ft.gettimesurface(1), tr2.time_surface(1))
    assert len(ft.ids) > 40
    ft.close()


def test_sae_time_surface_1280x720(oracle):
    """largest BASELINE resolution (C5): 21-bit keys (3 x 7-bit passes), ~0.9 M events per camera"""
    W, H = 1280, 720
    rng = np.random.default_rng(99)
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    t0 = 3_000_000_000
    for b in range(2):
        n = 900_000
        L = uniform_batch(W, H, n, t0 + b * 33333, 33333, rng)
        R = uniform_batch(W, H, n // 3, t0 + b * 33333, 33333, rng)
        # concentrate a third of the left events on a 40x40 patch: long same-pixel segments
        L["x"][::3] = 600 + (L["x"][::3] % 40)
        L["y"][::3] = 300 + (L["y"][::3] % 40)
        assert ft.detector.createSAE_stereo(L, R) == 0
        det.create_sae(0, L)
        det.create_sae(1, R)
        for cam in (0, 1):
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
        t = event_times(L)[-1]
        assert np.array_equal(ft.detector.SAEtoTimeSurface_left(t), det.time_surface(0, t))
    ft.close()


@pytest.mark.parametrize("W,H,why", [(1280, 960, "64x32 tiles, 11 pixel bits (32x32 would need 2401 buckets)"),
                                     (97, 61, "sensor smaller than four tiles, ragged tile edges")])
def test_sae_other_sensor_sizes(oracle, W, H, why):
    """the tiled SAE update picks its tile by sensor size (32x16 up to ~0.5 MP, 32x32 up to ~1 MP,
    64x32 beyond): planes, a time surface and the corner flags at sizes that take the other branches"""
    rng = np.random.default_rng(W)
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    n = 400_000 if W > 1000 else 30_000
    t0 = 2_000_000_000
    for b in range(2):
        L = uniform_batch(W, H, n, t0 + b * 33333, 33333, rng)
        R = uniform_batch(W, H, n // 2, t0 + b * 33333, 33333, rng)
        # a busy patch straddling tile borders (long per-pixel histories), some out-of-sensor events
        L["x"][::4] = (W // 2 - 20) + (L["x"][::4] % 40)
        L["y"][::4] = (H // 2 - 20) + (L["y"][::4] % 40)
        L["x"][5::1000] = W + 3
        assert ft.detector.createSAE_stereo(L, R) == det.create_sae(0, L) + det.create_sae(1, R)
        for cam in (0, 1):
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
        t = event_times(L)[-1]
        assert np.array_equal(ft.detector.SAEtoTimeSurface_left(t), det.time_surface(0, t))
        assert np.array_equal(ft.detector.isCorner(L[:100_000]), det.corner_flags(L[:100_000]))
    ft.close()


def test_event_path_on_a_sensor_beyond_the_lds_bitmap(oracle):
    """1920x1080 events (2.07 M pixels): the greedy selection's bitmap does not fit LDS and lives in
    device memory (k_select_gbm); the SAE update takes whichever form the size allows.  Planes, time
    surface, corner flags, the selected corners (with a mask) and three frames of trackEvent equal
    the oracle's."""
    W, H = 1920, 1080
    s = SceneStream(W, H, rate=6e6, seed=77)
    ft = _mk(W, H, max_cnt=200, min_dist=25)
    tr = oracle.Tracker(oracle.make_config(W, H, max_cnt=200, min_dist=25, f_ransac=1))
    det = oracle.Detector(W, H, min_dist=25)
    for f in range(3):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, True)
        _compare_tracks(ft, tr.track_event(t, L, R, True), ("hd", f))
        det.create_sae(0, L)
        det.create_sae(1, R)
        for cam in (0, 1):
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
    assert len(ft.ids) > 100
    ts = det.time_surface(0, t)
    assert np.array_equal(ft.detector.SAEtoTimeSurface_left(t), ts)
    assert np.array_equal(ft.detector.isCorner(L[:50_000]), det.corner_flags(L[:50_000]))
    rng = np.random.default_rng(5)
    mask = np.zeros((H, W), np.uint8)  # nonzero: blocked
    for _ in range(40):
        x, y = rng.integers(0, W - 80), rng.integers(0, H - 80)
        mask[y:y + rng.integers(10, 80), x:x + rng.integers(10, 80)] = 255
    for maxc in (1, 57, 200):
        xy_g, idx_g = ft.Event_FeaturesToTrack(L, maxc, mask)
        xy_c, idx_c = det.features_to_track(L, maxc, 25, mask, ts)
        assert np.array_equal(idx_g, idx_c) and np.array_equal(xy_g, xy_c), maxc
    ft.close()


@pytest.mark.parametrize("path", ["tiled", "sort_per_event"])
def test_sae_bucket_sizes_around_chunk_and_turn_boundaries(oracle, path, monkeypatch):
    """the tiled apply takes a bucket's events in chunks of 64 and turns of 256 (one wave per turn,
    ticket-ordered): buckets of exactly 0, 1, 63..65, 255..257, 511..513, 1023..1025, 2047..2049 and
    4096 + 1 events, packed onto a handful of pixels of one tile so that nearly every event has its
    predecessors in an earlier chunk / turn / wave; ties, polarity flips and backward stamps inside;
    two batches so the carried-in planes matter"""
    for k, v in SAE_PATHS[path].items():
        monkeypatch.setenv(k, v)
    W, H = 346, 260
    sizes = [0, 1, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4097]
    rng = np.random.default_rng(23)
    ft = _mk(W, H)
    det = oracle.Detector(W, H)
    for batch in range(2):
        xs, ys, ts, ps = [], [], [], []
        for i, n in enumerate(sizes):
            # tile i (32x16 tiles at this size: 11 tiles per row), 1-5 pixels inside it
            tx, ty = (i % 10) * 32, (i // 10) * 16 + 32 * batch
            npx = 1 + i % 5
            xs.append(tx + rng.integers(0, npx, n))
            ys.append(np.full(n, ty + 3))
            ts.append(8_000_000 + 50_000 * batch + np.cumsum(rng.choice([0, 0, 400, 11_000, -900], n)))
            ps.append((np.cumsum(rng.random(n) < 0.2) + i) % 2)
        x, y, t, p = (np.concatenate(v) for v in (xs, ys, ts, ps))
        t = np.maximum(t, 1)
        # interleave the buckets (each bucket's own order stays as generated)
        order = np.argsort(np.concatenate([np.arange(len(v)) * (1.0 + 1e-3 * k) for k, v in enumerate(xs)]),
                           kind="stable")
        ev = make_events(x[order], y[order], t[order], p[order])
        for cam, evc in ((0, ev), (1, ev[::-1].copy())):
            (ft.detector.createSAE_left if cam == 0 else ft.detector.createSAE_right)(evc)
            det.create_sae(cam, evc)
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
    ft.close()


def test_selection_large_radius(oracle):
    """min_dist 40 (> 31: two disc rows per lane in k_select) and a tiny max_cnt"""
    W, H = 640, 480
    s = SceneStream(W, H, rate=2e6, seed=8)
    ft = FE.FeatureTracker(FE.make_config(W, H, min_dist=40, max_cnt=25))
    det = oracle.Detector(W, H, min_dist=40)
    for b in range(2):
        L, R, _ = s.next_batch()
        ft.detector.createSAE_stereo(L, R)
        det.create_sae(0, L)
        t = event_times(L)[-1]
        ts = ft.detector.SAEtoTimeSurface_left(t)
        assert np.array_equal(ft.detector.isCorner(L), det.corner_flags(L))
        for maxc in (5, 25):
            xy_g, idx_g = ft.Event_FeaturesToTrack(L, maxc, None)
            xy_c, idx_c = det.features_to_track(L, maxc, 40, np.zeros((H, W), np.uint8), ts)
            assert np.array_equal(idx_g, idx_c) and np.array_equal(xy_g, xy_c)
            assert len(idx_g) >= 3
    ft.close()


def _motion(mod, L, accel=(4.0, 5.0, 3.0), omega=(0.9, -1.4, 2.1), W=640, H=480, dt_frac=0.8):
    """a Motion_correction_value as handle_stereo_event builds it (node:192-254); t1 (header stamp)
    is placed inside the batch so that both the compensated and the plain branch are exercised"""
    t = event_times(L)
    t1 = t[0] + dt_frac * (t[-1] - t[0])
    return mod.make_motion(t1, v=(0.8, -0.3, 0.2), v_pre=(0.7, -0.25, 0.15), accel=accel, omega=omega,
                           fx=0.9 * W, fy=0.9 * W, cx=W / 2.0 + 3.5, cy=H / 2.0 - 2.25)


@pytest.mark.parametrize("path", ["tiled", "sort_walk"])
def test_motion_compensated_sae(oracle, path, monkeypatch):
    """createSAE_left/right with Motion_correction_value (event_detector.cc:102-147,168-210) under
    trackEvent's per-event gate (feature_tracker.cpp:627-641): planes bit-exact, incl. |a| <= 5
    (no warp), large rotations (Pade-5 / Pade-7 branches of Matrix3f::exp) and border pixels.  Through
    the tiled update (the warp inside k_tile_hist, the warped pixel written into the partitioned
    record by k_tile_scatter) and through the radix-sort form (k_sae_keys<true>)."""
    for k, v in SAE_PATHS[path].items():
        monkeypatch.setenv(k, v)
    W, H = 640, 480
    s = SceneStream(W, H, rate=2e6, seed=31)
    cases = [dict(), dict(accel=(1.0, 2.0, 3.0)), dict(omega=(9.0, -14.0, 21.0)),
             dict(omega=(40.0, 35.0, -60.0)), dict(dt_frac=1.5), dict(dt_frac=-0.5)]
    for ci, kw in enumerate(cases):
        ft = _mk(W, H)
        det = oracle.Detector(W, H)
        plain = oracle.Detector(W, H)
        moved = 0
        for b in range(2):
            L, R, _ = s.next_batch()
            L["x"][:200] = np.arange(200) % 14            # border band: kBorder = 6
            L["y"][200:400] = H - 1 - (np.arange(200) % 14)
            mg, mo = _motion(FE, L, **kw), _motion(oracle, L, **kw)
            assert ft.detector.createSAE_stereo_mc(L, R, mg) == 0
            det.create_sae_mc(0, L, L[:1], mo)
            det.create_sae_mc(1, R, L[:1], mo)
            for cam in (0, 1):
                _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
            plain.create_sae(0, L)
            moved += int((plain.get_sae(0)[1] != det.get_sae(0)[1]).sum())
        if ci in (0, 2, 3):
            assert moved > 1000, (ci, moved)   # the warp really moves events
        if ci in (1, 5):
            assert moved == 0, (ci, moved)     # |a| <= 5 or dt <= 0: identical to the plain rule
        ft.close()


def test_track_event_motion_compensated_end_to_end(oracle):
    W, H = 640, 480
    s = SceneStream(W, H, rate=5e6, seed=5)
    ft = FE.FeatureTracker(FE.make_config(W, H))
    tr = oracle.Tracker(oracle.make_config(W, H))
    for f in range(6):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        om = (0.5 + 0.1 * f, -0.8, 1.2)
        ft.trackEvent(t, L, R, f % 3 != 2, measurements=_motion(FE, L, omega=om))
        r = tr.track_event(t, L, R, f % 3 != 2, motion=_motion(oracle, L, omega=om))
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
        assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1))
        _compare_tracks(ft, r, ("mc", f))
    assert len(ft.ids) > 100
    ft.close()


@pytest.mark.parametrize("space", ["host", "device"])
def test_motion_compensated_replay_schedule(oracle, space):
    """the motion-compensated overload announced ahead (esvio_fe_set_next_batch_mc: the warp runs on
    the prefetch stream with the rest of the batch's SAE update; Arc* is asked about the events' own
    pixels, not the warped ones), three batches ahead, lazy mode, Motion_correction_values that change
    from batch to batch incl. |a| <= 5 (no warp): bit-identical to the sequential oracle; a call whose
    Motion_correction_value differs from the announced one is refused"""
    W, H = 640, 480
    s = SceneStream(W, H, rate=5e6, seed=8)
    batches = [s.next_batch()[:2] for _ in range(14)]
    keep = []
    if space == "device":
        def to_dev(a):
            keep.append(FE.EventBuffer(a, FE.DEVICE))
            return keep[-1].arg
        dev = [(to_dev(L), to_dev(R)) for L, R in batches]
    pubs = [f % 3 != 1 for f in range(len(batches))]

    def motions(mod, f, L):
        om = (0.5 + 0.1 * f, -0.8, 1.2 - 0.2 * f)
        return _motion(mod, L, omega=om, accel=(1.0, 2.0, 3.0) if f % 5 == 4 else (4.0, 5.0, 3.0))
    ft = FE.FeatureTracker(FE.make_config(W, H))
    ft.set_lazy_new_stereo(True)
    tr = oracle.Tracker(oracle.make_config(W, H))
    announced = 0
    for f, (L, R) in enumerate(batches):
        arg = (lambda k: dev[k]) if space == "device" else (lambda k: batches[k])
        while announced < min(f + 3, len(batches) - 1):
            announced += 1
            Ln = batches[announced][0]
            ft.set_next_batch(event_times(Ln)[-1], arg(announced)[0], arg(announced)[1], pubs[announced],
                              measurements=motions(FE, announced, Ln))
        t = event_times(L)[-1]
        if f == 6:  # not the announced Motion_correction_value: refused, nothing consumed
            with pytest.raises(FE.FrontendError):
                ft.trackEvent(t, arg(f)[0], arg(f)[1], pubs[f], measurements=motions(FE, f + 1, L))
        ft.trackEvent(t, arg(f)[0], arg(f)[1], pubs[f], measurements=motions(FE, f, L))
        r = tr.track_event(t, L, R, pubs[f], motion=motions(oracle, f, L))
        ft.finish()
        _compare_tracks(ft, r, ("mc replay", space, f))
    assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
    assert len(ft.ids) > 100
    ft.close()
    for b in keep:
        b.free()


@pytest.mark.parametrize("equalize,hint,depth,lazy,launch", [
    (0, "none", 1, 0, 0), (0, "right", 1, 0, 0), (0, "wrong", 1, 0, 0), (1, "right", 1, 0, 0), (1, "none", 1, 0, 0),
    (0, "right", 2, 0, 0), (1, "right", 2, 0, 0), (0, "right", 2, 1, 0), (0, "none", 1, 1, 0), (1, "right", 2, 1, 0),
    (0, "right", 3, 0, 0), (0, "right", 3, 1, 0), (1, "right", 3, 1, 0),
    # ... and with the prefetch launches issued by the handle's launch thread (esvio_fe_set_launch_thread)
    (0, "right", 3, 1, 1), (1, "right", 2, 0, 1), (0, "wrong", 1, 0, 1), (0, "none", 1, 1, 1), (0, "right", 3, 0, 1)])
def test_next_batch_prefetch_is_transparent(oracle, equalize, hint, depth, lazy, launch):
    """esvio_fe_set_next_batch (replay mode: the next one or two batches' SAE update / images — and,
    with the PUB hint, their Arc* pass — run on a second stream, and the next frame's temporal LK
    is launched speculatively on a third) must not change a single result bit, whether the hint is
    absent, right or wrong; also a mismatching follow-up call is refused."""
    W, H = 640, 480
    s = SceneStream(W, H, rate=5e6, seed=6)
    batches = [s.next_batch() for _ in range(12)]
    kw = dict(f_ransac=1, equalize=equalize)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(bool(lazy))
    if launch:
        ft.set_launch_thread(True)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    pubs = [(f % 3) != 1 for f in range(len(batches))]
    announced = 0
    for f, (L, R, _) in enumerate(batches):
        t = event_times(L)[-1]
        # batch 5 is never announced (mixing prefetched and plain calls is legal)
        if f == 5:
            announced = 5
        hi = min(f + depth, len(batches) - 1)
        if f <= 4:
            hi = min(hi, 4)
        while announced < hi:
            announced += 1
            Ln, Rn, _ = batches[announced]
            h = {"none": False, "right": pubs[announced], "wrong": not pubs[announced]}[hint]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, h)
        if launch and f in (3, 6, 9):  # (switched off and on again with batches announced and in flight)
            ft.set_launch_thread(f == 6)
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        if lazy:
            # what a lazily returned frame already guarantees: the left side and the PointCloud rows
            for k in ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity"):
                assert np.array_equal(getattr(ft, k), getattr(r, k)), (k, f)
            if f % 2:   # ... and after finish() everything (odd frames only: the next call must
                ft.finish()   # complete a pending frame by itself as well)
                _compare_tracks(ft, r, ("lazy+finish", hint, depth, f))
        else:
            _compare_tracks(ft, r, ("prefetch", hint, depth, f))
        if f == 4:  # nothing pending: the taps show this very frame
            assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
    assert len(ft.ids) > 100
    # a call that does not match the announced batch is an error, not silent corruption
    L, R, _ = batches[0]
    ft.set_next_batch(1.0, L, R)
    ft.trackEvent(event_times(batches[1][0])[-1] + 1.0, batches[1][0], batches[1][1], False)
    with pytest.raises(FE.FrontendError):
        ft.trackEvent(2.0, batches[2][0], batches[2][1], False)
    ft.close()


@pytest.mark.parametrize("split,lk_accum,pattern", [("1", 2, "alt"), ("1", 2, "2of3"), ("0", 2, "alt"), (None, 2, "alt"),
                                                     ("1", 1, "alt"), (None, 1, "alt")])
def test_replay_with_the_unpublished_frames_stereo_lk_on_its_own_stream(oracle, monkeypatch, split, lk_accum, pattern):
    """the replay schedule the bench runs — device-resident batches announced three ahead, lazy mode, the launch
    thread — with the stereo LK of the unpublished frames on the second stereo stream (ESVIO_FE_STEREO_SPLIT=1: forced;
    unset: the library's own rule, float-order LK + launch thread + device-resident batches), switched off, and with
    publish patterns that put two published / two unpublished frames next to each other: every frame bit-identical to
    the sequential oracle, lazily returned frames completed by finish() or by the next call"""
    if split is None:
        monkeypatch.delenv("ESVIO_FE_STEREO_SPLIT", raising=False)
    else:
        monkeypatch.setenv("ESVIO_FE_STEREO_SPLIT", split)
    W, H = 640, 480
    s = SceneStream(W, H, rate=5e6, seed=21)
    batches = [s.next_batch()[:2] for _ in range(16)]
    keep = []

    def to_dev(a):
        keep.append(FE.EventBuffer(a, FE.DEVICE))
        return keep[-1].arg
    dev = [(to_dev(L), to_dev(R)) for L, R in batches]
    pubs = [f % 2 == 0 for f in range(len(batches))] if pattern == "alt" else [f % 3 != 1 for f in range(len(batches))]
    kw = dict(f_ransac=1, lk_accum=lk_accum)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    ft.set_launch_thread(True)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    announced = 0
    for f, (L, R) in enumerate(batches):
        while announced < min(f + 3, len(batches) - 1):
            announced += 1
            ft.set_next_batch(event_times(batches[announced][0])[-1], dev[announced][0], dev[announced][1], pubs[announced])
        t = event_times(L)[-1]
        if f == 9:  # (the launch thread switched off and on again with batches in flight: the rule follows it)
            ft.set_launch_thread(False)
        if f == 11:
            ft.set_launch_thread(True)
        ft.trackEvent(t, dev[f][0], dev[f][1], pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        for k in ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity"):
            assert np.array_equal(getattr(ft, k), getattr(r, k)), (k, f)
        if f % 3 == 2 or f == len(batches) - 1:
            ft.finish()
            _compare_tracks(ft, r, ("stereo split", split, lk_accum, pattern, f))
    assert len(ft.ids) > 100
    ft.close()
    for b in keep:
        b.free()


def test_two_batches_ahead_need_an_exact_pub_hint(monkeypatch):
    """with two batches in flight the SAE has moved past a frame by the time it is tracked, so a
    published frame whose hint was 0 (no prefetched Arc* pass) is refused instead of detecting on
    the wrong surface; a seventh announcement is refused too.  (Host batches staged by the helper
    threads are taken up when they have arrived, which would make "two in flight" a matter of timing
    here: the stager is off for this test.)"""
    monkeypatch.setenv("ESVIO_FE_STAGE_THREADS", "0")
    W, H = 346, 260
    s = SceneStream(W, H, rate=2e6, seed=2)
    b = [s.next_batch() for _ in range(8)]
    t = [event_times(x[0])[-1] for x in b]
    ft = FE.FeatureTracker(FE.make_config(W, H))
    ft.set_next_batch(t[1], b[1][0], b[1][1], False)
    ft.set_next_batch(t[2], b[2][0], b[2][1], False)
    for k in (3, 4, 5, 6):
        ft.set_next_batch(t[k], b[k][0], b[k][1], False)
    with pytest.raises(FE.FrontendError):
        ft.set_next_batch(t[7], b[7][0], b[7][1], False)
    ft.trackEvent(t[0], b[0][0], b[0][1], True)       # enqueues the prefetch of frames 1 and 2
    with pytest.raises(FE.FrontendError):
        ft.trackEvent(t[1], b[1][0], b[1][1], True)   # hint said "not published"
    ft.reset()
    ft.trackEvent(t[0], b[0][0], b[0][1], True)       # usable again after reset
    assert len(ft.ids) > 0
    ft.close()


# ------------------------------------------------------------------ image front-end
@pytest.mark.parametrize("W,H,md,use_mask", [(160, 120, 10, False), (346, 260, 20, True),
                                             (640, 480, 30, True), (640, 480, 1, False),
                                             (345, 259, 15, True), (1280, 720, 30, False)])
def test_good_features_to_track_matches_oracle(oracle, W, H, md, use_mask):
    """cv::goodFeaturesToTrack restated (feature_tracker.cpp:228): response map bit-exact (Sobel,
    structure tensor, OpenCV's running column sum, min eigenvalue), corners identical and in order"""
    from esvio_amd.synth import ImageStream
    s = ImageStream(W, H, seed=W + md)
    ft = _mk(W, H, max_cnt=150, min_dist=10)
    rng = np.random.default_rng(md)
    for k in range(2):
        img = s.next_frame()[0] if k == 0 else rng.integers(0, 256, (H, W), dtype=np.uint8)
        mask = None
        if use_mask:
            mask = np.full((H, W), 255, np.uint8)
            for _ in range(25):
                x, y = rng.integers(0, W - 30), rng.integers(0, H - 30)
                mask[y:y + rng.integers(5, 30), x:x + rng.integers(5, 30)] = 0
        for maxc in (150, 7):
            cg, eg = ft.goodFeaturesToTrack(img, maxc, 0.01, md, mask, want_eig=True)
            co, eo = oracle.good_features_to_track(img, maxc, 0.01, md, mask, want_eig=True)
            assert np.array_equal(eg.view(np.uint32), eo.view(np.uint32)), "cornerMinEigenVal differs"
            assert cg.shape == co.shape and np.array_equal(cg, co), (k, maxc)
            assert len(cg) > 3
    # degenerate inputs: flat image -> nothing; everything masked -> nothing
    assert len(ft.goodFeaturesToTrack(np.full((H, W), 9, np.uint8), 10, 0.01, md)) == 0
    assert len(ft.goodFeaturesToTrack(img, 10, 0.01, md, np.zeros((H, W), np.uint8))) == 0
    ft.close()


@pytest.mark.parametrize("equalize", [0, 1])
def test_track_image_end_to_end_matches_oracle(oracle, equalize):
    """FeatureTracker::trackImage (feature_tracker.cpp:164-338), 8 stereo frames 640x480, forward /
    backward checks on, published and unpublished frames, one frame without a right image:
    ids, track counts and every float result vector bit-exact"""
    from esvio_amd.synth import ImageStream
    W, H = 640, 480
    s = ImageStream(W, H, velocity=(4, -3), disparity=11, seed=5)
    kw = dict(max_cnt=150, min_dist=30, flow_back=1, equalize=equalize)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(8):
        L, R, t = s.next_frame()
        pub = f % 3 != 1
        if f == 5:
            R = None
        ft.trackImage(t, L, R, pub)
        _compare_tracks(ft, tr.track_image(t, L, R, pub), ("image", equalize, f))
    assert len(ft.ids) > 100 and ft.track_cnt.max() >= 6 and len(ft.ids_right) > 80
    ft.close()


@pytest.mark.parametrize("W,H,max_cnt,min_dist", [(1440, 1080, 175, 40), (1920, 1200, 200, 30), (1224, 1024, 150, 20)])
def test_track_image_at_the_shipped_frame_camera_sizes(oracle, W, H, max_cnt, min_dist):
    """the frame cameras of the shipped ESVIO configs (config/esvio_DSEC, esvio_ecmd, esvio_VECtor:
    cam0_esvio.yaml image_width/height, esvio.yaml max_cnt_img / min_dist_img): above ~1.3 M pixels
    the min-distance bitmap of the selection lives in device memory; trackImage equals the oracle's"""
    from esvio_amd.synth import ImageStream
    s = ImageStream(W, H, velocity=(5, 3), disparity=14, seed=W)
    kw = dict(max_cnt=max_cnt, min_dist=min_dist, flow_back=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(3):
        L, R, t = s.next_frame()
        ft.trackImage(t, L, R, f != 1)
        _compare_tracks(ft, tr.track_image(t, L, R, f != 1), ("image", W, f))
    assert len(ft.ids) > max_cnt // 2 and len(ft.ids_right) > max_cnt // 4
    ft.close()


@pytest.mark.parametrize("k", [1, 2])
def test_median_blur_kernel_size(oracle, k):
    """median_blur_kernel_size k > 0 (event_detector.cc:262-264): cv::medianBlur(2k+1) of each
    rendered surface — the standalone render, and the whole tracker on top of it"""
    W, H = 346, 260
    s = SceneStream(W, H, rate=2e6, seed=17, n_rect=14, size=(30.0, 90.0))
    kw = dict(max_cnt=100, min_dist=10, f_ransac=1, median_blur_kernel_size=k)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(5):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, f % 2 == 0)
        r = tr.track_event(t, L, R, f % 2 == 0)
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
        assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1))
        _compare_tracks(ft, r, ("median", k, f))
    assert len(ft.ids) > 30
    # the plain render entry point
    det = oracle.Detector(W, H, median_blur_kernel_size=k)
    L, R, _ = s.next_batch()
    det.create_sae(0, L)
    ft2 = FE.FeatureTracker(FE.make_config(W, H, median_blur_kernel_size=k))
    ft2.detector.createSAE_left(L)
    t = event_times(L)[-1]
    assert np.array_equal(ft2.detector.SAEtoTimeSurface_left(t), det.time_surface(0, t))
    ft.close()
    ft2.close()


@pytest.mark.parametrize("lazy,threads,ahead", [(0, 1, 2), (1, 1, 2), (1, 4, 3), (0, 1, 3)])
def test_replay_mode_soak(oracle, lazy, threads, ahead):
    """60 frames with two or three batches announced ahead (three: the temporal LK of the frame
    after an unpublished one is chained to it on the device), irregular publish pattern, event rate changing
    from batch to batch (buffers regrow, speculative temporal LK sizes change), an empty right batch
    now and then: every frame bit-identical to the sequential oracle"""
    W, H = 346, 260
    rng = np.random.default_rng(99)
    s = SceneStream(W, H, rate=2.5e6, seed=41, n_rect=10, size=(25.0, 80.0), t0_us=3_000_000_000)
    batches = []
    for f in range(60):
        L, R, _ = s.next_batch()
        if rng.integers(0, 2):  # a quarter of the events on some frames
            L, R = L[::4].copy(), R[::4].copy()
        if f % 11 == 7:
            R = R[:0]
        batches.append((L, R))
    pubs = [bool(rng.integers(0, 3) != 0) for _ in batches]
    kw = dict(max_cnt=120, min_dist=10, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(bool(lazy))
    ft.set_host_threads(threads)  # (RANSAC helpers: no effect on any result)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    from esvio_amd.node import pack_track_records
    announced = 0
    for f, (L, R) in enumerate(batches):
        while announced < min(f + ahead, len(batches) - 1):
            announced += 1
            Ln, Rn = batches[announced]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, pubs[announced])
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        if lazy:  # the published rows are complete even before the frame is finished
            rows = ft.pack_track_records()
            if rng.integers(0, 3) == 0:
                ft.finish()
                _compare_tracks(ft, r, ("soak lazy", f))
                assert np.array_equal(rows.view(np.uint32), pack_track_records(ft, 120).view(np.uint32))
        else:
            _compare_tracks(ft, r, ("soak", f))
    assert len(ft.ids) > 40 and ft.track_cnt.max() >= 3
    ft.close()


def test_replay_hd_shape(oracle):
    """BASELINE C5's sensor shape (1280x720, not a shipped resolution) with max_cnt 500 / min_dist 14
    through the replay schedule (three batches ahead, lazy, chained temporal LK, fused time surface +
    pyramid kernel on 160x90 level-3 tiles): bit-identical to the oracle"""
    W, H = 1280, 720
    s = SceneStream(W, H, rate=4e6, seed=9, n_rect=16, size=(40.0, 160.0))
    batches = [s.next_batch()[:2] for _ in range(8)]
    pubs = [f % 2 == 0 for f in range(len(batches))]
    kw = dict(max_cnt=500, min_dist=14, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    ft.set_host_threads(3)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    announced = 0
    for f, (L, R) in enumerate(batches):
        while announced < min(f + 3, len(batches) - 1):
            announced += 1
            Ln, Rn = batches[announced]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, pubs[announced])
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        ft.finish()
        _compare_tracks(ft, r, ("hd", f))
    assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
    assert len(ft.ids) > 100
    ft.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_replay_random_schedules(oracle, seed):
    """random publish pattern, a random number of batches (0..3) announced ahead of every call, lazy
    mode and RANSAC helper threads switched at random between calls, finish() only now and then:
    speculative / chained launches get used, skipped and cancelled in every combination, and every
    frame stays bit-identical to the sequential oracle"""
    W, H = 346, 260
    rng = np.random.default_rng(1000 + seed)
    s = SceneStream(W, H, rate=2.5e6, seed=50 + seed, n_rect=10, size=(25.0, 80.0))
    batches = [s.next_batch()[:2] for _ in range(36)]
    p_pub = [0.3, 0.5, 0.7, 0.9][seed - 1]
    pubs = [bool(rng.random() < p_pub) for _ in batches]
    kw = dict(max_cnt=100 + 30 * seed, min_dist=8 + 2 * seed, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    announced = 0
    for f, (L, R) in enumerate(batches):
        if rng.random() < 0.2:
            ft.set_lazy_new_stereo(bool(rng.integers(0, 2)))
        if rng.random() < 0.15:
            ft.set_host_threads(int(rng.integers(1, 5)))
        announced = max(announced, f)  # (a frame that was never announced is a plain call)
        want = min(f + int(rng.integers(0, 4)), len(batches) - 1)
        while announced < want:
            announced += 1
            Ln, Rn = batches[announced]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, pubs[announced])
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        for k in ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity"):
            assert np.array_equal(getattr(ft, k), getattr(r, k)), (k, f)
        if rng.random() < 0.5:
            ft.finish()
            _compare_tracks(ft, r, ("random", seed, f))
    ft.finish()
    _compare_tracks(ft, r, ("random", seed, "end"))
    assert len(ft.ids) > 20
    ft.close()


@pytest.mark.parametrize("opt", ["ESVIO_FE_NO_CHAIN", "ESVIO_FE_DEDUP", "ESVIO_FE_NO_DEDUP", "ESVIO_FE_NO_FUSE",
                                 "ESVIO_FE_SAE_SORT", "ESVIO_FE_SAE_SORT+ESVIO_FE_SAE_EV_MIN", "ESVIO_FE_SELECT_SERIAL"])
def test_replay_options_do_not_change_results(oracle, opt, monkeypatch):
    """the measurement switches read at esvio_fe_create (no chained temporal LK; the per-pixel dedup of the Arc* candidates forced on
    (handles with max_cnt <= 500 run without it) / off; unfused time surface +
    pyrDown kernels; the radix-sort form of the SAE update instead of the tiled one, with the
    per-pixel walk and with the per-event apply kernels for batches of >= 1 event instead of >= 2^20; the
    one-wave selection kernel that sensors without LDS for the 16-wave kernel's queue take)
    leave every result bit-identical to the oracle"""
    for o in opt.split("+"):
        monkeypatch.setenv(o, "1")
    W, H = 346, 260
    s = SceneStream(W, H, rate=3e6, seed=77, n_rect=10, size=(25.0, 80.0))
    batches = [s.next_batch()[:2] for _ in range(14)]
    pubs = [f % 2 == 0 for f in range(len(batches))]
    kw = dict(max_cnt=150, min_dist=10, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    announced = 0
    for f, (L, R) in enumerate(batches):
        while announced < min(f + 3, len(batches) - 1):
            announced += 1
            Ln, Rn = batches[announced]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, pubs[announced])
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        ft.finish()
        _compare_tracks(ft, r, (opt, f))
    assert len(ft.ids) > 40
    ft.close()


def test_pack_track_records_matches_the_node_packing():
    """esvio_fe_pack_track_records == the PointCloud packing of stereo_event_tracker_node.cpp:273-329
    (python mirror node.pack_track_records) on live tracker results"""
    from esvio_amd.node import pack_track_records
    W, H = 346, 260
    s = SceneStream(W, H, rate=2e6, seed=5, n_rect=12, size=(30.0, 90.0))
    ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=90, min_dist=10))
    for f in range(5):
        L, R, _ = s.next_batch()
        ft.trackEvent(event_times(L)[-1], L, R, True)
        a = ft.pack_track_records()
        b = pack_track_records(ft, 90)
        assert a.shape == b.shape == (180, 8) and np.array_equal(a.view(np.uint32), b.view(np.uint32)), f
    assert (a[:, 3] >= 0).sum() > 60
    ft.close()


@pytest.mark.parametrize("W,H", [(345, 259), (321, 243), (250, 187)])
def test_odd_sensor_sizes_end_to_end(oracle, W, H):
    """sizes that are not multiples of 2/4/32: pyramid levels round up ((n+1)/2), bitmap rows end in
    a partial word, image rows are not dword multiples — whole tracker + taps against the oracle"""
    s = SceneStream(W, H, rate=1.5e6, seed=W, n_rect=12, size=(25.0, 80.0))
    kw = dict(max_cnt=100, min_dist=9, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    batches = [s.next_batch() for _ in range(7)]
    for f, (L, R, _) in enumerate(batches):
        t = event_times(L)[-1]
        if 1 <= f < len(batches) - 1 and f != 3:  # mix prefetched and plain calls
            Ln, Rn, _ = batches[f + 1]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, (f + 1) % 2 == 0)
        ft.trackEvent(t, L, R, f % 2 == 0)
        _compare_tracks(ft, tr.track_event(t, L, R, f % 2 == 0), ("odd", W, H, f))
        if f in (0, 3):  # nothing pending: the taps show this frame
            assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
            assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1))
    assert len(ft.ids) > 30
    # pyramids and Scharr of an arbitrary image at this size
    img = np.random.default_rng(W).integers(0, 256, (H, W), dtype=np.uint8)
    levels = ft.build_pyramid(img, 3)
    assert len(levels) == oracle.pyr_levels(W, H) + 1
    cur = img
    for l, (im, dv) in enumerate(levels):
        if l > 0:
            cur = oracle.pyr_down(cur)
        assert np.array_equal(im, cur), ("level", l)
        assert np.array_equal(dv, oracle.scharr(cur)), ("scharr", l)
    ft.close()


def test_event_and_image_handles_interleaved(oracle):
    """ESVIO mode runs both front-ends (two nodes in the reference): an event handle in replay mode
    and an image handle used alternately in one process do not disturb each other"""
    from esvio_amd.synth import ImageStream
    W, H = 346, 260
    es = SceneStream(W, H, rate=1.5e6, seed=8, n_rect=12, size=(25.0, 80.0))
    ims = ImageStream(W, H, velocity=(2, 1), disparity=6, seed=8)
    ekw = dict(max_cnt=90, min_dist=10, f_ransac=1)
    ikw = dict(max_cnt=70, min_dist=20, flow_back=1)
    fe, fi = FE.FeatureTracker(FE.make_config(W, H, **ekw)), FE.FeatureTracker(FE.make_config(W, H, **ikw))
    oe, oi = oracle.Tracker(oracle.make_config(W, H, **ekw)), oracle.Tracker(
        oracle.make_config(W, H, **ikw))
    batches = [es.next_batch() for _ in range(6)]
    for f, (L, R, _) in enumerate(batches):
        if f + 1 < len(batches):
            Ln, Rn, _ = batches[f + 1]
            fe.set_next_batch(event_times(Ln)[-1], Ln, Rn, True)
        t = event_times(L)[-1]
        fe.trackEvent(t, L, R, True)
        IL, IR, ti = ims.next_frame()
        fi.trackImage(ti, IL, IR, True)
        _compare_tracks(fe, oe.track_event(t, L, R, True), ("ev", f))
        _compare_tracks(fi, oi.track_image(ti, IL, IR, True), ("img", f))
    assert len(fe.ids) > 30 and len(fi.ids) > 40
    fe.close()
    fi.close()


def test_randomized_small_cases(oracle):
    """40 seeded random cases: sensor sizes from 48x44 up, batch sizes around the sort tile (1, 2,
    2047..2049, 4097 ...), clustered pixels, tied / reversed stamps, both polarity modes: SAE planes,
    time surfaces (several sync times) and Arc* flags bit-exact"""
    rng = np.random.default_rng(2024)
    sizes = [1, 2, 3, 63, 64, 65, 255, 257, 2047, 2048, 2049, 4097, 6000, 10000]
    for case in range(40):
        W, H = int(rng.integers(48, 200)), int(rng.integers(44, 150))
        ign = int(rng.integers(0, 2))
        thr = float(rng.choice([0.0, 0.001, 0.01, 0.05]))
        ft = FE.FeatureTracker(FE.make_config(W, H, ignore_polarity=ign, feature_filter_threshold=thr,
                                              decay_ms=float(rng.choice([5.0, 20.0, 100.0]))))
        det = oracle.Detector(W, H, decay_ms=ft.cfg.decay_ms, ignore_polarity=ign, filter_threshold=thr)
        t0 = 9_000_000
        for b in range(3):
            n = int(rng.choice(sizes))
            if rng.integers(0, 2):  # clustered: few pixels, long per-pixel runs
                cx, cy = rng.integers(0, W, 6), rng.integers(0, H, 6)
                k = rng.integers(0, 6, n)
                x, y = cx[k], cy[k]
            else:
                x, y = rng.integers(0, W, n), rng.integers(0, H, n)
            dt = rng.choice([0, 0, 1, 50, 3000, -200], n)
            t = t0 + np.maximum(np.cumsum(dt), -t0 + 1)
            t0 = int(t[-1]) + int(rng.integers(0, 20000))
            ev = make_events(x, y, t, rng.integers(0, 2, n))
            cam = int(rng.integers(0, 2))
            (ft.detector.createSAE_right if cam else ft.detector.createSAE_left)(ev)
            det.create_sae(cam, ev)
            _planes_equal(ft.detector.get_sae(cam), det.get_sae(cam))
            if cam == 0:
                assert np.array_equal(ft.detector.isCorner(ev), det.corner_flags(ev)), (case, b)
            for ts in (t[-1] * 1e-6, t[-1] * 1e-6 + 0.013, t[0] * 1e-6 - 0.2):
                f = ft.detector.SAEtoTimeSurface_right if cam else ft.detector.SAEtoTimeSurface_left
                assert np.array_equal(f(ts), det.time_surface(cam, ts)), (case, b, ts)
        ft.close()


def test_create_destroy_many_handles():
    """handles can be created and destroyed repeatedly (streams, events, pinned and device memory are
    all released) and several can be alive at once"""
    probe = FE.FeatureTracker(FE.make_config(346, 260))  # (stays alive: only asks for the free memory)

    def free_bytes():
        return probe.device_memory()[0]

    W, H = 346, 260
    s = SceneStream(W, H, rate=1e6, seed=1, n_rect=8, size=(25.0, 80.0))
    L, R, _ = s.next_batch()
    free0 = None
    for k in range(25):
        fts = [FE.FeatureTracker(FE.make_config(W, H, equalize=k % 2, median_blur_kernel_size=k % 3))
               for _ in range(3)]
        for ft in fts:
            ft.trackEvent(event_times(L)[-1], L, R, True)
            assert len(ft.ids) > 10
        for ft in fts:
            ft.close()
        if k == 2:
            free0 = free_bytes()
    assert free_bytes() >= free0 - (64 << 20), "device memory leaks per handle"
    probe.close()
