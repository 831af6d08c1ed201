"""Host-resident event batches (ESVIO_FE_HOST — what the reference's `const dvs_msgs::EventArray&`
interface hands over, feature_tracker.h:51-52) on their way to the device: pinned chunks + DMA by
helper threads, started by esvio_fe_set_next_batch (fe_evstage.cpp).  Whatever the staging does —
helper count, a pinned source, batches taken up late because they were still on their way, a reset
in between, batches below the staging threshold — every frame equals the sequential oracle's."""
import numpy as np
import pytest

from esvio_amd import frontend as FE
from esvio_amd.events import event_times
from esvio_amd.synth import PoissonStream, SceneStream

pytestmark = pytest.mark.gpu

KEYS = ("cur_pts", "cur_un_pts", "pts_velocity", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity")


def _same(ft, r, tag):
    assert np.array_equal(ft.ids, r.ids) and np.array_equal(ft.track_cnt, r.track_cnt), tag
    assert np.array_equal(ft.ids_right, r.ids_right), tag
    for k in KEYS:
        a, b = getattr(ft, k), getattr(r, k)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (tag, k)


def _batches(W, H, n, rate, seed, thin=True):
    rng = np.random.default_rng(seed)
    s = SceneStream(W, H, rate=rate, seed=seed, n_rect=12, size=(30.0, 90.0))
    out = []
    for f in range(n):
        L, R, _ = s.next_batch()
        if thin and rng.integers(0, 3) == 0:  # sizes change from batch to batch: slots regrow, chunk counts change
            L, R = L[::5].copy(), R[::5].copy()
        if f % 9 == 5:
            R = R[:0]
        out.append((L, R))
    return out


@pytest.mark.parametrize("threads,ahead,pinned", [(2, 4, False), (1, 3, False), (4, 2, False), (0, 3, False),
                                                  (2, 4, True), (2, 1, False)])
def test_host_resident_replay_matches_the_oracle(oracle, monkeypatch, threads, ahead, pinned):
    """replay schedule over host batches, announced 1..4 ahead (4: one more than the prefetch depth,
    what bench.py does for host batches), lazy mode, with 0 (stager off: the runtime's own pageable
    copy), 1, 2 or 4 helper threads, and from a source that is pinned already (one DMA, no memcpy)"""
    monkeypatch.setenv("ESVIO_FE_STAGE_THREADS", str(threads))
    W, H = 640, 480
    batches = _batches(W, H, 30, 4e6, 17)
    bufs = []
    if pinned:  # copies in pinned memory of the library's own HIP runtime (esvio_fe_mem_alloc)
        pb = []
        for L, R in batches:
            bl, br = FE.EventBuffer(L), FE.EventBuffer(R)
            bufs += [bl, br]
            pb.append((bl.array, br.array))
        batches = pb
    rng = np.random.default_rng(5)
    pubs = [bool(rng.integers(0, 3) != 0) for _ in batches]
    kw = dict(max_cnt=200, min_dist=10, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    announced = 0
    for f, (L, R) in enumerate(batches):
        while announced < min(f + ahead, len(batches) - 1):
            announced += 1
            Ln, Rn = batches[announced]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, pubs[announced])
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        if f % 4 == 3:
            ft.finish()
            _same(ft, r, ("host replay", threads, ahead, f))
    ft.finish()
    _same(ft, r, ("host replay end", threads, ahead))
    assert len(ft.ids) > 60
    ft.close()
    for b in bufs:
        b.free()


def test_host_batches_without_announcement_and_across_a_reset(oracle):
    """the drop-in call of INTEGRATION.md: esvio_fe_track_event(ESVIO_FE_HOST) one batch at a time (the
    helpers and the calling thread stage the chunks together), small batches below the staging
    threshold in between, announced batches dropped by esvio_fe_reset while they are being staged"""
    W, H = 346, 260
    kw = dict(max_cnt=120, min_dist=10, f_ransac=1)
    batches = _batches(W, H, 14, 2.5e6, 3, thin=False)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f, (L, R) in enumerate(batches[:6]):
        if f == 3:
            L, R = L[:3000].copy(), R[:2000].copy()  # 80 KB: plain copy
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, f % 2 == 0)
        _same(ft, tr.track_event(t, L, R, f % 2 == 0), ("direct", f))
    # announce three, track one, reset
    for k in (6, 7, 8):
        ft.set_next_batch(event_times(batches[k][0])[-1], batches[k][0], batches[k][1], k % 2 == 0)
    L, R = batches[6]
    ft.trackEvent(event_times(L)[-1], L, R, True)
    _same(ft, tr.track_event(event_times(L)[-1], L, R, True), "before reset")
    ft.reset()
    tr2 = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f, (L, R) in enumerate(batches[9:]):
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, True)
        r = tr2.track_event(t, L, R, True)
        # (ids keep counting across a reset: n_id is a static in the reference, feature_tracker.cpp:9)
        off = int(ft.ids.min() - r.ids.min()) if len(r.ids) else 0
        assert np.array_equal(ft.ids, r.ids + off) and np.array_equal(ft.ids_right, r.ids_right + off), ("after reset", f)
        for k in KEYS:
            assert np.array_equal(getattr(ft, k).view(np.uint32), getattr(r, k).view(np.uint32)), ("after reset", f, k)
    assert len(ft.ids) > 30
    ft.close()


def test_a_batch_announced_late_is_taken_up_by_its_own_call(oracle):
    """announce a big host batch and track it at once: the call that needs it waits for its staging
    (helping with the chunks) instead of finding it prefetched"""
    W, H = 640, 480
    kw = dict(max_cnt=150, min_dist=10, f_ransac=1)
    s = SceneStream(W, H, rate=8e6, seed=11, n_rect=20, size=(40.0, 120.0))
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(6):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        if f:
            ft.set_next_batch(t, L, R, True)
        ft.trackEvent(t, L, R, True)
        _same(ft, tr.track_event(t, L, R, True), ("late", f))
    ft.close()


@pytest.mark.parametrize("space", ["host", "device"])
def test_announce_after_return_with_lazy_unpublished_frames(oracle, space):
    """track(k) returns, THEN batch k+1 is announced and tracked: the take-up of the late announcement
    runs on the prefetch stream while frame k — a first frame, or an unpublished one that returned lazily,
    so nobody has synchronised the main stream — may still be using the planes and the partition scratch
    there (round-3 review: the event the take-up waits for was only recorded for batches announced DURING
    the previous call).  Every frame must still equal the sequential oracle's."""
    W, H = 640, 480
    kw = dict(max_cnt=150, min_dist=10, f_ransac=1)
    s = SceneStream(W, H, rate=6e6, seed=23, n_rect=20, size=(40.0, 120.0))
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    pubs = [False, False, True, False, False, True, True, False, False, False, True, False]
    bufs = []
    for f, pub in enumerate(pubs):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        if space == "device":
            bl, br = FE.EventBuffer(L, FE.DEVICE), FE.EventBuffer(R, FE.DEVICE)
            bufs += [bl, br]
            aL, aR = bl.arg, br.arg
        else:
            aL, aR = L, R
        if f:  # (announced only now: the previous call has returned)
            ft.set_next_batch(t, aL, aR, pub)
        ft.trackEvent(t, aL, aR, pub)
        r = tr.track_event(t, L, R, pub)
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0)), ("left surface", f)
        assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1)), ("right surface", f)
        if f % 3 == 2:
            ft.finish()
            _same(ft, r, ("announce after return", space, f))
    ft.finish()
    _same(ft, r, ("announce after return end", space))
    ft.close()
    for b in bufs:
        b.free()


@pytest.mark.parametrize("space", ["host", "device", "registered"])
@pytest.mark.parametrize("split", [True, False])
def test_plain_calls_with_the_cameras_on_two_streams(oracle, monkeypatch, space, split):
    """A plain call (nothing announced, nothing lazy: the reference node's pattern) runs the left camera's
    SAE update + image on the main stream and the right camera's on the stereo stream, under the temporal
    LK — for a pageable batch with the left array as a DMA of its own (ESVIO_FE_NO_CAMSPLIT=1: both on the
    main stream, as before).  Calls that cannot split are mixed in (an empty right array, a batch below the
    staging threshold, an announced batch, a lazy stretch) and both cameras' surfaces and images are read
    back right after split calls (the main stream has to follow the stereo stream's right-camera chain):
    every frame equals the sequential oracle's."""
    if not split:
        monkeypatch.setenv("ESVIO_FE_NO_CAMSPLIT", "1")
    W, H = 640, 480
    kw = dict(max_cnt=150, min_dist=10, f_ransac=1)
    s = SceneStream(W, H, rate=5e6, seed=41, n_rect=20, size=(40.0, 120.0))
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    pubs = [True, False, True, True, False, False, True, False, True, True, False, True, True, False, True, True]
    batches = [s.next_batch()[:2] for _ in pubs]
    batches[4] = (batches[4][0], batches[4][1][:0])                                   # no right events
    batches[7] = (batches[7][0][:4000].copy(), batches[7][1][:3000].copy())           # 112 KB: plain copy
    bufs = []

    def args(L, R):
        if space == "host":
            return L, R
        if space == "registered":  # the caller's own arrays, page-locked where they lie (esvio_fe_register_host_buffer)
            regs = [FE.RegisteredEvents(a) if len(a) else None for a in (L, R)]
            bufs.extend(r for r in regs if r)
            return tuple(r.array if r else a for r, a in zip(regs, (L, R)))
        bl, br = FE.EventBuffer(L, FE.DEVICE), FE.EventBuffer(R, FE.DEVICE)
        bufs.extend([bl, br])
        return bl.arg, br.arg

    for f, pub in enumerate(pubs):
        L, R = batches[f]
        t = event_times(L)[-1]
        aL, aR = args(L, R)
        if f == 9:  # one announced batch in between: calls 9 and 10 are not plain
            Ln, Rn = batches[10]
            nL, nR = args(Ln, Rn)
            ft.set_next_batch(event_times(Ln)[-1], nL, nR, pubs[10])
        if f == 10:
            aL, aR = nL, nR
        if f == 12:
            ft.set_lazy_new_stereo(True)
        if f == 14:
            ft.finish()
            ft.set_lazy_new_stereo(False)
        ft.trackEvent(t, aL, aR, pub)
        r = tr.track_event(t, L, R, pub)
        if f % 2 == 0:
            for cam in (0, 1):
                assert np.array_equal(ft.gettimesurface(cam), tr.time_surface(cam)), ("surface", cam, f)
        if f in (12, 13):
            ft.finish()
        _same(ft, r, ("plain, two streams" if split else "plain, one stream", space, f))
    n = ft.plain_call_counters()
    assert n["plain_calls"] >= 10 and n["stereo_chained"] > 0 and n["chained_redone"] == 0, n
    # (13 plain calls; no. 4 has no right events; no. 7 from pageable memory is below the staging threshold)
    assert n["split_by_camera"] == ((n["plain_calls"] - (2 if space in ("host", "registered") else 1)) if split else 0), n
    assert len(ft.ids) > 60
    ft.close()
    for b in bufs:
        b.free()


@pytest.mark.parametrize("pack", [1, 0])
def test_plain_calls_from_pageable_memory_cross_pcie_packed(oracle, monkeypatch, pack):
    """A plain call's batch in pageable host memory is staged by camera in 64 KiB chunks and pulled over PCIe by a
    kernel; since round 6 the staging threads pack each chunk to 8 bytes per event where its events allow it and the
    pull kernel unpacks it on the way into the device buffer (ESVIO_FE_STAGE_PACK=0: every chunk raw).  Batches with
    garbage in the records' padding bytes and polarity bytes other than 0 / 1 (what a deserialised ROS message may
    hold), one whose stamps cross a second boundary, one that steps BACK over one (those chunks travel raw): every
    frame as the oracle's, and the counters say which form the chunks took."""
    monkeypatch.setenv("ESVIO_FE_STAGE_PACK", str(pack))
    W, H = 640, 480
    rng = np.random.default_rng(23)
    s = SceneStream(W, H, rate=4e6, seed=29, n_rect=12, size=(30.0, 90.0), t0_us=1_700_000_000_950_000)  # (frame 1 crosses a second)
    batches = []
    for f in range(8):
        L, R, _ = s.next_batch()
        for a in (L, R):
            raw = a.view(np.uint8).reshape(-1, 16)
            raw[:, 13:] = rng.integers(0, 256, (len(a), 3), dtype=np.uint8)
            a["polarity"] = np.where(a["polarity"] != 0, rng.integers(1, 256, len(a)), 0)
        batches.append((L, R))
    # batch 5: the second half of the left array one second EARLIER (time reversal across a second boundary)
    L5 = batches[5][0]
    L5["sec"][len(L5) // 2:] -= 1
    kw = dict(max_cnt=200, min_dist=10, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f, (L, R) in enumerate(batches):
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, f % 2 == 0)
        _same(ft, tr.track_event(t, L, R, f % 2 == 0), ("packed" if pack else "raw", f))
        for cam in (0, 1):
            for a, b in zip(ft.detector.get_sae(cam), tr.detector().get_sae(cam)):
                assert np.array_equal(a, b), (f, cam)
    n = ft.staging_counters()
    assert n["batches"] == len(batches)
    if pack:
        assert n["chunks_packed"] > 20 * len(batches) and 1 <= n["chunks_raw"] <= 40, n
    else:
        assert n["chunks_packed"] == 0 and n["chunks_raw"] == 0, n
    ft.close()


def test_plain_call_whose_groups_have_more_chunks_than_the_descriptor_cache(oracle):
    """k_stage_pull_packed keeps a group's chunk descriptors in LDS — up to 1024 of them; a larger group reads them
    from the pinned buffer as it goes.  9.3 M events per camera and batch (1280x720, 280 Mev/s: 2.8 times C5's rate)
    make 1139 chunks per group: planes and results as the oracle's, every chunk packed."""
    W, H = 1280, 720
    s = PoissonStream(W, H, rate=2.8e8, seed=41)
    kw = dict(max_cnt=120, min_dist=20, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(2):
        L, R, _ = s.next_batch()
        assert len(L) * 16 // 65536 // 2 > 1024 and len(R) * 16 // 65536 // 2 > 1024
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, True)
        _same(ft, tr.track_event(t, L, R, True), ("large groups", f))
        for cam in (0, 1):
            for a, b in zip(ft.detector.get_sae(cam), tr.detector().get_sae(cam)):
                assert np.array_equal(a, b), (f, cam)
    n = ft.staging_counters()
    assert n["batches"] == 2 and n["chunks_packed"] > 8000 and n["chunks_raw"] == 0, n
    ft.close()
