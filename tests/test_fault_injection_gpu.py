"""Every device-side wait is bounded and has a give-up path; here each one is made to expire on demand
(esvio_fe_debug_inject / ESVIO_FE_FAULT: the wait's bound becomes 0, so a wave that would have to wait
at all gives up) and the documented consequence is checked:

* k_tile_apply's turn ticket, k_radix_pass's look-back: the call fails with ESVIO_FE_EINTERNAL (the
  planes are partially updated), esvio_fe_reset makes the handle usable again;
* the speculative temporal LK waiting for k_select's corners, the chained temporal LK waiting for the
  previous frame's launch: the host notices the flag and redoes the launch the plain way — every frame
  still equals the sequential oracle's, and the redo counters show that the path was taken."""
import numpy as np
import pytest

from esvio_amd import frontend as FE
from esvio_amd.events import event_times
from esvio_amd.synth import SceneStream

pytestmark = pytest.mark.gpu

KEYS = ("cur_pts", "cur_un_pts", "pts_velocity", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity")


def _same(ft, r, tag, off=0):
    assert np.array_equal(ft.ids, r.ids + off) and np.array_equal(ft.track_cnt, r.track_cnt), tag
    assert np.array_equal(ft.ids_right, r.ids_right + off), tag
    for k in KEYS:
        a, b = getattr(ft, k), getattr(r, k)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (tag, k)


@pytest.mark.parametrize("which", ["ticket", "lookback"])
def test_an_expired_wait_of_the_sae_update_fails_the_call_and_reset_recovers(oracle, monkeypatch, which):
    if which == "lookback":
        monkeypatch.setenv("ESVIO_FE_SAE_SORT", "1")  # the radix-sort form of the update
    W, H = 640, 480
    kw = dict(max_cnt=150, min_dist=10, f_ransac=1)
    s = SceneStream(W, H, rate=2e7, seed=6, n_rect=20, size=(40.0, 120.0))  # ~1100 events per tile: several turns
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    L, R, _ = s.next_batch()
    ft.trackEvent(event_times(L)[-1], L, R, True)  # (a healthy frame first)
    ft.debug_inject(FE.FAULT_TICKET if which == "ticket" else FE.FAULT_LOOKBACK)
    L, R, _ = s.next_batch()
    with pytest.raises(FE.FrontendError, match="rc=-5"):
        ft.trackEvent(event_times(L)[-1], L, R, True)
    with pytest.raises(FE.FrontendError, match="rc=-5"):  # the stand-alone entry point reports it as well
        ft.detector.createSAE_stereo(L, R)
    ft.debug_inject(0)
    ft.reset()
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for f in range(4):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, True)
        r = tr.track_event(t, L, R, True)
        off = int(ft.ids.min() - r.ids.min()) if len(r.ids) else 0  # (ids keep counting across a reset)
        _same(ft, r, (which, "after reset", f), off)
        assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
        assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1))
    assert len(ft.ids) > 60
    ft.close()


@pytest.mark.parametrize("mask,counter", [(FE.FAULT_SPECULATIVE, "spec_redone"), (FE.FAULT_CHAINED, "chain_redone"),
                                          (FE.FAULT_SPECULATIVE | FE.FAULT_CHAINED, "spec_redone")])
def test_an_expired_speculative_or_chained_lk_wait_is_redone_with_the_same_results(oracle, mask, counter):
    W, H = 640, 480
    kw = dict(max_cnt=200, min_dist=10, f_ransac=1)
    s = SceneStream(W, H, rate=4e6, seed=23, n_rect=24, size=(40.0, 120.0))
    batches = [s.next_batch()[:2] for _ in range(36)]
    pubs = [f % 2 == 0 for f in range(len(batches))]  # every other frame publishes nothing: chains are launched
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    announced = 0
    for f, (L, R) in enumerate(batches):
        if f == 6:
            base = ft.debug_counters()
            ft.debug_inject(mask)
        if f == 18:
            after = ft.debug_counters()
            ft.debug_inject(0)
        while announced < min(f + 3, len(batches) - 1):
            announced += 1
            Ln, Rn = batches[announced]
            ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, pubs[announced])
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        ft.finish()
        _same(ft, r, ("fault", mask, f))
    assert after[counter] > base[counter], (base, after)  # the give-up path really ran
    end = ft.debug_counters()
    assert end["chain_used"] > after["chain_used"]  # ... and normal service resumed
    assert len(ft.ids) > 80
    ft.close()


def test_an_expired_wait_of_the_plain_calls_chained_stereo_lk_is_redone(oracle):
    """A plain call (nothing announced, the drop-in pattern of INTEGRATION.md) launches the stereo LK of the
    kept points together with the temporal LK, every wave waiting on the device for its point's forward
    result; made to give up, the call redoes that launch from the temporal results it has read meanwhile:
    every frame still equals the sequential oracle's, published or not."""
    W, H = 640, 480
    kw = dict(max_cnt=200, min_dist=10, f_ransac=1)
    s = SceneStream(W, H, rate=4e6, seed=29, n_rect=24, size=(40.0, 120.0))
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    redone = []
    for f in range(14):
        if f == 4:
            ft.debug_inject(FE.FAULT_CHAINED)
        if f == 10:
            ft.debug_inject(0)
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        pub = f % 3 != 1
        ft.trackEvent(t, L, R, pub)
        _same(ft, tr.track_event(t, L, R, pub), ("plain chained stereo", f))
        redone.append(ft.debug_counters()["chain_redone"])
    assert redone[3] == 0 and redone[9] > redone[3] and redone[-1] == redone[9], redone
    assert len(ft.ids) > 80
    ft.close()


def test_fault_mask_from_the_environment(oracle, monkeypatch):
    """ESVIO_FE_FAULT=<mask> applies from esvio_fe_create on"""
    monkeypatch.setenv("ESVIO_FE_FAULT", str(FE.FAULT_TICKET))
    W, H = 346, 260
    s = SceneStream(W, H, rate=1e7, seed=2)
    ft = FE.FeatureTracker(FE.make_config(W, H))
    L, R, _ = s.next_batch()
    with pytest.raises(FE.FrontendError, match="rc=-5"):
        ft.detector.createSAE_stereo(L, R)
    ft.close()


@pytest.mark.parametrize("late", [False, True])
def test_lazy_completions_at_their_latest_point_and_packed_in_between(oracle, late):
    """Replay mode leaves two things for later: the right-camera entries of the corners a published frame has just
    detected, and the whole right-camera tail of a frame that publishes nothing.  WHEN they are completed depends on
    whether the stereo LK they wait for is over (round 6); results may only depend on their ORDER — the new corners
    first: they extend the map the next frame's right-camera velocities read.  FAULT_LAZY_LATE pushes every completion
    to the latest point it can ever take place at; esvio_fe_pack_track_records after EVERY call (published or not) and
    esvio_fe_finish now and then complete them from outside the track calls.  Every frame's vectors as the oracle's."""
    from esvio_amd.node import pack_track_records
    W, H = 346, 260
    rng = np.random.default_rng(17)
    s = SceneStream(W, H, rate=2.5e6, seed=43, n_rect=10, size=(25.0, 80.0))
    batches = [s.next_batch()[:2] for _ in range(40)]
    pubs = [bool(rng.integers(0, 3) != 0) for _ in batches]
    kw = dict(max_cnt=120, min_dist=10, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    ft.set_launch_thread(True)
    if late:
        ft.debug_inject(FE.FAULT_LAZY_LATE)
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
        assert np.array_equal(ft.ids, r.ids) and np.array_equal(ft.cur_pts.view(np.uint32), r.cur_pts.view(np.uint32)), f
        mode = int(rng.integers(0, 3))
        if mode == 0:  # (completes an unpublished frame's tail from outside; a published frame's rows need nothing)
            rows = ft.pack_track_records()
            want = _oracle_rows(r, 120)
            assert np.array_equal(rows.view(np.uint32), want.view(np.uint32)), ("rows", f)
        elif mode == 1:
            ft.finish()
            for k in ("ids_right", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity"):
                a, b = getattr(ft, k), getattr(r, k)
                assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (f, k)
    ft.finish()
    for k in ("ids_right", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity"):
        a, b = getattr(ft, k), getattr(r, k)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("end", k)
    assert len(ft.ids) > 40
    ft.close()


def _oracle_rows(r, max_cnt):
    """the node's PointCloud rows (stereo_event_tracker_node.cpp:273-329) of an oracle result"""
    from esvio_amd.node import pack_track_records

    class T:
        pass
    t = T()
    for k in ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity", "ids_right", "cur_right_pts", "cur_un_right_pts",
              "right_pts_velocity"):
        setattr(t, k, getattr(r, k))
    return pack_track_records(t, max_cnt)
