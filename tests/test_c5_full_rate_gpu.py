"""BASELINE C5 at its own workload: 1280x720, 100 Mev/s per camera in 30 Hz batches (~3.3 M events
per camera per batch, ~11 events per touched pixel), through the paths a batch of that size takes
by itself (no ESVIO_FE_* switches): SAE planes, time surfaces, Arc* flags and the greedy selection
against the oracle per stage, then trackEvent end to end (plain and replay schedule) at max_cnt 500."""
import numpy as np
import pytest

from esvio_amd import frontend as FE
from esvio_amd.events import event_times
from esvio_amd.synth import SceneStream

pytestmark = pytest.mark.gpu

W, H = 1280, 720


def _stream(n, seed=21):
    s = SceneStream(W, H, rate=100e6, seed=seed, n_rect=40, size=(60.0, 220.0))
    return [s.next_batch()[:2] for _ in range(n)]


def test_c5_stages_at_full_rate(oracle):
    batches = _stream(3)
    ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=500, min_dist=10))
    det = oracle.Detector(W, H)
    n_corners = 0
    for b, (L, R) in enumerate(batches):
        assert len(L) >= 3_000_000 and len(R) >= 3_000_000  # >= 2^20 per submission by a wide margin
        assert ft.detector.createSAE_stereo(L, R) == 0
        det.create_sae(0, L)
        det.create_sae(1, R)
        for cam in (0, 1):
            for x, y, name in zip(ft.detector.get_sae(cam), det.get_sae(cam), ("L0", "L1", "S0", "S1")):
                assert np.array_equal(x, y), (b, cam, name, int((x != y).sum()))
        t = event_times(L)[-1]
        ts_cpu = det.time_surface(0, t)
        assert np.array_equal(ft.detector.SAEtoTimeSurface_left(t), ts_cpu)
        assert np.array_equal(ft.detector.SAEtoTimeSurface_right(t), det.time_surface(1, t))
        fg, fc = ft.detector.isCorner(L), det.corner_flags(L)
        assert np.array_equal(fg, fc), "%d flag mismatches" % int((fg != fc).sum())
        n_corners += int(fc.sum())
        mask = np.zeros((H, W), np.uint8)
        oracle.circle_fill(mask, 640, 360, 10)
        for maxc in (1, 500):
            xy_g, idx_g = ft.Event_FeaturesToTrack(L, maxc, mask)
            xy_c, idx_c = det.features_to_track(L, maxc, 10, mask, ts_cpu)
            assert np.array_equal(idx_g, idx_c) and np.array_equal(xy_g, xy_c), (b, maxc)
    assert n_corners > 100_000
    ft.close()


@pytest.mark.parametrize("replay", [False, True])
def test_c5_track_event_at_full_rate(oracle, replay):
    batches = _stream(4, seed=22)
    pubs = [True, False, True, True]
    kw = dict(max_cnt=500, min_dist=10, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    if replay:
        ft.set_lazy_new_stereo(True)
        ft.set_host_threads(3)
    announced = 0
    for f, (L, R) in enumerate(batches):
        if replay:
            while announced < min(f + 3, len(batches) - 1):
                announced += 1
                Ln, Rn = batches[announced]
                ft.set_next_batch(event_times(Ln)[-1], Ln, Rn, pubs[announced])
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, pubs[f])
        r = tr.track_event(t, L, R, pubs[f])
        if replay:
            ft.finish()
        assert np.array_equal(ft.ids, r.ids), f
        assert np.array_equal(ft.track_cnt, r.track_cnt) and np.array_equal(ft.ids_right, r.ids_right), f
        for k in ("cur_pts", "cur_un_pts", "pts_velocity", "cur_right_pts", "cur_un_right_pts",
                  "right_pts_velocity"):
            a, b = getattr(ft, k), getattr(r, k)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (f, k)
    assert np.array_equal(ft.gettimesurface(0), tr.time_surface(0))
    assert np.array_equal(ft.gettimesurface(1), tr.time_surface(1))
    assert len(ft.ids) > 200 and len(ft.ids_right) > 100
    ft.close()


def test_c5_named_split_eight_time_slices(oracle):
    """BASELINE C5's own split at its own size: ONE 1280x720 stream at 100 Mev/s per camera (3.3 M events
    per camera per batch), every batch cut into 8 time slices, one per handle (8 handles in one process
    — this box has one GPU — exchanging DEVICE plane sets, what 8 ranks all-gather over RCCL):
    esvio_fe_sae_slice_last / _apply / _commit; every handle's planes after every batch, and the tracks
    handle 0 derives from them, equal the oracle's"""
    import ctypes as C
    from esvio_amd.dist import time_slice
    N = 8
    batches = _stream(2, seed=23)
    kw = dict(max_cnt=500, min_dist=10, f_ransac=1)
    fts = [FE.FeatureTracker(FE.make_config(W, H, **kw)) for _ in range(N)]
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    nd = fts[0].sae_plane_doubles()
    assert nd == 4 * W * H
    lib = FE.load_library()
    bufs = []
    for _ in range(2):  # all slices' "last" sets, all slices' "S" sets
        p = C.c_void_p()
        assert lib.esvio_fe_mem_alloc(FE.DEVICE, N * nd * 8, C.byref(p)) == 0
        bufs.append(p)
    last_all, s_all = bufs[0].value, bufs[1].value
    for b, (L, R) in enumerate(batches):
        assert len(L) >= 3_000_000 and len(R) >= 3_000_000
        pub = b == 0
        t = event_times(L)[-1]
        r = tr.track_event(t, L, R, pub)
        det = tr.detector()
        cuts = [(L[slice(*time_slice(len(L), N, k))], R[slice(*time_slice(len(R), N, k))]) for k in range(N)]
        for k, ft in enumerate(fts):
            ft.sae_slice_last(cuts[k][0], cuts[k][1], (last_all + 8 * k * nd, 1))
        for k, ft in enumerate(fts):
            ft.sae_slice_apply(cuts[k][0], cuts[k][1], (last_all, k), k, (s_all + 8 * k * nd, 1))
        for k, ft in enumerate(fts):
            ft.sae_slice_commit((last_all, N), (s_all, N), N)
            for cam in (0, 1):
                for x, y, name in zip(ft.detector.get_sae(cam), det.get_sae(cam), ("L0", "L1", "S0", "S1")):
                    assert np.array_equal(x, y), (b, k, cam, name, int((x != y).sum()))
        # handle 0 (the rank the host feeds) runs the rest of trackEvent on the composed planes
        fts[0].trackEvent(t, L, R, pub)
        assert np.array_equal(fts[0].ids, r.ids) and np.array_equal(fts[0].ids_right, r.ids_right), b
        assert np.array_equal(fts[0].cur_pts.view(np.uint32), r.cur_pts.view(np.uint32)), b
        assert np.array_equal(fts[0].gettimesurface(0), tr.time_surface(0))
        # (the other handles took the commit as their update of the planes: drop the one-shot marker the
        # way a rank that never tracks does — its next slice_last starts from the committed planes)
    assert len(fts[0].ids) > 200
    for ft in fts:
        ft.close()
    for p in bufs:
        lib.esvio_fe_mem_free(FE.DEVICE, p)
