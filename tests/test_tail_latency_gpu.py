"""Tail latency of the replay schedule as a tested quantity (round-3 review: the driver-timed pass had
steps of 2-7 ms among 0.1 ms ones).  The two causes found (profiles/r04_stall_forensics.md) — a candidate
set's first use putting seven hipMallocs into some later call, and the calling thread waiting for a RANSAC
helper that had lost its CPU — are both held here: no allocation after esvio_fe_reserve, and over 1000 replay
steps no step slower than 5x the median."""
import time

import numpy as np
import pytest

from esvio_amd import frontend as FE
from esvio_amd.events import event_times
from esvio_amd.node import FreqControl
from esvio_amd.synth import SceneStream

pytestmark = pytest.mark.gpu

W, H = 640, 480


def _stream(n_distinct, cycles, rate, seed):
    """n_distinct batches of a scene stream, repeated `cycles` times with the stamps moved on by the
    stream's length each time (the scene jumps back once per cycle: a discontinuity, as after a bag loop)"""
    s = SceneStream(W, H, rate=rate, seed=seed)
    base = [s.next_batch()[:2] for _ in range(n_distinct)]
    span = int(np.ceil(event_times(base[-1][0])[-1] - event_times(base[0][0])[0])) + 1
    out = []
    for k in range(cycles):
        for L, R in base:
            L2, R2 = L.copy(), R.copy()
            L2["sec"] += k * span
            R2["sec"] += k * span
            out.append((L2, R2))
    return out


def _run(batches, host_threads, steps, warm, launch=False):
    bufs, dev = [], []
    for L, R in batches:
        bl, br = FE.EventBuffer(L, FE.DEVICE), FE.EventBuffer(R, FE.DEVICE)
        bufs += [bl, br]
        dev.append((bl.arg, br.arg, event_times(L)[-1]))
    fc = FreqControl(15)
    pubs = []
    for b in dev:
        pubs.append(fc.pub_this_frame(b[2]))
        if pubs[-1]:
            fc.published()
    ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=300, min_dist=10, f_ransac=1))
    ft.set_lazy_new_stereo(True)
    if host_threads > 1:
        ft.set_host_threads(host_threads)
    if launch:
        ft.set_launch_thread(True)
    ft.reserve(max(len(b[0]) for b in batches), max(len(b[1]) for b in batches))
    allocs_after_reserve = []
    announced = 0
    times = []
    FE.ransac_tail(reset=True)
    for i in range(warm + steps):
        if i == warm:
            allocs_after_reserve.append(ft.latency_stats(reset=True)["allocs"])
        t0 = time.perf_counter()
        while announced < min(i + 3, len(dev) - 1):
            announced += 1
            b = dev[announced]
            ft.set_next_batch(b[2], b[0], b[1], pubs[announced])
        b = dev[i]
        ft.trackEvent(b[2], b[0], b[1], pubs[i], copy=False)
        times.append(time.perf_counter() - t0)
    ft.finish(copy=False)
    lat = ft.latency_stats()
    # esvio_fe_latency_recent: the latest calls one by one — consecutive indices, one after the other in time, the
    # phases of a call inside its wall time, publish flags as called; the slowest of them no slower than the record's
    rec = ft.latency_recent(64)
    assert len(rec) == 64 and [r[0] for r in rec] == list(range(steps - 64, steps))  # (indices since the reset at `warm`)
    assert [r[1] for r in rec] == [bool(p) for p in pubs[warm + steps - 64:warm + steps]]
    for a, b in zip(rec, rec[1:]):
        assert a[2] + a[3] <= b[2] + 1e-6
    for r in rec:
        top = {k: v for k, v in r[4].items() if not k.startswith("pub:") and not k.startswith("sae:")}
        assert 0.0 < r[3] and sum(top.values()) <= r[3] + 1e-3, r
    assert max(r[3] for r in rec) <= lat["max_ms"] + 1e-9
    assert ft.latency_recent(300)[0][0] == steps - 256  # (256 are kept)
    allocs_after_reserve.append(lat["allocs"])
    tail = FE.ransac_tail()
    n_tracks = len(ft.ids)
    ft.close()
    for b in bufs:
        b.free()
    return np.array(times[warm:]) * 1e3, lat, tail, allocs_after_reserve, n_tracks


def test_thousand_replay_steps_without_a_latency_tail():
    steps, warm = 1000, 40
    batches = _stream(130, 8, 2.5e6, 4)
    assert len(batches) >= steps + warm
    report = []
    for attempt in range(2):  # (one repeat: a step can still lose its CPU to something else on the box)
        ms, lat, tail, allocs, n_tracks = _run(batches, 8, steps, warm)
        med, mx = float(np.median(ms)), float(ms.max())
        report.append(dict(median_ms=round(med, 4), p99_ms=round(float(np.percentile(ms, 99)), 4), max_ms=round(mx, 4),
                           argmax=int(ms.argmax()), library=lat, ransac_tail=tail, allocs=allocs))
        print("tail latency, attempt %d: %s" % (attempt, report[-1]))
        assert allocs == [0, 0], "allocations after esvio_fe_reserve: %s" % allocs
        assert n_tracks > 50
        if mx <= 5.0 * med:
            break
    else:
        pytest.fail("a step slower than 5x the median in both attempts: %s" % report)
    # the library's own record covers the same calls
    assert lat["calls"] == steps and lat["max_ms"] <= mx + 1e-3
    # the same with the prefetch launches issued by the handle's launch thread (esvio_fe_set_launch_thread)
    for attempt in range(2):
        ms, lat, tail, allocs, n_tracks = _run(batches, 8, 400, warm, launch=True)
        med, mx = float(np.median(ms)), float(ms.max())
        print("tail latency with the launch thread, attempt %d: median %.4f p99 %.4f max %.4f ms (call %d); %s"
              % (attempt, med, float(np.percentile(ms, 99)), mx, int(ms.argmax()), lat["max_phase_ms"]))
        assert allocs == [0, 0] and n_tracks > 50
        if mx <= 5.0 * med:
            break
    else:
        pytest.fail("launch thread: a step slower than 5x the median in both attempts")


def test_no_allocation_after_reserve_and_a_first_call_like_the_others():
    """esvio_fe_create has loaded the code object and woken every stream, esvio_fe_reserve has sized the
    buffers: the first track calls allocate nothing, and host batches (staging slots) neither"""
    s = SceneStream(W, H, rate=4e6, seed=9)
    batches = [s.next_batch()[:2] for _ in range(12)]
    for host in (False, True):
        ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=200, min_dist=10, f_ransac=1))
        ft.reserve(max(len(b[0]) for b in batches), max(len(b[1]) for b in batches), host_batches=host)
        ft.latency_stats(reset=True)
        bufs = []
        for f, (L, R) in enumerate(batches):
            if host:
                aL, aR = L, R
            else:
                bl, br = FE.EventBuffer(L, FE.DEVICE), FE.EventBuffer(R, FE.DEVICE)
                bufs += [bl, br]
                aL, aR = bl.arg, br.arg
            ft.trackEvent(event_times(L)[-1], aL, aR, f % 2 == 0)
        lat = ft.latency_stats()
        assert lat["calls"] == len(batches)
        assert lat["allocs"] == 0, (host, lat)
        ft.close()
        for b in bufs:
            b.free()
    # without the reserve the same calls do allocate (the counter counts)
    ft = FE.FeatureTracker(FE.make_config(W, H, max_cnt=200, min_dist=10, f_ransac=1))
    L, R = batches[0]
    ft.trackEvent(event_times(L)[-1], L, R, True)
    assert ft.latency_stats()["allocs"] > 0
    ft.close()


@pytest.mark.parametrize("max_cnt", [600, 1000])
def test_replay_with_more_points_than_waiting_launches_fit(max_cnt):
    """An LK launch whose waves wait on the device (the speculative launch for k_select's corners, a chained launch for
    its producer's points) holds one CU per four points while it waits; with max_cnt 1000 a launch is 250 workgroups on
    256 CUs, and one step in ~20 of the replay schedule ran into a wait's 20-40 ms bound and was redone (round 6,
    bench.py --max-cnt 1000: passes of 1.2-3.3 ms per step).  The handle now makes such launches only while they leave
    room (esvio_fe_ctx::waits_fit_*): 100 replay steps with the launch thread, none slower than 5 ms, results as the
    oracle's, no wait expired; the speculative launch still runs (250 workgroups leave room for k_select_mw), the chained
    one does not."""
    from oracle import oracle as O
    O.build()
    s = SceneStream(W, H, rate=5e6, seed=77)
    batches = [s.next_batch()[:2] for _ in range(100)]
    bufs, dev = [], []
    for L, R in batches:
        bl, br = FE.EventBuffer(L, FE.DEVICE), FE.EventBuffer(R, FE.DEVICE)
        bufs += [bl, br]
        dev.append((bl.arg, br.arg, event_times(L)[-1]))
    fc = FreqControl(15)
    pubs = []
    for b in dev:
        pubs.append(fc.pub_this_frame(b[2]))
        if pubs[-1]:
            fc.published()
    kw = dict(max_cnt=max_cnt, min_dist=10, f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    ft.set_lazy_new_stereo(True)
    ft.set_host_threads(8)
    ft.set_launch_thread(True)
    ft.reserve(max(len(b[0]) for b in batches), max(len(b[1]) for b in batches))
    tr = O.Tracker(O.make_config(W, H, **kw))
    announced = 0
    slow = []
    for i, b in enumerate(dev):
        while announced < min(i + 3, len(dev) - 1):
            announced += 1
            a = dev[announced]
            ft.set_next_batch(a[2], a[0], a[1], pubs[announced])
        t0 = time.perf_counter()
        ft.trackEvent(b[2], b[0], b[1], pubs[i])
        ms = (time.perf_counter() - t0) * 1e3
        if i >= 5 and ms > 5.0:
            slow.append((i, round(ms, 2)))
        r = tr.track_event(b[2], batches[i][0], batches[i][1], pubs[i])
        assert np.array_equal(ft.ids, r.ids) and np.array_equal(ft.cur_pts.view(np.uint32), r.cur_pts.view(np.uint32)), i
        if i % 10 == 9:  # (the right camera's results of a lazily returned frame: complete them now and then)
            ft.finish()
            assert np.array_equal(ft.ids_right, r.ids_right), i
            assert np.array_equal(ft.cur_right_pts.view(np.uint32), r.cur_right_pts.view(np.uint32)), i
    ft.finish()
    cnt = ft.debug_counters()
    ft.close()
    for b in bufs:
        b.free()
    assert not slow, slow
    assert cnt["spec_redone"] == 0 and cnt["chain_redone"] == 0, cnt
    assert cnt["chain_launched"] == 0, cnt  # (two waiting launches of 150 / 250 workgroups each do not fit 256 CUs)
