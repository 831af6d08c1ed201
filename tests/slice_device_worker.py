"""Run by tests/test_time_slice_gpu.py in a fresh interpreter (torch first, then the library): the
time-slice entry points with DEVICE plane sets (torch CUDA tensors) and the exchange over RCCL
(backend nccl, a process group of ONE rank — this box has one GPU): two handles in one process take
the two slices of every batch by hand (device buffers), a TimeSlicedSae of world size 1 runs the
collective path; planes against the oracle.  Prints OK."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    from esvio_amd import frontend as FE
    from esvio_amd.dist import TimeSlicedSae, time_slice
    from esvio_amd.events import event_times
    from oracle import oracle as O
    from slice_engine import adversarial_batches

    W, H = 346, 260
    # ---- two handles, device-side plane sets, slices by hand
    fts = [FE.FeatureTracker(FE.make_config(W, H, device=0)) for _ in range(2)]
    det = O.Detector(W, H)
    nd = fts[0].sae_plane_doubles()
    last_all = torch.empty(2 * nd, dtype=torch.float64, device="cuda")
    s_all = torch.empty(2 * nd, dtype=torch.float64, device="cuda")
    for b, (L, R) in enumerate(adversarial_batches(W, H, 3, seed=9, n=30000)):
        det.create_sae(0, L)
        det.create_sae(1, R)
        cuts = [(L[slice(*time_slice(len(L), 2, r))], R[slice(*time_slice(len(R), 2, r))]) for r in range(2)]
        for r, ft in enumerate(fts):
            ft.sae_slice_last(cuts[r][0], cuts[r][1], (last_all.data_ptr() + 8 * r * nd, 1))
        for r, ft in enumerate(fts):
            ft.sae_slice_apply(cuts[r][0], cuts[r][1], (last_all.data_ptr(), r), r, (s_all.data_ptr() + 8 * r * nd, 1))
        for ft in fts:
            ft.sae_slice_commit((last_all.data_ptr(), 2), (s_all.data_ptr(), 2), 2)
            for cam in (0, 1):
                for x, y in zip(ft.detector.get_sae(cam), det.get_sae(cam)):
                    assert np.array_equal(x, y), ("device planes", b, cam)
    for ft in fts:
        ft.close()
    # ---- the collective path over RCCL, world size 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ft = FE.FeatureTracker(FE.make_config(W, H, device=0, max_cnt=150))
    tr = O.Tracker(O.make_config(W, H, max_cnt=150))
    ts = TimeSlicedSae(ft, 0, 1, dist, device="cuda")
    from esvio_amd.synth import SceneStream
    s = SceneStream(W, H, rate=1e6, seed=4, n_rect=12, size=(30.0, 90.0))
    for b in range(4):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        ts.track(t, L, R, b % 2 == 0)
        r = tr.track_event(t, L, R, b % 2 == 0)
        assert np.array_equal(ft.ids, r.ids) and np.array_equal(ft.cur_pts, r.cur_pts), ("tracks", b)
        assert np.array_equal(ft.ids_right, r.ids_right), ("right", b)
    # ... and the track exchange of the Python mirror over the same communicator
    from esvio_amd.dist import TrackExchange
    ex = TrackExchange(150, 1, device="cuda", dist=dist, stream=torch.cuda.Stream())
    ex.submit_tracker(ft, async_op=True)
    g = ex.result()
    assert g.shape == (1, 300, 8) and np.array_equal(g[0], ft.pack_track_records())
    # ... and the library's own communicator (esvio_fe_comm_init / exchange_begin / exchange_end)
    ft.comm_init(FE.comm_unique_id(), 0, 1)
    for _ in range(3):
        ft.exchange_begin()
    g2 = ft.exchange_end()
    assert g2.shape == (1, 300, 8) and np.array_equal(g2[0], ft.pack_track_records())
    # ... automatic: every published frame, enqueued by the following call
    ft.set_auto_exchange(True)
    want = None
    for b in range(4, 9):
        L, R, _ = s.next_batch()
        t = event_times(L)[-1]
        pub = b % 2 == 0
        ft.trackEvent(t, L, R, pub)
        tr.track_event(t, L, R, pub)
        if pub:
            want = ft.pack_track_records().copy()
    g3 = ft.exchange_end()
    assert np.array_equal(g3[0], want)
    ft.close()
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()
