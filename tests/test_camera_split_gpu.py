"""Camera split over two ranks (BASELINE C4) equals the single-handle result bit for bit.
Runs two gloo ranks that both use cuda:0 (the GPU box has one GPU); on a real 2-GPU node the same
code runs with backend nccl and one device per rank."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity", "ids_right", "cur_right_pts",
        "cur_un_right_pts", "right_pts_velocity")
W, H, NB = 346, 260, 6


def _stream():
    from esvio_amd.synth import SceneStream
    s = SceneStream(W, H, rate=1e6, seed=9, n_rect=12, size=(30.0, 90.0))
    return [s.next_batch() for _ in range(NB)]


def _worker(rank, port, q):
    import torch.distributed as dist
    from esvio_amd import frontend as FE
    from esvio_amd.dist import CameraSplitRig
    from esvio_amd.events import event_times
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        ft = FE.FeatureTracker(FE.make_config(W, H, device=0, max_cnt=150))
        rig = CameraSplitRig(ft, rank, dist, device="cpu")
        out = []
        for b, (L, R, _) in enumerate(_stream()):
            t = event_times(L)[-1]
            r = rig.track(t, L, R, b % 3 != 2)
            if rank == 0:
                out.append({k: getattr(r, k).copy() for k in KEYS})
        dist.barrier()
        q.put((rank, out))
        ft.close()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_camera_split_matches_single_gpu():
    import torch.multiprocessing as mp
    from esvio_amd import frontend as FE
    from esvio_amd.events import event_times
    ft = FE.FeatureTracker(FE.make_config(W, H, device=0, max_cnt=150))
    ref = []
    for b, (L, R, _) in enumerate(_stream()):
        ft.trackEvent(event_times(L)[-1], L, R, b % 3 != 2)
        ref.append({k: getattr(ft, k).copy() for k in KEYS})
    ft.close()
    assert len(ref[-1]["ids"]) > 30 and len(ref[-1]["ids_right"]) > 20

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert isinstance(res[0], list), res[0]
    assert isinstance(res[1], list), res[1]
    for b in range(NB):
        for k in KEYS:
            assert np.array_equal(res[0][b][k], ref[b][k]), (b, k)
