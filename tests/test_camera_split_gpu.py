"""Camera split over two ranks (BASELINE C4) equals the single-handle result — and the oracle's — bit
for bit, at C4's own workload (stereo 640x480, 5 Mev/s per camera, C3's parameters) and at the
DAVIS346 shape.  Runs two gloo ranks that both use cuda:0 when the box has one GPU; with two GPUs
visible the ranks use one device each and RCCL (backend nccl), which is how bench.py --split camera
runs it on a multi-GPU node."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity", "ids_right", "cur_right_pts",
        "cur_un_right_pts", "right_pts_velocity")
NB = 6
CASES = {
    "davis346": dict(W=346, H=260, rate=1e6, seed=9, n_rect=12, size=(30.0, 90.0), max_cnt=150, min_dist=10),
    # BASELINE C4 = C3's stream and parameters
    "c4_640x480": dict(W=640, H=480, rate=5e6, seed=12345, n_rect=28, size=(50.0, 150.0), max_cnt=300,
                       min_dist=10),
}


def _stream(case):
    from esvio_amd.synth import SceneStream
    c = CASES[case]
    s = SceneStream(c["W"], c["H"], rate=c["rate"], seed=c["seed"], n_rect=c["n_rect"], size=c["size"])
    return [s.next_batch() for _ in range(NB)]


def _n_gpus():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def _worker(rank, port, q, case, two_gpus):
    import torch.distributed as dist
    from esvio_amd import frontend as FE
    from esvio_amd.dist import CameraSplitRig
    from esvio_amd.events import event_times
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    c = CASES[case]
    dev = rank if two_gpus else 0
    if two_gpus:
        import torch
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        ft = FE.FeatureTracker(FE.make_config(c["W"], c["H"], device=dev, max_cnt=c["max_cnt"],
                                              min_dist=c["min_dist"], f_ransac=1))
        rig = CameraSplitRig(ft, rank, dist, device="cuda" if two_gpus else "cpu")
        out = []
        for b, (L, R, _) in enumerate(_stream(case)):
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


@pytest.mark.parametrize("case", list(CASES))
def test_camera_split_matches_single_gpu(oracle, case):
    import torch.multiprocessing as mp
    from esvio_amd import frontend as FE
    from esvio_amd.events import event_times
    c = CASES[case]
    kw = dict(max_cnt=c["max_cnt"], min_dist=c["min_dist"], f_ransac=1)
    ft = FE.FeatureTracker(FE.make_config(c["W"], c["H"], device=0, **kw))
    tr = oracle.Tracker(oracle.make_config(c["W"], c["H"], **kw))
    ref = []
    for b, (L, R, _) in enumerate(_stream(case)):
        t = event_times(L)[-1]
        ft.trackEvent(t, L, R, b % 3 != 2)
        ref.append({k: getattr(ft, k).copy() for k in KEYS})
        r = tr.track_event(t, L, R, b % 3 != 2)
        for k in KEYS:  # the single-handle run itself is the oracle's, bit for bit
            assert np.array_equal(ref[-1][k], getattr(r, k)), (b, k)
    ft.close()
    assert len(ref[-1]["ids"]) > 30 and len(ref[-1]["ids_right"]) > 20

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    two_gpus = _n_gpus() >= 2
    procs = [ctx.Process(target=_worker, args=(r, port, q, case, two_gpus)) for r in range(2)]
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
