"""Time-sliced SAE update on the GPU (BASELINE C5's split): N handles each apply
one time slice of every batch through esvio_fe_sae_slice_last / _apply / _commit; planes and the
tracks rank 0 derives from them equal the single-handle run and the oracle bit for bit."""
import os
import socket

import numpy as np
import pytest

from slice_engine import adversarial_batches

pytestmark = pytest.mark.gpu

KEYS = ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity", "ids_right", "cur_right_pts",
        "cur_un_right_pts", "right_pts_velocity")


def _planes_equal(ft, det, tag):
    for cam in (0, 1):
        for x, y, name in zip(ft.detector.get_sae(cam), det.get_sae(cam), ("L0", "L1", "S0", "S1")):
            assert np.array_equal(x, y), (tag, cam, name, int((x != y).sum()))


@pytest.mark.parametrize("world", [2, 4])
def test_slice_entry_points_compose_in_one_process(oracle, world):
    """the three entry points driven by hand for N handles in one process (host plane buffers):
    adversarial per-pixel histories across the cuts, three batches"""
    from esvio_amd import frontend as FE
    from esvio_amd.dist import time_slice
    W, H = 346, 260
    fts = [FE.FeatureTracker(FE.make_config(W, H)) for _ in range(world)]
    det = oracle.Detector(W, H)
    nd = fts[0].sae_plane_doubles()
    assert nd == 4 * W * H
    for b, (L, R) in enumerate(adversarial_batches(W, H, 3, seed=5, n=40000)):
        det.create_sae(0, L)
        det.create_sae(1, R)
        cuts = [(L[slice(*time_slice(len(L), world, r))], R[slice(*time_slice(len(R), world, r))])
                for r in range(world)]
        last_all, s_all = np.empty(world * nd), np.empty(world * nd)
        for r, ft in enumerate(fts):
            ft.sae_slice_last(cuts[r][0], cuts[r][1], last_all[r * nd:(r + 1) * nd])
        for r, ft in enumerate(fts):
            ft.sae_slice_apply(cuts[r][0], cuts[r][1], last_all, r, s_all[r * nd:(r + 1) * nd])
        for r, ft in enumerate(fts):
            ft.sae_slice_commit(last_all, s_all, world)
            _planes_equal(ft, det, (b, r))
    for ft in fts:
        ft.close()


W2, H2, NB = 640, 480, 5


def _stream():
    from esvio_amd.synth import SceneStream
    s = SceneStream(W2, H2, rate=5e6, seed=31)
    return [s.next_batch()[:2] for _ in range(NB)]


def _worker(rank, port, q, two_gpus):
    import torch
    import torch.distributed as dist
    from esvio_amd import frontend as FE
    from esvio_amd.dist import TimeSlicedSae
    from esvio_amd.events import event_times
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank if two_gpus else 0
    if two_gpus:
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        ft = FE.FeatureTracker(FE.make_config(W2, H2, device=dev, max_cnt=300, min_dist=10, f_ransac=1))
        ts = TimeSlicedSae(ft, rank, 2, dist, device="cuda" if two_gpus else "cpu")
        out = []
        for b, (L, R) in enumerate(_stream()):
            r = ts.track(event_times(L)[-1], L, R, b % 3 != 2)
            planes = [np.stack(ft.detector.get_sae(cam)) for cam in (0, 1)]
            out.append((planes, {k: getattr(r, k).copy() for k in KEYS} if rank == 0 else None))
        dist.barrier()
        q.put((rank, out))
        ft.close()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_time_sliced_stream_two_ranks(oracle):
    """TimeSlicedSae over two ranks (gloo, both on cuda:0; nccl with one device each when two GPUs
    are visible) at C3's workload: every rank's planes after every batch, and rank 0's tracks, equal
    the oracle's"""
    import ctypes
    import torch.multiprocessing as mp
    from esvio_amd.events import event_times
    tr = oracle.Tracker(oracle.make_config(W2, H2, max_cnt=300, min_dist=10, f_ransac=1))
    ref = []
    for b, (L, R) in enumerate(_stream()):
        r = tr.track_event(event_times(L)[-1], L, R, b % 3 != 2)
        d = tr.detector()
        ref.append(([np.stack(d.get_sae(cam)) for cam in (0, 1)], {k: getattr(r, k).copy() for k in KEYS}))
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    two_gpus = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q, two_gpus)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank in (0, 1):
        assert isinstance(res[rank], list), res[rank]
        for b in range(NB):
            for cam in (0, 1):
                assert np.array_equal(res[rank][b][0][cam], ref[b][0][cam]), (rank, b, cam)
    for b in range(NB):
        for k in KEYS:
            assert np.array_equal(res[0][b][1][k], ref[b][1][k]), (b, k)
    assert len(ref[-1][1]["ids"]) > 100


def test_device_plane_sets_and_rccl_world1():
    """DEVICE-space plane sets (torch CUDA tensors) through the three entry points, and
    TimeSlicedSae / TrackExchange over RCCL with a process group of one rank, in a fresh interpreter
    (torch initialises the GPU before the library does)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "slice_device_worker.py")],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
