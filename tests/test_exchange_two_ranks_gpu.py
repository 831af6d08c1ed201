"""The library's own track exchange (esvio_fe_comm_init with world 2 -> ncclAllGather on the handle's
communicator, RCCL over xGMI) against the torch.distributed mirror (TrackExchange) and against the two
ranks' packed records themselves: two processes, one GPU each, each tracking its own rig.  RCCL refuses
two ranks on one device, so this needs a box with at least two GPUs and is skipped on a one-GPU box
(where tests/slice_device_worker.py exercises the same entry points with a one-rank communicator)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NB = 7
MAX_CNT = 150


def _n_gpus():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def _worker(rank, port, q):
    import torch
    import torch.distributed as dist
    from esvio_amd import frontend as FE
    from esvio_amd.dist import TrackExchange
    from esvio_amd.events import event_times
    from esvio_amd.synth import SceneStream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", rank))
    try:
        W, H = 346, 260
        ft = FE.FeatureTracker(FE.make_config(W, H, device=rank, max_cnt=MAX_CNT, min_dist=10, f_ransac=1))
        box = [FE.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ft.comm_init(box[0], rank, 2)
        ex = TrackExchange(MAX_CNT, 2, device="cuda", dist=dist, stream=torch.cuda.Stream())
        s = SceneStream(W, H, rate=1e6, seed=40 + rank, n_rect=12, size=(30.0, 90.0))
        out = []
        for b in range(NB):
            L, R, _ = s.next_batch()
            pub = b % 2 == 0  # (every rank publishes the same frames: timestamps only)
            ft.trackEvent(event_times(L)[-1], L, R, pub)
            if not pub:
                continue
            mine = ft.pack_track_records().copy()
            ft.exchange_begin()
            lib = ft.exchange_end()
            ex.submit_tracker(ft, async_op=True)
            mirror = ex.result().copy()
            out.append((mine, lib.copy(), mirror))
        # the automatic form: the records of the last published frame, enqueued by the following call
        ft.set_auto_exchange(True)
        last = None
        for b in range(NB, NB + 3):
            L, R, _ = s.next_batch()
            pub = b % 2 == 0
            ft.trackEvent(event_times(L)[-1], L, R, pub)
            if pub:
                last = ft.pack_track_records().copy()
        auto = ft.exchange_end().copy()
        dist.barrier()
        q.put((rank, (out, last, auto)))
        ft.close()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_library_exchange_of_two_ranks_matches_the_torch_mirror():
    if _n_gpus() < 2:
        pytest.skip("RCCL with two ranks needs two GPUs")
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(2):
        assert isinstance(res[r], tuple), res[r]
    (out0, last0, auto0), (out1, last1, auto1) = res[0], res[1]
    assert len(out0) == len(out1) > 2
    for (m0, l0, t0), (m1, l1, t1) in zip(out0, out1):
        want = np.stack([m0, m1])
        for got in (l0, l1, t0, t1):  # every rank sees both ranks' rows, from either exchange
            assert got.shape == (2, 2 * MAX_CNT, 8)
            assert np.array_equal(got, want)
    assert (out0[-1][0][:, 3] >= 0).sum() > 10  # (real tracks, not padding rows)
    want = np.stack([last0, last1])
    assert np.array_equal(auto0, want) and np.array_equal(auto1, want)
