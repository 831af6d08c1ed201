"""Time-sliced SAE update: composition rule and esvio_amd.dist.TimeSlicedSae on CPU,
with the oracle standing in for the handles (tests/slice_engine.py).  The GPU twin is
tests/test_time_slice_gpu.py."""
import os
import socket

import numpy as np
import pytest

from slice_engine import OracleSliceEngine, adversarial_batches

W, H = 48, 40


def _planes(det):
    return [np.concatenate([p.reshape(-1) for p in det.get_sae(cam)]) for cam in (0, 1)]


@pytest.mark.parametrize("world", [2, 3, 5])
def test_slices_compose_to_the_sequential_update(oracle, world):
    """N slices applied with the two-phase rule == the whole batch applied in stream order, over three
    batches of adversarial per-pixel histories (state carried between batches)"""
    from esvio_amd.dist import time_slice
    ref = oracle.Detector(W, H)
    engines = [OracleSliceEngine(oracle, W, H) for _ in range(world)]
    nd = engines[0].sae_plane_doubles()
    for L, R in adversarial_batches(W, H, 3, seed=world):
        ref.create_sae(0, L)
        ref.create_sae(1, R)
        cuts = [(L[slice(*time_slice(len(L), world, r))], R[slice(*time_slice(len(R), world, r))])
                for r in range(world)]
        last_all = np.empty(world * nd)
        s_all = np.empty(world * nd)
        for r, e in enumerate(engines):
            e.sae_slice_last(cuts[r][0], cuts[r][1], last_all[r * nd:(r + 1) * nd])
        for r, e in enumerate(engines):
            e.sae_slice_apply(cuts[r][0], cuts[r][1], last_all, r, s_all[r * nd:(r + 1) * nd])
        for e in engines:
            e.sae_slice_commit(last_all, s_all, world)
        for e in engines:
            for a, b in zip(_planes(e.det), _planes(ref)):
                assert np.array_equal(a, b), int((a != b).sum())


def test_first_event_rule_alone_is_not_enough(oracle):
    """why the exchange carries the exact L and not just "re-decide each slice's first event": with
    stamps that go backwards an event that is NOT its slice's first at the pixel still depends on
    the carried-in L of the other polarity (L[!p] > L[p] with L[p] set inside the slice)"""
    from esvio_amd.events import make_events
    # slice 0: polarity 1 at t=9.0 ; slice 1: polarity 0 at t=5.000 then polarity 0 at t=5.001
    a = make_events([3], [3], [9_000_000], [1])
    b = make_events([3, 3], [3, 3], [5_000_000, 5_001_000], [0, 0])
    ref = oracle.Detector(W, H)
    ref.create_sae(0, make_events([3, 3, 3], [3, 3, 3], [9_000_000, 5_000_000, 5_001_000], [1, 0, 0]))
    S0 = ref.get_sae(0)[2]
    assert S0[3, 3] == 5.001  # the SECOND event of slice 1 passes only because L[1] = 9.0 > L[0] = 5.0
    e = [OracleSliceEngine(oracle, W, H) for _ in range(2)]
    nd = e[0].sae_plane_doubles()
    last_all, s_all = np.empty(2 * nd), np.empty(2 * nd)
    empty = a[:0]
    for r, (ev, eng) in enumerate(zip((a, b), e)):
        eng.sae_slice_last(ev, empty, last_all[r * nd:(r + 1) * nd])
    for r, (ev, eng) in enumerate(zip((a, b), e)):
        eng.sae_slice_apply(ev, empty, last_all, r, s_all[r * nd:(r + 1) * nd])
    e[0].sae_slice_commit(last_all, s_all, 2)
    assert np.array_equal(_planes(e[0].det)[0], _planes(ref)[0])


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import oracle as O
    from esvio_amd.dist import TimeSlicedSae
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleSliceEngine(O, W, H)
        ts = TimeSlicedSae(eng, rank, world, dist, device="cpu")
        ref = O.Detector(W, H)
        for L, R in adversarial_batches(W, H, 3, seed=11):
            ts.apply_batch(L, R)
            ref.create_sae(0, L)
            ref.create_sae(1, R)
            for a, b in zip(_planes(eng.det), _planes(ref)):
                assert np.array_equal(a, b), (rank, int((a != b).sum()))
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_time_sliced_sae_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
