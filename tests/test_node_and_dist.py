"""Host-side logic around the kernels: the node harness mirror (handle_stereo_event) and the
multi-rank track exchange, on CPU (gloo, world_size 2)."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest

from esvio_amd.events import make_events
from esvio_amd.node import FreqControl, StereoEventTrackerNode, pack_point_cloud, pack_track_records


def _fake_tracker():
    ft = SimpleNamespace()
    ft.ids = np.array([7, 3, 9, 12], np.int32)
    ft.track_cnt = np.array([5, 1, 2, 3], np.int32)
    ft.cur_pts = np.arange(8, dtype=np.float32).reshape(4, 2)
    ft.cur_un_pts = ft.cur_pts / 100
    ft.pts_velocity = ft.cur_pts / 10
    ft.ids_right = np.array([3, 9, 12], np.int32)
    ft.cur_right_pts = np.arange(6, dtype=np.float32).reshape(3, 2) + 50
    ft.cur_un_right_pts = ft.cur_right_pts / 100
    ft.right_pts_velocity = ft.cur_right_pts / 10
    return ft


def test_point_cloud_packing_rules():
    """node:273-329: left entries need track_cnt > 1; right entries need their id in the left set;
    channel 0 is id*2+cam as float32; z == 1"""
    pc = pack_point_cloud(_fake_tracker())
    assert pc.shape == (5, 8)
    assert pc[:, 3].tolist() == [14.0, 18.0, 24.0, 19.0, 25.0]  # 7,9,12 left; 9,12 right (3 dropped)
    assert (pc[:, 2] == 1.0).all()
    v = (pc[:, 3] + 0.5).astype(int)  # the estimator's decode (stereo_estimator_node.cpp:388-401)
    assert (v // 2).tolist() == [7, 9, 12, 9, 12] and (v % 2).tolist() == [0, 0, 0, 1, 1]
    assert pc[3, 4] == 52.0 and pc[3, 5] == 53.0
    rec = pack_track_records(_fake_tracker(), 4)
    assert rec.shape == (8, 8) and (rec[5:, 3] == -1).all() and np.array_equal(rec[:5], pc)


def test_freq_control_publishes_at_freq():
    """node:177-188 at 30 Hz input and freq 15: about every other frame is published"""
    fc = FreqControl(15)
    t0 = 100.0
    pubs = []
    for k in range(300):
        p = fc.pub_this_frame(t0 + k / 30.0)
        pubs.append(p)
        if p:
            fc.published()
    rate = sum(pubs) / (299 / 30.0)
    assert 14.0 <= rate <= 16.5
    assert not pubs[0]


class _StubTracker:
    def __init__(self):
        self.calls = []
        self.resets = 0
        f = _fake_tracker()
        self.__dict__.update(f.__dict__)

    def trackEvent(self, t, L, R, pub):
        self.calls.append((t, len(L), len(R), pub))

    def reset(self):
        self.resets += 1


def test_handle_stereo_event_flow():
    tr = _StubTracker()
    node = StereoEventTrackerNode(tr, freq=15)
    ev = make_events([1, 2], [1, 2], [5_000_000, 5_010_000], [1, 0])
    empty = ev[:0]
    assert node.handle(empty, ev, 5.0) is None and not tr.calls            # node:150
    assert node.handle(ev, ev, 5.0) is None and not tr.calls               # first frame (node:155)
    out = node.handle(ev, ev, 5.0333)
    # pub_count starts at 1: 1/0.0333 = 30 Hz > FREQ -> tracked but not published (node:177-188);
    # cur_time is the last LEFT event's stamp (node:190)
    assert len(tr.calls) == 1 and tr.calls[0][0] == 5.01 and tr.calls[0][3] is False
    assert out is None
    out = node.handle(ev, ev, 5.0 + 2 / 30.0)                               # 1/0.0667 = 15 -> publish
    assert tr.calls[-1][3] is True and out is None                         # first publish swallowed
    out = node.handle(ev, ev, 5.0 + 4 / 30.0)
    assert tr.calls[-1][3] is True and out is not None and out.shape[1] == 8
    got = [node.handle(ev, ev, 5.2 + k / 30.0) for k in range(30)]
    assert any(g is not None and g.shape[1] == 8 for g in got)
    # time gap > 1 s: the node re-arms (restart flag published) and the next frame is treated as the
    # first one; the tracker itself is NOT reset (node:163-173 only touches the node's own flags)
    n = len(tr.calls)
    assert node.handle(ev, ev, 9.0) is None and node.restart_flag and node.restart_count == 1
    assert tr.resets == 0 and len(tr.calls) == n
    assert node.handle(ev, ev, 9.03) is None and len(tr.calls) == n and not node.restart_flag
    node.handle(ev, ev, 9.06)
    assert len(tr.calls) == n + 1
    # time going backwards re-arms as well
    node.handle(ev, ev, 8.0)
    assert node.restart_count == 2 and node.restart_flag and tr.resets == 0
    # the non-reference option clears the tracker too
    tr2 = _StubTracker()
    node2 = StereoEventTrackerNode(tr2, freq=15, reset_tracker_on_restart=True)
    node2.handle(ev, ev, 5.0)
    node2.handle(ev, ev, 9.0)
    assert tr2.resets == 1


def test_motion_correction_value_assembly():
    """node:102-125,195-252 by hand: the velocity pair and the acceleration come from the last two
    odometry messages (one consumed per batch), omega from the first IMU message not older than the
    batch's first event; IMU messages out of order are dropped; nothing is assembled without IMU."""
    from esvio_amd.node import MotionCorrection
    mc = MotionCorrection(300.0, 301.0, 160.0, 120.0)
    # no IMU yet: everything zero (the reference's locals are uninitialised there), odometry untouched
    mc.state_callback(9.0, (1.0, 2.0, 3.0))
    v = mc.value(10.0, 10.03)
    assert v["t1"] == 10.03 and v["v"] == (0.0, 0.0, 0.0) and v["accel"] == (0.0, 0.0, 0.0)
    assert v["omega"] == (0.0, 0.0, 0.0) and len(mc.odom_buffer_) == 1
    assert (v["fx"], v["fy"], v["cx"], v["cy"]) == (300.0, 301.0, 160.0, 120.0)
    # IMU: one before the batch, one inside, one out of order (dropped)
    mc.imu_callback(9.99, (0.1, 0.2, 0.3))
    mc.imu_callback(10.01, (0.4, 0.5, 0.6))
    mc.imu_callback(10.005, (9.0, 9.0, 9.0))
    assert len(mc.imu_buf) == 2
    v = mc.value(10.0, 10.03)
    # first odometry message: v_pre = old v_cur = 0, v_cur = (1,2,3), t_pre = 0, t_cur = 9 -> a = v / 9
    assert v["v"] == (1.0, 2.0, 3.0) and v["v_pre"] == (0.0, 0.0, 0.0)
    assert v["accel"] == tuple(np.float32(x / 9.0) for x in (1.0, 2.0, 3.0))
    assert v["omega"] == tuple(np.float32(x) for x in (0.4, 0.5, 0.6))  # 9.99 < t_left_0 was popped
    assert len(mc.imu_buf) == 1 and not mc.odom_buffer_
    # second odometry message 0.1 s later: a = (float32 difference) / 0.1 as float32
    mc.state_callback(9.1, (1.7, 1.4, 3.05))
    v = mc.value(10.033, 10.066)
    d = np.float32([1.7, 1.4, 3.05]) - np.float32([1.0, 2.0, 3.0])
    want = tuple((d.astype(np.float64) / (9.1 - 9.0)).astype(np.float32))
    assert v["accel"] == want and v["v_pre"] == (1.0, 2.0, 3.0) and v["v"] == (1.7, 1.4, 3.05)
    assert v["omega"] == (0.0, 0.0, 0.0) and not mc.imu_buf  # the 10.01 message is older than this batch
    # no new odometry: State_ / temp_a are zero, the previous velocity persists
    mc.imu_callback(10.07, (1.0, 1.0, 1.0))
    v = mc.value(10.066, 10.1)
    assert v["v"] == (0.0, 0.0, 0.0) and v["accel"] == (0.0, 0.0, 0.0) and v["v_pre"] == (1.0, 2.0, 3.0)
    assert v["omega"] == (1.0, 1.0, 1.0)


def test_node_passes_the_motion_value_to_the_tracker():
    from esvio_amd.node import MotionCorrection

    class Tr(_StubTracker):
        def trackEvent(self, t, L, R, pub, measurements=None):
            self.calls.append((t, len(L), len(R), pub, measurements))

    tr = Tr()
    mc = MotionCorrection(1.0, 2.0, 3.0, 4.0)
    node = StereoEventTrackerNode(tr, freq=15, motion=mc, make_motion=lambda **kw: kw)
    ev = make_events([1, 2], [1, 2], [5_000_000, 5_010_000], [1, 0])
    mc.imu_callback(5.005, (0.5, 0.0, -0.5))
    mc.state_callback(4.9, (1.0, 0.0, 0.0))
    node.handle(ev, ev, 5.0)
    node.handle(ev, ev, 5.0333, header_stamp=5.02)
    m = tr.calls[-1][4]
    assert m["t1"] == 5.02 and m["v"] == (1.0, 0.0, 0.0) and m["omega"] == (0.5, 0.0, -0.5)
    assert m["fx"] == 1.0 and m["cy"] == 4.0


def test_freq_control_rounds_like_c():
    """node:177 calls C round(): half away from zero.  pub_count / dt = 10.5 at FREQ 10 must NOT
    publish (round -> 11); Python's round() would give 10 and publish."""
    from esvio_amd.node import c_round
    assert [c_round(v) for v in (0.5, 1.5, 2.5, 10.5, 10.49, 11.5, 0.49999999999999994)] == \
        [1, 2, 3, 11, 10, 12, 0]
    fc = FreqControl(10)
    fc.pub_this_frame(0.0)           # first frame
    fc.pub_count = 21
    assert fc.peek(2.0) is False     # 21 / 2.0 = 10.5 -> 11 > 10
    assert fc.pub_this_frame(2.0) is False
    fc.pub_count = 20
    assert fc.pub_this_frame(2.0) is True


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from esvio_amd.dist import TrackExchange, merged_point_cloud, shard_units
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert shard_units(5, world, rank) == ([0, 2, 4] if rank == 0 else [1, 3])
        ex = TrackExchange(4, world, device="cpu", dist=dist)
        for frame in range(3):
            ft = _fake_tracker()
            ft.ids = ft.ids + 100 * rank + 1000 * frame
            ft.ids_right = ft.ids_right + 100 * rank + 1000 * frame
            ex.submit(pack_track_records(ft, 4), async_op=True)
            g = ex.result()
            assert g.shape == (world, 8, 8)
            merged = merged_point_cloud(g)
            ids = ((merged[:, 3] + 0.5).astype(int) // 2).tolist()
            exp = []
            for rk in range(world):
                exp += [x + 100 * rk + 1000 * frame for x in (7, 9, 12, 9, 12)]
            assert ids == exp, (ids, exp)
            assert merged[:, 8].tolist() == [0.0] * 5 + [1.0] * 5
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_track_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_freq_control_peek_predicts_next_decision():
    """FreqControl.peek (the PUB hint for esvio_fe_set_next_batch) equals the decision
    pub_this_frame takes afterwards, for regular and jittered batch timestamps."""
    from esvio_amd.node import FreqControl
    rng = np.random.default_rng(3)
    for freq, hz in [(15, 30.0), (20, 30.0), (30, 30.0), (10, 25.0)]:
        fc = FreqControl(freq)
        t = 0.5
        n_pub = 0
        for i in range(200):
            t += (1.0 / hz) * (1.0 + 0.2 * rng.uniform(-1, 1))
            guess = fc.peek(t)
            pub = fc.pub_this_frame(t)
            assert guess == pub, (freq, hz, i)
            if pub:
                fc.published()
                n_pub += 1
        assert n_pub > 20


def test_handle_stereo_image_flow():
    """stereo_image_tracker_node.cpp:54-183: first frame swallowed, 30 Hz input at freq 15 tracks
    every frame and publishes every other one, first publish swallowed, a discontinuity re-arms the
    node without resetting the tracker"""
    from esvio_amd.node import StereoImageTrackerNode

    class Stub:
        def __init__(self):
            self.calls = []
            self.__dict__.update(_fake_tracker().__dict__)

        def trackImage(self, t, left, right, pub):
            self.calls.append((t, pub))

    tr = Stub()
    node = StereoImageTrackerNode(tr, freq=15)
    img = np.zeros((4, 4), np.uint8)
    assert node.handle(img, img, 5.0) is None and not tr.calls
    assert node.handle(img, img, 5.0 + 1 / 30.0) is None and tr.calls[-1] == (5.0 + 1 / 30.0, False)
    assert node.handle(img, img, 5.0 + 2 / 30.0) is None and tr.calls[-1][1] is True   # swallowed
    out = node.handle(img, img, 5.0 + 4 / 30.0)
    assert tr.calls[-1][1] is True and out is not None and out.shape == (5, 8)
    n = len(tr.calls)
    assert node.handle(img, img, 7.5) is None and len(tr.calls) == n and node.restart_count == 1
    assert node.handle(img, img, 7.53) is None and len(tr.calls) == n      # treated as first frame
    node.handle(img, img, 7.56)
    assert len(tr.calls) == n + 1


def test_frequency_rule_at_a_duplicate_stamp_follows_c():
    """a message whose stamp equals first_image_time (dt == 0): the reference computes inf or NaN,
    `round(...) <= FREQ` is false and the frame is not published — no exception, like
    tools/replay_node.cpp; negative halves round away from zero like C's round()"""
    from esvio_amd.node import FreqControl, c_rate, c_round, rate_allows
    assert [c_round(v) for v in (-0.5, -1.5, -2.4, 2.5, 0.49999999999999994)] == [-1, -2, -2, 3, 0]
    assert not rate_allows(c_rate(3, 0.0), 15) and not rate_allows(c_rate(0, 0.0), 15)
    assert rate_allows(c_rate(3, np.float64(0.25)), 15) and not rate_allows(c_rate(30, np.float64(0.25)), 15)
    fc = FreqControl(15)
    assert fc.pub_this_frame(10.0) is False        # first frame
    assert fc.pub_this_frame(10.0) is False        # duplicate stamp: dt == 0
    assert fc.peek(10.0) is False
    assert fc.pub_this_frame(10.5) is True


def test_bench_gpus_n_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside a launcher re-executes itself under torch.distributed.run with
    N ranks on 127.0.0.1 (the command line the driver uses for N > 1); under a launcher (WORLD_SIZE set)
    it does not"""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_exec(prog, argv, env):
        seen.update(prog=prog, argv=list(argv), env=dict(env))
        raise SystemExit(0)
    monkeypatch.setattr(os, "execvpe", fake_exec)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in a
    assert a[a.index("--nproc-per-node") + 1] == "8" and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert a[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and a[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
