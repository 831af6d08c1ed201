"""tools/replay_node (C++: sync_process + handle_stereo_event over the C ABI) against a
test-side restatement of the reference node driving the ORACLE tracker: the PointCloud rows of every
published frame are bit-identical, through a stream discontinuity, in the one-batch-in-flight and in
the replay schedule, and with the RCCL all-gather hand-off (one-rank communicator) switched on."""
import math
import os
import struct
import subprocess

import numpy as np
import pytest

from esvio_amd import frontend as FE
from esvio_amd.events import event_times
from esvio_amd.synth import SceneStream

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, FREQ = 346, 260, 15
KW = dict(max_cnt=150, min_dist=10, f_ransac=1)


def _messages():
    """(cam, header_stamp, events) in arrival order; a 1.5 s hole after batch 9, one right message
    0.5 s late (thrown by the pairing rule), one left message with no partner in range"""
    s = SceneStream(W, H, rate=1e6, seed=17, n_rect=12, size=(30.0, 90.0))
    msgs = []
    for b in range(22):
        if b == 10:
            s.t_us += 1_500_000
        L, R, t_end = s.next_batch()
        stamp = t_end * 1e-6
        msgs.append((0, stamp, L))
        msgs.append((1, stamp + 0.001, R))
    return msgs


def _write_log(path, msgs):
    """kinds 0/1: (kind, stamp, events); 2: (2, stamp, 6 doubles — IMU); 3: (3, stamp, 3 doubles — odometry)"""
    version = 2 if any(m[0] > 1 for m in msgs) else 1
    with open(path, "wb") as f:
        f.write(b"ESVB" + struct.pack("<IIII", version, W, H, len(msgs)))
        for kind, stamp, body in msgs:
            f.write(struct.pack("<BBBBId", kind, 0, 0, 0, body.size if kind >= 4 else len(body), stamp))
            if kind <= 1 or kind >= 4:
                f.write(np.ascontiguousarray(body).tobytes())
            else:
                f.write(np.asarray(body, np.float64).tobytes())


def _read_dump(path):
    b = open(path, "rb").read()
    assert b[:4] == b"ESVD"
    n, = struct.unpack_from("<I", b, 4)
    o, frames = 8, []
    for _ in range(n):
        stamp, restart, pub, _, _, rows = struct.unpack_from("<dBBBBI", b, o)
        o += 16
        a = np.frombuffer(b, np.float32, rows * 8, o).reshape(rows, 8).copy()
        o += rows * 32
        frames.append((stamp, restart, pub, a))
    return frames


def _c_round(x):
    r = math.floor(x)
    return r + 1 if x - r >= 0.5 else r


class _Imu:
    """imu_callback / state_callback and the Motion_correction_value of node:102-125,195-252, restated
    for the oracle side (float32 where the node holds Eigen::Vector3f; zeros where it leaves locals
    uninitialised)"""

    def __init__(self, K):
        self.K, self.imu, self.odo, self.last = K, [], [], 0.0
        self.v_cur = np.zeros(3, np.float32)
        self.v_pre = np.zeros(3, np.float32)
        self.t_pre = self.t_cur = 0.0

    def deliver(self, kind, stamp, body):
        if kind == 2:
            if stamp > self.last:
                self.last = stamp
                self.imu.append((stamp, body[:3]))
        else:
            self.odo.append((stamp, body))

    def value(self, oracle, L, header_stamp):
        t0 = event_times(L[:1])[0]
        v, a, w = np.zeros(3), np.zeros(3, np.float32), np.zeros(3, np.float32)
        if self.imu:
            if self.odo:
                t, vel = self.odo.pop(0)
                v = np.asarray(vel, np.float64)
                self.v_pre, self.v_cur = self.v_cur, v.astype(np.float32)
                self.t_pre, self.t_cur = self.t_cur, t
                a = ((self.v_cur - self.v_pre).astype(np.float64) / (self.t_cur - self.t_pre)).astype(np.float32)
            while self.imu and self.imu[0][0] < t0:
                self.imu.pop(0)
            if self.imu:
                w = np.asarray(self.imu[0][1], np.float64).astype(np.float32)
        return oracle.make_motion(header_stamp, v=v, v_pre=self.v_pre, accel=a, omega=w, fx=self.K[0],
                                  fy=self.K[1], cx=self.K[2], cy=self.K[3])


def _oracle_node(oracle, msgs, mc_K=None):
    """stereo_event_tracker_node.cpp:372-418 (pairing) and :145-344 (handle_stereo_event), restated
    here, over the oracle tracker.  mc_K: Do_motion_correction with these fx, fy, cx, cy."""
    tr = oracle.Tracker(oracle.make_config(W, H, **KW))
    imu = _Imu(mc_K) if mc_K else None
    ql, qr, pairs = [], [], []
    for mi, (cam, stamp, ev) in enumerate(msgs):
        if cam > 1:
            continue
        (ql if cam == 0 else qr).append((stamp, ev))
        while ql and qr:
            tl, trr = ql[0][0], qr[0][0]
            if tl < trr - 0.2:
                ql.pop(0)
            elif tl > trr + 0.2:
                qr.pop(0)
            else:
                pairs.append((ql.pop(0), qr.pop(0), mi))
    first, first_t, last_t, pub_count, init_pub = True, 0.0, 0.0, 1, False
    out = []
    delivered = 0
    for (stamp, L), (_, R), ready in pairs:
        while delivered < ready:  # callbacks in log order; the pair is handled when it is complete
            if msgs[delivered][0] > 1 and imu:
                imu.deliver(*msgs[delivered])
            delivered += 1
        if len(L) == 0:
            continue
        if first:
            first, first_t, last_t = False, stamp, stamp
            continue
        if stamp - last_t > 1.0 or stamp < last_t:
            first, last_t, pub_count = True, 0.0, 1
            out.append((stamp, 1, 0, np.zeros((0, 8), np.float32)))
            continue
        last_t = stamp
        rate = 1.0 * pub_count / (stamp - first_t)
        pub = _c_round(rate) <= FREQ
        if pub and abs(rate - FREQ) < 0.01 * FREQ:
            first_t, pub_count = stamp, 0
        if imu is None:
            r = tr.track_event(event_times(L)[-1], L, R, pub)
        else:
            r = tr.track_event(event_times(L)[-1], L, R, pub, motion=imu.value(oracle, L, stamp))
        rows = np.zeros((0, 8), np.float32)
        published = 0
        if pub:
            pub_count += 1
            rr = []
            ids = set()
            for j in range(len(r.ids)):
                if r.track_cnt[j] > 1:
                    ids.add(int(r.ids[j]))
                    rr.append((r.cur_un_pts[j, 0], r.cur_un_pts[j, 1], 1.0, np.float32(int(r.ids[j]) * 2),
                               r.cur_pts[j, 0], r.cur_pts[j, 1], r.pts_velocity[j, 0], r.pts_velocity[j, 1]))
            for j in range(len(r.ids_right)):
                if int(r.ids_right[j]) in ids:
                    rr.append((r.cur_un_right_pts[j, 0], r.cur_un_right_pts[j, 1], 1.0,
                               np.float32(int(r.ids_right[j]) * 2 + 1), r.cur_right_pts[j, 0],
                               r.cur_right_pts[j, 1], r.right_pts_velocity[j, 0], r.right_pts_velocity[j, 1]))
            if not init_pub:
                init_pub = True
            else:
                published = 1
                rows = np.asarray(rr, np.float32).reshape(-1, 8)
        out.append((stamp, 0, published, rows))
    return out


def _run(tmp_path, log, name, *opts):
    from esvio_amd import build as B
    tool = B.build_tools()
    dump = str(tmp_path / (name + ".bin"))
    args = [tool, log, dump, "max_cnt=%d" % KW["max_cnt"], "min_dist=%d" % KW["min_dist"], "freq=%d" % FREQ,
            "lk_accum=%d" % FE.DEFAULT_LK_ACCUM] + list(opts)
    p = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return _read_dump(dump), p.stdout


def test_replay_node_matches_the_oracle_node(oracle, tmp_path):
    msgs = _messages()
    # a right message half a second late: the pairing rule throws the left one it meets first
    late = list(msgs)
    cam, stamp, ev = late[9]
    assert cam == 1
    late[9] = (cam, stamp + 0.5, ev)
    log = str(tmp_path / "log.esvb")
    _write_log(log, late)
    ref = _oracle_node(oracle, late)
    assert sum(f[2] for f in ref) >= 5 and sum(f[1] for f in ref) >= 1
    runs = {"plain": [], "replay": ["ahead=3", "lazy=1", "threads=3"]}
    if os.path.exists("/opt/rocm/lib/librccl.so.1"):
        runs["rccl"] = ["rccl=1"]
    for name, opts in runs.items():
        got, stdout = _run(tmp_path, log, name, *opts)
        assert len(got) == len(ref), (name, len(got), len(ref), stdout)
        for k, (g, r) in enumerate(zip(got, ref)):
            assert g[0] == r[0] and g[1] == r[1] and g[2] == r[2], (name, k, g[:3], r[:3])
            assert g[3].shape == r[3].shape, (name, k)
            assert np.array_equal(g[3].view(np.uint32), r[3].view(np.uint32)), (name, k)
        if name == "rccl":
            assert " 0 RCCL exchanges" not in stdout, stdout


def test_replay_node_motion_compensation(oracle, tmp_path):
    """Do_motion_correction: 1 — the harness assembles the Motion_correction_value from logged IMU and
    back-end odometry messages (node:102-125,195-252) and calls esvio_fe_track_event_mc.  The IMU
    runs at 200 Hz with one message out of order; the back end reports a velocity per batch that
    jumps, so that some batches exceed the 5 m/s^2 gate and are warped and others are not."""
    ev_msgs = _messages()[:28]
    K = (0.9 * W, 0.9 * W, W / 2.0 + 2.5, H / 2.0 - 1.25)
    rng = np.random.default_rng(3)
    msgs, t_imu, k_odo = [], None, 0
    for cam, stamp, ev in ev_msgs:
        if cam == 0:
            t0 = event_times(ev[:1])[0]
            if t_imu is None:
                t_imu = t0 - 0.01
            while t_imu < stamp:  # IMU messages up to this batch's stamp
                msgs.append((2, t_imu, list(rng.uniform(-0.3, 0.3, 3)) + [0.0, 0.0, 9.8]))
                if len(msgs) % 37 == 0:
                    msgs.append((2, t_imu - 0.002, [9.0, 9.0, 9.0, 0, 0, 0]))  # out of order: dropped
                t_imu += 0.005
            if k_odo % 5 != 4:  # (every fifth batch comes without a new back-end state)
                jump = 0.3 if k_odo % 2 else 0.02
                msgs.append((3, stamp - 0.004, [0.4 + jump, -0.2 - jump / 2, 0.1 * (k_odo % 3)]))
            k_odo += 1
        msgs.append((cam, stamp, ev))
    log = str(tmp_path / "log_mc.esvb")
    _write_log(log, msgs)
    ref = _oracle_node(oracle, msgs, mc_K=K)
    plain = _oracle_node(oracle, msgs)
    assert sum(f[2] for f in ref) >= 4
    # the warp matters: the published rows differ from those of the uncompensated node
    assert any(a[3].shape != b[3].shape or not np.array_equal(a[3], b[3]) for a, b in zip(ref, plain))
    got, stdout = _run(tmp_path, log, "mc", "mc=1", "fx=%r" % K[0], "fy=%r" % K[1], "cx=%r" % K[2], "cy=%r" % K[3])
    assert len(got) == len(ref), (len(got), len(ref), stdout)
    for k, (g, r) in enumerate(zip(got, ref)):
        assert g[:3] == r[:3], (k, g[:3], r[:3])
        assert g[3].shape == r[3].shape, k
        assert np.array_equal(g[3].view(np.uint32), r[3].view(np.uint32)), k


def _oracle_image_node(oracle, msgs, kw):
    """stereo_image_tracker_node.cpp:210-250 (pairing within a second) and :54-183
    (handle_stereo_image), restated here, over the oracle tracker"""
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    il, ir, out = [], [], []
    first, first_t, last_t, pub_count, init_pub = True, 0.0, 0.0, 1, False
    for kind, stamp, img in msgs:
        (il if kind == 4 else ir).append((stamp, img))
        while il and ir:
            tl, trr = il[0][0], ir[0][0]
            if tl <= trr - 1:
                il.pop(0)
                continue
            if tl > trr + 1:
                ir.pop(0)
                continue
            (stamp_l, L), (_, R) = il.pop(0), ir.pop(0)
            if first:
                first, first_t, last_t = False, stamp_l, stamp_l
                continue
            if stamp_l - last_t > 1.0 or stamp_l < last_t:
                first, last_t, pub_count = True, 0.0, 1
                out.append((stamp_l, 1, 0, np.zeros((0, 8), np.float32)))
                continue
            last_t = stamp_l
            rate = 1.0 * pub_count / (stamp_l - first_t)
            pub = _c_round(rate) <= FREQ
            if pub and abs(rate - FREQ) < 0.01 * FREQ:
                first_t, pub_count = stamp_l, 0
            r = tr.track_image(stamp_l, L, R, pub)
            rows, published = np.zeros((0, 8), np.float32), 0
            if pub:
                pub_count += 1
                rr, ids = [], set()
                for j in range(len(r.ids)):
                    if r.track_cnt[j] > 1:
                        ids.add(int(r.ids[j]))
                        rr.append((r.cur_un_pts[j, 0], r.cur_un_pts[j, 1], 1.0, np.float32(int(r.ids[j]) * 2),
                                   r.cur_pts[j, 0], r.cur_pts[j, 1], r.pts_velocity[j, 0], r.pts_velocity[j, 1]))
                for j in range(len(r.ids_right)):
                    if int(r.ids_right[j]) in ids:
                        rr.append((r.cur_un_right_pts[j, 0], r.cur_un_right_pts[j, 1], 1.0,
                                   np.float32(int(r.ids_right[j]) * 2 + 1), r.cur_right_pts[j, 0],
                                   r.cur_right_pts[j, 1], r.right_pts_velocity[j, 0], r.right_pts_velocity[j, 1]))
                if not init_pub:
                    init_pub = True
                else:
                    published, rows = 1, np.asarray(rr, np.float32).reshape(-1, 8)
            out.append((stamp_l, 0, published, rows))
    return out


@pytest.mark.parametrize("equalize", [0, 1])
def test_replay_image_node(oracle, tmp_path, equalize):
    """the reference's second node (stereo_image_tracker_node.cpp) through the same harness: a log of
    mono8 image messages, pairing within 1 s, handle_stereo_image -> esvio_fe_track_image, the
    PointCloud rows of every published frame; with a 1.4 s hole in the stream, a right image that
    arrives 1.2 s late and one that never arrives (the reference then pairs what it has)"""
    from esvio_amd.synth import ImageStream
    s = ImageStream(W, H, velocity=(3, -2), disparity=9, seed=8)
    msgs = []
    t_shift = 0.0
    for f in range(24):
        L, R, t = s.next_frame()
        if f == 9:
            t_shift = 1.4
        t += t_shift
        msgs.append((4, t, L))
        if f != 15:  # (a right image that never arrives: the pairing goes on with the next one)
            msgs.append((5, t + 0.0005 + (1.2 if f == 4 else 0.0), R))
    kw = dict(max_cnt=120, min_dist=20, flow_back=1, equalize=equalize)
    log = str(tmp_path / "log_img.esvb")
    _write_log(log, msgs)
    ref = _oracle_image_node(oracle, msgs, kw)
    assert sum(f[2] for f in ref) >= 4 and sum(f[1] for f in ref) >= 1
    assert max(f[3].shape[0] for f in ref) > 60
    from esvio_amd import build as B
    dump = str(tmp_path / "img.bin")
    p = subprocess.run([B.build_tools(), log, dump, "max_cnt=120", "min_dist=20", "freq=%d" % FREQ, "lk_accum=%d" % FE.DEFAULT_LK_ACCUM,
                        "equalize=%d" % equalize], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got = _read_dump(dump)
    assert len(got) == len(ref), (len(got), len(ref), p.stdout)
    for k, (g, r) in enumerate(zip(got, ref)):
        assert g[:3] == r[:3], (k, g[:3], r[:3])
        assert g[3].shape == r[3].shape, k
        assert np.array_equal(g[3].view(np.uint32), r[3].view(np.uint32)), k
