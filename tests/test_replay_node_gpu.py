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
    with open(path, "wb") as f:
        f.write(b"ESVB" + struct.pack("<IIII", 1, W, H, len(msgs)))
        for cam, stamp, ev in msgs:
            f.write(struct.pack("<BBBBId", cam, 0, 0, 0, len(ev), stamp))
            f.write(np.ascontiguousarray(ev).tobytes())


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


def _oracle_node(oracle, msgs):
    """stereo_event_tracker_node.cpp:372-418 (pairing) and :145-344 (handle_stereo_event), restated
    here, over the oracle tracker"""
    tr = oracle.Tracker(oracle.make_config(W, H, lk_accum=1, **KW))
    ql, qr, pairs = [], [], []
    for cam, stamp, ev in msgs:
        (ql if cam == 0 else qr).append((stamp, ev))
        while ql and qr:
            tl, trr = ql[0][0], qr[0][0]
            if tl < trr - 0.2:
                ql.pop(0)
            elif tl > trr + 0.2:
                qr.pop(0)
            else:
                pairs.append((ql.pop(0), qr.pop(0)))
    first, first_t, last_t, pub_count, init_pub = True, 0.0, 0.0, 1, False
    out = []
    for (stamp, L), (_, R) in pairs:
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
        r = tr.track_event(event_times(L)[-1], L, R, pub)
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
    args = [tool, log, dump, "max_cnt=%d" % KW["max_cnt"], "min_dist=%d" % KW["min_dist"], "freq=%d" % FREQ] + list(opts)
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
