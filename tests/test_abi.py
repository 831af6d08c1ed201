"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol
include/esvio_fe.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from esvio_amd import frontend as FE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="esvio_fe.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(esvio_fe_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    """two headers, two lists: the boundary (esvio_fe.h) and the test taps (esvio_fe_test.h) — disjoint"""
    assert _declared_symbols() == sorted(FE.ABI_SYMBOLS)
    assert _declared_symbols("esvio_fe_test.h") == sorted(FE.TEST_SYMBOLS)
    assert not set(FE.ABI_SYMBOLS) & set(FE.TEST_SYMBOLS)
    # nothing that smells like a tap in the public header
    assert not [s for s in FE.ABI_SYMBOLS if "_debug_" in s or s.startswith("esvio_fe_host_")]


def test_integration_md_covers_the_whole_public_header():
    """INTEGRATION.md section 3 maps every entry point of include/esvio_fe.h to what it replaces"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 3."):doc.index("## 4.")]
    missing = [s for s in FE.ABI_SYMBOLS if "`%s`" % s not in sec]
    assert not [s for s in FE.TEST_SYMBOLS if "| `%s`" % s in sec]
    assert not missing, missing


def test_library_exports_every_declared_symbol():
    L = FE.load_library()
    for s in _declared_symbols() + _declared_symbols("esvio_fe_test.h"):
        assert hasattr(L, s), s
    assert b"gfx950" in L.esvio_fe_version()


def test_struct_layouts_match_header():
    # esvio_fe_event is 16 B; config/camera sizes as laid out by the C compiler (natural alignment)
    from esvio_amd.events import EVENT_DTYPE
    assert EVENT_DTYPE.itemsize == 16
    assert [EVENT_DTYPE.fields[n][1] for n in ("x", "y", "sec", "nsec", "polarity")] == [0, 2, 4, 8, 12]
    assert C.sizeof(FE.Camera) == 64
    assert C.sizeof(FE.Config) == 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 2 * 64
    assert C.sizeof(FE.Tracks) == 8 + 9 * 8
    # esvio_fe_latency: u64, 4 doubles, u64, 3 x i32 (+ 4 padding), 2 x i64, 16 doubles, 2 x u64
    assert C.sizeof(FE.Latency) == 8 + 32 + 8 + 16 + 16 + 16 * 8 + 16 and FE.LATENCY_PHASES == 16
    # esvio_fe_latency_call: u64, 2 x i32, 2 doubles, 16 doubles
    assert C.sizeof(FE.LatencyCall) == 8 + 8 + 16 + 16 * 8
    L = FE.load_library()
    names = [L.esvio_fe_latency_phase_name(i).decode() for i in range(FE.LATENCY_PHASES)]
    assert names[0].startswith("enqueue") and names[4] == "host ransac" and L.esvio_fe_latency_phase_name(99) == b""


def test_create_fails_loudly_without_gpu(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    L = FE.load_library()
    h = C.c_void_p()
    rc = L.esvio_fe_create(C.byref(FE.make_config(640, 480)), C.byref(h))
    assert rc == -2 and not h  # ESVIO_FE_ENODEVICE: no CPU fallback exists
    with pytest.raises(FE.FrontendError):
        FE.FeatureTracker(FE.make_config(640, 480))


def test_create_rejects_bad_config():
    L = FE.load_library()
    h = C.c_void_p()
    for kw, rc_expected in ((dict(min_dist=2), -1), (dict(equalize=2), -1),
                            (dict(median_blur_kernel_size=8), -4), (dict(median_blur_kernel_size=-1), -1),
                            (dict(decay_ms=0.0), -1),
                            (dict(max_cnt=0), -1)):
        cfg = FE.make_config(640, 480, **kw)
        assert L.esvio_fe_create(C.byref(cfg), C.byref(h)) == rc_expected, kw
    assert L.esvio_fe_create(None, C.byref(h)) == -1
    assert L.esvio_fe_destroy(None) == -1
    assert L.esvio_fe_track_event(None, 0.0, None, 0, None, 0, 0, 1, None) == -1


def test_host_stages_run_without_gpu(oracle):
    """the two host-side stages of the boundary (no device needed) agree with the oracle"""
    cam = dict(fx=560.0, fy=555.0, cx=320.5, cy=239.0, k1=-0.31, k2=0.11, p1=4e-4, p2=-7e-4)
    rng = np.random.default_rng(0)
    for u, v in rng.uniform(0, 640, (50, 2)):
        a = FE.lift_projective(cam, u, v)
        b = oracle.lift_projective(cam, u, v)
        assert np.array_equal(a, b)
    nodist = dict(cam, k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    assert np.allclose(FE.lift_projective(nodist, 320.5, 239.0), [0, 0, 1])


def test_host_ransac_matches_oracle(oracle):
    """esvio_fe_find_fundamental_mat (host stage of rejectWithF_event) vs the oracle on two-view
    problems: general motion with outliers, pure translation, LMedS sizes (8..14), tiny sets."""
    rng = np.random.default_rng(5)
    K = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1.0]])
    for trial in range(60):
        n = int(rng.integers(8, 300))
        X = rng.uniform(-1, 1, (n, 3)) * np.array([2, 1.5, 1]) + np.array([0, 0, 4.0])
        w = rng.normal(0, 0.01, 3)
        R = np.eye(3) + np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])

        def proj(X, R, t):
            x = (K @ ((R @ X.T).T + t).T).T
            return (x[:, :2] / x[:, 2:]).astype(np.float32)
        p1 = proj(X, np.eye(3), np.zeros(3))
        p2 = proj(X, R if trial % 3 else np.eye(3), rng.normal(0, 0.05, 3))
        p2 += rng.normal(0, 0.05, p2.shape).astype(np.float32)
        k = int(0.15 * n)
        p2[:k] += rng.normal(0, 6, (k, 2)).astype(np.float32)
        cnt_o, st_o, _ = oracle.find_fundamental(p1, p2, 1.0, 0.99)
        cnt_p, st_p = FE.find_fundamental_mat(p1, p2, 1.0, 0.99)
        assert cnt_o == cnt_p and np.array_equal(st_o, st_p), (trial, n)
    assert FE.find_fundamental_mat(p1[:5], p2[:5])[0] == 0


@pytest.mark.parametrize("threads", [2, 4])
def test_host_ransac_with_helper_threads_is_bit_identical(oracle, threads):
    """esvio_fe_set_host_threads' RANSAC (helpers solve/score iterations, the caller draws and
    replays in order) gives the sequential loop's inlier flags whatever the interleaving: many
    outlier ratios (10..1000 iterations), degenerate inputs, sizes around the RANSAC/LMedS switch"""
    rng = np.random.default_rng(17 + threads)
    K = np.array([[460, 0, 320], [0, 460, 240], [0, 0, 1.0]])
    for trial in range(120):
        n = int(rng.integers(15, 320)) if trial % 10 else int(rng.integers(7, 17))
        X = rng.uniform(-1, 1, (n, 3)) * np.array([2, 1.5, 1]) + np.array([0, 0, 4.0])
        t = rng.normal(0, 0.05, 3)
        x = (K @ (X + t).T).T
        p1 = (K @ X.T).T
        p1 = (p1[:, :2] / p1[:, 2:]).astype(np.float32)
        p2 = (x[:, :2] / x[:, 2:]).astype(np.float32) + rng.normal(0, 0.05, (n, 2)).astype(np.float32)
        k = int(rng.uniform(0.0, 0.6) * n)
        p2[:k] += rng.normal(0, 8, (k, 2)).astype(np.float32)
        if trial % 17 == 3:
            p2[:] = p1  # no motion at all
        if trial % 17 == 5:
            p1[:, 1] = 100.0  # every point on one line: no subset passes the collinearity check
            p2[:, 1] = 100.0
        cnt_1, st_1 = FE.find_fundamental_mat(p1, p2, 1.0, 0.99)
        cnt_t, st_t = FE.find_fundamental_mat(p1, p2, 1.0, 0.99, threads=threads)
        assert cnt_1 == cnt_t and np.array_equal(st_1, st_t), (trial, n)
        if trial % 4 == 0:
            cnt_o, st_o, _ = oracle.find_fundamental(p1, p2, 1.0, 0.99)
            assert cnt_o == cnt_t and np.array_equal(st_o, st_t), (trial, n)


def test_host_ransac_does_not_wait_for_a_helper_that_lost_its_cpu(oracle):
    """A helper thread that is descheduled in the middle of a job keeps that job's buffer marked as in use
    (round 3's 1.8 ms RANSAC calls: the caller waited for it).  With one buffer held the call takes the
    other one, with both held it runs without the helpers; flags and count stay those of the sequential
    loop, for the RANSAC (>= 15 points) and the LMedS (8..14) branch, and the counters show which path ran."""
    rng = np.random.default_rng(5)
    K = np.array([[460, 0, 320], [0, 460, 240], [0, 0, 1.0]])
    for trial in range(24):
        n = int(rng.integers(15, 200)) if trial % 3 else int(rng.integers(8, 15))
        X = rng.uniform(-1, 1, (n, 3)) * np.array([2, 1.5, 1]) + np.array([0, 0, 4.0])
        t = rng.normal(0, 0.05, 3)
        x = (K @ (X + t).T).T
        p1 = (K @ X.T).T
        p1 = (p1[:, :2] / p1[:, 2:]).astype(np.float32)
        p2 = (x[:, :2] / x[:, 2:]).astype(np.float32) + rng.normal(0, 0.05, (n, 2)).astype(np.float32)
        k = int(rng.uniform(0.0, 0.5) * n)
        p2[:k] += rng.normal(0, 8, (k, 2)).astype(np.float32)
        cnt_1, st_1 = FE.find_fundamental_mat(p1, p2, 1.0, 0.99)
        for hold in (0, 1, 2, 3):
            FE.ransac_tail(reset=True)
            cnt_h, st_h = FE.find_fundamental_mat(p1, p2, 1.0, 0.99, threads=4, hold_mask=hold)
            assert cnt_h == cnt_1 and np.array_equal(st_h, st_1), (trial, n, hold)
            tail = FE.ransac_tail()
            # (a fresh pool's first job is epoch 1 = buffer 0: holding that one makes it move on to buffer 1)
            assert tail["solo_jobs"] == (1 if hold == 3 else 0), (hold, tail)
            assert tail["skipped_buffers"] == (1 if hold == 1 else 0), (hold, tail)
            assert max(tail["max_us"], tail["lmeds_max_us"]) > 0


def test_ransac_helpers_with_other_work_between_jobs():
    """The host-batch staging hands the spinning RANSAC helpers its chunks between jobs (and has every awake
    helper call the hook once when it is registered: a thread's first HIP calls belong there, not into a
    track call).  With units of other work arriving before every job the flags and the count stay those of
    the sequential loop and all units get done."""
    rng = np.random.default_rng(9)
    K = np.array([[460, 0, 320], [0, 460, 240], [0, 0, 1.0]])
    for trial in range(8):
        n = int(rng.integers(15, 200)) if trial % 3 else int(rng.integers(8, 15))
        X = rng.uniform(-1, 1, (n, 3)) * np.array([2, 1.5, 1]) + np.array([0, 0, 4.0])
        t = rng.normal(0, 0.05, 3)
        x = (K @ (X + t).T).T
        p1 = (K @ X.T).T
        p1 = (p1[:, :2] / p1[:, 2:]).astype(np.float32)
        p2 = (x[:, :2] / x[:, 2:]).astype(np.float32) + rng.normal(0, 0.05, (n, 2)).astype(np.float32)
        k = int(rng.uniform(0.0, 0.5) * n)
        p2[:k] += rng.normal(0, 8, (k, 2)).astype(np.float32)
        cnt_1, st_1 = FE.find_fundamental_mat(p1, p2, 1.0, 0.99)
        for threads, units in ((2, 16), (4, 64), (4, 0)):
            cnt, st, idle = FE.find_fundamental_mat_idle(p1, p2, 1.0, 0.99, threads=threads, repeats=5, idle_units=units)
            assert cnt == cnt_1 and np.array_equal(st, st_1), (trial, n, threads, units)
            assert idle["idle_left"] == 0 and idle["idle_done"] == 5 * units, (idle, threads, units)
            assert idle["idle_calls"] >= idle["idle_done"], idle  # (+ the call at registration of every helper that was idle then)
    L = FE.load_library()
    assert L.esvio_fe_find_fundamental_mat_idle(None, None, 0, 1.0, 0.99, 1, 1, 0, None, None, None) == -1


def test_header_is_plain_c(tmp_path):
    """include/esvio_fe.h is the drop-in boundary: it must compile as C99 (no C++ / HIP / torch types)
    and a C translation unit must link against the library's exports"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "esvio_fe.h"\n#include "esvio_fe_test.h"\n'
                   'int main(void) { esvio_fe_config c; esvio_fe_tracks t; esvio_fe_motion m; (void)c; (void)t; (void)m;\n'
                   '  return esvio_fe_kernel_count() > 0 && esvio_fe_version() != 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(root, "include"), str(src)])


def test_stage_copy_copies_exactly_the_bytes_asked_for():
    """the chunk copy of the host-batch staging (streaming stores for an aligned destination, memcpy for the
    rest): every length around the 64-byte step, every source alignment, aligned and unaligned
    destinations — the bytes arrive and the bytes around them stay"""
    L = FE.load_library()
    rng = np.random.default_rng(4)
    assert L.esvio_fe_host_stage_copy(None, None, 0) == 0 and L.esvio_fe_host_stage_copy(None, None, 16) == -1
    for n in [0, 1, 15, 16, 17, 63, 64, 65, 127, 128, 200, 4096, 65536, 65536 + 48, 262144 + 16]:
        for so in (0, 1, 8, 16):
            for do in (0, 16, 3):
                src = rng.integers(0, 256, n + so + 64, dtype=np.uint8)
                buf = np.full(n + 256 + do, 0xA5, np.uint8)
                base = (-buf.ctypes.data) % 64 + do  # dst = 64-byte aligned + do
                d = buf[base:base + n]
                assert L.esvio_fe_host_stage_copy(C.c_void_p(buf.ctypes.data + base), C.c_void_p(src.ctypes.data + so), n) == 0
                assert np.array_equal(d, src[so:so + n]), (n, so, do)
                assert (buf[:base] == 0xA5).all() and (buf[base + n:] == 0xA5).all(), (n, so, do)


def test_stage_pack_round_trip_and_refusals():
    """the 8-byte form a plain call's host batch crosses PCIe in (fe_evstage.cpp stage_pack, unpacked on the device by
    k_stage_pull_packed): x | y << 16, nsec | (polarity != 0) << 30 | (sec - base) << 31.  Every field a kernel reads
    comes back (polarity as != 0, the record's padding bytes as zero: no kernel reads them — and a ROS message leaves
    them uninitialised); a chunk whose stamps step back over a second, jump two, or carry an nsec no ros::Time has is
    refused (it then travels raw)."""
    from esvio_amd.events import EVENT_DTYPE
    L = FE.load_library()
    L.esvio_fe_host_stage_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(8)

    def pack(ev):
        raw = np.ascontiguousarray(ev).view(np.uint8).reshape(-1)
        buf = np.zeros(len(raw) // 2 + 64, np.uint8)
        off = (-buf.ctypes.data) % 16
        base = C.c_uint32(0)
        rc = L.esvio_fe_host_stage_pack(C.c_void_p(buf.ctypes.data + off), C.c_void_p(raw.ctypes.data), len(raw), C.byref(base))
        return rc, base.value, buf[off:off + len(raw) // 2].view(np.uint32).reshape(-1, 2)

    for n in (1, 2, 3, 4095, 4096, 4097):
        ev = np.zeros(n, EVENT_DTYPE)
        ev["x"], ev["y"] = rng.integers(0, 65536, n), rng.integers(0, 65536, n)
        t = np.sort(rng.integers(0, 1_400_000_000, n)) + 1_700_000_000 * 10 ** 9 + 300_000_000  # crosses one second boundary
        ev["sec"], ev["nsec"] = t // 10 ** 9, t % 10 ** 9
        ev["polarity"] = rng.choice([0, 1, 255, 7], n)
        raw = ev.view(np.uint8).reshape(-1, 16)
        raw[:, 13:] = rng.integers(0, 256, (n, 3))  # garbage in the padding
        rc, base, p = pack(ev)
        assert rc == 1 and base == ev["sec"][0], (n, rc)
        assert np.array_equal(p[:, 0], ev["x"].astype(np.uint32) | (ev["y"].astype(np.uint32) << 16))
        assert np.array_equal(p[:, 1] & 0x3fffffff, ev["nsec"]) and np.array_equal((p[:, 1] >> 30) & 1, (ev["polarity"] != 0))
        assert np.array_equal(base + (p[:, 1] >> 31), ev["sec"])
    ev = np.zeros(64, EVENT_DTYPE)
    ev["sec"], ev["nsec"] = 100, np.arange(64) * 1000
    assert pack(ev)[0] == 1
    for k in list(range(40, 48)) + [61, 62, 63]:  # (both events of the loop's round, and the odd last one)
        for bad in ("back", "jump", "nsec", "nsec_top"):
            e2 = ev[:63].copy() if k < 63 else ev.copy()
            k2 = min(k, len(e2) - 1)
            if bad == "back":
                e2["sec"][k2] = 99
            elif bad == "jump":
                e2["sec"][k2:] = 102
            elif bad == "nsec":
                e2["nsec"][k2] = 1 << 30
            else:
                e2["nsec"][k2] = (1 << 31) | 5
            assert pack(e2)[0] == 0, (bad, k)
    # one step forwards over a second is fine wherever it falls
    for k in range(1, 64):
        e2 = ev.copy()
        e2["sec"][k:] = 101
        rc, base, p = pack(e2)
        assert rc == 1 and base == 100 and np.array_equal(base + (p[:, 1] >> 31), e2["sec"]), k
    assert L.esvio_fe_host_stage_pack(None, None, 16, None) < 0


def test_host_hypot_is_cv_hypot():
    """cv::SVD's Jacobi rotations call hypot unqualified inside namespace cv, where lapack.cpp's own
    template (a * sqrt(1 + (b/a)^2)) hides libm's — so the library's rotations use that formula, in IEEE
    operations only.  Held here against the same formula evaluated by numpy (each step one correctly
    rounded IEEE operation, so bit for bit), over the magnitudes the solver sees, wide exponent ranges,
    near-equal and very unequal operands and zeros; and shown to differ from libm's hypot in the last
    bit on a visible share of inputs (which is why it matters which one the restatement uses)."""
    rng = np.random.default_rng(12)
    n = 1_000_000
    x = rng.normal(size=n) * 10.0 ** rng.uniform(-40, 40, n)
    y = x * rng.uniform(-3, 3, n)
    y[::5] = rng.normal(size=len(y[::5])) * 10.0 ** rng.uniform(-40, 40, len(y[::5]))
    x[::1001] = 0.0
    y[::1003] = 0.0
    xs = [x, rng.integers(0, 100000, n).astype(np.float64), rng.uniform(1e5, 1e23, n)]
    ys = [y, rng.integers(0, 100000, n).astype(np.float64), rng.uniform(1e5, 1e23, n) * 10.0 ** rng.uniform(-20, 3, n)]
    differs_from_libm = 0
    for a, b in zip(xs, ys):
        aa, bb = np.abs(a), np.abs(b)
        hi, lo = np.maximum(aa, bb), np.minimum(aa, bb)
        with np.errstate(all="ignore"):
            r = lo / hi
            want = np.where(hi > 0, hi * np.sqrt(1 + r * r), 0.0)
            libm = np.hypot(a, b)
        got = FE.host_hypot(a, b)
        bad = got.view(np.uint64) != want.view(np.uint64)
        assert not bad.any(), (int(bad.sum()), a[bad][:3], b[bad][:3])
        differs_from_libm += int((got != libm).sum())
    assert differs_from_libm > 1000


def test_round3_entry_points_validate_their_arguments(has_gpu):
    """the entry points added in round 3 (event memory, fault injection, announced motion compensation)
    refuse bad arguments with ESVIO_FE_EINVAL and never touch a device for that; without a GPU an
    allocation fails with ESVIO_FE_EHIP and leaves the out-pointer null"""
    L = FE.load_library()
    p = C.c_void_p(1)
    assert L.esvio_fe_mem_alloc(7, 64, C.byref(p)) == -1          # unknown memory space
    assert L.esvio_fe_mem_alloc(FE.HOST, 64, None) == -1
    assert L.esvio_fe_mem_free(7, None) == -1
    assert L.esvio_fe_mem_free(FE.HOST, None) == 0                 # free(NULL)
    assert L.esvio_fe_mem_upload(None, None, 0) == 0 and L.esvio_fe_mem_upload(None, None, 16) == -1
    assert L.esvio_fe_debug_inject(None, 0) == -1 and L.esvio_fe_debug_counters(None, None) == -1
    assert L.esvio_fe_plain_call_counters(None, None) == -1
    assert L.esvio_fe_set_next_batch_mc(None, 0.0, None, 0, None, 0, FE.HOST, 0, None) == -1
    if not has_gpu:
        q = C.c_void_p(1)
        assert L.esvio_fe_mem_alloc(FE.HOST, 4096, C.byref(q)) == -3 and not q
