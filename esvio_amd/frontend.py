"""Host-side mirror of the reference's feature_tracker interface over the C ABI.

``FeatureTracker`` / ``EventDetector`` keep the reference's names and argument meaning
(feature_tracker/src/feature_tracker.h:49-173, event_detector/event_detector.h:18-80) so the
parity tests read like calls into the reference.  Everything here is ctypes plumbing over
``libesvio_fe.so`` (include/esvio_fe.h); there is no Python compute path and no CPU fallback —
if the library or a GPU is missing, construction raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST, DEVICE = 0, 1
LK_USE_INITIAL_FLOW = 4


class FrontendError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")]


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("decay_ms", C.c_double),
        ("ignore_polarity", C.c_int32), ("median_blur_kernel_size", C.c_int32),
        ("feature_filter_threshold", C.c_double),
        ("ts_lk_threshold", C.c_double),
        ("max_cnt", C.c_int32), ("min_dist", C.c_int32),
        ("flow_back", C.c_int32), ("equalize", C.c_int32),
        ("f_threshold", C.c_double),
        ("f_ransac", C.c_int32), ("lk_accum", C.c_int32),
        ("focal_length", C.c_int32), ("device", C.c_int32),
        ("cam", Camera * 2),
    ]


class Motion(C.Structure):
    """esvio_fe_motion: the Motion_correction_value fields createSAE_* reads + detector.init's K"""
    _fields_ = [("t1", C.c_double), ("v", C.c_double * 3), ("v_pre", C.c_float * 3),
                ("accel", C.c_float * 3), ("omega", C.c_float * 3),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


def make_motion(t1, v, v_pre, accel, omega, fx, fy, cx, cy):
    m = Motion()
    m.t1 = float(t1)
    for i in range(3):
        m.v[i] = float(v[i])
        m.v_pre[i] = float(v_pre[i])
        m.accel[i] = float(accel[i])
        m.omega[i] = float(omega[i])
    m.fx, m.fy, m.cx, m.cy = float(fx), float(fy), float(cx), float(cy)
    return m


class Tracks(C.Structure):
    _fields_ = [
        ("n_left", C.c_int32), ("n_right", C.c_int32),
        ("ids", C.c_void_p), ("track_cnt", C.c_void_p),
        ("cur_pts", C.c_void_p), ("cur_un_pts", C.c_void_p), ("pts_velocity", C.c_void_p),
        ("ids_right", C.c_void_p), ("cur_right_pts", C.c_void_p),
        ("cur_un_right_pts", C.c_void_p), ("right_pts_velocity", C.c_void_p),
    ]


# every symbol include/esvio_fe.h declares
# the drop-in boundary: every entry point include/esvio_fe.h declares (INTEGRATION.md section 3 maps each to the
# reference call it replaces)
ABI_SYMBOLS = [
    "esvio_fe_calc_optical_flow_pyr_lk", "esvio_fe_comm_init", "esvio_fe_comm_unique_id", "esvio_fe_create",
    "esvio_fe_create_sae", "esvio_fe_create_sae_stereo", "esvio_fe_create_sae_stereo_mc", "esvio_fe_destroy",
    "esvio_fe_exchange_begin", "esvio_fe_exchange_end", "esvio_fe_exchange_tracks", "esvio_fe_export_image",
    "esvio_fe_features_to_track", "esvio_fe_find_fundamental_mat", "esvio_fe_finish", "esvio_fe_get_sae",
    "esvio_fe_get_time_surface", "esvio_fe_good_features_to_track", "esvio_fe_import_image", "esvio_fe_is_corner",
    "esvio_fe_last_error", "esvio_fe_mem_alloc", "esvio_fe_mem_free", "esvio_fe_mem_upload",
    "esvio_fe_pack_track_records", "esvio_fe_register_host_buffer", "esvio_fe_reserve", "esvio_fe_reset",
    "esvio_fe_sae_plane_doubles", "esvio_fe_sae_slice_apply", "esvio_fe_sae_slice_commit",
    "esvio_fe_sae_slice_last", "esvio_fe_sae_to_time_surface", "esvio_fe_set_auto_exchange",
    "esvio_fe_set_host_threads", "esvio_fe_set_launch_thread", "esvio_fe_set_lazy_new_stereo",
    "esvio_fe_set_next_batch", "esvio_fe_set_next_batch_mc", "esvio_fe_track_event", "esvio_fe_track_event_mc",
    "esvio_fe_track_image", "esvio_fe_unregister_host_buffer", "esvio_fe_version",
]
# test / measurement taps: include/esvio_fe_test.h (not part of the boundary)
TEST_SYMBOLS = [
    "esvio_fe_build_pyramid", "esvio_fe_debug_counters", "esvio_fe_debug_inject", "esvio_fe_device_memory",
    "esvio_fe_find_fundamental_mat_held", "esvio_fe_find_fundamental_mat_idle", "esvio_fe_find_fundamental_mat_mt",
    "esvio_fe_get_kernel_stats", "esvio_fe_host_hypot", "esvio_fe_host_nullspace", "esvio_fe_host_stage_copy",
    "esvio_fe_host_stage_pack", "esvio_fe_staging_counters",
    "esvio_fe_kernel_count", "esvio_fe_kernel_name", "esvio_fe_latency_phase_name", "esvio_fe_latency_recent",
    "esvio_fe_latency_stats", "esvio_fe_lift_projective", "esvio_fe_plain_call_counters", "esvio_fe_ransac_stats",
    "esvio_fe_ransac_tail", "esvio_fe_reset_kernel_stats", "esvio_fe_set_profiling", "esvio_fe_set_sae",
    "esvio_fe_stream",
]

LATENCY_PHASES = 16


class LatencyCall(C.Structure):  # esvio_fe_latency_call
    _fields_ = [("call", C.c_uint64), ("published", C.c_int32), ("reserved", C.c_int32), ("begin_ms", C.c_double),
                ("ms", C.c_double), ("phase_ms", C.c_double * 16)]


class Latency(C.Structure):  # esvio_fe_latency
    _fields_ = [("calls", C.c_uint64), ("mean_ms", C.c_double), ("p50_ms", C.c_double), ("p99_ms", C.c_double),
                ("max_ms", C.c_double), ("max_call", C.c_uint64), ("max_published", C.c_int32),
                ("max_cpu_begin", C.c_int32), ("max_cpu_end", C.c_int32), ("max_invol_switches", C.c_int64),
                ("max_allocs", C.c_int64), ("max_phase_ms", C.c_double * LATENCY_PHASES), ("allocs", C.c_uint64),
                ("invol_switches", C.c_uint64)]


_lib = None


def lib_path():
    # (ESVIO_FE_LIB: an experimental build of the library — kernel A/B measurements under tools/_bin)
    return os.environ.get("ESVIO_FE_LIB") or os.path.join(_HERE, "libesvio_fe.so")


def load_library(build_if_missing=True):
    """dlopen libesvio_fe.so (building it in-tree if absent).  Raises if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if build_if_missing and not os.environ.get("ESVIO_FE_LIB") and _build.needs_build():
        _build.build()
    if not os.path.exists(path):
        raise FrontendError("libesvio_fe.so is missing: run `python -m esvio_amd.build`")
    L = C.CDLL(path)
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    L.esvio_fe_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.esvio_fe_destroy.argtypes = [vp]
    L.esvio_fe_reset.argtypes = [vp]
    L.esvio_fe_last_error.restype = C.c_char_p
    L.esvio_fe_last_error.argtypes = [vp]
    L.esvio_fe_version.restype = C.c_char_p
    L.esvio_fe_create_sae.argtypes = [vp, i, vp, sz, i, C.POINTER(C.c_uint64)]
    L.esvio_fe_create_sae_stereo.argtypes = [vp, vp, sz, vp, sz, i, C.POINTER(C.c_uint64)]
    L.esvio_fe_sae_to_time_surface.argtypes = [vp, i, d, vp]
    L.esvio_fe_is_corner.argtypes = [vp, vp, sz, i, vp]
    L.esvio_fe_features_to_track.argtypes = [vp, vp, sz, i, i, vp, vp, vp, C.POINTER(C.c_int32)]
    L.esvio_fe_get_sae.argtypes = [vp, i, vp, vp, vp, vp]
    L.esvio_fe_set_sae.argtypes = [vp, i, vp, vp, vp, vp]
    L.esvio_fe_calc_optical_flow_pyr_lk.argtypes = [vp, vp, vp, i, i, vp, vp, vp, i, i, i, d, i]
    L.esvio_fe_build_pyramid.argtypes = [vp, vp, i, i, i, i, vp, vp, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.esvio_fe_find_fundamental_mat.argtypes = [vp, vp, i, d, d, vp, C.POINTER(C.c_int32)]
    L.esvio_fe_lift_projective.argtypes = [C.POINTER(Camera), d, d, vp]
    L.esvio_fe_track_event.argtypes = [vp, d, vp, sz, vp, sz, i, i, C.POINTER(Tracks)]
    L.esvio_fe_track_event_mc.argtypes = [vp, d, vp, sz, vp, sz, i, i, C.POINTER(Motion),
                                          C.POINTER(Tracks)]
    L.esvio_fe_create_sae_stereo_mc.argtypes = [vp, vp, sz, vp, sz, i, C.POINTER(Motion),
                                                C.POINTER(C.c_uint64)]
    L.esvio_fe_set_next_batch.argtypes = [vp, d, vp, sz, vp, sz, i, i]
    L.esvio_fe_set_next_batch_mc.argtypes = [vp, d, vp, sz, vp, sz, i, i, C.POINTER(Motion)]
    L.esvio_fe_mem_alloc.argtypes = [i, sz, vp]
    L.esvio_fe_mem_free.argtypes = [i, vp]
    L.esvio_fe_mem_upload.argtypes = [vp, vp, sz]
    L.esvio_fe_register_host_buffer.argtypes = [vp, sz]
    L.esvio_fe_unregister_host_buffer.argtypes = [vp]
    L.esvio_fe_debug_inject.argtypes = [vp, i]
    L.esvio_fe_debug_counters.argtypes = [vp, vp]
    L.esvio_fe_plain_call_counters.argtypes = [vp, vp]
    L.esvio_fe_good_features_to_track.argtypes = [vp, vp, i, d, d, vp, vp, vp, vp]
    L.esvio_fe_track_image.argtypes = [vp, d, vp, vp, i, vp]
    L.esvio_fe_pack_track_records.argtypes = [vp, vp, vp]
    L.esvio_fe_set_lazy_new_stereo.argtypes = [vp, i]
    L.esvio_fe_set_host_threads.argtypes = [vp, i]
    L.esvio_fe_ransac_stats.argtypes = [vp, i]
    L.esvio_fe_host_hypot.argtypes = [vp, vp, i, vp]
    L.esvio_fe_host_stage_copy.argtypes = [vp, vp, C.c_size_t]
    L.esvio_fe_host_nullspace.argtypes = [vp, i, i, vp, C.POINTER(C.c_int32)]
    L.esvio_fe_find_fundamental_mat_mt.argtypes = [vp, vp, i, d, d, i, vp, C.POINTER(C.c_int32)]
    L.esvio_fe_finish.argtypes = [vp, vp]
    L.esvio_fe_get_time_surface.argtypes = [vp, i, vp]
    L.esvio_fe_export_image.argtypes = [vp, i, vp, i]
    L.esvio_fe_import_image.argtypes = [vp, i, vp, i]
    L.esvio_fe_set_profiling.argtypes = [vp, i]
    L.esvio_fe_kernel_name.restype = C.c_char_p
    L.esvio_fe_kernel_name.argtypes = [i]
    L.esvio_fe_get_kernel_stats.argtypes = [vp, i, C.POINTER(d), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]
    L.esvio_fe_reset_kernel_stats.argtypes = [vp]
    L.esvio_fe_stream.restype = vp
    L.esvio_fe_stream.argtypes = [vp]
    L.esvio_fe_device_memory.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    L.esvio_fe_exchange_tracks.argtypes = [vp, vp, i, vp]
    L.esvio_fe_comm_unique_id.argtypes = [vp]
    L.esvio_fe_comm_init.argtypes = [vp, vp, i, i]
    L.esvio_fe_exchange_begin.argtypes = [vp]
    L.esvio_fe_exchange_end.argtypes = [vp, vp]
    L.esvio_fe_set_auto_exchange.argtypes = [vp, i]
    L.esvio_fe_sae_plane_doubles.restype = sz
    L.esvio_fe_sae_plane_doubles.argtypes = [vp]
    L.esvio_fe_sae_slice_last.argtypes = [vp, vp, sz, vp, sz, i, vp, i]
    L.esvio_fe_sae_slice_apply.argtypes = [vp, vp, sz, vp, sz, i, vp, i, i, vp, i]
    L.esvio_fe_sae_slice_commit.argtypes = [vp, vp, vp, i, i]
    L.esvio_fe_reserve.argtypes = [vp, sz, sz, i]
    L.esvio_fe_set_launch_thread.argtypes = [vp, i]
    L.esvio_fe_latency_stats.argtypes = [vp, C.POINTER(Latency), i]
    L.esvio_fe_latency_recent.argtypes = [vp, i, C.POINTER(LatencyCall)]
    L.esvio_fe_latency_phase_name.restype = C.c_char_p
    L.esvio_fe_latency_phase_name.argtypes = [i]
    L.esvio_fe_ransac_tail.argtypes = [vp, i]
    L.esvio_fe_find_fundamental_mat_held.argtypes = [vp, vp, i, d, d, i, i, vp, C.POINTER(C.c_int32)]
    L.esvio_fe_find_fundamental_mat_idle.argtypes = [vp, vp, i, d, d, i, i, i, vp, C.POINTER(C.c_int32), vp]
    _lib = L
    return L


# How calcOpticalFlowPyrLK's sums are accumulated when a caller does not say: 2 = in float, in the order of the
# x86 OpenCV build the reference node links (feature_tracker.cpp:410,417-418,490,495) — the reference's own
# arithmetic; 1 = exact integer sums.  ESVIO_LK_ACCUM=1 in the environment runs everything that does not say
# (the test suite, tools/) in the exact mode instead.
DEFAULT_LK_ACCUM = int(os.environ.get("ESVIO_LK_ACCUM", "2"))


def make_config(W, H, device=-1, **kw):
    """esvio_fe_config with the shipped defaults of config/esio_DSEC/esio.yaml:77-94
    (equalize forced 0) unless overridden."""
    c = Config()
    c.width, c.height = W, H
    c.decay_ms = kw.get("decay_ms", 20.0)
    c.ignore_polarity = kw.get("ignore_polarity", 0)
    c.median_blur_kernel_size = kw.get("median_blur_kernel_size", 0)
    c.feature_filter_threshold = kw.get("feature_filter_threshold", 0.01)
    c.ts_lk_threshold = kw.get("ts_lk_threshold", 128.0)
    c.max_cnt = kw.get("max_cnt", 300)
    c.min_dist = kw.get("min_dist", 10)
    c.flow_back = kw.get("flow_back", 1)
    c.equalize = kw.get("equalize", 0)
    c.f_threshold = kw.get("f_threshold", 1.0)
    c.f_ransac = kw.get("f_ransac", 1)
    c.lk_accum = kw.get("lk_accum", DEFAULT_LK_ACCUM)
    c.focal_length = kw.get("focal_length", 460)
    c.device = device
    cams = kw.get("cams")
    if cams is None:
        cams = [dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2.0, cy=H / 2.0, k1=-0.05, k2=0.01, p1=1e-4,
                     p2=-2e-4)] * 2
    for k in range(2):
        for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2"):
            setattr(c.cam[k], n, float(cams[k][n]))
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


FAULT_TICKET, FAULT_LOOKBACK, FAULT_SPECULATIVE, FAULT_CHAINED, FAULT_LAZY_LATE = 1, 2, 4, 8, 16


def _events_arg(ev):
    """numpy EVENT_DTYPE array -> (pointer, n, HOST); (device_ptr:int, n) tuple -> DEVICE."""
    if isinstance(ev, tuple):
        return C.c_void_p(ev[0]), int(ev[1]), DEVICE, None
    ev = np.ascontiguousarray(ev)
    if ev.dtype.itemsize != 16:
        raise ValueError("events must be 16-byte records (esvio_amd.events.EVENT_DTYPE)")
    return _p(ev), ev.shape[0], HOST, ev


class _Handle:
    def __init__(self, cfg):
        self.L = load_library()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.L.esvio_fe_create(C.byref(cfg), C.byref(h))
        if rc != 0 or not h:
            raise FrontendError("esvio_fe_create failed rc=%d (no GPU / bad config)" % rc)
        self.h = h

    def check(self, rc):
        if rc != 0:
            raise FrontendError("rc=%d: %s" % (rc, self.L.esvio_fe_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.esvio_fe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EventDetector:
    """esvio::EventDetector (event_detector.h:18-80) — batch forms of its per-event methods."""

    def __init__(self, cfg=None, handle=None, **kw):
        if handle is None:
            handle = _Handle(cfg if cfg is not None else make_config(**kw))
        self._hd = handle
        self.W, self.H = handle.cfg.width, handle.cfg.height

    def createSAE_left(self, events):
        return self._create_sae(0, events)

    def createSAE_right(self, events):
        return self._create_sae(1, events)

    def _create_sae(self, cam, events):
        ptr, n, space, keep = _events_arg(events)
        rej = C.c_uint64(0)
        self._hd.check(self._hd.L.esvio_fe_create_sae(self._hd.h, cam, ptr, n, space, C.byref(rej)))
        return rej.value

    def createSAE_stereo(self, left, right):
        pl, nl, sl, k1 = _events_arg(left)
        pr, nr, sr, k2 = _events_arg(right)
        assert sl == sr
        rej = C.c_uint64(0)
        self._hd.check(self._hd.L.esvio_fe_create_sae_stereo(self._hd.h, pl, nl, pr, nr, sl,
                                                             C.byref(rej)))
        return rej.value

    def createSAE_stereo_mc(self, left, right, motion):
        """createSAE_left/right(et, ex, ey, ep, measurements) loops (feature_tracker.cpp:627-641)"""
        pl, nl, sl, k1 = _events_arg(left)
        pr, nr, sr, k2 = _events_arg(right)
        assert sl == sr
        rej = C.c_uint64(0)
        self._hd.check(self._hd.L.esvio_fe_create_sae_stereo_mc(self._hd.h, pl, nl, pr, nr, sl,
                                                                C.byref(motion), C.byref(rej)))
        return rej.value

    def SAEtoTimeSurface_left(self, external_sync_time):
        return self._ts(0, external_sync_time)

    def SAEtoTimeSurface_right(self, external_sync_time):
        return self._ts(1, external_sync_time)

    def _ts(self, cam, t):
        out = np.empty((self.H, self.W), np.uint8)
        self._hd.check(self._hd.L.esvio_fe_sae_to_time_surface(self._hd.h, cam, float(t), _p(out)))
        return out

    def isCorner(self, events):
        """isCorner(e.ts.toSec(), e.x, e.y, e.polarity) for every event -> uint8 flags."""
        ptr, n, space, keep = _events_arg(events)
        flags = np.zeros(n, np.uint8)
        self._hd.check(self._hd.L.esvio_fe_is_corner(self._hd.h, ptr, n, space, _p(flags)))
        return flags

    def get_sae(self, cam):
        planes = [np.empty((self.H, self.W), np.float64) for _ in range(4)]
        self._hd.check(self._hd.L.esvio_fe_get_sae(self._hd.h, cam, *[_p(a) for a in planes]))
        return planes  # L0, L1, S0, S1

    def set_sae(self, cam, L0, L1, S0, S1):
        arrs = [np.ascontiguousarray(a, np.float64) for a in (L0, L1, S0, S1)]
        self._hd.check(self._hd.L.esvio_fe_set_sae(self._hd.h, cam, *[_p(a) for a in arrs]))


class FeatureTracker:
    """FeatureTracker (feature_tracker.h:49-173): trackEvent + the public result members."""

    def __init__(self, cfg=None, **kw):
        self.cfg = cfg if cfg is not None else make_config(**kw)
        self._hd = _Handle(self.cfg)
        self.detector = EventDetector(handle=self._hd)
        m = max(self.cfg.max_cnt, 1)
        self._bufs = dict(
            ids=np.zeros(m, np.int32), track_cnt=np.zeros(m, np.int32),
            cur_pts=np.zeros((m, 2), np.float32), cur_un_pts=np.zeros((m, 2), np.float32),
            pts_velocity=np.zeros((m, 2), np.float32),
            ids_right=np.zeros(m, np.int32), cur_right_pts=np.zeros((m, 2), np.float32),
            cur_un_right_pts=np.zeros((m, 2), np.float32),
            right_pts_velocity=np.zeros((m, 2), np.float32))
        self._tr = Tracks()
        for k, a in self._bufs.items():
            setattr(self._tr, k, a.ctypes.data)
        self._copy = False

    def close(self):
        self._hd.close()

    def trackEvent(self, cur_time, event_left, event_right, PUB_THIS_FRAME=True, copy=True,
                   measurements=None):
        """both overloads of FeatureTracker::trackEvent (feature_tracker.h:51-52); `measurements`
        is an esvio_fe_motion (frontend.make_motion) for the motion-compensated one"""
        pl, nl, sl, k1 = _events_arg(event_left)
        pr, nr, sr, k2 = _events_arg(event_right)
        assert sl == sr
        if measurements is None:
            self._hd.check(self._hd.L.esvio_fe_track_event(
                self._hd.h, float(cur_time), pl, nl, pr, nr, sl, int(PUB_THIS_FRAME), C.byref(self._tr)))
        else:
            self._hd.check(self._hd.L.esvio_fe_track_event_mc(
                self._hd.h, float(cur_time), pl, nl, pr, nr, sl, int(PUB_THIS_FRAME),
                C.byref(measurements), C.byref(self._tr)))
        return self._take(copy)

    _LEFT = frozenset(("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity"))
    _RIGHT = frozenset(("ids_right", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity"))
    _RESULTS = tuple(sorted(_LEFT | _RIGHT))

    def _take(self, copy):
        """the result members (.ids, .cur_pts, ...) are cut out of the call's output buffers when they are READ
        (__getattr__ below), not after every call: a replaying caller that looks at none of them pays nothing for
        them — nine slices and attribute stores were ~3 us of Python between two track calls"""
        d = self.__dict__
        for k in self._RESULTS:
            d.pop(k, None)
        self._copy = copy
        return self

    def __getattr__(self, name):  # (only reached when the attribute is not set: a result member not read yet)
        if name in FeatureTracker._LEFT:
            v = self._bufs[name][:self._tr.n_left]
        elif name in FeatureTracker._RIGHT:
            v = self._bufs[name][:self._tr.n_right]
        else:
            raise AttributeError(name)
        if self._copy:
            v = v.copy()
        self.__dict__[name] = v
        return v

    def trackImage(self, cur_time, img_left, img_right, PUB_THIS_FRAME=True, copy=True):
        """FeatureTracker::trackImage (feature_tracker.cpp:164-338); the handle's width/height/
        max_cnt/min_dist are the image camera's COL/ROW/MAX_CNT_IMG/MIN_DIST_IMG"""
        il = np.ascontiguousarray(img_left, np.uint8)
        assert il.shape == (self.cfg.height, self.cfg.width)
        ir = None
        if img_right is not None:
            ir = np.ascontiguousarray(img_right, np.uint8)
            assert ir.shape == il.shape
        self._hd.check(self._hd.L.esvio_fe_track_image(
            self._hd.h, float(cur_time), _p(il), None if ir is None else _p(ir), int(PUB_THIS_FRAME),
            C.byref(self._tr)))
        return self._take(copy)

    def goodFeaturesToTrack(self, image, maxCorners, qualityLevel=0.01, minDistance=30, mask=None,
                            want_eig=False):
        """cv::goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask) with
        trackImage's defaults (feature_tracker.cpp:228)"""
        img = np.ascontiguousarray(image, np.uint8)
        assert img.shape == (self.cfg.height, self.cfg.width)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        out = np.zeros((max(int(maxCorners), 1), 2), np.float32)
        eig = np.zeros(img.shape, np.float32) if want_eig else None
        n = C.c_int32()
        self._hd.check(self._hd.L.esvio_fe_good_features_to_track(
            self._hd.h, _p(img), int(maxCorners), float(qualityLevel), float(minDistance),
            None if m is None else _p(m), _p(out), C.byref(n), None if eig is None else _p(eig)))
        return (out[:n.value].copy(), eig) if want_eig else out[:n.value].copy()

    def set_lazy_new_stereo(self, on=True):
        """throughput option: published frames do not wait for the stereo LK of the corners they have
        just detected; finish() (or the next call) completes the right-camera vectors"""
        self._hd.check(self._hd.L.esvio_fe_set_lazy_new_stereo(self._hd.h, int(bool(on))))

    def set_host_threads(self, threads):
        """host threads for rejectWithF_event's RANSAC (results do not depend on the count)"""
        self._hd.check(self._hd.L.esvio_fe_set_host_threads(self._hd.h, int(threads)))

    def set_launch_thread(self, on=True):
        """replay mode: a thread of the handle issues the announced batches' prefetch launches"""
        self._hd.check(self._hd.L.esvio_fe_set_launch_thread(self._hd.h, int(bool(on))))

    def finish(self, copy=True):
        """complete a lazily returned frame and refresh the result members"""
        self._hd.check(self._hd.L.esvio_fe_finish(self._hd.h, C.byref(self._tr)))
        return self._take(copy)

    def pack_track_records(self, out=None):
        """node:273-329 packing of the current results into a (2*max_cnt, 8) float32 block
        (padding rows have id -1); `out` may be a preallocated (e.g. pinned) array"""
        if out is None:
            out = np.empty((2 * self.cfg.max_cnt, 8), np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == 16 * self.cfg.max_cnt
        n = C.c_int32()
        self._hd.check(self._hd.L.esvio_fe_pack_track_records(self._hd.h, _p(out), C.byref(n)))
        return out

    def set_next_batch(self, next_cur_time, event_left, event_right, PUB_NEXT_FRAME=False, measurements=None):
        """announce the batch of the FOLLOWING trackEvent call (throughput / replay mode);
        PUB_NEXT_FRAME is the PUB_THIS_FRAME that call is expected to carry (a hint);
        `measurements`: the esvio_fe_motion that call will pass (motion-compensated overload)"""
        pl, nl, sl, k1 = _events_arg(event_left)
        pr, nr, sr, k2 = _events_arg(event_right)
        assert sl == sr
        # host arrays must outlive the prefetch (up to six announced + the one being tracked)
        self._next_keep = (getattr(self, "_next_keep", []) + [(k1, k2)])[-8:]
        if measurements is None:
            self._hd.check(self._hd.L.esvio_fe_set_next_batch(self._hd.h, float(next_cur_time), pl, nl,
                                                              pr, nr, sl, int(bool(PUB_NEXT_FRAME))))
        else:
            self._hd.check(self._hd.L.esvio_fe_set_next_batch_mc(self._hd.h, float(next_cur_time), pl, nl, pr, nr, sl,
                                                                 int(bool(PUB_NEXT_FRAME)), C.byref(measurements)))

    def reset(self):
        self._hd.check(self._hd.L.esvio_fe_reset(self._hd.h))

    def reserve(self, max_left, max_right, host_batches=False):
        """size every event-proportional buffer for batches of up to max_left + max_right events now, so
        that no later call below that size allocates (esvio_fe_reserve)"""
        self._hd.check(self._hd.L.esvio_fe_reserve(self._hd.h, int(max_left), int(max_right), int(bool(host_batches))))

    def latency_stats(self, reset=False):
        """wall time of the trackEvent calls since the last reset as the calling thread saw them: count,
        mean / p50 / p99 / max [ms], and for the slowest call its index, its phases [ms], the CPUs it began
        and ended on, involuntary context switches and allocations inside it"""
        o = Latency()
        L = self._hd.L
        self._hd.check(L.esvio_fe_latency_stats(self._hd.h, C.byref(o), int(bool(reset))))
        phases = {L.esvio_fe_latency_phase_name(k).decode(): round(o.max_phase_ms[k], 4)
                  for k in range(LATENCY_PHASES) if o.max_phase_ms[k] > 0.0005}
        return dict(calls=int(o.calls), mean_ms=o.mean_ms, p50_ms=o.p50_ms, p99_ms=o.p99_ms, max_ms=o.max_ms,
                    max_call=int(o.max_call), max_published=bool(o.max_published),
                    max_cpu=(int(o.max_cpu_begin), int(o.max_cpu_end)), max_invol_switches=int(o.max_invol_switches),
                    max_allocs=int(o.max_allocs), max_phase_ms=phases, allocs=int(o.allocs),
                    invol_switches=int(o.invol_switches))

    def latency_recent(self, n):
        """the latest n (<= 256) track calls, oldest first: (call, published, begin_ms, ms, {phase: ms})"""
        L = self._hd.L
        names = [L.esvio_fe_latency_phase_name(k).decode() for k in range(LATENCY_PHASES)]
        out = []
        for back in range(n - 1, -1, -1):
            o = LatencyCall()
            if L.esvio_fe_latency_recent(self._hd.h, back, C.byref(o)) != 0:
                continue
            out.append((int(o.call), bool(o.published), o.begin_ms, o.ms,
                        {names[k]: o.phase_ms[k] for k in range(LATENCY_PHASES) if o.phase_ms[k] > 0.0005}))
        return out

    def debug_inject(self, mask):
        """make device-side waits expire on demand (FAULT_TICKET | FAULT_LOOKBACK | FAULT_SPECULATIVE |
        FAULT_CHAINED; 0: normal bounds), or replay mode's lazy completions always happen at their latest point
        (FAULT_LAZY_LATE)"""
        self._hd.check(self._hd.L.esvio_fe_debug_inject(self._hd.h, int(mask)))

    def staging_counters(self):
        """host batches staged / bytes / chunks that crossed PCIe packed to 8 B per event / chunks of such batches sent raw"""
        out = (C.c_uint64 * 4)()
        self._hd.check(self._hd.L.esvio_fe_staging_counters(self._hd.h, out))
        return dict(batches=out[0], bytes=out[1], chunks_packed=out[2], chunks_raw=out[3])

    def debug_counters(self):
        out = (C.c_uint64 * 4)()
        self._hd.check(self._hd.L.esvio_fe_debug_counters(self._hd.h, out))
        return dict(spec_redone=out[0], chain_redone=out[1], chain_launched=out[2], chain_used=out[3])

    def plain_call_counters(self):
        out = (C.c_uint64 * 4)()
        self._hd.check(self._hd.L.esvio_fe_plain_call_counters(self._hd.h, out))
        return dict(plain_calls=out[0], split_by_camera=out[1], stereo_chained=out[2], chained_redone=out[3])

    # ---- the handle's own RCCL communicator: asynchronous exchange of the track records
    def comm_init(self, unique_id, rank, world):
        """unique_id: the 128 bytes of comm_unique_id() made on rank 0"""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._hd.check(self._hd.L.esvio_fe_comm_init(self._hd.h, buf, int(rank), int(world)))
        self._x_world = int(world)

    def set_auto_exchange(self, on=True):
        self._hd.check(self._hd.L.esvio_fe_set_auto_exchange(self._hd.h, int(bool(on))))

    def exchange_begin(self):
        self._hd.check(self._hd.L.esvio_fe_exchange_begin(self._hd.h))

    def exchange_end(self, want=True):
        """wait for the latest exchange_begin; -> (world, 2*max_cnt, 8) float32, or None"""
        out = np.empty((self._x_world, 2 * self.cfg.max_cnt, 8), np.float32) if want else None
        self._hd.check(self._hd.L.esvio_fe_exchange_end(self._hd.h, _p(out)))
        return out

    def device_memory(self):
        """(free, total) bytes of the handle's device"""
        f, t = C.c_size_t(0), C.c_size_t(0)
        self._hd.check(self._hd.L.esvio_fe_device_memory(self._hd.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    # ---- one stream time-sliced across GPUs (esvio_fe_sae_slice_*); plane sets are numpy float64
    # arrays (host) or (device_ptr, n_sets) tuples
    def sae_plane_doubles(self):
        return int(self._hd.L.esvio_fe_sae_plane_doubles(self._hd.h))

    @staticmethod
    def _planes_arg(a):
        if isinstance(a, tuple):
            return C.c_void_p(a[0]), DEVICE
        return _p(a), HOST

    def sae_slice_last(self, event_left, event_right, out):
        pl, nl, sl, k1 = _events_arg(event_left)
        pr, nr, sr, k2 = _events_arg(event_right)
        assert sl == sr
        po, so = self._planes_arg(out)
        self._hd.check(self._hd.L.esvio_fe_sae_slice_last(self._hd.h, pl, nl, pr, nr, sl, po, so))

    def sae_slice_apply(self, event_left, event_right, last_before, n_before, s_out):
        pl, nl, sl, k1 = _events_arg(event_left)
        pr, nr, sr, k2 = _events_arg(event_right)
        assert sl == sr
        pi, si = self._planes_arg(last_before) if n_before else (None, HOST)
        po, so = self._planes_arg(s_out)
        self._hd.check(self._hd.L.esvio_fe_sae_slice_apply(self._hd.h, pl, nl, pr, nr, sl, pi, int(n_before),
                                                          si, po, so))

    def sae_slice_commit(self, last_all, s_all, n_slices):
        pa, sa = self._planes_arg(last_all)
        pb, sb = self._planes_arg(s_all)
        assert sa == sb
        self._hd.check(self._hd.L.esvio_fe_sae_slice_commit(self._hd.h, pa, pb, int(n_slices), sa))

    def gettimesurface(self, cam=0):
        out = np.empty((self.cfg.height, self.cfg.width), np.uint8)
        self._hd.check(self._hd.L.esvio_fe_get_time_surface(self._hd.h, cam, _p(out)))
        return out

    # ---- camera split (right camera on another GPU)
    def export_image(self, cam, dst=None):
        """current image of `cam` -> numpy (H,W) u8, or into a device buffer given as int pointer"""
        if dst is None:
            out = np.empty((self.cfg.height, self.cfg.width), np.uint8)
            self._hd.check(self._hd.L.esvio_fe_export_image(self._hd.h, cam, _p(out), HOST))
            return out
        self._hd.check(self._hd.L.esvio_fe_export_image(self._hd.h, cam, C.c_void_p(int(dst)), DEVICE))
        return None

    def import_image(self, cam, src):
        """numpy (H,W) u8 or device pointer (int): right image for the next trackEvent"""
        if isinstance(src, np.ndarray):
            src = np.ascontiguousarray(src, np.uint8)
            self._hd.check(self._hd.L.esvio_fe_import_image(self._hd.h, cam, _p(src), HOST))
        else:
            self._hd.check(self._hd.L.esvio_fe_import_image(self._hd.h, cam, C.c_void_p(int(src)), DEVICE))

    # ---- stage-level entry points used by the parity tests
    def Event_FeaturesToTrack(self, last_event, maxCorners, event_mask=None):
        ptr, n, space, keep = _events_arg(last_event)
        xy = np.zeros((max(maxCorners, 1), 2), np.float32)
        idx = np.zeros(max(maxCorners, 1), np.int32)
        k = C.c_int32(0)
        mask = None if event_mask is None else np.ascontiguousarray(event_mask, np.uint8)
        self._hd.check(self._hd.L.esvio_fe_features_to_track(
            self._hd.h, ptr, n, space, int(maxCorners), _p(mask), _p(xy), _p(idx), C.byref(k)))
        return xy[:k.value].copy(), idx[:k.value].copy()

    def calcOpticalFlowPyrLK(self, prev_img, next_img, prev_pts, next_pts=None, maxLevel=3,
                             max_count=30, eps=0.01, flags=0):
        prev_img = np.ascontiguousarray(prev_img, np.uint8)
        next_img = np.ascontiguousarray(next_img, np.uint8)
        h, w = prev_img.shape
        prev_pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        n = prev_pts.shape[0]
        nxt = (np.zeros((n, 2), np.float32) if next_pts is None
               else np.array(next_pts, np.float32).reshape(-1, 2).copy())
        status = np.zeros(n, np.uint8)
        self._hd.check(self._hd.L.esvio_fe_calc_optical_flow_pyr_lk(
            self._hd.h, _p(prev_img), _p(next_img), w, h, _p(prev_pts), _p(nxt), _p(status), n,
            maxLevel, max_count, eps, flags))
        return nxt, status

    def build_pyramid(self, img, max_level=3):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        nl, lw, lh = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._hd.check(self._hd.L.esvio_fe_build_pyramid(self._hd.h, _p(img), w, h, max_level, -1,
                                                         None, None, None, None, C.byref(nl)))
        levels = []
        for l in range(nl.value):
            self._hd.check(self._hd.L.esvio_fe_build_pyramid(
                self._hd.h, _p(img), w, h, max_level, l, None, None, C.byref(lw), C.byref(lh),
                C.byref(nl)))
            im = np.empty((lh.value, lw.value), np.uint8)
            dv = np.empty((lh.value, lw.value, 2), np.int16)
            self._hd.check(self._hd.L.esvio_fe_build_pyramid(
                self._hd.h, _p(img), w, h, max_level, l, _p(im), _p(dv), C.byref(lw), C.byref(lh),
                C.byref(nl)))
            levels.append((im, dv))
        return levels

    # ---- measurement
    def set_profiling(self, on=True):
        self._hd.check(self._hd.L.esvio_fe_set_profiling(self._hd.h, int(on)))

    def reset_kernel_stats(self):
        self._hd.check(self._hd.L.esvio_fe_reset_kernel_stats(self._hd.h))

    def kernel_stats(self):
        L = self._hd.L
        out = {}
        for k in range(L.esvio_fe_kernel_count()):
            ms, n, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
            self._hd.check(L.esvio_fe_get_kernel_stats(self._hd.h, k, C.byref(ms), C.byref(n),
                                                       C.byref(b)))
            out[L.esvio_fe_kernel_name(k).decode()] = dict(ms=ms.value, launches=n.value,
                                                           alg_bytes=b.value)
        return out

    @property
    def stream(self):
        return self._hd.L.esvio_fe_stream(self._hd.h)


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0); 128 bytes to hand to every rank's comm_init"""
    buf = (C.c_uint8 * 128)()
    rc = load_library().esvio_fe_comm_unique_id(buf)
    if rc:
        raise FrontendError("esvio_fe_comm_unique_id rc=%d (librccl.so not found?)" % rc)
    return bytes(buf)


class EventBuffer:
    """an event batch in memory from the library's own HIP runtime (esvio_fe_mem_alloc): pinned host memory
    (`.array`: a numpy view that can be passed wherever a host batch goes — it is DMA'd without the staging
    copy) or device memory (`.arg`: the (pointer, n) tuple the entry points take for ESVIO_FE_DEVICE)"""

    def __init__(self, events, space=HOST):
        ev = np.ascontiguousarray(events)
        assert ev.dtype.itemsize == 16
        self.space, self.n = space, len(ev)
        p = C.c_void_p()
        rc = load_library().esvio_fe_mem_alloc(space, ev.nbytes, C.byref(p))
        if rc:
            raise FrontendError("esvio_fe_mem_alloc rc=%d" % rc)
        self.ptr = p
        if space == HOST:
            raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(ev.nbytes, 16),))
            raw[:ev.nbytes] = ev.view(np.uint8).reshape(-1)
            self.array = raw[:ev.nbytes].view(ev.dtype)
        else:
            rc = load_library().esvio_fe_mem_upload(p, _p(ev), ev.nbytes)
            if rc:
                raise FrontendError("esvio_fe_mem_upload rc=%d" % rc)
            self.arg = (p.value, self.n)

    def free(self):
        if self.ptr:
            load_library().esvio_fe_mem_free(self.space, self.ptr)
            self.ptr = None
            self.array = None


class RegisteredEvents:
    """a caller-owned numpy event array page-locked where it lies (esvio_fe_register_host_buffer): batches handed
    over from it cross PCIe without the staging copy; `.array` is the array itself"""

    def __init__(self, events):
        self.array = np.ascontiguousarray(events)
        assert self.array.dtype.itemsize == 16 and len(self.array)
        rc = load_library().esvio_fe_register_host_buffer(_p(self.array), self.array.nbytes)
        if rc:
            raise FrontendError("esvio_fe_register_host_buffer rc=%d" % rc)
        self.registered = True

    def free(self):
        if self.registered:
            load_library().esvio_fe_unregister_host_buffer(_p(self.array))
            self.registered = False


def host_nullspace(systems, lanes):
    """run7Point's null-space basis of n 7x9 systems -> (f[n, 2, 9], systems redone one at a time);
    lanes False: the one-at-a-time routine, True: the vector-lane form the RANSAC loop uses"""
    a = np.ascontiguousarray(systems, np.float64).reshape(-1, 7, 9)
    f = np.empty((a.shape[0], 2, 9), np.float64)
    redone = C.c_int32(0)
    rc = load_library().esvio_fe_host_nullspace(_p(a), a.shape[0], int(bool(lanes)), _p(f), C.byref(redone))
    if rc != 0:
        raise FrontendError("host_nullspace rc=%d" % rc)
    return f, redone.value


def host_hypot(x, y):
    """the hypot of the 7-point solver's Jacobi rotations: cv::hypot of OpenCV's lapack.cpp, a*sqrt(1+(b/a)^2)"""
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    out = np.empty_like(x)
    rc = load_library().esvio_fe_host_hypot(_p(x), _p(y), x.size, _p(out))
    if rc != 0:
        raise FrontendError("host_hypot rc=%d" % rc)
    return out


def ransac_stats(reset=False):
    """process-wide counters of the host findFundamentalMat: RANSAC branch dict(calls, iterations,
    points, us) + LMedS branch (lmeds_calls, lmeds_us)"""
    out = (C.c_uint64 * 6)()
    L = load_library()
    rc = L.esvio_fe_ransac_stats(out, 1 if reset else 0)
    if rc != 0:
        raise FrontendError("ransac_stats rc=%d" % rc)
    return {"calls": int(out[0]), "iterations": int(out[1]), "points": int(out[2]), "us": out[3] / 1e3,
            "lmeds_calls": int(out[4]), "lmeds_us": out[5] / 1e3}


def ransac_tail(reset=False):
    """the tail of the host findFundamentalMat (process-wide): slowest RANSAC / LMedS call [us], iterations the
    calling thread redid for a helper that did not deliver, jobs run without the helpers, jobs that took the
    other job buffer because a helper was stuck in theirs, involuntary context switches of the helper threads"""
    out = (C.c_uint64 * 6)()
    rc = load_library().esvio_fe_ransac_tail(out, 1 if reset else 0)
    if rc != 0:
        raise FrontendError("ransac_tail rc=%d" % rc)
    return {"max_us": out[0] / 1e3, "lmeds_max_us": out[1] / 1e3, "redone_iterations": int(out[2]),
            "solo_jobs": int(out[3]), "skipped_buffers": int(out[4]), "helper_invol_switches": int(out[5])}


def find_fundamental_mat(p1, p2, thr=1.0, conf=0.99, threads=1, hold_mask=None):
    """cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, conf, status) (host-side stage).  hold_mask (test tap,
    threads >= 2): job buffers of the helper pool marked as still holding a descheduled helper"""
    L = load_library()
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = p1.shape[0]
    status = np.zeros(n, np.uint8)
    k = C.c_int32(0)
    if hold_mask is not None:
        rc = L.esvio_fe_find_fundamental_mat_held(_p(p1), _p(p2), n, thr, conf, threads, int(hold_mask), _p(status),
                                                  C.byref(k))
    elif threads > 1:
        rc = L.esvio_fe_find_fundamental_mat_mt(_p(p1), _p(p2), n, thr, conf, threads, _p(status),
                                                C.byref(k))
    else:
        rc = L.esvio_fe_find_fundamental_mat(_p(p1), _p(p2), n, thr, conf, _p(status), C.byref(k))
    if rc:
        raise FrontendError("find_fundamental_mat rc=%d" % rc)
    return k.value, status


def find_fundamental_mat_idle(p1, p2, thr=1.0, conf=0.99, threads=4, repeats=4, idle_units=64):
    """test tap: findFundamentalMat on a helper pool whose helpers are handed other work between jobs (what the
    host-batch staging does) -> (inliers, status, {idle_calls, idle_done, idle_left})"""
    L = load_library()
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = p1.shape[0]
    status = np.zeros(n, np.uint8)
    k = C.c_int32(0)
    out = (C.c_uint64 * 3)()
    rc = L.esvio_fe_find_fundamental_mat_idle(_p(p1), _p(p2), n, thr, conf, threads, repeats, idle_units, _p(status),
                                              C.byref(k), out)
    if rc:
        raise FrontendError("find_fundamental_mat_idle rc=%d" % rc)
    return k.value, status, dict(idle_calls=int(out[0]), idle_done=int(out[1]), idle_left=int(out[2]))


def lift_projective(cam, u, v):
    L = load_library()
    c = Camera(**{k: float(cam[k]) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")})
    out = np.empty(3, np.float64)
    rc = L.esvio_fe_lift_projective(C.byref(c), float(u), float(v), _p(out))
    if rc:
        raise FrontendError("lift_projective rc=%d" % rc)
    return out
