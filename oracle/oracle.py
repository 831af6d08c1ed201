"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from esvio_amd/.  PARITY UNPINNED (see esvio_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("esvio_oracle.cpp", "esvio_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


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
        ("focal_length", C.c_int32), ("_pad", C.c_int32),
        ("cam", Camera * 2),
    ]


class Motion(C.Structure):
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


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
        L.oracle_detector_create.restype = vp
        L.oracle_detector_create.argtypes = [i, i, d, i, d, i]
        L.oracle_detector_destroy.argtypes = [vp]
        L.oracle_detector_reset.argtypes = [vp]
        L.oracle_detector_set_median.argtypes = [vp, i]
        L.oracle_median_blur.argtypes = [vp, i, i, i]
        L.oracle_create_sae.restype = sz
        L.oracle_create_sae.argtypes = [vp, i, vp, sz]
        L.oracle_sae_to_time_surface.argtypes = [vp, i, d, vp]
        L.oracle_is_corner.restype = i
        L.oracle_is_corner.argtypes = [vp, d, i, i, i]
        L.oracle_corner_flags.argtypes = [vp, vp, sz, vp]
        L.oracle_get_sae.argtypes = [vp, i, vp, vp, vp, vp]
        L.oracle_set_sae.argtypes = [vp, i, vp, vp, vp, vp]
        L.oracle_disc_halfwidths.argtypes = [i, vp]
        L.oracle_circle_fill.argtypes = [vp, i, i, i, i, i, C.c_uint8]
        L.oracle_features_to_track.restype = i
        L.oracle_features_to_track.argtypes = [vp, vp, sz, i, i, vp, vp, d, vp, vp]
        L.oracle_pyr_down.argtypes = [vp, i, i, vp]
        L.oracle_scharr.argtypes = [vp, i, i, vp]
        L.oracle_pyr_levels.restype = i
        L.oracle_pyr_levels.argtypes = [i, i, i, i]
        L.oracle_lk.argtypes = [vp, vp, i, i, vp, vp, vp, i, i, i, i, d, i, i]
        L.oracle_lift_projective.argtypes = [C.POINTER(Camera), d, d, vp]
        L.oracle_clahe.argtypes = [vp, i, i, vp]
        L.oracle_normalize_minmax.argtypes = [vp, sz]
        L.oracle_find_fundamental_ransac.restype = i
        L.oracle_find_fundamental_ransac.argtypes = [vp, vp, i, d, d, vp, vp]
        L.oracle_set_nullspace_mode.argtypes = [i]
        L.oracle_svd_rows.restype = i
        L.oracle_svd_rows.argtypes = [vp, i, i, i, vp]
        L.oracle_seven_point.restype = i
        L.oracle_seven_point.argtypes = [vp, vp, vp]
        L.oracle_tracker_create.restype = vp
        L.oracle_tracker_create.argtypes = [C.POINTER(Config)]
        L.oracle_tracker_destroy.argtypes = [vp]
        L.oracle_track_event.restype = i
        L.oracle_track_event.argtypes = [vp, d, vp, sz, vp, sz, i, C.POINTER(Tracks)]
        L.oracle_track_event_mc.restype = i
        L.oracle_track_event_mc.argtypes = [vp, d, vp, sz, vp, sz, i, C.POINTER(Motion), C.POINTER(Tracks)]
        L.oracle_create_sae_mc.restype = sz
        L.oracle_create_sae_mc.argtypes = [vp, i, vp, sz, vp, C.POINTER(Motion)]
        L.oracle_matrix_exp3f.argtypes = [vp, vp]
        L.oracle_tracker_time_surface.argtypes = [vp, i, vp]
        L.oracle_tracker_detector.restype = vp
        L.oracle_tracker_detector.argtypes = [vp]
        L.oracle_tracker_stage_seconds.argtypes = [vp, vp]
        L.oracle_good_features_to_track.restype = i
        L.oracle_good_features_to_track.argtypes = [vp, i, i, i, d, d, vp, vp, vp, vp]
        L.oracle_euclid_halfwidths.argtypes = [i, vp]
        L.oracle_track_image.restype = i
        L.oracle_track_image.argtypes = [vp, d, vp, vp, i, C.POINTER(Tracks)]
        L.oracle_lk_pair_stats.argtypes = [vp, i]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ev(ev):
    ev = np.ascontiguousarray(ev)
    assert ev.dtype.itemsize == 16
    return ev


# How calcOpticalFlowPyrLK's sums are accumulated when a caller does not say: 2 = in float, in the order of the
# x86 OpenCV build the reference node links (feature_tracker.cpp:410,417-418,490,495) — the reference's own
# arithmetic; 1 = exact integer sums.  ESVIO_LK_ACCUM=1 in the environment runs everything that does not say
# (the test suite, tools/) in the exact mode instead.
DEFAULT_LK_ACCUM = int(os.environ.get("ESVIO_LK_ACCUM", "2"))


def make_config(W, H, **kw):
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
    cams = kw.get("cams")
    if cams is None:
        cams = [dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2.0, cy=H / 2.0, k1=-0.05, k2=0.01, p1=1e-4, p2=-2e-4)] * 2
    for k in range(2):
        for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2"):
            setattr(c.cam[k], n, float(cams[k][n]))
    return c


class Detector:
    """esvio::EventDetector restatement (event_detector.cc)."""

    def __init__(self, W, H, decay_ms=20.0, ignore_polarity=0, filter_threshold=0.01, min_dist=10,
                 handle=None, median_blur_kernel_size=0):
        self.W, self.H = W, H
        self._own = handle is None
        self.h = handle if handle is not None else lib().oracle_detector_create(
            W, H, decay_ms, ignore_polarity, filter_threshold, min_dist)
        if handle is None and median_blur_kernel_size:
            lib().oracle_detector_set_median(self.h, int(median_blur_kernel_size))

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().oracle_detector_destroy(self.h)
            self.h = None

    def reset(self):
        lib().oracle_detector_reset(self.h)

    def create_sae(self, cam, ev):
        ev = _ev(ev)
        return lib().oracle_create_sae(self.h, cam, _p(ev), ev.shape[0])

    def create_sae_mc(self, cam, ev, first_left, motion):
        ev = _ev(ev)
        first_left = _ev(first_left)
        return lib().oracle_create_sae_mc(self.h, cam, _p(ev), ev.shape[0], _p(first_left),
                                          C.byref(motion))

    def time_surface(self, cam, t_sync):
        out = np.empty((self.H, self.W), np.uint8)
        lib().oracle_sae_to_time_surface(self.h, cam, float(t_sync), _p(out))
        return out

    def is_corner(self, et, ex, ey, ep):
        return bool(lib().oracle_is_corner(self.h, float(et), int(ex), int(ey), int(ep)))

    def corner_flags(self, ev):
        ev = _ev(ev)
        out = np.empty(ev.shape[0], np.uint8)
        lib().oracle_corner_flags(self.h, _p(ev), ev.shape[0], _p(out))
        return out

    def get_sae(self, cam):
        planes = [np.empty((self.H, self.W), np.float64) for _ in range(4)]
        lib().oracle_get_sae(self.h, cam, *[_p(a) for a in planes])
        return planes  # L0, L1, S0, S1

    def set_sae(self, cam, L0, L1, S0, S1):
        arrs = [np.ascontiguousarray(a, np.float64) for a in (L0, L1, S0, S1)]
        lib().oracle_set_sae(self.h, cam, *[_p(a) for a in arrs])

    def features_to_track(self, ev, max_corners, min_dist, mask, ts, ts_lk_threshold=128.0):
        ev = _ev(ev)
        mask = np.ascontiguousarray(mask, np.uint8)
        ts = np.ascontiguousarray(ts, np.uint8)
        xy = np.empty((max(max_corners, 1), 2), np.float32)
        idx = np.empty(max(max_corners, 1), np.int32)
        n = lib().oracle_features_to_track(self.h, _p(ev), ev.shape[0], max_corners, min_dist,
                                           _p(mask), _p(ts), ts_lk_threshold, _p(xy), _p(idx))
        return xy[:n].copy(), idx[:n].copy()


def disc_halfwidths(r):
    hw = np.empty(r + 1, np.int32)
    lib().oracle_disc_halfwidths(r, _p(hw))
    return hw


def circle_fill(img, cx, cy, r, v=255):
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    H, W = img.shape
    lib().oracle_circle_fill(_p(img), W, H, cx, cy, r, v)


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().oracle_pyr_down(_p(img), w, h, _p(out))
    return out


def scharr(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().oracle_scharr(_p(img), w, h, _p(out))
    return out


def pyr_levels(w, h, win=21, max_level=3):
    return lib().oracle_pyr_levels(w, h, win, max_level)


def lk(prev, nxt, prev_pts, next_pts=None, win=21, max_level=3, max_count=30, eps=0.01, flags=0,
       accum=1):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    h, w = prev.shape
    prev_pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    n = prev_pts.shape[0]
    if next_pts is None:
        next_pts = np.zeros((n, 2), np.float32)
    else:
        next_pts = np.array(next_pts, np.float32).reshape(-1, 2).copy()
    status = np.zeros(n, np.uint8)
    lib().oracle_lk(_p(prev), _p(nxt), w, h, _p(prev_pts), _p(next_pts), _p(status), n, win,
                    max_level, max_count, eps, flags, accum)
    return next_pts, status


def median_blur(img, ksize):
    """cv::medianBlur(img, ksize) on u8 (BORDER_REPLICATE)"""
    out = np.ascontiguousarray(img, np.uint8).copy()
    lib().oracle_median_blur(_p(out), out.shape[1], out.shape[0], int(ksize))
    return out


def clahe(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty_like(img)
    lib().oracle_clahe(_p(img), w, h, _p(out))
    return out


def normalize_minmax(img):
    out = np.ascontiguousarray(img, np.uint8).copy()
    lib().oracle_normalize_minmax(_p(out), out.size)
    return out


def good_features_to_track(img, max_corners, quality=0.01, min_distance=30, mask=None, want_eig=False):
    """cv::goodFeaturesToTrack restatement (blockSize 3, gradientSize 3, Shi-Tomasi);
    returns corners (n,2) float32 [, cornerMinEigenVal map]"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((max(max_corners, 1) if max_corners > 0 else w * h, 2), np.float32)
    n = C.c_int32()
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    eig = np.zeros((h, w), np.float32) if want_eig else None
    lib().oracle_good_features_to_track(_p(img), w, h, int(max_corners), float(quality),
                                        float(min_distance), None if m is None else _p(m), _p(out),
                                        C.byref(n), None if eig is None else _p(eig))
    return (out[:n.value].copy(), eig) if want_eig else out[:n.value].copy()


def euclid_halfwidths(r):
    hw = np.zeros(r + 1, np.int32)
    lib().oracle_euclid_halfwidths(int(r), _p(hw))
    return hw


def lift_projective(cam, u, v):
    c = Camera(**{k: float(cam[k]) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")})
    out = np.empty(3, np.float64)
    lib().oracle_lift_projective(C.byref(c), float(u), float(v), _p(out))
    return out


def set_nullspace_mode(mode):
    """0: cv::SVDecomp's Jacobi route (default); 1: Householder QR (comparison only)"""
    lib().oracle_set_nullspace_mode(int(mode))


def svd_rows(a, n1=None):
    """One-sided Jacobi SVD of the n rows of `a` (n x m, m >= n) as cv::SVD runs it, completed to
    n1 orthonormal rows -> (rows[n1, m], w[n])"""
    a = np.ascontiguousarray(a, np.float64)
    n, m = a.shape
    n1 = n if n1 is None else n1
    at = np.zeros((n1, m), np.float64)
    at[:n] = a
    w = np.zeros(n, np.float64)
    if lib().oracle_svd_rows(_p(at), m, n, n1, _p(w)) != 0:
        raise ValueError("svd_rows: bad shape")
    return at, w


def seven_point(p1, p2):
    """run7Point on 7 pairs -> list of 3x3 F"""
    p1 = np.ascontiguousarray(p1, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    assert p1.shape == (7, 2) and p2.shape == (7, 2)
    F = np.zeros(27, np.float64)
    k = lib().oracle_seven_point(_p(p1), _p(p2), _p(F))
    return [F[9 * j:9 * j + 9].reshape(3, 3).copy() for j in range(max(k, 0))]


def find_fundamental(p1, p2, thr=1.0, conf=0.99):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = p1.shape[0]
    status = np.zeros(n, np.uint8)
    F = np.zeros(9, np.float64)
    cnt = lib().oracle_find_fundamental_ransac(_p(p1), _p(p2), n, thr, conf, _p(status), _p(F))
    return cnt, status, F.reshape(3, 3)


class TrackResult:
    pass


class Tracker:
    """FeatureTracker::trackEvent restatement (feature_tracker.cpp:340-603)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.h = lib().oracle_tracker_create(C.byref(cfg))
        if not self.h:
            raise ValueError("unsupported config")
        m = max(cfg.max_cnt, 1)
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

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_tracker_destroy(self.h)
            self.h = None

    def track_event(self, cur_time, left, right, pub_this_frame=True, motion=None):
        left, right = _ev(left), _ev(right)
        if motion is None:
            rc = lib().oracle_track_event(self.h, float(cur_time), _p(left), left.shape[0], _p(right),
                                          right.shape[0], int(pub_this_frame), C.byref(self._tr))
        else:
            rc = lib().oracle_track_event_mc(self.h, float(cur_time), _p(left), left.shape[0],
                                             _p(right), right.shape[0], int(pub_this_frame),
                                             C.byref(motion), C.byref(self._tr))
        if rc:
            raise RuntimeError("oracle_track_event rc=%d" % rc)
        r = TrackResult()
        nl, nr = self._tr.n_left, self._tr.n_right
        for k in ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity"):
            setattr(r, k, self._bufs[k][:nl].copy())
        for k in ("ids_right", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity"):
            setattr(r, k, self._bufs[k][:nr].copy())
        return r

    def _result(self):
        r = TrackResult()
        nl, nr = self._tr.n_left, self._tr.n_right
        for k in ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity"):
            setattr(r, k, self._bufs[k][:nl].copy())
        for k in ("ids_right", "cur_right_pts", "cur_un_right_pts", "right_pts_velocity"):
            setattr(r, k, self._bufs[k][:nr].copy())
        return r

    def track_image(self, cur_time, img_left, img_right, pub_this_frame=True):
        """FeatureTracker::trackImage (feature_tracker.cpp:164-338)"""
        il = np.ascontiguousarray(img_left, np.uint8)
        ir = None if img_right is None else np.ascontiguousarray(img_right, np.uint8)
        rc = lib().oracle_track_image(self.h, float(cur_time), _p(il), None if ir is None else _p(ir),
                                      int(pub_this_frame), C.byref(self._tr))
        if rc:
            raise RuntimeError("oracle_track_image rc=%d" % rc)
        return self._result()

    def time_surface(self, cam):
        out = np.empty((self.cfg.height, self.cfg.width), np.uint8)
        lib().oracle_tracker_time_surface(self.h, cam, _p(out))
        return out

    def detector(self):
        return Detector(self.cfg.width, self.cfg.height, handle=lib().oracle_tracker_detector(self.h))

    def stage_seconds(self):
        out = np.zeros(6, np.float64)
        lib().oracle_tracker_stage_seconds(self.h, _p(out))
        return dict(zip(("sae", "ts", "lk_temporal", "detect", "lk_stereo", "host"), out))


def lk_pair_stats(reset=False):
    """iterations per forward+backward pair of LK calls the trackers made (what one fused GPU launch
    runs per point): dict(pairs, slowest_mean, slowest_max, mean_per_point)"""
    out = np.zeros(5, np.uint64)
    lib().oracle_lk_pair_stats(_p(out), int(reset))
    pairs, mx_sum, pts, it_sum, mx_mx = (int(v) for v in out)
    return dict(pairs=pairs, slowest_mean=mx_sum / max(pairs, 1), slowest_max=mx_mx,
                mean_per_point=it_sum / max(pts, 1))
