"""Host harness mirroring the caller of trackEvent: ``handle_stereo_event`` of the reference's
stereo_event_tracker node (feature_tracker/src/stereo_event_tracker_node.cpp:145-344) without ROS:
first-frame drop, stream-discontinuity re-arm, publish-rate control, PointCloud-equivalent packing.

Plain host bookkeeping over the FeatureTracker mirror; no compute happens here.  The C++ twin of
this file is tools/replay_node.cpp (same rules, drives the C ABI directly).
"""
import math

import numpy as np


def c_round(x):
    """C `round()` (half away from zero), which the node's frequency control uses (node:177);
    Python's round() is half-to-even and differs when pub_count/dt lands on x.5 with even x."""
    a = abs(x)
    r = math.floor(a)
    if a - r >= 0.5:  # (a - floor(a) is exact in binary floating point; a + 0.5 is not)
        r += 1
    return int(math.copysign(r, x))


def c_rate(pub_count, dt):
    """`1.0 * pub_count / dt` as C computes it (IEEE division: x/0 is +-inf, 0/0 is NaN)"""
    dt = float(dt)
    if dt == 0.0:
        return math.nan if pub_count == 0 else math.copysign(math.inf, pub_count)
    return 1.0 * pub_count / dt


def rate_allows(rate, freq):
    """`round(rate) <= FREQ` (node:177): false for inf and NaN — a message whose stamp equals
    first_image_time (a duplicate stamp right after the first frame or after a frequency-control reset)
    is simply not published, as in the reference and in tools/replay_node.cpp"""
    return math.isfinite(rate) and c_round(rate) <= freq

from .events import event_times

NUM_OF_CAM_stereo = 2


class FreqControl:
    """PUB_THIS_FRAME logic of stereo_event_tracker_node.cpp:155-188 (first frame handled by the
    caller): publish while round(pub_count / (t - first_image_time)) <= FREQ, re-arm the window
    when the running rate is within 1 % of FREQ."""

    def __init__(self, freq):
        self.FREQ = freq if freq != 0 else 100  # parameters.cpp:278-279
        self.first_image_time = None
        self.pub_count = 1

    def pub_this_frame(self, msg_timestamp):
        if self.first_image_time is None:
            self.first_image_time = msg_timestamp
            return False
        dt = msg_timestamp - self.first_image_time
        rate = c_rate(self.pub_count, dt)
        if rate_allows(rate, self.FREQ):
            if abs(rate - self.FREQ) < 0.01 * self.FREQ:
                self.first_image_time = msg_timestamp
                self.pub_count = 0
            return True
        return False

    def published(self):
        self.pub_count += 1

    def peek(self, msg_timestamp):
        """the decision pub_this_frame(msg_timestamp) would take, without taking it (the rule reads
        timestamps only, so a replaying caller can hint the next frame's PUB_THIS_FRAME)"""
        if self.first_image_time is None:
            return False
        dt = msg_timestamp - self.first_image_time
        return rate_allows(c_rate(self.pub_count, dt), self.FREQ)


def pack_point_cloud(ft):
    """sensor_msgs/PointCloud equivalent (node:273-329): rows of
    (x_un, y_un, 1, id*2+cam as float32, u, v, vx, vy); left entries with track_cnt > 1 first, then
    right entries whose id is in the left set."""
    rows = []
    hash_ids = set()
    for j in range(len(ft.ids)):
        if ft.track_cnt[j] > 1:
            fid = int(ft.ids[j])
            hash_ids.add(fid)
            rows.append((ft.cur_un_pts[j, 0], ft.cur_un_pts[j, 1], 1.0,
                         np.float32(fid * NUM_OF_CAM_stereo + 0), ft.cur_pts[j, 0], ft.cur_pts[j, 1],
                         ft.pts_velocity[j, 0], ft.pts_velocity[j, 1]))
    for j in range(len(ft.ids_right)):
        fid = int(ft.ids_right[j])
        if fid in hash_ids:
            rows.append((ft.cur_un_right_pts[j, 0], ft.cur_un_right_pts[j, 1], 1.0,
                         np.float32(fid * NUM_OF_CAM_stereo + 1), ft.cur_right_pts[j, 0],
                         ft.cur_right_pts[j, 1], ft.right_pts_velocity[j, 0],
                         ft.right_pts_velocity[j, 1]))
    return np.asarray(rows, np.float32).reshape(-1, 8)


def pack_track_records(ft, max_cnt):
    """fixed-size (2*max_cnt, 8) float32 block of the PointCloud rows (padding rows have id -1):
    the unit the multi-GPU all_gather exchanges."""
    rec = np.zeros((2 * max_cnt, 8), np.float32)
    rec[:, 3] = -1.0
    pc = pack_point_cloud(ft)
    rec[:len(pc)] = pc[:2 * max_cnt]
    return rec


class MotionCorrection:
    """The node's side of motion compensation (Do_motion_correction: 1): imu_callback / state_callback
    (stereo_event_tracker_node.cpp:102-125) fill two queues, handle_stereo_event (node:195-252)
    builds the Motion_correction_value of a batch from them.  ``value(...)`` returns the arguments of
    ``frontend.make_motion`` / ``oracle.make_motion``.

    Where the reference leaves a local uninitialised — State_ and temp_a when the back end has sent no
    new odometry, omega_avg_ when there is no IMU message at or after the batch's first event — the
    field is zero here; zero acceleration is below the warp's threshold."""

    def __init__(self, fx, fy, cx, cy):
        self.K = (float(fx), float(fy), float(cx), float(cy))  # the YAML's fx, fy, cx, cy (parameters.cpp:221-224)
        self.imu_buf = []
        self.odom_buffer_ = []
        self.last_imu_t = 0.0
        self.v_cur = np.zeros(3, np.float32)  # file-scope Eigen::Vector3f (node:48-49): zero at start
        self.v_pre = np.zeros(3, np.float32)
        self.t_pre = 0.0
        self.t_cur = 0.0

    def imu_callback(self, stamp, angular_velocity, linear_acceleration=(0.0, 0.0, 0.0)):
        if stamp <= self.last_imu_t:  # "imu message in disorder!"
            return
        self.last_imu_t = stamp
        self.imu_buf.append((float(stamp), tuple(map(float, angular_velocity)), tuple(map(float, linear_acceleration))))

    def state_callback(self, stamp, linear_velocity):
        self.odom_buffer_.append((float(stamp), tuple(map(float, linear_velocity))))

    def value(self, t_left_0, t_left_1):
        v = np.zeros(3, np.float64)
        accel = np.zeros(3, np.float32)
        omega = np.zeros(3, np.float32)
        if self.imu_buf:
            if self.odom_buffer_:
                t, vel = self.odom_buffer_.pop(0)
                v[:] = vel
                self.v_pre = self.v_cur.copy()
                self.v_cur = v.astype(np.float32)
                self.t_pre, self.t_cur = self.t_cur, t
                with np.errstate(divide="ignore", invalid="ignore"):
                    # (float - float) / double, stored to a float
                    accel = ((self.v_cur - self.v_pre).astype(np.float64) / (self.t_cur - self.t_pre)).astype(np.float32)
            while self.imu_buf and self.imu_buf[0][0] < t_left_0:
                self.imu_buf.pop(0)
            if self.imu_buf:
                omega = np.asarray(self.imu_buf[0][1], np.float64).astype(np.float32)
        return dict(t1=float(t_left_1), v=tuple(v), v_pre=tuple(self.v_pre), accel=tuple(accel),
                    omega=tuple(omega), fx=self.K[0], fy=self.K[1], cx=self.K[2], cy=self.K[3])


class StereoEventTrackerNode:
    """handle_stereo_event (node:145-344).  ``handle(left, right, msg_timestamp)`` returns the
    published PointCloud rows or None (first frame, reset, non-published frame, swallowed first
    publish).  With ``motion`` (a MotionCorrection and the module whose ``make_motion`` builds the
    tracker's argument) the batch takes the motion-compensated trackEvent (node:194-254)."""

    def __init__(self, tracker, freq, reset_tracker_on_restart=False, motion=None, make_motion=None):
        self.trackerData = tracker
        self.motion = motion
        self.make_motion = make_motion
        self.freq = freq
        # NOT reference behaviour (kept for callers that own both ends): also clear the tracker when
        # the stream is discontinuous.  The reference only re-arms the node and publishes `restart`.
        self.reset_tracker_on_restart = reset_tracker_on_restart
        self.restart_flag = False  # what pub_restart would have carried for the latest call
        self.first_image_flag = True
        self.first_image_time = 0.0
        self.last_image_time = 0.0
        self.pub_count = 1
        self.init_pub = False
        self.restart_count = 0
        self.FREQ = freq if freq != 0 else 100

    def handle(self, event_left, event_right, msg_timestamp, header_stamp=None):
        """header_stamp: event_left.header.stamp (only read with motion compensation; default
        msg_timestamp, which is what sync_process passes: the same stamp)"""
        self.restart_flag = False
        if len(event_left) == 0:  # node:150
            return None
        if self.first_image_flag:  # node:155-161
            self.first_image_flag = False
            self.first_image_time = msg_timestamp
            self.last_image_time = msg_timestamp
            return None
        if msg_timestamp - self.last_image_time > 1.0 or msg_timestamp < self.last_image_time:
            self.first_image_flag = True  # node:163-173
            self.last_image_time = 0
            self.pub_count = 1
            self.restart_count += 1
            # the reference publishes restart_flag and returns: the tracker keeps its SAE planes,
            # tracks, ids and previous image across the gap
            self.restart_flag = True
            if self.reset_tracker_on_restart:
                self.trackerData.reset()
            return None
        self.last_image_time = msg_timestamp
        rate = c_rate(self.pub_count, msg_timestamp - self.first_image_time)
        if rate_allows(rate, self.FREQ):  # node:177-188
            pub = True
            if abs(rate - self.FREQ) < 0.01 * self.FREQ:
                self.first_image_time = msg_timestamp
                self.pub_count = 0
        else:
            pub = False
        msg_timestamp_left = event_times(event_left[-1:])[0]  # node:190
        if self.motion is None:
            self.trackerData.trackEvent(msg_timestamp_left, event_left, event_right, pub)
        else:
            t_left_1 = msg_timestamp if header_stamp is None else header_stamp
            mv = self.make_motion(**self.motion.value(event_times(event_left[:1])[0], t_left_1))
            self.trackerData.trackEvent(msg_timestamp_left, event_left, event_right, pub, measurements=mv)
        if not pub:
            return None
        self.pub_count += 1
        pc = pack_point_cloud(self.trackerData)
        if not self.init_pub:  # node:334-339: the first publishable frame is swallowed
            self.init_pub = True
            return None
        return pc


class StereoImageTrackerNode:
    """handle_stereo_image (stereo_image_tracker_node.cpp:54-183): same first-frame / discontinuity
    / publish-rate logic as the event node, `trackImage` instead of `trackEvent` (the tracker's own
    `equalize` setting stands for the node's CLAHE, :92-96).  A discontinuity only re-arms the node
    (:68-78: the restart flag is published, the tracker is not reset).  ``handle(img_left,
    img_right, msg_timestamp)`` returns the published PointCloud rows or None."""

    def __init__(self, tracker, freq):
        self.trackerData = tracker
        self.first_image_flag = True
        self.first_image_time = 0.0
        self.last_image_time = 0.0
        self.pub_count = 1
        self.init_pub = False
        self.restart_count = 0
        self.restart_flag = False
        self.FREQ = freq if freq != 0 else 100

    def handle(self, img_left, img_right, msg_timestamp):
        self.restart_flag = False
        if self.first_image_flag:  # :58-64
            self.first_image_flag = False
            self.first_image_time = msg_timestamp
            self.last_image_time = msg_timestamp
            return None
        if msg_timestamp - self.last_image_time > 1.0 or msg_timestamp < self.last_image_time:
            self.first_image_flag = True  # :66-78
            self.last_image_time = 0
            self.pub_count = 1
            self.restart_count += 1
            self.restart_flag = True
            return None
        self.last_image_time = msg_timestamp
        rate = c_rate(self.pub_count, msg_timestamp - self.first_image_time)
        if rate_allows(rate, self.FREQ):  # :81-91
            pub = True
            if abs(rate - self.FREQ) < 0.01 * self.FREQ:
                self.first_image_time = msg_timestamp
                self.pub_count = 0
        else:
            pub = False
        self.trackerData.trackImage(msg_timestamp, img_left, img_right, pub)  # :100
        if not pub:
            return None
        self.pub_count += 1  # :113-115
        pc = pack_point_cloud(self.trackerData)
        if not self.init_pub:  # :171-176: the first publishable frame is swallowed
            self.init_pub = True
            return None
        return pc
