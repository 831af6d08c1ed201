/*
 * esvio_fe.h — C ABI of the MI355X-native ESVIO event front-end (libesvio_fe.so).
 *
 * Drop-in boundary for the hot path of arclab-hku/ESVIO's `feature_tracker` event node:
 * the calls FeatureTracker::trackEvent makes into esvio::EventDetector and OpenCV, and the
 * result vectors the ROS node reads back.  The reference has no FFI of its
 * own; each entry point below names the reference interface it replaces (file:line relative
 * to the reference tree).  Plain pointers and sizes only — no C++/torch types.
 *
 * Threading: one caller per handle (the reference has exactly one worker thread,
 * feature_tracker/src/stereo_event_tracker_node.cpp:366).  Different handles are independent
 * (one per GPU / per rig).  All functions return 0 on success, <0 on error
 * (see ESVIO_FE_E*), never abort; esvio_fe_last_error() gives a message.
 *
 * Every compute entry point runs hand-written HIP kernels on the handle's device; there is no
 * CPU fallback — creating a handle without a usable GPU fails with ESVIO_FE_ENODEVICE.
 */
#ifndef ESVIO_FE_H
#define ESVIO_FE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESVIO_FE_OK 0
#define ESVIO_FE_EINVAL (-1)      /* bad argument / unsupported config value */
#define ESVIO_FE_ENODEVICE (-2)   /* no usable HIP device */
#define ESVIO_FE_EHIP (-3)        /* HIP runtime error */
#define ESVIO_FE_ENOTIMPL (-4)    /* config asks for a stage that is not built yet */
#define ESVIO_FE_EINTERNAL (-5)   /* device-side invariant violated (bounded spin expired …) */

/* dvs_msgs::Event as laid out in memory (feature_tracker/src/dvs_msgs/Event.h:42-52):
 * uint16 x; uint16 y; ros::Time ts {uint32 sec; uint32 nsec}; uint8 polarity => 16 B AoS.
 * A `std::vector<dvs_msgs::Event>::data()` pointer can be passed as-is. */
typedef struct esvio_fe_event {
  uint16_t x, y;
  uint32_t sec, nsec;
  uint8_t polarity;
  uint8_t _pad[3];
} esvio_fe_event;

/* camodocal pinhole + radtan parameters (camera_model/src/camera_models/PinholeCamera.cc) */
typedef struct esvio_fe_camera {
  double fx, fy, cx, cy, k1, k2, p1, p2;
} esvio_fe_camera;

/* The YAML knobs readParameters_event loads into globals
 * (feature_tracker/src/parameters.cpp:183-282), as one plain struct. */
typedef struct esvio_fe_config {
  int32_t width, height;            /* event_width / event_height -> COL_event, ROW_event */
  double decay_ms;                  /* decay_ms */
  int32_t ignore_polarity;          /* ignore_polarity */
  int32_t median_blur_kernel_size;  /* k: cv::medianBlur(2k+1) of each surface; 0 in every shipped config; k <= 7 */
  double feature_filter_threshold;  /* feature_filter_threshold [s] */
  double ts_lk_threshold;           /* TS_LK_threshold (128.0) */
  int32_t max_cnt;                  /* max_cnt */
  int32_t min_dist;                 /* min_dist */
  int32_t flow_back;                /* flow_back */
  int32_t equalize;                 /* equalize: 1 = CLAHE(40, 8x8) + normalize(0,255) before LK */
  double f_threshold;               /* F_threshold [px] */
  int32_t f_ransac;                 /* 1: run rejectWithF_event's RANSAC on host; 0: skip */
  int32_t lk_accum;                 /* how calcOpticalFlowPyrLK's sums are accumulated: 2 = in float in the
                                     * order of OpenCV 4.2's x86 SIMD128 build, i.e. the reference's own build
                                     * (restated from recall) — the default of every caller in this repository;
                                     * 1 = exactly (int64 sums; 0.65-0.8x the LK time, not that build's
                                     * arithmetic: within 1e-4 px of it on ~97 % of the points) */
  int32_t focal_length;             /* FOCAL_LENGTH, 460 (parameters.cpp:274) */
  int32_t device;                   /* HIP device ordinal, -1 = current device */
  esvio_fe_camera cam[2];           /* event_left_calib / event_right_calib */
} esvio_fe_config;

/* What stereo_event_tracker_node.cpp:289-322 reads from FeatureTracker after trackEvent.
 * Caller-owned buffers, each sized for max_cnt entries (xy arrays: 2*max_cnt floats). */
typedef struct esvio_fe_tracks {
  int32_t n_left;             /* ids.size() */
  int32_t n_right;            /* ids_right.size() */
  int32_t* ids;               /* FeatureTracker::ids */
  int32_t* track_cnt;         /* ::track_cnt */
  float* cur_pts;             /* ::cur_pts (u,v) */
  float* cur_un_pts;          /* ::cur_un_pts */
  float* pts_velocity;        /* ::pts_velocity */
  int32_t* ids_right;         /* ::ids_right */
  float* cur_right_pts;       /* ::cur_right_pts */
  float* cur_un_right_pts;    /* ::cur_un_right_pts */
  float* right_pts_velocity;  /* ::right_pts_velocity */
} esvio_fe_tracks;

typedef struct esvio_fe_ctx* esvio_fe_handle;

/* ---- lifetime ------------------------------------------------------------------------ */
/* replaces: global `esvio::EventDetector detector` (feature_tracker.cpp:7), `trackerData`
 * (stereo_event_tracker_node.cpp:45), EventDetector::init (event_detector.cc:47-70). */
/* Limits (ESVIO_FE_EINVAL beyond them): 42 <= width, height <= 8192; 1 <= max_cnt <= 65536;
 * 3 <= min_dist <= 63; median_blur_kernel_size <= 7.  Up to ~1.3 M pixels (1280x960) the greedy
 * corner selections keep their one-bit-per-pixel map in LDS; above that (the frame cameras of the
 * shipped ESVIO configs go up to 1920x1200) the map lives in device memory: same results, a slower
 * selection. */
int esvio_fe_create(const esvio_fe_config* cfg, esvio_fe_handle* out);
int esvio_fe_destroy(esvio_fe_handle h);
/* Clears SAE planes, images, tracks and ids as a freshly created handle (n_id keeps counting).
 * NOT what the reference does on a stream discontinuity: stereo_event_tracker_node.cpp:163-173 only
 * re-arms the node's own flags and publishes `restart`, the tracker keeps everything — a drop-in
 * caller does not call this there (tools/replay_node.cpp, esvio_amd/node.py). */
int esvio_fe_reset(esvio_fe_handle h);
const char* esvio_fe_last_error(esvio_fe_handle h);
const char* esvio_fe_version(void);

/* ---- EventDetector stages ------------------------------------------------------------ */
/* memory space of an event pointer argument */
#define ESVIO_FE_HOST 0
#define ESVIO_FE_DEVICE 1

/* createSAE_left (cam 0, event_detector.cc:149-166) / createSAE_right (cam 1, :212-228)
 * applied to n events in stream order.  Events with x>=width or y>=height are skipped and
 * counted in *n_rejected (the reference would abort on an Eigen assert). */
int esvio_fe_create_sae(esvio_fe_handle h, int cam, const esvio_fe_event* ev, size_t n,
                        int space, uint64_t* n_rejected);
/* both cameras in one submission (what trackEvent does at feature_tracker.cpp:356-362) */
int esvio_fe_create_sae_stereo(esvio_fe_handle h, const esvio_fe_event* left, size_t nL,
                               const esvio_fe_event* right, size_t nR, int space,
                               uint64_t* n_rejected);
/* SAEtoTimeSurface_left/right (event_detector.cc:230-305).  Renders the camera's time surface
 * (and, with equalize, the CLAHE+normalize image trackEvent derives from it, feature_tracker.cpp:
 * 375-382) into the handle; if out != NULL also copies the raw width*height u8 surface to host. */
int esvio_fe_sae_to_time_surface(esvio_fe_handle h, int cam, double external_sync_time,
                                 uint8_t* out);
/* isCorner (event_detector.cc:308-544) for n events against the LEFT planes; flags[i] in {0,1}
 * (host buffer).  Out-of-sensor events give 0. */
int esvio_fe_is_corner(esvio_fe_handle h, const esvio_fe_event* ev, size_t n, int space,
                       uint8_t* flags);
/* Event_FeaturesToTrack (feature_tracker.cpp:13-38): greedy scan of `ev` in stream order.
 * mask: width*height bytes on host, 255 = blocked (the reference's CV_64F 255.0), may be NULL.
 * Uses the handle's current left time surface for the TS_LK_threshold test.  Writes up to
 * max_corners (x,y) pairs and (optionally) their event indices. */
int esvio_fe_features_to_track(esvio_fe_handle h, const esvio_fe_event* ev, size_t n, int space,
                               int max_corners, const uint8_t* mask, float* out_xy,
                               int32_t* out_idx, int32_t* n_out);
/* read-out of a camera's four planes (each width*height doubles, index x + y*width): what the reference's
 * private sae_[2] / sae_latest_[2] (event_detector.h) hold; the write side (tests) is esvio_fe_set_sae in
 * esvio_fe_test.h */
int esvio_fe_get_sae(esvio_fe_handle h, int cam, double* L0, double* L1, double* S0, double* S1);

/* ---- OpenCV stages used by trackEvent ------------------------------------------------- */
#define ESVIO_FE_LK_USE_INITIAL_FLOW 4 /* cv::OPTFLOW_USE_INITIAL_FLOW */
/* cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(21,21), max_level,
 * TermCriteria(COUNT+EPS, max_count, eps), flags) as called at feature_tracker.cpp:410,417,
 * 490,495.  Host u8 images of w*h (w,h need not equal the config's sensor size). */
int esvio_fe_calc_optical_flow_pyr_lk(esvio_fe_handle h, const uint8_t* prev_img,
                                      const uint8_t* next_img, int w, int hgt,
                                      const float* prev_pts, float* next_pts, uint8_t* status,
                                      int n, int max_level, int max_count, double eps, int flags);
/* cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, conf, status) as used by rejectWithF_event
 * (feature_tracker.cpp:935); host-side. Returns the inlier count in *n_inliers. */
int esvio_fe_find_fundamental_mat(const float* p1, const float* p2, int n, double thr,
                                  double conf, uint8_t* status, int32_t* n_inliers);

/* ---- the fused per-frame call -------------------------------------------------------- */
/* FeatureTracker::trackEvent(cur_time, event_left, event_right) (feature_tracker.cpp:340-603)
 * with PUB_THIS_FRAME passed explicitly (global at parameters.cpp:276, set by
 * stereo_event_tracker_node.cpp:177-188).  nL must be > 0 (node:150 returns early otherwise).
 * `out` may be NULL. */
int esvio_fe_track_event(esvio_fe_handle h, double cur_time, const esvio_fe_event* left,
                         size_t nL, const esvio_fe_event* right, size_t nR, int space,
                         int pub_this_frame, esvio_fe_tracks* out);
/* The fields of Motion_correction_value (event_detector.h:16) that createSAE_left/right with
 * motion compensation read (event_detector.cc:102-147,168-210), as handle_stereo_event fills them
 * (stereo_event_tracker_node.cpp:192-254), plus the K that detector.init(COL,ROW,fx,fy,cx,cy)
 * receives (feature_tracker.cpp:616; the fx,fy,cx,cy globals of parameters.cpp:222-225). */
typedef struct esvio_fe_motion {
  double t1;       /* event_left.header.stamp.toSec() (feature_tracker.cpp:622) */
  double v[3];     /* State_[0..2]: current linear velocity (node:215-217) */
  float v_pre[3];  /* previous velocity (node:220-222) */
  float accel[3];  /* temp_a (node:230-232); the warp is applied when |accel| > 5 m/s^2 */
  float omega[3];  /* IMU angular velocity (node:244-246) */
  double fx, fy, cx, cy;
} esvio_fe_motion;

/* FeatureTracker::trackEvent(cur_time, event_left, event_right, measurements)
 * (feature_tracker.cpp:605-877, configs with Do_motion_correction: 1): events of the first part of
 * the batch are warped to the batch start before the SAE update; everything else as
 * esvio_fe_track_event. */
int esvio_fe_track_event_mc(esvio_fe_handle h, double cur_time, const esvio_fe_event* left,
                            size_t nL, const esvio_fe_event* right, size_t nR, int space,
                            int pub_this_frame, const esvio_fe_motion* motion,
                            esvio_fe_tracks* out);
/* the createSAE_left/right(…, measurements) loops alone (feature_tracker.cpp:627-641) */
int esvio_fe_create_sae_stereo_mc(esvio_fe_handle h, const esvio_fe_event* left, size_t nL,
                                  const esvio_fe_event* right, size_t nR, int space,
                                  const esvio_fe_motion* motion, uint64_t* n_rejected);

/* Throughput (replay) mode: announce the batch the FOLLOWING esvio_fe_track_event call will be
 * given.  The current call then enqueues that batch's SAE update, time surfaces and pyramids on a
 * second HIP stream as soon as the current frame has finished reading the SAE planes, so they overlap
 * the current frame's LK / selection and the host work between calls.  `pub_hint` is the
 * PUB_THIS_FRAME the caller expects to pass with that batch (the node's frequency control depends
 * on timestamps only, stereo_event_tracker_node.cpp:177-188): when non-zero the batch's Arc* pass
 * is prefetched as well; a wrong hint costs time, never correctness.  The following call must
 * pass exactly these pointers, sizes, space and cur_time (else ESVIO_FE_EINVAL) and the event memory
 * must stay valid until then.  Results are identical to the non-pipelined sequence.  After a call
 * that prefetched, the get_sae / time-surface taps already reflect the latest prefetched batch.
 * Up to three batches are taken up ahead of their track calls (the calls that follow must come in
 * the announced order): the later ones' SAE updates then run whole frames early; up to six may be
 * announced and not yet tracked (ESVIO_FE_EINVAL beyond).  Batches in ESVIO_FE_HOST memory start on their way to the
 * device inside this call: helper threads copy them into pinned chunks and enqueue the DMAs on a copy
 * stream of the handle (ESVIO_FE_STAGE_THREADS, default 2; a pinned source is DMA'd as it is), and a
 * track call takes up an announced batch only once that is done — so announce host batches one call
 * further ahead than device batches.  With more than one batch in flight the hint of a published frame must be exact
 * (the SAE has moved on by the time it is tracked): a published frame whose hint was 0 is refused
 * with ESVIO_FE_EINVAL.  With two batches' pyramids in flight and a next frame that publishes
 * nothing (hint 0: no new corners, no RANSAC — the frame after it tracks exactly its forward LK
 * results) the temporal LK of the frame AFTER next is launched together with the next frame's,
 * chained to it point by point on the device; a wrong hint only discards that launch.
 * The reference has no counterpart: it processes one batch at a time (depth-1 queues,
 * node:128-142). */
int esvio_fe_set_next_batch(esvio_fe_handle h, double next_cur_time, const esvio_fe_event* left,
                            size_t nL, const esvio_fe_event* right, size_t nR, int space,
                            int pub_hint);
/* The same for a batch that esvio_fe_track_event_mc will be given (configs with
 * Do_motion_correction: 1): the Motion_correction_value the node has assembled for it travels with the
 * announcement (copied), the following esvio_fe_track_event_mc call must pass the same values. */
int esvio_fe_set_next_batch_mc(esvio_fe_handle h, double next_cur_time, const esvio_fe_event* left,
                               size_t nL, const esvio_fe_event* right, size_t nR, int space,
                               int pub_hint, const esvio_fe_motion* motion);

/* ---- image front-end: FeatureTracker::trackImage ------------------------ */
/* For this path the handle is the image tracker's own instance (stereo_image_tracker_node.cpp:31):
 * width/height = image_width/image_height (parameters.cpp:103-104), max_cnt = max_cnt_img,
 * min_dist = min_dist_img (:100-102); equalize = the node's CLAHE (node:92-96).
 *
 * cv::goodFeaturesToTrack(img, corners, max_corners, quality, min_distance, mask) as trackImage
 * calls it (feature_tracker.cpp:228: blockSize 3, gradientSize 3, Shi-Tomasi) [OpenCV, restated].
 * img: width*height bytes (host); mask: width*height bytes, nonzero = allowed, or NULL;
 * 1 <= max_corners <= max_cnt; out_xy: 2*max_corners floats; eig_out (optional): the
 * cornerMinEigenVal map, width*height floats. */
int esvio_fe_good_features_to_track(esvio_fe_handle h, const uint8_t* img, int max_corners,
                                    double quality, double min_distance, const uint8_t* mask,
                                    float* out_xy, int32_t* n_out, float* eig_out);
/* FeatureTracker::trackImage(cur_time, img_left, img_right) (feature_tracker.cpp:164-338), host
 * images of width*height bytes; img_right may be NULL (the right block is then skipped like the
 * reference's `!img_right.empty()` test).  Results as esvio_fe_track_event. */
int esvio_fe_track_image(esvio_fe_handle h, double cur_time, const uint8_t* img_left,
                         const uint8_t* img_right, int pub_this_frame, esvio_fe_tracks* out);

/* The node's PointCloud packing of the current results (stereo_event_tracker_node.cpp:273-329):
 * out = 2*max_cnt rows of 8 floats (x_un, y_un, 1, id*2+cam as float32, u, v, vx, vy): left entries
 * with track_cnt > 1, then right entries whose id is among them, then padding rows with id -1.
 * This fixed-size block is the unit the multi-GPU all_gather exchanges. */
int esvio_fe_pack_track_records(esvio_fe_handle h, float* out, int32_t* n_rows);

/* The north-star's merge step from C/C++: one ncclAllGather (RCCL over xGMI) of this handle's
 * esvio_fe_pack_track_records block over `nccl_comm` (a ncclComm_t of `world` ranks whose rank uses
 * this handle's device), enqueued on the handle's stream; `gathered` (host) receives
 * world x 2*max_cnt x 8 floats, rank by rank, on every rank.  19.2 KB per rank at max_cnt 300:
 * latency-bound, xGMI bandwidth is irrelevant.  RCCL is dlopen'ed on first use
 * (ESVIO_FE_ENOTIMPL if librccl.so cannot be found).  Replaces: the ROS topic hop of
 * stereo_event_tracker_node.cpp:340 when several GPUs feed one estimator. */
int esvio_fe_exchange_tracks(esvio_fe_handle h, void* nccl_comm, int world, float* gathered);
/* The same exchange without a stop on the caller's thread, over a communicator the handle owns:
 * rank 0 makes an id (esvio_fe_comm_unique_id = ncclGetUniqueId) and passes it to the other ranks by
 * whatever side channel the launcher has; every rank calls esvio_fe_comm_init once
 * (ncclCommInitRank).  esvio_fe_exchange_begin packs the current frame's records and enqueues
 * upload + ncclAllGather + download on a stream of its own — it returns at once, the next frames'
 * kernels run beside it; esvio_fe_exchange_end waits for the latest begun exchange and copies the
 * world x 2*max_cnt x 8 floats out (gathered may be NULL: wait only).  One exchange in flight:
 * begin waits for the previous one. */
int esvio_fe_comm_unique_id(uint8_t id[128]);
int esvio_fe_comm_init(esvio_fe_handle h, const uint8_t id[128], int rank, int world);
int esvio_fe_exchange_begin(esvio_fe_handle h);
int esvio_fe_exchange_end(esvio_fe_handle h, float* gathered);
/* With `on`, every esvio_fe_track_event call with pub_this_frame != 0 exchanges its records by itself:
 * they are packed at the end of that call and the upload / ncclAllGather / download are enqueued by
 * the NEXT call at the point where it waits for its temporal LK anyway (or at its end), so the
 * exchange costs the calling thread no time of its own.  esvio_fe_exchange_end returns the latest
 * one (enqueuing it first if the next call has not come yet).  Every rank must publish the same
 * frames (the node's frequency control reads timestamps only). */
int esvio_fe_set_auto_exchange(esvio_fe_handle h, int on);

/* Throughput option: with `on`, a published esvio_fe_track_event call returns without waiting for
 * the stereo LK of the corners it has just detected.  Everything else in its results is complete
 * (the node's PointCloud never contains corners of track_cnt 1, node:289); the right-camera entries
 * of those new corners (ids_right / cur_right_pts / cur_un_right_pts / right_pts_velocity tails) are
 * appended by a later call's right-camera bookkeeping — the next call's, or, when that call itself
 * returns lazily while their stereo LK is still running, the one after it: always before they can
 * influence anything — or by esvio_fe_finish, after which the state is bit-identical to the eager sequence.
 * A call with pub_this_frame == 0 (the node publishes nothing of it, stereo_event_tracker_node.cpp
 * :262) returns without waiting for its stereo LK at all: its left-camera members are complete, its
 * right-camera members (feature_tracker.cpp:475-575) still show the previous frame until the next
 * call, esvio_fe_finish or esvio_fe_pack_track_records completes them, again bit-identically. */
int esvio_fe_set_lazy_new_stereo(esvio_fe_handle h, int on);
/* Throughput option: rejectWithF_event's RANSAC (feature_tracker.cpp:935, ~100-350 iterations of
 * the 7-point solver per published frame, on the frame's critical path) uses `threads` host threads
 * (1 = the calling thread only, the default; at most 16).  The helpers only solve and score
 * iterations; the cv::RNG draws and the best-model bookkeeping stay sequential on the calling
 * thread, so the result is bit-identical for any thread count.  Helpers spin while frames keep
 * coming and sleep after 2 ms without work. */
int esvio_fe_set_host_threads(esvio_fe_handle h, int threads);
/* Throughput option for replay mode: with `on`, the HIP calls that start an announced batch's SAE update,
 * rendering, pyramids and Arc* pass (~10 launches and event calls, 35-45 us of host time per batch) are
 * issued by a thread of the handle instead of by the calling thread — the calling thread, which bounds
 * the replay rate, keeps only the bookkeeping and checks that a batch's job has been issued before it
 * consumes that batch (three frames later in the steady state).  Results do not depend on it.  The
 * thread spins while batches keep coming and blocks after 2 ms without one: one more busy CPU. */
int esvio_fe_set_launch_thread(esvio_fe_handle h, int on);
/* complete a lazily returned frame (no-op otherwise) and copy the result members into `out` */
int esvio_fe_finish(esvio_fe_handle h, esvio_fe_tracks* out);

/* FeatureTracker::gettimesurface() tap (feature_tracker.cpp:894): current left/right image */
int esvio_fe_get_time_surface(esvio_fe_handle h, int cam, uint8_t* out);

/* ---- camera split across GPUs (BASELINE config C4) ----------------------- */
/* The left and right cameras have disjoint SAE state (sae_/sae_latest_ vs sae_right/
 * sae_latest_right, event_detector.h:74-79), so a second GPU can own the right camera: it runs
 * esvio_fe_create_sae(cam=1) + esvio_fe_sae_to_time_surface(cam=1, t_sync = LEFT batch end,
 * feature_tracker.cpp:367-368) and ships the 1-byte/pixel image.  export: copy the current image
 * of `cam` into a contiguous width*height buffer (host or device).  import: hand the left GPU's
 * handle the right image for the NEXT esvio_fe_track_event call (one shot), which then skips the
 * right camera's SAE update and rendering (pass nR = 0) and uses this image for stereo LK. */
int esvio_fe_export_image(esvio_fe_handle h, int cam, uint8_t* dst, int space);
int esvio_fe_import_image(esvio_fe_handle h, int cam, const uint8_t* src, int space);

/* ---- one stream time-sliced across GPUs (BASELINE config C5) ------------- */
/* createSAE_left/right (event_detector.cc:149-166) for ONE batch cut into N consecutive slices of the
 * stream, one per GPU.  Every rank holds the planes as they were before the batch.  Per batch:
 *   1. rank r: esvio_fe_sae_slice_last(its slice) -> last_r: per (camera, pixel, polarity) the time of
 *      the slice's last event, ESVIO_FE_SLICE_NONE where it has none (L[p] = t is unconditional,
 *      :158, so this does not depend on what came before);          all-gather last_0..last_{N-1}
 *   2. rank r: esvio_fe_sae_slice_apply(its slice, last_0..last_{r-1}) -> s_r: the time of the
 *      slice's last event that PASSES `t > L[p] + thr || L[!p] > L[p]` (:155), evaluated with the
 *      exact carried-in L (planes before the batch overlaid with the earlier slices), NONE where no
 *      event passes — the decisions are the sequential loop's for any timestamps; all-gather s_r
 *   3. every rank: esvio_fe_sae_slice_commit(all last, all s): planes after the batch = planes
 *      before it overlaid with the slices in order.  The next esvio_fe_track_event call on this
 *      handle then takes the batch's SAE update as done (it still needs the batch's left events
 *      for Arc*) — one shot, like esvio_fe_import_image.
 * A plane set is esvio_fe_sae_plane_doubles(h) = 2 cameras x width*height x 2 polarities doubles in
 * the handle's own order (opaque to the caller; it only travels between handles of the same size).
 * Not to be mixed with esvio_fe_set_next_batch. */
#define ESVIO_FE_SLICE_NONE (-1.0)
size_t esvio_fe_sae_plane_doubles(esvio_fe_handle h);
int esvio_fe_sae_slice_last(esvio_fe_handle h, const esvio_fe_event* left, size_t nL,
                            const esvio_fe_event* right, size_t nR, int space, double* last_out,
                            int out_space);
int esvio_fe_sae_slice_apply(esvio_fe_handle h, const esvio_fe_event* left, size_t nL,
                             const esvio_fe_event* right, size_t nR, int space,
                             const double* last_before /* [n_before] plane sets */, int n_before,
                             int in_space, double* s_out, int out_space);
int esvio_fe_sae_slice_commit(esvio_fe_handle h, const double* last_all, const double* s_all,
                              int n_slices, int space);

/* ---- event memory ------------------------------------------------------------------------- */
/* Memory for event batches from the HIP runtime the library itself is linked to (a process may hold a
 * second one, e.g. the copy PyTorch bundles).  ESVIO_FE_HOST: pinned host memory — a batch kept there
 * (a dvs_msgs::EventArray_<Allocator> whose allocator calls this, or a buffer the driver's callback
 * deserialises into) goes to the device as one DMA, without the staging copy pageable memory needs.
 * ESVIO_FE_DEVICE: device memory on the current device; esvio_fe_mem_upload fills it from host memory
 * (synchronous).  Freed with esvio_fe_mem_free(space, p). */
int esvio_fe_mem_alloc(int space, size_t bytes, void** out);
int esvio_fe_mem_free(int space, void* p);
int esvio_fe_mem_upload(void* dst_device, const void* src_host, size_t bytes);
/* ... or the caller's own storage, page-locked where it lies: esvio_fe_register_host_buffer(p, bytes) once for a
 * buffer that event batches are handed over from again and again (the deserialisation buffer of the driver callback,
 * a ring of EventArray storage, stereo_event_tracker_node.cpp:128-142,399) — every later batch inside [p, p + bytes)
 * then crosses PCIe straight from there: no staging copy by the CPU at all.  Registering costs a system call and
 * a page walk (~0.1 ms per MB): per buffer, not per batch.  esvio_fe_unregister_host_buffer(p) before the memory
 * is freed.  (hipHostRegister / hipHostUnregister through the library's own HIP runtime.) */
int esvio_fe_register_host_buffer(void* p, size_t bytes);
int esvio_fe_unregister_host_buffer(void* p);

/* ---- capacity ----------------------------------------------------------------------------- */
/* Every event-proportional device buffer (partition scratch, candidate sets, staging lanes) grows on
 * demand, by a hipFree + hipMalloc inside the call that first needs more — a stall of 0.1-2 ms in the
 * middle of a stream.  esvio_fe_reserve makes them all large enough for batches of up to
 * max_events_left + max_events_right events (and for max_events_left Arc* candidates in every candidate
 * set) now, so that no call below that size allocates (esvio_fe_latency.allocs counts the ones that
 * do); with host_batches != 0 also the staging slots of batches handed over in ESVIO_FE_HOST memory
 * (8 slots of 32 B per event, half of it pinned).  Not while batches are announced.  The reference has
 * no counterpart: its std::vectors grow inside the callbacks. */
int esvio_fe_reserve(esvio_fe_handle h, size_t max_events_left, size_t max_events_right, int host_batches);

#ifdef __cplusplus
}
#endif
#endif /* ESVIO_FE_H */
