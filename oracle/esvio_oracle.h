/*
 * esvio_oracle.h — C interface of the CPU ORACLE for the ESVIO event front-end hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (esvio_amd/, include/) may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (arclab-hku/ESVIO @2024-11-15) ships no tests, golden
 * vectors or fixtures for this path, and none of its sources for the path can be built in
 * this image without stand-ins for Eigen/OpenCV/ROS headers (not permitted).  This oracle
 * is therefore a line-by-line restatement of the reference's algorithm, checked by
 * hand-derived known-answer tests (tests/test_oracle_*.py), not by reference outputs.
 * Every function cites the reference file:line it restates; arithmetic that lives in
 * OpenCV (un-vendored, unpinned; ROS Noetic ships 4.2.0) is restated from the published
 * algorithm and marked [OpenCV].
 */
#ifndef ESVIO_ORACLE_H
#define ESVIO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dvs_msgs::Event in-memory layout (feature_tracker/src/dvs_msgs/Event.h:42-52):
 * uint16 x; uint16 y; ros::Time ts {uint32 sec; uint32 nsec}; uint8 polarity  => 16 B. */
typedef struct {
  uint16_t x, y;
  uint32_t sec, nsec;
  uint8_t polarity;
  uint8_t _pad[3];
} oracle_event;

/* pinhole + radtan camera (camera_model/src/camera_models/PinholeCamera.cc) */
typedef struct {
  double fx, fy, cx, cy, k1, k2, p1, p2;
} oracle_camera;

/* YAML knobs read by readParameters_event (feature_tracker/src/parameters.cpp:183-282) */
typedef struct {
  int32_t width, height;             /* event_width / event_height (COL_event, ROW_event) */
  double decay_ms;                   /* para_decay_ms */
  int32_t ignore_polarity;           /* para_ignore_polarity */
  int32_t median_blur_kernel_size;   /* must be 0 (all shipped configs) */
  double feature_filter_threshold;   /* para_feature_filter_threshold */
  double ts_lk_threshold;            /* TS_LK_THRESHOLD (128.0) */
  int32_t max_cnt;                   /* MAX_CNT */
  int32_t min_dist;                  /* MIN_DIST */
  int32_t flow_back;                 /* FLOW_BACK */
  int32_t equalize;                  /* EQUALIZE: CLAHE + normalize before LK */
  double f_threshold;                /* F_THRESHOLD */
  int32_t f_ransac;                  /* 0: skip rejectWithF_event; 1: restated RANSAC */
  int32_t lk_accum;                  /* 0: float scalar-order sums; 1: exact int64 sums; 2: float sums in the x86 SIMD128 lane order of OpenCV 4.2, on real SSE2 registers where the host has them; 4: the scalar emulation of those lanes (the cross-check of 2) */
  int32_t focal_length;              /* FOCAL_LENGTH = 460 (parameters.cpp:274) */
  int32_t _pad;
  oracle_camera cam[2];
} oracle_config;

/* ------------------------------------------------------------------ detector */
void* oracle_detector_create(int W, int H, double decay_ms, int ignore_polarity,
                             double filter_threshold, int min_dist);
void oracle_detector_destroy(void* d);
void oracle_detector_reset(void* d);
/* median_blur_kernel_size k (event_detector.cc:262-264): cv::medianBlur(2k+1) on every rendered surface */
void oracle_detector_set_median(void* d, int k);
void oracle_median_blur(uint8_t* img, int w, int h, int ksize);
/* createSAE_left (cam 0) / createSAE_right (cam 1) applied to n events in stream order.
 * returns number of events rejected because x>=W or y>=H (the reference would abort). */
size_t oracle_create_sae(void* d, int cam, const oracle_event* ev, size_t n);
/* SAEtoTimeSurface_left/right; out is H*W row-major u8 */
void oracle_sae_to_time_surface(void* d, int cam, double t_sync, uint8_t* out);
/* isCorner on the LEFT planes (sae_/sae_latest_) */
int oracle_is_corner(void* d, double et, int ex, int ey, int ep);
void oracle_corner_flags(void* d, const oracle_event* ev, size_t n, uint8_t* flags);
/* copy the 4 state planes of a camera (each H*W doubles, index x + y*W): L0,L1,S0,S1 */
void oracle_get_sae(void* d, int cam, double* L0, double* L1, double* S0, double* S1);
void oracle_set_sae(void* d, int cam, const double* L0, const double* L1, const double* S0,
                    const double* S1);

/* ------------------------------------------------------------------ mask / selection */
/* half-widths of cv::circle(..., r, -1) rows dy=0..r; hw[dy] = max |dx| painted */
void oracle_disc_halfwidths(int r, int* hw /* r+1 */);
/* paint a filled disc with value v into a W*H u8 image */
void oracle_circle_fill(uint8_t* img, int W, int H, int cx, int cy, int r, uint8_t v);
/* Event_FeaturesToTrack (feature_tracker.cpp:13-38); mask: W*H u8, 255 = blocked (input
 * is not modified); ts: raw left time surface; returns number of corners written to
 * out_xy (x,y float pairs) and their event indices to out_idx (may be NULL) */
int oracle_features_to_track(void* d, const oracle_event* ev, size_t n, int max_corners,
                             int min_dist, const uint8_t* mask, const uint8_t* ts,
                             double ts_lk_threshold, float* out_xy, int32_t* out_idx);

/* ------------------------------------------------------------------ image ops [OpenCV] */
void oracle_pyr_down(const uint8_t* src, int sw, int sh, uint8_t* dst);
/* dst: interleaved (Ix,Iy) int16, 2*w*h */
void oracle_scharr(const uint8_t* src, int w, int h, int16_t* dst);
/* number of pyramid levels actually built for a w x h image (returns maxLevel clamp) */
int oracle_pyr_levels(int w, int h, int win, int max_level);
/* calcOpticalFlowPyrLK(prev,next,prev_pts,next_pts,status,err,Size(win,win),max_level,
 *                      TermCriteria(COUNT+EPS,max_count,eps),flags)
 * flags bit 2 (=4) = OPTFLOW_USE_INITIAL_FLOW. next_pts is in/out. accum: see lk_accum */
void oracle_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts,
               float* next_pts, uint8_t* status, int n, int win, int max_level, int max_count,
               double eps, int flags, int accum);

/* cv::createCLAHE()->apply (clip 40, 8x8 tiles) and cv::normalize(.,0,255,NORM_MINMAX) on u8
 * (feature_tracker.cpp:377-381) */
void oracle_clahe(const uint8_t* src, int w, int h, uint8_t* dst);
void oracle_normalize_minmax(uint8_t* img, size_t n);

/* ------------------------------------------------------------------ camera */
void oracle_lift_projective(const oracle_camera* cam, double u, double v, double* out3);

/* ------------------------------------------------------------------ F-RANSAC [OpenCV] */
/* cv::findFundamentalMat(p1,p2,FM_RANSAC,thr,conf,status); returns #inliers, 0 on failure
 * (status then all 0 like OpenCV's empty-F return). */
int oracle_find_fundamental_ransac(const float* p1, const float* p2, int n, double thr,
                                   double conf, uint8_t* status, double* F9);
/* run7Point's null space: 0 = cv::SVDecomp's one-sided Jacobi route (default, what OpenCV calls),
 * 1 = Householder QR of A^T (same plane, another basis; only for measuring what the basis moves) */
void oracle_set_nullspace_mode(int mode);
/* the Jacobi SVD itself on n rows of length m (m >= n, n <= 16), completed to n1 rows; `at` is
 * n1 x m row-major, rows >= n ignored on input.  Out: orthonormal rows, w[n] singular values. */
int oracle_svd_rows(double* at, int m, int n, int n1, double* w);
/* FMEstimatorCallback::run7Point on 7 point pairs: up to 3 row-major F into F27; returns how many */
int oracle_seven_point(const float* p1, const float* p2, double* F27);

/* ------------------------------------------------------------------ motion compensation */
/* the fields of Motion_correction_value that createSAE_* (5 args) reads
 * (stereo_event_tracker_node.cpp:192-254, event_detector.cc:102-147) + detector.init's K */
typedef struct {
  double t1;          /* event_left.header.stamp (feature_tracker.cpp:622) */
  double v[3];        /* State_[0..2] (node:215-217) */
  float v_pre[3];     /* node:220-222 */
  float accel[3];     /* temp_a (node:230-232): gates the warp at |a| > 5 */
  float omega[3];     /* IMU angular velocity (node:244-246) */
  double fx, fy, cx, cy; /* detector.init(COL,ROW,fx,fy,cx,cy) (feature_tracker.cpp:616) */
} oracle_motion;
size_t oracle_create_sae_mc(void* d, int cam, const oracle_event* ev, size_t n,
                            const oracle_event* first_left, const oracle_motion* motion);
void oracle_matrix_exp3f(const float* a9, float* out9);

/* ------------------------------------------------------------------ tracker */
typedef struct {
  int32_t n_left;
  int32_t n_right;
  int32_t* ids;          /* [max_cnt] */
  int32_t* track_cnt;    /* [max_cnt] */
  float* cur_pts;        /* [2*max_cnt] */
  float* cur_un_pts;     /* [2*max_cnt] */
  float* pts_velocity;   /* [2*max_cnt] */
  int32_t* ids_right;    /* [max_cnt] */
  float* cur_right_pts;  /* [2*max_cnt] */
  float* cur_un_right_pts;
  float* right_pts_velocity;
} oracle_tracks;

void* oracle_tracker_create(const oracle_config* cfg);
void oracle_tracker_destroy(void* t);
/* FeatureTracker::trackEvent (feature_tracker.cpp:340-603) */
int oracle_track_event(void* t, double cur_time, const oracle_event* left, size_t nL,
                       const oracle_event* right, size_t nR, int pub_this_frame,
                       oracle_tracks* out);
/* trackEvent overload with Motion_correction_value (feature_tracker.cpp:605-877) */
int oracle_track_event_mc(void* t, double cur_time, const oracle_event* left, size_t nL,
                          const oracle_event* right, size_t nR, int pub_this_frame,
                          const oracle_motion* motion, oracle_tracks* out);
/* ---- image front-end */
/* cv::goodFeaturesToTrack(img, maxCorners, quality, minDistance, mask) with blockSize 3,
 * gradientSize 3, Shi-Tomasi [OpenCV restated]; eig_out (w*h floats, optional) = cornerMinEigenVal */
int oracle_good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double quality,
                                  double min_distance, const uint8_t* mask, float* out_xy,
                                  int32_t* n_out, float* eig_out);
void oracle_euclid_halfwidths(int r, int* hw /*[r+1]*/);
/* FeatureTracker::trackImage (feature_tracker.cpp:164-338); right may be NULL */
int oracle_track_image(void* t, double cur_time, const uint8_t* left, const uint8_t* right,
                       int pub_this_frame, oracle_tracks* out);
/* taps */
void oracle_tracker_time_surface(void* t, int cam, uint8_t* out);
void* oracle_tracker_detector(void* t);
/* diagnostics: {LK iterations, (point,level) visits, visits that hit maxCount} */
void oracle_lk_iter_stats(unsigned long long* out3, int reset);
/* per forward+backward pair of LK calls of the trackers (what one fused GPU launch runs per point):
 * {pairs, sum of the slowest point's iterations, points, iterations, slowest point of any pair} */
void oracle_lk_pair_stats(unsigned long long* out5, int reset);
/* per-stage wall-clock accumulators (seconds): sae, ts, lk_temporal, detect, lk_stereo, host */
void oracle_tracker_stage_seconds(void* t, double* out6);

#ifdef __cplusplus
}
#endif
#endif
