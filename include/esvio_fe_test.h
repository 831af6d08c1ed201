/* esvio_fe_test.h — test and measurement taps of libesvio_fe.so.
 *
 * NOT part of the drop-in boundary (include/esvio_fe.h): nothing here replaces a
 * call of the reference.  These entry points exist so that tests/ and bench.py can reach pieces of the host
 * side in isolation (the RANSAC pool under adverse scheduling, the staging copy, the 7-point solver's
 * null-space basis), inject faults into the bounded device-side waits, and read counters.  A caller of the
 * front-end never needs them.
 */
#ifndef ESVIO_FE_TEST_H
#define ESVIO_FE_TEST_H
#include "esvio_fe.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- host RANSAC (esvio_fe_find_fundamental_mat) under test conditions ------------------- */
/* The same with `threads` - 1 helper threads solving / scoring the RANSAC iterations (a temporary
 * pool; test tap for esvio_fe_set_host_threads): status and count are those of the call above. */
int esvio_fe_find_fundamental_mat_mt(const float* p1, const float* p2, int n, double thr,
                                     double conf, int threads, uint8_t* status, int32_t* n_inliers);
/* Test tap: the same with job buffers of the pool marked as still holding a helper (hold_mask bit 0 / 1:
 * buffer 0 / 1), as when a helper thread loses its CPU in the middle of a job: the call must not wait for
 * it — it takes the other buffer, or runs without the helpers when both are held — and must return the
 * same status and count (esvio_fe_ransac_tail counts both cases).  threads >= 2. */
int esvio_fe_find_fundamental_mat_held(const float* p1, const float* p2, int n, double thr, double conf,
                                       int threads, int hold_mask, uint8_t* status, int32_t* n_inliers);
/* Test tap: the same while the pool's helpers have other work between jobs, the way the host-batch staging hands
 * them chunks (fe_evstage.cpp): `repeats` calls, `idle_units` units of ~5 us arriving before each; status and
 * count must equal esvio_fe_find_fundamental_mat's; out3 = {calls of the idle hook (a helper that is idle when the
 * hook is registered calls it once then), units done, units left after a bounded wait (0)}.  threads >= 2. */
int esvio_fe_find_fundamental_mat_idle(const float* p1, const float* p2, int n, double thr, double conf,
                                       int threads, int repeats, int idle_units, uint8_t* status, int32_t* n_inliers,
                                       uint64_t out3[3]);
/* Test tap: the hypot inside that function's 7-point solver: cv::SVD's Jacobi rotations call hypot
 * unqualified inside namespace cv, which resolves to lapack.cpp's own a*sqrt(1+(b/a)^2) template, not
 * to libm's; IEEE operations only, so the result does not depend on the host's libm. */
int esvio_fe_host_hypot(const double* x, const double* y, int n, double* out);
/* Test tap: the copy the staging threads move a chunk of a host-resident event batch with (pageable source ->
 * pinned buffer; streaming stores where dst is 16-byte aligned, memcpy otherwise and for the last < 64 bytes):
 * dst[0, len) = src[0, len), nothing else written.  No device involved. */
int esvio_fe_host_stage_copy(void* dst, const void* src, size_t len);
/* Test tap: the basis of the 7x9 epipolar system's null space that run7Point takes from
 * cv::SVDecomp(A, W, U, Vt, MODIFY_A + FULL_UV) (rows 7 and 8 of Vt; OpenCV calib3d/fundam.cpp, reached
 * from feature_tracker.cpp:935), for n systems of 63 doubles -> f12 = n x (f1[9] | f2[9]).  lanes = 0:
 * one system at a time; lanes != 0: side by side in vector lanes with the sweep's independent row pairs
 * scheduled together, as the RANSAC loop solves its hypotheses — both must give the same bits.
 * *redone (may be NULL) = systems the lane form handed back to the one-at-a-time routine. */
int esvio_fe_host_nullspace(const double* systems, int n, int lanes, double* f12, int32_t* redone);
/* One chunk of a host batch as the staging threads pack it for a plain call's pull over PCIe (fe_evstage.cpp stage_pack):
 * len bytes of 16-byte records at src -> len / 2 bytes at dst (16-byte aligned), 8 bytes per event: x | y << 16, then
 * nsec | (polarity != 0) << 30 | (sec - *base_sec) << 31.  Returns 1 if every event fits that form (nsec < 2^30, second
 * = the first event's or the one after), 0 if not (the chunk then travels raw; dst's contents are unspecified), < 0 on
 * bad arguments. */
int esvio_fe_host_stage_pack(void* dst, const void* src, size_t len, uint32_t* base_sec);
/* {host batches staged, bytes staged, chunks that went to the device packed, chunks of packing batches that went raw} */
int esvio_fe_staging_counters(esvio_fe_handle h, uint64_t out4[4]);

/* ---- fault injection (tests) ----------------------------------------------------------- */
/* Every device-side wait is bounded: a wave that gives up raises a host-visible flag and the call
 * either fails with ESVIO_FE_EINTERNAL (the SAE update's turn ticket and the radix sort's look-back:
 * the planes are then partially updated; esvio_fe_reset makes the handle usable again) or redoes the
 * launch the plain way (the speculative / chained temporal LK of replay mode: results unchanged).
 * esvio_fe_debug_inject makes the chosen waits expire the first time they would have to wait, for the
 * launches that follow (0: normal bounds again); ESVIO_FE_FAULT=<mask> in the environment does the
 * same from esvio_fe_create on.  esvio_fe_debug_counters: {speculative launches redone, chained
 * launches redone, chained launches made, chained launches used}. */
#define ESVIO_FE_FAULT_TICKET 1
#define ESVIO_FE_FAULT_LOOKBACK 2
#define ESVIO_FE_FAULT_SPECULATIVE 4
#define ESVIO_FE_FAULT_CHAINED 8
/* replay mode's lazy completions as late as they may ever happen, whether or not the stereo LK they wait for is over: a
 * lazily returning call always leaves the previous published frame's new corners to the next call, a published call
 * always runs the previous frame's right-camera tail in its own tail — the order of the two, not their timing, is what
 * results depend on */
#define ESVIO_FE_FAULT_LAZY_LATE 16
int esvio_fe_debug_inject(esvio_fe_handle h, int mask);
int esvio_fe_debug_counters(esvio_fe_handle h, uint64_t out4[4]);
/* What the plain calls (nothing announced, not lazy: the reference node's pattern) did since create:
 * {plain calls, of them with the two cameras' SAE update + image on two streams, stereo LK launches
 * chained to the temporal one on the device, chained launches redone}. */
int esvio_fe_plain_call_counters(esvio_fe_handle h, uint64_t out4[4]);

/* Tail of the host RANSAC (process-wide, like esvio_fe_ransac_stats; reset together with it or here):
 * out6 = {slowest RANSAC call [ns], slowest LMedS call [ns], iterations the calling thread redid because
 * the helper that took them did not deliver, jobs the calling thread ran alone because helpers were
 * still inside both job buffers, jobs that took the other buffer because a helper was still inside
 * theirs (a helper that lost its CPU in the middle of a job), involuntary context switches of the helper
 * threads}. */
int esvio_fe_ransac_tail(uint64_t out6[6], int reset);



/* ---- taps on single stages (tests compare them with the oracle; the reference has no such calls) -------- */
/* write a camera's four planes (each width*height doubles, index x + y*width); the read side is
 * esvio_fe_get_sae in esvio_fe.h */
int esvio_fe_set_sae(esvio_fe_handle h, int cam, const double* L0, const double* L1,
                     const double* S0, const double* S1);

/* test tap: the pyramid the LK stage builds for a host image: level `level` u8 image
 * (lw*lh bytes) and its Scharr derivatives (lw*lh*2 int16, interleaved Ix,Iy). Either output
 * may be NULL. Returns the number of levels built (maxLevel+1) in *n_levels. */
int esvio_fe_build_pyramid(esvio_fe_handle h, const uint8_t* img, int w, int hgt, int max_level,
                           int level, uint8_t* out_img, int16_t* out_deriv, int32_t* lw,
                           int32_t* lh, int32_t* n_levels);
/* camodocal PinholeCamera::liftProjective (PinholeCamera.cc:450-510); host-side. */
int esvio_fe_lift_projective(const esvio_fe_camera* cam, double u, double v, double* out3);

/* Measurement tap: process-wide counters of that function since the last reset — out6 = {calls,
 * loop iterations, points, nanoseconds inside the calls} of its RANSAC branch (>= 15 points) and
 * {calls, nanoseconds} of its LMedS branch (8..14 points, what OpenCV runs below 15). */
int esvio_fe_ransac_stats(uint64_t* out6, int reset);

/* ---- measurement --------------------------------------------------------------------- */
/* Wall time of the esvio_fe_track_event(_mc) calls on this handle since the last reset, as the calling
 * thread sees them (always on: a dozen clock reads per call).  The percentiles cover the latest 4096
 * calls.  For the slowest call: its index since the reset, whether it published, where its time went
 * (esvio_fe_latency_phase_name(i) names max_phase_ms[i]; entries 8.. are parts of entry 5 on published
 * frames), the CPUs the calling thread was on when it began / ended, the involuntary context switches
 * the thread suffered inside it (getrusage(RUSAGE_THREAD)) and the device / pinned allocations it
 * made. */
#define ESVIO_FE_LATENCY_PHASES 16
typedef struct esvio_fe_latency {
  uint64_t calls;
  double mean_ms, p50_ms, p99_ms, max_ms;
  uint64_t max_call;
  int32_t max_published;
  int32_t max_cpu_begin, max_cpu_end;
  int64_t max_invol_switches;
  int64_t max_allocs;
  double max_phase_ms[ESVIO_FE_LATENCY_PHASES];
  uint64_t allocs;          /* allocations inside track / announce calls since the reset */
  uint64_t invol_switches;  /* involuntary context switches inside track calls since the reset */
} esvio_fe_latency;
int esvio_fe_latency_stats(esvio_fe_handle h, esvio_fe_latency* out, int reset);
const char* esvio_fe_latency_phase_name(int i);
/* One of the latest 256 track calls as the record above saw it (back = 0: the last call, 1: the one before ...):
 * when it began (ms since the first track call after the last reset), how long it took, whether it published, and its phases —
 * read AFTER a run, so that looking does not change the schedule that is looked at.  ESVIO_FE_EINVAL for a call
 * that is not kept (any more). */
typedef struct esvio_fe_latency_call {
  uint64_t call;      /* index of the call since the last reset of esvio_fe_latency_stats (as max_call there) */
  int32_t published;
  int32_t reserved;
  double begin_ms, ms;
  double phase_ms[ESVIO_FE_LATENCY_PHASES];
} esvio_fe_latency_call;
int esvio_fe_latency_recent(esvio_fe_handle h, int back, esvio_fe_latency_call* out);
/* Per-kernel HIP-event timing on the handle's stream (off by default; when on, every launch is
 * bracketed by hipEventRecord and resolved lazily). */
int esvio_fe_set_profiling(esvio_fe_handle h, int on);
int esvio_fe_kernel_count(void);
const char* esvio_fe_kernel_name(int kernel_id);
/* total_ms / launches / algorithmic bytes accumulated since the last reset_kernel_stats */
int esvio_fe_get_kernel_stats(esvio_fe_handle h, int kernel_id, double* total_ms,
                              uint64_t* launches, uint64_t* alg_bytes);
int esvio_fe_reset_kernel_stats(esvio_fe_handle h);
/* the hipStream_t the handle launches on (as void*) */
void* esvio_fe_stream(esvio_fe_handle h);
/* hipMemGetInfo on the handle's device, through the HIP runtime the library itself is linked to */
int esvio_fe_device_memory(esvio_fe_handle h, size_t* free_bytes, size_t* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ESVIO_FE_TEST_H */
