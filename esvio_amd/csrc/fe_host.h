// fe_host.h — host-side pieces of trackEvent that stay on the CPU (<= max_cnt points per frame):
// pinhole lift, cv::circle disc table, the blocked-pixel bitmap, F-matrix RANSAC.
#pragma once
#include <atomic>
#include <stdint.h>

#include <vector>

#include "../../include/esvio_fe.h"
#include "../../include/esvio_fe_test.h"

namespace esvio {
// one spin-loop hint for every host-side wait (x86: pause; elsewhere: nothing — the host files build for any
// CPU that has a ROCm)
inline void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
  __builtin_ia32_pause();
#endif
}
namespace host {

// OpenCV rounding helpers [core/fast_math.hpp]
int cv_round(double v);   // round-half-even; INT_MIN when out of int32 range
int cv_floor(float v);

// camodocal PinholeCamera::liftProjective (camera_model/src/camera_models/PinholeCamera.cc:450-510)
void lift_projective(const esvio_fe_camera& cam, double u, double v, double out[3]);
// the same for n points (uv = interleaved float pairs), results xu[i], yu[i] with z = 1
void lift_projective_batch(const esvio_fe_camera& cam, const float* uv, int n, double* xu, double* yu);

// half-width per |dy| of cv::circle(img, c, r, color, -1) [OpenCV imgproc/drawing.cpp Circle()]
std::vector<int> disc_halfwidths(int r);

// Blocked-pixel bitmap, one bit per pixel, wpr 32-bit words per row (bit set = 255.0 in the
// reference's CV_64F mask_event).
struct BitMask {
  int W = 0, H = 0, wpr = 0;
  std::vector<uint32_t> bits;
  void reset(int w, int h);
  bool test(int x, int y) const { return (bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u; }
  void stamp_disc(int cx, int cy, int r, const std::vector<int>& hw);
  void from_bytes(const uint8_t* mask);  // 255 -> set
};

// cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, conf, status) [OpenCV calib3d fundam.cpp +
// ptsetreg.cpp]: 7-point minimal solver, RANSAC for n >= 15, LMedS for 8..14, cv::RNG sequence.
// Returns the number of inliers; status has n entries.
// With a pool (helper threads, see fe_host.cpp) the RANSAC iterations are solved and scored in
// parallel; the outcome is that of the sequential loop, bit for bit.
struct RansacPool;
RansacPool* ransac_pool_create(int helpers);
void ransac_pool_destroy(RansacPool* p);
void ransac_pool_wake(RansacPool* p);  // a job is coming: helpers that went to sleep start spinning
// test tap: the hypot the 7-point solver's Jacobi rotations use (cv::hypot of OpenCV's lapack.cpp)
void host_hypot(const double* x, const double* y, int n, double* out);
// test tap: run7Point's null-space basis (rows 7, 8 of cv::SVDecomp's Vt) of n 7x9 systems (63 doubles
// each) -> f1 | f2 (18 doubles each), by the one-at-a-time routine (lanes = 0) or in groups of
// vector lanes as the RANSAC loop solves them (lanes = 1; returns how many systems fell back to the
// one-at-a-time routine)
int host_nullspace(const double* A, int n, int lanes, double* f);
// process-wide counters of find_fundamental_mat: its RANSAC branch (>= 15 points) — calls,
// hypotheses replayed (the loop's iteration count), points, nanoseconds inside the call — and its
// LMedS branch (8..14 points: a fixed 300 hypotheses) — calls, nanoseconds
struct RansacStats {
  uint64_t calls, iterations, points, ns, lmeds_calls, lmeds_ns;
};
RansacStats ransac_stats(bool reset);
// its tail: {slowest RANSAC call ns, slowest LMedS call ns, iterations the calling thread redid because
// the helper that had taken them did not deliver, jobs run without the helpers because a helper was
// still inside both job buffers, jobs that took the other buffer because a helper was still inside
// theirs, involuntary context switches of the helper threads}
void ransac_tail(uint64_t out6[6], bool reset);
// test tap: mark job buffer 0 / 1 (bits of `mask`) of the pool as still holding a helper, as if one had
// lost its CPU in the middle of a job (on = false: undo)
void ransac_pool_hold(RansacPool* p, int mask, bool on);
// Other work the helpers take while they spin without a RANSAC job (the handle's host-batch staging:
// chunks of a batch to copy into pinned memory): `pending` is polled without a lock, `fn(arg)` takes one
// unit and returns whether there was one.  nullptr: none.
void ransac_pool_set_idle_work(RansacPool* p, const std::atomic<int>* pending, bool (*fn)(void*), void* arg);
int find_fundamental_mat(const float* p1, const float* p2, int n, double thr, double conf,
                         uint8_t* status, RansacPool* pool = nullptr);

}  // namespace host
}  // namespace esvio
