// esvio_oracle.cpp — CPU ORACLE (test infrastructure, see esvio_oracle.h header note).
//
// PARITY UNPINNED (no reference tests/fixtures exist; reference sources for this path need
// Eigen/OpenCV/ROS which are absent and may not be stubbed).  Single-threaded like the
// reference's worker thread (feature_tracker/src/stereo_event_tracker_node.cpp:366).
//
// Build: g++ -O3 -std=c++17 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
// -ffp-contract=off + x86-64 baseline (no FMA) mirrors the reference build flags
// (feature_tracker/CMakeLists.txt:4-6: -O3 -Wall -g, no -march).
#include "esvio_oracle.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#define ORACLE_LK_SSE2 1
#else
#define ORACLE_LK_SSE2 0
#endif

namespace {

// ---------------------------------------------------------------- OpenCV scalar helpers
// cvRound(double): SSE2 cvtsd2si, round-half-even, "integer indefinite" 0x80000000 when the
// value does not fit int32 [OpenCV core/fast_math.hpp].
inline int cv_round_d(double v) {
  if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;  // also NaN
  double r = std::nearbyint(v);  // default rounding mode: ties-to-even
  if (r > 2147483647.0 || r < -2147483648.0) return INT_MIN;
  return (int)r;
}
inline int cv_round_f(float v) { return cv_round_d((double)v); }
inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}
inline uint8_t saturate_u8(int v) { return (uint8_t)((unsigned)v <= 255 ? v : v > 0 ? 255 : 0); }
// cv::borderInterpolate(p, len, BORDER_REFLECT_101) for |overshoot| < len
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}
#define CV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

inline double ev_time(const oracle_event& e) {
  // ros::Time::toSec(): (double)sec + 1e-9*(double)nsec
  return (double)e.sec + 1e-9 * (double)e.nsec;
}

// ---------------------------------------------------------------- EventDetector
// feature_tracker/src/event_detector/event_detector.{h,cc}
struct Detector {
  int W, H;
  double decay_ms;
  bool ignore_polarity;
  double filter_threshold;
  int min_dist;
  int median_blur_kernel_size = 0;  // k: cv::medianBlur(ksize 2k+1) on the rendered surface (:262-264)
  // planes indexed x + y*W (Eigen MatrixXd(W,H) col-major indexed (x,y), event_detector.cc:52-63)
  // [cam][pol]; cam 0 = sae_/sae_latest_ (left), cam 1 = sae_right/sae_latest_right
  std::vector<double> sae[2][2];         // S: latest ACCEPTED time
  std::vector<double> sae_latest[2][2];  // L: latest time incl. filtered
  void reset() {
    for (int c = 0; c < 2; c++)
      for (int p = 0; p < 2; p++) {
        sae[c][p].assign((size_t)W * H, 0.0);
        sae_latest[c][p].assign((size_t)W * H, 0.0);
      }
  }
};

// event_detector.cc:14-22
const int kSmallCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                 {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                 {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
const int kLargeCircle[20][2] = {{0, 4},   {1, 4},   {2, 3},   {3, 2},  {4, 1},  {4, 0},  {4, -1},
                                 {3, -2},  {2, -3},  {1, -4},  {0, -4}, {-1, -4}, {-2, -3}, {-3, -2},
                                 {-4, -1}, {-4, 0},  {-4, 1},  {-3, 2}, {-2, 3},  {-1, 4}};

// createSAE_left (event_detector.cc:149-166) / createSAE_right (:212-228), one event
inline void create_sae_one(Detector* d, int cam, double et, int ex, int ey, bool ep) {
  const int pol = ep ? 1 : 0;
  const int pol_inv = (!ep) ? 1 : 0;
  const size_t idx = (size_t)ex + (size_t)ey * d->W;
  double& t_last = d->sae_latest[cam][pol][idx];
  double& t_last_inv = d->sae_latest[cam][pol_inv][idx];
  if ((et > t_last + d->filter_threshold) || (t_last_inv > t_last)) {
    t_last = et;
    d->sae[cam][pol][idx] = et;
  } else {
    t_last = et;
  }
}

// SAEtoTimeSurface_left/right (event_detector.cc:230-305)
// cv::medianBlur(src, dst, ksize) on CV_8U, BORDER_REPLICATE [OpenCV imgproc/median_blur]: the
// exact median of the ksize x ksize neighbourhood (OpenCV's sorting networks / histogram variants
// all return it); in place in the reference, i.e. computed from a copy of the source
void median_blur_u8(uint8_t* img, int W, int H, int ksize) {
  const int k = ksize / 2, n = ksize * ksize;
  std::vector<uint8_t> src(img, img + (size_t)W * H), win(n);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int m = 0;
      for (int dy = -k; dy <= k; dy++)
        for (int dx = -k; dx <= k; dx++) {
          const int yy = std::min(std::max(y + dy, 0), H - 1), xx = std::min(std::max(x + dx, 0), W - 1);
          win[m++] = src[(size_t)yy * W + xx];
        }
      std::nth_element(win.begin(), win.begin() + n / 2, win.end());
      img[(size_t)y * W + x] = win[n / 2];
    }
}

void sae_to_ts(const Detector* d, int cam, double external_sync_time, uint8_t* out) {
  const double decay_sec = d->decay_ms / 1000.0;
  const std::vector<double>& s0 = d->sae[cam][0];
  const std::vector<double>& s1 = d->sae[cam][1];
  for (int y = 0; y < d->H; ++y) {
    for (int x = 0; x < d->W; ++x) {
      const size_t i = (size_t)x + (size_t)y * d->W;
      double v = 0.0;  // cv::Mat::zeros
      double most_recent = (s1[i] > s0[i]) ? s1[i] : s0[i];
      if (most_recent > 0) {
        const double dt = external_sync_time - most_recent;
        double expVal = std::exp(-dt / decay_sec);
        if (!d->ignore_polarity) {
          double polarity = (s1[i] > s0[i]) ? 1.0 : -1.0;
          expVal *= polarity;
        }
        v = expVal;
      }
      // :256-260  255.0*(M+1.0)/2.0  [OpenCV MatExpr folds to convertTo(alpha=127.5,beta=127.5)],
      // else 255.0*M (alpha=255,beta=0); then convertTo(CV_8U) = saturate_cast<uchar>(cvRound())
      double scaled = d->ignore_polarity ? (v * 255.0 + 0.0) : (v * 127.5 + 127.5);
      out[(size_t)y * d->W + x] = saturate_u8(cv_round_d(scaled));
    }
  }
  if (d->median_blur_kernel_size > 0)  // :262-264
    median_blur_u8(out, d->W, d->H, 2 * d->median_blur_kernel_size + 1);
}

// one Arc* ring (event_detector.cc:337-435 small, :438-541 large) — same code, N/min/max differ
template <int N>
inline bool arc_ring(const std::vector<double>& S, int W, int ex, int ey, const int (*circle)[2],
                     int kMin, int kMax) {
  auto at = [&](int i) -> double {
    return S[(size_t)(ex + circle[i][0]) + (size_t)(ey + circle[i][1]) * W];
  };
  double segment_new_min_t = at(0);
  int arc_right_idx = 0;
  int arc_left_idx;
  for (int i = 1; i < N; i++) {
    const double t = at(i);
    if (t > segment_new_min_t) {
      segment_new_min_t = t;
      arc_right_idx = i;
    }
  }
  arc_left_idx = (arc_right_idx - 1 + N) % N;
  arc_right_idx = (arc_right_idx + 1) % N;
  double arc_left_value = at(arc_left_idx);
  double arc_right_value = at(arc_right_idx);
  double arc_left_min_t = arc_left_value;
  double arc_right_min_t = arc_right_value;

  int iteration = 1;
  for (; iteration < kMin; iteration++) {
    if (arc_right_value > arc_left_value) {
      if (arc_right_min_t < segment_new_min_t) segment_new_min_t = arc_right_min_t;
      arc_right_idx = (arc_right_idx + 1) % N;
      arc_right_value = at(arc_right_idx);
      if (arc_right_value < arc_right_min_t) arc_right_min_t = arc_right_value;
    } else {
      if (arc_left_min_t < segment_new_min_t) segment_new_min_t = arc_left_min_t;
      arc_left_idx = (arc_left_idx - 1 + N) % N;
      arc_left_value = at(arc_left_idx);
      if (arc_left_value < arc_left_min_t) arc_left_min_t = arc_left_value;
    }
  }
  int newest_segment_size = kMin;
  for (; iteration < N; iteration++) {
    if (arc_right_value > arc_left_value) {
      if (arc_right_value >= segment_new_min_t) {
        newest_segment_size = iteration + 1;
        if (arc_right_min_t < segment_new_min_t) segment_new_min_t = arc_right_min_t;
      }
      arc_right_idx = (arc_right_idx + 1) % N;
      arc_right_value = at(arc_right_idx);
      if (arc_right_value < arc_right_min_t) arc_right_min_t = arc_right_value;
    } else {
      if (arc_left_value >= segment_new_min_t) {
        newest_segment_size = iteration + 1;
        if (arc_left_min_t < segment_new_min_t) segment_new_min_t = arc_left_min_t;
      }
      arc_left_idx = (arc_left_idx - 1 + N) % N;
      arc_left_value = at(arc_left_idx);
      if (arc_left_value < arc_left_min_t) arc_left_min_t = arc_left_value;
    }
  }
  return (newest_segment_size <= kMax) ||
         ((newest_segment_size >= (N - kMax)) && (newest_segment_size <= (N - kMin)));
}

// EventDetector::isCorner (event_detector.cc:308-544) — always the LEFT planes
bool is_corner(const Detector* d, double et, int ex, int ey, bool ep) {
  const int pol = ep ? 1 : 0;
  const int pol_inv = (!ep) ? 1 : 0;
  const size_t idx = (size_t)ex + (size_t)ey * d->W;
  const double t_last = d->sae_latest[0][pol][idx];
  const double t_last_inv = d->sae_latest[0][pol_inv][idx];
  if ((et > t_last + d->filter_threshold) || (t_last_inv > t_last)) return false;  // :315
  const int kBorderLimit = d->min_dist + 1;                                        // :320
  if (ex < kBorderLimit || ex >= (d->W - kBorderLimit) || ey < kBorderLimit ||
      ey >= (d->H - kBorderLimit))
    return false;
  if (!arc_ring<16>(d->sae[0][pol], d->W, ex, ey, kSmallCircle, 4, 6)) return false;
  return arc_ring<20>(d->sae[0][pol], d->W, ex, ey, kLargeCircle, 5, 8);
}

// ---------------------------------------------------------------- IMU motion compensation
// EventDetector::motioncorrection (event_detector.cc:547-591) and the 5-argument createSAE_*
// (:102-147, :168-210).  All matrix arithmetic is Eigen single precision; the evaluation order of
// Eigen's fixed-size 3x3 kernels is restated as recalled [upstream-Eigen 3.3, recalled — unpinned]:
// coefficient-based products with the 3-term reduction a0 + (a1 + a2), the cofactor inverse,
// MatrixBase::exp() = Pade 3/5/7 + partial-pivot LU solve + squarings (unsupported/MatrixFunctions).
struct Mat3f {
  float m[3][3];
};
inline float red3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
inline Mat3f m3_mul(const Mat3f& A, const Mat3f& B) {
  Mat3f C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C.m[i][j] = red3(A.m[i][0] * B.m[0][j], A.m[i][1] * B.m[1][j], A.m[i][2] * B.m[2][j]);
  return C;
}
inline Mat3f m3_identity() {
  Mat3f I = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  return I;
}
inline Mat3f m3_transpose(const Mat3f& A) {
  Mat3f T;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T.m[i][j] = A.m[j][i];
  return T;
}
// Eigen compute_inverse<.,.,3>: cofactors of column 0, det = sum(cof0 .* col0), rest * invdet
inline float cof3(const Mat3f& M, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return M.m[i1][j1] * M.m[i2][j2] - M.m[i1][j2] * M.m[i2][j1];
}
inline Mat3f m3_inverse(const Mat3f& M) {
  const float c0 = cof3(M, 0, 0), c1 = cof3(M, 1, 0), c2 = cof3(M, 2, 0);
  const float det = red3(c0 * M.m[0][0], c1 * M.m[1][0], c2 * M.m[2][0]);
  const float invdet = 1.0f / det;
  Mat3f R;
  R.m[0][0] = c0 * invdet;
  R.m[0][1] = c1 * invdet;
  R.m[0][2] = c2 * invdet;
  R.m[1][0] = cof3(M, 0, 1) * invdet;
  R.m[1][1] = cof3(M, 1, 1) * invdet;
  R.m[1][2] = cof3(M, 2, 1) * invdet;
  R.m[2][0] = cof3(M, 0, 2) * invdet;
  R.m[2][1] = cof3(M, 1, 2) * invdet;
  R.m[2][2] = cof3(M, 2, 2) * invdet;
  return R;
}
// denom.partialPivLu().solve(numer), 3x3: unblocked LU with row pivoting, then the matrix-RHS
// triangular solves (axpy order, upper solve multiplies by the reciprocal diagonal)
inline Mat3f m3_lu_solve(Mat3f LU, Mat3f X) {
  for (int k = 0; k < 3; k++) {
    int piv = k;
    float best = std::fabs(LU.m[k][k]);
    for (int i = k + 1; i < 3; i++)
      if (std::fabs(LU.m[i][k]) > best) {
        best = std::fabs(LU.m[i][k]);
        piv = i;
      }
    if (piv != k)
      for (int j = 0; j < 3; j++) {
        std::swap(LU.m[k][j], LU.m[piv][j]);
        std::swap(X.m[k][j], X.m[piv][j]);
      }
    if (best != 0.0f)
      for (int i = k + 1; i < 3; i++) LU.m[i][k] /= LU.m[k][k];
    for (int i = k + 1; i < 3; i++)
      for (int j = k + 1; j < 3; j++) LU.m[i][j] -= LU.m[i][k] * LU.m[k][j];
  }
  for (int c = 0; c < 3; c++) {  // unit lower, then upper
    for (int k = 0; k < 3; k++)
      for (int i = k + 1; i < 3; i++) X.m[i][c] -= X.m[k][c] * LU.m[i][k];
    for (int k = 2; k >= 0; k--) {
      const float a = 1.0f / LU.m[k][k];
      X.m[k][c] *= a;
      for (int i = 0; i < k; i++) X.m[i][c] -= X.m[k][c] * LU.m[i][k];
    }
  }
  return X;
}
inline Mat3f m3_lin(float a, const Mat3f& A, float b, const Mat3f& B) {  // a*A + b*B per coefficient
  Mat3f C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.m[i][j] = a * A.m[i][j] + b * B.m[i][j];
  return C;
}
Mat3f m3_exp(const Mat3f& arg) {
  float l1 = 0;
  for (int j = 0; j < 3; j++) {
    const float cs = red3(std::fabs(arg.m[0][j]), std::fabs(arg.m[1][j]), std::fabs(arg.m[2][j]));
    if (j == 0 || cs > l1) l1 = cs;
  }
  const Mat3f I = m3_identity();
  Mat3f U, V, A = arg;
  int squarings = 0;
  if (l1 < 4.258730016922831e-001f) {
    const Mat3f A2 = m3_mul(A, A);
    const Mat3f tmp = m3_lin(1.f, A2, 60.f, I);
    U = m3_mul(A, tmp);
    V = m3_lin(12.f, A2, 120.f, I);
  } else if (l1 < 1.880152677804762e+000f) {
    const Mat3f A2 = m3_mul(A, A), A4 = m3_mul(A2, A2);
    Mat3f tmp, v;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        tmp.m[i][j] = (1.f * A4.m[i][j] + 420.f * A2.m[i][j]) + 15120.f * I.m[i][j];
        v.m[i][j] = (30.f * A4.m[i][j] + 3360.f * A2.m[i][j]) + 30240.f * I.m[i][j];
      }
    U = m3_mul(A, tmp);
    V = v;
  } else {
    const float maxnorm = 3.925724783138660f;
    std::frexp(l1 / maxnorm, &squarings);
    if (squarings < 0) squarings = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A.m[i][j] = std::ldexp(arg.m[i][j], -squarings);
    const Mat3f A2 = m3_mul(A, A), A4 = m3_mul(A2, A2), A6 = m3_mul(A4, A2);
    Mat3f tmp, v;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        tmp.m[i][j] = ((1.f * A6.m[i][j] + 1512.f * A4.m[i][j]) + 277200.f * A2.m[i][j]) +
                      8648640.f * I.m[i][j];
        v.m[i][j] = ((56.f * A6.m[i][j] + 25200.f * A4.m[i][j]) + 1995840.f * A2.m[i][j]) +
                    17297280.f * I.m[i][j];
      }
    U = m3_mul(A, tmp);
    V = v;
  }
  Mat3f numer, denom;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      numer.m[i][j] = U.m[i][j] + V.m[i][j];
      denom.m[i][j] = -U.m[i][j] + V.m[i][j];
    }
  Mat3f R = m3_lu_solve(denom, numer);
  for (int i = 0; i < squarings; i++) R = m3_mul(R, R);
  return R;
}

struct MotionComp {  // what createSAE_* (5 args) reads from Motion_correction_value
  double t0;         // first LEFT event time (feature_tracker.cpp:621)
  double dt_batch;   // header stamp - t0 (:623)
  bool active;       // |accel| > a_motion_compensation_threshold (event_detector.cc:125)
  float v[3], v_pre[3], omega[3];
  Mat3f K, Kinv;
};

// motioncorrection (event_detector.cc:547-591): returns the (possibly warped) pixel
inline void motion_correct(const MotionComp& mc, int W, int H, int ex_i, int ey_i, double dt,
                           int* ox, int* oy) {
  const double ex = ex_i, ey = ey_i;
  const int kBorder = 6;
  *ox = ex_i;
  *oy = ey_i;
  if (ex > kBorder && ex <= (W - kBorder) && ey > kBorder && ey <= (H - kBorder)) {
    const float fdt = (float)dt;  // Vector3f * double: the scalar is converted to float
    const float rx = mc.omega[0] * fdt, ry = mc.omega[1] * fdt, rz = mc.omega[2] * fdt;
    const Mat3f skew = {{{0, -rz, ry}, {rz, 0, -rx}, {-ry, rx, 0}}};
    const Mat3f R = m3_exp(skew);
    const Mat3f rot_K = m3_mul(m3_mul(mc.K, m3_transpose(R)), mc.Kinv);
    const float h = (float)(0.5 * dt);
    float tk[3];
    for (int i = 0; i < 3; i++) tk[i] = h * (mc.v[i] + mc.v_pre[i]);
    float kt[3], tr[3];
    for (int i = 0; i < 3; i++)
      kt[i] = red3(mc.Kinv.m[i][0] * tk[0], mc.Kinv.m[i][1] * tk[1], mc.Kinv.m[i][2] * tk[2]);
    for (int i = 0; i < 3; i++)
      tr[i] = red3((-rot_K.m[i][0]) * kt[0], (-rot_K.m[i][1]) * kt[1], (-rot_K.m[i][2]) * kt[2]);
    float ev[3] = {(float)ex, (float)ey, 1.f}, w[3];
    for (int i = 0; i < 3; i++)
      w[i] = red3(rot_K.m[i][0] * ev[0], rot_K.m[i][1] * ev[1], rot_K.m[i][2] * ev[2]) + tr[i];
    w[0] = w[0] / w[2];  // ConvertToHomogeneous (feature_tracker.h)
    w[1] = w[1] / w[2];
    const int xc = (int)std::floor(w[0]), yc = (int)std::floor(w[1]);
    if (xc > 0 && xc < W - 1 && yc > 0 && yc < H - 1) {
      *ox = xc;
      *oy = yc;
    }
  }
}

// ---------------------------------------------------------------- cv::circle fill [OpenCV]
// imgproc/src/drawing.cpp Circle(): midpoint circle with horizontal fills. Returns the union
// half-width per |row offset| (hlines are centred, so the union is the max half-width).
void disc_halfwidths(int r, int* hw) {
  for (int i = 0; i <= r; i++) hw[i] = -1;
  int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
  while (dx >= dy) {
    hw[dy] = std::max(hw[dy], dx);  // rows cy±dy span cx±dx
    hw[dx] = std::max(hw[dx], dy);  // rows cy±dx span cx±dy
    dy++;
    err += plus;
    plus += 2;
    int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

void circle_fill(uint8_t* img, int W, int H, int cx, int cy, int r, uint8_t v) {
  std::vector<int> hw(r + 1);
  disc_halfwidths(r, hw.data());
  for (int oy = -r; oy <= r; oy++) {
    int y = cy + oy;
    if ((unsigned)y >= (unsigned)H) continue;
    int h = hw[std::abs(oy)];
    if (h < 0) continue;
    int x0 = std::max(cx - h, 0), x1 = std::min(cx + h, W - 1);
    for (int x = x0; x <= x1; x++) img[(size_t)y * W + x] = v;
  }
}

// ---------------------------------------------------------------- pyramid / Scharr [OpenCV]
// cv::pyrDown u8 (imgproc/src/pyramids.cpp): separable [1 4 6 4 1], (sum+128)>>8, REFLECT_101
void pyr_down(const uint8_t* src, int sw, int sh, uint8_t* dst) {
  const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  // horizontal pass on every source row (all-integer, so pass order does not change the result)
  std::vector<uint16_t> hbuf((size_t)sh * dw);
  const int xlo = 1, xhi = (sw - 3) / 2;  // 2x-2 >= 0 and 2x+2 <= sw-1
  for (int sy = 0; sy < sh; sy++) {
    const uint8_t* s = src + (size_t)sy * sw;
    uint16_t* row = hbuf.data() + (size_t)sy * dw;
    for (int x = 0; x < dw; x++) {
      if (x >= xlo && x <= xhi) continue;
      row[x] = (uint16_t)(s[reflect101(2 * x - 2, sw)] + s[reflect101(2 * x + 2, sw)] +
                          4 * (s[reflect101(2 * x - 1, sw)] + s[reflect101(2 * x + 1, sw)]) +
                          6 * s[reflect101(2 * x, sw)]);
    }
    for (int x = xlo; x <= xhi; x++) {
      const uint8_t* q = s + 2 * x;
      row[x] = (uint16_t)(q[-2] + q[2] + 4 * (q[-1] + q[1]) + 6 * q[0]);
    }
  }
  for (int y = 0; y < dh; y++) {
    const uint16_t* r0 = hbuf.data() + (size_t)reflect101(2 * y - 2, sh) * dw;
    const uint16_t* r1 = hbuf.data() + (size_t)reflect101(2 * y - 1, sh) * dw;
    const uint16_t* r2 = hbuf.data() + (size_t)reflect101(2 * y, sh) * dw;
    const uint16_t* r3 = hbuf.data() + (size_t)reflect101(2 * y + 1, sh) * dw;
    const uint16_t* r4 = hbuf.data() + (size_t)reflect101(2 * y + 2, sh) * dw;
    uint8_t* d = dst + (size_t)y * dw;
    for (int x = 0; x < dw; x++)
      d[x] = (uint8_t)((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
  }
}

// calcSharrDeriv (video/src/lkpyramid.cpp): Ix=[3 10 3]^T x [-1 0 1], Iy=[-1 0 1]^T x [3 10 3]
void scharr(const uint8_t* src, int cols, int rows, int16_t* dst) {
  std::vector<int> t0(cols + 2), t1(cols + 2);
  for (int y = 0; y < rows; y++) {
    const uint8_t* srow0 = src + (size_t)(y > 0 ? y - 1 : rows > 1 ? 1 : 0) * cols;
    const uint8_t* srow1 = src + (size_t)y * cols;
    const uint8_t* srow2 = src + (size_t)(y < rows - 1 ? y + 1 : rows > 1 ? rows - 2 : 0) * cols;
    int* trow0 = t0.data() + 1;
    int* trow1 = t1.data() + 1;
    for (int x = 0; x < cols; x++) {
      trow0[x] = (int16_t)((srow0[x] + srow2[x]) * 3 + srow1[x] * 10);
      trow1[x] = (int16_t)(srow2[x] - srow0[x]);
    }
    int x0 = (cols > 1 ? 1 : 0), x1 = (cols > 1 ? cols - 2 : 0);
    trow0[-1] = trow0[x0];
    trow0[cols] = trow0[x1];
    trow1[-1] = trow1[x0];
    trow1[cols] = trow1[x1];
    int16_t* drow = dst + (size_t)y * cols * 2;
    for (int x = 0; x < cols; x++) {
      drow[x * 2] = (int16_t)(trow0[x + 1] - trow0[x - 1]);
      drow[x * 2 + 1] = (int16_t)((trow1[x + 1] + trow1[x - 1]) * 3 + trow1[x] * 10);
    }
  }
}

// buildOpticalFlowPyramid level count (video/src/lkpyramid.cpp)
int pyr_levels(int w, int h, int win, int max_level) {
  int sw = w, sh = h;
  for (int level = 0; level <= max_level; ++level) {
    sw = (sw + 1) / 2;
    sh = (sh + 1) / 2;
    if (sw <= win || sh <= win) return level;
  }
  return max_level;
}

struct Pyr {
  int levels;  // = maxLevel (inclusive)
  int pad;     // = winSize: each level is stored padded by `pad` px, BORDER_REFLECT_101
  std::vector<int> w, h;
  std::vector<std::vector<uint8_t>> img;   // unpadded levels (pyrDown input)
  std::vector<std::vector<uint8_t>> pimg;  // padded levels, (w+2pad) x (h+2pad)
};

// copyMakeBorder(level, temp, win, win, win, win, BORDER_REFLECT_101) (buildOpticalFlowPyramid)
void pad_reflect(const uint8_t* src, int w, int h, int pad, std::vector<uint8_t>& dst) {
  const int pw = w + 2 * pad, ph = h + 2 * pad;
  dst.resize((size_t)pw * ph);
  std::vector<int> xmap(pw);
  for (int x = 0; x < pw; x++) xmap[x] = reflect101(x - pad, w);
  for (int y = 0; y < ph; y++) {
    const uint8_t* s = src + (size_t)reflect101(y - pad, h) * w;
    uint8_t* d = dst.data() + (size_t)y * pw;
    for (int x = 0; x < pw; x++) d[x] = s[xmap[x]];
  }
}

void build_pyr(const uint8_t* img, int w, int h, int win, int max_level, Pyr& p) {
  p.levels = pyr_levels(w, h, win, max_level);
  p.pad = win;
  p.w.assign(p.levels + 1, 0);
  p.h.assign(p.levels + 1, 0);
  p.img.assign(p.levels + 1, std::vector<uint8_t>());
  p.pimg.assign(p.levels + 1, std::vector<uint8_t>());
  p.w[0] = w;
  p.h[0] = h;
  p.img[0].assign(img, img + (size_t)w * h);
  for (int l = 1; l <= p.levels; l++) {
    p.w[l] = (p.w[l - 1] + 1) / 2;
    p.h[l] = (p.h[l - 1] + 1) / 2;
    p.img[l].resize((size_t)p.w[l] * p.h[l]);
    pyr_down(p.img[l - 1].data(), p.w[l - 1], p.h[l - 1], p.img[l].data());
  }
  for (int l = 0; l <= p.levels; l++) pad_reflect(p.img[l].data(), p.w[l], p.h[l], win, p.pimg[l]);
}

// calcSharrDeriv into the interior of a buffer padded by `pad` with BORDER_CONSTANT 0
void scharr_padded(const uint8_t* src, int w, int h, int pad, std::vector<int16_t>& dst) {
  const int pw = w + 2 * pad, ph = h + 2 * pad;
  dst.assign((size_t)pw * ph * 2, 0);
  std::vector<int16_t> tmp((size_t)w * h * 2);
  scharr(src, w, h, tmp.data());
  for (int y = 0; y < h; y++)
    std::memcpy(dst.data() + ((size_t)(y + pad) * pw + pad) * 2, tmp.data() + (size_t)y * w * 2,
                sizeof(int16_t) * 2 * w);
}

// diagnostics: total LK iterations / (point,level) visits since the last reset
unsigned long long g_lk_iters = 0, g_lk_visits = 0, g_lk_maxed = 0;
// ... and per forward+backward PAIR of calls (what one fused k_lk launch runs per point): the
// iterations of the slowest point and of the average point, summed over the pairs
std::vector<int> g_lk_pair;          // per-point iterations of the pair being measured
bool g_lk_pair_on = false;
unsigned long long g_lk_pairs = 0, g_lk_pair_max_sum = 0, g_lk_pair_pts = 0, g_lk_pair_iter_sum = 0,
                   g_lk_pair_max_max = 0;
void lk_pair_begin(int n) {
  g_lk_pair.assign((size_t)n, 0);
  g_lk_pair_on = true;
}
void lk_pair_end() {
  g_lk_pair_on = false;
  if (g_lk_pair.empty()) return;
  int mx = 0;
  unsigned long long sum = 0;
  for (int v : g_lk_pair) {
    mx = std::max(mx, v);
    sum += (unsigned long long)v;
  }
  g_lk_pairs++;
  g_lk_pair_max_sum += (unsigned long long)mx;
  g_lk_pair_max_max = std::max(g_lk_pair_max_max, (unsigned long long)mx);
  g_lk_pair_pts += g_lk_pair.size();
  g_lk_pair_iter_sum += sum;
}

// LKTrackerInvoker::operator() for all points at one level (video/src/lkpyramid.cpp) [OpenCV]
// I, J: level images padded by `win` with BORDER_REFLECT_101; dI: Scharr (Ix,Iy) of I padded by
// `win` with BORDER_CONSTANT 0 — exactly the buffers OpenCV's tracker indexes.  cols/rows are
// the unpadded level size; pointers address the padded buffers' origin.
void lk_level(const uint8_t* Ipad, const int16_t* dIpad, const uint8_t* Jpad, int cols, int rows,
              const float* prevPts, float* nextPts, uint8_t* status, int npoints, int win,
              int level, int maxLevel, int maxCount, double epsilon /*squared*/, int flags,
              float minEigThreshold, int accum) {
  const float halfWinX = (win - 1) * 0.5f, halfWinY = (win - 1) * 0.5f;
  const int W_BITS = 14, W_BITS1 = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  std::vector<int16_t> IWinBuf((size_t)win * win), dWinBuf((size_t)win * win * 2);
  const int stepI = cols + 2 * win, stepJ = stepI, dstep = stepI * 2;
  const uint8_t* I = Ipad + (size_t)win * stepI + win;      // -> pixel (0,0)
  const uint8_t* J = Jpad + (size_t)win * stepJ + win;
  const int16_t* derivI = dIpad + ((size_t)win * stepI + win) * 2;

  for (int ptidx = 0; ptidx < npoints; ptidx++) {
    float prevX = prevPts[ptidx * 2] * (float)(1. / (1 << level));
    float prevY = prevPts[ptidx * 2 + 1] * (float)(1. / (1 << level));
    float nextX, nextY;
    if (level == maxLevel) {
      if (flags & 4) {  // OPTFLOW_USE_INITIAL_FLOW
        nextX = nextPts[ptidx * 2] * (float)(1. / (1 << level));
        nextY = nextPts[ptidx * 2 + 1] * (float)(1. / (1 << level));
      } else {
        nextX = prevX;
        nextY = prevY;
      }
    } else {
      nextX = nextPts[ptidx * 2] * 2.f;
      nextY = nextPts[ptidx * 2 + 1] * 2.f;
    }
    nextPts[ptidx * 2] = nextX;
    nextPts[ptidx * 2 + 1] = nextY;

    prevX -= halfWinX;
    prevY -= halfWinY;
    int iprevX = cv_floor_f(prevX), iprevY = cv_floor_f(prevY);
    if (iprevX < -win || iprevX >= cols || iprevY < -win || iprevY >= rows) {
      if (level == 0) status[ptidx] = 0;
      continue;
    }
    float a = prevX - iprevX;
    float b = prevY - iprevY;
    int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    float fA11 = 0, fA12 = 0, fA22 = 0;
    float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0};
    int64_t iA11 = 0, iA12 = 0, iA22 = 0;
    for (int y = 0; y < win; y++) {
      const uint8_t* src = I + (ptrdiff_t)(y + iprevY) * stepI + iprevX;
      const int16_t* dsrc = derivI + (ptrdiff_t)(y + iprevY) * dstep + iprevX * 2;
      int16_t* Iptr = &IWinBuf[(size_t)y * win];
      int16_t* dIptr = &dWinBuf[(size_t)y * win * 2];
      for (int x = 0; x < win; x++, dsrc += 2, dIptr += 2) {
        int ival = CV_DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 +
                                  src[x + stepI + 1] * iw11,
                              W_BITS1 - 5);
        int ixval = CV_DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 +
                                   dsrc[dstep + 2] * iw11,
                               W_BITS1);
        int iyval = CV_DESCALE(dsrc[1] * iw00 + dsrc[2 + 1] * iw01 + dsrc[dstep + 1] * iw10 +
                                   dsrc[dstep + 2 + 1] * iw11,
                               W_BITS1);
        Iptr[x] = (int16_t)ival;
        dIptr[0] = (int16_t)ixval;
        dIptr[1] = (int16_t)iyval;
        if (accum == 0) {
          fA11 += (float)(ixval * ixval);
          fA12 += (float)(ixval * iyval);
          fA22 += (float)(iyval * iyval);
        } else if (accum == 2 || accum == 4) {
          // x86 SIMD128 build [OpenCV 4.2 lkpyramid.cpp, `#if CV_SIMD128 && !CV_NEON`, recalled]:
          // the vector loop runs while x <= win - 8 (8 pixels per step as two halves of 4), lane k
          // of qA11/qA12/qA22 takes pixel x = 4m + k; the rest of the row goes to the scalar float
          // accumulator.  (v_muladd without FMA = mul then add; the products are < 2^24: exact.)
          // accum 4 (and 2 on a host without SSE2): these lanes emulated one by one; accum 2 with SSE2:
          // real __m128 accumulators, after the row is complete (below).
          if (accum == 2 && ORACLE_LK_SSE2) {
            if (x >= (win / 8) * 8) {
              fA11 += (float)(ixval * ixval);
              fA12 += (float)(ixval * iyval);
              fA22 += (float)(iyval * iyval);
            }
            continue;
          }
          const float fx = (float)ixval, fy = (float)iyval;
          if (x < (win / 8) * 8) {
            qA11[x & 3] += fx * fx;
            qA12[x & 3] += fx * fy;
            qA22[x & 3] += fy * fy;
          } else {
            fA11 += (float)(ixval * ixval);
            fA12 += (float)(ixval * iyval);
            fA22 += (float)(iyval * iyval);
          }
        } else {
          iA11 += (int64_t)(ixval * ixval);
          iA12 += (int64_t)(ixval * iyval);
          iA22 += (int64_t)(iyval * iyval);
        }
      }
#if ORACLE_LK_SSE2
      if (accum == 2) {  // the row's vector part on real SSE registers: two float32x4 halves per step of 8
        __m128 a11 = _mm_loadu_ps(qA11), a12 = _mm_loadu_ps(qA12), a22 = _mm_loadu_ps(qA22);
        const int16_t* dp = &dWinBuf[(size_t)y * win * 2];
        for (int x = 0; x <= win - 8; x += 8)
          for (int half = 0; half < 2; half++) {
            const int16_t* q = dp + 2 * (x + 4 * half);
            const __m128 fx = _mm_cvtepi32_ps(_mm_setr_epi32(q[0], q[2], q[4], q[6]));
            const __m128 fy = _mm_cvtepi32_ps(_mm_setr_epi32(q[1], q[3], q[5], q[7]));
            a22 = _mm_add_ps(_mm_mul_ps(fy, fy), a22);  // v_muladd without FMA
            a12 = _mm_add_ps(_mm_mul_ps(fx, fy), a12);
            a11 = _mm_add_ps(_mm_mul_ps(fx, fx), a11);
          }
        _mm_storeu_ps(qA11, a11);
        _mm_storeu_ps(qA12, a12);
        _mm_storeu_ps(qA22, a22);
      }
#endif
    }
    float A11, A12, A22;
    if (accum == 0) {  // typedef float acctype (default OpenCV build, scalar loop order)
      A11 = fA11 * FLT_SCALE;
      A12 = fA12 * FLT_SCALE;
      A22 = fA22 * FLT_SCALE;
    } else if (accum == 2 || accum == 4) {  // iA11 += v_reduce_sum(qA11): SSE horizontal sum (q0+q2)+(q1+q3)
      fA11 += (qA11[0] + qA11[2]) + (qA11[1] + qA11[3]);
      fA12 += (qA12[0] + qA12[2]) + (qA12[1] + qA12[3]);
      fA22 += (qA22[0] + qA22[2]) + (qA22[1] + qA22[3]);
      A11 = fA11 * FLT_SCALE;
      A12 = fA12 * FLT_SCALE;
      A22 = fA22 * FLT_SCALE;
    } else {  // typedef int64 acctype (OpenCV's integer-accumulator build): exact sums
      A11 = (float)iA11 * FLT_SCALE;
      A12 = (float)iA12 * FLT_SCALE;
      A22 = (float)iA22 * FLT_SCALE;
    }
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                   (float)(2 * win * win);
    if (minEig < minEigThreshold || D < FLT_EPSILON) {
      if (level == 0) status[ptidx] = 0;
      continue;
    }
    D = 1.f / D;

    nextX -= halfWinX;
    nextY -= halfWinY;
    float prevDeltaX = 0, prevDeltaY = 0;
    g_lk_visits++;
    for (int j = 0; j < maxCount; j++) {
      g_lk_iters++;
      if (g_lk_pair_on && (size_t)ptidx < g_lk_pair.size()) g_lk_pair[ptidx]++;
      if (j == maxCount - 1) g_lk_maxed++;
      int inextX = cv_floor_f(nextX), inextY = cv_floor_f(nextY);
      if (inextX < -win || inextX >= cols || inextY < -win || inextY >= rows) {
        if (level == 0) status[ptidx] = 0;
        break;
      }
      a = nextX - inextX;
      b = nextY - inextY;
      iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
      iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
      iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      float fb1 = 0, fb2 = 0;
      float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0};
      int64_t ib1 = 0, ib2 = 0;
      for (int y = 0; y < win; y++) {
        const uint8_t* Jptr = J + (ptrdiff_t)(y + inextY) * stepJ + inextX;
        const int16_t* Iptr = &IWinBuf[(size_t)y * win];
        const int16_t* dIptr = &dWinBuf[(size_t)y * win * 2];
        if (accum == 0) {
          for (int x = 0; x < win; x++, dIptr += 2) {
            int diff = CV_DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                      Jptr[x + stepJ + 1] * iw11,
                                  W_BITS1 - 5) -
                       Iptr[x];
            fb1 += (float)(diff * dIptr[0]);
            fb2 += (float)(diff * dIptr[1]);
          }
        } else if (accum == 2 && ORACLE_LK_SSE2) {
#if ORACLE_LK_SSE2
          // The x86 SIMD128 build's loop as it is written there [OpenCV 4.2 lkpyramid.cpp, recalled], on real SSE2
          // registers — the lane order below comes from the hardware's unpack / pmaddwd / cvtdq2ps / addps, not
          // from an emulation of them (accum 4 is that emulation; tests/test_lk_float_orders.py holds the two
          // bit-identical): per step of 8 pixels the bilinear sums as pmaddwd of (J[x], J[x+1]) against
          // (iw00, iw01) and of the row below against (iw10, iw11), + delta, >> 9, packed to int16, minus the
          // patch; v_zip(diff, diff) / v_zip(dI lo, dI hi) / v_zip again put (It_k, It_k+4) against
          // (Ix_k, Ix_k+4) and (Iy_k, Iy_k+4); pmaddwd adds each pair in int32, cvtdq2ps rounds, one addps.
          const __m128i z = _mm_setzero_si128();
          const __m128i qw0 = _mm_set1_epi32(iw00 + (iw01 << 16)), qw1 = _mm_set1_epi32(iw10 + (iw11 << 16));
          const __m128i qdelta = _mm_set1_epi32(1 << (W_BITS1 - 5 - 1));
          __m128 vb0 = _mm_loadu_ps(qb0), vb1 = _mm_loadu_ps(qb1);
          int x = 0;
          for (; x <= win - 8; x += 8) {
            const __m128i v00 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x)), z);
            const __m128i v01 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x + 1)), z);
            const __m128i v10 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x + stepJ)), z);
            const __m128i v11 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x + stepJ + 1)), z);
            __m128i t0 = _mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(v00, v01), qw0),
                                       _mm_madd_epi16(_mm_unpacklo_epi16(v10, v11), qw1));
            __m128i t1 = _mm_add_epi32(_mm_madd_epi16(_mm_unpackhi_epi16(v00, v01), qw0),
                                       _mm_madd_epi16(_mm_unpackhi_epi16(v10, v11), qw1));
            t0 = _mm_srai_epi32(_mm_add_epi32(t0, qdelta), W_BITS1 - 5);
            t1 = _mm_srai_epi32(_mm_add_epi32(t1, qdelta), W_BITS1 - 5);
            const __m128i diff0 = _mm_sub_epi16(_mm_packs_epi32(t0, t1), _mm_loadu_si128((const __m128i*)(Iptr + x)));
            const __m128i diff2 = _mm_unpacklo_epi16(diff0, diff0);  // It0 It0 It1 It1 It2 It2 It3 It3
            const __m128i diff1 = _mm_unpackhi_epi16(diff0, diff0);  // It4 It4 ... It7 It7
            const __m128i d00 = _mm_loadu_si128((const __m128i*)(dIptr + 2 * x));      // Ix0 Iy0 ... Ix3 Iy3
            const __m128i d01 = _mm_loadu_si128((const __m128i*)(dIptr + 2 * x + 8));  // Ix4 Iy4 ... Ix7 Iy7
            const __m128i d10 = _mm_unpacklo_epi16(d00, d01);  // Ix0 Ix4 Iy0 Iy4 Ix1 Ix5 Iy1 Iy5
            const __m128i d11 = _mm_unpackhi_epi16(d00, d01);  // Ix2 Ix6 Iy2 Iy6 Ix3 Ix7 Iy3 Iy7
            const __m128i e00 = _mm_unpacklo_epi16(diff2, diff1);  // It0 It4 It0 It4 It1 It5 It1 It5
            const __m128i e01 = _mm_unpackhi_epi16(diff2, diff1);  // It2 It6 It2 It6 It3 It7 It3 It7
            vb0 = _mm_add_ps(vb0, _mm_cvtepi32_ps(_mm_madd_epi16(e00, d10)));
            vb1 = _mm_add_ps(vb1, _mm_cvtepi32_ps(_mm_madd_epi16(e01, d11)));
          }
          _mm_storeu_ps(qb0, vb0);
          _mm_storeu_ps(qb1, vb1);
          for (; x < win; x++) {
            const int diff = CV_DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                            Jptr[x + stepJ + 1] * iw11,
                                        W_BITS1 - 5) -
                             Iptr[x];
            fb1 += (float)(diff * dIptr[2 * x]);
            fb2 += (float)(diff * dIptr[2 * x + 1]);
          }
#endif
        } else if (accum == 2 || accum == 4) {
          // x86 SIMD128 build [OpenCV 4.2, recalled]: per step of 8 pixels x0..x0+7 the zips put
          // (It_k, It_k+4) against (Ix_k, Ix_k+4) / (Iy_k, Iy_k+4), v_dotprod (pmaddwd) adds each
          // pair exactly in int32, v_cvt_f32 rounds, then one float add per lane:
          //   qb0 = {bx(0,4), by(0,4), bx(1,5), by(1,5)},  qb1 = {bx(2,6), by(2,6), bx(3,7), by(3,7)}
          int dv[64];  // win <= 64 (21 at every call site)
          for (int x = 0; x < win; x++)
            dv[x] = CV_DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                   Jptr[x + stepJ + 1] * iw11,
                               W_BITS1 - 5) -
                    Iptr[x];
          int x = 0;
          for (; x <= win - 8; x += 8) {
            for (int k = 0; k < 4; k++) {
              const int sx = dv[x + k] * dIptr[2 * (x + k)] + dv[x + k + 4] * dIptr[2 * (x + k + 4)];
              const int sy = dv[x + k] * dIptr[2 * (x + k) + 1] + dv[x + k + 4] * dIptr[2 * (x + k + 4) + 1];
              float* q = k < 2 ? qb0 : qb1;
              q[(k & 1) * 2] += (float)sx;
              q[(k & 1) * 2 + 1] += (float)sy;
            }
          }
          for (; x < win; x++) {
            fb1 += (float)(dv[x] * dIptr[2 * x]);
            fb2 += (float)(dv[x] * dIptr[2 * x + 1]);
          }
        } else {
          int64_t r1 = 0, r2 = 0;  // row sums fit easily; exact
          for (int x = 0; x < win; x++, dIptr += 2) {
            int diff = CV_DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                      Jptr[x + stepJ + 1] * iw11,
                                  W_BITS1 - 5) -
                       Iptr[x];
            r1 += (int64_t)(diff * dIptr[0]);
            r2 += (int64_t)(diff * dIptr[1]);
          }
          ib1 += r1;
          ib2 += r2;
        }
      }
      float b1, b2;
      if (accum == 0) {
        b1 = fb1 * FLT_SCALE;
        b2 = fb2 * FLT_SCALE;
      } else if (accum == 2 || accum == 4) {
        // v_recombine(v_interleave_pairs(qb0 + qb1), 0, qf0, qf1); ib1 += v_reduce_sum(qf0) ...
        const float s0 = qb0[0] + qb1[0], s1 = qb0[1] + qb1[1], s2 = qb0[2] + qb1[2], s3 = qb0[3] + qb1[3];
        fb1 += (s0 + 0.f) + (s2 + 0.f);
        fb2 += (s1 + 0.f) + (s3 + 0.f);
        b1 = fb1 * FLT_SCALE;
        b2 = fb2 * FLT_SCALE;
      } else {
        b1 = (float)ib1 * FLT_SCALE;
        b2 = (float)ib2 * FLT_SCALE;
      }
      float deltaX = (float)((A12 * b2 - A22 * b1) * D);
      float deltaY = (float)((A12 * b1 - A11 * b2) * D);
      nextX += deltaX;
      nextY += deltaY;
      nextPts[ptidx * 2] = nextX + halfWinX;
      nextPts[ptidx * 2 + 1] = nextY + halfWinY;
      if ((double)deltaX * deltaX + (double)deltaY * deltaY <= epsilon) break;
      if (j > 0 && std::abs(deltaX + prevDeltaX) < 0.01 && std::abs(deltaY + prevDeltaY) < 0.01) {
        nextPts[ptidx * 2] -= deltaX * 0.5f;
        nextPts[ptidx * 2 + 1] -= deltaY * 0.5f;
        break;
      }
      prevDeltaX = deltaX;
      prevDeltaY = deltaY;
    }
    // the reference passes an `err` vector and no GET_MIN_EIGENVALS flag, so the err block
    // runs at level 0 and re-validates the final window position
    if (status[ptidx] && level == 0) {
      float npX = nextPts[ptidx * 2] - halfWinX, npY = nextPts[ptidx * 2 + 1] - halfWinY;
      int ix = cv_floor_f(npX), iy = cv_floor_f(npY);
      if (ix < -win || ix >= cols || iy < -win || iy >= rows) status[ptidx] = 0;
    }
  }
}

void lk_pyr(const Pyr& P, const std::vector<std::vector<int16_t>>& dP, const Pyr& N,
            const float* prev_pts, float* next_pts, uint8_t* status, int n, int win,
            int max_count, double eps, int flags, int accum, int maxLevel) {
  // TermCriteria normalisation (lkpyramid.cpp calc()): COUNT and EPS both set by every call site
  max_count = std::min(std::max(max_count, 0), 100);
  eps = std::min(std::max(eps, 0.), 10.);
  eps *= eps;
  for (int i = 0; i < n; i++) status[i] = 1;
  for (int level = maxLevel; level >= 0; level--)
    lk_level(P.pimg[level].data(), dP[level].data(), N.pimg[level].data(), P.w[level], P.h[level],
             prev_pts, next_pts, status, n, win, level, maxLevel, max_count, eps, flags, 1e-4f,
             accum);
}

// cv::calcOpticalFlowPyrLK: both pyramids and the prev-image derivatives are rebuilt per call
void calc_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts,
             float* next_pts, uint8_t* status, int n, int win, int max_level, int max_count,
             double eps, int flags, int accum) {
  Pyr P, N;
  build_pyr(prev, w, h, win, max_level, P);
  build_pyr(next, w, h, win, max_level, N);
  std::vector<std::vector<int16_t>> dP(P.levels + 1);
  for (int l = 0; l <= P.levels; l++) scharr_padded(P.img[l].data(), P.w[l], P.h[l], win, dP[l]);
  if (!(flags & 4))
    for (int i = 0; i < 2 * n; i++) next_pts[i] = 0.f;  // _nextPts.create(): overwritten at top level
  lk_pyr(P, dP, N, prev_pts, next_pts, status, n, win, max_count, eps, flags, accum, P.levels);
}

// ---------------------------------------------------------------- CLAHE + normalize [OpenCV]
// cv::createCLAHE() defaults: clipLimit 40.0, tileGridSize 8x8 (feature_tracker.cpp:377-379);
// imgproc/src/clahe.cpp for CV_8UC1: per-tile clipped histogram -> LUT, bilinear LUT blending.
void clahe_apply(const uint8_t* src, int W, int H, uint8_t* dst) {
  const int tilesX = 8, tilesY = 8, histSize = 256;
  const double clipLimitD = 40.0;
  // extend with BORDER_REFLECT_101 when the size is not divisible by the grid
  int EW = W, EH = H;
  if (!(W % tilesX == 0 && H % tilesY == 0)) {
    EW = W + (tilesX - (W % tilesX));
    EH = H + (tilesY - (H % tilesY));
  }
  const int tw = EW / tilesX, th = EH / tilesY;
  const int tileSizeTotal = tw * th;
  const float lutScale = static_cast<float>(histSize - 1) / tileSizeTotal;
  int clipLimit = static_cast<int>(clipLimitD * tileSizeTotal / histSize);
  clipLimit = std::max(clipLimit, 1);
  std::vector<uint8_t> lut((size_t)tilesX * tilesY * histSize);
  for (int k = 0; k < tilesX * tilesY; k++) {  // CLAHE_CalcLut_Body
    const int ty = k / tilesX, tx = k % tilesX;
    int tileHist[256] = {0};
    for (int y = ty * th; y < (ty + 1) * th; y++)
      for (int x = tx * tw; x < (tx + 1) * tw; x++)
        tileHist[src[(size_t)reflect101(y, H) * W + reflect101(x, W)]]++;
    int clipped = 0;
    for (int i = 0; i < histSize; ++i) {
      if (tileHist[i] > clipLimit) {
        clipped += tileHist[i] - clipLimit;
        tileHist[i] = clipLimit;
      }
    }
    int redistBatch = clipped / histSize;
    int residual = clipped - redistBatch * histSize;
    for (int i = 0; i < histSize; ++i) tileHist[i] += redistBatch;
    if (residual != 0) {
      int residualStep = std::max(histSize / residual, 1);
      for (int i = 0; i < histSize && residual > 0; i += residualStep, residual--) tileHist[i]++;
    }
    int sum = 0;
    uint8_t* tileLut = &lut[(size_t)k * histSize];
    for (int i = 0; i < histSize; ++i) {
      sum += tileHist[i];
      tileLut[i] = saturate_u8(cv_round_f(sum * lutScale));
    }
  }
  // CLAHE_Interpolation_Body
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  for (int y = 0; y < H; y++) {
    float tyf = y * inv_th - 0.5f;
    int ty1 = cv_floor_f(tyf);
    int ty2 = ty1 + 1;
    float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = std::max(ty1, 0);
    ty2 = std::min(ty2, tilesY - 1);
    const uint8_t* lutPlane1 = &lut[(size_t)ty1 * tilesX * histSize];
    const uint8_t* lutPlane2 = &lut[(size_t)ty2 * tilesX * histSize];
    for (int x = 0; x < W; x++) {
      float txf = x * inv_tw - 0.5f;
      int tx1 = cv_floor_f(txf);
      int tx2 = tx1 + 1;
      float xa = txf - tx1, xa1 = 1.0f - xa;
      tx1 = std::max(tx1, 0);
      tx2 = std::min(tx2, tilesX - 1);
      int srcVal = src[(size_t)y * W + x];
      int ind1 = tx1 * histSize + srcVal;
      int ind2 = tx2 * histSize + srcVal;
      float res = (lutPlane1[ind1] * xa1 + lutPlane1[ind2] * xa) * ya1 +
                  (lutPlane2[ind1] * xa1 + lutPlane2[ind2] * xa) * ya;
      dst[(size_t)y * W + x] = saturate_u8(cv_round_f(res));
    }
  }
}

// cv::normalize(img, img, 0, 255, NORM_MINMAX) on CV_8U (core/src/norm.cpp + convertTo 8u->8u with
// float scale/shift, unfused multiply-add — the SSE2 baseline of convert_scale)
void normalize_minmax_u8(uint8_t* img, size_t n) {
  if (!n) return;
  int mn = 255, mx = 0;
  for (size_t i = 0; i < n; i++) {
    mn = std::min(mn, (int)img[i]);
    mx = std::max(mx, (int)img[i]);
  }
  const double smin = mn, smax = mx, dmin = 0, dmax = 255;
  const double scale = (dmax - dmin) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
  const double shift = dmin - smin * scale;
  const float a = (float)scale, b = (float)shift;
  for (size_t i = 0; i < n; i++) img[i] = saturate_u8(cv_round_f(img[i] * a + b));
}

// ---------------------------------------------------------------- camera
// PinholeCamera::liftProjective (camera_model/src/camera_models/PinholeCamera.cc:450-510),
// distortion (:646-662), m_inv_K* (:824-827)
void distortion(const oracle_camera* c, double ux, double uy, double* dx, double* dy) {
  double k1 = c->k1, k2 = c->k2, p1 = c->p1, p2 = c->p2;
  double mx2_u = ux * ux, my2_u = uy * uy, mxy_u = ux * uy;
  double rho2_u = mx2_u + my2_u;
  double rad_dist_u = k1 * rho2_u + k2 * rho2_u * rho2_u;
  *dx = ux * rad_dist_u + 2.0 * p1 * mxy_u + p2 * (rho2_u + 2.0 * mx2_u);
  *dy = uy * rad_dist_u + 2.0 * p2 * mxy_u + p1 * (rho2_u + 2.0 * my2_u);
}

void lift_projective(const oracle_camera* c, double u, double v, double* P) {
  double inv_K11 = 1.0 / c->fx, inv_K13 = -c->cx / c->fx;
  double inv_K22 = 1.0 / c->fy, inv_K23 = -c->cy / c->fy;
  double mx_d = inv_K11 * u + inv_K13;
  double my_d = inv_K22 * v + inv_K23;
  double mx_u, my_u;
  bool noDistortion = (c->k1 == 0.0) && (c->k2 == 0.0) && (c->p1 == 0.0) && (c->p2 == 0.0);
  if (noDistortion) {
    mx_u = mx_d;
    my_u = my_d;
  } else {
    int n = 8;
    double dx, dy;
    distortion(c, mx_d, my_d, &dx, &dy);
    mx_u = mx_d - dx;
    my_u = my_d - dy;
    for (int i = 1; i < n; ++i) {
      distortion(c, mx_u, my_u, &dx, &dy);
      mx_u = mx_d - dx;
      my_u = my_d - dy;
    }
  }
  P[0] = mx_u;
  P[1] = my_u;
  P[2] = 1.0;
}

// ---------------------------------------------------------------- F-RANSAC [OpenCV calib3d]
struct CvRNG {  // cv::RNG (core/operations.hpp): multiply-with-carry
  uint64_t state;
  explicit CvRNG(uint64_t s) : state(s ? s : 0xffffffff) {}
  unsigned next() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// cv::hypot: the template of core/src/lapack.cpp (declared above JacobiImpl_) that JacobiSVDImpl_'s
// unqualified `hypot((double)p, beta)` resolves to inside namespace cv — not libm's hypot
// [OpenCV 4.2, recalled]
static double cv_hypot(double a, double b) {
  a = std::fabs(a);
  b = std::fabs(b);
  if (a > b) {
    b /= a;
    return a * std::sqrt(1 + b * b);
  }
  if (b > 0) {
    a /= b;
    return b * std::sqrt(1 + a * a);
  }
  return 0;
}

// One-sided Jacobi SVD as OpenCV's core/src/lapack.cpp runs it for doubles (JacobiSVDImpl_ with
// minval = DBL_MIN, eps = DBL_EPSILON*10; restated from the published algorithm, OpenCV 4.2 is not in
// /root/reference).  `at` holds n rows of length m (row stride `step`), m >= n: the rows are rotated
// pairwise (Hestenes) until they are mutually orthogonal, their norms are the singular values (sorted
// in decreasing order, rows swapped along), and each row is divided by its norm.  Rows n..n1-1 — the
// part of a FULL_UV basis the data does not determine — are made the way OpenCV makes them: a
// vector of +-1/m drawn from cv::RNG(0x12345678) (bit 8 of each 32-bit output), two rounds of
// Gram-Schmidt against the rows above with an L1 rescale after each projection, then an L2
// normalisation; the same is done for a row whose singular value is <= minval.
// (The rotations of the second factor, V^T in OpenCV's call, do not feed back into `at` and are
// left out.)
void jacobi_svd_rows(double* at, int step, double* w, int m, int n, int n1) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  double W[16];
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) sd += at[i * step + k] * at[i * step + k];
    W[i] = sd;
  }
  const int max_iter = std::max(m, 30);
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        double *ai = at + i * step, *aj = at + j * step;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += ai[k] * aj[k];
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = cv_hypot(p, beta);
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = std::sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const double t0 = c * ai[k] + s * aj[k];
          const double t1 = -s * ai[k] + c * aj[k];
          ai[k] = t0;
          aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) sd += at[i * step + k] * at[i * step + k];
    W[i] = std::sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {  // selection sort, largest first
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < m; k++) std::swap(at[i * step + k], at[j * step + k]);
    }
  }
  for (int i = 0; i < n; i++) w[i] = W[i];
  CvRNG rng(0x12345678);
  for (int i = 0; i < n1; i++) {
    double sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      const double val0 = 1. / m;
      for (int k = 0; k < m; k++) at[i * step + k] = (rng.next() & 256) != 0 ? val0 : -val0;
      for (int iter = 0; iter < 2; iter++)
        for (int j = 0; j < i; j++) {
          sd = 0;
          for (int k = 0; k < m; k++) sd += at[i * step + k] * at[j * step + k];
          double asum = 0;
          for (int k = 0; k < m; k++) {
            const double t = at[i * step + k] - sd * at[j * step + k];
            at[i * step + k] = t;
            asum += std::fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < m; k++) at[i * step + k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < m; k++) sd += at[i * step + k] * at[i * step + k];
      sd = std::sqrt(sd);
    }
    const double s = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < m; k++) at[i * step + k] *= s;
  }
}

int g_nullspace_mode = 0;  // 0: cv::SVDecomp's route (what run7Point calls); 1: Householder QR

// FMEstimatorCallback::run7Point's SVDecomp(A, W, U, Vt, MODIFY_A + FULL_UV) for the 7x9 system: rows
// < cols, so cv::SVD works on A itself as the row set (m = 9, n = 7) and completes it to 9 rows;
// f1, f2 are rows 7 and 8 of that V^T.
void nullspace_7x9_svd(const double* a /*7x9 row-major*/, double* f1, double* f2) {
  double at[9 * 9] = {0}, w[7];
  std::memcpy(at, a, 7 * 9 * sizeof(double));
  jacobi_svd_rows(at, 9, w, 9, 7, 9);
  std::memcpy(f1, at + 7 * 9, 9 * sizeof(double));
  std::memcpy(f2, at + 8 * 9, 9 * sizeof(double));
}

// The same plane from a Householder QR of A^T (last two columns of Q): any orthonormal basis of the
// 2-D null space yields the same cubic roots / F matrices up to rounding.  Kept as the comparison
// that measures how much the choice of basis moves RANSAC's inlier decisions
// (tests/test_ransac_nullspace.py); not what the product does.
void nullspace_7x9_qr(const double* a /*7x9 row-major*/, double* f1, double* f2) {
  const int m = 9, n = 7;
  double R[9][7];
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) R[i][j] = a[j * 9 + i];
  double Q[9][9];
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) Q[i][j] = (i == j);
  for (int k = 0; k < n; k++) {
    double norm = 0;
    for (int i = k; i < m; i++) norm += R[i][k] * R[i][k];
    norm = std::sqrt(norm);
    if (norm == 0) continue;
    double alpha = R[k][k] > 0 ? -norm : norm;
    double v[9] = {0};
    for (int i = k; i < m; i++) v[i] = R[i][k];
    v[k] -= alpha;
    double vn = 0;
    for (int i = k; i < m; i++) vn += v[i] * v[i];
    if (vn == 0) continue;
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int i = k; i < m; i++) s += v[i] * R[i][j];
      s = 2 * s / vn;
      for (int i = k; i < m; i++) R[i][j] -= s * v[i];
    }
    for (int j = 0; j < m; j++) {  // Q = Q * H
      double s = 0;
      for (int i = k; i < m; i++) s += Q[j][i] * v[i];
      s = 2 * s / vn;
      for (int i = k; i < m; i++) Q[j][i] -= s * v[i];
    }
  }
  for (int i = 0; i < 9; i++) {
    f1[i] = Q[i][7];
    f2[i] = Q[i][8];
  }
}

void nullspace_7x9(const double* a, double* f1, double* f2) {
  if (g_nullspace_mode == 1)
    nullspace_7x9_qr(a, f1, f2);
  else
    nullspace_7x9_svd(a, f1, f2);
}

// cv::solveCubic (core/src/mathfuncs.cpp)
int solve_cubic(const double* c, double* r) {
  double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
  double x0 = 0., x1 = 0., x2 = 0.;
  int n = 0;
  if (a0 == 0) {
    if (a1 == 0) {
      if (a2 == 0)
        n = a3 == 0 ? -1 : 0;
      else {
        x0 = -a3 / a2;
        n = 1;
      }
    } else {
      double d = a2 * a2 - 4 * a1 * a3;
      if (d >= 0) {
        d = std::sqrt(d);
        double q1 = (-a2 + d) * 0.5;
        double q2 = (a2 + d) * -0.5;
        if (std::fabs(q1) > std::fabs(q2)) {
          x0 = q1 / a1;
          x1 = a3 / q1;
        } else {
          x0 = q2 / a1;
          x1 = a3 / q2;
        }
        n = d > 0 ? 2 : 1;
      }
    }
  } else {
    a0 = 1. / a0;
    a1 *= a0;
    a2 *= a0;
    a3 *= a0;
    double Q = (a1 * a1 - 3 * a2) * (1. / 9);
    double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
    double Qcubed = Q * Q * Q;
    double d = Qcubed - R * R;
    const double PI = 3.1415926535897932384626433832795;
    if (d > 0) {
      double theta = std::acos(R / std::sqrt(Qcubed));
      double sqrtQ = std::sqrt(Q);
      double t0 = -2 * sqrtQ;
      double t1 = theta * (1. / 3);
      double t2 = a1 * (1. / 3);
      x0 = t0 * std::cos(t1) - t2;
      x1 = t0 * std::cos(t1 + (2. * PI / 3)) - t2;
      x2 = t0 * std::cos(t1 + (4. * PI / 3)) - t2;
      n = 3;
    } else if (d == 0) {
      if (R >= 0) {
        x0 = -2 * std::pow(R, 1. / 3) - a1 / 3;
        x1 = std::pow(R, 1. / 3) - a1 / 3;
      } else {
        x0 = 2 * std::pow(-R, 1. / 3) - a1 / 3;
        x1 = -std::pow(-R, 1. / 3) - a1 / 3;
      }
      x2 = 0;
      n = x0 == x1 ? 1 : 2;
      x1 = x0 == x1 ? 0 : x1;
    } else {
      double e;
      d = std::sqrt(-d);
      e = std::pow(d + std::fabs(R), 1. / 3);
      if (R > 0) e = -e;
      x0 = (e + Q / e) - a1 * (1. / 3);
      n = 1;
    }
  }
  r[0] = x0;
  r[1] = x1;
  r[2] = x2;
  return n;
}

// FMEstimatorCallback::run7Point (calib3d/src/fundam.cpp)
int run_7point(const float* m1, const float* m2, double* fmatrix /*up to 27*/) {
  double a[7 * 9], c[4], r[3] = {0};
  double f1[9], f2[9];
  for (int i = 0; i < 7; i++) {
    double x0 = m1[i * 2], y0 = m1[i * 2 + 1];
    double x1 = m2[i * 2], y1 = m2[i * 2 + 1];
    a[i * 9 + 0] = x1 * x0;
    a[i * 9 + 1] = x1 * y0;
    a[i * 9 + 2] = x1;
    a[i * 9 + 3] = y1 * x0;
    a[i * 9 + 4] = y1 * y0;
    a[i * 9 + 5] = y1;
    a[i * 9 + 6] = x0;
    a[i * 9 + 7] = y0;
    a[i * 9 + 8] = 1;
  }
  nullspace_7x9(a, f1, f2);
  for (int i = 0; i < 9; i++) f1[i] -= f2[i];
  double t0 = f2[4] * f2[8] - f2[5] * f2[7];
  double t1 = f2[3] * f2[8] - f2[5] * f2[6];
  double t2 = f2[3] * f2[7] - f2[4] * f2[6];
  c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
  c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
         f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
         f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
         f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
  t0 = f1[4] * f1[8] - f1[5] * f1[7];
  t1 = f1[3] * f1[8] - f1[5] * f1[6];
  t2 = f1[3] * f1[7] - f1[4] * f1[6];
  c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
  c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
         f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
         f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
         f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
  int n = solve_cubic(c, r);
  if (n < 1 || n > 3) return n;
  for (int k = 0; k < n; k++, fmatrix += 9) {
    double lambda = r[k], mu = 1.;
    double s = f1[8] * r[k] + f2[8];
    if (std::fabs(s) > DBL_EPSILON) {
      mu = 1. / s;
      lambda *= mu;
      fmatrix[8] = 1.;
    } else
      fmatrix[8] = 0.;
    for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
  }
  return n;
}

// FMEstimatorCallback::computeError
void fm_compute_error(const float* m1, const float* m2, int count, const double* F, float* err) {
  for (int i = 0; i < count; i++) {
    double a, b, c, d1, d2, s1, s2;
    double x1 = m1[i * 2], y1 = m1[i * 2 + 1], x2 = m2[i * 2], y2 = m2[i * 2 + 1];
    a = F[0] * x1 + F[1] * y1 + F[2];
    b = F[3] * x1 + F[4] * y1 + F[5];
    c = F[6] * x1 + F[7] * y1 + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = x1 * a + y1 * b + c;
    err[i] = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
  }
}

bool have_collinear(const float* m, int count) {
  int i = count - 1;
  for (int j = 0; j < i; j++) {
    double dx1 = m[j * 2] - m[i * 2];
    double dy1 = m[j * 2 + 1] - m[i * 2 + 1];
    for (int k = 0; k < j; k++) {
      double dx2 = m[k * 2] - m[i * 2];
      double dy2 = m[k * 2 + 1] - m[i * 2 + 1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <=
          FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
        return true;
    }
  }
  return false;
}

// RANSACPointSetRegistrator::getSubset (calib3d/src/ptsetreg.cpp), modelPoints = 7
bool get_subset(const float* m1, const float* m2, int count, float* ms1, float* ms2, CvRNG& rng,
                int maxAttempts) {
  const int modelPoints = 7;
  int idx[7];
  int i = 0, j, iters = 0;
  for (; iters < maxAttempts; iters++) {
    for (i = 0; i < modelPoints && iters < maxAttempts;) {
      int idx_i = 0;
      for (;;) {
        idx_i = idx[i] = rng.uniform(0, count);
        for (j = 0; j < i; j++)
          if (idx_i == idx[j]) break;
        if (j == i) break;
      }
      ms1[i * 2] = m1[idx_i * 2];
      ms1[i * 2 + 1] = m1[idx_i * 2 + 1];
      ms2[i * 2] = m2[idx_i * 2];
      ms2[i * 2 + 1] = m2[idx_i * 2 + 1];
      i++;
    }
    if (i == modelPoints && (have_collinear(ms1, i) || have_collinear(ms2, i))) continue;
    break;
  }
  return i == modelPoints && iters < maxAttempts;
}

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = std::max(p, 0.);
  p = std::min(p, 1.);
  ep = std::max(ep, 0.);
  ep = std::min(ep, 1.);
  double num = std::max(1. - p, DBL_MIN);
  double denom = 1. - std::pow(1. - ep, modelPoints);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv_round_d(num / denom);
}

int find_inliers(const float* m1, const float* m2, int count, const double* F, float* err,
                 uint8_t* mask, double thresh) {
  fm_compute_error(m1, m2, count, F, err);
  float t = (float)(thresh * thresh);
  int nz = 0;
  for (int i = 0; i < count; i++) {
    int f = err[i] <= t;
    mask[i] = (uint8_t)f;
    nz += f;
  }
  return nz;
}

// cv::findFundamentalMat(..., FM_RANSAC, thr, conf, mask): RANSAC for n>=15, LMedS for 8..14
int find_fundamental(const float* m1, const float* m2, int count, double thr, double conf,
                     uint8_t* status, double* Fout) {
  const int modelPoints = 7, maxIters = 1000;
  for (int i = 0; i < count; i++) status[i] = 0;
  if (count < 7) return 0;
  if (thr <= 0) thr = 3;
  if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
  std::vector<float> err(count);
  std::vector<uint8_t> mask(count);
  float ms1[14], ms2[14];
  double model[27], best[9];
  CvRNG rng((uint64_t)-1);
  if (count == 7) {
    int n = run_7point(m1, m2, model);
    for (int i = 0; i < count; i++) status[i] = 1;
    if (n <= 0) return 0;
    if (Fout) std::memcpy(Fout, model, sizeof(double) * 9);
    return count;
  }
  if (count >= 15) {  // RANSACPointSetRegistrator::run
    int niters = std::max(maxIters, 1), maxGoodCount = 0;
    for (int iter = 0; iter < niters; iter++) {
      bool found = get_subset(m1, m2, count, ms1, ms2, rng, 10000);
      if (!found) {
        if (iter == 0) return 0;
        break;
      }
      int nmodels = run_7point(ms1, ms2, model);
      if (nmodels <= 0) continue;
      for (int i = 0; i < nmodels; i++) {
        int goodCount = find_inliers(m1, m2, count, model + 9 * i, err.data(), mask.data(), thr);
        if (goodCount > std::max(maxGoodCount, modelPoints - 1)) {
          std::memcpy(status, mask.data(), count);
          std::memcpy(best, model + 9 * i, sizeof(best));
          maxGoodCount = goodCount;
          niters = ransac_update_num_iters(conf, (double)(count - goodCount) / count, modelPoints,
                                           niters);
        }
      }
    }
    if (maxGoodCount > 0 && Fout) std::memcpy(Fout, best, sizeof(best));
    return maxGoodCount;
  }
  // LMeDSPointSetRegistrator::run
  const double outlierRatio = 0.45;
  int niters = ransac_update_num_iters(conf, outlierRatio, modelPoints, maxIters);
  niters = std::max(niters, 3);
  double minMedian = DBL_MAX;
  std::vector<float> errs(count);
  for (int iter = 0; iter < niters; iter++) {
    bool found = get_subset(m1, m2, count, ms1, ms2, rng, 1000);  // getSubset's default maxAttempts
    if (!found) {
      if (iter == 0) return 0;
      break;
    }
    int nmodels = run_7point(ms1, ms2, model);
    if (nmodels <= 0) continue;
    for (int i = 0; i < nmodels; i++) {
      fm_compute_error(m1, m2, count, model + 9 * i, errs.data());
      // OpenCV nth_element's the float bits as int; errors are >= 0 so the order is the same
      std::nth_element(errs.begin(), errs.begin() + count / 2, errs.end());
      double median = errs[count / 2];
      if (median < minMedian) {
        minMedian = median;
        std::memcpy(best, model + 9 * i, sizeof(best));
      }
    }
  }
  if (minMedian < DBL_MAX) {
    double sigma = 2.5 * 1.4826 * (1 + 5. / (count - modelPoints)) * std::sqrt(minMedian);
    sigma = std::max(sigma, 0.001);
    int cnt = find_inliers(m1, m2, count, best, err.data(), status, sigma);
    if (Fout) std::memcpy(Fout, best, sizeof(best));
    return cnt;
  }
  return 0;
}

// ---------------------------------------------------------------- FeatureTracker
struct P2f {
  float x, y;
};

struct Tracker {
  oracle_config cfg;
  Detector det;
  bool detector_nostart = true;
  int n_id = 0;
  double cur_time = 0, prev_time = 0;
  std::vector<uint8_t> prev_img_left, cur_img_left, cur_img_right, ts_left, ts_right;
  std::vector<P2f> prev_pts, cur_pts, cur_right_pts, n_pts;
  std::vector<P2f> cur_un_pts, cur_un_right_pts, pts_velocity, right_pts_velocity;
  std::vector<int> ids, ids_right, track_cnt, track_cnt_right;
  std::map<int, P2f> cur_un_pts_map, prev_un_pts_map, cur_un_right_pts_map, prev_un_right_pts_map;
  std::vector<uint8_t> mask_event;  // 255 = blocked (reference: CV_64FC1 0.0/255.0)
  double stage_s[6] = {0, 0, 0, 0, 0, 0};
};

template <class T>
void reduce_vector(std::vector<T>& v, const std::vector<uint8_t>& status) {  // feature_tracker.cpp:56-81
  int j = 0;
  for (int i = 0; i < int(v.size()); i++)
    if (status[i]) v[j++] = v[i];
  v.resize(j);
}

bool in_border_event(const Tracker* t, const P2f& pt) {  // feature_tracker.cpp:48-54
  const int BORDER_SIZE = 1;
  int img_x = cv_round_f(pt.x);
  int img_y = cv_round_f(pt.y);
  return BORDER_SIZE <= img_x && img_x < t->cfg.width - BORDER_SIZE && BORDER_SIZE <= img_y &&
         img_y < t->cfg.height - BORDER_SIZE;
}

double pt_distance(const P2f& a, const P2f& b) {  // feature_tracker.cpp:1314-1319
  double dx = a.x - b.x;
  double dy = a.y - b.y;
  return std::sqrt(dx * dx + dy * dy);
}

// FeatureTracker::Event_setMask (feature_tracker.cpp:123-151)
void event_set_mask(Tracker* t) {
  const int W = t->cfg.width, H = t->cfg.height;
  t->mask_event.assign((size_t)W * H, 0);
  std::vector<std::pair<int, std::pair<P2f, int>>> cnt_pts_id;
  for (unsigned int i = 0; i < t->cur_pts.size(); i++)
    cnt_pts_id.push_back(std::make_pair(t->track_cnt[i], std::make_pair(t->cur_pts[i], t->ids[i])));
  std::sort(cnt_pts_id.begin(), cnt_pts_id.end(),
            [](const std::pair<int, std::pair<P2f, int>>& a,
               const std::pair<int, std::pair<P2f, int>>& b) { return a.first > b.first; });
  t->cur_pts.clear();
  t->ids.clear();
  t->track_cnt.clear();
  for (auto& it : cnt_pts_id) {
    int px = cv_round_f(it.second.first.x), py = cv_round_f(it.second.first.y);
    if (t->mask_event[(size_t)py * W + px] == 0) {
      t->cur_pts.push_back(it.second.first);
      t->ids.push_back(it.second.second);
      t->track_cnt.push_back(it.first);
      circle_fill(t->mask_event.data(), W, H, px, py, t->cfg.min_dist, 255);
    }
  }
}

// Event_FeaturesToTrack (feature_tracker.cpp:13-38)
int features_to_track(const Detector* d, const oracle_event* ev, size_t n, int maxCorners,
                      int min_dist, const uint8_t* event_mask, const uint8_t* ts,
                      double ts_lk_threshold, std::vector<P2f>& n_pts, std::vector<int>* idx_out) {
  int ncorners = 0;
  n_pts.clear();
  if (idx_out) idx_out->clear();
  std::vector<uint8_t> event_mask_cur(event_mask, event_mask + (size_t)d->W * d->H);
  if (maxCorners > 0) {
    for (size_t i = 0; i < n; i++) {
      const oracle_event& e = ev[i];
      if (ncorners >= maxCorners) break;
      if (e.x >= d->W || e.y >= d->H) continue;  // reference would abort on such input
      if (event_mask_cur[(size_t)e.y * d->W + e.x] != 255) {
        if ((double)ts[(size_t)e.y * d->W + e.x] != ts_lk_threshold) {
          if (is_corner(d, ev_time(e), e.x, e.y, e.polarity != 0)) {
            n_pts.push_back(P2f{(float)e.x, (float)e.y});
            if (idx_out) idx_out->push_back((int)i);
            ncorners++;
            circle_fill(event_mask_cur.data(), d->W, d->H, e.x, e.y, min_dist, 255);
          }
        }
      }
    }
  }
  return ncorners;
}

std::vector<P2f> undistorted_pts(const std::vector<P2f>& pts, const oracle_camera* cam) {  // :991-1002
  std::vector<P2f> un;
  for (unsigned int i = 0; i < pts.size(); i++) {
    double b[3];
    lift_projective(cam, (double)pts[i].x, (double)pts[i].y, b);
    un.push_back(P2f{(float)(b[0] / b[2]), (float)(b[1] / b[2])});
  }
  return un;
}

// FeatureTracker::ptsVelocity (feature_tracker.cpp:1004-1045); note the else-branch sizes the
// result by the LEFT cur_pts even when called for the right camera.
std::vector<P2f> pts_velocity_fn(Tracker* t, std::vector<int>& ids, std::vector<P2f>& pts,
                                 std::map<int, P2f>& cur_id_pts, std::map<int, P2f>& prev_id_pts) {
  std::vector<P2f> vel;
  cur_id_pts.clear();
  for (unsigned int i = 0; i < ids.size(); i++) cur_id_pts.insert(std::make_pair(ids[i], pts[i]));
  if (!prev_id_pts.empty()) {
    double dt = t->cur_time - t->prev_time;
    for (unsigned int i = 0; i < pts.size(); i++) {
      if (ids[i] != -1) {
        auto it = prev_id_pts.find(ids[i]);
        if (it != prev_id_pts.end()) {
          double v_x = (pts[i].x - it->second.x) / dt;
          double v_y = (pts[i].y - it->second.y) / dt;
          vel.push_back(P2f{(float)v_x, (float)v_y});
        } else
          vel.push_back(P2f{0, 0});
      } else
        vel.push_back(P2f{0, 0});
    }
  } else {
    for (unsigned int i = 0; i < t->cur_pts.size(); i++) vel.push_back(P2f{0, 0});
  }
  return vel;
}

// FeatureTracker::rejectWithF_event (feature_tracker.cpp:910-947)
void reject_with_f_event(Tracker* t) {
  if (t->cur_pts.size() >= 8) {
    const oracle_camera* cam = &t->cfg.cam[0];
    std::vector<float> un_cur(t->cur_pts.size() * 2), un_prev(t->prev_pts.size() * 2);
    const double FOCAL = t->cfg.focal_length;
    for (unsigned int i = 0; i < t->prev_pts.size(); i++) {
      double p[3];
      lift_projective(cam, t->prev_pts[i].x, t->prev_pts[i].y, p);
      p[0] = FOCAL * p[0] / p[2] + t->cfg.width / 2.0;
      p[1] = FOCAL * p[1] / p[2] + t->cfg.height / 2.0;
      un_prev[i * 2] = (float)p[0];
      un_prev[i * 2 + 1] = (float)p[1];
      lift_projective(cam, t->cur_pts[i].x, t->cur_pts[i].y, p);
      p[0] = FOCAL * p[0] / p[2] + t->cfg.width / 2.0;
      p[1] = FOCAL * p[1] / p[2] + t->cfg.height / 2.0;
      un_cur[i * 2] = (float)p[0];
      un_cur[i * 2 + 1] = (float)p[1];
    }
    std::vector<uint8_t> status(t->cur_pts.size());
    find_fundamental(un_prev.data(), un_cur.data(), (int)t->cur_pts.size(), t->cfg.f_threshold,
                     0.99, status.data(), nullptr);
    reduce_vector(t->prev_pts, status);
    reduce_vector(t->cur_pts, status);
    // (reference also reduces the stale cur_un_pts here, :940 — rebuilt at :471, no effect)
    reduce_vector(t->ids, status);
    reduce_vector(t->track_cnt, status);
  }
}

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) {
  return std::chrono::duration<double>(b - a).count();
}

// FeatureTracker::trackEvent (feature_tracker.cpp:340-603)
MotionComp make_motion(const oracle_motion* m, const oracle_event* left) {
  MotionComp mc;
  mc.t0 = ev_time(left[0]);
  mc.dt_batch = m->t1 - mc.t0;
  const double an = std::sqrt(std::pow((double)m->accel[0], 2) + std::pow((double)m->accel[1], 2) +
                              std::pow((double)m->accel[2], 2));
  mc.active = an > 5;
  for (int i = 0; i < 3; i++) {
    mc.v[i] = (float)m->v[i];
    mc.v_pre[i] = m->v_pre[i];
    mc.omega[i] = m->omega[i];
  }
  const Mat3f K = {{{(float)m->fx, 0.f, (float)m->cx}, {0.f, (float)m->fy, (float)m->cy}, {0.f, 0.f, 1.f}}};
  mc.K = K;
  mc.Kinv = m3_inverse(K);
  return mc;
}

// createSAE_left/right with Motion_correction_value (event_detector.cc:102-147,168-210) under the
// per-event gate of trackEvent (feature_tracker.cpp:627-641)
inline void create_sae_mc(Detector* d, int cam, const MotionComp& mc, const oracle_event& e) {
  const double et = ev_time(e);
  int ex = e.x, ey = e.y;
  if (mc.dt_batch > 0 && (et - mc.t0) / mc.dt_batch < 1) {
    if (mc.active) motion_correct(mc, d->W, d->H, ex, ey, et - mc.t0, &ex, &ey);
  }
  create_sae_one(d, cam, et, ex, ey, e.polarity != 0);
}

int track_event(Tracker* t, double _cur_time, const oracle_event* left, size_t nL,
                const oracle_event* right, size_t nR, bool PUB_THIS_FRAME,
                const oracle_motion* motion = nullptr) {
  const oracle_config& c = t->cfg;
  const int W = c.width, H = c.height, WIN = 21;
  t->cur_time = _cur_time;
  if (t->detector_nostart) {
    t->detector_nostart = false;
    t->det.reset();
  }
  auto t0 = clk::now();
  if (motion) {  // overload with Motion_correction_value (:605-641)
    const MotionComp mc = make_motion(motion, left);
    for (size_t i = 0; i < nL; i++)
      if (left[i].x < W && left[i].y < H) create_sae_mc(&t->det, 0, mc, left[i]);
    for (size_t i = 0; i < nR; i++)
      if (right[i].x < W && right[i].y < H) create_sae_mc(&t->det, 1, mc, right[i]);
  } else {
    for (size_t i = 0; i < nL; i++)  // :356-358
      if (left[i].x < W && left[i].y < H)
        create_sae_one(&t->det, 0, ev_time(left[i]), left[i].x, left[i].y, left[i].polarity != 0);
    for (size_t i = 0; i < nR; i++)  // :360-362
      if (right[i].x < W && right[i].y < H)
        create_sae_one(&t->det, 1, ev_time(right[i]), right[i].x, right[i].y, right[i].polarity != 0);
  }
  auto t1 = clk::now();
  t->ts_left.resize((size_t)W * H);
  t->ts_right.resize((size_t)W * H);
  sae_to_ts(&t->det, 0, t->cur_time, t->ts_left.data());   // :367
  sae_to_ts(&t->det, 1, t->cur_time, t->ts_right.data());  // :368
  auto t2 = clk::now();
  t->stage_s[0] += secs(t0, t1);
  t->stage_s[1] += secs(t1, t2);
  std::vector<uint8_t> eq_left, eq_right;
  if (c.equalize) {  // :375-382: CLAHE then normalize(0,255,MINMAX); detection keeps the raw surface
    eq_left.resize((size_t)W * H);
    eq_right.resize((size_t)W * H);
    clahe_apply(t->ts_left.data(), W, H, eq_left.data());
    clahe_apply(t->ts_right.data(), W, H, eq_right.data());
    normalize_minmax_u8(eq_left.data(), eq_left.size());
    normalize_minmax_u8(eq_right.data(), eq_right.size());
  }
  const std::vector<uint8_t>& img_left = c.equalize ? eq_left : t->ts_left;
  const std::vector<uint8_t>& img_right = c.equalize ? eq_right : t->ts_right;
  if (t->cur_img_left.empty()) {  // :390-395
    t->prev_img_left = t->cur_img_left = img_left;
  } else {
    t->cur_img_left = img_left;
  }
  t->cur_pts.clear();
  t->cur_img_right = img_right;  // :398-403
  t->cur_right_pts.clear();

  auto t3 = clk::now();
  if (t->prev_pts.size() > 0) {  // :405-437
    int n = (int)t->prev_pts.size();
    std::vector<uint8_t> status(n);
    t->cur_pts.resize(n);
    lk_pair_begin(n);
    calc_lk(t->prev_img_left.data(), t->cur_img_left.data(), W, H, &t->prev_pts[0].x,
            &t->cur_pts[0].x, status.data(), n, WIN, 3, 30, 0.01, 0, c.lk_accum);
    if (!c.flow_back) lk_pair_end();
    if (c.flow_back) {
      std::vector<uint8_t> reverse_status(n);
      std::vector<P2f> reverse_pts = t->prev_pts;
      calc_lk(t->cur_img_left.data(), t->prev_img_left.data(), W, H, &t->cur_pts[0].x,
              &reverse_pts[0].x, reverse_status.data(), n, WIN, 1, 30, 0.01, 4, c.lk_accum);
      lk_pair_end();
      for (size_t i = 0; i < status.size(); i++) {
        if (status[i] && reverse_status[i] && pt_distance(t->prev_pts[i], reverse_pts[i]) <= 0.5)
          status[i] = 1;
        else
          status[i] = 0;
      }
    }
    for (int i = 0; i < int(t->cur_pts.size()); i++)
      if (status[i] && !in_border_event(t, t->cur_pts[i])) status[i] = 0;
    reduce_vector(t->prev_pts, status);
    reduce_vector(t->cur_pts, status);
    reduce_vector(t->ids, status);
    reduce_vector(t->track_cnt, status);
  }
  auto t4 = clk::now();
  t->stage_s[2] += secs(t3, t4);

  for (auto& n : t->track_cnt) n++;  // :439-440

  if (PUB_THIS_FRAME) {  // :442-469
    auto h0 = clk::now();
    if (c.f_ransac) reject_with_f_event(t);
    event_set_mask(t);
    auto h1 = clk::now();
    t->stage_s[5] += secs(h0, h1);
    int n_max_cnt = c.max_cnt - static_cast<int>(t->cur_pts.size());
    if (n_max_cnt > 0) {
      features_to_track(&t->det, left, nL, c.max_cnt - (int)t->cur_pts.size(), c.min_dist,
                        t->mask_event.data(), t->ts_left.data(), c.ts_lk_threshold, t->n_pts,
                        nullptr);
    } else
      t->n_pts.clear();
    for (auto& p : t->n_pts) {
      t->cur_pts.push_back(p);
      t->ids.push_back(t->n_id++);
      t->track_cnt.push_back(1);
    }
    t->stage_s[3] += secs(h1, clk::now());
  }
  auto t5 = clk::now();
  t->cur_un_pts = undistorted_pts(t->cur_pts, &c.cam[0]);  // :470-473
  t->pts_velocity = pts_velocity_fn(t, t->ids, t->cur_un_pts, t->cur_un_pts_map, t->prev_un_pts_map);
  auto t6 = clk::now();
  t->stage_s[5] += secs(t5, t6);

  {  // :475-575  (img_right is never empty: the right time surface is always rendered)
    t->ids_right.clear();
    t->cur_right_pts.clear();
    t->cur_un_right_pts.clear();
    t->right_pts_velocity.clear();
    t->cur_un_right_pts_map.clear();
    t->track_cnt_right.clear();
    if (!t->cur_pts.empty()) {
      int n = (int)t->cur_pts.size();
      std::vector<P2f> reverseLeftPts(n);
      std::vector<uint8_t> status(n), statusRightLeft(n);
      t->cur_right_pts.resize(n);
      lk_pair_begin(n);
      calc_lk(t->cur_img_left.data(), t->cur_img_right.data(), W, H, &t->cur_pts[0].x,
              &t->cur_right_pts[0].x, status.data(), n, WIN, 3, 30, 0.01, 0, c.lk_accum);
      if (!(c.flow_back && !t->cur_right_pts.empty())) lk_pair_end();
      if (c.flow_back && !t->cur_right_pts.empty()) {
        calc_lk(t->cur_img_right.data(), t->cur_img_left.data(), W, H, &t->cur_right_pts[0].x,
                &reverseLeftPts[0].x, statusRightLeft.data(), n, WIN, 3, 30, 0.01, 0, c.lk_accum);
        lk_pair_end();
        for (size_t i = 0; i < status.size(); i++) {
          if (status[i] && statusRightLeft[i] && in_border_event(t, t->cur_right_pts[i]) &&
              pt_distance(t->cur_pts[i], reverseLeftPts[i]) <= 0.5)
            status[i] = 1;
          else
            status[i] = 0;
        }
      }
      t->ids_right = t->ids;
      reduce_vector(t->cur_right_pts, status);
      reduce_vector(t->ids_right, status);
      for (size_t i = 0; i < t->cur_right_pts.size(); i++) t->track_cnt_right.push_back(1);
      auto t7 = clk::now();
      t->stage_s[4] += secs(t6, t7);
      t->cur_un_right_pts = undistorted_pts(t->cur_right_pts, &c.cam[1]);
      t->right_pts_velocity = pts_velocity_fn(t, t->ids_right, t->cur_un_right_pts,
                                              t->cur_un_right_pts_map, t->prev_un_right_pts_map);
      t->stage_s[5] += secs(t7, clk::now());
    }
    t->prev_un_right_pts_map = t->cur_un_right_pts_map;
  }
  t->prev_img_left = t->cur_img_left;  // :585-590
  t->prev_pts = t->cur_pts;
  t->prev_un_pts_map = t->cur_un_pts_map;
  t->prev_time = t->cur_time;
  return 0;
}


// ============================================================================ image front-end
// FeatureTracker::trackImage (feature_tracker.cpp:164-338) — SURVEY 8(f) N4.
//
// cv::goodFeaturesToTrack(image, corners, maxCorners, qualityLevel, minDistance, mask) with the
// defaults trackImage uses (blockSize 3, gradientSize 3, Shi-Tomasi min-eigenvalue response).
// [OpenCV imgproc: featureselect.cpp goodFeaturesToTrack, corner.cpp cornerEigenValsVecs /
//  calcMinEigenVal, deriv.cpp Sobel, filter.cpp SymmRowSmallFilter / SymmColumnSmallFilter,
//  box_filter.cpp RowSum (ksize 3 branch) / ColumnSum — restated as recalled; OpenCV is not in the
//  reference tree and its version is unpinned, so this arithmetic is "parity unpinned".]
//   scale = 1 / (2^(3-1) * blockSize * 255);  smoothing taps {1,2,1} * (float)scale
//   Dx = Sobel(1,0): row [-1,0,1] (exact integers), column ((r0+r2)*f1 + r1*f0)
//   Dy = Sobel(0,1): row ((s0+s2)*k1 + s1*k0) in float, column (R2 - R0)
//   cov = (Dx*Dx, Dx*Dy, Dy*Dy);  3x3 unnormalised box: rows (c0+c1)+c2, columns as the running
//   sum SUM += next; D = SUM; SUM -= oldest (order matters in float);  BORDER_REFLECT_101 throughout
//   eig = (a+c) - sqrt((a-c)^2 + b^2), a = cov0*0.5f, b = cov1, c = cov2*0.5f
void corner_min_eigen_val(const uint8_t* img, int W, int H, std::vector<float>& eig) {
  double scale = (double)(1 << (3 - 1)) * 3;
  scale *= 255.0;
  scale = 1.0 / scale;
  const float fs = (float)scale;
  const float k0 = 2.f * fs, k1 = 1.f * fs;  // centre / neighbour smoothing taps
  auto px = [&](int y, int x) { return (int)img[(size_t)reflect101(y, H) * W + reflect101(x, W)]; };
  std::vector<float> cov((size_t)W * H * 3);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      // Dx: differentiate along x (exact), smooth along y
      const int r0 = px(y - 1, x + 1) - px(y - 1, x - 1);
      const int r1 = px(y, x + 1) - px(y, x - 1);
      const int r2 = px(y + 1, x + 1) - px(y + 1, x - 1);
      const float dx = (float)(r0 + r2) * k1 + (float)r1 * k0;
      // Dy: smooth along x in float, differentiate along y
      const float R0 = (float)(px(y - 1, x - 1) + px(y - 1, x + 1)) * k1 + (float)px(y - 1, x) * k0;
      const float R2 = (float)(px(y + 1, x - 1) + px(y + 1, x + 1)) * k1 + (float)px(y + 1, x) * k0;
      const float dy = R2 - R0;
      float* c = &cov[((size_t)y * W + x) * 3];
      c[0] = dx * dx;
      c[1] = dx * dy;
      c[2] = dy * dy;
    }
  // RowSum, ksize == 3
  std::vector<float> rs((size_t)W * H * 3);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      for (int k = 0; k < 3; k++) {
        const float* r = &cov[(size_t)y * W * 3];
        rs[((size_t)y * W + x) * 3 + k] =
            (r[reflect101(x - 1, W) * 3 + k] + r[x * 3 + k]) + r[reflect101(x + 1, W) * 3 + k];
      }
  // ColumnSum (running sum over the rows -1 .. H), then calcMinEigenVal
  eig.assign((size_t)W * H, 0.f);
  std::vector<float> SUM((size_t)W * 3, 0.f);
  auto row = [&](int y) { return &rs[(size_t)reflect101(y, H) * W * 3]; };
  for (int y = -1; y <= 0; y++) {
    const float* Sp = row(y);
    for (int i = 0; i < W * 3; i++) SUM[i] += Sp[i];
  }
  for (int y = 0; y < H; y++) {
    const float* Sp = row(y + 1);
    const float* Sm = row(y - 1);
    for (int x = 0; x < W; x++) {
      float cv3[3];
      for (int k = 0; k < 3; k++) {
        const int i = x * 3 + k;
        const float s0 = SUM[i] + Sp[i];
        cv3[k] = s0;
        SUM[i] = s0 - Sm[i];
      }
      const float a = cv3[0] * 0.5f, b = cv3[1], c = cv3[2] * 0.5f;
      eig[(size_t)y * W + x] = (a + c) - std::sqrt((a - c) * (a - c) + b * b);
    }
  }
}

// half-widths of the open Euclidean disc dx*dx + dy*dy < r*r by |dy| (-1: row not touched)
void euclid_halfwidths(int r, int* hw /*[r+1]*/) {
  for (int dy = 0; dy <= r; dy++) {
    int w = -1;
    for (int dx = 0; dx <= r; dx++)
      if (dx * dx + dy * dy < r * r) w = dx;
    hw[dy] = w;
  }
}

int good_features_to_track(const uint8_t* img, int W, int H, int maxCorners, double qualityLevel,
                           double minDistance, const uint8_t* mask /*nonzero = allowed, or NULL*/,
                           std::vector<P2f>& corners, std::vector<float>* eig_out) {
  corners.clear();
  std::vector<float> eig;
  corner_min_eigen_val(img, W, H, eig);
  if (eig_out) *eig_out = eig;
  // minMaxLoc(eig, 0, &maxVal, 0, 0, mask); threshold(eig, eig, maxVal*qualityLevel, 0, THRESH_TOZERO)
  double maxVal = 0;
  bool any = false;
  for (size_t i = 0; i < eig.size(); i++)
    if (!mask || mask[i]) {
      if (!any || (double)eig[i] > maxVal) maxVal = eig[i];
      any = true;
    }
  if (!any) return 0;
  const float thr = (float)(maxVal * qualityLevel);
  std::vector<float> th(eig.size());
  for (size_t i = 0; i < eig.size(); i++) th[i] = eig[i] > thr ? eig[i] : 0.f;
  // dilate 3x3 + local-maximum test, rows/cols 1 .. size-2
  std::vector<int> cand;  // pixel offsets
  for (int y = 1; y < H - 1; y++)
    for (int x = 1; x < W - 1; x++) {
      const float val = th[(size_t)y * W + x];
      if (val == 0) continue;
      float tmp = val;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) tmp = std::max(tmp, th[(size_t)(y + dy) * W + (x + dx)]);
      if (val == tmp && (!mask || mask[(size_t)y * W + x])) cand.push_back(y * W + x);
    }
  // std::sort(tmpCorners, greaterThanPtr): by value, ties by address, both descending
  std::sort(cand.begin(), cand.end(), [&](int a, int b) {
    return th[a] > th[b] ? true : th[a] < th[b] ? false : a > b;
  });
  if (minDistance >= 1) {
    // the cell grid only accelerates the search; the test is the plain Euclidean one
    const double md2 = minDistance * minDistance;
    for (int o : cand) {
      const int y = o / W, x = o - y * W;
      bool good = true;
      for (const P2f& q : corners) {
        const float dx = (float)x - q.x, dy = (float)y - q.y;
        if (dx * dx + dy * dy < md2) {
          good = false;
          break;
        }
      }
      if (good) {
        corners.push_back(P2f{(float)x, (float)y});
        if (maxCorners > 0 && (int)corners.size() == maxCorners) break;
      }
    }
  } else {
    for (int o : cand) {
      const int y = o / W, x = o - y * W;
      corners.push_back(P2f{(float)x, (float)y});
      if (maxCorners > 0 && (int)corners.size() == maxCorners) break;
    }
  }
  return (int)corners.size();
}

// FeatureTracker::Image_setMask (feature_tracker.cpp:90-119; FISHEYE = 0): mask_image 255 = free
void image_set_mask(Tracker* t, std::vector<uint8_t>& mask_image) {
  const int W = t->cfg.width, H = t->cfg.height;
  mask_image.assign((size_t)W * H, 255);
  std::vector<std::pair<int, std::pair<P2f, int>>> cnt_pts_id;
  for (unsigned int i = 0; i < t->cur_pts.size(); i++)
    cnt_pts_id.push_back(std::make_pair(t->track_cnt[i], std::make_pair(t->cur_pts[i], t->ids[i])));
  std::sort(cnt_pts_id.begin(), cnt_pts_id.end(),
            [](const std::pair<int, std::pair<P2f, int>>& a,
               const std::pair<int, std::pair<P2f, int>>& b) { return a.first > b.first; });
  t->cur_pts.clear();
  t->ids.clear();
  t->track_cnt.clear();
  for (auto& it : cnt_pts_id) {
    // Mat::at<uchar>(Point2f) -> Point(cvRound(x), cvRound(y)); inside the image after inBorder
    const int px = cv_round_f(it.second.first.x), py = cv_round_f(it.second.first.y);
    if (px < 0 || px >= W || py < 0 || py >= H) continue;
    if (mask_image[(size_t)py * W + px] == 255) {
      t->cur_pts.push_back(it.second.first);
      t->ids.push_back(it.second.second);
      t->track_cnt.push_back(it.first);
      circle_fill(mask_image.data(), W, H, px, py, t->cfg.min_dist, 0);
    }
  }
}

// trackImage: cfg.width/height = COL/ROW, max_cnt = MAX_CNT_IMG, min_dist = MIN_DIST_IMG;
// cfg.equalize applies the node's CLAHE (stereo_image_tracker_node.cpp:92-96) to both images first.
// inBorder (feature_tracker.cpp:40-46) == in_border_event with COL/ROW.  The reference's
// track_cnt_right bookkeeping (:303-311) is not an output and indexes out of range; not restated.
int track_image(Tracker* t, double _cur_time, const uint8_t* img_left_in, const uint8_t* img_right_in,
                bool PUB_THIS_FRAME) {
  const oracle_config& c = t->cfg;
  const int W = c.width, H = c.height, WIN = 21;
  t->cur_time = _cur_time;
  std::vector<uint8_t> img_left(img_left_in, img_left_in + (size_t)W * H), img_right;
  if (img_right_in) img_right.assign(img_right_in, img_right_in + (size_t)W * H);
  if (c.equalize) {
    std::vector<uint8_t> tmp(img_left.size());
    clahe_apply(img_left.data(), W, H, tmp.data());
    img_left.swap(tmp);
    if (!img_right.empty()) {
      clahe_apply(img_right.data(), W, H, tmp.data());
      img_right.swap(tmp);
    }
  }
  if (t->cur_img_left.empty())
    t->prev_img_left = t->cur_img_left = img_left;
  else
    t->cur_img_left = img_left;
  t->cur_pts.clear();
  if (t->prev_pts.size() > 0) {  // :180-209
    const int n = (int)t->prev_pts.size();
    std::vector<uint8_t> status(n);
    t->cur_pts.resize(n);
    calc_lk(t->prev_img_left.data(), t->cur_img_left.data(), W, H, &t->prev_pts[0].x,
            &t->cur_pts[0].x, status.data(), n, WIN, 3, 30, 0.01, 0, c.lk_accum);
    if (c.flow_back) {
      std::vector<uint8_t> reverse_status(n);
      std::vector<P2f> reverse_pts = t->prev_pts;  // (initial values unused: flags 0)
      calc_lk(t->cur_img_left.data(), t->prev_img_left.data(), W, H, &t->cur_pts[0].x,
              &reverse_pts[0].x, reverse_status.data(), n, WIN, 3, 30, 0.01, 0, c.lk_accum);
      for (size_t i = 0; i < status.size(); i++)
        status[i] = status[i] && reverse_status[i] && pt_distance(t->prev_pts[i], reverse_pts[i]) <= 0.5;
    }
    for (int i = 0; i < int(t->cur_pts.size()); i++)
      if (status[i] && !in_border_event(t, t->cur_pts[i])) status[i] = 0;
    reduce_vector(t->prev_pts, status);
    reduce_vector(t->cur_pts, status);
    reduce_vector(t->ids, status);
    reduce_vector(t->track_cnt, status);
  }
  for (auto& n : t->track_cnt) n++;
  if (PUB_THIS_FRAME) {  // :214-241
    std::vector<uint8_t> mask_image;
    image_set_mask(t, mask_image);
    const int n_max_cnt = c.max_cnt - static_cast<int>(t->cur_pts.size());
    if (n_max_cnt > 0)
      good_features_to_track(t->cur_img_left.data(), W, H, n_max_cnt, 0.01, c.min_dist,
                             mask_image.data(), t->n_pts, nullptr);
    else
      t->n_pts.clear();
    for (auto& p : t->n_pts) {
      t->cur_pts.push_back(p);
      t->ids.push_back(t->n_id++);
      t->track_cnt.push_back(1);
    }
  }
  t->cur_un_pts = undistorted_pts(t->cur_pts, &c.cam[0]);
  t->pts_velocity = pts_velocity_fn(t, t->ids, t->cur_un_pts, t->cur_un_pts_map, t->prev_un_pts_map);
  if (!img_right.empty()) {  // :249-318
    t->ids_right.clear();
    t->cur_right_pts.clear();
    t->cur_un_right_pts.clear();
    t->right_pts_velocity.clear();
    t->cur_un_right_pts_map.clear();
    if (!t->cur_pts.empty()) {
      const int n = (int)t->cur_pts.size();
      std::vector<P2f> reverseLeftPts(n);
      std::vector<uint8_t> status(n), statusRightLeft(n);
      t->cur_right_pts.resize(n);
      calc_lk(t->cur_img_left.data(), img_right.data(), W, H, &t->cur_pts[0].x, &t->cur_right_pts[0].x,
              status.data(), n, WIN, 3, 30, 0.01, 0, c.lk_accum);
      if (c.flow_back && !t->cur_right_pts.empty()) {
        calc_lk(img_right.data(), t->cur_img_left.data(), W, H, &t->cur_right_pts[0].x,
                &reverseLeftPts[0].x, statusRightLeft.data(), n, WIN, 3, 30, 0.01, 0, c.lk_accum);
        for (size_t i = 0; i < status.size(); i++)
          status[i] = status[i] && statusRightLeft[i] && in_border_event(t, t->cur_right_pts[i]) &&
                      pt_distance(t->cur_pts[i], reverseLeftPts[i]) <= 0.5;
      }
      t->ids_right = t->ids;
      reduce_vector(t->cur_right_pts, status);
      reduce_vector(t->ids_right, status);
      t->cur_un_right_pts = undistorted_pts(t->cur_right_pts, &c.cam[1]);
      t->right_pts_velocity = pts_velocity_fn(t, t->ids_right, t->cur_un_right_pts,
                                              t->cur_un_right_pts_map, t->prev_un_right_pts_map);
    }
    t->prev_un_right_pts_map = t->cur_un_right_pts_map;
  }
  t->prev_img_left = t->cur_img_left;
  t->prev_pts = t->cur_pts;
  t->prev_un_pts_map = t->cur_un_pts_map;
  t->prev_time = t->cur_time;
  return 0;
}

}  // namespace

// ================================================================== C interface
extern "C" {

void* oracle_detector_create(int W, int H, double decay_ms, int ignore_polarity,
                             double filter_threshold, int min_dist) {
  Detector* d = new Detector();
  d->W = W;
  d->H = H;
  d->decay_ms = decay_ms;
  d->ignore_polarity = ignore_polarity != 0;
  d->filter_threshold = filter_threshold;
  d->min_dist = min_dist;
  d->reset();
  return d;
}
void oracle_detector_destroy(void* d) { delete (Detector*)d; }
void oracle_detector_reset(void* d) { ((Detector*)d)->reset(); }
void oracle_detector_set_median(void* d, int k) { ((Detector*)d)->median_blur_kernel_size = k; }
void oracle_median_blur(uint8_t* img, int w, int h, int ksize) { median_blur_u8(img, w, h, ksize); }

size_t oracle_create_sae(void* dv, int cam, const oracle_event* ev, size_t n) {
  Detector* d = (Detector*)dv;
  size_t rejected = 0;
  for (size_t i = 0; i < n; i++) {
    if (ev[i].x >= d->W || ev[i].y >= d->H) {
      rejected++;
      continue;
    }
    create_sae_one(d, cam, ev_time(ev[i]), ev[i].x, ev[i].y, ev[i].polarity != 0);
  }
  return rejected;
}
void oracle_sae_to_time_surface(void* d, int cam, double t_sync, uint8_t* out) {
  sae_to_ts((Detector*)d, cam, t_sync, out);
}
int oracle_is_corner(void* d, double et, int ex, int ey, int ep) {
  return is_corner((Detector*)d, et, ex, ey, ep != 0) ? 1 : 0;
}
void oracle_corner_flags(void* dv, const oracle_event* ev, size_t n, uint8_t* flags) {
  Detector* d = (Detector*)dv;
  for (size_t i = 0; i < n; i++) {
    if (ev[i].x >= d->W || ev[i].y >= d->H) {
      flags[i] = 0;
      continue;
    }
    flags[i] = is_corner(d, ev_time(ev[i]), ev[i].x, ev[i].y, ev[i].polarity != 0) ? 1 : 0;
  }
}
void oracle_get_sae(void* dv, int cam, double* L0, double* L1, double* S0, double* S1) {
  Detector* d = (Detector*)dv;
  size_t n = (size_t)d->W * d->H * sizeof(double);
  std::memcpy(L0, d->sae_latest[cam][0].data(), n);
  std::memcpy(L1, d->sae_latest[cam][1].data(), n);
  std::memcpy(S0, d->sae[cam][0].data(), n);
  std::memcpy(S1, d->sae[cam][1].data(), n);
}
void oracle_set_sae(void* dv, int cam, const double* L0, const double* L1, const double* S0,
                    const double* S1) {
  Detector* d = (Detector*)dv;
  size_t n = (size_t)d->W * d->H;
  d->sae_latest[cam][0].assign(L0, L0 + n);
  d->sae_latest[cam][1].assign(L1, L1 + n);
  d->sae[cam][0].assign(S0, S0 + n);
  d->sae[cam][1].assign(S1, S1 + n);
}

void oracle_disc_halfwidths(int r, int* hw) { disc_halfwidths(r, hw); }
void oracle_circle_fill(uint8_t* img, int W, int H, int cx, int cy, int r, uint8_t v) {
  circle_fill(img, W, H, cx, cy, r, v);
}
int oracle_features_to_track(void* dv, const oracle_event* ev, size_t n, int max_corners,
                             int min_dist, const uint8_t* mask, const uint8_t* ts,
                             double ts_lk_threshold, float* out_xy, int32_t* out_idx) {
  std::vector<P2f> pts;
  std::vector<int> idx;
  int nc = features_to_track((Detector*)dv, ev, n, max_corners, min_dist, mask, ts,
                             ts_lk_threshold, pts, &idx);
  for (int i = 0; i < nc; i++) {
    out_xy[i * 2] = pts[i].x;
    out_xy[i * 2 + 1] = pts[i].y;
    if (out_idx) out_idx[i] = idx[i];
  }
  return nc;
}

void oracle_pyr_down(const uint8_t* src, int sw, int sh, uint8_t* dst) { pyr_down(src, sw, sh, dst); }
void oracle_scharr(const uint8_t* src, int w, int h, int16_t* dst) { scharr(src, w, h, dst); }
int oracle_pyr_levels(int w, int h, int win, int max_level) { return pyr_levels(w, h, win, max_level); }
void oracle_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts,
               float* next_pts, uint8_t* status, int n, int win, int max_level, int max_count,
               double eps, int flags, int accum) {
  calc_lk(prev, next, w, h, prev_pts, next_pts, status, n, win, max_level, max_count, eps, flags,
          accum);
}
void oracle_clahe(const uint8_t* src, int w, int h, uint8_t* dst) { clahe_apply(src, w, h, dst); }
void oracle_normalize_minmax(uint8_t* img, size_t n) { normalize_minmax_u8(img, n); }
void oracle_lift_projective(const oracle_camera* cam, double u, double v, double* out3) {
  lift_projective(cam, u, v, out3);
}
int oracle_find_fundamental_ransac(const float* p1, const float* p2, int n, double thr, double conf,
                                   uint8_t* status, double* F9) {
  return find_fundamental(p1, p2, n, thr, conf, status, F9);
}
void oracle_set_nullspace_mode(int mode) { g_nullspace_mode = mode; }
int oracle_svd_rows(double* at, int m, int n, int n1, double* w) {
  if (m < n || n1 < n || n1 > m || n > 16) return -1;
  jacobi_svd_rows(at, m, w, m, n, n1);
  return 0;
}
int oracle_seven_point(const float* p1, const float* p2, double* F27) { return run_7point(p1, p2, F27); }

void* oracle_tracker_create(const oracle_config* cfg) {
  if (cfg->median_blur_kernel_size < 0) return nullptr;
  Tracker* t = new Tracker();
  t->cfg = *cfg;
  t->det.W = cfg->width;
  t->det.H = cfg->height;
  t->det.decay_ms = cfg->decay_ms;
  t->det.ignore_polarity = cfg->ignore_polarity != 0;
  t->det.filter_threshold = cfg->feature_filter_threshold;
  t->det.min_dist = cfg->min_dist;
  t->det.median_blur_kernel_size = cfg->median_blur_kernel_size;
  t->det.reset();
  return t;
}
void oracle_tracker_destroy(void* t) { delete (Tracker*)t; }

static void fill_tracks(Tracker* t, oracle_tracks* out) {
  if (out) {
    out->n_left = (int32_t)t->ids.size();
    for (size_t i = 0; i < t->ids.size(); i++) {
      out->ids[i] = t->ids[i];
      out->track_cnt[i] = t->track_cnt[i];
      out->cur_pts[2 * i] = t->cur_pts[i].x;
      out->cur_pts[2 * i + 1] = t->cur_pts[i].y;
      out->cur_un_pts[2 * i] = t->cur_un_pts[i].x;
      out->cur_un_pts[2 * i + 1] = t->cur_un_pts[i].y;
      out->pts_velocity[2 * i] = t->pts_velocity[i].x;
      out->pts_velocity[2 * i + 1] = t->pts_velocity[i].y;
    }
    out->n_right = (int32_t)t->ids_right.size();
    for (size_t i = 0; i < t->ids_right.size(); i++) {
      out->ids_right[i] = t->ids_right[i];
      out->cur_right_pts[2 * i] = t->cur_right_pts[i].x;
      out->cur_right_pts[2 * i + 1] = t->cur_right_pts[i].y;
      out->cur_un_right_pts[2 * i] = t->cur_un_right_pts[i].x;
      out->cur_un_right_pts[2 * i + 1] = t->cur_un_right_pts[i].y;
      out->right_pts_velocity[2 * i] = t->right_pts_velocity[i].x;
      out->right_pts_velocity[2 * i + 1] = t->right_pts_velocity[i].y;
    }
  }
}

int oracle_track_event(void* tv, double cur_time, const oracle_event* left, size_t nL,
                       const oracle_event* right, size_t nR, int pub_this_frame,
                       oracle_tracks* out) {
  Tracker* t = (Tracker*)tv;
  if (nL == 0) return -1;
  int rc = track_event(t, cur_time, left, nL, right, nR, pub_this_frame != 0);
  if (rc) return rc;
  fill_tracks(t, out);
  return 0;
}
int oracle_track_event_mc(void* tv, double cur_time, const oracle_event* left, size_t nL,
                          const oracle_event* right, size_t nR, int pub_this_frame,
                          const oracle_motion* motion, oracle_tracks* out) {
  Tracker* t = (Tracker*)tv;
  if (nL == 0 || !motion) return -1;
  int rc = track_event(t, cur_time, left, nL, right, nR, pub_this_frame != 0, motion);
  if (rc) return rc;
  fill_tracks(t, out);
  return 0;
}
size_t oracle_create_sae_mc(void* dv, int cam, const oracle_event* ev, size_t n,
                            const oracle_event* first_left, const oracle_motion* motion) {
  Detector* d = (Detector*)dv;
  const MotionComp mc = make_motion(motion, first_left);
  size_t rejected = 0;
  for (size_t i = 0; i < n; i++) {
    if (ev[i].x >= d->W || ev[i].y >= d->H) {
      rejected++;
      continue;
    }
    create_sae_mc(d, cam, mc, ev[i]);
  }
  return rejected;
}
void oracle_matrix_exp3f(const float* a9, float* out9) {
  Mat3f A;
  std::memcpy(A.m, a9, sizeof(A.m));
  Mat3f R = m3_exp(A);
  std::memcpy(out9, R.m, sizeof(R.m));
}
void oracle_tracker_time_surface(void* tv, int cam, uint8_t* out) {
  Tracker* t = (Tracker*)tv;
  const std::vector<uint8_t>& s = cam ? t->ts_right : t->ts_left;
  std::memcpy(out, s.data(), s.size());
}
void* oracle_tracker_detector(void* tv) { return &((Tracker*)tv)->det; }
int oracle_good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double quality,
                                  double min_distance, const uint8_t* mask, float* out_xy,
                                  int32_t* n_out, float* eig_out) {
  std::vector<P2f> corners;
  std::vector<float> eig;
  const int n = good_features_to_track(img, w, h, max_corners, quality, min_distance, mask, corners,
                                       eig_out ? &eig : nullptr);
  if (out_xy) std::memcpy(out_xy, corners.data(), corners.size() * sizeof(P2f));
  if (eig_out) std::memcpy(eig_out, eig.data(), eig.size() * sizeof(float));
  if (n_out) *n_out = n;
  return n;
}
void oracle_euclid_halfwidths(int r, int* hw) { euclid_halfwidths(r, hw); }
int oracle_track_image(void* tv, double cur_time, const uint8_t* left, const uint8_t* right,
                       int pub_this_frame, oracle_tracks* out) {
  Tracker* t = (Tracker*)tv;
  if (!left) return -1;
  const int rc = track_image(t, cur_time, left, right, pub_this_frame != 0);
  if (rc == 0 && out) fill_tracks(t, out);
  return rc;
}
void oracle_lk_iter_stats(unsigned long long* out3, int reset) {
  out3[0] = g_lk_iters;
  out3[1] = g_lk_visits;
  out3[2] = g_lk_maxed;
  if (reset) g_lk_iters = g_lk_visits = g_lk_maxed = 0;
}
void oracle_lk_pair_stats(unsigned long long* out5, int reset) {
  out5[0] = g_lk_pairs;          // forward+backward pairs (= fused launches)
  out5[1] = g_lk_pair_max_sum;   // sum over pairs of the slowest point's iterations
  out5[2] = g_lk_pair_pts;       // points over all pairs
  out5[3] = g_lk_pair_iter_sum;  // iterations over all points of all pairs
  out5[4] = g_lk_pair_max_max;   // slowest point of any pair
  if (reset) g_lk_pairs = g_lk_pair_max_sum = g_lk_pair_pts = g_lk_pair_iter_sum = g_lk_pair_max_max = 0;
}
void oracle_tracker_stage_seconds(void* tv, double* out6) {
  std::memcpy(out6, ((Tracker*)tv)->stage_s, sizeof(double) * 6);
}

}  // extern "C"
