// fe_host.cpp — host-side geometry of the event front-end (see fe_host.h).
// Compiled with -ffp-contract=off like the kernels (the reference build has no FMA contraction).
#include "fe_host.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

// The per-point loops below are written over plain arrays so that the compiler vectorises them;
// an AVX2 clone is selected at load time where the CPU has it.  Every element still goes through
// the same IEEE operations in the same order (no contraction, no reassociation), so results do not
// depend on the vector width.
#if defined(__HIP_DEVICE_COMPILE__)
#define ESVIO_SIMD_CLONES
#else
#define ESVIO_SIMD_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif

namespace esvio {
namespace host {

int cv_round(double v) {
  // SSE2 cvtsd2si semantics: ties-to-even, 0x80000000 when the value does not fit
  if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
  const double r = std::nearbyint(v);
  if (r > 2147483647.0 || r < -2147483648.0) return INT_MIN;
  return (int)r;
}

int cv_floor(float v) {
  int i = (int)v;
  return i - (i > v);
}

// ---------------------------------------------------------------------------- pinhole
// PinholeCamera::distortion (PinholeCamera.cc:646-662): radial-tangential displacement d_u(p_u)
static void radtan(const esvio_fe_camera& c, double x, double y, double& dx, double& dy) {
  const double x2 = x * x, y2 = y * y, xy = x * y;
  const double rho2 = x2 + y2;
  const double rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
  dx = x * rad + 2.0 * c.p1 * xy + c.p2 * (rho2 + 2.0 * x2);
  dy = y * rad + 2.0 * c.p2 * xy + c.p1 * (rho2 + 2.0 * y2);
}

void lift_projective(const esvio_fe_camera& c, double u, double v, double out[3]) {
  // m_inv_K11 = 1/fx, m_inv_K13 = -cx/fx, ... (PinholeCamera.cc:824-827)
  const double xd = (1.0 / c.fx) * u + (-c.cx / c.fx);
  const double yd = (1.0 / c.fy) * v + (-c.cy / c.fy);
  double xu = xd, yu = yd;
  const bool no_distortion = c.k1 == 0.0 && c.k2 == 0.0 && c.p1 == 0.0 && c.p2 == 0.0;
  if (!no_distortion) {
    // "recursive distortion model", n = 8 fixed-point iterations (:490-504)
    double dx, dy;
    radtan(c, xd, yd, dx, dy);
    xu = xd - dx;
    yu = yd - dy;
    for (int it = 1; it < 8; ++it) {
      radtan(c, xu, yu, dx, dy);
      xu = xd - dx;
      yu = yd - dy;
    }
  }
  out[0] = xu;
  out[1] = yu;
  out[2] = 1.0;
}

// liftProjective of n points (uv interleaved float pairs) -> xu[], yu[] (z = 1): the 8 fixed-point
// iterations run over all points together
ESVIO_SIMD_CLONES
void lift_projective_batch(const esvio_fe_camera& c, const float* uv, int n, double* __restrict xu,
                           double* __restrict yu) {
  const double ifx = 1.0 / c.fx, ify = 1.0 / c.fy, ox = -c.cx / c.fx, oy = -c.cy / c.fy;
  const double k1 = c.k1, k2 = c.k2, p1 = c.p1, p2 = c.p2;
  const bool no_distortion = k1 == 0.0 && k2 == 0.0 && p1 == 0.0 && p2 == 0.0;
  constexpr int B = 64;
  double xd[B], yd[B], x[B], y[B];
  for (int i0 = 0; i0 < n; i0 += B) {
    const int m = std::min(B, n - i0);
    for (int i = 0; i < m; i++) {
      xd[i] = ifx * (double)uv[2 * (i0 + i)] + ox;
      yd[i] = ify * (double)uv[2 * (i0 + i) + 1] + oy;
      x[i] = xd[i];
      y[i] = yd[i];
    }
    if (!no_distortion) {
      for (int it = 0; it < 8; ++it) {
        for (int i = 0; i < m; i++) {
          const double x2 = x[i] * x[i], y2 = y[i] * y[i], xy = x[i] * y[i];
          const double rho2 = x2 + y2;
          const double rad = k1 * rho2 + k2 * rho2 * rho2;
          const double dx = x[i] * rad + 2.0 * p1 * xy + p2 * (rho2 + 2.0 * x2);
          const double dy = y[i] * rad + 2.0 * p2 * xy + p1 * (rho2 + 2.0 * y2);
          x[i] = xd[i] - dx;
          y[i] = yd[i] - dy;
        }
      }
    }
    for (int i = 0; i < m; i++) {
      xu[i0 + i] = x[i];
      yu[i0 + i] = y[i];
    }
  }
}

// ---------------------------------------------------------------------------- disc / bitmap
std::vector<int> disc_halfwidths(int r) {
  std::vector<int> hw(r + 1, -1);
  int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
  while (dx >= dy) {
    hw[dy] = std::max(hw[dy], dx);
    hw[dx] = std::max(hw[dx], dy);
    dy++;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
  return hw;
}

void BitMask::reset(int w, int h) {
  W = w;
  H = h;
  wpr = (w + 31) / 32;
  bits.assign((size_t)wpr * h, 0u);
}

void BitMask::stamp_disc(int cx, int cy, int r, const std::vector<int>& hw) {
  for (int oy = -r; oy <= r; oy++) {
    const int y = cy + oy;
    if (y < 0 || y >= H) continue;
    const int h = hw[oy < 0 ? -oy : oy];
    if (h < 0) continue;
    const int x0 = std::max(cx - h, 0), x1 = std::min(cx + h, W - 1);
    if (x1 < x0) continue;
    for (int w = x0 >> 5; w <= (x1 >> 5); w++) {
      const int lo = std::max(x0 - (w << 5), 0), hi = std::min(x1 - (w << 5), 31);
      const uint32_t m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
      bits[(size_t)y * wpr + w] |= m;
    }
  }
}

void BitMask::from_bytes(const uint8_t* mask) {
  std::fill(bits.begin(), bits.end(), 0u);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      if (mask[(size_t)y * W + x] == 255) bits[(size_t)y * wpr + (x >> 5)] |= 1u << (x & 31);
}

// ---------------------------------------------------------------------------- F-matrix RANSAC
namespace {

struct OcvRng {  // cv::RNG multiply-with-carry generator, seeded with (uint64)-1 by the registrators
  uint64_t s = ~0ull;
  unsigned next() {
    s = (uint64_t)(unsigned)s * 4164903690U + (unsigned)(s >> 32);
    return (unsigned)s;
  }
  int uniform(int lo, int hi) { return lo == hi ? lo : (int)(next() % (unsigned)(hi - lo) + lo); }
};

// Two orthonormal vectors spanning the null space of the 7x9 epipolar system, obtained the way
// run7Point obtains them: SVDecomp(A, W, U, Vt, MODIFY_A + FULL_UV) and rows 7, 8 of Vt.  For a
// matrix with fewer rows than columns cv::SVD runs its one-sided Jacobi sweep on the 7 rows of A
// themselves (length 9): cyclic Hestenes rotations (pairs (0,1),(0,2)..(5,6), at most 30 sweeps, a
// pair skipped when |<ri,rj>| <= 10 eps sqrt(|ri|^2 |rj|^2), squared norms carried along), rows
// sorted by norm (selection sort, largest first) and normalised.  Rows 7 and 8 are not determined by
// the data; OpenCV fills each with +-1/9 from cv::RNG(0x12345678) (bit 8 of the generator's 32-bit
// outputs), projects the rows above out of it twice (rescaling by the L1 norm after every
// projection) and normalises — restated here step for step [OpenCV 4.2 core/src/lapack.cpp,
// JacobiSVDImpl_<double>: minval = DBL_MIN, eps = 10 DBL_EPSILON; not in /root/reference].  Every
// sum runs in index order in double, mul and add stay separate (-ffp-contract=off).
//
// Which basis of the null plane is used matters only through rounding (the three F solutions are
// those of the same cubic), but that is enough to move a point that lies within float rounding of
// the 1 px threshold across it, so the basis is OpenCV's and not a cheaper one.
ESVIO_SIMD_CLONES
void epipolar_nullspace(const double A[7][9], double f1[9], double f2[9]) {
  constexpr int M = 9, N = 7, N1 = 9, MP = 12;
  constexpr double kMin = DBL_MIN, kEps = DBL_EPSILON * 10;
  alignas(32) double R[N1][MP];
  double W[N];
  for (int i = 0; i < N1; i++)
    for (int k = 0; k < MP; k++) R[i][k] = (i < N && k < M) ? A[i][k] : 0.0;
  for (int i = 0; i < N; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += R[i][k] * R[i][k];
    W[i] = sd;
  }
  for (int sweep = 0; sweep < 30; sweep++) {  // max(m, 30)
    bool changed = false;
    for (int i = 0; i < N - 1; i++)
      for (int j = i + 1; j < N; j++) {
        double* __restrict ri = R[i];
        double* __restrict rj = R[j];
        double a = W[i], b = W[j], p = 0;
        for (int k = 0; k < M; k++) p += ri[k] * rj[k];
        if (std::fabs(p) <= kEps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = hypot(p, beta);
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = std::sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        alignas(32) double t0[MP], t1[MP];
        for (int k = 0; k < MP; k++) {  // (the padding lanes stay 0)
          t0[k] = c * ri[k] + s * rj[k];
          t1[k] = -s * ri[k] + c * rj[k];
        }
        a = b = 0;
        for (int k = 0; k < M; k++) {
          a += t0[k] * t0[k];
          b += t1[k] * t1[k];
        }
        for (int k = 0; k < MP; k++) {
          ri[k] = t0[k];
          rj[k] = t1[k];
        }
        W[i] = a;
        W[j] = b;
        changed = true;
      }
    if (!changed) break;
  }
  for (int i = 0; i < N; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += R[i][k] * R[i][k];
    W[i] = std::sqrt(sd);
  }
  for (int i = 0; i < N - 1; i++) {
    int j = i;
    for (int k = i + 1; k < N; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < MP; k++) std::swap(R[i][k], R[j][k]);
    }
  }
  uint64_t rng = 0x12345678;  // cv::RNG(0x12345678)
  for (int i = 0; i < N1; i++) {
    double sd = i < N ? W[i] : 0;
    for (int attempt = 0; attempt < 100 && sd <= kMin; attempt++) {
      const double v0 = 1. / M;
      for (int k = 0; k < M; k++) {
        rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
        R[i][k] = ((unsigned)rng & 256) != 0 ? v0 : -v0;
      }
      for (int round = 0; round < 2; round++)
        for (int j = 0; j < i; j++) {
          sd = 0;
          for (int k = 0; k < M; k++) sd += R[i][k] * R[j][k];
          double asum = 0;
          for (int k = 0; k < M; k++) {
            const double t = R[i][k] - sd * R[j][k];
            R[i][k] = t;
            asum += std::fabs(t);
          }
          asum = asum > kEps * 100 ? 1 / asum : 0;
          for (int k = 0; k < M; k++) R[i][k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < M; k++) sd += R[i][k] * R[i][k];
      sd = std::sqrt(sd);
    }
    const double s = sd > kMin ? 1 / sd : 0.;
    for (int k = 0; k < M; k++) R[i][k] *= s;
  }
  for (int k = 0; k < M; k++) {
    f1[k] = R[7][k];
    f2[k] = R[8][k];
  }
}

// cv::solveCubic [OpenCV core/mathfuncs.cpp]
int solve_cubic(const double coef[4], double roots[3]) {
  double a0 = coef[0], a1 = coef[1], a2 = coef[2], a3 = coef[3];
  double x0 = 0, x1 = 0, x2 = 0;
  int n = 0;
  if (a0 == 0) {
    if (a1 == 0) {
      if (a2 == 0) {
        n = a3 == 0 ? -1 : 0;
      } else {
        x0 = -a3 / a2;
        n = 1;
      }
    } else {
      double d = a2 * a2 - 4 * a1 * a3;
      if (d >= 0) {
        d = std::sqrt(d);
        const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
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
    const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
    const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
    const double Qcubed = Q * Q * Q;
    double d = Qcubed - R * R;
    const double kPi = 3.1415926535897932384626433832795;
    if (d > 0) {
      const double theta = std::acos(R / std::sqrt(Qcubed));
      const double sqrtQ = std::sqrt(Q);
      const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
      x0 = t0 * std::cos(t1) - t2;
      x1 = t0 * std::cos(t1 + (2. * kPi / 3)) - t2;
      x2 = t0 * std::cos(t1 + (4. * kPi / 3)) - t2;
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
      d = std::sqrt(-d);
      double e = std::pow(d + std::fabs(R), 1. / 3);
      if (R > 0) e = -e;
      x0 = (e + Q / e) - a1 * (1. / 3);
      n = 1;
    }
  }
  roots[0] = x0;
  roots[1] = x1;
  roots[2] = x2;
  return n;
}

// FMEstimatorCallback::run7Point [OpenCV calib3d/fundam.cpp]; F: up to 3 row-major 3x3
int seven_point(const float* m1, const float* m2, double* F) {
  double A[7][9];
  for (int i = 0; i < 7; i++) {
    const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
    const double row[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1};
    std::memcpy(A[i], row, sizeof(row));
  }
  double f1[9], f2[9];
  epipolar_nullspace(A, f1, f2);
  for (int i = 0; i < 9; i++) f1[i] -= f2[i];
  double c[4], r[3] = {0, 0, 0};
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
  const int n = solve_cubic(c, r);
  if (n < 1 || n > 3) return n;
  for (int k = 0; k < n; k++, F += 9) {
    double lambda = r[k], mu = 1.;
    const double s = f1[8] * r[k] + f2[8];
    if (std::fabs(s) > DBL_EPSILON) {
      mu = 1. / s;
      lambda *= mu;
      F[8] = 1.;
    } else {
      F[8] = 0.;
    }
    for (int i = 0; i < 8; i++) F[i] = f1[i] * lambda + f2[i] * mu;
  }
  return n;
}

// FMEstimatorCallback::computeError: max of the two squared point-to-epipolar-line distances
void epipolar_errors(const float* m1, const float* m2, int n, const double* F, float* err) {
  for (int i = 0; i < n; i++) {
    const double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
    double a = F[0] * x1 + F[1] * y1 + F[2];
    double b = F[3] * x1 + F[4] * y1 + F[5];
    double c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    err[i] = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
  }
}

bool last_point_collinear(const float* m, int count) {  // haveCollinearPoints
  const int i = count - 1;
  for (int j = 0; j < i; j++) {
    const double dx1 = m[2 * j] - m[2 * i], dy1 = m[2 * j + 1] - m[2 * i + 1];
    for (int k = 0; k < j; k++) {
      const double dx2 = m[2 * k] - m[2 * i], dy2 = m[2 * k + 1] - m[2 * i + 1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <=
          FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
        return true;
    }
  }
  return false;
}

// RANSACPointSetRegistrator::getSubset for modelPoints = 7
bool draw_subset(const float* m1, const float* m2, int count, float* s1, float* s2, OcvRng& rng,
                 int max_attempts) {
  int idx[7];
  int i = 0, iters = 0;
  for (; iters < max_attempts; iters++) {
    for (i = 0; i < 7 && iters < max_attempts;) {
      int pick, j;
      for (;;) {
        pick = idx[i] = rng.uniform(0, count);
        for (j = 0; j < i; j++)
          if (pick == idx[j]) break;
        if (j == i) break;
      }
      s1[2 * i] = m1[2 * pick];
      s1[2 * i + 1] = m1[2 * pick + 1];
      s2[2 * i] = m2[2 * pick];
      s2[2 * i + 1] = m2[2 * pick + 1];
      i++;
    }
    if (i == 7 && (last_point_collinear(s1, i) || last_point_collinear(s2, i))) continue;
    break;
  }
  return i == 7 && iters < max_attempts;
}

int update_num_iters(double p, double ep, int model_points, int max_iters) {  // RANSACUpdateNumIters
  p = std::min(std::max(p, 0.), 1.);
  ep = std::min(std::max(ep, 0.), 1.);
  double num = std::max(1. - p, DBL_MIN);
  double denom = 1. - std::pow(1. - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

int mark_inliers(const float* m1, const float* m2, int n, const double* F, std::vector<float>& err,
                 uint8_t* mask, double thresh) {
  epipolar_errors(m1, m2, n, F, err.data());
  const float t = (float)(thresh * thresh);
  int good = 0;
  for (int i = 0; i < n; i++) {
    const int f = err[i] <= t;
    mask[i] = (uint8_t)f;
    good += f;
  }
  return good;
}

// RANSAC scoring of one model over points held as double arrays: inlier flags + count for the
// points [i0, i1) (FMEstimatorCallback::computeError + findInliers, element for element)
ESVIO_SIMD_CLONES
int score_block(const double* __restrict x1, const double* __restrict y1, const double* __restrict x2,
                const double* __restrict y2, int i0, int i1, const double* F, float t,
                uint8_t* __restrict mask) {
  const double F0 = F[0], F1 = F[1], F2 = F[2], F3 = F[3], F4 = F[4], F5 = F[5], F6 = F[6], F7 = F[7],
               F8 = F[8];
  int good = 0;
  for (int i = i0; i < i1; i++) {
    double a = F0 * x1[i] + F1 * y1[i] + F2;
    double b = F3 * x1[i] + F4 * y1[i] + F5;
    double c = F6 * x1[i] + F7 * y1[i] + F8;
    const double s2 = 1. / (a * a + b * b);
    const double d2 = x2[i] * a + y2[i] * b + c;
    a = F0 * x2[i] + F3 * y2[i] + F6;
    b = F1 * x2[i] + F4 * y2[i] + F7;
    c = F2 * x2[i] + F5 * y2[i] + F8;
    const double s1 = 1. / (a * a + b * b);
    const double d1 = x1[i] * a + y1[i] * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    const float err = (float)(e1 < e2 ? e2 : e1);  // std::max(e1, e2)
    const int f = err <= t;
    mask[i] = (uint8_t)f;
    good += f;
  }
  return good;
}

// Inlier count of a model if it can still beat `need` (> need wins), else a value <= need: the
// RANSAC loop only looks at the flags of a model that improves on the best one, so scoring stops
// as soon as that has become impossible.
int mark_inliers_bounded(const double* x1, const double* y1, const double* x2, const double* y2, int n,
                         const double* F, uint8_t* mask, double thresh, int need) {
  const float t = (float)(thresh * thresh);
  constexpr int kBlock = 32;
  int good = 0;
  for (int i0 = 0; i0 < n; i0 += kBlock) {
    const int i1 = std::min(n, i0 + kBlock);
    good += score_block(x1, y1, x2, y2, i0, i1, F, t, mask);
    if (good + (n - i1) <= need) return good;  // cannot exceed `need` any more
  }
  return good;
}

}  // namespace

// ---------------------------------------------------------------------------- helper threads
// The RANSAC loop is ~100-350 iterations of (draw 7 points, 7-point solver, score <= 3 models);
// only the draws (the cv::RNG sequence) and the "is this model better than the best so far"
// bookkeeping are sequential.  With a pool the calling thread draws the subsets in order and
// replays the results in order — so best model, inlier flags and iteration count are those of the
// sequential loop, bit for bit — while helper threads solve and score the iterations in between.
// The helpers spin while a tracker is busy (a futex wake-up costs more than the whole job) and go
// to sleep after kIdleSpinUs without work.
namespace {
constexpr int kRansacMaxIters = 1000;
constexpr int kRansacWindow = 48;      // iterations drawn ahead of the replay position
constexpr int kIdleSpinUs = 2000;

inline void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
  __builtin_ia32_pause();
#endif
}

struct alignas(64) IterResult {
  int nm;
  int good[3];
  double models[27];
};
constexpr uint32_t kCollinear = 0xffffffffu;  // summary code: the drawn subset fails checkSubset
}  // namespace

// State of one job.  Two of them alternate (job parity): a helper that is descheduled in the middle
// of an iteration keeps working on its own job's buffers and cannot touch the next job's; the
// caller does not wait for it (it redoes an iteration whose result does not arrive, and only a
// buffer's next use, two jobs later, waits for helpers still inside it).
struct RansacJob {
  alignas(64) std::atomic<int> next{0};    // next iteration to hand out
  alignas(64) std::atomic<int> avail{0};   // subsets drawn so far
  alignas(64) std::atomic<int> bound{0};   // best inlier count among the replayed iterations
  alignas(64) std::atomic<int> active{0};  // helpers inside this job
  std::vector<double> xy;                  // x1 | y1 | x2 | y2, count each
  std::vector<float> pts;                  // m1 | m2 (interleaved x,y), 2*count each
  int count = 0;
  double thr = 0;
  // The caller only runs the cv::RNG index draws (7 distinct indices per iteration, ~15 ns); the
  // gather and getSubset's collinearity test are part of the evaluation.  A subset that fails the
  // test (the reference then redraws, consuming more random numbers) is reported as kCollinear and
  // the caller continues sequentially from that iteration's saved generator state.
  std::vector<int32_t> picks;  // [iteration][8]
  std::unique_ptr<IterResult[]> res;
  // per iteration (epoch << 32) | max inlier count of its models (or kCollinear): 8 iterations per
  // cache line, so the in-order replay costs ~1/8 of a line transfer per iteration and reads the
  // full result only when the iteration improves on the best model
  std::unique_ptr<std::atomic<uint64_t>[]> summary;

  uint32_t evaluate(int idx, IterResult& r, std::vector<uint8_t>& scratch) const {
    float s1[14], s2[14];
    const float *m1 = pts.data(), *m2 = m1 + 2 * (size_t)count;
    const int32_t* pk = &picks[(size_t)idx * 8];
    for (int i = 0; i < 7; i++) {
      s1[2 * i] = m1[2 * pk[i]];
      s1[2 * i + 1] = m1[2 * pk[i] + 1];
      s2[2 * i] = m2[2 * pk[i]];
      s2[2 * i + 1] = m2[2 * pk[i] + 1];
    }
    if (last_point_collinear(s1, 7) || last_point_collinear(s2, 7)) {
      r.nm = -1;
      return kCollinear;
    }
    r.nm = seven_point(s1, s2, r.models);
    const int need = bound.load(std::memory_order_relaxed);
    if ((int)scratch.size() < count) scratch.resize(count);
    const double *x1 = xy.data(), *y1 = x1 + count, *x2 = y1 + count, *y2 = x2 + count;
    int best = 0;
    for (int k = 0; k < r.nm; k++) {
      r.good[k] = mark_inliers_bounded(x1, y1, x2, y2, count, r.models + 9 * k, scratch.data(), thr, need);
      best = std::max(best, r.good[k]);
    }
    return (uint32_t)best;
  }
};

struct RansacPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t wake_seq = 0;  // (under mu)
  std::atomic<int> sleepers{0};
  std::atomic<bool> quit{false};
  alignas(64) std::atomic<uint32_t> epoch{0};  // odd while a job is open; job (epoch >> 1) & 1
  RansacJob job[2];
  // helpers are kept on the cores that share the caller's L3 (one CCD): an iteration is ~0.5 us of
  // work, so the hand-over has to cost a same-die cache line transfer, not a cross-socket one
  std::vector<int> near_cpus;  // the caller's L3 domain minus the caller's own core (empty: unknown)
  std::vector<int> l3_cpus;    // the whole domain
  int near_of = -1;            // the CPU that set was made for

  void pin_near_caller();

  void helper() {
    std::vector<uint8_t> scratch;
    uint32_t seen = 0;
    auto idle_since = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
      if (quit.load(std::memory_order_acquire)) return;
      const uint32_t e = epoch.load(std::memory_order_acquire);
      if (!(e & 1) || e == seen) {
        cpu_relax();
        if ((++spins & 1023) == 0 &&
            std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(kIdleSpinUs)) {
          std::unique_lock<std::mutex> lk(mu);
          const uint64_t my = wake_seq;
          sleepers.fetch_add(1, std::memory_order_acq_rel);
          cv.wait(lk, [&] { return wake_seq != my || quit.load(std::memory_order_acquire); });
          sleepers.fetch_sub(1, std::memory_order_acq_rel);
          idle_since = std::chrono::steady_clock::now();
        }
        continue;
      }
      RansacJob& J = job[(e >> 1) & 1];
      J.active.fetch_add(1, std::memory_order_acq_rel);
      // (the job may have closed — and this buffer's next job opened — in between: then leave)
      while (epoch.load(std::memory_order_acquire) == e) {
        const int idx = J.next.fetch_add(1, std::memory_order_relaxed);
        if (idx >= kRansacMaxIters) break;
        bool go = true;
        while (J.avail.load(std::memory_order_acquire) <= idx) {
          if (epoch.load(std::memory_order_acquire) != e) {
            go = false;
            break;
          }
          cpu_relax();
        }
        if (!go) break;
        const uint32_t code = J.evaluate(idx, J.res[idx], scratch);
        J.summary[idx].store(((uint64_t)e << 32) | code, std::memory_order_release);
      }
      J.active.fetch_sub(1, std::memory_order_release);
      seen = e;
      idle_since = std::chrono::steady_clock::now();
    }
  }
};

namespace {
// "0-7,128-135" -> cpu numbers
std::vector<int> read_cpu_list(const char* path) {
  std::vector<int> out;
  FILE* f = std::fopen(path, "r");
  if (!f) return out;
  char buf[512];
  if (std::fgets(buf, sizeof buf, f)) {
    const char* p = buf;
    while (*p) {
      char* e;
      const long a = std::strtol(p, &e, 10);
      if (e == p) break;
      long b = a;
      p = e;
      if (*p == '-') {
        b = std::strtol(p + 1, &e, 10);
        p = e;
      }
      for (long v = a; v <= b && out.size() < 1024; v++) out.push_back((int)v);
      if (*p == ',') p++;
    }
  }
  std::fclose(f);
  return out;
}
}  // namespace

void RansacPool::pin_near_caller() {
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
  const int cpu = sched_getcpu();
  if (cpu < 0) return;
  if (near_of >= 0) {
    if (cpu == near_of) return;
    for (int v : l3_cpus)
      if (v == cpu) return;  // (moved within the die — fine)
  }
  char path[128];
  std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  std::vector<int> l3 = read_cpu_list(path);
  std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
  const std::vector<int> self = read_cpu_list(path);
  near_of = cpu;
  l3_cpus = l3;
  near_cpus.clear();
  std::vector<int> primary;  // one logical CPU per physical core of the domain, caller's core left out
  for (int v : l3) {
    bool mine = v == cpu;
    for (int w : self) mine = mine || v == w;
    if (mine) continue;
    near_cpus.push_back(v);
    std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", v);
    const std::vector<int> sib = read_cpu_list(path);
    if (sib.empty() || sib[0] == v) primary.push_back(v);
  }
  // a spinning helper on the SMT sibling of another helper (or of the caller) takes issue slots
  // from it: keep to one hardware thread per core when the domain has enough cores
  if (primary.size() >= th.size()) near_cpus = primary;
  if (near_cpus.size() < th.size()) {  // unknown topology or a tiny L3 domain: leave it to the OS
    near_cpus.clear();
    return;
  }
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int v : near_cpus)
    if (v < CPU_SETSIZE) CPU_SET(v, &set);
  for (auto& t : th) (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
#endif
}

RansacPool* ransac_pool_create(int helpers) {
  if (helpers <= 0) return nullptr;
  RansacPool* p = new RansacPool();
  for (RansacJob& J : p->job) {
    J.picks.resize((size_t)kRansacMaxIters * 8);
    J.res.reset(new IterResult[kRansacMaxIters]);
    J.summary.reset(new std::atomic<uint64_t>[kRansacMaxIters]);
    for (int i = 0; i < kRansacMaxIters; i++) J.summary[i].store(0, std::memory_order_relaxed);
  }
  for (int i = 0; i < helpers; i++) p->th.emplace_back([p] { p->helper(); });
  p->pin_near_caller();
  return p;
}

void ransac_pool_wake(RansacPool* p) {
  if (!p) return;
  p->pin_near_caller();  // (a vDSO call unless the calling thread has moved to another die)
  if (p->sleepers.load(std::memory_order_acquire) == 0) return;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->wake_seq++;
  }
  p->cv.notify_all();
}

void ransac_pool_destroy(RansacPool* p) {
  if (!p) return;
  p->quit.store(true, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->wake_seq++;
  }
  p->cv.notify_all();
  for (auto& t : p->th) t.join();
  delete p;
}

namespace {
// the RANSAC loop of find_fundamental_mat with helpers (count >= 15)
int ransac_pooled(RansacPool* P, const float* m1, const float* m2, int count, double thr, double conf,
                  const double* xy, uint8_t* status) {
  const int kModelPoints = 7;
  const uint32_t e = P->epoch.load(std::memory_order_relaxed) + 1;  // (odd: this job's tag)
  RansacJob& J = P->job[(e >> 1) & 1];
  // a helper that stalled inside this buffer's previous job (two jobs ago) is waited for here
  while (J.active.load(std::memory_order_acquire) != 0) cpu_relax();
  J.xy.assign(xy, xy + 4 * (size_t)count);
  J.pts.resize(4 * (size_t)count);
  std::memcpy(J.pts.data(), m1, 2 * (size_t)count * sizeof(float));
  std::memcpy(J.pts.data() + 2 * (size_t)count, m2, 2 * (size_t)count * sizeof(float));
  J.count = count;
  J.thr = thr;
  J.next.store(0, std::memory_order_relaxed);
  J.avail.store(0, std::memory_order_relaxed);
  J.bound.store(kModelPoints - 1, std::memory_order_relaxed);
  P->epoch.store(e, std::memory_order_release);  // open
  ransac_pool_wake(P);

  OcvRng rng;
  std::vector<uint8_t> scratch(count);
  std::vector<uint64_t> rng_before;  // generator state ahead of each iteration's draw
  rng_before.reserve(256);
  IterResult local;
  double best_F[9];
  const double *x1 = J.xy.data(), *y1 = x1 + count, *x2 = y1 + count, *y2 = x2 + count;
  int drawn = 0, rp = 0, niters = kRansacMaxIters, best_good = 0;
  bool have_best = false;
  int collinear_at = -1;
  unsigned stalled = 0;
  while (collinear_at < 0) {
    // replay, in iteration order, what has been evaluated
    while (rp < niters && rp < drawn) {
      const IterResult* r = &J.res[rp];
      const uint64_t sm = J.summary[rp].load(std::memory_order_acquire);
      uint32_t code;
      if ((uint32_t)(sm >> 32) == e) {
        code = (uint32_t)sm;
      } else if (stalled > 64) {
        // whoever took this iteration is not delivering (descheduled?): do it here, the late
        // result is simply never looked at
        code = J.evaluate(rp, local, scratch);
        r = &local;
      } else {
        break;
      }
      stalled = 0;
      if (code == kCollinear) {
        collinear_at = rp;
        break;
      }
      if ((int)code > std::max(best_good, kModelPoints - 1))
        for (int k = 0; k < r->nm; k++)
          if (r->good[k] > std::max(best_good, kModelPoints - 1)) {
            best_good = r->good[k];
            std::memcpy(best_F, r->models + 9 * k, sizeof(best_F));
            have_best = true;
            niters = update_num_iters(conf, (double)(count - best_good) / count, kModelPoints, niters);
          }
      rp++;
      J.bound.store(std::max(best_good, kModelPoints - 1), std::memory_order_relaxed);
    }
    if (rp >= niters || collinear_at >= 0) break;
    if (drawn < niters && drawn - rp < kRansacWindow) {
      for (int j = 0; j < 8 && drawn < niters; j++) {  // RANSACPointSetRegistrator::getSubset's draws
        rng_before.push_back(rng.s);
        int32_t* pk = &J.picks[(size_t)drawn * 8];
        for (int i = 0; i < 7; i++) {
          int pick, q;
          for (;;) {
            pick = pk[i] = rng.uniform(0, count);
            for (q = 0; q < i; q++)
              if (pick == pk[q]) break;
            if (q == i) break;
          }
        }
        drawn++;
      }
      J.avail.store(drawn, std::memory_order_release);
      continue;
    }
    // nothing to draw: take an iteration like a helper does
    int idx = J.next.load(std::memory_order_relaxed);
    if (idx < drawn && J.next.compare_exchange_strong(idx, idx + 1, std::memory_order_relaxed)) {
      const uint32_t code = J.evaluate(idx, J.res[idx], scratch);
      J.summary[idx].store(((uint64_t)e << 32) | code, std::memory_order_release);
    } else {
      cpu_relax();
      stalled++;
    }
  }
  P->epoch.store(e + 1, std::memory_order_release);  // closed; helpers drop out on their own
  if (collinear_at >= 0) {
    // the loop as the reference runs it, from the iteration whose first subset was rejected
    rng.s = rng_before[collinear_at];
    float s1[14], s2[14];
    for (int iter = collinear_at; iter < niters; iter++) {
      if (!draw_subset(m1, m2, count, s1, s2, rng, 10000)) {
        if (iter == 0) return 0;
        break;
      }
      const int nm = seven_point(s1, s2, local.models);
      for (int k = 0; k < nm; k++) {
        const int good = mark_inliers_bounded(x1, y1, x2, y2, count, local.models + 9 * k, scratch.data(), thr,
                                              std::max(best_good, kModelPoints - 1));
        if (good > std::max(best_good, kModelPoints - 1)) {
          best_good = good;
          std::memcpy(best_F, local.models + 9 * k, sizeof(best_F));
          have_best = true;
          niters = update_num_iters(conf, (double)(count - good) / count, kModelPoints, niters);
        }
      }
    }
  }
  if (have_best) score_block(x1, y1, x2, y2, 0, count, best_F, (float)(thr * thr), status);
  return best_good;
}
}  // namespace

int find_fundamental_mat(const float* m1, const float* m2, int count, double thr, double conf,
                         uint8_t* status, RansacPool* pool) {
  const int kModelPoints = 7, kMaxIters = kRansacMaxIters;
  std::fill(status, status + count, (uint8_t)0);
  if (count < 7) return 0;
  if (thr <= 0) thr = 3;
  if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
  std::vector<float> err(count);
  std::vector<uint8_t> mask(count);
  float s1[14], s2[14];
  double models[27], best[9];
  OcvRng rng;
  if (count == 7) {
    const int n = seven_point(m1, m2, models);
    std::fill(status, status + count, (uint8_t)1);
    return n > 0 ? count : 0;
  }
  if (count >= 15) {  // RANSAC
    std::vector<double> xy((size_t)4 * count);
    double *x1 = xy.data(), *y1 = x1 + count, *x2 = y1 + count, *y2 = x2 + count;
    for (int i = 0; i < count; i++) {
      x1[i] = m1[2 * i];
      y1[i] = m1[2 * i + 1];
      x2[i] = m2[2 * i];
      y2[i] = m2[2 * i + 1];
    }
    if (pool) return ransac_pooled(pool, m1, m2, count, thr, conf, xy.data(), status);
    int niters = kMaxIters, best_good = 0;
    for (int iter = 0; iter < niters; iter++) {
      if (!draw_subset(m1, m2, count, s1, s2, rng, 10000)) {
        if (iter == 0) return 0;
        break;
      }
      const int nm = seven_point(s1, s2, models);
      if (nm <= 0) continue;
      for (int k = 0; k < nm; k++) {
        const int good = mark_inliers_bounded(x1, y1, x2, y2, count, models + 9 * k, mask.data(), thr,
                                              std::max(best_good, kModelPoints - 1));
        if (good > std::max(best_good, kModelPoints - 1)) {
          std::memcpy(status, mask.data(), count);
          best_good = good;
          niters = update_num_iters(conf, (double)(count - good) / count, kModelPoints, niters);
        }
      }
    }
    return best_good;
  }
  // LMedS for 8..14 points
  const int niters = update_num_iters(conf, 0.45, kModelPoints, kMaxIters);
  double min_median = DBL_MAX;
  std::vector<float> errs(count);
  for (int iter = 0; iter < niters; iter++) {
    if (!draw_subset(m1, m2, count, s1, s2, rng, 300)) {
      if (iter == 0) return 0;
      break;
    }
    const int nm = seven_point(s1, s2, models);
    if (nm <= 0) continue;
    for (int k = 0; k < nm; k++) {
      epipolar_errors(m1, m2, count, models + 9 * k, errs.data());
      std::nth_element(errs.begin(), errs.begin() + count / 2, errs.end());
      const double median = errs[count / 2];
      if (median < min_median) {
        min_median = median;
        std::memcpy(best, models + 9 * k, sizeof(best));
      }
    }
  }
  if (min_median < DBL_MAX) {
    double sigma = 2.5 * 1.4826 * (1 + 5. / (count - kModelPoints)) * std::sqrt(min_median);
    sigma = std::max(sigma, 0.001);
    return mark_inliers(m1, m2, count, best, err, status, sigma);
  }
  return 0;
}

}  // namespace host
}  // namespace esvio
