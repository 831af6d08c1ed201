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
#include <sys/resource.h>
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
#if defined(__HIP_DEVICE_COMPILE__) || defined(ESVIO_NO_SIMD_CLONES)  // (sanitizer builds: no ifunc resolvers)
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

// JacobiSVDImpl_'s sweep limit, max(m, 30) = 30 for the 7x9 system.  (A macro only so that
// tests/jacobi_cap_check.cpp can make the limit bite: real systems converge in ~5 sweeps.)
#ifndef ESVIO_JACOBI_MAX_SWEEPS
#define ESVIO_JACOBI_MAX_SWEEPS 30
#endif
constexpr int kJacobiMaxSweeps = ESVIO_JACOBI_MAX_SWEEPS;

// cv::hypot [OpenCV 4.2 core/src/lapack.cpp]: JacobiSVDImpl_'s `hypot((double)p, beta)` is an
// unqualified call inside namespace cv, where the file's own
//     template<typename _Tp> static inline _Tp hypot(_Tp a, _Tp b)
// (declared just above JacobiImpl_) hides ::hypot — so the rotations get a * sqrt(1 + (b/a)^2), not
// libm's correctly-scaled hypot; the two differ in the last bit often enough to matter.
// Restated from the published source, unpinned like every OpenCV restatement here.  IEEE division,
// square root, multiply and add only (mul and add stay separate: -ffp-contract=off): no libm, so the
// value does not depend on the host's glibc, and the same code runs in the vector lanes.
inline double cv_hypot(double a, double b) {
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
  for (int sweep = 0; sweep < kJacobiMaxSweeps; sweep++) {  // max(m, 30)
    bool changed = false;
    for (int i = 0; i < N - 1; i++)
      for (int j = i + 1; j < N; j++) {
        double* __restrict ri = R[i];
        double* __restrict rj = R[j];
        double a = W[i], b = W[j], p = 0;
        for (int k = 0; k < M; k++) p += ri[k] * rj[k];
        if (std::fabs(p) <= kEps * std::sqrt(a * b)) continue;
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

// The same decomposition for kLanes independent systems at once, one per vector lane.  A single
// system is a chain of dependent operations (dot product -> hypot -> sqrt/div -> rotation, ~85
// times), so one at a time the vector units idle; side by side every lane performs exactly the
// scalar sequence above on its own data: a pair that a lane skips leaves that lane's rows and norms
// untouched (selected by mask), a lane that has converged sees only skipped pairs in the sweeps the
// others still need, and the sweep limit is common.  hypot is cv_hypot, all lanes at once.  The
// made-up rows use the first draw of the sign vector (the same for every system); a lane that
// would need OpenCV's retry (a zero singular value or a vanished projection) makes the function
// return false and the caller does those systems with the scalar routine.
#ifndef ESVIO_RANSAC_LANES
#define ESVIO_RANSAC_LANES 8
#endif
constexpr int kLanes = ESVIO_RANSAC_LANES;

// The cyclic sweep (0,1),(0,2)..(5,6) taken as levels of pairs (i, j) with i + j = level: two rotations
// that share no row commute exactly — each reads and writes only its own two rows and norms — and the
// pairs containing a given row r come in the sweep's own order ((0,r)..(r-1,r),(r,r+1)..(r,6): i + j
// strictly increasing), so every row goes through the same rotations in the same order with the same
// operands as in the sequential sweep: the result is bit-identical.  What changes is the dependent
// chain: a pair is dot product -> hypot -> sqrt/div -> rotation -> norms, ~200 cycles of latency with
// the vector units idle in between; the pairs of a level are independent chains written side by
// side (K per stage).  Levels 8..11 of a sweep leave rows 0..4 alone, so levels 1..4 of the NEXT
// sweep go with them: 7 steps of 3 pairs per sweep instead of 21 of one.  (A sweep that turns out to
// have rotated nothing ends the loop; the next sweep's pairs taken along with its last levels saw
// the rows and norms this sweep's own tests saw and skipped like them, so nothing has run ahead.)

// The Hestenes rotations of K row pairs that share no row, all lanes at once; bit q of the result:
// some lane of pair q rotated.  (always_inline: the body is compiled for the ISA of the clone that calls it.)
template <int K, int I0, int J0, int I1 = 0, int J1 = 0, int I2 = 0, int J2 = 0>
static inline __attribute__((always_inline)) unsigned jacobi_pairs(double (*R)[9][kLanes], double (*W)[kLanes]) {
  constexpr int M = 9, L = kLanes;
  constexpr int ij[3][2] = {{I0, J0}, {I1, J1}, {I2, J2}};
  constexpr double kEps = DBL_EPSILON * 10;
  alignas(64) double p[K][L], lim[K][L];
  alignas(64) int64_t on[K][L];
  for (int q = 0; q < K; q++)
    for (int l = 0; l < L; l++) p[q][l] = 0;
  for (int k = 0; k < M; k++)
    for (int q = 0; q < K; q++) {
      const double* __restrict ri = R[ij[q][0]][k];
      const double* __restrict rj = R[ij[q][1]][k];
      for (int l = 0; l < L; l++) p[q][l] += ri[l] * rj[l];
    }
  unsigned rotated = 0;  // bit q: some lane of pair q rotates
  for (int q = 0; q < K; q++) {
    const double* __restrict wi = W[ij[q][0]];
    const double* __restrict wj = W[ij[q][1]];
    for (int l = 0; l < L; l++) lim[q][l] = kEps * std::sqrt(wi[l] * wj[l]);
    int64_t any = 0;
    for (int l = 0; l < L; l++) {
      on[q][l] = !(std::fabs(p[q][l]) <= lim[q][l]) ? -1 : 0;
      any |= on[q][l];
    }
    rotated |= any ? 1u << q : 0u;
  }
  if (!rotated) return 0;
  alignas(64) double c[K][L], s[K][L], a[K][L], b[K][L];
  for (int q = 0; q < K; q++) {
    const double* __restrict wi = W[ij[q][0]];
    const double* __restrict wj = W[ij[q][1]];
    for (int l = 0; l < L; l++) {
      const double p2 = p[q][l] * 2, beta = wi[l] - wj[l];
      // gamma = cv_hypot(p2, beta) for the lanes that rotate: hi * sqrt(1 + (lo / hi)^2) with hi / lo the
      // larger / smaller magnitude is the scalar routine's value on either of its branches (a lane
      // that rotates has p != 0, so hi > 0); a lane that does not rotate computes on 1.0
      const double x = on[q][l] ? std::fabs(p2) : 1.0, y = on[q][l] ? std::fabs(beta) : 1.0;
      const double hi = x > y ? x : y, lo = x > y ? y : x;
      const double r = lo / hi;
      const double gamma = hi * std::sqrt(1 + r * r);
      // beta < 0:  s = sqrt(((gamma - beta) * 0.5) / gamma),  c = p2 / (gamma * s * 2)
      // else:      c = sqrt((gamma + beta) / (gamma * 2)),    s = p2 / (gamma * c * 2)
      // — the lane's own branch, operands selected before the division (one sqrt and two divisions
      // per lane instead of both branches' two and four)
      const bool neg = beta < 0;
      const double num = neg ? (gamma - beta) * 0.5 : gamma + beta;
      const double den = neg ? gamma : gamma * 2;
      const double first = std::sqrt(num / den);
      const double second = p2 / (gamma * first * 2);
      c[q][l] = neg ? second : first;
      s[q][l] = neg ? first : second;
      a[q][l] = 0;
      b[q][l] = 0;
    }
  }
  for (int k = 0; k < M; k++)
    for (int q = 0; q < K; q++) {
      double* __restrict ri = R[ij[q][0]][k];
      double* __restrict rj = R[ij[q][1]][k];
      for (int l = 0; l < L; l++) {
        const double x = ri[l], y = rj[l];
        const double t0 = c[q][l] * x + s[q][l] * y, t1 = -s[q][l] * x + c[q][l] * y;
        ri[l] = on[q][l] ? t0 : x;
        rj[l] = on[q][l] ? t1 : y;
        a[q][l] += t0 * t0;
        b[q][l] += t1 * t1;
      }
    }
  for (int q = 0; q < K; q++) {
    double* __restrict wi = W[ij[q][0]];
    double* __restrict wj = W[ij[q][1]];
    for (int l = 0; l < L; l++) {
      wi[l] = on[q][l] ? a[q][l] : wi[l];
      wj[l] = on[q][l] ? b[q][l] : wj[l];
    }
  }
  return rotated;
}

ESVIO_SIMD_CLONES
bool epipolar_nullspace_lanes(const double (*A)[7][9], double (*f1)[9], double (*f2)[9]) {
  constexpr int M = 9, N = 7, L = kLanes;
  constexpr double kMin = DBL_MIN, kEps = DBL_EPSILON * 10;
  alignas(64) double R[9][M][L];
  alignas(64) double W[N][L];
  for (int i = 0; i < N; i++)
    for (int k = 0; k < M; k++)
      for (int l = 0; l < L; l++) R[i][k][l] = A[l][i][k];
  for (int i = 0; i < N; i++) {
    alignas(64) double sd[L];
    for (int l = 0; l < L; l++) sd[l] = 0;
    for (int k = 0; k < M; k++)
      for (int l = 0; l < L; l++) sd[l] += R[i][k][l] * R[i][k][l];
    for (int l = 0; l < L; l++) W[i][l] = sd[l];
  }
  // sweep 0, levels 1..4
  unsigned cur = jacobi_pairs<1, 0, 1>(R, W);
  cur |= jacobi_pairs<1, 0, 2>(R, W);
  cur |= jacobi_pairs<2, 0, 3, 1, 2>(R, W);
  cur |= jacobi_pairs<2, 0, 4, 1, 3>(R, W);
  for (int sweep = 0; sweep < kJacobiMaxSweeps; sweep++) {  // max(m, 30)
    cur |= jacobi_pairs<3, 0, 5, 1, 4, 2, 3>(R, W);  // levels 5..7
    cur |= jacobi_pairs<3, 0, 6, 1, 5, 2, 4>(R, W);
    cur |= jacobi_pairs<3, 1, 6, 2, 5, 3, 4>(R, W);
    unsigned next = 0;
    if (sweep + 1 < kJacobiMaxSweeps) {
      // levels 8..11 of this sweep (the first pair(s)) with levels 1..4 of the next one
      unsigned r = jacobi_pairs<3, 2, 6, 3, 5, 0, 1>(R, W);
      cur |= r & 3u;
      next |= r & 4u;
      r = jacobi_pairs<3, 3, 6, 4, 5, 0, 2>(R, W);
      cur |= r & 3u;
      next |= r & 4u;
      r = jacobi_pairs<3, 4, 6, 0, 3, 1, 2>(R, W);
      cur |= r & 1u;
      next |= r & 6u;
      r = jacobi_pairs<3, 5, 6, 0, 4, 1, 3>(R, W);
      cur |= r & 1u;
      next |= r & 6u;
    } else {
      cur |= jacobi_pairs<2, 2, 6, 3, 5>(R, W);
      cur |= jacobi_pairs<2, 3, 6, 4, 5>(R, W);
      cur |= jacobi_pairs<1, 4, 6>(R, W);
      cur |= jacobi_pairs<1, 5, 6>(R, W);
    }
    if (!cur) break;
    cur = next;
  }
  for (int i = 0; i < N; i++) {
    alignas(64) double sd[L];
    for (int l = 0; l < L; l++) sd[l] = 0;
    for (int k = 0; k < M; k++)
      for (int l = 0; l < L; l++) sd[l] += R[i][k][l] * R[i][k][l];
    for (int l = 0; l < L; l++) W[i][l] = std::sqrt(sd[l]);
  }
  bool plain = true;
  for (int l = 0; l < L; l++) {
    for (int i = 0; i < N - 1; i++) {
      int j = i;
      for (int k = i + 1; k < N; k++)
        if (W[j][l] < W[k][l]) j = k;
      if (i != j) {
        std::swap(W[i][l], W[j][l]);
        for (int k = 0; k < M; k++) std::swap(R[i][k][l], R[j][k][l]);
      }
    }
    for (int i = 0; i < N; i++) plain = plain && W[i][l] > kMin;
  }
  if (!plain) return false;
  for (int i = 0; i < N; i++) {
    alignas(64) double inv[L];
    for (int l = 0; l < L; l++) inv[l] = 1 / W[i][l];
    for (int k = 0; k < M; k++)
      for (int l = 0; l < L; l++) R[i][k][l] *= inv[l];
  }
  uint64_t rng = 0x12345678;
  for (int i = N; i < 9; i++) {
    const double v0 = 1. / M;
    for (int k = 0; k < M; k++) {
      rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
      const double v = ((unsigned)rng & 256) != 0 ? v0 : -v0;
      for (int l = 0; l < L; l++) R[i][k][l] = v;
    }
    alignas(64) double sd[L], asum[L];
    for (int round = 0; round < 2; round++)
      for (int j = 0; j < i; j++) {
        for (int l = 0; l < L; l++) {
          sd[l] = 0;
          asum[l] = 0;
        }
        for (int k = 0; k < M; k++)
          for (int l = 0; l < L; l++) sd[l] += R[i][k][l] * R[j][k][l];
        for (int k = 0; k < M; k++)
          for (int l = 0; l < L; l++) {
            const double t = R[i][k][l] - sd[l] * R[j][k][l];
            R[i][k][l] = t;
            asum[l] += std::fabs(t);
          }
        for (int l = 0; l < L; l++) asum[l] = asum[l] > kEps * 100 ? 1 / asum[l] : 0;
        for (int k = 0; k < M; k++)
          for (int l = 0; l < L; l++) R[i][k][l] *= asum[l];
      }
    for (int l = 0; l < L; l++) sd[l] = 0;
    for (int k = 0; k < M; k++)
      for (int l = 0; l < L; l++) sd[l] += R[i][k][l] * R[i][k][l];
    for (int l = 0; l < L; l++) {
      sd[l] = std::sqrt(sd[l]);
      plain = plain && sd[l] > kMin;
    }
    if (!plain) return false;
    for (int l = 0; l < L; l++) sd[l] = 1 / sd[l];
    for (int k = 0; k < M; k++)
      for (int l = 0; l < L; l++) R[i][k][l] *= sd[l];
  }
  for (int l = 0; l < L; l++)
    for (int k = 0; k < M; k++) {
      f1[l][k] = R[7][k][l];
      f2[l][k] = R[8][k][l];
    }
  return true;
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

// FMEstimatorCallback::run7Point [OpenCV calib3d/fundam.cpp]: the 7x9 system ...
void epipolar_system(const float* m1, const float* m2, double A[7][9]) {
  for (int i = 0; i < 7; i++) {
    const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
    const double row[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1};
    std::memcpy(A[i], row, sizeof(row));
  }
}

// ... and, from the basis f1, f2 of its null space, the matrices lambda f1 + (1 - lambda) f2 of
// determinant 0; F: up to 3 row-major 3x3 (f1 is overwritten)
int seven_point_from_basis(double f1[9], const double f2[9], double* F) {
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

int seven_point(const float* m1, const float* m2, double* F) {
  double A[7][9], f1[9], f2[9];
  epipolar_system(m1, m2, A);
  epipolar_nullspace(A, f1, f2);
  return seven_point_from_basis(f1, f2, F);
}

// seven_point for cnt <= kLanes subsets side by side (s1/s2: 14 floats per subset); nm[l] and
// models[l] are what seven_point returns for subset l
void seven_point_lanes(int cnt, const float (*s1)[14], const float (*s2)[14], int* nm, double (*models)[27]) {
  double A[kLanes][7][9], f1[kLanes][9], f2[kLanes][9];
  for (int l = 0; l < kLanes; l++) epipolar_system(s1[l < cnt ? l : 0], s2[l < cnt ? l : 0], A[l]);
  if (!epipolar_nullspace_lanes(A, f1, f2))
    for (int l = 0; l < cnt; l++) epipolar_nullspace(A[l], f1[l], f2[l]);
  for (int l = 0; l < cnt; l++) nm[l] = seven_point_from_basis(f1[l], f2[l], models[l]);
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
                 int max_attempts, int32_t* picked = nullptr) {
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
  if (picked)
    for (int k = 0; k < 7; k++) picked[k] = idx[k];
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
// points [i0, i1) (FMEstimatorCallback::computeError + findInliers, element for element).
// (Measured and dropped: deciding a point from bounds on d^2 against t (a^2 + b^2) and dividing only in
// blocks that hold a point within 2^-20 of the threshold — same flags on 2.5 M points, but on the EPYC
// 9575F, where a zmm division costs little, the extra compares made a RANSAC call slower: 44.6 -> 50.2
// us on one thread, 10-11 -> 15 us of each bench step with 8.)
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
constexpr int kRansacWindow = 128;     // iterations drawn ahead of the replay position (>= threads x kLanes)
constexpr int kIdleSpinUs = 2000;

struct alignas(64) IterResult {
  int nm;
  int good[3];
  double models[27];
};
constexpr uint32_t kCollinear = 0xffffffffu;  // summary code: the drawn subset fails checkSubset
}  // namespace

namespace {
std::atomic<uint64_t> g_rs_calls{0}, g_rs_iters{0}, g_rs_points{0}, g_rs_ns{0}, g_rs_lm_calls{0}, g_rs_lm_ns{0};
std::atomic<uint64_t> g_rs_max_ns{0}, g_rs_lm_max_ns{0}, g_rs_redone{0}, g_rs_solo{0}, g_rs_skipped{0}, g_rs_helper_sw{0};
void atomic_max(std::atomic<uint64_t>& a, uint64_t v) {
  uint64_t cur = a.load(std::memory_order_relaxed);
  while (v > cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {
  }
}
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
  bool lmeds = false;  // score = median of the errors (as float bits in good[]), not an inlier count
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
    if (!gather(idx, s1, s2)) {
      r.nm = -1;
      return kCollinear;
    }
    r.nm = seven_point(s1, s2, r.models);
    return score(r, scratch);
  }

  // Iterations [idx0, idx0 + cnt), cnt <= kLanes, their 7-point systems solved side by side; each
  // iteration's summary is published as soon as it is scored.  Same results as evaluate() one by one.
  void evaluate_lanes(int idx0, int cnt, uint32_t e, std::vector<uint8_t>& scratch) {
    float s1[kLanes][14], s2[kLanes][14];
    int lane_of[kLanes], nm[kLanes], live = 0;
    double models[kLanes][27];
    for (int i = 0; i < cnt; i++) {
      if (gather(idx0 + i, s1[live], s2[live])) {
        lane_of[i] = live++;
      } else {
        lane_of[i] = -1;
        res[idx0 + i].nm = -1;
        summary[idx0 + i].store(((uint64_t)e << 32) | kCollinear, std::memory_order_release);
      }
    }
    if (!live) return;
    seven_point_lanes(live, s1, s2, nm, models);
    for (int i = 0; i < cnt; i++) {
      if (lane_of[i] < 0) continue;
      IterResult& r = res[idx0 + i];
      r.nm = nm[lane_of[i]];
      if (r.nm > 0) std::memcpy(r.models, models[lane_of[i]], sizeof(double) * 9 * r.nm);
      summary[idx0 + i].store(((uint64_t)e << 32) | score(r, scratch), std::memory_order_release);
    }
  }

 private:
  bool gather(int idx, float* s1, float* s2) const {  // false: getSubset's checkSubset rejects it
    const float *m1 = pts.data(), *m2 = m1 + 2 * (size_t)count;
    const int32_t* pk = &picks[(size_t)idx * 8];
    for (int i = 0; i < 7; i++) {
      s1[2 * i] = m1[2 * pk[i]];
      s1[2 * i + 1] = m1[2 * pk[i] + 1];
      s2[2 * i] = m2[2 * pk[i]];
      s2[2 * i + 1] = m2[2 * pk[i] + 1];
    }
    return !(last_point_collinear(s1, 7) || last_point_collinear(s2, 7));
  }
  uint32_t score(IterResult& r, std::vector<uint8_t>& scratch) const {
    if (lmeds) {  // LMeDSPointSetRegistrator::run: the median of computeError's values per model
      const float *m1 = pts.data(), *m2 = m1 + 2 * (size_t)count;
      float errs[16];
      for (int k = 0; k < r.nm; k++) {
        epipolar_errors(m1, m2, count, r.models + 9 * k, errs);
        std::nth_element(errs, errs + count / 2, errs + count);
        std::memcpy(&r.good[k], &errs[count / 2], sizeof(float));
      }
      return 0;
    }
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
  // how long an idle helper keeps polling before it blocks (ESVIO_FE_HELPER_SPIN_US; default
  // kIdleSpinUs: longer than the gap between two published frames, so helpers never sleep while a
  // tracker is busy).  With more threads than CPUs (N ranks sharing a small quota) a few tens of
  // microseconds is right: the helpers block between jobs and the wake-up at the start of a
  // published frame's call (ransac_pool_wake) has them back before the RANSAC begins.
  int idle_spin_us = kIdleSpinUs;
  // (ransac_pool_set_idle_work) what a spinning helper takes when no job is open
  // `idle_pending` points into memory that outlives the pool (the handle): polled without any guard;
  // only a helper that sees work there enters the guarded section and looks at the function
  std::atomic<const std::atomic<int>*> idle_pending{nullptr};
  std::atomic<bool (*)(void*)> idle_fn{nullptr};
  void* idle_arg = nullptr;
  alignas(64) std::atomic<int> idle_inside{0};  // helpers inside idle_fn right now
  std::atomic<uint32_t> idle_gen{0};  // bumped when idle_fn is set: every awake helper calls it once at once (per-thread set-up)
  alignas(64) std::atomic<uint32_t> epoch{0};  // odd while a job is open; job (epoch >> 1) & 1
  RansacJob job[2];
  // helpers are kept on the cores that share the caller's L3 (one CCD): an iteration is ~0.5 us of
  // work, so the hand-over has to cost a same-die cache line transfer, not a cross-socket one
  std::vector<int> near_cpus;  // the caller's L3 domain minus the caller's own core (empty: unknown)
  std::vector<int> l3_cpus;    // the whole domain ...
  std::vector<int> l3_core;    // ... and the core (its first hardware thread) each of them belongs to
  int near_of = -1;            // the CPU the calling thread was last seen on
  uint64_t repins = 0;         // times the helpers had to be moved out of the calling thread's way

  void pin_near_caller();

  void helper() {
    std::vector<uint8_t> scratch;
    uint32_t seen = 0;
    auto idle_since = std::chrono::steady_clock::now();
    unsigned spins = 0;
    uint32_t idle_gen_seen = 0;
    [[maybe_unused]] long last_nivcsw = 0;
    for (;;) {
      if (quit.load(std::memory_order_acquire)) return;
      const uint32_t e = epoch.load(std::memory_order_acquire);
      if (!(e & 1) || e == seen) {
        const std::atomic<int>* pend = idle_pending.load(std::memory_order_acquire);
        const uint32_t ig = idle_gen.load(std::memory_order_acquire);
        const bool first = ig != idle_gen_seen;
        idle_gen_seen = ig;
        if (pend && (first || pend->load(std::memory_order_relaxed) > 0)) {
          idle_inside.fetch_add(1, std::memory_order_seq_cst);
          bool (*fn)(void*) = idle_fn.load(std::memory_order_seq_cst);  // (still there?)
          const bool did = fn && fn(idle_arg);
          idle_inside.fetch_sub(1, std::memory_order_acq_rel);
          if (did) {
            idle_since = std::chrono::steady_clock::now();
            continue;
          }
        }
        cpu_relax();
        if ((++spins & 1023) == 0 &&
            std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(idle_spin_us)) {
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
        // (subsets are published in groups of kLanes starting at multiples of kLanes; only the last
        // group of a job can be shorter)
        const int idx = J.next.fetch_add(kLanes, std::memory_order_relaxed);
        if (idx >= kRansacMaxIters) break;
        bool go = true;
        int avail;
        while ((avail = J.avail.load(std::memory_order_acquire)) <= idx) {
          if (epoch.load(std::memory_order_acquire) != e) {
            go = false;
            break;
          }
          cpu_relax();
        }
        if (!go) break;
        J.evaluate_lanes(idx, std::min(kLanes, avail - idx), e, scratch);
      }
      J.active.fetch_sub(1, std::memory_order_release);
      seen = e;
      idle_since = std::chrono::steady_clock::now();
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
      {  // (esvio_fe_ransac_tail: how often the helpers lose their CPU — a syscall per job, off the caller's path)
        struct rusage ru;
        if (getrusage(RUSAGE_THREAD, &ru) == 0) {
          if (ru.ru_nivcsw > last_nivcsw) g_rs_helper_sw.fetch_add((uint64_t)(ru.ru_nivcsw - last_nivcsw), std::memory_order_relaxed);
          last_nivcsw = ru.ru_nivcsw;
        }
      }
#endif
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
  if (cpu < 0 || cpu == near_of) return;
  if (near_of >= 0) {
    // The calling thread has moved.  Inside the die that is fine as long as it has not landed on a core
    // one of the helpers is pinned to: a spinning helper and the caller on one core take turns by the
    // scheduler's tick — milliseconds — until the load balancer moves the caller again (the helpers
    // cannot move: their set has exactly one CPU per helper when the domain has 8 cores).
    bool in_domain = false, on_helper_core = false;
    for (size_t i = 0; i < l3_cpus.size(); i++)
      if (l3_cpus[i] == cpu) {
        in_domain = true;
        for (int v : near_cpus) on_helper_core = on_helper_core || v == l3_core[i];
      }
    if (in_domain && !on_helper_core) {
      near_of = cpu;
      return;
    }
    repins++;
  }
  char path[128];
  std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  std::vector<int> l3 = read_cpu_list(path);
  std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
  const std::vector<int> self = read_cpu_list(path);
  near_of = cpu;
  l3_cpus = l3;
  l3_core.assign(l3.size(), -1);
  near_cpus.clear();
  // only CPUs this process may run on (N ranks on one node: bench.py gives every rank its own block of
  // cores; a helper pinned to the whole L3 domain would sit on the neighbours' cores)
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  const bool have_allowed = sched_getaffinity(0, sizeof allowed, &allowed) == 0;
  std::vector<int> primary;  // one logical CPU per physical core of the domain, caller's core left out
  std::vector<int> all;
  for (size_t i = 0; i < l3.size(); i++) {
    const int v = l3[i];
    std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", v);
    const std::vector<int> sib = read_cpu_list(path);
    l3_core[i] = sib.empty() ? v : sib[0];  // (a core is named by its first hardware thread)
    if (have_allowed && (v >= CPU_SETSIZE || !CPU_ISSET(v, &allowed))) continue;
    bool mine = v == cpu;
    for (int w : self) mine = mine || v == w;
    if (mine) continue;
    all.push_back(v);
    if (sib.empty() || sib[0] == v) primary.push_back(v);
  }
  // a spinning helper on the SMT sibling of another helper (or of the caller) takes issue slots
  // from it: keep to one hardware thread per core when the domain has enough cores
  near_cpus = primary.size() >= th.size() ? primary : all;
  if (near_cpus.size() < th.size()) {  // unknown topology or a tiny L3 domain: leave it to the OS
    near_cpus.clear();
    cpu_set_t any;
    CPU_ZERO(&any);
    if (have_allowed) any = allowed;
    if (have_allowed && repins)  // (they were pinned to another die before)
      for (auto& t : th) (void)pthread_setaffinity_np(t.native_handle(), sizeof any, &any);
    return;
  }
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int v : near_cpus)
    if (v < CPU_SETSIZE) CPU_SET(v, &set);
  for (auto& t : th) (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
#endif
}

namespace {
RansacPool* ransac_pool_alloc() {
  RansacPool* p = new RansacPool();
  for (RansacJob& J : p->job) {
    J.picks.resize((size_t)kRansacMaxIters * 8);
    J.res.reset(new IterResult[kRansacMaxIters]);
    J.summary.reset(new std::atomic<uint64_t>[kRansacMaxIters]);
    for (int i = 0; i < kRansacMaxIters; i++) J.summary[i].store(0, std::memory_order_relaxed);
  }
  return p;
}
}  // namespace

RansacPool* ransac_pool_create(int helpers) {
  if (helpers <= 0) return nullptr;
  RansacPool* p = ransac_pool_alloc();
  if (const char* v = getenv("ESVIO_FE_HELPER_SPIN_US")) p->idle_spin_us = std::max(0, atoi(v));
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

void ransac_pool_set_idle_work(RansacPool* p, const std::atomic<int>* pending, bool (*fn)(void*), void* arg) {
  if (!p) return;
  // store-then-load against the helpers' add-then-load: both sides sequentially consistent, or the setter's load
  // could pass its own buffered store and miss a helper that still reads the old hook
  p->idle_fn.store(nullptr, std::memory_order_seq_cst);
  while (p->idle_inside.load(std::memory_order_seq_cst) != 0) cpu_relax();  // (nobody is left inside the old one)
  p->idle_pending.store(nullptr, std::memory_order_release);
  if (!pending || !fn) return;
  p->idle_arg = arg;
  p->idle_pending.store(pending, std::memory_order_release);  // (the handle's counter: outlives the pool)
  p->idle_fn.store(fn, std::memory_order_release);
  p->idle_gen.fetch_add(1, std::memory_order_acq_rel);
}

void ransac_pool_hold(RansacPool* p, int mask, bool on) {
  if (!p) return;
  for (int b = 0; b < 2; b++)
    if (mask & (1 << b)) p->job[b].active.fetch_add(on ? 1 : -1, std::memory_order_acq_rel);
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


void ransac_tail(uint64_t out6[6], bool reset) {
  out6[0] = g_rs_max_ns.load();
  out6[1] = g_rs_lm_max_ns.load();
  out6[2] = g_rs_redone.load();
  out6[3] = g_rs_solo.load();
  out6[4] = g_rs_skipped.load();
  out6[5] = g_rs_helper_sw.load();
  if (reset) {
    g_rs_skipped = 0;
    g_rs_helper_sw = 0;
    g_rs_max_ns = 0;
    g_rs_lm_max_ns = 0;
    g_rs_redone = 0;
    g_rs_solo = 0;
  }
}

void host_hypot(const double* x, const double* y, int n, double* out) {
  for (int i = 0; i < n; i++) out[i] = cv_hypot(x[i], y[i]);
}

int host_nullspace(const double* A, int n, int lanes, double* f) {
  if (!lanes) {
    for (int i = 0; i < n; i++)
      epipolar_nullspace((const double(*)[9])(A + (size_t)i * 63), f + (size_t)i * 18, f + (size_t)i * 18 + 9);
    return 0;
  }
  int redone = 0;
  for (int i0 = 0; i0 < n; i0 += kLanes) {
    const int cnt = std::min(kLanes, n - i0);
    double Al[kLanes][7][9], f1[kLanes][9], f2[kLanes][9];
    for (int l = 0; l < kLanes; l++) std::memcpy(Al[l], A + (size_t)(i0 + (l < cnt ? l : 0)) * 63, sizeof(Al[l]));
    if (!epipolar_nullspace_lanes(Al, f1, f2)) {  // (what seven_point_lanes does with such a group)
      for (int l = 0; l < cnt; l++) epipolar_nullspace(Al[l], f1[l], f2[l]);
      redone += cnt;
    }
    for (int l = 0; l < cnt; l++) {
      std::memcpy(f + (size_t)(i0 + l) * 18, f1[l], sizeof(f1[l]));
      std::memcpy(f + (size_t)(i0 + l) * 18 + 9, f2[l], sizeof(f2[l]));
    }
  }
  return redone;
}

RansacStats ransac_stats(bool reset) {
  RansacStats r{g_rs_calls.load(), g_rs_iters.load(), g_rs_points.load(), g_rs_ns.load(),
                g_rs_lm_calls.load(), g_rs_lm_ns.load()};
  if (reset) {
    g_rs_lm_calls = 0;
    g_rs_lm_ns = 0;
    g_rs_calls = 0;
    g_rs_iters = 0;
    g_rs_points = 0;
    g_rs_ns = 0;
    g_rs_max_ns = 0;
    g_rs_lm_max_ns = 0;
    g_rs_redone = 0;
    g_rs_solo = 0;
    g_rs_skipped = 0;
    g_rs_helper_sw = 0;
  }
  return r;
}

namespace {
// the RANSAC loop of find_fundamental_mat with helpers (count >= 15)
int ransac_pooled(RansacPool* P, const float* m1, const float* m2, int count, double thr, double conf,
                  const double* xy, uint8_t* status, int* iterations) {
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
  J.lmeds = false;
  J.next.store(0, std::memory_order_relaxed);
  J.avail.store(0, std::memory_order_relaxed);
  J.bound.store(kModelPoints - 1, std::memory_order_relaxed);
  P->epoch.store(e, std::memory_order_release);  // open
  if (!P->th.empty()) ransac_pool_wake(P);

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
        g_rs_redone.fetch_add(1, std::memory_order_relaxed);
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
      for (int j = 0; j < kLanes && drawn < niters; j++) {  // RANSACPointSetRegistrator::getSubset's draws
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
    if (idx < drawn && J.next.compare_exchange_strong(idx, idx + kLanes, std::memory_order_relaxed)) {
      J.evaluate_lanes(idx, std::min(kLanes, drawn - idx), e, scratch);
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
  *iterations = niters;
  if (have_best) score_block(x1, y1, x2, y2, 0, count, best_F, (float)(thr * thr), status);
  return best_good;
}

// LMeDSPointSetRegistrator::run for 8..14 points: a fixed number of hypotheses (300 at confidence
// 0.99), the one with the smallest median error wins (the first one on ties), inliers within
// 2.5 * 1.4826 * (1 + 5/(n - 7)) * sqrt(median).  The subsets do not depend on the models, so they
// are all drawn first — getSubset's own loop, collinearity retries included — and then solved
// kLanes at a time by the calling thread and the pool's helpers; the medians are compared in
// iteration order.
int lmeds_pooled(RansacPool* P, const float* m1, const float* m2, int count, double conf, uint8_t* status) {
  const int kModelPoints = 7;
  const int niters = std::max(update_num_iters(conf, 0.45, kModelPoints, kRansacMaxIters), 3);
  const uint32_t e = P->epoch.load(std::memory_order_relaxed) + 1;
  RansacJob& J = P->job[(e >> 1) & 1];
  while (J.active.load(std::memory_order_acquire) != 0) cpu_relax();
  OcvRng rng;
  float s1[14], s2[14];
  int nsub = 0;
  for (; nsub < niters; nsub++)  // (getSubset's default of 1000 attempts; the RANSAC loop passes 10000)
    if (!draw_subset(m1, m2, count, s1, s2, rng, 1000, &J.picks[(size_t)nsub * 8])) break;
  if (nsub == 0) return 0;
  J.pts.resize(4 * (size_t)count);
  std::memcpy(J.pts.data(), m1, 2 * (size_t)count * sizeof(float));
  std::memcpy(J.pts.data() + 2 * (size_t)count, m2, 2 * (size_t)count * sizeof(float));
  J.count = count;
  J.lmeds = true;
  J.next.store(0, std::memory_order_relaxed);
  J.avail.store(nsub, std::memory_order_relaxed);
  P->epoch.store(e, std::memory_order_release);  // open
  if (!P->th.empty()) ransac_pool_wake(P);
  std::vector<uint8_t> scratch;
  IterResult local;
  double best[9], min_median = DBL_MAX;
  unsigned stalled = 0;
  for (int rp = 0; rp < nsub;) {
    const IterResult* r = &J.res[rp];
    const uint64_t sm = J.summary[rp].load(std::memory_order_acquire);
    if ((uint32_t)(sm >> 32) != e) {
      int idx = J.next.load(std::memory_order_relaxed);
      if (idx < nsub && J.next.compare_exchange_strong(idx, idx + kLanes, std::memory_order_relaxed)) {
        J.evaluate_lanes(idx, std::min(kLanes, nsub - idx), e, scratch);
        continue;
      }
      if (++stalled <= 64) {
        cpu_relax();
        continue;
      }
      J.evaluate(rp, local, scratch);  // (its taker is not delivering)
      r = &local;
      g_rs_redone.fetch_add(1, std::memory_order_relaxed);
    }
    stalled = 0;
    for (int k = 0; k < r->nm; k++) {
      float med;
      std::memcpy(&med, &r->good[k], sizeof(float));
      if ((double)med < min_median) {
        min_median = med;
        std::memcpy(best, r->models + 9 * k, sizeof(best));
      }
    }
    rp++;
  }
  P->epoch.store(e + 1, std::memory_order_release);  // closed
  if (min_median < DBL_MAX) {
    double sigma = 2.5 * 1.4826 * (1 + 5. / (count - kModelPoints)) * std::sqrt(min_median);
    sigma = std::max(sigma, 0.001);
    std::vector<float> err(count);
    return mark_inliers(m1, m2, count, best, err, status, sigma);
  }
  return 0;
}
}  // namespace

int find_fundamental_mat(const float* m1, const float* m2, int count, double thr, double conf,
                         uint8_t* status, RansacPool* pool) {
  std::fill(status, status + count, (uint8_t)0);
  if (count < 7) return 0;
  if (thr <= 0) thr = 3;
  if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
  if (count == 7) {
    double models[27];
    const int n = seven_point(m1, m2, models);
    std::fill(status, status + count, (uint8_t)1);
    return n > 0 ? count : 0;
  }
  // without helper threads the same loops run on a pool of none: the calling thread draws, solves the
  // subsets kLanes at a time and replays them in order
  static thread_local std::unique_ptr<RansacPool> solo;
  if (pool) {
    // This job's buffer (two alternate) may still hold a helper that was descheduled in the middle of
    // the job before last.  Waiting for it costs a scheduler quantum — the 1.8 ms RANSAC calls and 2-7 ms
    // steps of round 3's cold runs — so after a few microseconds the job takes the other buffer (the
    // epoch moves on by two: the held helper finds a foreign epoch when it comes back and leaves), and
    // if that one is held as well it runs without the helpers (same result, by construction).
    auto held = [&](uint32_t e) { return pool->job[(e >> 1) & 1].active.load(std::memory_order_acquire) != 0; };
    const uint32_t e = pool->epoch.load(std::memory_order_relaxed) + 1;
    int spins = 0;
    while (held(e) && ++spins < 256) cpu_relax();
    if (held(e)) {
      if (!held(e + 2)) {
        pool->epoch.store(e + 1, std::memory_order_release);  // (even: no job open; the next one is e + 2)
        g_rs_skipped.fetch_add(1, std::memory_order_relaxed);
      } else {
        g_rs_solo.fetch_add(1, std::memory_order_relaxed);
        pool = nullptr;
      }
    }
  }
  if (!pool) {
    if (!solo) solo.reset(ransac_pool_alloc());
    pool = solo.get();
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
    const auto t0 = std::chrono::steady_clock::now();
    int iters = 0;
    const int good = ransac_pooled(pool, m1, m2, count, thr, conf, xy.data(), status, &iters);
    g_rs_calls.fetch_add(1, std::memory_order_relaxed);
    g_rs_iters.fetch_add((uint64_t)iters, std::memory_order_relaxed);
    g_rs_points.fetch_add((uint64_t)count, std::memory_order_relaxed);
    const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                            std::chrono::steady_clock::now() - t0).count();
    g_rs_ns.fetch_add(ns, std::memory_order_relaxed);
    atomic_max(g_rs_max_ns, ns);
    return good;
  }
  // LMedS for 8..14 points
  const auto t0 = std::chrono::steady_clock::now();
  const int good = lmeds_pooled(pool, m1, m2, count, conf, status);
  g_rs_lm_calls.fetch_add(1, std::memory_order_relaxed);
  const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::steady_clock::now() - t0).count();
  g_rs_lm_ns.fetch_add(ns, std::memory_order_relaxed);
  atomic_max(g_rs_lm_max_ns, ns);
  return good;
}

}  // namespace host
}  // namespace esvio
