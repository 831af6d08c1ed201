// fe_mc.h — IMU motion compensation of event coordinates (reference:
// feature_tracker/src/event_detector/event_detector.cc:102-147 createSAE_left with
// Motion_correction_value, :547-591 motioncorrection).  Single-precision 3x3 arithmetic in the
// evaluation order of Eigen's fixed-size kernels as recalled [upstream-Eigen 3.3, unpinned]:
// coefficient products reduced as a0 + (a1 + a2), cofactor inverse, Matrix3f::exp() as Pade 3/5/7 +
// partial-pivot LU solve + squarings.  Shared by the key kernel (device) and the handle (host: K, K^-1).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace esvio {

struct M3f {
  float m[3][3];
};

struct McParams {
  int enabled;       // this batch goes through the motion-compensation overload
  int active;        // |accel| > 5 m/s^2 (event_detector.cc:125): the warp is applied
  double t1;         // event_left.header.stamp (feature_tracker.cpp:622); the kernels read t0, the first
                     // LEFT event's time (:621), from the batch itself: dt = t1 - t0 (:623)
  float vsum[3];     // tmp_v + tmp_v_pre
  float omega[3];
  M3f K, Kinv;
};

#define ESVIO_HD __host__ __device__ __forceinline__

ESVIO_HD float mc_red3(float a0, float a1, float a2) { return a0 + (a1 + a2); }

ESVIO_HD M3f mc_mul(const M3f& A, const M3f& B) {
  M3f C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C.m[i][j] = mc_red3(A.m[i][0] * B.m[0][j], A.m[i][1] * B.m[1][j], A.m[i][2] * B.m[2][j]);
  return C;
}

ESVIO_HD float mc_cof(const M3f& M, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return M.m[i1][j1] * M.m[i2][j2] - M.m[i1][j2] * M.m[i2][j1];
}

ESVIO_HD M3f mc_inverse(const M3f& M) {
  const float c0 = mc_cof(M, 0, 0), c1 = mc_cof(M, 1, 0), c2 = mc_cof(M, 2, 0);
  const float det = mc_red3(c0 * M.m[0][0], c1 * M.m[1][0], c2 * M.m[2][0]);
  const float invdet = 1.0f / det;
  M3f R;
  R.m[0][0] = c0 * invdet;
  R.m[0][1] = c1 * invdet;
  R.m[0][2] = c2 * invdet;
  for (int r = 1; r < 3; r++)
    for (int k = 0; k < 3; k++) R.m[r][k] = mc_cof(M, k, r) * invdet;
  return R;
}

// denom.partialPivLu().solve(numer)
ESVIO_HD M3f mc_lu_solve(M3f LU, M3f X) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int piv = k;
    float best = fabsf(LU.m[k][k]);
#pragma unroll
    for (int i = k + 1; i < 3; i++)
      if (fabsf(LU.m[i][k]) > best) {
        best = fabsf(LU.m[i][k]);
        piv = i;
      }
    // row k <-> row piv, written with compile-time row numbers (a run-time row index would put the
    // matrices into scratch / LDS on the device: 30 us for the 0.33 M events of a C3 batch)
#pragma unroll
    for (int r = k + 1; r < 3; r++)
      if (piv == r) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
          float t = LU.m[k][j];
          LU.m[k][j] = LU.m[r][j];
          LU.m[r][j] = t;
          t = X.m[k][j];
          X.m[k][j] = X.m[r][j];
          X.m[r][j] = t;
        }
      }
    if (best != 0.0f)
      for (int i = k + 1; i < 3; i++) LU.m[i][k] /= LU.m[k][k];
    for (int i = k + 1; i < 3; i++)
      for (int j = k + 1; j < 3; j++) LU.m[i][j] -= LU.m[i][k] * LU.m[k][j];
  }
  for (int c = 0; c < 3; c++) {
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

// MatrixBase<Matrix3f>::exp() (unsupported/Eigen/MatrixFunctions, float specialisation)
ESVIO_HD M3f mc_exp(const M3f& arg) {
  float l1 = 0;
  for (int j = 0; j < 3; j++) {
    const float cs = mc_red3(fabsf(arg.m[0][j]), fabsf(arg.m[1][j]), fabsf(arg.m[2][j]));
    if (j == 0 || cs > l1) l1 = cs;
  }
  M3f A = arg, U, V;
  int squarings = 0;
  if (l1 < 4.258730016922831e-001f) {
    const M3f A2 = mc_mul(A, A);
    M3f tmp;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const float id = i == j ? 1.f : 0.f;
        tmp.m[i][j] = 1.f * A2.m[i][j] + 60.f * id;
        V.m[i][j] = 12.f * A2.m[i][j] + 120.f * id;
      }
    U = mc_mul(A, tmp);
  } else if (l1 < 1.880152677804762e+000f) {
    const M3f A2 = mc_mul(A, A), A4 = mc_mul(A2, A2);
    M3f tmp;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const float id = i == j ? 1.f : 0.f;
        tmp.m[i][j] = (1.f * A4.m[i][j] + 420.f * A2.m[i][j]) + 15120.f * id;
        V.m[i][j] = (30.f * A4.m[i][j] + 3360.f * A2.m[i][j]) + 30240.f * id;
      }
    U = mc_mul(A, tmp);
  } else {
    const float maxnorm = 3.925724783138660f;
    (void)frexpf(l1 / maxnorm, &squarings);
    if (squarings < 0) squarings = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A.m[i][j] = ldexpf(arg.m[i][j], -squarings);
    const M3f A2 = mc_mul(A, A), A4 = mc_mul(A2, A2), A6 = mc_mul(A4, A2);
    M3f tmp;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const float id = i == j ? 1.f : 0.f;
        tmp.m[i][j] = ((1.f * A6.m[i][j] + 1512.f * A4.m[i][j]) + 277200.f * A2.m[i][j]) + 8648640.f * id;
        V.m[i][j] = ((56.f * A6.m[i][j] + 25200.f * A4.m[i][j]) + 1995840.f * A2.m[i][j]) + 17297280.f * id;
      }
    U = mc_mul(A, tmp);
  }
  M3f numer, denom;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      numer.m[i][j] = U.m[i][j] + V.m[i][j];
      denom.m[i][j] = -U.m[i][j] + V.m[i][j];
    }
  M3f R = mc_lu_solve(denom, numer);
  for (int i = 0; i < squarings; i++) R = mc_mul(R, R);
  return R;
}

// motioncorrection(ex, ey, v, v_pre, accel, omega, dt) -> pixel the event is written to
ESVIO_HD void mc_warp(const McParams& p, int W, int H, int ex_i, int ey_i, double dt, int* ox,
                      int* oy) {
  const double ex = ex_i, ey = ey_i;
  const int kBorder = 6;
  *ox = ex_i;
  *oy = ey_i;
  if (ex > kBorder && ex <= (W - kBorder) && ey > kBorder && ey <= (H - kBorder)) {
    const float fdt = (float)dt;
    const float rx = p.omega[0] * fdt, ry = p.omega[1] * fdt, rz = p.omega[2] * fdt;
    M3f skew;
    skew.m[0][0] = 0;   skew.m[0][1] = -rz; skew.m[0][2] = ry;
    skew.m[1][0] = rz;  skew.m[1][1] = 0;   skew.m[1][2] = -rx;
    skew.m[2][0] = -ry; skew.m[2][1] = rx;  skew.m[2][2] = 0;
    const M3f R = mc_exp(skew);
    M3f Rt;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rt.m[i][j] = R.m[j][i];
    const M3f rot_K = mc_mul(mc_mul(p.K, Rt), p.Kinv);
    const float h = (float)(0.5 * dt);
    float tk[3], kt[3], tr[3], w[3];
    for (int i = 0; i < 3; i++) tk[i] = h * p.vsum[i];
    for (int i = 0; i < 3; i++)
      kt[i] = mc_red3(p.Kinv.m[i][0] * tk[0], p.Kinv.m[i][1] * tk[1], p.Kinv.m[i][2] * tk[2]);
    for (int i = 0; i < 3; i++)
      tr[i] = mc_red3((-rot_K.m[i][0]) * kt[0], (-rot_K.m[i][1]) * kt[1], (-rot_K.m[i][2]) * kt[2]);
    const float ev[3] = {(float)ex, (float)ey, 1.f};
    for (int i = 0; i < 3; i++)
      w[i] = mc_red3(rot_K.m[i][0] * ev[0], rot_K.m[i][1] * ev[1], rot_K.m[i][2] * ev[2]) + tr[i];
    w[0] = w[0] / w[2];
    w[1] = w[1] / w[2];
    const int xc = (int)floorf(w[0]), yc = (int)floorf(w[1]);
    if (xc > 0 && xc < W - 1 && yc > 0 && yc < H - 1) {
      *ox = xc;
      *oy = yc;
    }
  }
}

}  // namespace esvio
