// score_check.cpp — the division-free inlier test of the RANSAC scoring (score_block) against the
// reference's formula (score_block_exact: computeError + findInliers, element for element) on random
// two-view points and on points placed within 1e-12 .. 1e-3 (relative) of the threshold, where the
// bounds do not decide and the block must fall back to the exact form; also degenerate models (zero
// rows: 1 / 0 in the reference's formula).  Built and run by tests/test_ransac_kat.py (host only).
#include "../esvio_amd/csrc/fe_host.cpp"

#include <cstdio>
#include <random>

int main() {
  using namespace esvio::host;
  std::mt19937_64 g(11);
  std::uniform_real_distribution<double> U(-1, 1);
  const int n = 64;
  std::vector<double> x1(n), y1(n), x2(n), y2(n);
  std::vector<uint8_t> a(n), b(n);
  long cases = 0, differing = 0, near = 0, inl = 0, fast_blocks = 0, blocks = 0;
  for (int trial = 0; trial < 40000; trial++) {
    double F[9];
    for (double& f : F) f = U(g) * std::pow(10.0, 3 * U(g));
    F[8] = 1;
    const int kind = trial % 5;
    if (kind == 3) F[0] = F[1] = F[3] = F[4] = 0;          // epipolar lines of constant direction
    if (kind == 4 && trial % 25 == 4) F[0] = F[1] = F[2] = F[3] = F[4] = F[5] = 0;  // a = b = 0 for image 2: 1 / 0
    const float t = (float)std::pow(10.0, kind == 2 ? 2 * U(g) : 0.0);  // thr^2 (1.0 as shipped, or 0.01 .. 100)
    for (int i = 0; i < n; i++) {
      x1[i] = (float)(320 + 320 * U(g));
      y1[i] = (float)(240 + 240 * U(g));
      // the epipolar line of point 1 in image 2, and a point 2 at a chosen distance from it
      const double la = F[0] * x1[i] + F[1] * y1[i] + F[2], lb = F[3] * x1[i] + F[4] * y1[i] + F[5],
                   lc = F[6] * x1[i] + F[7] * y1[i] + F[8];
      const double nn = std::sqrt(la * la + lb * lb);
      double px = 320 + 320 * U(g), py = 240 + 240 * U(g);
      if (nn > 0 && i % 2 == 0 && trial % 2) {  // (even trials: random points only — the bounds decide)
        const double d0 = (la * px + lb * py + lc) / nn;
        const double rel = std::pow(10.0, -12 + 9 * (double)(i % 32) / 32) * (i % 4 ? 1 : -1);
        const double want = std::sqrt((double)t) * (1 + rel) * (i % 8 < 4 ? 1 : -1);
        px += (want - d0) * la / nn;
        py += (want - d0) * lb / nn;
        near++;
      }
      x2[i] = (i % 16 == 1) ? (double)(float)px : px;  // (mostly not float-rounded: stay close to the threshold)
      y2[i] = (i % 16 == 1) ? (double)(float)py : py;
    }
    const int ga = score_block_exact(x1.data(), y1.data(), x2.data(), y2.data(), 0, n, F, t, a.data());
    const uint64_t fb = g_score_fallbacks.load();
    const int gb = score_block(x1.data(), y1.data(), x2.data(), y2.data(), 0, n, F, t, b.data());
    fast_blocks += g_score_fallbacks.load() == fb;
    blocks++;
    cases += n;
    inl += ga;
    if (ga != gb || std::memcmp(a.data(), b.data(), n)) differing++;
  }
  std::printf("%ld points (%ld placed next to the threshold), %ld inliers, %ld of %ld blocks decided by the bounds alone, "
              "blocks with a differing flag: %ld\n", cases, near, inl, fast_blocks, blocks, differing);
  return differing == 0 && inl > cases / 50 && fast_blocks > blocks / 4 && blocks - fast_blocks > blocks / 4 ? 0 : 1;
}
