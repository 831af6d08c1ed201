// jacobi_cap_check.cpp — the lane form of run7Point's null space against the one-at-a-time routine with
// the Jacobi sweep limit lowered until it bites (-DESVIO_JACOBI_MAX_SWEEPS=2..4; OpenCV's 30 is never
// reached by real systems): the lane form runs levels of the NEXT sweep alongside the last levels of the
// current one, so "the current sweep was the last one allowed" is a path of its own.
// Built and run by tests/test_ransac_nullspace.py (host only, no GPU).
#include "../esvio_amd/csrc/fe_host.cpp"

#include <cstdio>
#include <random>

int main() {
  using namespace esvio::host;
  std::mt19937 g(7);
  std::uniform_real_distribution<double> U(0, 1);
  std::normal_distribution<double> N(0, 8);
  const int n = 2003;
  std::vector<double> A((size_t)n * 63), one((size_t)n * 18), lanes((size_t)n * 18);
  for (int s = 0; s < n; s++)
    for (int i = 0; i < 7; i++) {
      const double x0 = 640 * U(g), y0 = 480 * U(g), x1 = x0 + N(g), y1 = y0 + N(g);
      const double row[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1};
      std::memcpy(&A[(size_t)s * 63 + i * 9], row, sizeof(row));
    }
  host_nullspace(A.data(), n, 0, one.data());
  const int redone = host_nullspace(A.data(), n, 1, lanes.data());
  const bool same = std::memcmp(one.data(), lanes.data(), one.size() * sizeof(double)) == 0;
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the basis' bytes: compared across builds for other vector widths
  const unsigned char* bytes = (const unsigned char*)lanes.data();
  for (size_t i = 0; i < lanes.size() * sizeof(double); i++) h = (h ^ bytes[i]) * 1099511628211ull;
  std::printf("sweep limit %d: %s (%d systems redone one at a time) basis %016llx\n", kJacobiMaxSweeps,
              same ? "identical" : "DIFFERENT", redone, (unsigned long long)h);
  return same && redone == 0 ? 0 : 1;
}
