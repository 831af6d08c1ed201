// ransac_bench.cpp — host-only timing of rejectWithF_event's RANSAC with 0..N helper threads
//   g++ -O3 -std=c++17 -ffp-contract=off -fno-math-errno -pthread tools/ransac_bench.cpp esvio_amd/csrc/fe_host.cpp -Iinclude -o /tmp/ransac_bench
//   ransac_bench [points [outlier fraction [repetitions]]]   (8..14 points take the LMedS branch)
// tests/test_host_sanitizers.py builds it with -fsanitize=thread and -fsanitize=address,undefined.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../esvio_amd/csrc/fe_host.h"

using namespace esvio::host;

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 160;
  const double outl = argc > 2 ? atof(argv[2]) : 0.37;
  const int reps = argc > 3 ? atoi(argv[3]) : 2000;
  std::mt19937 g(5);
  std::uniform_real_distribution<double> U(-1, 1);
  std::normal_distribution<double> N(0, 1);
  std::vector<std::vector<float>> P1, P2;
  for (int s = 0; s < 16; s++) {
    std::vector<float> p1(2 * n), p2(2 * n);
    const double t[3] = {0.05 * N(g), 0.05 * N(g), 0.05 * N(g)};
    for (int i = 0; i < n; i++) {
      const double X = 2 * U(g), Y = 1.5 * U(g), Z = 4 + U(g);
      p1[2 * i] = (float)(460 * X / Z + 320);
      p1[2 * i + 1] = (float)(460 * Y / Z + 240);
      p2[2 * i] = (float)(460 * (X + t[0]) / (Z + t[2]) + 320 + 0.05 * N(g));
      p2[2 * i + 1] = (float)(460 * (Y + t[1]) / (Z + t[2]) + 240 + 0.05 * N(g));
      if (i < outl * n) {
        p2[2 * i] += (float)(8 * N(g));
        p2[2 * i + 1] += (float)(8 * N(g));
      }
    }
    P1.push_back(p1);
    P2.push_back(p2);
  }
  std::vector<uint8_t> ref(n), st(n);
  for (int helpers : {0, 1, 2, 3, 5, 7}) {
    RansacPool* pool = ransac_pool_create(helpers);
    int bad = 0;
    long inl = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) {
      const int s = r & 15;
      inl += find_fundamental_mat(P1[s].data(), P2[s].data(), n, 1.0, 0.99, st.data(), pool);
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    for (int s = 0; s < 16; s++) {
      find_fundamental_mat(P1[s].data(), P2[s].data(), n, 1.0, 0.99, ref.data(), nullptr);
      find_fundamental_mat(P1[s].data(), P2[s].data(), n, 1.0, 0.99, st.data(), pool);
      bad += memcmp(ref.data(), st.data(), n) != 0;
    }
    printf("helpers=%d  %.1f us/call  mean inliers %.1f  mismatches %d\n", helpers, us, (double)inl / reps, bad);
    ransac_pool_destroy(pool);
  }
  return 0;
}
