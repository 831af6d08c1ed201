// Pin for the one OpenCV step no Python binding reaches: the cv::MatExpr of event_detector.cc:257-259,
//     time_surface_map = 255.0 * (time_surface_map + 1.0) / 2.0;   time_surface_map.convertTo(time_surface_map, CV_8U);
// The oracle (oracle/esvio_oracle.cpp sae_to_ts) and the GPU kernel restate it as ONE scale-and-shift in double,
//     v * 127.5 + 127.5, then saturate_cast<uchar>(cvRound(.)),
// i.e. they assume that OpenCV folds the expression's three scalar operations into a single convertTo(alpha, beta).
// This program holds that assumption against OpenCV itself on a dense grid of v in [-1, 1] (every value of
// +-exp(-k * 2^-22 / 0.02), the same ages tests/test_exp_sweep_gpu.py sweeps) and on the ignore_polarity form 255.0 * M.
// Self-checking: prints the number of differing bytes (0 = the restatement is pinned) and exits with 1 otherwise.
//     g++ -O2 cv_matexpr_pin.cpp $(pkg-config --cflags --libs opencv4) -o cv_matexpr_pin && ./cv_matexpr_pin
// (Against OpenCV 4.2.0, what ROS Noetic ships.)  Not built by this repository: the image has no OpenCV.
#include <cmath>
#include <cstdio>
#include <opencv2/core.hpp>

int main() {
  const int N = 1 << 21;
  long bad = 0;
  for (int form = 0; form < 3; form++) {  // 0: polarity +, 1: polarity -, 2: ignore_polarity
    cv::Mat m(1, N, CV_64FC1);
    for (int k = 0; k < N; k++) {
      const double e = std::exp(-(k * std::ldexp(1.0, -22)) / 0.02);
      m.at<double>(0, k) = form == 1 ? -e : e;
    }
    cv::Mat expr = form == 2 ? cv::Mat(255.0 * m) : cv::Mat(255.0 * (m + 1.0) / 2.0);
    cv::Mat u8;
    expr.convertTo(u8, CV_8U);
    for (int k = 0; k < N; k++) {
      const double v = m.at<double>(0, k);
      const double folded = form == 2 ? v * 255.0 : v * 127.5 + 127.5;
      const int want = cv::saturate_cast<uchar>(cvRound(folded));
      if (u8.at<uchar>(0, k) != want) {
        if (bad < 10) std::printf("form %d k %d: OpenCV %d, folded %d\n", form, k, (int)u8.at<uchar>(0, k), want);
        bad++;
      }
    }
  }
  std::printf("OpenCV %s: %ld of %d bytes differ from the folded form\n", CV_VERSION, bad, 3 * N);
  return bad ? 1 : 0;
}
