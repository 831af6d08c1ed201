// Driver of the ThreadSanitizer build of the library's host side (tests/hipstub, tests/test_host_tsan.py): the soak's
// toggling pattern over the C ABI — replay mode with batches announced up to four ahead from pageable memory (the
// staging helpers), the launch thread switched on and off mid-stream, 0..7 RANSAC helpers, lazy mode on and off,
// plain calls in between, calls that must fail (a published frame announced as unpublished), esvio_fe_reset with
// batches announced and in flight, handles created and destroyed.  The device is fake (fake_device.cpp): what is
// under test is every thread the library starts and every hand-over between them.
//   drive <seed> <frames>      exit 0: done; the sanitizer reports on stderr
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "esvio_fe.h"
#if __has_include("esvio_fe_test.h")  // (trees from before the header was split have the taps in esvio_fe.h)
#include "esvio_fe_test.h"
extern "C" void hipstub_arm_faults(int on);  // tests/hipstub/hip_stub.cpp
#endif

static uint32_t rs;
static uint32_t rnd() { return rs = rs * 1664525u + 1013904223u; }

struct Batch {
  std::vector<esvio_fe_event> L, R;
  double t;
};
static void make_batch(Batch& b, int W, int H, int frame, int n) {
  b.L.resize((size_t)n);
  b.R.resize((size_t)n * 3 / 4);
  const uint32_t sec = 1700000000u + (uint32_t)frame / 30u;
  for (auto* v : {&b.L, &b.R}) {
    uint32_t ns = (uint32_t)(frame % 30) * 33000000u;
    for (auto& e : *v) {
      std::memset(&e, 0, sizeof(e));
      e.x = (uint16_t)(rnd() % (uint32_t)W);
      e.y = (uint16_t)(rnd() % (uint32_t)H);
      ns += rnd() % 2000u;
      e.sec = sec;
      e.nsec = ns;
      e.polarity = (uint8_t)(rnd() & 1u);
    }
  }
  b.t = (double)sec + 1e-9 * (double)b.L.back().nsec;
}

int main(int argc, char** argv) {
  rs = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
  const int frames = argc > 2 ? atoi(argv[2]) : 200;
  const int W = 640, H = 480, M = 120;
  esvio_fe_config c;
  std::memset(&c, 0, sizeof(c));
  c.width = W; c.height = H; c.decay_ms = 20; c.feature_filter_threshold = 0.01; c.ts_lk_threshold = 128;
  c.max_cnt = M; c.min_dist = 10; c.flow_back = 1; c.f_threshold = 1.0; c.f_ransac = 1; c.lk_accum = 1;
  c.focal_length = 460; c.device = -1;
  for (int k = 0; k < 2; k++) { c.cam[k].fx = c.cam[k].fy = 0.9 * W; c.cam[k].cx = W / 2.0; c.cam[k].cy = H / 2.0; }
  std::vector<int32_t> ids(M), cnt(M), idr(M);
  std::vector<float> f2[6];
  for (auto& v : f2) v.resize(2 * M);
  esvio_fe_tracks t;
  std::memset(&t, 0, sizeof(t));
  t.ids = ids.data(); t.track_cnt = cnt.data(); t.cur_pts = f2[0].data(); t.cur_un_pts = f2[1].data();
  t.pts_velocity = f2[2].data(); t.ids_right = idr.data(); t.cur_right_pts = f2[3].data();
  t.cur_un_right_pts = f2[4].data(); t.right_pts_velocity = f2[5].data();
  long calls = 0, failed = 0, resets = 0, handles = 0;
  int f = 0;
  while (f < frames) {
    esvio_fe_handle h = nullptr;
    // (both LK modes, and the batches handed over as host or as "device" memory — the stub's device memory is the
    // host's: with the launch thread on, the float-order mode and device batches the handle takes its second stereo
    // stream and the chained launch's device-side gate, fe_track.cpp)
    const bool split_case = handles == 0;  // (the first handle of a run: exactly that configuration, from its first frame)
    c.lk_accum = split_case ? 2 : 1 + (int)(rnd() & 1u);
    const int space = split_case ? ESVIO_FE_DEVICE : (rnd() & 1u) ? ESVIO_FE_HOST : ESVIO_FE_DEVICE;
    // (HIPSTUB_FAIL_EVERY: the injected failures are for the calls; a creation — whose warm-up alone records 384
    // events and checks every return code — is let through)
    hipstub_arm_faults(0);
    const int crc = esvio_fe_create(&c, &h);
    hipstub_arm_faults(1);
    if (crc != ESVIO_FE_OK) { fprintf(stderr, "create failed\n"); return 3; }
    handles++;
    esvio_fe_reserve(h, 1u << 16, 1u << 16, 1);
    const int stretch = 20 + (int)(rnd() % 40u);  // frames on this handle
    std::vector<Batch> bs((size_t)stretch);
    for (int i = 0; i < stretch; i++) make_batch(bs[(size_t)i], W, H, f + i, 9000 + (int)(rnd() % 6000u));
    std::vector<int> pub((size_t)stretch);
    for (int i = 0; i < stretch; i++) pub[(size_t)i] = (rnd() % 3u) != 0;
    int announced = 0;                                 // batches [i + 1, announced] are announced
    bool replay = split_case || (rnd() & 1u) != 0;
    if (split_case) {
      esvio_fe_set_launch_thread(h, 1);
      esvio_fe_set_lazy_new_stereo(h, 1);
    }
    for (int i = 0; i < stretch; i++) {
      if (rnd() % 7u == 0) esvio_fe_set_launch_thread(h, (int)(rnd() & 1u));
      if (rnd() % 9u == 0) esvio_fe_set_host_threads(h, (int)(rnd() % 8u));
      if (rnd() % 11u == 0) esvio_fe_set_lazy_new_stereo(h, (int)(rnd() & 1u));
      if (rnd() % 13u == 0) replay = !replay;
      if (rnd() % 29u == 0) {  // a clean slate in the middle of the stream, whatever is announced or in flight
        esvio_fe_reset(h);
        resets++;
        announced = i;
      }
      if (announced < i) announced = i;
      if (replay) {
        const int ahead = 1 + (int)(rnd() % 4u);
        while (announced < i + ahead && announced + 1 < stretch) {
          announced++;
          const Batch& n = bs[(size_t)announced];
          int hint = pub[(size_t)announced];
          if (rnd() % 41u == 0) hint = 0;  // (a wrong hint: with more than one batch ahead the call for it is refused)
          esvio_fe_set_next_batch(h, n.t, n.L.data(), n.L.size(), n.R.data(), n.R.size(), space, hint);
        }
      }
      const Batch& b = bs[(size_t)i];
      const int rc = esvio_fe_track_event(h, b.t, b.L.data(), b.L.size(), b.R.data(), b.R.size(), space,
                                          pub[(size_t)i], &t);
      calls++;
      if (rc != ESVIO_FE_OK) {  // refused (wrong hint) or failed: the handle must be usable after a reset
        failed++;
        esvio_fe_reset(h);
        announced = i;
      }
      if (rnd() % 17u == 0) esvio_fe_finish(h, &t);
    }
    esvio_fe_finish(h, &t);
    {
      esvio_fe_latency_call lc;
      if (esvio_fe_latency_recent(h, 0, &lc) == ESVIO_FE_OK && lc.ms < 0) { fprintf(stderr, "latency record\n"); return 4; }
    }
    esvio_fe_destroy(h);
    f += stretch;
  }
  uint64_t tail[6];
  esvio_fe_ransac_tail(tail, 0);
  printf("drive ok: %ld calls on %ld handles, %ld refused/failed, %ld resets, tracks last %d / %d, ransac redone %llu\n", calls,
         handles, failed, resets, t.n_left, t.n_right, (unsigned long long)tail[2]);
  return 0;
}
