// The few "kernels" of the ThreadSanitizer build that do something (tests/hipstub): enough for the host side to have
// real work — corners appear, LK moves them a little, so tracks live on and rejectWithF_event runs its RANSAC on
// the helper pool every published frame — and the staged events are really read where the device would read them.
#include <cstring>

#include "fe_kernels.h"

namespace esvio {
namespace {
uint32_t lcg(uint32_t& s) { return s = s * 1664525u + 1013904223u; }
}  // namespace

// H2D of staged events by a kernel: the pinned buffer the staging threads filled is READ here
void launch_stage_pull(hipStream_t s, const void* pinned_src, void* dst, size_t bytes) {
  hipstub_stream_begin(s);
  std::memcpy(dst, pinned_src, bytes);
  hipstub_stream_end(s);
}

// ... of packed chunks (fe_evstage.cpp stage_pack): unpacked here as k_stage_pull_packed does
void launch_stage_pull_packed(hipStream_t s, const void* pinned_src, void* dst, size_t bytes, const void* desc, uint32_t epc) {
  hipstub_stream_begin(s);
  const uint32_t* d = (const uint32_t*)desc;
  const uint8_t* src = (const uint8_t*)pinned_src;
  uint32_t* out = (uint32_t*)dst;
  for (size_t i = 0; i < bytes / 16; i++) {
    const size_t c = i / epc, j = i - c * epc;
    const uint8_t* base = src + c * (size_t)epc * 16;
    if (d[2 * c + 1]) {
      uint32_t v[2];
      std::memcpy(v, base + 8 * j, 8);
      out[4 * i] = v[0];
      out[4 * i + 1] = d[2 * c] + (v[1] >> 31);
      out[4 * i + 2] = v[1] & 0x3fffffffu;
      out[4 * i + 3] = (v[1] >> 30) & 1u;
    } else {
      std::memcpy(out + 4 * i, base + 16 * j, 16);
    }
  }
  hipstub_stream_end(s);
}

void launch_compact(hipStream_t s, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, uint32_t*,
                    uint32_t* total, uint32_t*) {
  hipstub_stream_begin(s);
  if (total) *total = 0;
  hipstub_stream_end(s);
}

// Event_FeaturesToTrack: fills the free places with corners on a jittered grid (deterministic)
KernelId launch_select(hipStream_t s, const SelectArgs& a, size_t) {
  hipstub_stream_begin(s);
  static uint32_t seed = 12345;
  const int want = a.max_corners > 0 ? a.max_corners : 0;
  int k = 0;
  for (; k < want; k++) {
    const float x = 8.f + (float)(lcg(seed) % (uint32_t)(a.W - 16)), y = 8.f + (float)(lcg(seed) % (uint32_t)(a.H - 16));
    a.out_pts[a.out_base + k] = make_float2(x, y);
    if (a.out_idx) a.out_idx[a.out_base + k] = k;
    if (a.pub_slots)
      a.pub_slots[a.out_base + k] = ((unsigned long long)a.pub_seq << 32) | ((unsigned long long)(uint32_t)y << 16) | (uint32_t)x;
  }
  if (a.n_out) *a.n_out = k;
  if (a.n_total) *a.n_total = a.out_base + k;
  if (a.host_counts) {
    a.host_counts[0] = k;
    a.host_counts[1] = a.out_base + k;
    a.host_counts[2] = k;
  }
  if (a.pub_done) *a.pub_done = ((unsigned long long)a.pub_seq << 32) | (uint32_t)(a.out_base + k);
  hipstub_stream_end(s);
  return K_SELECT_MW;
}

// calcOpticalFlowPyrLK forward (+ backward): every point found, moved by a fraction of a pixel that depends on the
// point (so that the epipolar geometry is not degenerate), the backward pass lands on the start
void launch_lk(hipStream_t s, const LkArgs& f, const LkArgs* b, float2* back_pts, uint8_t* back_status) {
  hipstub_stream_begin(s);
  const int n = f.n_ptr ? *f.n_ptr : f.n_max;
  for (int i = 0; i < n && i < f.n_max; i++) {
    float2 p = f.prev_pts ? f.prev_pts[i] : make_float2(0.f, 0.f);
    if (f.chain_in) {  // the previous launch's forward result
      const unsigned long long vx = f.chain_in[2 * i], vy = f.chain_in[2 * i + 1];
      uint32_t ux = (uint32_t)vx, uy = (uint32_t)vy;
      std::memcpy(&p.x, &ux, 4);
      std::memcpy(&p.y, &uy, 4);
    } else if (f.poll_slots && i >= f.poll_from) {
      const unsigned long long v = f.poll_slots[i];
      p = make_float2((float)(uint32_t)(v & 0xffffu), (float)(uint32_t)((v >> 16) & 0xffffu));
    }
    const float2 q = make_float2(p.x + 0.25f + 0.001f * (float)(i % 37), p.y - 0.125f + 0.002f * (float)(i % 11));
    f.next_pts[i] = q;
    f.status[i] = 1;
    if (f.chain_out) {
      uint32_t ux, uy;
      std::memcpy(&ux, &q.x, 4);
      std::memcpy(&uy, &q.y, 4);
      const unsigned long long hi = (unsigned long long)((f.chain_seq << 2) | 1u) << 32;
      f.chain_out[2 * i] = hi | ux;
      f.chain_out[2 * i + 1] = hi | uy;
    }
    if (b && back_pts) {
      back_pts[i] = p;
      back_status[i] = 1;
    }
  }
  hipstub_stream_end(s);
}

}  // namespace esvio
