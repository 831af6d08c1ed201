// fe_api.cpp — handle, device memory, per-frame orchestration and the C ABI (include/esvio_fe.h).
//
// The per-frame sequence mirrors FeatureTracker::trackEvent (reference:
// feature_tracker/src/feature_tracker.cpp:340-603); every data-parallel stage is a HIP kernel
// from fe_kernels.hip, the <= max_cnt-point bookkeeping stays on the host exactly where the
// reference has it.  There is no CPU fallback: without a HIP device esvio_fe_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <utility>
#include <deque>
#include <vector>

#include "../../include/esvio_fe.h"
#include "fe_host.h"
#include "fe_kernels.h"
#include "fe_mc.h"

using namespace esvio;

namespace {

struct P2f {
  float x, y;
};

// std::map<int, cv::Point2f> as ptsVelocity uses it (insert-if-absent, find, empty, clear), kept as a
// sorted flat vector: same semantics, no node allocations per frame
struct IdMap {
  std::vector<std::pair<int, P2f>> v;
  bool empty() const { return v.empty(); }
  void clear() { v.clear(); }
  void swap(IdMap& o) { v.swap(o.v); }
  void build(const std::vector<int>& ids, const std::vector<P2f>& pts) {
    v.clear();
    v.reserve(ids.size());
    for (size_t i = 0; i < ids.size(); i++) v.emplace_back(ids[i], pts[i]);
    std::stable_sort(v.begin(), v.end(),
                     [](const std::pair<int, P2f>& a, const std::pair<int, P2f>& b) { return a.first < b.first; });
    // map::insert keeps the first element of equal keys
    v.erase(std::unique(v.begin(), v.end(),
                        [](const std::pair<int, P2f>& a, const std::pair<int, P2f>& b) { return a.first == b.first; }),
            v.end());
  }
  const P2f* find(int id) const {
    auto it = std::lower_bound(v.begin(), v.end(), id,
                               [](const std::pair<int, P2f>& a, int k) { return a.first < k; });
    return (it != v.end() && it->first == id) ? &it->second : nullptr;
  }
};

const char* const kKernelNames[K_COUNT] = {
    "k_sae_keys", "k_radix_pass", "k_sae_apply",
    "k_time_surface", "k_clahe", "k_pyr_down", "k_pyr_pad", "k_scharr", "k_lk", "k_arc", "k_compact", "k_select",
    "k_arc_map"};

struct KStat {
  double ms = 0;
  uint64_t launches = 0;
  uint64_t bytes = 0;
};

struct ProfRec {
  int id;
  hipEvent_t a, b;
  uint64_t bytes;
};

// buildOpticalFlowPyramid's level count [OpenCV video/lkpyramid.cpp]
int pyr_levels(int w, int h, int win, int max_level) {
  int sw = w, sh = h;
  for (int level = 0; level <= max_level; ++level) {
    sw = (sw + 1) / 2;
    sh = (sh + 1) / 2;
    if (sw <= win || sh <= win) return level;
  }
  return max_level;
}

// esvio_fe_set_next_batch: a batch announced ahead of its trackEvent call ...
struct Batch {
  const esvio_fe_event *left = nullptr, *right = nullptr;
  size_t nL = 0, nR = 0;
  int space = 0;
  double time = 0;
  int pub = 0;  // caller's PUB_THIS_FRAME hint
};
// ... and, once its SAE update / images / pyramids (/ Arc*) are enqueued on the prefetch stream,
// the resources they were given
struct Inflight : Batch {
  int lane = 0;  // staging buffer + event pair
  int slotL = 0, slotR = 0, raw = 0, cand = 0;
  const EventRec *dL = nullptr, *dR = nullptr;
  bool arc_done = false;
};
constexpr int kPrefetchDepth = 3;
constexpr int kLeftSlots = 2 + kPrefetchDepth;   // prev, cur, prefetched...
constexpr int kRightSlots = 1 + kPrefetchDepth;  // cur, prefetched...

struct PyrStore {
  PyrDesc d{};
  void* mem = nullptr;
  size_t bytes = 0;
  int w = 0, h = 0, max_level = -1;
};

}  // namespace

struct esvio_fe_ctx {
  esvio_fe_config cfg{};
  int dev = 0;
  hipStream_t stream = nullptr;   // main stream
  hipStream_t stream2 = nullptr;  // prefetch stream (next batch's SAE update / images)
  hipStream_t stream3 = nullptr;  // speculative temporal LK of the next frame
  // stereo LK of the temporal survivors: nothing on the frame's chain reads its results before the
  // right-camera tail, and on the main stream it would hold up the corner selection behind it
  hipStream_t stream4 = nullptr;
  hipEvent_t ev_planes_free = nullptr;
  hipEvent_t ev_pts_ready = nullptr, ev_spec_done = nullptr, ev_sel_host = nullptr;
  std::string err;
  int W = 0, H = 0;
  uint32_t P = 0;
  int key_bits = 0;
  uint32_t invalid_key = 0;
  std::vector<int> hw;  // disc half-widths for min_dist

  // ---- device state
  double2* L2 = nullptr;  // [2P] {L[0],L[1]} per (cam,pixel)
  double2* S2 = nullptr;  // [2P] {S[0],S[1]}
  EventRec* d_ev = nullptr;
  size_t ev_cap = 0;
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *hist = nullptr;
  size_t sort_cap = 0, hist_cap = 0;
  size_t sae_ev_min = (size_t)1 << 20;  // batches of at least this many events: k_sae_apply_ev
  // tiled SAE update (default; ESVIO_FE_SAE_SORT=1 or a sensor too large for one digit: the radix
  // sort form above)
  bool tiled = false;
  TileGeom tgeom{};
  EventRec* d_part = nullptr;  // the batch's events partitioned by bucket
  size_t part_cap = 0;
  uint32_t* d_tile = nullptr;  // TileScratch
  size_t tile_cap = 0;
  uint8_t* sae_marks = nullptr;         // [sort_cap] its per-event "stores L / stores S" marks
  unsigned long long* d_rejected = nullptr;
  // left: slots 0..kLeftSlots-1 rotate (prev, cur, up to kPrefetchDepth being prefetched);
  // right: the kRightSlots after them (cur + prefetched)
  PyrStore pyr[kLeftSlots + kRightSlots];
  int slot_prevL = 0, slot_curL = 0, slot_curR = kLeftSlots;
  bool have_img = false;
  bool ext_right_pending = false;  // esvio_fe_import_image(cam=1) done for the next frame
  // time-sliced stream (esvio_fe_sae_slice_*): scratch planes a slice is applied to, and the one-shot
  // "the planes already hold the next frame's batch" set by esvio_fe_sae_slice_commit
  double2 *L2s = nullptr, *S2s = nullptr;
  double* slice_stage = nullptr;  // device staging for host-side slice planes
  size_t slice_stage_doubles = 0;
  bool ext_sae_pending = false;
  // esvio_fe_exchange_tracks: send / receive buffers of the all-gather and the pinned pack area
  float *x_send = nullptr, *x_recv = nullptr, *x_pin = nullptr;
  size_t x_recv_cap = 0;
  // ---- next-batch prefetch (esvio_fe_set_next_batch)
  std::deque<Batch> announced;     // announced, nothing enqueued yet (<= kPrefetchDepth)
  std::deque<Inflight> inflight;   // SAE update / images / pyramids enqueued on the prefetch stream
  bool cur_prefetched = false;     // the frame being processed came from the prefetch stream
  EventRec* d_evp[kPrefetchDepth] = {};  // host-event staging, one per prefetch lane
  size_t evp_cap[kPrefetchDepth] = {};
  hipEvent_t ev_lane_done[kPrefetchDepth] = {}, ev_lane_arc[kPrefetchDepth] = {};
  PyrStore tmp_pyr[2];  // standalone LK / pyramid taps on arbitrary host images
  PyrStore med_tmp[2];  // median_blur_kernel_size > 0: the surfaces before cv::medianBlur
  // equalize: raw time surfaces (single padded level each, left/right) + CLAHE scratch
  PyrStore raw[kRightSlots][2];  // [buffer][cam], rotating like the right pyramids
  int raw_cur = 0;
  uint8_t* d_lut = nullptr;
  int* d_minmax = nullptr;
  // device-side point / status buffers of the standalone entry points (LK, featuresToTrack) and the
  // selection counters; one allocation with the layout of ResLayout
  uint8_t* d_res = nullptr;
  size_t res_bytes = 0;
  float2 *d_ptsA = nullptr, *d_ptsB = nullptr, *d_ptsC = nullptr, *d_ptsD = nullptr;
  uint8_t *d_stA = nullptr, *d_stB = nullptr;
  int* d_counts = nullptr;  // [0]=n_out (select) [1]=n_total (kept + new: the LK kernels' n_ptr)
  // The per-frame path works on the pinned host block itself (device-visible): the LK kernels read
  // their points from it and write results into it, k_select mirrors its counters into it — no
  // H2D / D2H copy calls on the frame's critical path (each costs more host time than the few
  // hundred bytes take over PCIe).  z_* = device-side addresses of the h_pin / h_spec regions.
  uint8_t *z_res = nullptr, *z_spec = nullptr;
  // (set 1 — temporal LK, then stereo LK of the survivors — exists twice, see pin_of(); its device
  // addresses come from zdev())
  float2 *z_new = nullptr, *z_ptsB2 = nullptr, *z_ptsC2 = nullptr;
  uint8_t *z_stA2 = nullptr, *z_stB2 = nullptr;
  int* z_counts = nullptr;
  int res_set = 0;  // which copy of set 1 the current frame works in
  int lks_last = -1;  // copy the latest stereo LK launch (stream4) writes to, -1: none so far
  // ---- speculative temporal LK of the next frame (replay mode): once this frame's kept points
  // and new corners are final, next frame's calcOpticalFlowPyrLK(cur -> next) pair is launched on
  // stream3 against the prefetched pyramids, so it overlaps this frame's stereo LK and host tail
  uint8_t* h_spec = nullptr;  // pinned, device-visible: [ptsB | ptsC | stA | stB] of that launch
  size_t spec_bytes = 0;
  bool spec_valid = false;
  int spec_n = 0;             // number of points of that launch (= the next frame's prev_pts.size())
  // ---- chained temporal LK of the frame after next: when the next frame publishes nothing, the
  // frame after it tracks exactly the next frame's forward results, point by point, so its launch
  // (stream4) is made together with the speculative one and each of its waves starts the moment the
  // producer's wave of the same index publishes its forward result (LkArgs::chain_*).  Results:
  // second half of h_spec, indexed like the producer's points; the intermediate frame's temporal
  // filter gives the map from the final frame's prev_pts to those indices.
  unsigned long long* d_chain = nullptr;  // [2 * max_cnt] published forward results
  uint32_t chain_seq = 0;
  bool chain_enabled = true;   // (ESVIO_FE_NO_CHAIN=1 turns it off: A/B measurements)
  bool chain_valid = false;    // a chained launch has been made ...
  uint64_t chain_for = 0;      // ... for the frame with this number
  bool chain_map_ok = false;
  std::vector<int> chain_map;  // final frame's prev_pts[j] = producer point chain_map[j]
  uint64_t frame_no = 0;       // trackEvent calls so far
  hipEvent_t ev_chain_done = nullptr;
  // k_select publishes each new corner as it accepts it; the speculative launch, already resident,
  // picks them up one by one instead of starting after the whole selection
  unsigned long long *d_pub_slots = nullptr, *d_pub_done = nullptr;
  uint32_t pub_seq = 0;
  // ---- lazy stereo of new corners (esvio_fe_set_lazy_new_stereo): a published frame returns
  // without waiting for the stereo LK of the corners it has just detected; their right-camera
  // entries are appended by the next call (before anything reads them) or by esvio_fe_finish
  bool lazy_new = false;
  struct PendingNew {
    bool active = false;
    bool prev_map_was_empty = false;
    std::vector<int> ids;       // the new corners' ids
    std::vector<P2f> left;      // ... and left positions
  } pend;
  // ... and a frame that publishes nothing returns without waiting for its stereo LK at all: the
  // whole right-camera tail (:475-575) is run by the next call, in the shadow of its own kernels,
  // or by esvio_fe_finish.  The next frame works in the other copy of set 1 meanwhile.
  struct PendingRight {
    bool active = false;
    int set = 0;                // copy of set 1 that holds this frame's stereo LK results
    double dt = 0;              // cur_time - prev_time of that frame
    std::vector<int> ids;       // the frame's ids / left points (no new corners: nothing published)
    std::vector<P2f> left;
  } pend_right;
  hipEvent_t ev_lks_done[2] = {nullptr, nullptr}, ev_lknew_done = nullptr;
  host::RansacPool* pool = nullptr;  // esvio_fe_set_host_threads
  // the prefetch stream's per-batch launch sequence as a HIP graph (fe_kernels.h)
  // Off by default: measured on MI355X / ROCm 7.2 it saves ~17 us of host time per batch but the
  // graph's kernels complete ~50 us later than the same kernels launched one by one, and the
  // chained temporal LK then waits for the pyramids.  ESVIO_FE_GRAPH=1 turns it on.
  bool graphs_enabled = false;
  LaunchList rec;
  LaunchGraph pf_graph;
  // arc / select
  uint8_t* d_flags = nullptr;
  // per-block ordered candidate lists written by k_arc; two sets so that the Arc* of a prefetched
  // batch (prefetch stream) never overwrites the set the current frame's selection still reads
  struct CandSet {
    uint32_t *xy = nullptr, *idx = nullptr, *cnt = nullptr;
    // ... and their ordered compaction into one stream (k_compact, launched right behind k_arc)
    uint32_t *comp_xy = nullptr, *comp_idx = nullptr, *total = nullptr;
    size_t cap = 0;
  } cand[kRightSlots];
  // per-pixel earliest candidate of a set's latest Arc* pass (ArcArgs::first_map / launch_dedup)
  uint32_t* d_first[kRightSlots] = {};
  // per-pixel, per-polarity result of the event-independent part of isCorner (k_arc_map), one map
  // per candidate set
  uint32_t* d_cmap[kRightSlots] = {};
  uint8_t* d_touched[kRightSlots] = {};  // (pixel, polarity) pairs a batch's left events hit
  uint32_t first_epoch[kRightSlots] = {};  // Arc* passes into the set so far
  bool dedup_enabled = true;               // (ESVIO_FE_NO_DEDUP=1: A/B measurements)
  bool fuse_ts_pyr = true;                 // (ESVIO_FE_NO_FUSE=1: k_time_surface + 3 x k_pyr_down)
  bool disc_tab_only = false;              // (ESVIO_FE_DISC_TABLE=1: k_select's table look-ups)
  int cand_cur = 0;
  size_t arc_cap = 0;
  uint32_t* d_mask_bits = nullptr;
  // goodFeaturesToTrack scratch (image front-end), allocated on first use
  float4 *d_gftt_cov = nullptr, *d_gftt_rowsum = nullptr;
  float* d_gftt_eig = nullptr;
  uint32_t* d_gftt_max = nullptr;
  int32_t* d_sel_idx = nullptr;
  // pinned host staging (layout: pin_of())
  uint8_t* h_pin = nullptr;
  size_t h_pin_bytes = 0;

  // ---- FeatureTracker state (feature_tracker.h:119-173)
  int n_id = 0;
  double cur_time = 0, prev_time = 0;
  std::vector<P2f> prev_pts, cur_pts, cur_right_pts, n_pts;
  std::vector<P2f> cur_un_pts, cur_un_right_pts, pts_velocity, right_pts_velocity;
  std::vector<int> ids, ids_right, track_cnt, track_cnt_right;
  std::vector<int> src_idx;  // per cur_pts entry: index into the speculative stereo-LK results
  IdMap cur_un_pts_map, prev_un_pts_map, cur_un_right_pts_map, prev_un_right_pts_map;
  host::BitMask mask_event;

  // ---- host phase trace (ESVIO_FE_TRACE=1): stage, sae+ts enqueue, sync A, host A, enqueue B,
  // sync B, host B
  bool trace = false;
  double phase_ms[2][8] = {};  // [published?][phase]
  double pub_ms[6] = {};       // published frames: the parts of "host mask + enqueue detect/stereo"
  uint64_t phase_count[2] = {0, 0};
  uint64_t phase_frames = 0, tr_cand = 0, tr_new = 0, tr_detect = 0, tr_surv = 0;
  double tr_fm_ms = 0;  // time inside find_fundamental_mat alone
  double tr_fm_max_ms = 0, tr_lift_ms = 0;  // ... its slowest call; the two liftProjective batches
  uint64_t tr_chain_launch = 0, tr_chain_used = 0, tr_chain_cancel = 0, tr_spec_used = 0;
  // (trace only) device-side intervals of the published frame's chain, from timing events
  hipEvent_t ev_dbg_sel_start = nullptr;
  double tr_gpu_sel = 0, tr_gpu_spec = 0, tr_gpu_chain = 0, tr_host_chain = 0, tr_gpu_pyr = 0;
  int tr_lane = -1;  // prefetch lane of the frame being tracked
  uint64_t tr_gpu_n = 0;
  std::chrono::steady_clock::time_point tr_sel_launch;

  // ---- profiling
  bool prof_on = false;
  KStat stats[K_COUNT];
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> ev_pool;
};

namespace {

// The stream the helpers enqueue on: the main stream unless the calling thread has switched to
// another one (prefetch -> stream2, speculative LK -> stream3).
thread_local hipStream_t t_stream_override = nullptr;
inline hipStream_t cur_stream(const esvio_fe_ctx* c) {
  return t_stream_override ? t_stream_override : c->stream;
}
struct StreamScope {
  hipStream_t saved;
  explicit StreamScope(hipStream_t s) : saved(t_stream_override) { t_stream_override = s; }
  ~StreamScope() { t_stream_override = saved; }
};

int fail(esvio_fe_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    c->err = buf;
  }
  return code;
}

#define HIPCHK(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail((c), ESVIO_FE_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                  __FILE__, __LINE__);                                                       \
  } while (0)

// ---------------------------------------------------------------- profiling
hipEvent_t get_event(esvio_fe_ctx* c) {
  if (!c->ev_pool.empty()) {
    hipEvent_t e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

struct ScopedKernel {  // brackets one launch with HIP events on the handle's stream
  esvio_fe_ctx* c;
  int id;
  uint64_t bytes;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedKernel(esvio_fe_ctx* ctx, int kid, uint64_t alg_bytes) : c(ctx), id(kid), bytes(alg_bytes) {
    if (c->prof_on) {
      a = get_event(c);
      b = get_event(c);
      (void)hipEventRecord(a, cur_stream(c));
    }
  }
  ~ScopedKernel() {
    if (a) {
      (void)hipEventRecord(b, cur_stream(c));
      c->pending.push_back(ProfRec{id, a, b, bytes});
    }
  }
};

void resolve_profile(esvio_fe_ctx* c) {  // main stream idle; prefetch-stream records may be pending
  std::vector<ProfRec> keep;
  for (auto& r : c->pending) {
    float ms = 0;
    const hipError_t e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e == hipErrorNotReady) {
      keep.push_back(r);
      continue;
    }
    if (e == hipSuccess) {
      c->stats[r.id].ms += ms;
      c->stats[r.id].launches++;
      c->stats[r.id].bytes += r.bytes;
    }
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->pending.swap(keep);
  (void)hipGetLastError();
}

// ---------------------------------------------------------------- memory
template <class T>
int dev_alloc(esvio_fe_ctx* c, T** p, size_t count) {
  HIPCHK(c, hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
  return 0;
}

int ensure_event_capacity(esvio_fe_ctx* c, size_t n) {
  if (n <= c->ev_cap) return 0;
  size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
  if (c->d_ev) (void)hipFree(c->d_ev);
  c->d_ev = nullptr;
  c->ev_cap = 0;
  if (int rc = dev_alloc(c, &c->d_ev, cap)) return rc;
  c->ev_cap = cap;
  return 0;
}

int ensure_sort_capacity(esvio_fe_ctx* c, size_t n) {
  if (n > c->sort_cap) {
    size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
    for (int i = 0; i < 2; i++) {
      if (c->keys[i]) (void)hipFree(c->keys[i]);
      if (c->vals[i]) (void)hipFree(c->vals[i]);
      c->keys[i] = c->vals[i] = nullptr;
    }
    c->sort_cap = 0;
    for (int i = 0; i < 2; i++) {
      if (int rc = dev_alloc(c, &c->keys[i], cap)) return rc;
      if (int rc = dev_alloc(c, &c->vals[i], cap)) return rc;
    }
    c->sort_cap = cap;
    if (c->sae_marks) (void)hipFree(c->sae_marks);
    c->sae_marks = nullptr;
    if (int rc = dev_alloc(c, &c->sae_marks, cap)) return rc;
  }
  // [ghist + tickets | lookback for every pass]
  const size_t head = ((size_t)kRadixMaxPasses << kRadixMaxBits) + 64;
  size_t hneed = head + (size_t)kRadixMaxPasses * (radix_blocks((uint32_t)c->sort_cap) << kRadixMaxBits);
  if (hneed > c->hist_cap) {
    if (c->hist) (void)hipFree(c->hist);
    c->hist = nullptr;
    c->hist_cap = 0;
    if (int rc = dev_alloc(c, &c->hist, hneed)) return rc;
    HIPCHK(c, hipMemsetAsync(c->hist, 0, hneed * 4, cur_stream(c)));
    c->hist_cap = hneed;
  }
  return 0;
}

int ensure_cand_capacity(esvio_fe_ctx* c, int set, size_t n) {
  esvio_fe_ctx::CandSet& s = c->cand[set];
  if (n <= s.cap) return 0;
  size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
  cap = (cap + kArcBlock - 1) / kArcBlock * kArcBlock;
  void* ptrs[] = {s.xy, s.idx, s.cnt, s.comp_xy, s.comp_idx, s.total};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  s = esvio_fe_ctx::CandSet();
  if (int rc = dev_alloc(c, &s.xy, cap)) return rc;
  if (int rc = dev_alloc(c, &s.idx, cap)) return rc;
  if (int rc = dev_alloc(c, &s.cnt, cap / kArcBlock)) return rc;
  if (int rc = dev_alloc(c, &s.comp_xy, cap)) return rc;
  if (int rc = dev_alloc(c, &s.comp_idx, cap)) return rc;
  if (int rc = dev_alloc(c, &s.total, 1)) return rc;
  s.cap = cap;
  return 0;
}

// per-event flags (standalone isCorner) and candidate set `set`
int ensure_arc_capacity(esvio_fe_ctx* c, size_t n, int set) {
  if (int rc = ensure_cand_capacity(c, set, n)) return rc;
  if (n <= c->arc_cap) return 0;
  size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
  cap = (cap + kArcBlock - 1) / kArcBlock * kArcBlock;
  if (c->d_flags) (void)hipFree(c->d_flags);
  c->d_flags = nullptr;
  c->arc_cap = 0;
  if (int rc = dev_alloc(c, &c->d_flags, cap)) return rc;
  c->arc_cap = cap;
  return 0;
}

int pyr_alloc(esvio_fe_ctx* c, PyrStore& ps, int w, int h, int max_level) {
  if (ps.mem && ps.w == w && ps.h == h && ps.max_level == max_level) return 0;
  if (ps.mem) (void)hipFree(ps.mem);
  ps = PyrStore();
  const int levels = pyr_levels(w, h, kLkWin, max_level);
  size_t off = 0, img_off[kMaxLevels], der_off[kMaxLevels];
  int lw = w, lh = h;
  for (int l = 0; l <= levels; l++) {
    ps.d.stride[l] = pyr_stride(lw);
    const size_t area = (size_t)ps.d.stride[l] * (lh + 2 * kPad);
    img_off[l] = off;
    off += (area + 255) / 256 * 256;
    der_off[l] = off;
    off += (area * 4 + 255) / 256 * 256;
    ps.d.w[l] = lw;
    ps.d.h[l] = lh;
    lw = (lw + 1) / 2;
    lh = (lh + 1) / 2;
  }
  HIPCHK(c, hipMalloc(&ps.mem, off));
  HIPCHK(c, hipMemsetAsync(ps.mem, 0, off, cur_stream(c)));  // derivative borders stay 0 forever
  for (int l = 0; l <= levels; l++) {
    ps.d.img[l] = (uint8_t*)ps.mem + img_off[l];
    ps.d.deriv[l] = (int16_t*)((uint8_t*)ps.mem + der_off[l]);
  }
  for (int l = levels + 1; l < kMaxLevels; l++) {
    ps.d.img[l] = ps.d.img[levels];
    ps.d.deriv[l] = ps.d.deriv[levels];
    ps.d.w[l] = ps.d.w[levels];
    ps.d.h[l] = ps.d.h[levels];
    ps.d.stride[l] = ps.d.stride[levels];
  }
  ps.d.levels = levels;
  ps.bytes = off;
  ps.w = w;
  ps.h = h;
  ps.max_level = max_level;
  return 0;
}

// level 0 interior already written -> pyrDown chain, border fill, Scharr
void pyr_build(esvio_fe_ctx* c, const PyrDesc* p, int nimg) {
  uint64_t px0 = (uint64_t)p[0].w[0] * p[0].h[0] * nimg;
  for (int l = 0; l < p[0].levels; l++) {
    uint64_t src = (uint64_t)p[0].w[l] * p[0].h[l], dst = (uint64_t)p[0].w[l + 1] * p[0].h[l + 1];
    ScopedKernel k(c, K_PYR_DOWN, (src + dst) * nimg);
    launch_pyr_down(cur_stream(c), p, nimg, l);
  }
  {
    ScopedKernel k(c, K_PYR_PAD, 0);
    launch_pyr_pad(cur_stream(c), p, nimg);
  }
  {
    uint64_t all = 0;
    for (int l = 0; l <= p[0].levels; l++) all += (uint64_t)p[0].w[l] * p[0].h[l];
    ScopedKernel k(c, K_SCHARR, all * 5 * nimg);  // 1 B read + 4 B written per pixel
    launch_scharr(cur_stream(c), p, nimg);
  }
  (void)px0;
}

// both cameras' LK images of a batch + their pyramids: render_lk_images + pyr_build, with the
// time-surface and pyrDown launches fused into one when nothing sits between them
void render_and_build(esvio_fe_ctx* c, double t_sync, int slotL, int slotR, int rawbuf);

// ---------------------------------------------------------------- SAE update (both cameras)
// Motion_correction_value -> kernel parameters; first_left_host: left.events[0] (host copy)
McParams make_mc_params(const esvio_fe_motion* m, const esvio_fe_event& first_left) {
  McParams p;
  std::memset(&p, 0, sizeof(p));
  p.enabled = 1;
  p.t0 = (double)first_left.sec + 1e-9 * (double)first_left.nsec;  // ros::Time::toSec()
  p.dt_batch = m->t1 - p.t0;
  const double an = std::sqrt(std::pow((double)m->accel[0], 2) + std::pow((double)m->accel[1], 2) +
                              std::pow((double)m->accel[2], 2));
  p.active = an > 5;  // a_motion_compensation_threshold (event_detector.h:51)
  for (int i = 0; i < 3; i++) {
    p.vsum[i] = (float)m->v[i] + m->v_pre[i];
    p.omega[i] = m->omega[i];
  }
  M3f K;
  std::memset(&K, 0, sizeof(K));
  K.m[0][0] = (float)m->fx;
  K.m[0][2] = (float)m->cx;
  K.m[1][1] = (float)m->fy;
  K.m[1][2] = (float)m->cy;
  K.m[2][2] = 1.f;
  p.K = K;
  p.Kinv = mc_inverse(K);
  return p;
}

int first_event_host(esvio_fe_ctx* c, const esvio_fe_event* left, int space, esvio_fe_event* out) {
  if (space == ESVIO_FE_HOST) {
    *out = left[0];
    return 0;
  }
  HIPCHK(c, hipMemcpy(out, left, sizeof(*out), hipMemcpyDeviceToHost));
  return 0;
}

int sae_update_tiled(esvio_fe_ctx* c, const EventRec* evL, uint32_t nL, const EventRec* evR, uint32_t nR,
                     double2* L2, double2* S2, uint8_t* arc_touched) {
  const uint32_t n = nL + nR;
  if (n > c->part_cap) {
    const size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
    if (c->d_part) (void)hipFree(c->d_part);
    c->d_part = nullptr;
    c->part_cap = 0;
    if (int rc = dev_alloc(c, &c->d_part, cap)) return rc;
    c->part_cap = cap;
  }
  const size_t nblk_cap = (c->part_cap + 2047) / 2048;  // (2048 events per scatter block at least)
  const size_t head = (size_t)3 * kTileMaxBins + 64;
  const size_t need = head + (nblk_cap + 2 * (size_t)kTileMaxGroups) * kTileMaxBins;
  if (need > c->tile_cap) {
    if (c->d_tile) (void)hipFree(c->d_tile);
    c->d_tile = nullptr;
    c->tile_cap = 0;
    if (int rc = dev_alloc(c, &c->d_tile, need)) return rc;
    c->tile_cap = need;
  }
  TileScratch sc;
  sc.totals = c->d_tile;
  sc.tile_off = c->d_tile + kTileMaxBins;
  sc.tile_order = c->d_tile + 2 * kTileMaxBins + 32;
  sc.P = c->d_tile + head;
  sc.T = sc.P + nblk_cap * kTileMaxBins;
  sc.C = sc.T + (size_t)kTileMaxGroups * kTileMaxBins;
  {
    ScopedKernel k(c, K_SAE_KEYS, (uint64_t)n * 16);  // ingest: the raw records, read once
    launch_tile_hist(cur_stream(c), evL, nL, evR, nR, c->tgeom, sc, c->d_rejected);
  }
  {
    ScopedKernel k(c, K_RADIX_PASS, (uint64_t)n * 32);  // the partition's own traffic: 16 B in, 16 B out
    launch_tile_scatter(cur_stream(c), evL, nL, evR, nR, c->tgeom, sc, c->d_part);
  }
  {
    ScopedKernel k(c, K_SAE_APPLY, (uint64_t)n * 32);
    launch_tile_apply(cur_stream(c), c->d_part, n, c->tgeom, sc, L2, S2, c->cfg.feature_filter_threshold,
                      arc_touched, c->z_counts + 3);
  }
  return 0;
}

// arc_set >= 0: this batch's Arc* pass will run into candidate set arc_set; *arc_marked tells
// whether the update has set that set's touched flags on its way (else run_arc does it)
int sae_update(esvio_fe_ctx* c, const EventRec* evL, uint32_t nL, const EventRec* evR,
               uint32_t nR, const McParams* mc = nullptr, double2* L2 = nullptr, double2* S2 = nullptr,
               int arc_set = -1, bool* arc_marked = nullptr) {
  const uint32_t n = nL + nR;
  if (!n) return 0;
  if (!L2) L2 = c->L2;  // (other planes: the scratch pair of the time-slice entry points)
  if (!S2) S2 = c->S2;
  if (arc_marked) *arc_marked = false;
  if (c->tiled && !mc) {
    uint8_t* mark = arc_set >= 0 && nL ? c->d_touched[arc_set] : nullptr;
    if (arc_marked) *arc_marked = mark != nullptr;
    return sae_update_tiled(c, evL, nL, evR, nR, L2, S2, mark);
  }
  if (int rc = ensure_sort_capacity(c, n)) return rc;
  const int passes = (c->key_bits + 6) / 7;
  const int bits = (c->key_bits + passes - 1) / passes;
  const uint32_t nblk = radix_blocks(n);
  const uint32_t head = ((uint32_t)kRadixMaxPasses << kRadixMaxBits) + 64;
  uint32_t* ghist = c->hist;                                        // [passes << bits]
  uint32_t* tickets = c->hist + ((size_t)kRadixMaxPasses << kRadixMaxBits);  // [passes]
  uint32_t* lookback = c->hist + head;                              // [passes][nblk << bits]
  const uint32_t lb_words = (uint32_t)passes * (nblk << bits);
  {
    ScopedKernel k(c, K_SAE_KEYS, (uint64_t)n * 16);
    launch_sae_keys(cur_stream(c), evL, nL, evR, nR, c->W, c->H, c->keys[0], c->vals[0], c->invalid_key,
                    c->d_rejected, passes, bits, ghist, lookback, lb_words, mc);
  }
  int cur = 0;
  for (int p = 0; p < passes; p++) {
    ScopedKernel k(c, K_RADIX_PASS, (uint64_t)n * 16);
    launch_radix_pass(cur_stream(c), c->keys[cur], c->vals[cur], n, p * bits, bits, ghist + ((size_t)p << bits),
                      lookback + (size_t)p * (nblk << bits), tickets + p, c->keys[cur ^ 1],
                      c->vals[cur ^ 1], c->z_counts + 3);
    cur ^= 1;
  }
  {
    ScopedKernel k(c, K_SAE_APPLY, (uint64_t)n * 32);
    if (n >= c->sae_ev_min)  // many events per pixel: one lane per event
      launch_sae_apply_ev(cur_stream(c), c->keys[cur], c->vals[cur], n, evL, nL, evR, L2, S2,
                          c->cfg.feature_filter_threshold, c->invalid_key, c->hist, head, c->sae_marks);
    else
      launch_sae_apply(cur_stream(c), c->keys[cur], c->vals[cur], n, evL, nL, evR, L2, S2,
                       c->cfg.feature_filter_threshold, c->invalid_key, c->hist, head);
  }
  return 0;
}

// stage host events into the handle's device buffer; returns device pointers
int stage_events(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL,
                 const esvio_fe_event* right, size_t nR, int space, const EventRec** dL,
                 const EventRec** dR, int lane = -1) {
  if (space == ESVIO_FE_DEVICE) {
    *dL = (const EventRec*)left;
    *dR = (const EventRec*)right;
    return 0;
  }
  if (space != ESVIO_FE_HOST) return fail(c, ESVIO_FE_EINVAL, "bad memory space %d", space);
  EventRec** buf = lane >= 0 ? &c->d_evp[lane] : &c->d_ev;
  size_t* cap = lane >= 0 ? &c->evp_cap[lane] : &c->ev_cap;
  if (nL + nR > *cap) {
    const size_t ncap = std::max<size_t>(nL + nR + (nL + nR) / 4, 1 << 16);
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    if (int rc = dev_alloc(c, buf, ncap)) return rc;
    *cap = ncap;
  }
  if (nL) HIPCHK(c, hipMemcpyAsync(*buf, left, nL * 16, hipMemcpyHostToDevice, cur_stream(c)));
  if (nR) HIPCHK(c, hipMemcpyAsync(*buf + nL, right, nR * 16, hipMemcpyHostToDevice, cur_stream(c)));
  *dL = *buf;
  *dR = *buf + nL;
  return 0;
}

void render_ts(esvio_fe_ctx* c, double t_sync, uint8_t* dst0, uint8_t* dst1, int ncam,
               const double2* S2) {
  const int stride = c->pyr[0].d.stride[0];
  const int mk = c->cfg.median_blur_kernel_size;
  uint8_t* r0 = mk > 0 ? c->med_tmp[0].d.img[0] : dst0;
  uint8_t* r1 = mk > 0 ? c->med_tmp[ncam == 2 ? 1 : 0].d.img[0] : dst1;
  {
    ScopedKernel k(c, K_TIME_SURFACE, (uint64_t)c->P * 17 * ncam);
    launch_time_surface(cur_stream(c), S2, c->W, c->H, t_sync, c->cfg.decay_ms / 1000.0,
                        c->cfg.ignore_polarity, r0, r1, stride, ncam);
  }
  if (mk > 0) {  // cv::medianBlur(2k+1) of the rendered surface (event_detector.cc:262-264)
    const size_t o = (size_t)kPad * stride + kPad;
    ScopedKernel k(c, K_TIME_SURFACE, 0);
    launch_median(cur_stream(c), r0 + o, r1 + o, stride, dst0 + o, dst1 + o, stride, c->W, c->H, mk, ncam);
  }
}

inline uint8_t* px00(const PyrDesc& d) { return d.img[0] + (size_t)kPad * d.stride[0] + kPad; }

// the image trackEvent feeds to LK: the raw time surface, or CLAHE + normalize of it when
// `equalize` (feature_tracker.cpp:375-387).  cams: bit 0 left, bit 1 right.  Raw surfaces stay
// available for the TS_LK_THRESHOLD test and gettimesurface().
void render_lk_images(esvio_fe_ctx* c, double t_sync, int cams, int slotL, int slotR, int rawbuf) {
  const PyrDesc& L = c->pyr[slotL].d;
  const PyrDesc& R = c->pyr[slotR].d;
  if (!c->cfg.equalize) {
    if (cams == 3) render_ts(c, t_sync, L.img[0], R.img[0], 2, c->S2);
    else if (cams == 1) render_ts(c, t_sync, L.img[0], L.img[0], 1, c->S2);
    else if (cams == 2) render_ts(c, t_sync, R.img[0], R.img[0], 1, c->S2 + c->P);
    return;
  }
  const PyrDesc& rl = c->raw[rawbuf][0].d;
  const PyrDesc& rr = c->raw[rawbuf][1].d;
  int nimg;
  const uint8_t *s0, *s1;
  uint8_t *d0, *d1;
  if (cams == 3) {
    render_ts(c, t_sync, rl.img[0], rr.img[0], 2, c->S2);
    nimg = 2; s0 = px00(rl); s1 = px00(rr); d0 = px00(L); d1 = px00(R);
  } else if (cams == 1) {
    render_ts(c, t_sync, rl.img[0], rl.img[0], 1, c->S2);
    nimg = 1; s0 = s1 = px00(rl); d0 = d1 = px00(L);
  } else {
    render_ts(c, t_sync, rr.img[0], rr.img[0], 1, c->S2 + c->P);
    nimg = 1; s0 = s1 = px00(rr); d0 = d1 = px00(R);
  }
  for (int stage = 0; stage < 3; stage++) {
    ScopedKernel k(c, K_CLAHE, stage == 0 ? (uint64_t)c->P * nimg : (uint64_t)c->P * 2 * nimg);
    launch_clahe(cur_stream(c), s0, s1, rl.stride[0], d0, d1, L.stride[0], c->W, c->H, c->d_lut,
                 c->d_minmax, nimg, stage);
  }
}

void render_and_build(esvio_fe_ctx* c, double t_sync, int slotL, int slotR, int rawbuf) {
  PyrDesc two[2] = {c->pyr[slotL].d, c->pyr[slotR].d};
  const bool fused = c->fuse_ts_pyr && !c->cfg.equalize && c->cfg.median_blur_kernel_size <= 0 &&
                     two[0].levels == 3 && two[1].levels == 3;
  if (!fused) {
    render_lk_images(c, t_sync, 3, slotL, slotR, rawbuf);
    pyr_build(c, two, 2);
    return;
  }
  {
    uint64_t px = 0;
    for (int l = 0; l <= 3; l++) px += (uint64_t)two[0].w[l] * two[0].h[l];
    ScopedKernel k(c, K_TIME_SURFACE, ((uint64_t)c->P * 16 + px) * 2);
    launch_ts_pyr(cur_stream(c), c->S2, t_sync, c->cfg.decay_ms / 1000.0, c->cfg.ignore_polarity, two);
  }
  {
    uint64_t all = 0;
    for (int l = 0; l <= 3; l++) all += (uint64_t)two[0].w[l] * two[0].h[l];
    ScopedKernel k(c, K_SCHARR, all * 5 * 2);
    launch_pad_scharr(cur_stream(c), two, 2);
  }
}

const PyrDesc& raw_ts_desc(const esvio_fe_ctx* c, int cam) {
  if (c->cfg.equalize) return c->raw[c->raw_cur][cam].d;
  return cam ? c->pyr[c->slot_curR].d : c->pyr[c->slot_curL].d;
}

LkArgs make_lk(const PyrDesc& P, const PyrDesc& N, const float2* prev, const float2* init,
               float2* next, uint8_t* status, const int* n_ptr, int n_max, int max_level,
               int max_count, double eps, int flags) {
  LkArgs a;
  a.P = P;
  a.N = N;
  a.prev_pts = prev;
  a.init_pts = init ? init : next;
  a.next_pts = next;
  a.status = status;
  a.n_ptr = n_ptr;
  a.n_max = n_max;
  a.max_level = std::min(max_level, P.levels);
  // TermCriteria normalisation of calcOpticalFlowPyrLK [OpenCV]
  a.max_count = std::min(std::max(max_count, 0), 100);
  double e = std::min(std::max(eps, 0.), 10.);
  a.eps2 = e * e;
  a.flags = flags;
  return a;
}

// forward call (+ optional backward call fused into the same launch)
void run_lk(esvio_fe_ctx* c, const LkArgs& f, const LkArgs* b, float2* back_pts,
            uint8_t* back_status) {
  uint64_t bytes = (uint64_t)f.n_max * (f.max_level + 1) * kLkWin * kLkWin * 5;
  if (b) bytes += (uint64_t)f.n_max * (b->max_level + 1) * kLkWin * kLkWin * 5;
  ScopedKernel k(c, K_LK, bytes);
  launch_lk(cur_stream(c), f, b, back_pts, back_status);
}

int copy_level0_out(esvio_fe_ctx* c, const PyrDesc& d, uint8_t* out) {
  const int stride = d.stride[0];
  HIPCHK(c, hipMemcpy2DAsync(out, d.w[0], d.img[0] + (size_t)kPad * stride + kPad, stride, d.w[0],
                             d.h[0], hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  return 0;
}

int copy_level0_in(esvio_fe_ctx* c, const PyrDesc& d, const uint8_t* in) {
  const int stride = d.stride[0];
  HIPCHK(c, hipMemcpy2DAsync(d.img[0] + (size_t)kPad * stride + kPad, stride, in, d.w[0], d.w[0],
                             d.h[0], hipMemcpyHostToDevice, cur_stream(c)));
  return 0;
}

// ---------------------------------------------------------------- host bookkeeping (reference
// helpers in feature_tracker.cpp)
template <class T>
void reduce_vector(std::vector<T>& v, const std::vector<uint8_t>& status) {  // :56-81
  int j = 0;
  for (int i = 0; i < int(v.size()); i++)
    if (status[i]) v[j++] = v[i];
  v.resize(j);
}

bool in_border_event(const esvio_fe_ctx* c, const P2f& pt) {  // :48-54
  const int BORDER_SIZE = 1;
  const int img_x = host::cv_round(pt.x), img_y = host::cv_round(pt.y);
  return BORDER_SIZE <= img_x && img_x < c->W - BORDER_SIZE && BORDER_SIZE <= img_y &&
         img_y < c->H - BORDER_SIZE;
}

double pt_distance(const P2f& a, const P2f& b) {  // :1314-1319
  const double dx = a.x - b.x, dy = a.y - b.y;
  return std::sqrt(dx * dx + dy * dy);
}

// Event_setMask (:123-151): std::sort on the same element type/comparator as the reference so
// the (unstable) permutation of equal track counts is inherited from libstdc++.
void event_set_mask(esvio_fe_ctx* c) {
  c->mask_event.reset(c->W, c->H);
  // (the sort only ever compares .first, so carrying src_idx along as payload leaves the
  // permutation — std::sort is not stable — exactly what it is for the reference's pair type)
  struct Item {
    int first;
    std::pair<P2f, int> second;
    int src;
  };
  std::vector<Item> cnt_pts_id;
  cnt_pts_id.reserve(c->cur_pts.size());
  for (unsigned int i = 0; i < c->cur_pts.size(); i++)
    cnt_pts_id.push_back(Item{c->track_cnt[i], std::make_pair(c->cur_pts[i], c->ids[i]), c->src_idx[i]});
  std::sort(cnt_pts_id.begin(), cnt_pts_id.end(),
            [](const Item& a, const Item& b) { return a.first > b.first; });
  c->cur_pts.clear();
  c->ids.clear();
  c->track_cnt.clear();
  c->src_idx.clear();
  for (auto& it : cnt_pts_id) {
    const int px = host::cv_round(it.second.first.x), py = host::cv_round(it.second.first.y);
    if (px < 0 || px >= c->W || py < 0 || py >= c->H) continue;  // cannot happen after inBorder
    if (!c->mask_event.test(px, py)) {
      c->cur_pts.push_back(it.second.first);
      c->ids.push_back(it.second.second);
      c->track_cnt.push_back(it.first);
      c->src_idx.push_back(it.src);
      c->mask_event.stamp_disc(px, py, c->cfg.min_dist, c->hw);
    }
  }
}

std::vector<P2f> undistorted_pts(const std::vector<P2f>& pts, const esvio_fe_camera& cam) {  // :991
  const size_t n = pts.size();
  std::vector<P2f> un(n);
  if (!n) return un;
  std::vector<double> lx(n), ly(n);
  host::lift_projective_batch(cam, &pts[0].x, (int)n, lx.data(), ly.data());
  for (size_t i = 0; i < n; i++) un[i] = P2f{(float)lx[i], (float)ly[i]};  // b[2] == 1.0
  return un;
}

// ptsVelocity (:1004-1045) incl. its quirk: with no previous map the result is sized by the LEFT
// cur_pts whichever camera it is called for.
// (dt = cur_time - prev_time and the left point count of the frame the call belongs to are passed
// in: the right-camera tail of a frame may run during the next call, see finalize_right.)
std::vector<P2f> pts_velocity_fn(std::vector<int>& ids, std::vector<P2f>& pts, IdMap& cur_id_pts,
                                 IdMap& prev_id_pts, double dt, size_t n_left) {
  std::vector<P2f> vel;
  cur_id_pts.build(ids, pts);
  if (!prev_id_pts.empty()) {
    vel.reserve(pts.size());
    for (unsigned int i = 0; i < pts.size(); i++) {
      const P2f* prev = ids[i] != -1 ? prev_id_pts.find(ids[i]) : nullptr;
      if (prev) {
        const double v_x = (pts[i].x - prev->x) / dt;
        const double v_y = (pts[i].y - prev->y) / dt;
        vel.push_back(P2f{(float)v_x, (float)v_y});
      } else {
        vel.push_back(P2f{0, 0});
      }
    }
  } else {
    vel.assign(n_left, P2f{0, 0});
  }
  return vel;
}

void reject_with_f_event(esvio_fe_ctx* c) {  // :910-947
  if (c->cur_pts.size() >= 8) {
    const esvio_fe_camera& cam = c->cfg.cam[0];
    const double FOCAL = c->cfg.focal_length;
    const size_t n = c->prev_pts.size();
    std::vector<float> un_cur(n * 2), un_prev(n * 2);
    std::vector<double> lx(n), ly(n);
    const double cx = c->W / 2.0, cy = c->H / 2.0;
    const auto tl = std::chrono::steady_clock::now();
    host::lift_projective_batch(cam, &c->prev_pts[0].x, (int)n, lx.data(), ly.data());
    for (size_t i = 0; i < n; i++) {  // p[2] == 1.0: x / 1.0 is exact
      un_prev[2 * i] = (float)(FOCAL * lx[i] / 1.0 + cx);
      un_prev[2 * i + 1] = (float)(FOCAL * ly[i] / 1.0 + cy);
    }
    host::lift_projective_batch(cam, &c->cur_pts[0].x, (int)n, lx.data(), ly.data());
    for (size_t i = 0; i < n; i++) {
      un_cur[2 * i] = (float)(FOCAL * lx[i] / 1.0 + cx);
      un_cur[2 * i + 1] = (float)(FOCAL * ly[i] / 1.0 + cy);
    }
    std::vector<uint8_t> status(c->cur_pts.size());
    const auto t0 = std::chrono::steady_clock::now();
    host::find_fundamental_mat(un_prev.data(), un_cur.data(), (int)c->cur_pts.size(),
                               c->cfg.f_threshold, 0.99, status.data(), c->pool);
    if (c->trace) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      c->tr_fm_ms += ms;
      c->tr_fm_max_ms = std::max(c->tr_fm_max_ms, ms);
      c->tr_lift_ms += std::chrono::duration<double, std::milli>(t0 - tl).count();
    }
    reduce_vector(c->prev_pts, status);
    reduce_vector(c->cur_pts, status);
    reduce_vector(c->ids, status);
    reduce_vector(c->track_cnt, status);
    reduce_vector(c->src_idx, status);
  }
}

// device result block (and its pinned mirror): set 1 = temporal LK, then stereo LK of the temporal
// survivors; set 2 = stereo LK of the newly selected corners
struct ResLayout {
  size_t B1[2], C1[2], SA1[2], SB1[2], A[2], CNT, NEW, B2, C2, SA2, SB2, total;
};

ResLayout res_layout(size_t M) {
  const size_t stM = (M + 63) / 64 * 64;
  ResLayout L;
  size_t o = 0;
  for (int s = 0; s < 2; s++) {
    L.B1[s] = o;  o += M * 8;
    L.C1[s] = o;  o += M * 8;
    L.SA1[s] = o; o += stM;
    L.SB1[s] = o; o += stM;
    L.A[s] = o;   o += M * 8;
  }
  L.CNT = o; o += 64;
  L.NEW = o; o += M * 8;
  L.B2 = o;  o += M * 8;
  L.C2 = o;  o += M * 8;
  L.SA2 = o; o += stM;
  L.SB2 = o; o += stM;
  L.total = o;
  return L;
}

// pinned staging: a mirror of the device result block (D2H) + upload areas (H2D)
struct Pin {
  float2 *ptsB, *ptsC;    // set 1 (the copy asked for)
  uint8_t *stA, *stB;
  int* counts;            // [16]
  float2* news;           // [kept points (as uploaded) | newly selected corners]
  float2 *ptsB2, *ptsC2;  // set 2
  uint8_t *stA2, *stB2;
  float2* A;              // LK input points (read by the kernels in place)
  uint32_t* mask;         // H2D H*wpr words
};

Pin pin_of(esvio_fe_ctx* c, int set = 0) {
  const size_t M = std::max(c->cfg.max_cnt, 1);
  const ResLayout L = res_layout(M);
  Pin p;
  uint8_t* b = c->h_pin;
  p.ptsB = (float2*)(b + L.B1[set]);
  p.ptsC = (float2*)(b + L.C1[set]);
  p.stA = b + L.SA1[set];
  p.stB = b + L.SB1[set];
  p.counts = (int*)(b + L.CNT);
  p.news = (float2*)(b + L.NEW);
  p.A = (float2*)(b + L.A[set]);
  p.ptsB2 = (float2*)(b + L.B2);
  p.ptsC2 = (float2*)(b + L.C2);
  p.stA2 = b + L.SA2;
  p.stB2 = b + L.SB2;
  b += (L.total + 255) / 256 * 256;
  p.mask = (uint32_t*)b;
  return p;
}

// device-side address of a location inside the pinned block
template <typename T>
T* zdev(esvio_fe_ctx* c, T* host) {
  return (T*)(c->z_res + ((uint8_t*)host - c->h_pin));
}

size_t pin_bytes(const esvio_fe_config& cfg) {
  const size_t M = std::max(cfg.max_cnt, 1);
  const ResLayout L = res_layout(M);
  return (L.total + 255) / 256 * 256 +
         (size_t)cfg.height * ((cfg.width + 31) / 32) * 4 + 256;
}

void clear_tracker_state(esvio_fe_ctx* c) {
  c->prev_pts.clear();
  c->cur_pts.clear();
  c->cur_right_pts.clear();
  c->n_pts.clear();
  c->cur_un_pts.clear();
  c->cur_un_right_pts.clear();
  c->pts_velocity.clear();
  c->right_pts_velocity.clear();
  c->ids.clear();
  c->ids_right.clear();
  c->track_cnt.clear();
  c->track_cnt_right.clear();
  c->cur_un_pts_map.clear();
  c->prev_un_pts_map.clear();
  c->cur_un_right_pts_map.clear();
  c->prev_un_right_pts_map.clear();
  c->have_img = false;
  c->slot_prevL = c->slot_curL = 0;
  c->slot_curR = kLeftSlots;
  c->ext_right_pending = false;
  c->ext_sae_pending = false;
  c->cur_time = c->prev_time = 0;
}

SelectArgs make_select_args(esvio_fe_ctx* c, int set, int max_corners, float2* out_pts, int out_base,
                            int32_t* out_idx) {
  SelectArgs s{};
  s.comp_xy = c->cand[set].comp_xy;
  s.comp_idx = c->cand[set].comp_idx;
  s.total = c->cand[set].total;
  s.W = c->W;
  s.H = c->H;
  s.wpr = (c->W + 31) / 32;
  s.max_corners = max_corners;
  s.radius = c->cfg.min_dist;
  for (int i = 0; i <= kMaxDiscR; i++) s.hw[i] = i < (int)c->hw.size() ? (int8_t)c->hw[i] : -1;
  s.disc_c = c->disc_tab_only ? -1 : disc_threshold(s.hw, s.radius);
  s.out_pts = out_pts;
  s.out_idx = out_idx;
  s.out_base = out_base;
  s.n_out = c->d_counts;
  s.n_total = c->d_counts + 1;
  s.host_counts = nullptr;
  s.init_bits = nullptr;
  s.pub_slots = nullptr;
  s.pub_done = nullptr;
  s.pub_seq = 0;
  return s;
}

size_t select_lds_bytes(const esvio_fe_ctx* c) {
  // bitmap + half-width table + the kept points whose discs seed the bitmap
  return ((size_t)c->H * ((c->W + 31) / 32) + 4 + 64 + (size_t)std::max(c->cfg.max_cnt, 1)) * 4;
}

// ordered compaction of candidate set `set` (right behind the k_arc that filled it)
void run_compact(esvio_fe_ctx* c, uint32_t n_events, int set) {
  const uint32_t nblk = (n_events + kArcBlock - 1) / kArcBlock;
  const esvio_fe_ctx::CandSet& cs = c->cand[set];
  ScopedKernel k(c, K_COMPACT, 0);
  launch_compact(cur_stream(c), cs.xy, cs.idx, cs.cnt, nblk, cs.comp_xy, cs.comp_idx, cs.total);
}

// the sequential greedy (Event_FeaturesToTrack) over the compacted candidates of set `set`;
// `mask_bits`: blocked pixels the disc bitmap starts from (null: none, or already applied by k_arc)
void run_select(esvio_fe_ctx* c, int set, int max_corners, float2* out_pts, int out_base,
                int32_t* out_idx, const uint32_t* mask_bits = nullptr, int* host_counts = nullptr,
                bool publish = false, const float2* stamp_pts = nullptr, int n_stamp = 0) {
  SelectArgs s = make_select_args(c, set, max_corners, out_pts, out_base, out_idx);
  s.host_counts = host_counts;
  s.init_bits = mask_bits;
  s.stamp_pts = stamp_pts;
  s.n_stamp = n_stamp;
  if (publish) {
    s.pub_slots = c->d_pub_slots;
    s.pub_done = c->d_pub_done;
    s.pub_seq = c->pub_seq;
  }
  ScopedKernel k(c, K_SELECT, 0);
  launch_select(cur_stream(c), s, select_lds_bytes(c));
}

// Arc* flags (+ ordered per-block candidate lists into set `set`) for the left events; `ts` is the
// RAW left time surface the TS_LK_THRESHOLD test reads (null: no test)
void run_arc(esvio_fe_ctx* c, const EventRec* ev, uint32_t n, const PyrDesc* ts, bool use_mask,
             bool want_flags, bool want_cand, int set, bool marked = false) {
  ArcArgs a{};
  a.ev = ev;
  a.n = n;
  a.L2 = c->L2;
  a.S2 = c->S2;
  a.W = c->W;
  a.H = c->H;
  a.filter_threshold = c->cfg.feature_filter_threshold;
  a.border = c->cfg.min_dist + 1;
  a.ts = ts ? ts->img[0] : nullptr;  // RAW left time surface (:26)
  a.ts_stride = ts ? ts->stride[0] : 0;
  a.ts_lk_threshold = c->cfg.ts_lk_threshold;
  a.mask_bits = use_mask ? c->d_mask_bits : nullptr;
  a.wpr = (c->W + 31) / 32;
  a.flags = want_flags ? c->d_flags : nullptr;
  a.cand_xy = want_cand ? c->cand[set].xy : nullptr;
  a.cand_idx = want_cand ? c->cand[set].idx : nullptr;
  a.cand_cnt = want_cand ? c->cand[set].cnt : nullptr;
  // Worth it for batches of the usual size (k_select, on the frame's device chain, sees half the
  // candidates: 54 against 57 us at 0.17 M left events, the atomics and k_dedup run on the prefetch
  // stream); at 3.3 M left events the 0.8 M atomics cost k_arc_ev 43 us and save k_select 8.
  const bool dedup = want_cand && c->dedup_enabled && c->d_first[set] && n < (1u << 20);
  if (dedup) {
    // keys count down from launch to launch: 0xfe.. for the first, 0x01.. for the 254th, then the
    // map is cleared (to all ones) and the count starts again
    const uint32_t e = c->first_epoch[set]++ % 254u;
    if (e == 0)
      (void)hipMemsetAsync(c->d_first[set], 0xff, (size_t)c->P * 4, cur_stream(c));
    a.first_map = c->d_first[set];
    a.first_key = (254u - e) << 24;
  }
  a.cmap = c->d_cmap[set];
  a.touched = c->d_touched[set];
  {
    // the events' x,y,p once more (16 B records) -> touched bits; then per touched pair its 16/20
    // ring values (counted once per pixel: 16 B) + {L0,L1}
    ScopedKernel k(c, K_ARC_MAP, (uint64_t)n * 16 + (uint64_t)c->P * 32);
    if (!marked) launch_arc_mark(cur_stream(c), a);  // (else: done by the SAE update's first pass)
    launch_arc_map(cur_stream(c), a);
  }
  {
    ScopedKernel k(c, K_ARC, (uint64_t)n * 16);
    launch_arc(cur_stream(c), a);
  }
  if (dedup) {
    ScopedKernel k(c, K_COMPACT, 0);
    launch_dedup(cur_stream(c), a.cand_xy, a.cand_idx, a.cand_cnt, (n + kArcBlock - 1) / kArcBlock,
                 a.first_map, a.first_key, c->W);
  }
}

// wait for the main stream with a short busy poll first: the two per-frame host syncs are on the
// critical path and an interrupt-driven hipStreamSynchronize wakes up tens of microseconds late
hipError_t sync_main(esvio_fe_ctx* c) {
  for (int i = 0; i < 20000; i++) {
    const hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) return hipSuccess;
    if (e != hipErrorNotReady) return e;
  }
  return hipStreamSynchronize(c->stream);
}

hipError_t sync_event(hipEvent_t ev) {
  for (int i = 0; i < 20000; i++) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return hipSuccess;
    if (e != hipErrorNotReady) return e;
  }
  return hipEventSynchronize(ev);
}

// ---------------------------------------------------------------- next-batch prefetch
// Enqueue the SAE update, time surfaces and pyramids of the batch announced with
// esvio_fe_set_next_batch on the second stream; they overlap the rest of the current frame (stereo
// LK, selection) and the host work between calls.  Waits for ev_planes_free (recorded on the main
// stream once the current frame has finished reading the SAE planes) when `wait_planes`; a frame
// that itself came from the prefetch stream and runs no Arc* on the main stream reads neither the
// planes nor the raw surfaces there, so the next prefetch only has to follow its own stream.
// With the caller's PUB hint the Arc* pass of the batch runs here too (into the other candidate
// set), which takes it off the main stream's per-frame chain.
int prefetch_next(esvio_fe_ctx* c, bool wait_planes) {
  int rc = 0;
  StreamScope on_prefetch_stream(c->stream2);
  while (!rc && !c->announced.empty() && (int)c->inflight.size() < kPrefetchDepth) {
    Inflight b;
    static_cast<Batch&>(b) = c->announced.front();
    // resources nobody is using: not the current frame's, not another prefetched batch's
    auto taken = [&](int Inflight::*m, int v) {
      for (const Inflight& o : c->inflight)
        if (o.*m == v) return true;
      return false;
    };
    b.lane = 0;
    while (taken(&Inflight::lane, b.lane)) b.lane++;
    b.slotL = 0;
    while (b.slotL == c->slot_prevL || b.slotL == c->slot_curL || taken(&Inflight::slotL, b.slotL))
      b.slotL++;
    b.slotR = kLeftSlots;
    while (b.slotR == c->slot_curR || taken(&Inflight::slotR, b.slotR)) b.slotR++;
    b.raw = 0;
    while (b.raw == c->raw_cur || taken(&Inflight::raw, b.raw)) b.raw++;
    b.cand = 0;
    while (b.cand == c->cand_cur || taken(&Inflight::cand, b.cand)) b.cand++;
    do {
      if (wait_planes && hipStreamWaitEvent(c->stream2, c->ev_planes_free, 0) != hipSuccess) {
        rc = fail(c, ESVIO_FE_EHIP, "hipStreamWaitEvent failed");
        break;
      }
      wait_planes = false;  // later batches simply follow on the same stream
      // the right-camera pyramid slot this batch gets may be the one an earlier frame's stereo LK
      // (stream4) still reads — in lazy mode nobody has waited for that launch yet
      if (c->lks_last >= 0 &&
          hipStreamWaitEvent(c->stream2, c->ev_lks_done[c->lks_last], 0) != hipSuccess) {
        rc = fail(c, ESVIO_FE_EHIP, "hipStreamWaitEvent failed");
        break;
      }
      if ((rc = stage_events(c, b.left, b.nL, b.right, b.nR, b.space, &b.dL, &b.dR, b.lane))) break;
      // the ~11 dependent launches up to the pyramids go out as one graph (fe_kernels.h); with the
      // per-kernel timers on they are launched one by one so that each can be bracketed
      const bool as_graph = c->graphs_enabled && !c->prof_on;
      if (as_graph) {
        c->rec.clear();
        set_launch_recorder(&c->rec);
      }
      bool arc_marked = false;
      rc = sae_update(c, b.dL, (uint32_t)b.nL, b.dR, (uint32_t)b.nR, nullptr, nullptr, nullptr,
                      b.pub && b.nL ? b.cand : -1, &arc_marked);
      if (!rc) {
        render_and_build(c, b.time, b.slotL, b.slotR, b.raw);
        if (record_event(c->ev_lane_done[b.lane], c->stream2) != hipSuccess)
          rc = fail(c, ESVIO_FE_EHIP, "hipEventRecord failed");
      }
      if (as_graph) {
        set_launch_recorder(nullptr);
        if (!rc && launch_as_graph(c->pf_graph, c->rec, c->stream2) != hipSuccess) {
          // (not expected; the plain path still works)
          (void)hipGetLastError();
          c->graphs_enabled = false;
          destroy_launch_graph(c->pf_graph);
          if (launch_plain(c->rec, c->stream2) != hipSuccess)
            rc = fail(c, ESVIO_FE_EHIP, "kernel launch failed");
        }
      }
      if (rc) break;
      b.arc_done = false;
      if (b.pub && b.nL) {
        if ((rc = ensure_cand_capacity(c, b.cand, b.nL))) break;
        const PyrDesc& ts = c->cfg.equalize ? c->raw[b.raw][0].d : c->pyr[b.slotL].d;
        run_arc(c, b.dL, (uint32_t)b.nL, &ts, false, false, true, b.cand, arc_marked);
        run_compact(c, (uint32_t)b.nL, b.cand);
        if (hipEventRecord(c->ev_lane_arc[b.lane], c->stream2) != hipSuccess) {
          rc = fail(c, ESVIO_FE_EHIP, "hipEventRecord failed");
          break;
        }
        b.arc_done = true;
      }
      c->inflight.push_back(b);
      c->announced.pop_front();
    } while (0);
  }
  return rc;
}

// Launch the NEXT frame's temporal forward/backward LK (feature_tracker.cpp:410,417 of the next
// call) now: its inputs are final once this frame's kept points (written to z_new[0..n_kept)) and
// new corners (written by k_select behind them, total count in d_counts[1]) are known, and the next
// frame's pyramids are already being built on the prefetch stream.
int enqueue_spec_temporal(esvio_fe_ctx* c, const Inflight& nxt /* the next frame's batch */,
                          int n_kept, bool with_new) {
  const size_t M = std::max(c->cfg.max_cnt, 1);
  const size_t stM = (M + 63) / 64 * 64;
  // (kept points: already in host memory; new corners: published one by one by the k_select that
  // has just been launched — the waves of points >= n_kept wait for their slot)
  HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_lane_done[nxt.lane], 0));
  float2* B = (float2*)c->z_spec;  // results land in the pinned block itself
  float2* Cb = B + M;
  uint8_t* sA = c->z_spec + M * 16;
  uint8_t* sB = sA + stM;
  const PyrDesc& P = c->pyr[c->slot_curL].d;
  const PyrDesc& N = c->pyr[nxt.slotL].d;
  const int n_max = with_new ? (int)M : n_kept;
  LkArgs f = make_lk(P, N, c->z_new, nullptr, B, sA, nullptr, n_max, 3, 30, 0.01, 0);
  LkArgs b = make_lk(N, P, nullptr, nullptr, nullptr, nullptr, nullptr, n_max, 1, 30, 0.01,
                     ESVIO_FE_LK_USE_INITIAL_FLOW);
  if (with_new) {
    f.poll_slots = c->d_pub_slots;
    f.poll_done = c->d_pub_done;
    f.poll_seq = c->pub_seq;
    f.poll_from = n_kept;
    f.poll_err = (int*)(c->z_spec + M * 16 + 2 * stM);
  }
  // the frame after next, chained to this launch point by point (see esvio_fe_ctx::d_chain)
  const Inflight* nxt2 = nullptr;
  if (c->chain_enabled && !nxt.pub && c->inflight.size() >= 2 && c->inflight[0].lane == nxt.lane &&
      !c->chain_valid)
    nxt2 = &c->inflight[1];
  if (nxt2) {
    c->chain_seq = (c->chain_seq + 1) & 0x3fffffffu;
    if (!c->chain_seq) c->chain_seq = 1;
    f.chain_out = c->d_chain;
    f.chain_seq = c->chain_seq;
  }
  {
    StreamScope on_spec_stream(c->stream3);
    run_lk(c, f, c->cfg.flow_back ? &b : nullptr, Cb, sB);
  }
  HIPCHK(c, hipEventRecord(c->ev_spec_done, c->stream3));
  c->spec_valid = true;
  if (nxt2) {
    HIPCHK(c, hipStreamWaitEvent(c->stream4, c->ev_lane_done[nxt.lane], 0));
    HIPCHK(c, hipStreamWaitEvent(c->stream4, c->ev_lane_done[nxt2->lane], 0));
    uint8_t* zc = c->z_spec + c->spec_bytes;
    const PyrDesc& N2 = c->pyr[nxt2->slotL].d;
    LkArgs f2 = make_lk(N, N2, nullptr, nullptr, (float2*)zc, zc + M * 16, nullptr, n_max, 3, 30, 0.01, 0);
    LkArgs b2 = make_lk(N2, N, nullptr, nullptr, nullptr, nullptr, nullptr, n_max, 1, 30, 0.01,
                        ESVIO_FE_LK_USE_INITIAL_FLOW);
    f2.chain_in = c->d_chain;
    f2.chain_seq = c->chain_seq;
    f2.poll_err = (int*)(zc + M * 16 + 2 * stM);
    {
      StreamScope on_chain_stream(c->stream4);
      run_lk(c, f2, c->cfg.flow_back ? &b2 : nullptr, (float2*)zc + M, zc + M * 16 + stM);
    }
    HIPCHK(c, hipEventRecord(c->ev_chain_done, c->stream4));
    c->chain_valid = true;
    c->tr_chain_launch++;
    c->chain_for = c->frame_no + 2;
    c->chain_map_ok = false;
  }
  return 0;
}

// give up a chained launch whose results cannot be used (its kernel only waits for bounded times)
int cancel_chain(esvio_fe_ctx* c) {
  if (!c->chain_valid) return 0;
  c->chain_valid = false;
  c->chain_map_ok = false;
  c->tr_chain_cancel++;
  HIPCHK(c, hipStreamSynchronize(c->stream4));
  return 0;
}

// The right-camera tail of trackEvent (:475-575) for the first n points of a frame (all of them, or
// only the kept ones in lazy mode): the stereo LK results of the kept points are in set 1 (by
// survivor index, src == nullptr: identity), those of the new corners in set 2.  n_left = the
// frame's left point count (ptsVelocity's sizing quirk).
void right_tail(esvio_fe_ctx* c, const Pin& pin, const P2f* left, const int* ids, const int* src,
                int n, int n_kept, double dt, size_t n_left) {
  const esvio_fe_config& cfg = c->cfg;
  c->ids_right.clear();
  c->cur_right_pts.clear();
  c->cur_un_right_pts.clear();
  c->right_pts_velocity.clear();
  c->cur_un_right_pts_map.clear();
  c->track_cnt_right.clear();
  if (n_left) {
    // gather the stereo results: kept points from set 1, new ones from set 2
    std::vector<uint8_t> status(n), statusRightLeft(n);
    std::vector<P2f> reverseLeftPts(n);
    c->cur_right_pts.resize(n);
    const P2f *B1 = (const P2f*)pin.ptsB, *C1 = (const P2f*)pin.ptsC;
    const P2f *B2 = (const P2f*)pin.ptsB2, *C2 = (const P2f*)pin.ptsC2;
    for (int i = 0; i < n; i++) {
      if (i < n_kept) {
        const int j = src ? src[i] : i;
        c->cur_right_pts[i] = B1[j];
        status[i] = pin.stA[j];
        reverseLeftPts[i] = C1[j];
        statusRightLeft[i] = pin.stB[j];
      } else {
        const int j = i - n_kept;
        c->cur_right_pts[i] = B2[j];
        status[i] = pin.stA2[j];
        reverseLeftPts[i] = C2[j];
        statusRightLeft[i] = pin.stB2[j];
      }
    }
    if (cfg.flow_back && !c->cur_right_pts.empty()) {
      for (int i = 0; i < n; i++) {
        if (status[i] && statusRightLeft[i] && in_border_event(c, c->cur_right_pts[i]) &&
            pt_distance(left[i], reverseLeftPts[i]) <= 0.5)
          status[i] = 1;
        else
          status[i] = 0;
      }
    }
    c->ids_right.assign(ids, ids + n);
    reduce_vector(c->cur_right_pts, status);
    reduce_vector(c->ids_right, status);
    c->track_cnt_right.assign(c->cur_right_pts.size(), 1);
    c->cur_un_right_pts = undistorted_pts(c->cur_right_pts, cfg.cam[1]);
    c->right_pts_velocity =
        pts_velocity_fn(c->ids_right, c->cur_un_right_pts, c->cur_un_right_pts_map,
                        c->prev_un_right_pts_map, dt, n_left);
  }
  // reference: prev = cur (copy); cur is cleared before its next use in ptsVelocity, so a swap
  // is equivalent and avoids re-allocating ~300 map nodes per frame
  c->prev_un_right_pts_map.swap(c->cur_un_right_pts_map);
}

// Lazy mode: the right-camera tail of the previous call's frame, which published nothing and
// returned with its stereo LK still in flight.
int finalize_right(esvio_fe_ctx* c) {
  if (!c->pend_right.active) return 0;
  esvio_fe_ctx::PendingRight& pr = c->pend_right;
  pr.active = false;
  const int n = (int)pr.left.size();
  if (n) HIPCHK(c, sync_event(c->ev_lks_done[pr.set]));
  if (pin_of(c).counts[3] != 0) return fail(c, ESVIO_FE_EINTERNAL, "radix sort look-back spin expired");
  right_tail(c, pin_of(c, pr.set), pr.left.data(), pr.ids.data(), nullptr, n, n, pr.dt, (size_t)n);
  return 0;
}

// Lazy mode: append the right-camera entries of the corners the previous published frame detected
// (their stereo LK has run meanwhile).  Equal to what the eager tail would have produced: the new
// ids are the largest, come last in every vector, are absent from the previous frame's map (zero
// velocity, feature_tracker.cpp:1026-1040) and extend the (sorted) map the next frame reads.
int finalize_pending(esvio_fe_ctx* c) {
  if (!c->pend.active) return 0;
  c->pend.active = false;
  HIPCHK(c, sync_event(c->ev_lknew_done));
  Pin pin = pin_of(c);
  const esvio_fe_config& cfg = c->cfg;
  const P2f *B2 = (const P2f*)pin.ptsB2, *C2 = (const P2f*)pin.ptsC2;
  std::vector<P2f> add;
  std::vector<int> add_ids;
  for (size_t j = 0; j < c->pend.ids.size(); j++) {
    bool ok = pin.stA2[j] != 0;
    if (cfg.flow_back)
      ok = ok && pin.stB2[j] && in_border_event(c, B2[j]) && pt_distance(c->pend.left[j], C2[j]) <= 0.5;
    if (ok) {
      add.push_back(B2[j]);
      add_ids.push_back(c->pend.ids[j]);
    }
  }
  if (add.empty()) return 0;
  const std::vector<P2f> un = undistorted_pts(add, cfg.cam[1]);
  for (size_t j = 0; j < add.size(); j++) {
    c->ids_right.push_back(add_ids[j]);
    c->cur_right_pts.push_back(add[j]);
    c->cur_un_right_pts.push_back(un[j]);
    c->track_cnt_right.push_back(1);
    if (!c->pend.prev_map_was_empty) c->right_pts_velocity.push_back(P2f{0, 0});
    c->prev_un_right_pts_map.v.emplace_back(add_ids[j], un[j]);  // (already swapped: next frame's prev)
  }
  return 0;
}

// ---------------------------------------------------------------- trackEvent
int track_event_impl(esvio_fe_ctx* c, double _cur_time, const esvio_fe_event* left, size_t nL,
                     const esvio_fe_event* right, size_t nR, int space, bool PUB_THIS_FRAME,
                     const esvio_fe_motion* motion = nullptr) {
  const esvio_fe_config& cfg = c->cfg;
  const int M = cfg.max_cnt;
  // set 1 alternates between its two copies: the previous frame's stereo LK may still be in flight
  // (lazy mode, pend_right) while this frame's kernels are enqueued
  c->res_set ^= 1;
  c->frame_no++;
  Pin pin = pin_of(c, c->res_set);
  if (PUB_THIS_FRAME && c->pool) host::ransac_pool_wake(c->pool);
  c->cur_time = _cur_time;
  using clk = std::chrono::steady_clock;
  auto tp = clk::now();
  auto lap = [&](int i) {
    if (!c->trace) return;
    auto now = clk::now();
    c->phase_ms[PUB_THIS_FRAME ? 1 : 0][i] += std::chrono::duration<double, std::milli>(now - tp).count();
    tp = now;
  };

  const EventRec *dL = nullptr, *dR = nullptr;
  const bool first = !c->have_img;
  bool arc_done = false, arc_prefetched = false, arc_marked_main = false;
  int arc_lane = 0;
  if (!c->inflight.empty()) {
    // this batch was announced with esvio_fe_set_next_batch and its SAE update, images and
    // pyramids were enqueued on the prefetch stream during an earlier call
    const Inflight b = c->inflight.front();
    if (left != b.left || nL != b.nL || right != b.right || nR != b.nR || space != b.space ||
        _cur_time != b.time || motion)
      return fail(c, ESVIO_FE_EINVAL, "batch differs from the one given to esvio_fe_set_next_batch");
    if (PUB_THIS_FRAME && !b.arc_done && c->inflight.size() > 1)
      return fail(c, ESVIO_FE_EINVAL,
                  "PUB hint was 0 for a published frame and a later batch is already applied to "
                  "the SAE: with more than one batch announced the hint must be exact");
    c->inflight.pop_front();
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_lane_done[b.lane], 0));
    c->tr_lane = b.lane;
    dL = b.dL;
    dR = b.dR;
    c->slot_curL = b.slotL;
    c->slot_curR = b.slotR;
    c->raw_cur = b.raw;
    c->cur_prefetched = true;
    if (b.arc_done) {  // candidates of this batch are in its own set
      c->cand_cur = b.cand;
      arc_lane = b.lane;
      arc_done = arc_prefetched = true;
    }
  } else {
    c->cur_prefetched = false;
    if (int rc = stage_events(c, left, nL, right, nR, space, &dL, &dR)) return rc;
    // createSAE_left / createSAE_right loops (:356-362), or their motion-compensated forms (:627-641)
    if (c->ext_sae_pending) {
      // esvio_fe_sae_slice_commit has put this batch into the planes already (its SAE update ran
      // time-sliced over several GPUs); the events are still needed below for Arc*
      if (motion) return fail(c, ESVIO_FE_EINVAL, "time-sliced SAE update has no motion-compensated form");
      c->ext_sae_pending = false;
    } else if (motion) {
      esvio_fe_event first_ev;
      if (int rc = first_event_host(c, left, space, &first_ev)) return rc;
      const McParams mc = make_mc_params(motion, first_ev);
      if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR, &mc)) return rc;
    } else if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR, nullptr, nullptr, nullptr,
                                   PUB_THIS_FRAME ? c->cand_cur : -1, &arc_marked_main)) {
      return rc;
    }
    // SAEtoTimeSurface_left/right(cur_time) (:367-368) -> cur images; slot rotation replaces the
    // cv::Mat header swaps of :390-403,:585.  Left slots 0..2: {prev, cur, free}.
    int sl = 0;
    while (!first && (sl == c->slot_prevL || sl == c->slot_curL)) sl++;
    c->slot_curL = sl;
    // camera split: the right image was imported into slot_curR by esvio_fe_import_image
    if (!c->ext_right_pending) c->slot_curR = c->slot_curR == kLeftSlots ? kLeftSlots + 1 : kLeftSlots;
    c->raw_cur = (c->raw_cur + 1) % kRightSlots;
    if (c->ext_right_pending) {
      render_lk_images(c, c->cur_time, 1, c->slot_curL, c->slot_curR, c->raw_cur);
      PyrDesc cur2[2] = {c->pyr[c->slot_curL].d, c->pyr[c->slot_curR].d};
      pyr_build(c, cur2, 2);
    } else {
      render_and_build(c, c->cur_time, c->slot_curL, c->slot_curR, c->raw_cur);
    }
    c->ext_right_pending = false;
  }
  // the next frame's batch, if it is already in flight (two announced ahead), else once this
  // frame's early_work has put it there
  bool have_next = !c->inflight.empty();
  Inflight next_b = have_next ? c->inflight.front() : Inflight();
  const bool had_announced = !c->announced.empty();
  auto next_batch = [&]() -> const Inflight* {
    if (!have_next && had_announced) {
      if (!c->inflight.empty()) {
        next_b = c->inflight.front();
        have_next = true;
      }
    }
    return have_next ? &next_b : nullptr;
  };
  if (first) c->slot_prevL = c->slot_curL;  // prev_img_left = cur_img_left = img_left (:391)
  c->have_img = true;
  const PyrDesc& prevL = c->pyr[c->slot_prevL].d;
  const PyrDesc& curL = c->pyr[c->slot_curL].d;
  const PyrDesc& curR = c->pyr[c->slot_curR].d;
  // what THIS frame enqueues on the main stream that reads the SAE planes / raw time surfaces:
  // its own SAE update + rendering unless prefetched, and Arc* unless that ran with the prefetch
  bool main_reads_planes = !c->cur_prefetched;

  c->cur_pts.clear();
  c->cur_right_pts.clear();
  lap(0);

  // Arc* for every left event does not depend on the tracks: on published frames it is enqueued
  // now (behind the temporal LK) without the blocked-pixel mask, so it runs under the host-side
  // filtering / RANSAC / Event_setMask; the mask becomes k_select's initial bitmap.  After it
  // nothing of this frame reads the planes on the main stream, so the announced next batch is
  // started on the prefetch stream.
  bool early_done = false;
  auto early_work = [&]() -> int {
    if (early_done) return 0;
    early_done = true;
    if (PUB_THIS_FRAME && !arc_done) {
      if (int rc = ensure_arc_capacity(c, nL, c->cand_cur)) return rc;
      const PyrDesc ts = raw_ts_desc(c, 0);
      run_arc(c, dL, (uint32_t)nL, &ts, false, false, true, c->cand_cur, arc_marked_main);
      run_compact(c, (uint32_t)nL, c->cand_cur);
      arc_done = true;
      main_reads_planes = true;
    }
    if (!had_announced) return 0;
    if (main_reads_planes) HIPCHK(c, hipEventRecord(c->ev_planes_free, c->stream));
    return prefetch_next(c, main_reads_planes);
  };

  // a speculative launch of this very temporal LK may have been made by the previous call
  bool use_spec = false;
  if (c->spec_valid) {
    c->spec_valid = false;
    use_spec = c->cur_prefetched && (int)c->prev_pts.size() == c->spec_n;
    if (!use_spec) HIPCHK(c, hipStreamSynchronize(c->stream3));
  }
  // ... or a chained one by the call before that; if it was made for the NEXT frame, this frame
  // is the one in between: it must publish nothing and track with the speculative results
  bool use_chain = false;
  if (c->chain_valid && c->chain_for == c->frame_no) {
    use_chain = !use_spec && c->cur_prefetched && c->chain_map_ok &&
                c->chain_map.size() == c->prev_pts.size();
    if (!use_chain)
      if (int rc = cancel_chain(c)) return rc;
    c->chain_valid = false;
  } else if (c->chain_valid && (c->chain_for != c->frame_no + 1 || PUB_THIS_FRAME || !use_spec)) {
    if (int rc = cancel_chain(c)) return rc;
  }
  const bool chain_covers_next = c->chain_valid;  // (then: for frame_no + 1)
  const bool early_results = use_spec || use_chain;
  // When to enqueue the ~12 launches of the announced batch's prefetch (early_work):
  //  * before the wait for this frame's temporal LK when that is a speculative / chained launch
  //    still running and the frame publishes nothing: the host would only wait there;
  //  * late — a published frame whose successor is already in flight: after everything else of the
  //    frame, while the corner selection runs (RANSAC + mask + selection sit behind the temporal
  //    LK wait, so nothing is put in front of them);
  //  * else right away (Arc* still has to run on the main stream, or nothing to overlap with).
  // (Handing them to a second host thread was tried: the two threads' launches serialise inside
  // the runtime and the frame got slower, so everything stays on the calling thread.)
  const bool before_sync = early_results && !PUB_THIS_FRAME;
  const bool defer_late = !(PUB_THIS_FRAME && !arc_done) && have_next && !before_sync;
  if (c->prev_pts.size() > 0) {  // :405-437
    const int n = (int)c->prev_pts.size();
    const uint8_t *t_stA, *t_stB;
    const P2f *t_ptsB, *t_ptsC;
    bool spec_ok = false;
    if (use_spec) {
      if (!defer_late)
        if (int rc = early_work()) return rc;
      lap(1);
      HIPCHK(c, sync_event(c->ev_spec_done));
      lap(2);
      const size_t stM = ((size_t)std::max(M, 1) + 63) / 64 * 64;
      t_ptsB = (const P2f*)c->h_spec;
      t_ptsC = (const P2f*)(c->h_spec + (size_t)std::max(M, 1) * 8);
      t_stA = c->h_spec + (size_t)std::max(M, 1) * 16;
      t_stB = t_stA + stM;
      int* wait_expired = (int*)(c->h_spec + (size_t)std::max(M, 1) * 16 + 2 * stM);
      spec_ok = *wait_expired == 0;  // (a wave gave up waiting for k_select: redo the launch below)
      *wait_expired = 0;
    }
    std::vector<P2f> g_ptsB, g_ptsC;
    std::vector<uint8_t> g_stA, g_stB;
    if (use_chain) {
      if (!defer_late)
        if (int rc = early_work()) return rc;
      lap(1);
      HIPCHK(c, sync_event(c->ev_chain_done));
      lap(2);
      const size_t Mx = (size_t)std::max(M, 1), stM = (Mx + 63) / 64 * 64;
      const uint8_t* hc = c->h_spec + c->spec_bytes;
      int* wait_expired = (int*)(hc + Mx * 16 + 2 * stM);
      spec_ok = *wait_expired == 0;
      *wait_expired = 0;
      c->tr_chain_used += spec_ok;
      if (c->trace && spec_ok) {
        float a = 0, b = 0, d = 0;
        if (hipEventElapsedTime(&a, c->ev_dbg_sel_start, c->ev_sel_host) == hipSuccess &&
            hipEventElapsedTime(&b, c->ev_sel_host, c->ev_spec_done) == hipSuccess &&
            hipEventElapsedTime(&d, c->ev_sel_host, c->ev_chain_done) == hipSuccess) {
          c->tr_gpu_sel += a;
          c->tr_gpu_spec += b;
          c->tr_gpu_chain += d;
          float e2 = 0;
          if (c->tr_lane >= 0 &&
              hipEventElapsedTime(&e2, c->ev_sel_host, c->ev_lane_done[c->tr_lane]) == hipSuccess)
            c->tr_gpu_pyr += e2;
          else
            (void)hipGetLastError();
          c->tr_host_chain += std::chrono::duration<double, std::milli>(clk::now() - c->tr_sel_launch).count();
          c->tr_gpu_n++;
        } else {
          (void)hipGetLastError();
        }
      }
      if (spec_ok) {  // gather: prev_pts[j] was the producer's point chain_map[j]
        const P2f *sB = (const P2f*)hc, *sC = (const P2f*)(hc + Mx * 8);
        const uint8_t *sa = hc + Mx * 16, *sb = sa + stM;
        g_ptsB.resize(n);
        g_ptsC.resize(n);
        g_stA.resize(n);
        g_stB.resize(n);
        for (int j = 0; j < n; j++) {
          const int k = c->chain_map[j];
          g_ptsB[j] = sB[k];
          g_ptsC[j] = sC[k];
          g_stA[j] = sa[k];
          g_stB[j] = sb[k];
        }
        t_ptsB = g_ptsB.data();
        t_ptsC = g_ptsC.data();
        t_stA = g_stA.data();
        t_stB = g_stB.data();
      }
    }
    if (!spec_ok) {
      std::memcpy(pin.A, c->prev_pts.data(), (size_t)n * 8);
      // forward: prevL -> curL, maxLevel 3 (:410); reverse: curL -> prevL, maxLevel 1,
      // USE_INITIAL_FLOW seeded with prev_pts (:416-418) — fused into the same launch
      LkArgs f = make_lk(prevL, curL, zdev(c, pin.A), nullptr, zdev(c, pin.ptsB), zdev(c, pin.stA), nullptr, n, 3, 30, 0.01, 0);
      LkArgs b = make_lk(curL, prevL, nullptr, nullptr, nullptr, nullptr, nullptr, n, 1, 30, 0.01,
                         ESVIO_FE_LK_USE_INITIAL_FLOW);
      run_lk(c, f, cfg.flow_back ? &b : nullptr, zdev(c, pin.ptsC), zdev(c, pin.stB));
      if (int rc = early_work()) return rc;
      lap(1);
      HIPCHK(c, sync_main(c));
      lap(2);
      t_ptsB = (const P2f*)pin.ptsB;
      t_ptsC = (const P2f*)pin.ptsC;
      t_stA = pin.stA;
      t_stB = pin.stB;
    }
    std::vector<uint8_t> status(t_stA, t_stA + n);
    c->cur_pts.resize(n);
    std::memcpy(c->cur_pts.data(), t_ptsB, (size_t)n * 8);
    if (cfg.flow_back) {
      const P2f* reverse_pts = t_ptsC;
      for (int i = 0; i < n; i++) {
        if (status[i] && t_stB[i] && pt_distance(c->prev_pts[i], reverse_pts[i]) <= 0.5)
          status[i] = 1;
        else
          status[i] = 0;
      }
    }
    for (int i = 0; i < n; i++)
      if (status[i] && !in_border_event(c, c->cur_pts[i])) status[i] = 0;
    if (chain_covers_next) {
      if (use_spec && spec_ok) {  // (the producer's point i is this frame's prev_pts[i])
        c->chain_map.clear();
        for (int i = 0; i < n; i++)
          if (status[i]) c->chain_map.push_back(i);
        c->chain_map_ok = true;
      } else if (int rc = cancel_chain(c)) {
        return rc;
      }
    }
    reduce_vector(c->prev_pts, status);
    reduce_vector(c->cur_pts, status);
    reduce_vector(c->ids, status);
    reduce_vector(c->track_cnt, status);
  } else if (chain_covers_next) {
    if (int rc = cancel_chain(c)) return rc;
  }

  if (!defer_late)
    if (int rc = early_work()) return rc;  // (no previous points: nothing was synchronised above)
  for (auto& n : c->track_cnt) n++;  // :439-440

  // ---- speculative stereo LK of every temporal survivor (a superset of the points that survive
  // rejectWithF_event / Event_setMask): per-point results do not depend on the other points, so
  // this is exactly cv::calcOpticalFlowPyrLK(curL, curR, cur_pts, ...) (:490) and its reverse (:495)
  // for the kept points — launched now so that it overlaps the host-side RANSAC + mask.
  const int n_surv = (int)c->cur_pts.size();
  c->src_idx.resize(n_surv);
  for (int i = 0; i < n_surv; i++) c->src_idx[i] = i;
  lap(3);
  bool detect = false;
  int n_kept = n_surv;
  // the next batch's pyramids are in flight on the prefetch stream: next frame's temporal LK can be
  // launched as soon as this frame's points are final
  const bool will_spec = have_next || had_announced;
  auto upload_kept = [&]() -> int {
    if (!will_spec || !n_kept) return 0;
    // pin.news is a single buffer and the previous published frame's lazy stereo LK of its new
    // corners reads its points from there (z_new + its n_kept) in place.  Up to ~1000 points every
    // wave of that launch is resident from the start and has loaded its point long before the host
    // gets here (it had to wait for this frame's temporal LK first); a larger launch runs in
    // several rounds of blocks, so its completion is awaited before the slots are overwritten.
    if (c->pend.active && M > 1024) HIPCHK(c, sync_event(c->ev_lknew_done));
    std::memcpy(pin.news, c->cur_pts.data(), (size_t)n_kept * 8);  // read in place by the LK
    return 0;
  };
  if (!PUB_THIS_FRAME) {  // (ahead of the stereo LK so that the two launches overlap)
    if (int rc = upload_kept()) return rc;
    // (c->chain_valid here: the next frame's temporal LK is already running, chained to this one's)
    if (will_spec && n_kept && !c->chain_valid)
      if (const Inflight* nb = next_batch())
        if (int rc = enqueue_spec_temporal(c, *nb, n_kept, false)) return rc;
  }
  if (n_surv) {
    std::memcpy(pin.A, c->cur_pts.data(), (size_t)n_surv * 8);
    LkArgs f = make_lk(curL, curR, zdev(c, pin.A), nullptr, zdev(c, pin.ptsB), zdev(c, pin.stA), nullptr, n_surv, 3, 30,
                       0.01, 0);
    LkArgs b = make_lk(curR, curL, nullptr, nullptr, nullptr, nullptr, nullptr, n_surv, 3, 30, 0.01, 0);
    {
      // on its own stream.  Its inputs are complete without a device-side wait: the host has just
      // read this frame's temporal LK results, and that launch ran behind the frame's pyramids.
      StreamScope on_stereo_stream(c->stream4);
      run_lk(c, f, cfg.flow_back ? &b : nullptr, zdev(c, pin.ptsC), zdev(c, pin.stB));
      HIPCHK(c, hipEventRecord(c->ev_lks_done[c->res_set], c->stream4));
    }
    c->lks_last = c->res_set;
  }

  if (PUB_THIS_FRAME) {  // :442-469
    if (cfg.f_ransac) reject_with_f_event(c);
    lap(4);
    auto tq = clk::now();
    auto sub = [&](int i) {
      if (!c->trace) return;
      const auto now = clk::now();
      c->pub_ms[i] += std::chrono::duration<double, std::milli>(now - tq).count();
      tq = now;
    };
    event_set_mask(c);
    sub(0);
    n_kept = (int)c->cur_pts.size();
    const int n_max_cnt = M - n_kept;
    if (int rc = upload_kept()) return rc;
    if (n_max_cnt <= 0 && will_spec && n_kept)
      if (const Inflight* nb = next_batch())
        if (int rc = enqueue_spec_temporal(c, *nb, n_kept, false)) return rc;
    if (n_max_cnt > 0) {
      detect = true;
      // Event_setMask's blocked pixels are the discs of the kept points: k_select stamps them
      // into its bitmap itself from the points just written to pin.news (1-2 KB read in place
      // instead of a 38 KB bitmap copied over); candidates on them are skipped there
      if (!will_spec && n_kept) std::memcpy(pin.news, c->cur_pts.data(), (size_t)n_kept * 8);
      if (arc_prefetched) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_lane_arc[arc_lane], 0));
      // new corners go behind the kept points: z_new = next frame's prev_pts
      c->pub_seq++;
      if (c->trace) {
        HIPCHK(c, hipEventRecord(c->ev_dbg_sel_start, cur_stream(c)));
        c->tr_sel_launch = clk::now();
      }
      run_select(c, c->cand_cur, n_max_cnt, c->z_new, n_kept, nullptr, nullptr, c->z_counts, will_spec,
                 c->z_new, n_kept);
      sub(1);
      if (will_spec)
        if (const Inflight* nb = next_batch())
          if (int rc = enqueue_spec_temporal(c, *nb, n_kept, true)) return rc;
      sub(2);
      // the selection result is in host memory once k_select is done: an event right behind it lets
      // the left-camera bookkeeping below run under the stereo LK of the new corners
      HIPCHK(c, hipEventRecord(c->ev_sel_host, cur_stream(c)));
      if (int rc = finalize_pending(c)) return rc;  // (its results live where this launch writes)
      if (int rc = finalize_right(c)) return rc;    // (idle time: k_select is running)
      sub(3);
      // stereo LK of the new corners only (count known on the device)
      LkArgs f = make_lk(curL, curR, c->z_new + n_kept, nullptr, c->z_ptsB2, c->z_stA2, c->d_counts,
                         n_max_cnt, 3, 30, 0.01, 0);
      LkArgs b = make_lk(curR, curL, nullptr, nullptr, nullptr, nullptr, c->d_counts, n_max_cnt, 3, 30,
                         0.01, 0);
      run_lk(c, f, cfg.flow_back ? &b : nullptr, c->z_ptsC2, c->z_stB2);
      if (c->lazy_new) HIPCHK(c, hipEventRecord(c->ev_lknew_done, cur_stream(c)));
      sub(4);
    }
    if (defer_late)
      if (int rc = early_work()) return rc;
    sub(5);
  } else if (defer_late) {
    if (int rc = early_work()) return rc;
  }
  lap(5);
  if (detect) HIPCHK(c, sync_event(c->ev_sel_host));

  int n_new = 0;
  if (PUB_THIS_FRAME) {
    c->n_pts.clear();
    if (detect) {
      n_new = pin.counts[0];
      c->tr_cand += (uint64_t)pin.counts[2];
      c->tr_new += (uint64_t)n_new;
      c->tr_detect++;
      const P2f* np = (const P2f*)pin.news + n_kept;
      for (int i = 0; i < n_new; i++) c->n_pts.push_back(np[i]);
    }
    for (auto& p : c->n_pts) {  // :463-468
      c->cur_pts.push_back(p);
      c->ids.push_back(c->n_id++);
      c->track_cnt.push_back(1);
    }
  }
  c->cur_un_pts = undistorted_pts(c->cur_pts, cfg.cam[0]);  // :470-473
  c->pts_velocity = pts_velocity_fn(c->ids, c->cur_un_pts, c->cur_un_pts_map, c->prev_un_pts_map,
                                    c->cur_time - c->prev_time, c->cur_pts.size());
  lap(7);
  if (int rc = finalize_pending(c)) return rc;  // (the previous published frame's new corners,
  if (int rc = finalize_right(c)) return rc;    //  or the previous unpublished frame's whole tail)
  const bool lazy = c->lazy_new && detect;      // leave this frame's new corners to the next call
  const bool defer_right = c->lazy_new && !PUB_THIS_FRAME;  // ... or its whole right-camera tail
  if (defer_right) {
    // (returns with the stereo LK in flight)
  } else {
    if (!lazy) HIPCHK(c, sync_main(c));  // stereo LK results of the new corners
    if (n_surv) HIPCHK(c, sync_event(c->ev_lks_done[c->res_set]));  // ... of the kept points
  }
  lap(6);
  if (!defer_right && (n_surv || detect) && pin.counts[3] != 0)
    return fail(c, ESVIO_FE_EINTERNAL, "radix sort look-back spin expired");

  if (defer_right) {
    // nothing of this frame is published: its right-camera tail waits for the next call
    c->pend_right.active = true;
    c->pend_right.set = c->res_set;
    c->pend_right.dt = c->cur_time - c->prev_time;
    c->pend_right.ids = c->ids;
    c->pend_right.left = c->cur_pts;
  } else {
    if (lazy) {
      c->pend.active = true;
      c->pend.prev_map_was_empty = c->prev_un_right_pts_map.empty();
      c->pend.ids.assign(c->ids.begin() + n_kept, c->ids.end());
      c->pend.left.assign(c->cur_pts.begin() + n_kept, c->cur_pts.end());
    }
    right_tail(c, pin, c->cur_pts.data(), c->ids.data(), c->src_idx.data(),
               lazy ? n_kept : (int)c->cur_pts.size(), n_kept, c->cur_time - c->prev_time,
               c->cur_pts.size());
  }
  c->slot_prevL = c->slot_curL;  // prev_img_left = cur_img_left (:585)
  c->prev_pts = c->cur_pts;
  c->prev_un_pts_map.swap(c->cur_un_pts_map);
  c->prev_time = c->cur_time;
  c->spec_n = (int)c->prev_pts.size();
  lap(7);
  c->phase_frames++;
  c->phase_count[PUB_THIS_FRAME ? 1 : 0]++;
  c->tr_surv += (uint64_t)n_surv;
  if (c->prof_on) resolve_profile(c);
  return 0;
}


// ================================================================ image front-end
// half-widths of the open Euclidean disc dx*dx + dy*dy < md*md (goodFeaturesToTrack's distance test)
void euclid_halfwidths(double md, int8_t* hw /*[kMaxDiscR+1]*/, int* radius) {
  const double md2 = md * md;
  *radius = 0;
  for (int dy = 0; dy <= kMaxDiscR; dy++) {
    int w = -1;
    for (int dx = 0; dx <= kMaxDiscR; dx++)
      if ((double)dx * dx + (double)dy * dy < md2) w = dx;
    hw[dy] = (int8_t)w;
    if (w >= 0) *radius = dy;
  }
}

// cv::goodFeaturesToTrack on the level-0 image of pyramid `d` (padded, so no border arithmetic);
// up to max_corners corners are written at out_pts[out_base ..], counts mirrored to host_counts.
// `use_mask`: d_mask_bits holds the blocked pixels.  Synchronises the stream once (the number of
// local maxima sizes the sort).
int gftt_run(esvio_fe_ctx* c, const PyrDesc& d, int max_corners, double quality, double min_distance,
             bool use_mask, float2* out_pts, int out_base, int* host_counts) {
  const size_t P = (size_t)c->W * c->H;
  if (!c->d_gftt_cov) {
    if (int rc = dev_alloc(c, &c->d_gftt_cov, P)) return rc;
    if (int rc = dev_alloc(c, &c->d_gftt_rowsum, P)) return rc;
    if (int rc = dev_alloc(c, &c->d_gftt_eig, P)) return rc;
    if (int rc = dev_alloc(c, &c->d_gftt_max, 1)) return rc;
  }
  const int set = c->cand_cur;
  if (int rc = ensure_cand_capacity(c, set, P)) return rc;
  const esvio_fe_ctx::CandSet& cs = c->cand[set];
  GfttArgs g{};
  g.img = px00(d);
  g.stride = d.stride[0];
  g.W = c->W;
  g.H = c->H;
  g.cov = c->d_gftt_cov;
  g.rowsum = c->d_gftt_rowsum;
  g.eig = c->d_gftt_eig;
  g.mask_bits = use_mask ? c->d_mask_bits : nullptr;
  g.wpr = (c->W + 31) / 32;
  g.max_key = c->d_gftt_max;
  g.quality = quality;
  g.cand_xy = cs.xy;
  g.cand_val = cs.idx;
  g.cand_cnt = cs.cnt;
  launch_gftt_response(cur_stream(c), g);
  launch_gftt_collect(cur_stream(c), g);
  const uint32_t nblk = (uint32_t)((P + kArcBlock - 1) / kArcBlock);
  launch_compact(cur_stream(c), cs.xy, cs.idx, cs.cnt, nblk, cs.comp_xy, cs.comp_idx, cs.total);
  uint32_t n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, cs.total, 4, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  const uint32_t* sorted_xy = cs.comp_xy;
  if (n > 1) {  // by response, then address, both descending: 4 x 8-bit stable radix passes
    if (int rc = ensure_sort_capacity(c, n)) return rc;
    const uint32_t head = ((uint32_t)kRadixMaxPasses << kRadixMaxBits) + 64;
    const uint32_t nb = radix_blocks(n);
    uint32_t* ghist = c->hist;
    uint32_t* tickets = c->hist + ((size_t)kRadixMaxPasses << kRadixMaxBits);
    uint32_t* lookback = c->hist + head;
    HIPCHK(c, hipMemsetAsync(c->hist, 0, (size_t)head * 4, cur_stream(c)));
    launch_gftt_sortprep(cur_stream(c), cs.comp_xy, cs.comp_idx, n, c->keys[0], c->vals[0], ghist, lookback,
                         4u * (nb << 8));
    int cur = 0;
    for (int p = 0; p < 4; p++) {
      launch_radix_pass(cur_stream(c), c->keys[cur], c->vals[cur], n, 8 * p, 8, ghist + ((size_t)p << 8),
                        lookback + (size_t)p * (nb << 8), tickets + p, c->keys[cur ^ 1],
                        c->vals[cur ^ 1], c->z_counts + 3);
      cur ^= 1;
    }
    HIPCHK(c, hipMemsetAsync(c->hist, 0, (size_t)head * 4, cur_stream(c)));  // as k_sae_apply leaves it
    sorted_xy = c->vals[cur];
  }
  SelectArgs sa{};
  sa.comp_xy = sorted_xy;
  sa.comp_idx = sorted_xy;
  sa.total = cs.total;
  sa.W = c->W;
  sa.H = c->H;
  sa.wpr = (c->W + 31) / 32;
  sa.max_corners = max_corners;
  euclid_halfwidths(min_distance, sa.hw, &sa.radius);
  sa.disc_c = c->disc_tab_only ? -1 : disc_threshold(sa.hw, sa.radius);
  sa.out_pts = out_pts;
  sa.out_idx = nullptr;
  sa.out_base = out_base;
  sa.n_out = c->d_counts;
  sa.n_total = c->d_counts + 1;
  sa.host_counts = host_counts;
  sa.init_bits = nullptr;
  sa.pub_slots = nullptr;
  sa.pub_done = nullptr;
  sa.pub_seq = 0;
  ScopedKernel k(c, K_SELECT, 0);
  launch_select(cur_stream(c), sa, select_lds_bytes(c));
  return 0;
}

// Image_setMask (feature_tracker.cpp:90-119, FISHEYE 0): like Event_setMask on a CV_8UC1 mask;
// c->mask_event then holds the BLOCKED pixels (the reference's mask_image == 0)
void image_set_mask(esvio_fe_ctx* c) {
  c->mask_event.reset(c->W, c->H);
  std::vector<std::pair<int, std::pair<P2f, int>>> cnt_pts_id;
  cnt_pts_id.reserve(c->cur_pts.size());
  for (unsigned int i = 0; i < c->cur_pts.size(); i++)
    cnt_pts_id.push_back(std::make_pair(c->track_cnt[i], std::make_pair(c->cur_pts[i], c->ids[i])));
  std::sort(cnt_pts_id.begin(), cnt_pts_id.end(),
            [](const std::pair<int, std::pair<P2f, int>>& a,
               const std::pair<int, std::pair<P2f, int>>& b) { return a.first > b.first; });
  c->cur_pts.clear();
  c->ids.clear();
  c->track_cnt.clear();
  for (auto& it : cnt_pts_id) {
    const int px = host::cv_round(it.second.first.x), py = host::cv_round(it.second.first.y);
    if (px < 0 || px >= c->W || py < 0 || py >= c->H) continue;  // cannot happen after inBorder
    if (!c->mask_event.test(px, py)) {
      c->cur_pts.push_back(it.second.first);
      c->ids.push_back(it.second.second);
      c->track_cnt.push_back(it.first);
      c->mask_event.stamp_disc(px, py, c->cfg.min_dist, c->hw);
    }
  }
}

// FeatureTracker::trackImage (feature_tracker.cpp:164-338) for a handle whose width/height/max_cnt/
// min_dist are the image camera's COL/ROW/MAX_CNT_IMG/MIN_DIST_IMG.  cfg.equalize applies the
// node's CLAHE (stereo_image_tracker_node.cpp:92-96, no normalisation) to both images first.
// No pipelining here: one frame at a time on the main stream.
int track_image_impl(esvio_fe_ctx* c, double _cur_time, const uint8_t* img_left,
                     const uint8_t* img_right, bool PUB_THIS_FRAME) {
  const esvio_fe_config& cfg = c->cfg;
  const int M = cfg.max_cnt;
  if (int rc = finalize_pending(c)) return rc;  // (a lazy trackEvent call came before)
  if (int rc = finalize_right(c)) return rc;
  if (int rc = cancel_chain(c)) return rc;
  Pin pin = pin_of(c);
  c->cur_time = _cur_time;
  const bool first = !c->have_img;
  const bool have_right = img_right != nullptr;
  // slot rotation as in trackEvent's plain path
  int sl = 0;
  while (!first && (sl == c->slot_prevL || sl == c->slot_curL)) sl++;
  c->slot_curL = sl;
  c->slot_curR = c->slot_curR == kLeftSlots ? kLeftSlots + 1 : kLeftSlots;
  const PyrDesc& L = c->pyr[c->slot_curL].d;
  const PyrDesc& R = c->pyr[c->slot_curR].d;
  if (cfg.equalize) {
    c->raw_cur = (c->raw_cur + 1) % kRightSlots;
    const PyrDesc& rl = c->raw[c->raw_cur][0].d;
    const PyrDesc& rr = c->raw[c->raw_cur][1].d;
    if (int rc = copy_level0_in(c, rl, img_left)) return rc;
    if (have_right)
      if (int rc = copy_level0_in(c, rr, img_right)) return rc;
    const int nimg = have_right ? 2 : 1;
    for (int stage = 0; stage < 2; stage++) {
      ScopedKernel k(c, K_CLAHE, 0);
      launch_clahe(cur_stream(c), px00(rl), have_right ? px00(rr) : px00(rl), rl.stride[0], px00(L),
                   have_right ? px00(R) : px00(L), L.stride[0], c->W, c->H, c->d_lut, c->d_minmax, nimg,
                   stage);
    }
  } else {
    if (int rc = copy_level0_in(c, L, img_left)) return rc;
    if (have_right)
      if (int rc = copy_level0_in(c, R, img_right)) return rc;
  }
  {
    PyrDesc two[2] = {L, R};
    pyr_build(c, two, have_right ? 2 : 1);
  }
  if (first) c->slot_prevL = c->slot_curL;
  c->have_img = true;
  const PyrDesc& prevL = c->pyr[c->slot_prevL].d;
  c->cur_pts.clear();

  if (c->prev_pts.size() > 0) {  // :180-209: forward, and backward with maxLevel 3 / no initial flow
    const int n = (int)c->prev_pts.size();
    std::memcpy(pin.A, c->prev_pts.data(), (size_t)n * 8);
    LkArgs f = make_lk(prevL, L, zdev(c, pin.A), nullptr, zdev(c, pin.ptsB), zdev(c, pin.stA), nullptr, n, 3, 30, 0.01, 0);
    LkArgs b = make_lk(L, prevL, nullptr, nullptr, nullptr, nullptr, nullptr, n, 3, 30, 0.01, 0);
    run_lk(c, f, cfg.flow_back ? &b : nullptr, zdev(c, pin.ptsC), zdev(c, pin.stB));
    HIPCHK(c, sync_main(c));
    std::vector<uint8_t> status(pin.stA, pin.stA + n);
    c->cur_pts.resize(n);
    std::memcpy(c->cur_pts.data(), pin.ptsB, (size_t)n * 8);
    if (cfg.flow_back) {
      const P2f* reverse_pts = (const P2f*)pin.ptsC;
      for (int i = 0; i < n; i++)
        status[i] = status[i] && pin.stB[i] && pt_distance(c->prev_pts[i], reverse_pts[i]) <= 0.5;
    }
    for (int i = 0; i < n; i++)
      if (status[i] && !in_border_event(c, c->cur_pts[i])) status[i] = 0;
    reduce_vector(c->prev_pts, status);
    reduce_vector(c->cur_pts, status);
    reduce_vector(c->ids, status);
    reduce_vector(c->track_cnt, status);
  }
  for (auto& n : c->track_cnt) n++;

  if (PUB_THIS_FRAME) {  // :214-241
    image_set_mask(c);
    const int n_max_cnt = M - (int)c->cur_pts.size();
    c->n_pts.clear();
    if (n_max_cnt > 0) {
      std::memcpy(pin.mask, c->mask_event.bits.data(), c->mask_event.bits.size() * 4);
      HIPCHK(c, hipMemcpyAsync(c->d_mask_bits, pin.mask, c->mask_event.bits.size() * 4,
                               hipMemcpyHostToDevice, cur_stream(c)));
      if (int rc = gftt_run(c, L, n_max_cnt, 0.01, (double)cfg.min_dist, true, c->z_new, 0, c->z_counts))
        return rc;
      HIPCHK(c, sync_main(c));
      if (pin.counts[3] != 0) return fail(c, ESVIO_FE_EINTERNAL, "radix sort look-back spin expired");
      const int n_new = pin.counts[0];
      const P2f* np = (const P2f*)pin.news;
      for (int i = 0; i < n_new; i++) c->n_pts.push_back(np[i]);
    }
    for (auto& p : c->n_pts) {
      c->cur_pts.push_back(p);
      c->ids.push_back(c->n_id++);
      c->track_cnt.push_back(1);
    }
  }
  c->cur_un_pts = undistorted_pts(c->cur_pts, cfg.cam[0]);
  c->pts_velocity = pts_velocity_fn(c->ids, c->cur_un_pts, c->cur_un_pts_map, c->prev_un_pts_map,
                                    c->cur_time - c->prev_time, c->cur_pts.size());

  if (have_right) {  // :249-318
    c->ids_right.clear();
    c->cur_right_pts.clear();
    c->cur_un_right_pts.clear();
    c->right_pts_velocity.clear();
    c->cur_un_right_pts_map.clear();
    c->track_cnt_right.clear();
    if (!c->cur_pts.empty()) {
      const int n = (int)c->cur_pts.size();
      std::memcpy(pin.A, c->cur_pts.data(), (size_t)n * 8);
      LkArgs f = make_lk(L, R, zdev(c, pin.A), nullptr, zdev(c, pin.ptsB), zdev(c, pin.stA), nullptr, n, 3, 30, 0.01, 0);
      LkArgs b = make_lk(R, L, nullptr, nullptr, nullptr, nullptr, nullptr, n, 3, 30, 0.01, 0);
      run_lk(c, f, cfg.flow_back ? &b : nullptr, zdev(c, pin.ptsC), zdev(c, pin.stB));
      HIPCHK(c, sync_main(c));
      std::vector<uint8_t> status(pin.stA, pin.stA + n);
      c->cur_right_pts.resize(n);
      std::memcpy(c->cur_right_pts.data(), pin.ptsB, (size_t)n * 8);
      if (cfg.flow_back) {
        const P2f* reverseLeftPts = (const P2f*)pin.ptsC;
        for (int i = 0; i < n; i++)
          status[i] = status[i] && pin.stB[i] && in_border_event(c, c->cur_right_pts[i]) &&
                      pt_distance(c->cur_pts[i], reverseLeftPts[i]) <= 0.5;
      }
      c->ids_right = c->ids;
      reduce_vector(c->cur_right_pts, status);
      reduce_vector(c->ids_right, status);
      c->cur_un_right_pts = undistorted_pts(c->cur_right_pts, cfg.cam[1]);
      c->right_pts_velocity =
          pts_velocity_fn(c->ids_right, c->cur_un_right_pts, c->cur_un_right_pts_map,
                          c->prev_un_right_pts_map, c->cur_time - c->prev_time, c->cur_pts.size());
    }
    c->prev_un_right_pts_map.swap(c->cur_un_right_pts_map);
  }
  c->slot_prevL = c->slot_curL;
  c->prev_pts = c->cur_pts;
  c->prev_un_pts_map.swap(c->cur_un_pts_map);
  c->prev_time = c->cur_time;
  if (c->prof_on) resolve_profile(c);
  return 0;
}

}  // namespace

// ==================================================================================== C ABI
extern "C" {

const char* esvio_fe_version(void) { return "esvio_fe 0.1 (gfx950)"; }

const char* esvio_fe_last_error(esvio_fe_handle h) { return h ? h->err.c_str() : "null handle"; }

int esvio_fe_destroy(esvio_fe_handle c) {
  if (!c) return ESVIO_FE_EINVAL;
  (void)hipSetDevice(c->dev);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream4) (void)hipStreamSynchronize(c->stream4);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  host::ransac_pool_destroy(c->pool);
  c->pool = nullptr;
  destroy_launch_graph(c->pf_graph);
  if (c->trace && c->phase_frames) {
    static const char* nm[8] = {"enqueue sae+ts+pyr", "enqueue temporal LK", "sync A", "host filter",
                                "host ransac", "host mask + enqueue detect/stereo", "sync B", "host tail"};
    for (int pub = 0; pub < 2; pub++) {
      if (!c->phase_count[pub]) continue;
      double tot = 0;
      fprintf(stderr, "[esvio_fe trace] %llu %s frames, ms/frame:", (unsigned long long)c->phase_count[pub],
              pub ? "published" : "unpublished");
      for (int i = 0; i < 8; i++) {
        fprintf(stderr, " %s=%.3f", nm[i], c->phase_ms[pub][i] / c->phase_count[pub]);
        tot += c->phase_ms[pub][i] / c->phase_count[pub];
      }
      fprintf(stderr, " | total=%.3f\n", tot);
    }
    if (c->phase_count[1]) {
      static const char* pn[6] = {"Event_setMask", "points + k_select launch", "speculative + chained LK launches",
                                  "previous frame's right-camera tail", "stereo LK of new corners launch",
                                  "next batch's prefetch launches"};
      fprintf(stderr, "[esvio_fe trace] published frames, parts of 'host mask + enqueue', ms/frame:");
      for (int i = 0; i < 6; i++) fprintf(stderr, " %s=%.3f", pn[i], c->pub_ms[i] / c->phase_count[1]);
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "[esvio_fe trace]");
    fprintf(stderr, "\n[esvio_fe trace] findFundamentalMat alone: %.3f ms per published frame (slowest call %.3f ms, "
            "%.3f without it); the two liftProjective batches before it: %.3f ms",
            c->phase_count[1] ? c->tr_fm_ms / c->phase_count[1] : 0.0, c->tr_fm_max_ms,
            c->phase_count[1] > 1 ? (c->tr_fm_ms - c->tr_fm_max_ms) / (c->phase_count[1] - 1) : 0.0,
            c->phase_count[1] ? c->tr_lift_ms / c->phase_count[1] : 0.0);
    if (c->tr_gpu_n)
      fprintf(stderr, "\n[esvio_fe trace] device: k_select %.1f us; select end -> next frame's temporal LK done "
              "%.1f us, -> chained one done %.1f us (its frame's pyramids: %.1f us); host: select launch -> "
              "chained results read %.1f us",
              1e3 * c->tr_gpu_sel / c->tr_gpu_n, 1e3 * c->tr_gpu_spec / c->tr_gpu_n,
              1e3 * c->tr_gpu_chain / c->tr_gpu_n, 1e3 * c->tr_gpu_pyr / c->tr_gpu_n,
              1e3 * c->tr_host_chain / c->tr_gpu_n);
    fprintf(stderr, "\n[esvio_fe trace] chained temporal LK: %llu launched, %llu used, %llu cancelled",
            (unsigned long long)c->tr_chain_launch, (unsigned long long)c->tr_chain_used,
            (unsigned long long)c->tr_chain_cancel);
    fprintf(stderr, "\n[esvio_fe trace] survivors/frame=%.1f; detect frames=%llu: candidates/frame=%.0f new/frame=%.1f\n",
            (double)c->tr_surv / c->phase_frames, (unsigned long long)c->tr_detect,
            c->tr_detect ? (double)c->tr_cand / c->tr_detect : 0.0,
            c->tr_detect ? (double)c->tr_new / c->tr_detect : 0.0);
  }
  if (c->x_pin) (void)hipHostFree(c->x_pin);
  void* ptrs[] = {c->x_send, c->x_recv, c->d_part, c->d_tile, c->L2s, c->S2s, c->slice_stage, c->L2, c->S2, c->d_ev, c->keys[0], c->keys[1], c->vals[0], c->vals[1], c->hist, c->sae_marks,
                  c->d_rejected, c->d_res, c->d_ptsD, c->d_flags, c->d_pub_slots, c->d_pub_done, c->d_chain, c->d_gftt_cov, c->d_gftt_rowsum, c->d_gftt_eig, c->d_gftt_max,
                  c->d_mask_bits, c->d_sel_idx,
                  c->tmp_pyr[0].mem, c->tmp_pyr[1].mem, c->med_tmp[0].mem, c->med_tmp[1].mem, c->d_lut,
                  c->d_minmax};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& cs : c->cand)
    for (void* p : {(void*)cs.xy, (void*)cs.idx, (void*)cs.cnt, (void*)cs.comp_xy, (void*)cs.comp_idx,
                    (void*)cs.total})
      if (p) (void)hipFree(p);
  for (uint32_t* p : c->d_first)
    if (p) (void)hipFree(p);
  for (uint32_t* p : c->d_cmap)
    if (p) (void)hipFree(p);
  for (uint8_t* p : c->d_touched)
    if (p) (void)hipFree(p);
  for (PyrStore& ps : c->pyr)
    if (ps.mem) (void)hipFree(ps.mem);
  for (auto& rb : c->raw)
    for (PyrStore& ps : rb)
      if (ps.mem) (void)hipFree(ps.mem);
  for (int i = 0; i < kPrefetchDepth; i++) {
    if (c->d_evp[i]) (void)hipFree(c->d_evp[i]);
    if (c->ev_lane_done[i]) (void)hipEventDestroy(c->ev_lane_done[i]);
    if (c->ev_lane_arc[i]) (void)hipEventDestroy(c->ev_lane_arc[i]);
  }
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_spec) (void)hipHostFree(c->h_spec);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream4) (void)hipStreamDestroy(c->stream4);
  for (auto& r : c->pending) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->stream2) {
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamDestroy(c->stream2);
  }
  if (c->ev_planes_free) (void)hipEventDestroy(c->ev_planes_free);
  if (c->ev_pts_ready) (void)hipEventDestroy(c->ev_pts_ready);
  if (c->ev_spec_done) (void)hipEventDestroy(c->ev_spec_done);
  if (c->ev_chain_done) (void)hipEventDestroy(c->ev_chain_done);
  if (c->ev_dbg_sel_start) (void)hipEventDestroy(c->ev_dbg_sel_start);
  if (c->ev_sel_host) (void)hipEventDestroy(c->ev_sel_host);
  for (hipEvent_t e : c->ev_lks_done)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_lknew_done) (void)hipEventDestroy(c->ev_lknew_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int esvio_fe_create(const esvio_fe_config* cfg, esvio_fe_handle* out) {
  if (!cfg || !out) return ESVIO_FE_EINVAL;
  *out = nullptr;
  if (cfg->width < 2 * kLkWin || cfg->height < 2 * kLkWin || cfg->width > 8192 || cfg->height > 8192)
    return ESVIO_FE_EINVAL;
  if (cfg->max_cnt < 1 || cfg->max_cnt > 65536) return ESVIO_FE_EINVAL;
  if (cfg->min_dist < 3 || cfg->min_dist > kMaxDiscR) return ESVIO_FE_EINVAL;  // Arc* ring r=4
  if (cfg->lk_accum != 1) return ESVIO_FE_EINVAL;
  if (cfg->median_blur_kernel_size < 0) return ESVIO_FE_EINVAL;
  if (cfg->median_blur_kernel_size > kMaxMedianK) return ESVIO_FE_ENOTIMPL;  // ksize > 15
  if (cfg->equalize != 0 && cfg->equalize != 1) return ESVIO_FE_EINVAL;
  if (!(cfg->decay_ms > 0)) return ESVIO_FE_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ESVIO_FE_ENODEVICE;
  int dev = cfg->device;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) return ESVIO_FE_ENODEVICE;
  }
  if (dev >= ndev) return ESVIO_FE_ENODEVICE;
  if (hipSetDevice(dev) != hipSuccess) return ESVIO_FE_ENODEVICE;

  esvio_fe_ctx* c = new esvio_fe_ctx();
  c->cfg = *cfg;
  c->dev = dev;
  c->W = cfg->width;
  c->H = cfg->height;
  c->P = (uint32_t)c->W * c->H;
  c->invalid_key = 2 * c->P;
  c->key_bits = 1;
  while ((1ull << c->key_bits) <= (unsigned long long)c->invalid_key) c->key_bits++;
  c->hw = host::disc_halfwidths(cfg->min_dist);
  c->trace = getenv("ESVIO_FE_TRACE") != nullptr;
  c->mask_event.reset(c->W, c->H);

  auto bail = [&](int rc) {
    esvio_fe_destroy(c);
    return rc;
  };
  // the frame's own chain (LK, selection: few, latency-bound waves) outranks the prefetch stream's
  // wide kernels, which have a whole frame of slack
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  const bool streams_ok =
      hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
      hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_least) == hipSuccess &&
      hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
      hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, prio_least) == hipSuccess;
  if (!streams_ok ||
      hipEventCreateWithFlags(&c->ev_pts_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_spec_done, c->trace ? 0 : hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_chain_done, c->trace ? 0 : hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_dbg_sel_start, 0) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_sel_host, c->trace ? 0 : hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_lks_done[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_lks_done[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_lknew_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_planes_free, hipEventDisableTiming) != hipSuccess)
    return bail(ESVIO_FE_EHIP);
  for (int i = 0; i < kPrefetchDepth; i++)
    if (hipEventCreateWithFlags(&c->ev_lane_done[i], c->trace ? 0 : hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_lane_arc[i], hipEventDisableTiming) != hipSuccess)
      return bail(ESVIO_FE_EHIP);
  const size_t M = cfg->max_cnt;
  int rc = 0;
  if ((rc = dev_alloc(c, &c->L2, (size_t)2 * c->P))) return bail(rc);
  if ((rc = dev_alloc(c, &c->S2, (size_t)2 * c->P))) return bail(rc);
  if ((rc = dev_alloc(c, &c->d_rejected, 1))) return bail(rc);
  {
    const ResLayout L = res_layout(M);
    c->res_bytes = L.total;
    if ((rc = dev_alloc(c, &c->d_res, c->res_bytes))) return bail(rc);
    c->d_ptsB = (float2*)(c->d_res + L.B1[0]);
    c->d_ptsC = (float2*)(c->d_res + L.C1[0]);
    c->d_stA = c->d_res + L.SA1[0];
    c->d_stB = c->d_res + L.SB1[0];
    c->d_counts = (int*)(c->d_res + L.CNT);
    c->d_ptsA = (float2*)(c->d_res + L.A[0]);
  }
  {
    const size_t stM = (std::max<size_t>(M, 1) + 63) / 64 * 64;
    c->spec_bytes = std::max<size_t>(M, 1) * 16 + 2 * stM + 64;  // results + the wait-expired flag
    c->spec_bytes = (c->spec_bytes + 255) / 256 * 256;
    // (twice: the speculative launch's block, then the chained launch's)
    if (hipHostMalloc((void**)&c->h_spec, 2 * c->spec_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->z_spec, c->h_spec, 0) != hipSuccess)
      return bail(ESVIO_FE_EHIP);
    std::memset(c->h_spec, 0, 2 * c->spec_bytes);
  }
  if ((rc = dev_alloc(c, &c->d_chain, 2 * std::max<size_t>(M, 1)))) return bail(rc);
  c->chain_enabled = getenv("ESVIO_FE_NO_CHAIN") == nullptr;
  c->graphs_enabled = getenv("ESVIO_FE_GRAPH") != nullptr;
  c->dedup_enabled = getenv("ESVIO_FE_NO_DEDUP") == nullptr;
  c->fuse_ts_pyr = getenv("ESVIO_FE_NO_FUSE") == nullptr;
  c->disc_tab_only = getenv("ESVIO_FE_DISC_TABLE") != nullptr;
  if (const char* v = getenv("ESVIO_FE_SAE_EV_MIN")) c->sae_ev_min = (size_t)strtoull(v, nullptr, 10);
  c->tiled = make_tile_geom(c->W, c->H, &c->tgeom) && getenv("ESVIO_FE_SAE_SORT") == nullptr;
  for (int i = 0; i < kRightSlots; i++)
    if ((rc = dev_alloc(c, &c->d_first[i], (size_t)c->P))) return bail(rc);
  for (int i = 0; i < kRightSlots; i++) {
    const size_t words = arc_bitmap_words(c->W, c->H);
    if ((rc = dev_alloc(c, &c->d_cmap[i], words))) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_touched[i], arc_flag_bytes(c->W, c->H)))) return bail(rc);
    if (hipMemsetAsync(c->d_cmap[i], 0, words * 4, cur_stream(c)) != hipSuccess ||
        hipMemsetAsync(c->d_touched[i], 0, arc_flag_bytes(c->W, c->H), cur_stream(c)) != hipSuccess)
      return bail(ESVIO_FE_EHIP);
  }
  if ((rc = dev_alloc(c, &c->d_pub_slots, std::max<size_t>(M, 1)))) return bail(rc);
  if ((rc = dev_alloc(c, &c->d_pub_done, 1))) return bail(rc);
  if (hipMemsetAsync(c->d_chain, 0, std::max<size_t>(M, 1) * 16, cur_stream(c)) != hipSuccess ||
      hipMemsetAsync(c->d_pub_slots, 0, std::max<size_t>(M, 1) * 8, cur_stream(c)) != hipSuccess ||
      hipMemsetAsync(c->d_pub_done, 0, 8, cur_stream(c)) != hipSuccess)
    return bail(ESVIO_FE_EHIP);
  if ((rc = dev_alloc(c, &c->d_ptsD, M))) return bail(rc);
  if ((rc = dev_alloc(c, &c->d_sel_idx, M))) return bail(rc);
  if ((rc = dev_alloc(c, &c->d_mask_bits, (size_t)c->H * ((c->W + 31) / 32)))) return bail(rc);
  for (PyrStore& ps : c->pyr)
    if ((rc = pyr_alloc(c, ps, c->W, c->H, 3))) return bail(rc);
  if (cfg->median_blur_kernel_size > 0)
    for (PyrStore& ps : c->med_tmp)
      if ((rc = pyr_alloc(c, ps, c->W, c->H, 0))) return bail(rc);
  if (cfg->equalize) {
    for (auto& rb : c->raw)
      for (PyrStore& ps : rb)
        if ((rc = pyr_alloc(c, ps, c->W, c->H, 0))) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_lut, (size_t)2 * 64 * 256))) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_minmax, 4))) return bail(rc);
  }
  c->h_pin_bytes = pin_bytes(*cfg);
  if (hipHostMalloc((void**)&c->h_pin, c->h_pin_bytes, hipHostMallocDefault) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->z_res, c->h_pin, 0) != hipSuccess)
    return bail(ESVIO_FE_EHIP);
  std::memset(c->h_pin, 0, c->h_pin_bytes);
  {
    const ResLayout L = res_layout(std::max<size_t>(M, 1));
    c->z_counts = (int*)(c->z_res + L.CNT);
    c->z_new = (float2*)(c->z_res + L.NEW);
    c->z_ptsB2 = (float2*)(c->z_res + L.B2);
    c->z_ptsC2 = (float2*)(c->z_res + L.C2);
    c->z_stA2 = c->z_res + L.SA2;
    c->z_stB2 = c->z_res + L.SB2;
  }
  if (hipMemsetAsync(c->L2, 0, (size_t)2 * c->P * 16, cur_stream(c)) != hipSuccess ||
      hipMemsetAsync(c->S2, 0, (size_t)2 * c->P * 16, cur_stream(c)) != hipSuccess ||
      hipMemsetAsync(c->d_rejected, 0, 8, cur_stream(c)) != hipSuccess ||
      hipMemsetAsync(c->d_counts, 0, 64, cur_stream(c)) != hipSuccess ||
      hipStreamSynchronize(cur_stream(c)) != hipSuccess)
    return bail(ESVIO_FE_EHIP);
  size_t lds = select_lds_bytes(c);
  if (lds > 160 * 1024) return bail(ESVIO_FE_EINVAL);
  *out = c;
  return 0;
}

int esvio_fe_reset(esvio_fe_handle c) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream3));
  HIPCHK(c, hipStreamSynchronize(c->stream4));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  c->announced.clear();
  c->inflight.clear();
  c->spec_valid = false;
  c->chain_valid = false;
  c->chain_map_ok = false;
  c->pend.active = false;
  c->pend_right.active = false;
  HIPCHK(c, hipMemsetAsync(c->L2, 0, (size_t)2 * c->P * 16, cur_stream(c)));
  HIPCHK(c, hipMemsetAsync(c->S2, 0, (size_t)2 * c->P * 16, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  clear_tracker_state(c);
  return 0;
}

int esvio_fe_create_sae_stereo(esvio_fe_handle c, const esvio_fe_event* left, size_t nL,
                               const esvio_fe_event* right, size_t nR, int space,
                               uint64_t* n_rejected) {
  if (!c || (nL && !left) || (nR && !right)) return ESVIO_FE_EINVAL;
  if (nL + nR >= (1ull << 31)) return fail(c, ESVIO_FE_EINVAL, "batch too large");
  if (!c->inflight.empty()) return fail(c, ESVIO_FE_EINVAL, "a prefetched batch is pending");
  HIPCHK(c, hipSetDevice(c->dev));
  const EventRec *dL, *dR;
  if (int rc = stage_events(c, left, nL, right, nR, space, &dL, &dR)) return rc;
  HIPCHK(c, hipMemsetAsync(c->d_rejected, 0, 8, cur_stream(c)));
  if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR)) return rc;
  unsigned long long rej = 0;
  HIPCHK(c, hipMemcpyAsync(&rej, c->d_rejected, 8, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (pin_of(c).counts[3]) return fail(c, ESVIO_FE_EINTERNAL, "radix sort look-back spin expired");
  if (n_rejected) *n_rejected = rej;
  if (c->prof_on) resolve_profile(c);
  return 0;
}

int esvio_fe_create_sae_stereo_mc(esvio_fe_handle c, const esvio_fe_event* left, size_t nL,
                                  const esvio_fe_event* right, size_t nR, int space,
                                  const esvio_fe_motion* motion, uint64_t* n_rejected) {
  if (!c || !motion || !nL || !left || (nR && !right)) return ESVIO_FE_EINVAL;
  if (nL + nR >= (1ull << 31)) return fail(c, ESVIO_FE_EINVAL, "batch too large");
  if (!c->inflight.empty()) return fail(c, ESVIO_FE_EINVAL, "a prefetched batch is pending");
  HIPCHK(c, hipSetDevice(c->dev));
  esvio_fe_event first;
  if (int rc = first_event_host(c, left, space, &first)) return rc;
  const McParams mc = make_mc_params(motion, first);
  const EventRec *dL, *dR;
  if (int rc = stage_events(c, left, nL, right, nR, space, &dL, &dR)) return rc;
  HIPCHK(c, hipMemsetAsync(c->d_rejected, 0, 8, cur_stream(c)));
  if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR, &mc)) return rc;
  unsigned long long rej = 0;
  HIPCHK(c, hipMemcpyAsync(&rej, c->d_rejected, 8, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (pin_of(c).counts[3]) return fail(c, ESVIO_FE_EINTERNAL, "radix sort look-back spin expired");
  if (n_rejected) *n_rejected = rej;
  if (c->prof_on) resolve_profile(c);
  return 0;
}

// ---- one stream time-sliced across GPUs -----------------------------------------
namespace {
int slice_scratch(esvio_fe_ctx* c) {
  if (c->L2s) return 0;
  if (int rc = dev_alloc(c, &c->L2s, (size_t)2 * c->P)) return rc;
  if (int rc = dev_alloc(c, &c->S2s, (size_t)2 * c->P)) return rc;
  return 0;
}
// device address of `k` consecutive plane sets given in `space` (host ones are staged)
int slice_planes_in(esvio_fe_ctx* c, const double* p, size_t sets, int space, const double** dev) {
  const size_t nd = sets * 4 * (size_t)c->P;
  if (space == ESVIO_FE_DEVICE || !nd) {
    *dev = p;
    return 0;
  }
  if (space != ESVIO_FE_HOST) return fail(c, ESVIO_FE_EINVAL, "bad memory space %d", space);
  if (nd > c->slice_stage_doubles) {
    if (c->slice_stage) (void)hipFree(c->slice_stage);
    c->slice_stage = nullptr;
    c->slice_stage_doubles = 0;
    if (int rc = dev_alloc(c, &c->slice_stage, nd)) return rc;
    c->slice_stage_doubles = nd;
  }
  HIPCHK(c, hipMemcpyAsync(c->slice_stage, p, nd * 8, hipMemcpyHostToDevice, cur_stream(c)));
  *dev = c->slice_stage;
  return 0;
}
int slice_planes_out(esvio_fe_ctx* c, const double2* src, double* out, int space) {
  const size_t bytes = (size_t)4 * c->P * 8;
  HIPCHK(c, hipMemcpyAsync(out, src, bytes, space == ESVIO_FE_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                           cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (pin_of(c).counts[3]) return fail(c, ESVIO_FE_EINTERNAL, "radix sort look-back spin expired");
  return 0;
}
}  // namespace

size_t esvio_fe_sae_plane_doubles(esvio_fe_handle c) { return c ? (size_t)4 * c->P : 0; }

int esvio_fe_sae_slice_last(esvio_fe_handle c, const esvio_fe_event* left, size_t nL,
                            const esvio_fe_event* right, size_t nR, int space, double* last_out,
                            int out_space) {
  if (!c || !last_out || (nL && !left) || (nR && !right)) return ESVIO_FE_EINVAL;
  if (nL + nR >= (1ull << 31)) return fail(c, ESVIO_FE_EINVAL, "batch too large");
  if (!c->inflight.empty() || !c->announced.empty())
    return fail(c, ESVIO_FE_EINVAL, "time-sliced SAE update cannot be mixed with esvio_fe_set_next_batch");
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = slice_scratch(c)) return rc;
  const EventRec *dL, *dR;
  if (int rc = stage_events(c, left, nL, right, nR, space, &dL, &dR)) return rc;
  // L[p] is overwritten by every event whatever the carried-in state is (event_detector.cc:158), so
  // the slice's last event time per (pixel, polarity) is what the ordinary update leaves in planes
  // that start out as "nothing"
  launch_fill_f64(cur_stream(c), (double*)c->L2s, (size_t)4 * c->P, kSliceNone);
  launch_fill_f64(cur_stream(c), (double*)c->S2s, (size_t)4 * c->P, kSliceNone);
  if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR, nullptr, c->L2s, c->S2s)) return rc;
  return slice_planes_out(c, c->L2s, last_out, out_space);
}

int esvio_fe_sae_slice_apply(esvio_fe_handle c, const esvio_fe_event* left, size_t nL,
                             const esvio_fe_event* right, size_t nR, int space,
                             const double* last_before, int n_before, int in_space, double* s_out,
                             int out_space) {
  if (!c || !s_out || n_before < 0 || (n_before && !last_before) || (nL && !left) || (nR && !right))
    return ESVIO_FE_EINVAL;
  if (nL + nR >= (1ull << 31)) return fail(c, ESVIO_FE_EINVAL, "batch too large");
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = slice_scratch(c)) return rc;
  const size_t nd = (size_t)4 * c->P;
  // carried-in L = the planes before the batch overlaid with the earlier slices in stream order;
  // with it every pass decision of this slice is the sequential loop's (S never enters a decision)
  HIPCHK(c, hipMemcpyAsync(c->L2s, c->L2, nd * 8, hipMemcpyDeviceToDevice, cur_stream(c)));
  const double* dl = nullptr;
  if (int rc = slice_planes_in(c, last_before, (size_t)n_before, in_space, &dl)) return rc;
  for (int k = 0; k < n_before; k++)
    launch_overlay_f64(cur_stream(c), (double*)c->L2s, dl + (size_t)k * nd, nd, kSliceNone);
  launch_fill_f64(cur_stream(c), (double*)c->S2s, nd, kSliceNone);
  const EventRec *dL, *dR;
  if (int rc = stage_events(c, left, nL, right, nR, space, &dL, &dR)) return rc;
  if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR, nullptr, c->L2s, c->S2s)) return rc;
  return slice_planes_out(c, c->S2s, s_out, out_space);
}

int esvio_fe_sae_slice_commit(esvio_fe_handle c, const double* last_all, const double* s_all, int n_slices,
                              int space) {
  if (!c || n_slices < 1 || !last_all || !s_all) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t nd = (size_t)4 * c->P;
  for (int pass = 0; pass < 2; pass++) {  // (one staging buffer: L first, then S)
    const double* dp = nullptr;
    if (int rc = slice_planes_in(c, pass ? s_all : last_all, (size_t)n_slices, space, &dp)) return rc;
    double* dst = pass ? (double*)c->S2 : (double*)c->L2;
    for (int k = 0; k < n_slices; k++) launch_overlay_f64(cur_stream(c), dst, dp + (size_t)k * nd, nd, kSliceNone);
    HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  }
  c->ext_sae_pending = true;
  return 0;
}

int esvio_fe_create_sae(esvio_fe_handle c, int cam, const esvio_fe_event* ev, size_t n, int space,
                        uint64_t* n_rejected) {
  if (cam == 0) return esvio_fe_create_sae_stereo(c, ev, n, nullptr, 0, space, n_rejected);
  if (cam == 1) return esvio_fe_create_sae_stereo(c, nullptr, 0, ev, n, space, n_rejected);
  return ESVIO_FE_EINVAL;
}

int esvio_fe_sae_to_time_surface(esvio_fe_handle c, int cam, double t_sync, uint8_t* out) {
  if (!c || (cam != 0 && cam != 1)) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (!c->inflight.empty()) return fail(c, ESVIO_FE_EINVAL, "a prefetched batch is pending");
  render_lk_images(c, t_sync, cam ? 2 : 1, c->slot_curL, c->slot_curR, c->raw_cur);
  if (out) return copy_level0_out(c, raw_ts_desc(c, cam), out);
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (c->prof_on) resolve_profile(c);
  return 0;
}

int esvio_fe_get_time_surface(esvio_fe_handle c, int cam, uint8_t* out) {
  if (!c || !out || (cam != 0 && cam != 1)) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  return copy_level0_out(c, raw_ts_desc(c, cam), out);
}

int esvio_fe_export_image(esvio_fe_handle c, int cam, uint8_t* dst, int space) {
  if (!c || !dst || (cam != 0 && cam != 1)) return ESVIO_FE_EINVAL;
  if (space != ESVIO_FE_HOST && space != ESVIO_FE_DEVICE) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  const PyrDesc& d = cam ? c->pyr[c->slot_curR].d : c->pyr[c->slot_curL].d;
  const int stride = d.stride[0];
  HIPCHK(c, hipMemcpy2DAsync(dst, c->W, d.img[0] + (size_t)kPad * stride + kPad, stride, c->W, c->H,
                             space == ESVIO_FE_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                             cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  return 0;
}

int esvio_fe_import_image(esvio_fe_handle c, int cam, const uint8_t* src, int space) {
  if (!c || !src) return ESVIO_FE_EINVAL;
  if (cam != 1) return fail(c, ESVIO_FE_EINVAL, "only the right camera's image can be imported");
  if (space != ESVIO_FE_HOST && space != ESVIO_FE_DEVICE) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (!c->announced.empty() || !c->inflight.empty())
    return fail(c, ESVIO_FE_EINVAL, "import_image cannot be combined with set_next_batch");
  c->slot_curR = c->slot_curR == kLeftSlots ? kLeftSlots + 1 : kLeftSlots;  // next trackEvent's curR
  const PyrDesc& d = c->pyr[c->slot_curR].d;
  const int stride = d.stride[0];
  HIPCHK(c, hipMemcpy2DAsync(d.img[0] + (size_t)kPad * stride + kPad, stride, src, c->W, c->W, c->H,
                             space == ESVIO_FE_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                             cur_stream(c)));
  if (space == ESVIO_FE_HOST) HIPCHK(c, hipStreamSynchronize(cur_stream(c)));  // caller may reuse src
  c->ext_right_pending = true;
  return 0;
}

int esvio_fe_is_corner(esvio_fe_handle c, const esvio_fe_event* ev, size_t n, int space,
                       uint8_t* flags) {
  if (!c || (n && (!ev || !flags))) return ESVIO_FE_EINVAL;
  if (!n) return 0;
  HIPCHK(c, hipSetDevice(c->dev));
  const EventRec *dL, *dR;
  if (int rc = stage_events(c, ev, n, nullptr, 0, space, &dL, &dR)) return rc;
  if (int rc = ensure_arc_capacity(c, n, c->cand_cur)) return rc;
  run_arc(c, dL, (uint32_t)n, nullptr, false, true, false, c->cand_cur);
  HIPCHK(c, hipMemcpyAsync(flags, c->d_flags, n, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (c->prof_on) resolve_profile(c);
  return 0;
}

int esvio_fe_features_to_track(esvio_fe_handle c, const esvio_fe_event* ev, size_t n, int space,
                               int max_corners, const uint8_t* mask, float* out_xy,
                               int32_t* out_idx, int32_t* n_out) {
  if (!c || !n_out || (n && !ev)) return ESVIO_FE_EINVAL;
  *n_out = 0;
  if (max_corners <= 0 || !n) return 0;
  if (max_corners > c->cfg.max_cnt) return fail(c, ESVIO_FE_EINVAL, "max_corners > max_cnt");
  if (!out_xy) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  const EventRec *dL, *dR;
  if (int rc = stage_events(c, ev, n, nullptr, 0, space, &dL, &dR)) return rc;
  if (int rc = ensure_arc_capacity(c, n, c->cand_cur)) return rc;
  Pin pin = pin_of(c);
  host::BitMask bm;
  bm.reset(c->W, c->H);
  if (mask) bm.from_bytes(mask);
  std::memcpy(pin.mask, bm.bits.data(), bm.bits.size() * 4);
  HIPCHK(c, hipMemcpyAsync(c->d_mask_bits, pin.mask, bm.bits.size() * 4, hipMemcpyHostToDevice,
                           cur_stream(c)));
  const PyrDesc ts = raw_ts_desc(c, 0);
  run_arc(c, dL, (uint32_t)n, &ts, true, false, true, c->cand_cur);
  run_compact(c, (uint32_t)n, c->cand_cur);
  run_select(c, c->cand_cur, max_corners, c->d_ptsD, 0, c->d_sel_idx);
  HIPCHK(c, hipMemcpyAsync(pin.counts, c->d_counts, 8, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  const int k = pin.counts[0];
  if (k > 0) {
    HIPCHK(c, hipMemcpy(out_xy, c->d_ptsD, (size_t)k * 8, hipMemcpyDeviceToHost));
    if (out_idx) HIPCHK(c, hipMemcpy(out_idx, c->d_sel_idx, (size_t)k * 4, hipMemcpyDeviceToHost));
  }
  *n_out = k;
  if (c->prof_on) resolve_profile(c);
  return 0;
}

static int planes_io(esvio_fe_handle c, int cam, double* L0, double* L1, double* S0, double* S1,
                     const double* iL0, const double* iL1, const double* iS0, const double* iS1,
                     bool set) {
  if (!c || (cam != 0 && cam != 1)) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  std::vector<double> l((size_t)2 * c->P), s((size_t)2 * c->P);
  if (set) {
    for (uint32_t i = 0; i < c->P; i++) {
      l[2 * i] = iL0[i];
      l[2 * i + 1] = iL1[i];
      s[2 * i] = iS0[i];
      s[2 * i + 1] = iS1[i];
    }
    HIPCHK(c, hipMemcpy(c->L2 + (size_t)cam * c->P, l.data(), (size_t)c->P * 16, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->S2 + (size_t)cam * c->P, s.data(), (size_t)c->P * 16, hipMemcpyHostToDevice));
  } else {
    HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
    HIPCHK(c, hipMemcpy(l.data(), c->L2 + (size_t)cam * c->P, (size_t)c->P * 16, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(s.data(), c->S2 + (size_t)cam * c->P, (size_t)c->P * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < c->P; i++) {
      L0[i] = l[2 * i];
      L1[i] = l[2 * i + 1];
      S0[i] = s[2 * i];
      S1[i] = s[2 * i + 1];
    }
  }
  return 0;
}

int esvio_fe_get_sae(esvio_fe_handle c, int cam, double* L0, double* L1, double* S0, double* S1) {
  if (!L0 || !L1 || !S0 || !S1) return ESVIO_FE_EINVAL;
  return planes_io(c, cam, L0, L1, S0, S1, nullptr, nullptr, nullptr, nullptr, false);
}
int esvio_fe_set_sae(esvio_fe_handle c, int cam, const double* L0, const double* L1,
                     const double* S0, const double* S1) {
  if (!L0 || !L1 || !S0 || !S1) return ESVIO_FE_EINVAL;
  return planes_io(c, cam, nullptr, nullptr, nullptr, nullptr, L0, L1, S0, S1, true);
}

static int prep_tmp_pyr(esvio_fe_handle c, int slot, const uint8_t* img, int w, int hgt,
                        int max_level) {
  if (int rc = pyr_alloc(c, c->tmp_pyr[slot], w, hgt, max_level)) return rc;
  return copy_level0_in(c, c->tmp_pyr[slot].d, img);
}

int esvio_fe_calc_optical_flow_pyr_lk(esvio_fe_handle c, const uint8_t* prev_img,
                                      const uint8_t* next_img, int w, int hgt,
                                      const float* prev_pts, float* next_pts, uint8_t* status,
                                      int n, int max_level, int max_count, double eps, int flags) {
  if (!c || !prev_img || !next_img || n < 0 || (n && (!prev_pts || !next_pts || !status)))
    return ESVIO_FE_EINVAL;
  if (w < 2 * kLkWin || hgt < 2 * kLkWin || max_level < 0 || max_level >= kMaxLevels)
    return ESVIO_FE_EINVAL;
  if (n > c->cfg.max_cnt) return fail(c, ESVIO_FE_EINVAL, "n > max_cnt");
  if (!n) return 0;
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = prep_tmp_pyr(c, 0, prev_img, w, hgt, max_level)) return rc;
  if (int rc = prep_tmp_pyr(c, 1, next_img, w, hgt, max_level)) return rc;
  PyrDesc two[2] = {c->tmp_pyr[0].d, c->tmp_pyr[1].d};
  pyr_build(c, two, 2);
  HIPCHK(c, hipMemcpyAsync(c->d_ptsA, prev_pts, (size_t)n * 8, hipMemcpyHostToDevice, cur_stream(c)));
  if (flags & ESVIO_FE_LK_USE_INITIAL_FLOW)
    HIPCHK(c, hipMemcpyAsync(c->d_ptsB, next_pts, (size_t)n * 8, hipMemcpyHostToDevice, cur_stream(c)));
  LkArgs f = make_lk(two[0], two[1], c->d_ptsA, c->d_ptsB, c->d_ptsB, c->d_stA, nullptr, n, max_level,
                     max_count, eps, flags);
  run_lk(c, f, nullptr, nullptr, nullptr);
  HIPCHK(c, hipMemcpyAsync(next_pts, c->d_ptsB, (size_t)n * 8, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipMemcpyAsync(status, c->d_stA, (size_t)n, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (c->prof_on) resolve_profile(c);
  return 0;
}

int esvio_fe_build_pyramid(esvio_fe_handle c, const uint8_t* img, int w, int hgt, int max_level,
                           int level, uint8_t* out_img, int16_t* out_deriv, int32_t* lw,
                           int32_t* lh, int32_t* n_levels) {
  if (!c || !img || max_level < 0 || max_level >= kMaxLevels || w < 2 * kLkWin || hgt < 2 * kLkWin)
    return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = prep_tmp_pyr(c, 0, img, w, hgt, max_level)) return rc;
  const PyrDesc& d = c->tmp_pyr[0].d;
  pyr_build(c, &d, 1);
  if (n_levels) *n_levels = d.levels + 1;
  if (level < 0 || level > d.levels) {
    HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
    return level < 0 ? 0 : ESVIO_FE_EINVAL;
  }
  if (lw) *lw = d.w[level];
  if (lh) *lh = d.h[level];
  const int stride = d.stride[level];
  if (out_img)
    HIPCHK(c, hipMemcpy2DAsync(out_img, d.w[level], d.img[level] + (size_t)kPad * stride + kPad,
                               stride, d.w[level], d.h[level], hipMemcpyDeviceToHost, cur_stream(c)));
  if (out_deriv)
    HIPCHK(c, hipMemcpy2DAsync(out_deriv, (size_t)d.w[level] * 4,
                               d.deriv[level] + ((size_t)kPad * stride + kPad) * 2, (size_t)stride * 4,
                               (size_t)d.w[level] * 4, d.h[level], hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (c->prof_on) resolve_profile(c);
  return 0;
}

int esvio_fe_find_fundamental_mat(const float* p1, const float* p2, int n, double thr, double conf,
                                  uint8_t* status, int32_t* n_inliers) {
  if (n < 0 || (n && (!p1 || !p2 || !status))) return ESVIO_FE_EINVAL;
  const int k = host::find_fundamental_mat(p1, p2, n, thr, conf, status);
  if (n_inliers) *n_inliers = k;
  return 0;
}

int esvio_fe_find_fundamental_mat_mt(const float* p1, const float* p2, int n, double thr, double conf,
                                     int threads, uint8_t* status, int32_t* n_inliers) {
  if (n < 0 || (n && (!p1 || !p2 || !status)) || threads < 1 || threads > 16) return ESVIO_FE_EINVAL;
  host::RansacPool* pool = host::ransac_pool_create(threads - 1);
  const int k = host::find_fundamental_mat(p1, p2, n, thr, conf, status, pool);
  host::ransac_pool_destroy(pool);
  if (n_inliers) *n_inliers = k;
  return 0;
}

int esvio_fe_lift_projective(const esvio_fe_camera* cam, double u, double v, double* out3) {
  if (!cam || !out3) return ESVIO_FE_EINVAL;
  host::lift_projective(*cam, u, v, out3);
  return 0;
}

static int fill_tracks(esvio_fe_handle c, esvio_fe_tracks* out);
static int track_event_entry(esvio_fe_handle c, double cur_time, const esvio_fe_event* left,
                             size_t nL, const esvio_fe_event* right, size_t nR, int space,
                             int pub_this_frame, const esvio_fe_motion* motion,
                             esvio_fe_tracks* out) {
  if (!c) return ESVIO_FE_EINVAL;
  if (nL == 0 || !left) return fail(c, ESVIO_FE_EINVAL, "left batch must not be empty (node:150)");
  if (nR && !right) return ESVIO_FE_EINVAL;
  if (nL + nR >= (1ull << 31)) return fail(c, ESVIO_FE_EINVAL, "batch too large");
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = track_event_impl(c, cur_time, left, nL, right, nR, space, pub_this_frame != 0, motion))
    return rc;
  return fill_tracks(c, out);
}

// copy the result members (feature_tracker.h:126-138) into the caller's buffers
static int fill_tracks(esvio_fe_handle c, esvio_fe_tracks* out) {
  if (out) {
    out->n_left = (int32_t)c->ids.size();
    out->n_right = (int32_t)c->ids_right.size();
    const size_t nl = c->ids.size(), nr = c->ids_right.size();
    if (out->ids) std::memcpy(out->ids, c->ids.data(), nl * 4);
    if (out->track_cnt) std::memcpy(out->track_cnt, c->track_cnt.data(), nl * 4);
    if (out->cur_pts) std::memcpy(out->cur_pts, c->cur_pts.data(), nl * 8);
    if (out->cur_un_pts) std::memcpy(out->cur_un_pts, c->cur_un_pts.data(), nl * 8);
    if (out->pts_velocity) std::memcpy(out->pts_velocity, c->pts_velocity.data(), nl * 8);
    if (out->ids_right) std::memcpy(out->ids_right, c->ids_right.data(), nr * 4);
    if (out->cur_right_pts) std::memcpy(out->cur_right_pts, c->cur_right_pts.data(), nr * 8);
    if (out->cur_un_right_pts) std::memcpy(out->cur_un_right_pts, c->cur_un_right_pts.data(), nr * 8);
    if (out->right_pts_velocity)
      std::memcpy(out->right_pts_velocity, c->right_pts_velocity.data(), nr * 8);
  }
  return 0;
}

int esvio_fe_track_event(esvio_fe_handle c, double cur_time, const esvio_fe_event* left, size_t nL,
                         const esvio_fe_event* right, size_t nR, int space, int pub_this_frame,
                         esvio_fe_tracks* out) {
  return track_event_entry(c, cur_time, left, nL, right, nR, space, pub_this_frame, nullptr, out);
}

int esvio_fe_track_event_mc(esvio_fe_handle c, double cur_time, const esvio_fe_event* left,
                            size_t nL, const esvio_fe_event* right, size_t nR, int space,
                            int pub_this_frame, const esvio_fe_motion* motion,
                            esvio_fe_tracks* out) {
  if (!motion) return ESVIO_FE_EINVAL;
  return track_event_entry(c, cur_time, left, nL, right, nR, space, pub_this_frame, motion, out);
}

// ---- image front-end
int esvio_fe_good_features_to_track(esvio_fe_handle c, const uint8_t* img, int max_corners,
                                    double quality, double min_distance, const uint8_t* mask,
                                    float* out_xy, int32_t* n_out, float* eig_out) {
  if (!c || !img || !n_out) return ESVIO_FE_EINVAL;
  *n_out = 0;
  if (max_corners <= 0 || max_corners > c->cfg.max_cnt)
    return fail(c, ESVIO_FE_EINVAL, "max_corners must be in 1..max_cnt");
  if (!(quality > 0) || min_distance < 1 || min_distance > kMaxDiscR)
    return fail(c, ESVIO_FE_EINVAL, "quality must be > 0, min_distance in [1, %d]", kMaxDiscR);
  if (!out_xy) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (!c->inflight.empty()) return fail(c, ESVIO_FE_EINVAL, "a prefetched batch is pending");
  if (int rc = prep_tmp_pyr(c, 0, img, c->W, c->H, 0)) return rc;
  PyrDesc d = c->tmp_pyr[0].d;
  pyr_build(c, &d, 1);  // materialises the reflect-101 border the Sobel taps read
  Pin pin = pin_of(c);
  if (mask) {  // nonzero = allowed; the device bitmap holds the BLOCKED pixels
    host::BitMask bm;
    bm.reset(c->W, c->H);
    for (int y = 0; y < c->H; y++)
      for (int x = 0; x < c->W; x++)
        if (!mask[(size_t)y * c->W + x]) bm.bits[(size_t)y * bm.wpr + (x >> 5)] |= 1u << (x & 31);
    std::memcpy(pin.mask, bm.bits.data(), bm.bits.size() * 4);
    HIPCHK(c, hipMemcpyAsync(c->d_mask_bits, pin.mask, bm.bits.size() * 4, hipMemcpyHostToDevice,
                             cur_stream(c)));
  }
  if (int rc = gftt_run(c, d, max_corners, quality, min_distance, mask != nullptr, c->z_new, 0,
                        c->z_counts))
    return rc;
  if (eig_out)
    HIPCHK(c, hipMemcpyAsync(eig_out, c->d_gftt_eig, (size_t)c->W * c->H * 4, hipMemcpyDeviceToHost,
                             cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  if (pin.counts[3]) return fail(c, ESVIO_FE_EINTERNAL, "radix sort look-back spin expired");
  const int k = pin.counts[0];
  std::memcpy(out_xy, pin.news, (size_t)k * 8);
  *n_out = k;
  if (c->prof_on) resolve_profile(c);
  return 0;
}

int esvio_fe_track_image(esvio_fe_handle c, double cur_time, const uint8_t* img_left,
                         const uint8_t* img_right, int pub_this_frame, esvio_fe_tracks* out) {
  if (!c || !img_left) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (!c->announced.empty() || !c->inflight.empty())
    return fail(c, ESVIO_FE_EINVAL, "event batches are announced on this handle");
  if (int rc = track_image_impl(c, cur_time, img_left, img_right, pub_this_frame != 0)) return rc;
  return fill_tracks(c, out);
}

// The node's sensor_msgs/PointCloud packing (stereo_event_tracker_node.cpp:273-329) of the current
// result members, as a fixed-size block: left entries with track_cnt > 1, then right entries whose
// id is among them; rows (x_un, y_un, 1, id*2+cam as float32, u, v, vx, vy); padding rows id -1.
int esvio_fe_pack_track_records(esvio_fe_handle c, float* out, int32_t* n_rows) {
  if (!c || !out) return ESVIO_FE_EINVAL;
  if (c->pend_right.active) {  // (lazy mode, packing a frame that was not to be published)
    HIPCHK(c, hipSetDevice(c->dev));
    if (int rc = finalize_right(c)) return rc;
  }
  const int rows = 2 * std::max(c->cfg.max_cnt, 1);
  int k = 0;
  std::vector<int> left_ids;
  left_ids.reserve(c->ids.size());
  for (size_t j = 0; j < c->ids.size() && k < rows; j++)
    if (c->track_cnt[j] > 1) {
      float* r = out + (size_t)k++ * 8;
      r[0] = c->cur_un_pts[j].x;
      r[1] = c->cur_un_pts[j].y;
      r[2] = 1.f;
      r[3] = (float)(c->ids[j] * 2 + 0);
      r[4] = c->cur_pts[j].x;
      r[5] = c->cur_pts[j].y;
      r[6] = c->pts_velocity[j].x;
      r[7] = c->pts_velocity[j].y;
      left_ids.push_back(c->ids[j]);
    }
  std::sort(left_ids.begin(), left_ids.end());
  for (size_t j = 0; j < c->ids_right.size() && k < rows; j++)
    if (std::binary_search(left_ids.begin(), left_ids.end(), c->ids_right[j])) {
      float* r = out + (size_t)k++ * 8;
      r[0] = c->cur_un_right_pts[j].x;
      r[1] = c->cur_un_right_pts[j].y;
      r[2] = 1.f;
      r[3] = (float)(c->ids_right[j] * 2 + 1);
      r[4] = c->cur_right_pts[j].x;
      r[5] = c->cur_right_pts[j].y;
      r[6] = c->right_pts_velocity[j].x;
      r[7] = c->right_pts_velocity[j].y;
    }
  if (n_rows) *n_rows = k;
  for (; k < rows; k++) {
    float* r = out + (size_t)k * 8;
    for (int i = 0; i < 8; i++) r[i] = 0.f;
    r[3] = -1.f;
  }
  return 0;
}

// ---- RCCL hand-off (north-star: "a single RCCL all-gather over xGMI to merge tracked corners") ----
// RCCL is looked up at run time (dlopen librccl.so), so the library does not depend on it unless
// this entry point is used.
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/,
                                 hipStream_t);
nccl_allgather_fn rccl_all_gather() {
  static nccl_allgather_fn fn = []() -> nccl_allgather_fn {
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return lib ? (nccl_allgather_fn)dlsym(lib, "ncclAllGather") : nullptr;
  }();
  return fn;
}
}  // namespace

int esvio_fe_exchange_tracks(esvio_fe_handle c, void* nccl_comm, int world, float* gathered) {
  if (!c || !nccl_comm || world < 1 || !gathered) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  nccl_allgather_fn all_gather = rccl_all_gather();
  if (!all_gather) return fail(c, ESVIO_FE_ENOTIMPL, "librccl.so not found (dlopen): %s", dlerror());
  const size_t rows = (size_t)2 * std::max(c->cfg.max_cnt, 1), cnt = rows * 8;
  if (!c->x_send) {
    if (int rc = dev_alloc(c, &c->x_send, cnt)) return rc;
    HIPCHK(c, hipHostMalloc((void**)&c->x_pin, cnt * sizeof(float), hipHostMallocDefault));
  }
  if ((size_t)world * cnt > c->x_recv_cap) {
    if (c->x_recv) (void)hipFree(c->x_recv);
    c->x_recv = nullptr;
    c->x_recv_cap = 0;
    if (int rc = dev_alloc(c, &c->x_recv, (size_t)world * cnt)) return rc;
    c->x_recv_cap = (size_t)world * cnt;
  }
  if (int rc = esvio_fe_pack_track_records(c, c->x_pin, nullptr)) return rc;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->x_send, c->x_pin, cnt * 4, hipMemcpyHostToDevice, st));
  const int nrc = all_gather(c->x_send, c->x_recv, cnt, 7 /* ncclFloat32 */, nccl_comm, st);
  if (nrc != 0) return fail(c, ESVIO_FE_EHIP, "ncclAllGather failed: %d", nrc);
  HIPCHK(c, hipMemcpyAsync(gathered, c->x_recv, (size_t)world * cnt * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return 0;
}

int esvio_fe_set_lazy_new_stereo(esvio_fe_handle c, int on) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = finalize_pending(c)) return rc;
  if (int rc = finalize_right(c)) return rc;
  c->lazy_new = on != 0;
  return 0;
}

int esvio_fe_set_host_threads(esvio_fe_handle c, int threads) {
  if (!c || threads < 1 || threads > 16) return ESVIO_FE_EINVAL;
  host::ransac_pool_destroy(c->pool);
  c->pool = host::ransac_pool_create(threads - 1);
  return 0;
}

int esvio_fe_finish(esvio_fe_handle c, esvio_fe_tracks* out) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = finalize_pending(c)) return rc;
  if (int rc = finalize_right(c)) return rc;
  return fill_tracks(c, out);
}

int esvio_fe_set_next_batch(esvio_fe_handle c, double next_cur_time, const esvio_fe_event* left,
                            size_t nL, const esvio_fe_event* right, size_t nR, int space,
                            int pub_hint) {
  if (!c) return ESVIO_FE_EINVAL;
  if (nL == 0 || !left || (nR && !right)) return fail(c, ESVIO_FE_EINVAL, "bad next batch");
  if (space != ESVIO_FE_HOST && space != ESVIO_FE_DEVICE) return ESVIO_FE_EINVAL;
  if (c->ext_right_pending) return fail(c, ESVIO_FE_EINVAL, "not with an imported right image");
  if ((int)c->announced.size() >= kPrefetchDepth)
    return fail(c, ESVIO_FE_EINVAL, "at most %d batches can be announced ahead", kPrefetchDepth);
  Batch b;
  b.time = next_cur_time;
  b.left = left;
  b.nL = nL;
  b.right = right;
  b.nR = nR;
  b.space = space;
  b.pub = pub_hint != 0;
  c->announced.push_back(b);
  return 0;
}

int esvio_fe_set_profiling(esvio_fe_handle c, int on) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  resolve_profile(c);
  c->prof_on = on != 0;
  return 0;
}
int esvio_fe_device_memory(esvio_fe_handle c, size_t* free_bytes, size_t* total_bytes) {
  if (!c || !free_bytes || !total_bytes) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}

int esvio_fe_kernel_count(void) { return K_COUNT; }
const char* esvio_fe_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int esvio_fe_get_kernel_stats(esvio_fe_handle c, int id, double* total_ms, uint64_t* launches,
                              uint64_t* alg_bytes) {
  if (!c || id < 0 || id >= K_COUNT) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  resolve_profile(c);
  if (total_ms) *total_ms = c->stats[id].ms;
  if (launches) *launches = c->stats[id].launches;
  if (alg_bytes) *alg_bytes = c->stats[id].bytes;
  return 0;
}
int esvio_fe_reset_kernel_stats(esvio_fe_handle c) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  resolve_profile(c);
  for (auto& s : c->stats) s = KStat();
  return 0;
}
void* esvio_fe_stream(esvio_fe_handle c) { return c ? (void*)cur_stream(c) : nullptr; }

}  // extern "C"
