// fe_ctx.h — the handle (esvio_fe_ctx) and the small helpers every part of the library uses.
// Internal: the public boundary is include/esvio_fe.h.
//
//   fe_stages.cpp  device memory, the per-stage launches (SAE update, rendering, pyramids, LK, Arc*,
//                  selection) and the <= max_cnt-point host bookkeeping of the reference
//   fe_track.cpp   FeatureTracker::trackEvent: the per-frame sequence and its replay-mode scheduler
//                  (prefetch stream, speculative / chained temporal LK, lazy right-camera tails)
//   fe_image.cpp   FeatureTracker::trackImage and goodFeaturesToTrack
//   fe_api.cpp     the C ABI entry points
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
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
#include "../../include/esvio_fe_test.h"
#include "fe_host.h"
#include "fe_kernels.h"
#include "fe_mc.h"

using namespace esvio;

namespace esvio {
namespace fe {

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
    "k_tile_hist", "k_tile_scan", "k_tile_scatter", "k_tile_apply", "k_sae_keys", "k_radix_pass", "k_sae_apply",
    "k_time_surface4", "k_time_surface", "k_median", "k_clahe", "k_norm_pyr", "k_pyr3", "k_pyr_down", "k_pyr_pad",
    "k_scharr", "k_pad_scharr", "k_lk_f32", "k_lk", "k_arc_map", "k_arc_ev", "k_dedup", "k_compact", "k_select_mw",
    "k_select", "k_select_gbm"};

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
inline int pyr_levels(int w, int h, int win, int max_level) {
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
  int stage = -1;  // host events: the staging slot they travel through (fe_evstage.cpp), -1: none
  bool has_motion = false;  // esvio_fe_set_next_batch_mc: the motion-compensated overload
  esvio_fe_motion motion{};
};
// ... and, once its SAE update / images / pyramids (/ Arc*) are enqueued on the prefetch stream,
// the resources they were given
struct Inflight : Batch {
  int lane = 0;  // staging buffer + event pair
  int slotL = 0, slotR = 0, raw = 0, cand = 0;
  const EventRec *dL = nullptr, *dR = nullptr;
  bool arc_done = false;
  uint32_t gate = 0;  // != 0: the value the batch's prefetch sequence leaves in d_lane_gate[lane] (issued by the launch thread)
};
constexpr int kPrefetchDepth = 3;
// host-event staging slots: one per batch the handle can know about at a time (2 * kPrefetchDepth
// announced or prefetched and not yet tracked, the one being tracked, one spare)
constexpr int kStageSlots = 2 * kPrefetchDepth + 2;
struct EventStager;  // fe_evstage.cpp
struct Launcher;     // fe_track.cpp: the thread that issues announced batches' prefetch sequences
constexpr int kLeftSlots = 2 + kPrefetchDepth;   // prev, cur, prefetched...
constexpr int kRightSlots = 1 + kPrefetchDepth;  // cur, prefetched...

struct PyrStore {
  PyrDesc d{};
  void* mem = nullptr;
  size_t bytes = 0;
  int w = 0, h = 0, max_level = -1;
};

}  // namespace fe
}  // namespace esvio
using namespace esvio::fe;

struct esvio_fe_ctx {
  esvio_fe_config cfg{};
  int dev = 0;
  hipStream_t stream = nullptr;   // main stream
  hipStream_t stream2 = nullptr;  // prefetch stream (next batch's SAE update / images)
  hipStream_t stream3 = nullptr;  // speculative temporal LK of the next frame
  // stereo LK of the temporal survivors: nothing on the frame's chain reads its results before the
  // right-camera tail, and on the main stream it would hold up the corner selection behind it
  hipStream_t stream4 = nullptr;
  // ... of an UNPUBLISHED frame (stereo_stream() below).  The chained temporal LK of the frame after next sits on
  // stream4 from the published call that launched it until its last point is done; the unpublished frame's stereo
  // LK queued behind it, the next published frame's behind that, and the published call waited 35-45 us for its
  // stereo results at its end (and 10-55 for the previous frame's): with the unpublished frame's launch on a stream
  // of its own both waits are gone — cycle 272 -> 246 us in the trace, 0.131 -> 0.119 ms/step.  (WHICH launches
  // share a stream matters more than how many streams there are: the published frame's stereo LK and the chained
  // launch on the new stream instead — `k_select_mw` 17 -> 40 us, cycle 300 us; KERNELS.md.)
  hipStream_t stream6 = nullptr;
  std::vector<double> pre_lx, pre_ly;  // prev_pts lifted through the left camera model under the temporal LK's wait
  bool pre_lift_valid = false;
  bool stereo_unpub = false;  // the frame being tracked publishes nothing (and stereo_split is on)
  // -1: lk_accum == 2 && launch thread on (decided per call); 0 / 1: ESVIO_FE_STEREO_SPLIT
  int stereo_split_env = -1;
  bool stereo_split = false;
  int n_queue_conflicts = 0;  // pairs of the handle's streams found on one hardware queue (ESVIO_FE_QUEUE_PROBE)
  hipEvent_t ev_planes_free = nullptr;
  // a plain (not announced) call: the frame's images are built (main stream) -> its Arc* pass on the
  // prefetch stream and its stereo LK on stream4 start beside the temporal LK; the Arc* pass is done
  hipEvent_t ev_imgs_ready = nullptr, ev_arc_side = nullptr;
  // ... split by camera: the left camera's update + image on the main stream (ev_imgs_ready: the LEFT image
  // then), the right camera's behind it on the stereo stream — behind ev_sae_left (the one partition
  // scratch) and, for a batch in pageable memory, behind the right array's own DMA; ev_right_ready: done
  hipEvent_t ev_sae_left = nullptr, ev_right_ready = nullptr;
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
  uint32_t* d_warp = nullptr;  // [part_cap] motion compensation: the pixel each event is warped to
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
  float *x_send = nullptr, *x_recv = nullptr, *x_pin = nullptr, *x_pin_recv = nullptr;
  size_t x_recv_cap = 0;
  // esvio_fe_comm_init: the handle's own RCCL communicator, the side stream the asynchronous
  // exchange runs on and the event that marks its end
  void* x_comm = nullptr;
  int x_world = 0;
  hipStream_t x_stream = nullptr;
  hipEvent_t x_done = nullptr;
  bool x_pending = false;
  bool x_auto = false;      // esvio_fe_set_auto_exchange: every published frame's records are exchanged
  bool x_deferred = false;  // packed records waiting to be enqueued (by the next call, under its device wait)
  // ---- next-batch prefetch (esvio_fe_set_next_batch)
  std::deque<Batch> announced;     // announced, nothing enqueued yet (<= kPrefetchDepth)
  std::deque<Inflight> inflight;   // SAE update / images / pyramids enqueued on the prefetch stream
  bool cur_prefetched = false;     // the frame being processed came from the prefetch stream
  EventStager* stager = nullptr;  // host-resident batches: pinned chunks + DMA by helper threads
  // chunks queued in the stager right now; lives here (not in the stager) because the RANSAC helpers poll
  // it while they spin, and the handle outlives both
  std::atomic<int> stage_pending{0};
  int cur_stage = -1;             // staging slot of the batch being tracked
  int stage_threads = 0;          // helper threads of the stager (0: off)
  // esvio_fe_set_launch_thread: the HIP calls of an announced batch's prefetch sequence (~10 launches and
  // event calls, 35-45 us of host time per batch) are issued by a thread of the handle; the calling
  // thread only does the bookkeeping and, before it consumes a lane, checks that its job has been issued
  Launcher* launcher = nullptr;
  uint64_t lane_job[kPrefetchDepth] = {};  // job number that (re)records the lane's events
  int launch_err = 0;  // a launch-thread job failed and the thread has been stopped since: sticky until esvio_fe_reset
  uint64_t lks_wait_job[2] = {};  // latest job whose prefetch sequence waits on ev_lks_done[set] (record_lks_done)
  // bounds of the device-side waits handed to the launches; esvio_fe_debug_inject / ESVIO_FE_FAULT set
  // chosen ones to 0 (the wait expires the first time it would have to wait)
  struct WaitLimits {
    uint32_t lookback = kSpinLookback, ticket = kSpinTicket;
    unsigned long long poll = kTicksPoll, chain = kTicksChain;
  } lim;
  bool lazy_late = false;  // ESVIO_FE_FAULT_LAZY_LATE (test): the lazy completions always at their latest point
  uint64_t n_spec_expired = 0, n_chain_expired = 0;  // speculative / chained temporal LK launches redone
  EventRec* d_evp[kPrefetchDepth] = {};  // host-event staging, one per prefetch lane (stager off)
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
  // one word per prefetch lane: the serial number of the last prefetch sequence that has RUN there (LkArgs::gate_*)
  uint32_t* d_lane_gate = nullptr;
  uint32_t gate_seq = 0;
  // LK launches whose waves WAIT on the device for another kernel's results (the speculative temporal launch for
  // k_select's corners, a chained launch for its producer's points) hold their CU while they wait — in the float-order
  // mode one workgroup of four points per CU.  If the waiting workgroups can fill the device, the kernel they wait for
  // may find no CU: with max_cnt 1000 (250 workgroups per launch on 256 CUs) one step in ~20 ran into the wait's 20-40 ms
  // bound and was redone (correct, but 40 ms).  So: a speculative launch only if one launch's workgroups leave two CUs
  // free (k_select_mw's up to 160 KB of LDS need a CU without an LK workgroup), a chained one (two waiting launches at
  // once) only if two do.  (max_cnt <= 1016 / <= 508 on MI355X's 256 CUs.)
  int n_cu = 0;
  bool waits_fit_spec = true, waits_fit_chain = true;
  uint32_t chain_seq = 0;
  bool chain_enabled = true;   // (ESVIO_FE_NO_CHAIN=1 turns it off: A/B measurements)
  bool cam_split_enabled = true;  // a plain call runs the two cameras' updates on two streams (ESVIO_FE_NO_CAMSPLIT=1: one)
  uint64_t n_plain_calls = 0, n_cam_split = 0, n_stereo_chained = 0;  // esvio_fe_plain_call_counters
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
  // arc / select
  uint8_t* d_flags = nullptr;
  // per-block ordered candidate lists written by k_arc; two sets so that the Arc* of a prefetched
  // batch (prefetch stream) never overwrites the set the current frame's selection still reads
  struct CandSet {
    uint32_t *xy = nullptr, *idx = nullptr, *cnt = nullptr;
    // ... and their ordered compaction into one stream (k_compact, launched right behind k_arc)
    uint32_t *comp_xy = nullptr, *comp_idx = nullptr, *total = nullptr, *grp = nullptr;
    size_t cap = 0;
  } cand[kRightSlots];
  // per-pixel earliest candidate of a set's latest Arc* pass (ArcArgs::first_map / launch_dedup)
  uint32_t* d_first[kRightSlots] = {};
  // per-pixel, per-polarity result of the event-independent part of isCorner (k_arc_map), one map
  // per candidate set
  uint32_t* d_cmap[kRightSlots] = {};
  uint8_t* d_touched[kRightSlots] = {};  // (pixel, polarity) pairs a batch's left events hit
  uint32_t first_epoch[kRightSlots] = {};  // Arc* passes into the set so far
  bool dedup_enabled = true;               // (ESVIO_FE_NO_DEDUP=1, test-only: the path batches >= 2^20 events take)
  bool fuse_ts_pyr = true;                 // (ESVIO_FE_NO_FUSE=1, test-only: k_time_surface + 3 x k_pyr_down, the median / equalize path)
  int cand_cur = 0;
  size_t arc_cap = 0;
  uint32_t* d_mask_bits = nullptr;
  // goodFeaturesToTrack scratch (image front-end), allocated on first use
  float4 *d_gftt_cov = nullptr, *d_gftt_rowsum = nullptr;
  float* d_gftt_eig = nullptr;
  uint32_t* d_gftt_max = nullptr;
  int32_t* d_sel_idx = nullptr;
  // pinned host staging (layout: pin_of())
  uint8_t* h_img = nullptr;  // copy_level0_in's staging ring: pinned host side ...
  uint8_t* d_img = nullptr;  // ... and its device side (linear images)
  size_t img_stage_bytes = 0;
  unsigned img_stage_next = 0;
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
  uint8_t* d_eq_tmp = nullptr;  // equalize: the two CLAHE outputs before normalisation (linear W x H each)
  bool select_ok = true;  // the greedy selection's bitmap fits LDS
  bool select_one_wave = false;  // (ESVIO_FE_SELECT_SERIAL=1, test-only: the one-wave selection kernel)
  uint32_t* d_sel_bitmap = nullptr;  // ... else it lives here (k_select_gbm)
  bool trace = false;
  double phase_ms[2][8] = {};  // [published?][phase]
  double pub_ms[6] = {};       // published frames: the parts of "host mask + enqueue detect/stereo"
  double tail_ms[2][4] = {};   // [published?]: left bookkeeping, previous frames' right tails, this frame's right tail, the rest
  uint64_t phase_count[2] = {0, 0};
  uint64_t phase_frames = 0, tr_cand = 0, tr_new = 0, tr_detect = 0, tr_surv = 0;
  uint64_t tr_fm_class[3] = {};  // rejectWithF_event calls with < 8 points (skipped), 8..14 (LMedS), >= 15 (RANSAC)
  double tr_fm_ms = 0;  // time inside find_fundamental_mat alone
  double tr_fm_max_ms = 0, tr_lift_ms = 0;  // ... its slowest call; the two liftProjective batches
  uint64_t tr_chain_launch = 0, tr_chain_used = 0, tr_chain_cancel = 0, tr_spec_used = 0;
  // (trace only) device-side intervals of the published frame's chain, from timing events
  hipEvent_t ev_dbg_sel_start = nullptr;
  double tr_gpu_sel = 0, tr_gpu_spec = 0, tr_gpu_chain = 0, tr_host_chain = 0, tr_gpu_pyr = 0;
  int tr_lane = -1;  // prefetch lane of the frame being tracked
  uint64_t tr_gpu_n = 0;
  std::chrono::steady_clock::time_point tr_sel_launch;

  // ---- per-call latency record (esvio_fe_latency_stats; always on)
  struct Latency {
    static constexpr int kRing = 4096;
    float ring[kRing] = {};
    uint64_t calls = 0;
    double sum_ms = 0, max_ms = 0;
    uint64_t max_call = 0;
    int max_pub = 0, max_cpu0 = -1, max_cpu1 = -1;
    long max_nivcsw = 0, max_allocs = 0;
    double max_phase[ESVIO_FE_LATENCY_PHASES] = {};
    uint64_t allocs = 0, nivcsw = 0;
    double cur_phase[ESVIO_FE_LATENCY_PHASES] = {};  // the call in progress
    // the latest calls, whole (esvio_fe_latency_recent)
    static constexpr int kRecent = 256;
    esvio_fe_latency_call recent[kRecent] = {};
    uint64_t total_calls = 0;
    bool have_t0 = false;
    std::chrono::steady_clock::time_point t0;
  } lat;
  uint64_t n_allocs = 0;  // hipMalloc / hipHostMalloc calls of this handle so far
  double slow_call_ms = 0;  // ESVIO_FE_SLOW_CALL_MS: calls slower than this are reported on stderr (0: off)

  // ---- profiling
  bool prof_on = false;
  KStat stats[K_COUNT];
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> ev_pool;
};

namespace esvio {
namespace fe {

// The stream the helpers enqueue on: the main stream unless the calling thread has switched to
// another one (prefetch -> stream2, speculative LK -> stream3).
inline thread_local hipStream_t t_stream_override = nullptr;
// the stereo stream of the frame being tracked
inline hipStream_t stereo_stream(const esvio_fe_ctx* c) { return c->stereo_unpub ? c->stream6 : c->stream4; }
inline hipStream_t cur_stream(const esvio_fe_ctx* c) {
  return t_stream_override ? t_stream_override : c->stream;
}
struct StreamScope {
  hipStream_t saved;
  explicit StreamScope(hipStream_t s) : saved(t_stream_override) { t_stream_override = s; }
  ~StreamScope() { t_stream_override = saved; }
};

// (a thread of the handle that is not the calling thread — the launch thread — gives its error texts a place of its
// own here: c->err belongs to the calling thread)
inline std::string*& fail_sink() {
  static thread_local std::string* sink = nullptr;
  return sink;
}
inline int fail(esvio_fe_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (std::string* sink = fail_sink()) *sink = buf;
    else c->err = buf;
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
inline hipEvent_t get_event(esvio_fe_ctx* c) {
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

inline void resolve_profile(esvio_fe_ctx* c) {  // main stream idle; prefetch-stream records may be pending
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

}  // namespace fe
}  // namespace esvio
