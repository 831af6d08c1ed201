// fe_kernels.h — internal launch API of the gfx950 kernels (fe_kernels.hip).
// Not part of the public boundary; the C ABI is include/esvio_fe.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

namespace esvio {


// dvs_msgs::Event, 16 B AoS (reference: feature_tracker/src/dvs_msgs/Event.h:42-52)
struct __attribute__((aligned(16))) EventRec {
  uint16_t x, y;
  uint32_t sec, nsec;
  uint8_t pol;
  uint8_t pad[3];
};
static_assert(sizeof(EventRec) == 16, "event record must be 16 B");

constexpr int kPad = 24;        // image border kept around every pyramid level (>= LK win 21)
constexpr int kMaxLevels = 4;   // LK maxLevel 3 -> 4 levels
constexpr int kLkWin = 21;      // cv::Size(21,21) at every call site (feature_tracker.cpp:410..495)
constexpr int kArcBlock = 256;  // events per block of the Arc* kernel
constexpr int kMaxDiscR = 63;   // max min_dist supported by the selection kernel

// One image pyramid resident in HBM.  Level l is stored padded by kPad on every side
// (BORDER_REFLECT_101 for the image, zeros for the Scharr derivatives), row stride
// w[l]+2*kPad; pointers address the padded buffer's origin.
struct PyrDesc {
  uint8_t* img[kMaxLevels];
  int16_t* deriv[kMaxLevels];  // interleaved (Ix,Iy), same stride (in pixels) as the image
  int w[kMaxLevels], h[kMaxLevels];
  int stride[kMaxLevels];  // row stride in pixels: (w + 2*kPad) rounded up to 16
  int levels;  // maxLevel (inclusive) actually built
};
inline int pyr_stride(int w) { return (w + 2 * kPad + 15) & ~15; }

// Slots of the per-kernel timers (esvio_fe_set_profiling): one per kernel FUNCTION, named as rocprofv3 names it
// (kKernelNames, fe_ctx.h), so that a bench line's `kernels` can be matched with a profile without a table.  A
// launch site that picks between forms (k_lk / k_lk_f32, k_time_surface4 / k_time_surface, the selection kernels)
// books the one it launched.  The few slots that cover more than one function say so in their name's comment.
enum KernelId {
  K_TILE_HIST = 0,   // (+ k_mc_warp in front of it for a motion-compensated batch)
  K_TILE_SCAN,
  K_TILE_SCATTER,
  K_TILE_APPLY,
  K_SAE_KEYS,        // the radix-sort form of the update: k_sae_keys, k_radix_pass, k_sae_apply (_ev, _ev_write)
  K_RADIX_PASS,
  K_SAE_APPLY,
  K_TIME_SURFACE4,
  K_TIME_SURFACE,
  K_MEDIAN,
  K_CLAHE,           // k_clahe_lut, k_clahe_interp (k_normalize in the unfused form)
  K_NORM_PYR,
  K_PYR3,
  K_PYR_DOWN,
  K_PYR_PAD,
  K_SCHARR,
  K_PAD_SCHARR,
  K_LK_F32,
  K_LK,
  K_ARC_MAP,         // (+ k_arc_mark where the SAE update has not left the touched flags)
  K_ARC_EV,
  K_DEDUP,
  K_COMPACT,
  K_SELECT_MW,
  K_SELECT,
  K_SELECT_GBM,
  K_COUNT
};

// ---- SAE update -------------------------------------------------------------------------
// Bounds of the device-side waits (every one gives up instead of hanging the GPU; the host then fails
// the call or redoes the launch).  They are launch parameters so that a test can make a wait expire on
// demand (esvio_fe_debug_inject, fe_ctx.h: WaitLimits).
constexpr uint32_t kSpinLookback = 1u << 20;  // k_radix_pass: polls of an earlier tile's look-back word
constexpr uint32_t kSpinTicket = 1u << 21;    // k_tile_apply: polls of the block's turn ticket
constexpr unsigned long long kTicksPoll = 2000000ull;   // k_lk waiting for k_select's corner: 20 ms
constexpr unsigned long long kTicksChain = 4000000ull;  // k_lk waiting for the previous frame's k_lk: 40 ms

constexpr int kRadixTile = 2048;  // keys per block
constexpr int kRadixMaxBits = 8;
constexpr int kRadixMaxPasses = 4;
inline uint32_t radix_blocks(uint32_t n) { return (n + kRadixTile - 1) / kRadixTile; }
// sort scratch layout (uint32 words): [ghist: passes<<bits][tickets: passes] cleared by k_sae_apply,
// then [lookback: passes * radix_blocks(n) << bits] cleared by k_sae_keys
// keys[i] = cam*P + y*W + x (or invalid_key for out-of-sensor events), vals[i] = i, for the
// virtual concatenation [left; right]; also fills the per-pass global digit histograms.
void launch_sae_keys(hipStream_t s, const EventRec* evL, uint32_t nL, const EventRec* evR,
                     uint32_t nR, int W, int H, uint32_t* keys, uint32_t* vals,
                     uint32_t invalid_key, unsigned long long* n_rejected, int passes, int bits,
                     uint32_t* ghist, uint32_t* lookback, uint32_t lookback_words,
                     const struct McParams* mc /* NULL: no motion compensation */);
// one stable LSD pass on digit (key >> shift) & ((1<<bits)-1) with decoupled look-back
void launch_radix_pass(hipStream_t s, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t n,
                       int shift, int bits, const uint32_t* ghist, uint32_t* lookback,
                       uint32_t* ticket, uint32_t* keys_out, uint32_t* vals_out, int* err,
                       uint32_t spin_limit = kSpinLookback);
// walk every same-pixel segment of the sorted keys in stream order applying the SAE rule
// (event_detector.cc:149-166). L2/S2: double2 per (cam,pixel): {L[0],L[1]} and {S[0],S[1]}.
void launch_sae_apply(hipStream_t s, const uint32_t* keys, const uint32_t* vals, uint32_t n,
                      const EventRec* evL, uint32_t nL, const EventRec* evR, double2* L2,
                      double2* S2, double filter_threshold, uint32_t invalid_key,
                      uint32_t* sort_scratch, uint32_t sort_scratch_words);
// the same update with one lane per EVENT instead of one per pixel (see k_sae_apply_ev): for batches
// with many events per pixel, where the per-pixel walk leaves most lanes idle
void launch_sae_apply_ev(hipStream_t s, const uint32_t* keys, const uint32_t* vals, uint32_t n,
                         const EventRec* evL, uint32_t nL, const EventRec* evR, double2* L2,
                         double2* S2, double filter_threshold, uint32_t invalid_key,
                         uint32_t* sort_scratch, uint32_t sort_scratch_words, uint8_t* marks /* [n] */);

// ---- SAE update, tiled form (default) --------------------------------------------------------
// The sensor is cut into tiles of tw x th pixels; a tile of one camera is a bucket.  One stable
// partition of the event records by bucket (k_tile_hist + k_tile_scan + k_tile_scatter: one digit of
// <= 11 bits, no atomics or spins across blocks; the partitioned records are 8 bytes — tile-local
// pixel, polarity, seconds relative to the batch's smallest, nsec — whenever the batch's stamps allow
// it, else the raw 16), then one block per bucket applies its events in stream order with the tile's
// L planes in LDS (k_tile_apply).  Replaces key generation + 3 radix passes over (key, index) pairs +
// a gathering apply: per event 16 B are read twice, 8 written and 8 read.
// points (= waves) of an LK launch per workgroup; in the float-order mode a workgroup's 122 KB of LDS make it the only
// one on its CU (fe_kernels.hip kLkWaves; what the host needs it for: esvio_fe_ctx::waits_fit_*)
#ifndef ESVIO_LK_WAVES
#define ESVIO_LK_WAVES 4
#endif
constexpr int kLkPointsPerBlock = ESVIO_LK_WAVES;

struct TileGeom {
  int W, H;
  int tw, th;            // tile size in pixels (tw * th <= kTileMaxPx)
  int tiles_x, tiles_y;  // tiles per camera
  int nt_cam;            // tiles_x * tiles_y
  int nbins;             // 2 * nt_cam + 1 (last bin: out-of-sensor events)
  int bits;              // ceil(log2(nbins))
  int pix_bits;          // ceil(log2(tw * th))
};
constexpr int kTileMaxBins = 2048;
constexpr int kTileMaxPx = 2048;
constexpr int kTileScatterThreads = 256;
// smallest tile whose bucket count fits one digit; false: sensor too large for the tiled form
// (host-only arithmetic: inline here, so that a build without the kernels — tests/hipstub — has it too)
inline bool make_tile_geom(int W, int H, TileGeom* g) {
  // the smallest tile whose bucket count fits (a larger one was measured at C5: 64x32 takes 10 % off the
  // scatter and adds 50 % to the apply)
  static const int cand[][2] = {{32, 16}, {32, 32}, {64, 32}};
  for (const auto& c : cand) {
    const int tx = (W + c[0] - 1) / c[0], ty = (H + c[1] - 1) / c[1];
    const int nb = 2 * tx * ty + 1;
    if (nb > kTileMaxBins || c[0] * c[1] > kTileMaxPx) continue;
    g->W = W;
    g->H = H;
    g->tw = c[0];
    g->th = c[1];
    g->tiles_x = tx;
    g->tiles_y = ty;
    g->nt_cam = tx * ty;
    g->nbins = nb;
    g->bits = 1;
    while ((1 << g->bits) < nb) g->bits++;
    g->pix_bits = 1;
    while ((1 << g->pix_bits) < c[0] * c[1]) g->pix_bits++;
    return true;
  }
  return false;
}

// events per scatter block.  (Until round 6: 4096 from 2^20 events on, in 16 rounds per wave — 212 VGPRs, two resident
// blocks per CU; with one camera's buckets per table (below) the 8-round kernel's 33 KiB and 128 VGPRs let three run,
// and at 6.7 M events 2048 measured 59-66 us against 70 for 4096, profiles/r06_scatter_grid_and_nt_store_sweep.txt.)
constexpr uint32_t kTileScatterEvents = 2048;
// The batch is [left array; right array], and every bucket belongs to one camera: a scatter block never mixes the two.
// Scatter block b < nblkL owns the left events [b*TE, (b+1)*TE), block nblkL + b the right events [b*TE, (b+1)*TE) —
// so a block's bucket table (LDS in k_tile_scatter, a row of the count matrices) has one camera's nt_cam buckets + the
// out-of-sensor bin: half the columns the whole batch has (round 6; before, blocks were cut from the concatenated
// stream and every table had both cameras' buckets).  k_tile_hist block `seg` counts the buckets of `group`
// consecutive scatter blocks of ONE camera, so that the count matrices stay small: at most kTileMaxGroups groups.
constexpr uint32_t kTileMaxGroups = 512;
struct TileSplit {
  uint32_t te;            // events per scatter block
  uint32_t nblkL, nblkR;  // scatter blocks per camera
  uint32_t group;         // scatter blocks per k_tile_hist block
  uint32_t nsegL, nsegR;  // k_tile_hist blocks (= groups) per camera
  __host__ __device__ uint32_t nblk() const { return nblkL + nblkR; }
  __host__ __device__ uint32_t nseg() const { return nsegL + nsegR; }
};
inline TileSplit tile_split(uint32_t nL, uint32_t nR) {
  TileSplit t;
  t.te = kTileScatterEvents;
  t.nblkL = (nL + t.te - 1) / t.te;
  t.nblkR = (nR + t.te - 1) / t.te;
  t.group = (t.nblkL + t.nblkR + kTileMaxGroups - 3) / (kTileMaxGroups - 2);  // (each camera's last group may be short)
  if (!t.group) t.group = 1;
  t.nsegL = (t.nblkL + t.group - 1) / t.group;
  t.nsegR = (t.nblkR + t.group - 1) / t.group;
  return t;
}
// scratch (uint32 words).  Every k_tile_hist block leaves the range of seconds and the OR of the nsec words
// of its events in its own slot of `ranges` (three same-address global atomics per block — 410 blocks
// ending together — were 6 of the kernel's 32 us at 6.7 M events); k_tile_scan reduces the slots and
// writes the record format of the partition into `meta`
enum { kTileMetaCompact = 0, kTileMetaSecBase, kTileMetaWords = 8 };
constexpr int kTileRecSecBits = 20;
struct TileScratch {
  uint32_t* meta;      // [kTileMetaWords], written by k_tile_scan
  uint32_t* ranges;    // [kTileMaxGroups][4] {sec min, sec max, nsec OR, -} per k_tile_hist block
  uint32_t* totals;    // [nbins] events per bucket
  uint32_t* tile_off;  // [nbins + 1] exclusive bucket offsets into `part`, written by k_tile_scatter
  uint32_t* tile_order;  // [nbins - 1] buckets by descending size class, written by k_tile_scatter
  // the count matrices: one row per scatter block / group, one column per bucket of the row's camera (nt_cam + 1: the
  // last one is the out-of-sensor bin)
  uint32_t* P;         // [nblk][nt_cam + 1] bucket counts of the earlier scatter blocks of the same group
  uint32_t* T;         // [ngroups][nt_cam + 1] bucket counts per group
  uint32_t* C;         // [ngroups][nt_cam + 1] ... of all earlier groups of the camera (out-of-sensor bin: of all earlier groups)
};
// k_tile_hist: P, T and the blocks' ranges; launch_tile_scan: C, totals, the record format; out-of-sensor
// events added to *n_rejected
// mc (optional, enabled): the motion-compensated overload — buckets by the warped pixel, which is kept
// in warp_xy[nL + nR] for launch_tile_scatter
void launch_tile_hist(hipStream_t s, const EventRec* evL, uint32_t nL, const EventRec* evR, uint32_t nR,
                      const TileGeom& g, const TileScratch& sc, unsigned long long* n_rejected,
                      const struct McParams* mc = nullptr, uint32_t* warp_xy = nullptr);
void launch_tile_scan(hipStream_t s, uint32_t nL, uint32_t nR, const TileGeom& g, const TileScratch& sc,
                      unsigned long long* n_rejected);
// stable partition of [left; right] into `part` by bucket (warp_xy non-null: the records get the
// warped pixels launch_tile_hist computed)
void launch_tile_scatter(hipStream_t s, const EventRec* evL, uint32_t nL, const EventRec* evR, uint32_t nR,
                         const TileGeom& g, const TileScratch& sc, EventRec* part, const uint32_t* warp_xy = nullptr);
// createSAE_left/right (event_detector.cc:149-166, :212-228) per bucket, events in stream order.
// arc_touched (optional): ArcArgs::touched of the Arc* pass this batch will get — the left camera's
// touched (pixel, polarity) flags are written here, from the tiles' own bookkeeping, instead of by
// launch_arc_mark
void launch_tile_apply(hipStream_t s, const EventRec* part, uint32_t n, const TileGeom& g, const TileScratch& sc,
                       double2* L2, double2* S2, double filter_threshold, uint8_t* arc_touched, int* err,
                       uint32_t spin_limit = kSpinTicket);

// ---- time-slice composition (one stream cut into N slices, one per GPU) -------------------
constexpr double kSliceNone = -1.0;  // "this slice wrote nothing here" (event times are >= 0)
void launch_fill_f64(hipStream_t s, double* p, size_t n, double v);
void launch_spin(hipStream_t s, unsigned long long ticks);  // (100 MHz ticks)
void launch_set_u32(hipStream_t s, uint32_t* p, uint32_t v);  // *p = v, visible device-wide, behind everything enqueued on s so far
// dst[i] = src[i] unless src[i] == none
void launch_overlay_f64(hipStream_t s, double* dst, const double* src, size_t n, double none);

// ---- time surface -----------------------------------------------------------------------
// renders ncam cameras (S2 + cam*P) into level-0 interiors of dst[cam]
KernelId launch_time_surface(hipStream_t s, const double2* S2, int W, int H, double t_sync,
                         double decay_sec, int ignore_polarity, uint8_t* dst0, uint8_t* dst1,
                         int dst_stride, int ncam);

// ---- cv::medianBlur(ksize = 2k+1) of the rendered time surface (event_detector.cc:262-264) ----
// src/dst: pixel (0,0) pointers of nimg images with the given strides; BORDER_REPLICATE; k <= 7
constexpr int kMaxMedianK = 7;
void launch_median(hipStream_t s, const uint8_t* src0, const uint8_t* src1, int src_stride,
                   uint8_t* dst0, uint8_t* dst1, int dst_stride, int W, int H, int k, int nimg);

// ---- CLAHE + normalize (equalize: 1) ---------------------------------------------------------
// stage 0: per-tile LUTs, stage 1: LUT blending + min/max, stage 2: MINMAX normalisation in place
void launch_norm_pyr(hipStream_t s, const uint8_t* src0, const uint8_t* src1, int src_stride, const int* minmax,
                     const PyrDesc* p);
void launch_clahe(hipStream_t s, const uint8_t* raw0, const uint8_t* raw1, int raw_stride,
                  uint8_t* dst0, uint8_t* dst1, int dst_stride, int W, int H, uint8_t* lut,
                  int* minmax, int nimg, int stage);

// ---- pyramid ----------------------------------------------------------------------------
void launch_pyr_down(hipStream_t s, const PyrDesc* p, int nimg, int src_level);
// levels 1..3 of nimg images whose level 0 is in place, one launch (maxLevel 3)
void launch_pyr3(hipStream_t s, const PyrDesc* p, int nimg);
void launch_pyr_pad(hipStream_t s, const PyrDesc* p, int nimg);
void launch_pad_scharr(hipStream_t s, const PyrDesc* p, int nimg);  // k_pyr_pad + k_scharr, one launch
void launch_scharr(hipStream_t s, const PyrDesc* p, int nimg);

// ---- LK ---------------------------------------------------------------------------------
struct LkArgs {
  PyrDesc P;  // prev pyramid (+ derivatives)
  PyrDesc N;  // next pyramid (images only)
  const float2* prev_pts;
  const float2* init_pts;  // initial nextPts when flags has USE_INITIAL_FLOW (may alias next_pts)
  float2* next_pts;        // out
  uint8_t* status;
  const int* n_ptr;  // device count (may be NULL -> n_max)
  int n_max;
  // optional: points with index >= poll_from are not in prev_pts yet; their wave waits (bounded)
  // for k_select to publish them (SelectArgs::pub_*) or to announce a total below their index
  const unsigned long long* poll_slots = nullptr;
  const unsigned long long* poll_done = nullptr;
  uint32_t poll_seq = 0;
  int poll_from = 0;
  int* poll_err = nullptr;  // set to 1 if a wait expired (host-visible)
  // optional: two temporal launches chained point by point (frames g and g+1 when g publishes
  // nothing, so g+1 tracks exactly g's forward results).  The producer (chain_out) publishes each
  // point's forward result the moment it has it — two self-validating words per point,
  // (chain_seq << 2 | alive) << 32 | float bits of x resp. y; the consumer's wave of the same index
  // (chain_in) waits (bounded, poll_err) for them instead of reading prev_pts, and leaves
  // status 0 for a point whose forward status was 0 or that does not exist.
  unsigned long long* chain_out = nullptr;
  const unsigned long long* chain_in = nullptr;
  uint32_t chain_seq = 0;  // 30 bits, never 0
  unsigned long long poll_ticks = kTicksPoll, chain_ticks = kTicksChain;  // bounds of the two waits
  // gate: the kernel's waves go on only when *gate_ptr has reached gate_val (serial-number compare) — the `next`
  // pyramid of a chained launch is built by a prefetch sequence another thread is still issuing, so a stream wait
  // on that sequence's event cannot be enqueued yet; the sequence's last launch writes the word (launch_set_u32).
  // Bounded by chain_ticks, failure in *poll_err like the other waits.
  const uint32_t* gate_ptr = nullptr;
  uint32_t gate_val = 0;
  int max_level;
  int max_count;
  double eps2;
  int flags;
  int accum = 1;  // 1: exact integer sums (k_lk); 2: float sums in the reference build's order (k_lk_f32)
};
// one launch runs call `f`; if `b` != NULL the same wave then runs call `b` with prevPts = f's
// result and initial flow = f's prevPts (the forward/backward check of feature_tracker.cpp:410-418
// and :490-495); b->prev_pts/init_pts/next_pts/status are ignored.
void launch_lk(hipStream_t s, const LkArgs& f, const LkArgs* b, float2* back_pts,
               uint8_t* back_status);

// ---- Arc* + selection -------------------------------------------------------------------
struct ArcArgs {
  const EventRec* ev;
  uint32_t n;
  const double2* L2;  // left planes
  const double2* S2;
  int W, H;
  double filter_threshold;
  int border;                  // MIN_DIST + 1
  const uint8_t* ts;           // left level-0 padded image origin, or NULL (skip TS test)
  int ts_stride;
  double ts_lk_threshold;
  const uint32_t* mask_bits;   // H * wpr words, bit set = blocked; or NULL
  int wpr;
  uint8_t* flags;              // [n] or NULL
  uint32_t* cand_xy;           // [nblk*kArcBlock] per-block ordered candidates (x | y<<16), or NULL
  uint32_t* cand_idx;          // [nblk*kArcBlock]
  uint32_t* cand_cnt;          // [nblk]
  // per-pixel earliest candidate of this launch (optional): every candidate does
  // atomicMin(first_map[pixel], first_key | event index); keys of later launches are smaller
  // (first_key's top byte counts down), so the map is only cleared when that byte wraps
  uint32_t* first_map;
  uint32_t first_key;
  // touched: one flag byte per (pixel, polarity), index 2*pixel + polarity (arc_flag_bytes()), set by
  //          launch_arc_mark for every pair the batch has an event of, consumed (and cleared) by
  //          launch_arc_map; must start out zeroed
  // cmap:    bitmap, bit 2*pixel + polarity (arc_bitmap_words() words): result of the
  //          event-independent part of isCorner for the touched pairs, written by launch_arc_map
  //          from the planes as they are, read by launch_arc
  uint8_t* touched;
  uint32_t* cmap;
};
inline size_t arc_bitmap_words(int W, int H) { return ((size_t)2 * W * H + 31) / 32 + 64; }
inline size_t arc_flag_bytes(int W, int H) { return (size_t)2 * W * H + 1024; }
void launch_stage_pull(hipStream_t s, const void* pinned_src, void* dst, size_t bytes);
// ... of chunks of `epc` events each, packed to 8 bytes per event or raw as desc[chunk] = {base second, packed?} says
void launch_stage_pull_packed(hipStream_t s, const void* pinned_src, void* dst, size_t bytes, const void* desc, uint32_t epc);  // H2D of staged events by a kernel
void launch_arc_mark(hipStream_t s, const ArcArgs& a);  // per event: which (pixel, polarity) pairs occur
void launch_arc_map(hipStream_t s, const ArcArgs& a);   // per touched pair (rings, L[!p] > L[p], TS, border)
void launch_arc(hipStream_t s, const ArcArgs& a);       // per event, in stream order (needs the map)

// Only the earliest candidate of a pixel can ever be accepted by the greedy selection
// (feature_tracker.cpp:13-38: if it is accepted its disc blocks the pixel, if it is refused the
// pixel was blocked already, and blocked pixels stay blocked), so the later ones (more than half
// of the Arc* corners of a 5 Mev/s stream) are dropped from the per-block lists, in place and in
// order, before the ordered compaction; the sequential stage then has less than half the work.
void launch_dedup(hipStream_t s, uint32_t* cand_xy, uint32_t* cand_idx, uint32_t* cand_cnt,
                  uint32_t nblk, const uint32_t* first_map, uint32_t first_key, int W);

// ordered compaction of the per-block candidate lists (parallel; one block per Arc* block)
// grp_scratch (optional, [nblk / 64 + 1]): with more than 2048 blocks the counts are summed per 64
// blocks first (k_compact_groups)
void launch_compact(hipStream_t s, const uint32_t* cand_xy, const uint32_t* cand_idx,
                    const uint32_t* cand_cnt, uint32_t nblk, uint32_t* comp_xy, uint32_t* comp_idx,
                    uint32_t* total, uint32_t* grp_scratch = nullptr);

// ---- goodFeaturesToTrack (image front-end) ----------------------------------
// cv::goodFeaturesToTrack(img, n, quality, minDistance, mask) with blockSize 3 / gradientSize 3 /
// Shi-Tomasi, as FeatureTracker::trackImage calls it (feature_tracker.cpp:228) [OpenCV, restated]:
//   k_gftt_cov      Sobel (float, scale 1/3060) -> (Dx*Dx, Dx*Dy, Dy*Dy) per pixel
//   k_gftt_rowsum   box filter rows: (c[x-1] + c[x]) + c[x+1]
//   k_gftt_eig      box filter columns as OpenCV's running sum (one thread per column, rows in
//                   order: float addition order matters), min eigenvalue, masked maximum
//   k_gftt_collect  threshold at quality*max, 3x3 local maximum, mask -> per-block ordered lists
//   (k_compact), k_gftt_sortprep + 4 x k_radix_pass: by value then address, both descending
//   k_select with the open Euclidean disc: the minDistance greedy
struct GfttArgs {
  const uint8_t* img;  // level-0 pixel (0,0) inside its reflect-101 padded buffer
  int stride;
  int W, H;
  float4* cov;         // [W*H] scratch
  float4* rowsum;      // [W*H] scratch
  float* eig;          // [W*H] cornerMinEigenVal
  const uint32_t* mask_bits;  // blocked pixels (1 = not allowed), may be NULL
  int wpr;
  uint32_t* max_key;   // [1] order-preserving key of the masked maximum (cleared by k_gftt_cov)
  double quality;
  uint32_t* cand_xy;   // [nblk*kArcBlock] per-block ordered candidates: x | y<<16 ...
  uint32_t* cand_val;  // ... and the response bits
  uint32_t* cand_cnt;  // [nblk]
};
void launch_gftt_response(hipStream_t s, const GfttArgs& a);  // cov, rowsum, eig
void launch_gftt_collect(hipStream_t s, const GfttArgs& a);
// keys = ~value bits, vals = xy, read back to front (so that the stable sort leaves equal values
// in descending address order); accumulates the 4 x 8-bit digit histograms, clears the look-back
void launch_gftt_sortprep(hipStream_t s, const uint32_t* comp_xy, const uint32_t* comp_val, uint32_t n,
                          uint32_t* keys, uint32_t* vals, uint32_t* ghist, uint32_t* lookback,
                          uint32_t lookback_words);

struct SelectArgs {
  const uint32_t* comp_xy;   // compacted candidates in stream order
  const uint32_t* comp_idx;
  const uint32_t* total;     // number of candidates
  int W, H, wpr;
  int max_corners;
  int radius;
  int8_t hw[kMaxDiscR + 1];  // cv::circle half-widths per |dy|
  // >= 0: "|dx| <= hw[|dy|]" is the same set as "dx*dx + dy*dy <= disc_c" (true for cv::circle's
  // table at every radius 1..63 and for the open Euclidean disc; disc_threshold() checks it), so
  // disc membership is two multiply-adds instead of a table look-up; -1: use the table
  int disc_c;
  float2* out_pts;           // accepted corners are written at out_pts[out_base + k]
  int32_t* out_idx;          // may be NULL
  int out_base;
  int* n_out;                // number accepted
  int* n_total;              // out_base + accepted (feeds the LK kernels' n_ptr), may be NULL
  int* host_counts;          // optional host-mapped mirror: {accepted, out_base + accepted, total}
  // non-null: the disc bitmap lives here (H*wpr + 4 words of device memory) instead of in LDS — for
  // sensors whose bitmap does not fit the 160 KiB (k_select_gbm; LDS then holds the tables only)
  uint32_t* gbitmap;
  const uint32_t* init_bits; // optional H*wpr words the disc bitmap starts from (blocked pixels)
  // ... or, instead, the points whose discs ARE those blocked pixels (Event_setMask stamps
  // cv::circle(mask, cvRound(pt), MIN_DIST, 0, -1) for every point it keeps, feature_tracker.cpp
  // :77-86): the wave stamps them itself, so the host uploads n_stamp points, not a bitmap.
  // Needs n_stamp more LDS words behind the half-width table.
  const float2* stamp_pts;
  int n_stamp;
  // optional publication of each accepted corner the moment it is accepted, for an LK launch that
  // is already waiting (LkArgs::poll_*): slot[out_base + k] = seq<<32 | y<<16 | x, and at the end
  // *done = seq<<32 | (out_base + accepted).  Device memory, relaxed agent-scope atomics; the data
  // travels in the flag word itself, so no fence is needed.
  unsigned long long* pub_slots;
  unsigned long long* pub_done;
  uint32_t pub_seq;
  // != 0: the one-wave walk (k_select) even where the 16-wave kernel's queue fits LDS beside the bitmap
  // (sensors between ~1.2 and 1.3 M pixels take it for lack of LDS; ESVIO_FE_SELECT_SERIAL=1, test-only)
  int one_wave;
};
KernelId launch_select(hipStream_t s, const SelectArgs& a, size_t lds_bytes);  // (returns the form it launched)
// the threshold described at SelectArgs::disc_c for a half-width table, or -1 if there is none
inline int disc_threshold(const int8_t* hw, int radius) {
  long inside = -1, outside = (long)(radius + 1) * (radius + 1);
  for (int dy = 0; dy <= radius && dy <= kMaxDiscR; dy++) {
    const long h = hw[dy];
    if (h >= 0) inside = inside > h * h + (long)dy * dy ? inside : h * h + (long)dy * dy;
    const long o = (h + 1) * (h + 1) + (long)dy * dy;  // (h = -1: the row's centre pixel is outside)
    outside = outside < o ? outside : o;
  }
  return inside >= 0 && inside < outside ? (int)inside : -1;
}


}  // namespace esvio
