// fe_image.cpp — the image front-end: FeatureTracker::trackImage
// (feature_tracker.cpp:164-338) and cv::goodFeaturesToTrack on the kernels of fe_kernels.hip.
#include "fe_internal.h"

namespace esvio {
namespace fe {

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
                        c->vals[cur ^ 1], c->z_counts + 3, c->lim.lookback);
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
  sa.disc_c = disc_threshold(sa.hw, sa.radius);
  sa.out_pts = out_pts;
  sa.out_idx = nullptr;
  sa.out_base = out_base;
  sa.n_out = c->d_counts;
  sa.n_total = c->d_counts + 1;
  sa.host_counts = host_counts;
  sa.init_bits = nullptr;
  sa.gbitmap = nullptr;
  sa.pub_slots = nullptr;
  sa.pub_done = nullptr;
  sa.pub_seq = 0;
  sa.one_wave = c->select_one_wave ? 1 : 0;
  size_t lds = select_lds_bytes(c);
  if (!c->select_ok) {  // a frame camera's size: the min-distance bitmap goes to device memory
    if (!c->d_sel_bitmap)
      if (int rc = dev_alloc(c, &c->d_sel_bitmap, (size_t)c->H * sa.wpr + 4)) return rc;
    sa.gbitmap = c->d_sel_bitmap;
    lds = select_tables_lds_bytes(c);
  }
  ScopedKernel k(c, K_SELECT_MW, 0);
  k.id = launch_select(cur_stream(c), sa, lds);
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


}  // namespace fe
}  // namespace esvio
