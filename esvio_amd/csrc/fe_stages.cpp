// fe_stages.cpp — device memory and the per-stage launches of the event front-end, plus the host
// bookkeeping helpers the reference keeps on the CPU (feature_tracker.cpp:48-151,910-1045).
// Every data-parallel stage is a HIP kernel from fe_kernels.hip; there is no CPU fallback.
#include "fe_internal.h"

namespace esvio {
namespace fe {

// ---------------------------------------------------------------- memory

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

static int grow_cand_set(esvio_fe_ctx* c, int set, size_t cap) {
  esvio_fe_ctx::CandSet& s = c->cand[set];
  void* ptrs[] = {s.xy, s.idx, s.cnt, s.comp_xy, s.comp_idx, s.total, s.grp};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  s = esvio_fe_ctx::CandSet();
  if (int rc = dev_alloc(c, &s.xy, cap)) return rc;
  if (int rc = dev_alloc(c, &s.idx, cap)) return rc;
  if (int rc = dev_alloc(c, &s.cnt, cap / kArcBlock)) return rc;
  if (int rc = dev_alloc(c, &s.grp, cap / kArcBlock / 64 + 2)) return rc;
  if (int rc = dev_alloc(c, &s.comp_xy, cap)) return rc;
  if (int rc = dev_alloc(c, &s.comp_idx, cap)) return rc;
  if (int rc = dev_alloc(c, &s.total, 1)) return rc;
  s.cap = cap;
  return 0;
}

int ensure_cand_capacity(esvio_fe_ctx* c, int set, size_t n) {
  if (n <= c->cand[set].cap) return 0;
  size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
  cap = (cap + kArcBlock - 1) / kArcBlock * kArcBlock;
  if (int rc = grow_cand_set(c, set, cap)) return rc;
  // the sets that have never been used get the same size now: in replay mode the Arc* passes rotate
  // through them, and each one's first use would otherwise put seven hipMallocs into some later call
  // (a set that holds candidates is left alone: only its own Arc* pass may replace it)
  for (int k = 0; k < kRightSlots; k++)
    if (k != set && c->cand[k].cap == 0)
      if (int rc = grow_cand_set(c, k, cap)) return rc;
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
  c->n_allocs++;
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
// Motion_correction_value -> kernel parameters (the kernels take t_0, the first left event's time,
// feature_tracker.cpp:621, from the batch itself: no host copy of an event for a device batch)
McParams make_mc_params(const esvio_fe_motion* m) {
  McParams p;
  std::memset(&p, 0, sizeof(p));
  p.enabled = 1;
  p.t1 = m->t1;
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

// the partition's buffers for batches of up to n events (d_warp only once a motion-compensated batch comes)
int ensure_part_capacity(esvio_fe_ctx* c, size_t n, bool mc) {
  if (n > c->part_cap) {
    const size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
    if (c->d_part) (void)hipFree(c->d_part);
    if (c->d_warp) (void)hipFree(c->d_warp);
    c->d_part = nullptr;
    c->d_warp = nullptr;
    c->part_cap = 0;
    if (int rc = dev_alloc(c, &c->d_part, cap)) return rc;
    c->part_cap = cap;
  }
  if (mc && !c->d_warp)  // the motion-compensated overload: 4 B per event for the warped pixels
    if (int rc = dev_alloc(c, &c->d_warp, c->part_cap)) return rc;
  const size_t nblk_cap = (c->part_cap + 2047) / 2048 + 2;  // (2048 events per scatter block at least; each camera's last block may be short)
  const size_t head = (size_t)3 * kTileMaxBins + 64 + 4 * (size_t)kTileMaxGroups;
  const size_t need = head + (nblk_cap + 2 * (size_t)kTileMaxGroups) * kTileMaxBins;
  if (need > c->tile_cap) {
    if (c->d_tile) (void)hipFree(c->d_tile);
    c->d_tile = nullptr;
    c->tile_cap = 0;
    if (int rc = dev_alloc(c, &c->d_tile, need)) return rc;
    c->tile_cap = need;
  }
  return 0;
}

int sae_update_tiled(esvio_fe_ctx* c, const EventRec* evL, uint32_t nL, const EventRec* evR, uint32_t nR,
                     double2* L2, double2* S2, uint8_t* arc_touched, const McParams* mc) {
  const uint32_t n = nL + nR;
  if (int rc = ensure_part_capacity(c, n, mc != nullptr)) return rc;
  const size_t nblk_cap = (c->part_cap + 2047) / 2048 + 2;
  const size_t head = (size_t)3 * kTileMaxBins + 64 + 4 * (size_t)kTileMaxGroups;
  TileScratch sc;
  sc.meta = c->d_tile + 3 * kTileMaxBins + 32;  // (the 32 free words behind tile_order)
  sc.ranges = c->d_tile + 3 * kTileMaxBins + 64;
  sc.totals = c->d_tile;
  sc.tile_off = c->d_tile + kTileMaxBins;
  sc.tile_order = c->d_tile + 2 * kTileMaxBins + 32;
  sc.P = c->d_tile + head;
  sc.T = sc.P + nblk_cap * kTileMaxBins;
  sc.C = sc.T + (size_t)kTileMaxGroups * kTileMaxBins;
  {
    {
      ScopedKernel k(c, K_TILE_HIST, (uint64_t)n * 16);  // ingest: the raw records, read once
      launch_tile_hist(cur_stream(c), evL, nL, evR, nR, c->tgeom, sc, c->d_rejected, mc, mc ? c->d_warp : nullptr);
    }
    {
      ScopedKernel k(c, K_TILE_SCAN, 0);  // (the count matrices: not in SURVEY's accounting)
      launch_tile_scan(cur_stream(c), nL, nR, c->tgeom, sc, c->d_rejected);
    }
    {
      ScopedKernel k(c, K_TILE_SCATTER, (uint64_t)n * 24);  // the partition's own traffic: 16 B in, 8 B out (16 for wide records)
      launch_tile_scatter(cur_stream(c), evL, nL, evR, nR, c->tgeom, sc, c->d_part, mc ? c->d_warp : nullptr);
    }
  }
  {
    ScopedKernel k(c, K_TILE_APPLY, (uint64_t)n * 32);
    launch_tile_apply(cur_stream(c), c->d_part, n, c->tgeom, sc, L2, S2, c->cfg.feature_filter_threshold,
                      arc_touched, c->z_counts + 3, c->lim.ticket);
  }
  return 0;
}

// arc_set >= 0: this batch's Arc* pass will run into candidate set arc_set; *arc_marked tells
// whether the update has set that set's touched flags on its way (else run_arc does it)
int sae_update(esvio_fe_ctx* c, const EventRec* evL, uint32_t nL, const EventRec* evR,
               uint32_t nR, const McParams* mc, double2* L2, double2* S2, int arc_set, bool* arc_marked) {
  const uint32_t n = nL + nR;
  if (!n) return 0;
  if (!L2) L2 = c->L2;  // (other planes: the scratch pair of the time-slice entry points)
  if (!S2) S2 = c->S2;
  if (arc_marked) *arc_marked = false;
  if (c->tiled) {
    // (motion compensation: the update happens at the warped pixels, Arc* is asked about the events'
    // own pixels (feature_tracker.cpp:698 -> :13-38), so the tiles' touched flags are not Arc*'s)
    uint8_t* mark = arc_set >= 0 && nL && !mc ? c->d_touched[arc_set] : nullptr;
    if (arc_marked) *arc_marked = mark != nullptr;
    return sae_update_tiled(c, evL, nL, evR, nR, L2, S2, mark, mc);
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
                      c->vals[cur ^ 1], c->z_counts + 3, c->lim.lookback);
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
                 const EventRec** dR, int lane) {
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
    ScopedKernel k(c, K_TIME_SURFACE4, (uint64_t)c->P * 17 * ncam);
    k.id = launch_time_surface(cur_stream(c), S2, c->W, c->H, t_sync, c->cfg.decay_ms / 1000.0,
                               c->cfg.ignore_polarity, r0, r1, stride, ncam);
  }
  if (mk > 0) {  // cv::medianBlur(2k+1) of the rendered surface (event_detector.cc:262-264)
    const size_t o = (size_t)kPad * stride + kPad;
    ScopedKernel k(c, K_MEDIAN, 0);
    launch_median(cur_stream(c), r0 + o, r1 + o, stride, dst0 + o, dst1 + o, stride, c->W, c->H, mk, ncam);
  }
}


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
  const bool fused_eq = c->fuse_ts_pyr && c->cfg.equalize && c->cfg.median_blur_kernel_size <= 0 &&
                        two[0].levels == 3 && two[1].levels == 3;
  if (fused_eq) {
    // time surfaces -> raw; CLAHE LUTs; CLAHE output -> a linear scratch pair (no in-place normalise:
    // the fused kernel's blocks read their neighbours' pixels); normalise + pyramid levels; borders + Scharr
    const PyrDesc& rl = c->raw[rawbuf][0].d;
    const PyrDesc& rr = c->raw[rawbuf][1].d;
    render_ts(c, t_sync, rl.img[0], rr.img[0], 2, c->S2);
    for (int stage = 0; stage < 2; stage++) {
      ScopedKernel k(c, K_CLAHE, stage == 0 ? (uint64_t)c->P * 2 : (uint64_t)c->P * 4);
      launch_clahe(cur_stream(c), px00(rl), px00(rr), rl.stride[0], c->d_eq_tmp, c->d_eq_tmp + c->P, c->W, c->W,
                   c->H, c->d_lut, c->d_minmax, 2, stage);
    }
    {
      uint64_t px = 0;
      for (int l = 0; l <= 3; l++) px += (uint64_t)two[0].w[l] * two[0].h[l];
      ScopedKernel k(c, K_NORM_PYR, ((uint64_t)c->P + px) * 2);
      launch_norm_pyr(cur_stream(c), c->d_eq_tmp, c->d_eq_tmp + c->P, c->W, c->d_minmax, two);
    }
    {
      uint64_t all = 0;
      for (int l = 0; l <= 3; l++) all += (uint64_t)two[0].w[l] * two[0].h[l];
      ScopedKernel k(c, K_PAD_SCHARR, all * 5 * 2);
      launch_pad_scharr(cur_stream(c), two, 2);
    }
    return;
  }
  if (!fused) {
    render_lk_images(c, t_sync, 3, slotL, slotR, rawbuf);
    pyr_build(c, two, 2);
    return;
  }
  {
    ScopedKernel k(c, K_TIME_SURFACE4, (uint64_t)c->P * 17 * 2);
    k.id = launch_time_surface(cur_stream(c), c->S2, c->W, c->H, t_sync, c->cfg.decay_ms / 1000.0, c->cfg.ignore_polarity,
                               two[0].img[0], two[1].img[0], two[0].stride[0], 2);
  }
  {
    uint64_t px = 0;
    for (int l = 0; l <= 3; l++) px += (uint64_t)two[0].w[l] * two[0].h[l];
    ScopedKernel k(c, K_PYR3, px * 2);
    launch_pyr3(cur_stream(c), two, 2);
  }
  {
    uint64_t all = 0;
    for (int l = 0; l <= 3; l++) all += (uint64_t)two[0].w[l] * two[0].h[l];
    ScopedKernel k(c, K_PAD_SCHARR, all * 5 * 2);
    launch_pad_scharr(cur_stream(c), two, 2);
  }
}

// one camera's LK image + pyramid (the fused kernels; the caller has checked render_cam_ok)
bool render_cam_ok(const esvio_fe_ctx* c) {
  return c->fuse_ts_pyr && !c->cfg.equalize && c->cfg.median_blur_kernel_size <= 0 && c->pyr[0].d.levels == 3;
}
void render_and_build_cam(esvio_fe_ctx* c, double t_sync, int cam, int slot) {
  const PyrDesc one = c->pyr[slot].d;
  uint64_t px = 0;
  for (int l = 0; l <= 3; l++) px += (uint64_t)one.w[l] * one.h[l];
  {
    ScopedKernel k(c, K_TIME_SURFACE4, (uint64_t)c->P * 17);
    k.id = launch_time_surface(cur_stream(c), c->S2 + (size_t)cam * c->P, c->W, c->H, t_sync, c->cfg.decay_ms / 1000.0,
                               c->cfg.ignore_polarity, one.img[0], one.img[0], one.stride[0], 1);
  }
  {
    ScopedKernel k(c, K_PYR3, px);
    launch_pyr3(cur_stream(c), &one, 1);
  }
  {
    ScopedKernel k(c, K_PAD_SCHARR, px * 5);
    launch_pad_scharr(cur_stream(c), &one, 1);
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
  ScopedKernel k(c, c->cfg.lk_accum == 2 ? K_LK_F32 : K_LK, bytes);
  LkArgs fa = f;
  fa.accum = c->cfg.lk_accum;
  launch_lk(cur_stream(c), fa, b, back_pts, back_status);
}

// A caller's W x H image <-> level 0 of a padded pyramid.  Not as one 2-D copy between the caller's
// pageable memory and the pitched device image: the runtime does that row by row (4.5 ms for a
// stereo pair of 346 x 260 images, more than the rest of trackImage together).  The rows go through
// a pinned staging ring as one block, one linear copy crosses PCIe, and the re-pitching is a
// device-to-device copy.
static int image_stage_slot(esvio_fe_ctx* c, size_t bytes, size_t* off) {
  constexpr int kSlots = 4;  // (a frame stages at most two images; a slot is reused two frames later,
                             // and every call that stages an image waits for its stream before it returns
                             // or, on the way in, before the next trackImage call can come)
  if (!c->h_img || c->img_stage_bytes < bytes) {
    HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
    if (c->h_img) (void)hipHostFree(c->h_img);
    if (c->d_img) (void)hipFree(c->d_img);
    c->h_img = nullptr;
    c->d_img = nullptr;
    c->img_stage_bytes = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_img, kSlots * bytes, hipHostMallocDefault));
    c->n_allocs++;
    if (int rc = dev_alloc(c, &c->d_img, kSlots * bytes)) return rc;
    c->img_stage_bytes = bytes;
  }
  *off = (size_t)(c->img_stage_next++ % kSlots) * c->img_stage_bytes;
  return 0;
}

int copy_level0_in(esvio_fe_ctx* c, const PyrDesc& d, const uint8_t* in) {
  const int stride = d.stride[0];
  const size_t bytes = (size_t)d.w[0] * d.h[0];
  size_t off = 0;
  if (int rc = image_stage_slot(c, bytes, &off)) return rc;
  std::memcpy(c->h_img + off, in, bytes);
  HIPCHK(c, hipMemcpyAsync(c->d_img + off, c->h_img + off, bytes, hipMemcpyHostToDevice, cur_stream(c)));
  HIPCHK(c, hipMemcpy2DAsync(d.img[0] + (size_t)kPad * stride + kPad, stride, c->d_img + off, d.w[0], d.w[0],
                             d.h[0], hipMemcpyDeviceToDevice, cur_stream(c)));
  return 0;
}

int copy_level0_out(esvio_fe_ctx* c, const PyrDesc& d, uint8_t* out) {
  const int stride = d.stride[0];
  const size_t bytes = (size_t)d.w[0] * d.h[0];
  size_t off = 0;
  if (int rc = image_stage_slot(c, bytes, &off)) return rc;
  HIPCHK(c, hipMemcpy2DAsync(c->d_img + off, d.w[0], d.img[0] + (size_t)kPad * stride + kPad, stride, d.w[0],
                             d.h[0], hipMemcpyDeviceToDevice, cur_stream(c)));
  HIPCHK(c, hipMemcpyAsync(c->h_img + off, c->d_img + off, bytes, hipMemcpyDeviceToHost, cur_stream(c)));
  HIPCHK(c, hipStreamSynchronize(cur_stream(c)));
  std::memcpy(out, c->h_img + off, bytes);
  return 0;
}

// ---------------------------------------------------------------- host bookkeeping (reference
// helpers in feature_tracker.cpp)

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
  if (c->trace) c->tr_fm_class[c->cur_pts.size() < 8 ? 0 : c->cur_pts.size() < 15 ? 1 : 2]++;
  if (c->cur_pts.size() >= 8) {
    const esvio_fe_camera& cam = c->cfg.cam[0];
    const double FOCAL = c->cfg.focal_length;
    const size_t n = c->prev_pts.size();
    std::vector<float> un_cur(n * 2), un_prev(n * 2);
    std::vector<double> lx(n), ly(n);
    const double cx = c->W / 2.0, cy = c->H / 2.0;
    const auto tl = std::chrono::steady_clock::now();
    // (prev_pts' lifts: done under the temporal LK's wait when the frame publishes, TrackCall::temporal)
    const bool pre = c->pre_lift_valid && c->pre_lx.size() == n;
    if (!pre) host::lift_projective_batch(cam, &c->prev_pts[0].x, (int)n, lx.data(), ly.data());
    const double* plx = pre ? c->pre_lx.data() : lx.data();
    const double* ply = pre ? c->pre_ly.data() : ly.data();
    c->pre_lift_valid = false;  // (one use: the vectors belong to the call that made them)
    for (size_t i = 0; i < n; i++) {  // p[2] == 1.0: x / 1.0 is exact
      un_prev[2 * i] = (float)(FOCAL * plx[i] / 1.0 + cx);
      un_prev[2 * i + 1] = (float)(FOCAL * ply[i] / 1.0 + cy);
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


Pin pin_of(esvio_fe_ctx* c, int set) {
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
  s.disc_c = disc_threshold(s.hw, s.radius);
  s.out_pts = out_pts;
  s.out_idx = out_idx;
  s.out_base = out_base;
  s.n_out = c->d_counts;
  s.n_total = c->d_counts + 1;
  s.host_counts = nullptr;
  s.init_bits = nullptr;
  s.gbitmap = nullptr;
  s.pub_slots = nullptr;
  s.pub_done = nullptr;
  s.pub_seq = 0;
  s.one_wave = c->select_one_wave ? 1 : 0;
  return s;
}

size_t select_lds_bytes(const esvio_fe_ctx* c) {
  // bitmap + half-width table + the kept points whose discs seed the bitmap
  return ((size_t)c->H * ((c->W + 31) / 32) + 4 + 64 + (size_t)std::max(c->cfg.max_cnt, 1)) * 4;
}
size_t select_tables_lds_bytes(const esvio_fe_ctx* c) {  // with the bitmap in global memory
  return (4 + 64 + (size_t)std::max(c->cfg.max_cnt, 1)) * 4;
}

// ordered compaction of candidate set `set` (right behind the k_arc that filled it)
void run_compact(esvio_fe_ctx* c, uint32_t n_events, int set) {
  const uint32_t nblk = (n_events + kArcBlock - 1) / kArcBlock;
  const esvio_fe_ctx::CandSet& cs = c->cand[set];
  ScopedKernel k(c, K_COMPACT, 0);
  launch_compact(cur_stream(c), cs.xy, cs.idx, cs.cnt, nblk, cs.comp_xy, cs.comp_idx, cs.total, cs.grp);
}

// the sequential greedy (Event_FeaturesToTrack) over the compacted candidates of set `set`;
// `mask_bits`: blocked pixels the disc bitmap starts from (null: none, or already applied by k_arc)
void run_select(esvio_fe_ctx* c, int set, int max_corners, float2* out_pts, int out_base,
                int32_t* out_idx, const uint32_t* mask_bits, int* host_counts, bool publish,
                const float2* stamp_pts, int n_stamp) {
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
  size_t lds = select_lds_bytes(c);
  if (!c->select_ok) {  // the bitmap does not fit LDS: it lives in device memory (slower, same result)
    if (!c->d_sel_bitmap && dev_alloc(c, &c->d_sel_bitmap, (size_t)c->H * s.wpr + 4) != 0) return;
    s.gbitmap = c->d_sel_bitmap;
    lds = select_tables_lds_bytes(c);
  }
  ScopedKernel k(c, K_SELECT_MW, 0);
  k.id = launch_select(cur_stream(c), s, lds);
}

// Arc* flags (+ ordered per-block candidate lists into set `set`) for the left events; `ts` is the
// RAW left time surface the TS_LK_THRESHOLD test reads (null: no test)
void run_arc(esvio_fe_ctx* c, const EventRec* ev, uint32_t n, const PyrDesc* ts, bool use_mask,
             bool want_flags, bool want_cand, int set, bool marked) {
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
  // Only a pixel's earliest candidate can be accepted (a later one finds the pixel blocked whatever happened to the
  // first): an atomicMin per candidate here + k_dedup halve the list k_select walks.  That paid with the one-wave
  // k_select (54 against 57 us at 0.17 M left events); k_select_mw does not care until it has to dig through the
  // whole list — max_cnt 1000: 63 us with, 98 without — while the atomics cost k_arc_ev 2.4 us at C3, 9 at 20 Mev/s
  // (0.131 -> 0.1225 ms per replay step without them), 43 at 3.3 M left events, and most of its HBM writes.  So:
  // handles with max_cnt > 500 only (esvio_fe_create; ESVIO_FE_DEDUP=1 / ESVIO_FE_NO_DEDUP=1 force it), batches < 2^20.
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
    ScopedKernel k(c, K_ARC_EV, (uint64_t)n * 16);
    launch_arc(cur_stream(c), a);
  }
  if (dedup) {
    ScopedKernel k(c, K_DEDUP, 0);
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

}  // namespace fe
}  // namespace esvio
