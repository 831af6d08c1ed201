// fe_api.cpp — the C ABI (include/esvio_fe.h): handle lifetime and the entry points, each a thin
// layer over fe_stages.cpp / fe_track.cpp / fe_image.cpp.
#include "fe_internal.h"

// RCCL is looked up at run time (dlopen librccl.so), so the library does not depend on it unless the
// exchange entry points are used.
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/,
                                 hipStream_t);
struct NcclId {
  char internal[128];  // ncclUniqueId
};
typedef int (*nccl_get_id_fn)(NcclId*);
typedef int (*nccl_init_rank_fn)(void** /*ncclComm_t* */, int, NcclId, int);
typedef int (*nccl_destroy_fn)(void*);
// an RCCL that is already in the process (e.g. the one a PyTorch process group uses) is preferred
// to loading a second one
void* rccl_lib() {
  static void* lib = []() -> void* {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if (void* l = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)) return l;
    for (const char* n : names)
      if (void* l = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return l;
    return dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  }();
  return lib;
}
template <class F>
F rccl_sym(const char* name) {
  void* l = rccl_lib();
  return l ? (F)dlsym(l, name) : nullptr;
}
nccl_allgather_fn rccl_all_gather() {
  static nccl_allgather_fn fn = rccl_sym<nccl_allgather_fn>("ncclAllGather");
  return fn;
}
int exchange_buffers(esvio_fe_ctx* c, int world) {
  const size_t cnt = (size_t)2 * std::max(c->cfg.max_cnt, 1) * 8;
  if (!c->x_send) {
    if (int rc = dev_alloc(c, &c->x_send, cnt)) return rc;
    HIPCHK(c, hipHostMalloc((void**)&c->x_pin, cnt * sizeof(float), hipHostMallocDefault));
  }
  if ((size_t)world * cnt > c->x_recv_cap) {
    if (c->x_recv) (void)hipFree(c->x_recv);
    if (c->x_pin_recv) (void)hipHostFree(c->x_pin_recv);
    c->x_recv = nullptr;
    c->x_pin_recv = nullptr;
    c->x_recv_cap = 0;
    if (int rc = dev_alloc(c, &c->x_recv, (size_t)world * cnt)) return rc;
    HIPCHK(c, hipHostMalloc((void**)&c->x_pin_recv, (size_t)world * cnt * sizeof(float), hipHostMallocDefault));
    c->x_recv_cap = (size_t)world * cnt;
  }
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
  (void)launcher_set(c, false);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream4) (void)hipStreamSynchronize(c->stream4);
  if (c->stream6) (void)hipStreamSynchronize(c->stream6);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  host::ransac_pool_destroy(c->pool);
  c->pool = nullptr;
  stager_destroy(c);
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
    for (int pub = 0; pub < 2; pub++)
      if (c->phase_count[pub])
        fprintf(stderr, "\n[esvio_fe trace] %s frames, host bookkeeping, ms/frame: left undistort + velocity=%.4f previous "
                "frames' right tails=%.4f this frame's right tail=%.4f copies + profile + exchange=%.4f",
                pub ? "published" : "unpublished", c->tail_ms[pub][0] / c->phase_count[pub],
                c->tail_ms[pub][1] / c->phase_count[pub], c->tail_ms[pub][2] / c->phase_count[pub],
                c->tail_ms[pub][3] / c->phase_count[pub]);
    fprintf(stderr, "\n[esvio_fe trace] rejectWithF_event calls by point count: %llu with < 8 (skipped), %llu with "
            "8..14 (LMedS, 300 hypotheses), %llu with >= 15 (RANSAC)", (unsigned long long)c->tr_fm_class[0],
            (unsigned long long)c->tr_fm_class[1], (unsigned long long)c->tr_fm_class[2]);
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
  if (c->x_pending) (void)hipEventSynchronize(c->x_done);
  if (c->x_comm)
    if (nccl_destroy_fn destroy = rccl_sym<nccl_destroy_fn>("ncclCommDestroy")) (void)destroy(c->x_comm);
  if (c->x_stream) (void)hipStreamDestroy(c->x_stream);
  if (c->x_done) (void)hipEventDestroy(c->x_done);
  if (c->x_pin) (void)hipHostFree(c->x_pin);
  if (c->x_pin_recv) (void)hipHostFree(c->x_pin_recv);
  void* ptrs[] = {c->x_send, c->x_recv, c->d_part, c->d_warp, c->d_tile, c->L2s, c->S2s, c->slice_stage, c->L2, c->S2, c->d_ev, c->keys[0], c->keys[1], c->vals[0], c->vals[1], c->hist, c->sae_marks,
                  c->d_rejected, c->d_res, c->d_ptsD, c->d_flags, c->d_pub_slots, c->d_pub_done, c->d_chain, c->d_lane_gate, c->d_gftt_cov, c->d_gftt_rowsum, c->d_gftt_eig, c->d_gftt_max,
                  c->d_mask_bits, c->d_sel_idx, c->d_sel_bitmap, c->d_eq_tmp,
                  c->tmp_pyr[0].mem, c->tmp_pyr[1].mem, c->med_tmp[0].mem, c->med_tmp[1].mem, c->d_lut,
                  c->d_minmax};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& cs : c->cand)
    for (void* p : {(void*)cs.xy, (void*)cs.idx, (void*)cs.cnt, (void*)cs.grp, (void*)cs.comp_xy, (void*)cs.comp_idx,
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
  if (c->h_img) (void)hipHostFree(c->h_img);
  if (c->d_img) (void)hipFree(c->d_img);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_spec) (void)hipHostFree(c->h_spec);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream4) (void)hipStreamDestroy(c->stream4);
  if (c->stream6) (void)hipStreamDestroy(c->stream6);
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
  if (c->ev_imgs_ready) (void)hipEventDestroy(c->ev_imgs_ready);
  if (c->ev_arc_side) (void)hipEventDestroy(c->ev_arc_side);
  if (c->ev_sae_left) (void)hipEventDestroy(c->ev_sae_left);
  if (c->ev_right_ready) (void)hipEventDestroy(c->ev_right_ready);
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
  if (cfg->lk_accum != 1 && cfg->lk_accum != 2) return ESVIO_FE_EINVAL;
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
  if (const char* v = getenv("ESVIO_FE_SLOW_CALL_MS")) c->slow_call_ms = atof(v);
  c->mask_event.reset(c->W, c->H);

  auto bail = [&](int rc) {
    esvio_fe_destroy(c);
    return rc;
  };
  // the frame's own chain (LK, selection: few, latency-bound waves) outranks the prefetch stream's
  // wide kernels, which have a whole frame of slack
  {
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) {
      (void)hipGetLastError();
      n_cu = 64;  // (unknown: assume a small device)
    }
    const int lk_blocks = (std::max(cfg->max_cnt, 1) + kLkPointsPerBlock - 1) / kLkPointsPerBlock;
    c->n_cu = n_cu;
    c->waits_fit_spec = lk_blocks + 2 <= n_cu;
    c->waits_fit_chain = 2 * lk_blocks + 2 <= n_cu;
  }
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
      hipEventCreateWithFlags(&c->ev_planes_free, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_imgs_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_arc_side, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_sae_left, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_right_ready, hipEventDisableTiming) != hipSuccess)
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
  if ((rc = dev_alloc(c, &c->d_lane_gate, 16))) return bail(rc);
  c->stage_threads = stager_threads_from_env();
  if (const char* v = getenv("ESVIO_FE_FAULT")) esvio_fe_debug_inject(c, atoi(v));
  if (const char* e = getenv("ESVIO_FE_STEREO_SPLIT")) c->stereo_split_env = atoi(e) != 0;  // (else: fe_track.cpp decides)
  c->chain_enabled = getenv("ESVIO_FE_NO_CHAIN") == nullptr;
  c->cam_split_enabled = getenv("ESVIO_FE_NO_CAMSPLIT") == nullptr;
  // (the per-pixel dedup of the Arc* candidates pays only where the selection digs deep into its list: fe_stages.cpp run_arc)
  c->dedup_enabled = getenv("ESVIO_FE_NO_DEDUP") == nullptr && (cfg->max_cnt > 500 || getenv("ESVIO_FE_DEDUP") != nullptr);
  c->fuse_ts_pyr = getenv("ESVIO_FE_NO_FUSE") == nullptr;
  c->select_one_wave = getenv("ESVIO_FE_SELECT_SERIAL") != nullptr;
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
  if (hipMemsetAsync(c->d_lane_gate, 0, 64, cur_stream(c)) != hipSuccess ||
      hipMemsetAsync(c->d_chain, 0, std::max<size_t>(M, 1) * 16, cur_stream(c)) != hipSuccess ||
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
    // (CLAHE scratch: d_lut, d_minmax and d_eq_tmp are single buffers shared by the main stream and the
    // prefetch stream; rendering on one is ordered behind the other's through ev_planes_free /
    // ev_lane_done, like the SAE planes they are derived from)
    if ((rc = dev_alloc(c, &c->d_eq_tmp, (size_t)2 * c->P))) return bail(rc);
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
  // One kernel of this library on every stream, now: the runtime loads the code object with the first
  // launch from it and creates a stream's hardware queue with the stream's first use — 2.0-2.4 ms that
  // would otherwise sit inside the first esvio_fe_track_event call (profiles/r04_stall_forensics.md).
  for (hipStream_t st : {c->stream, c->stream2, c->stream3, c->stream4}) {
    launch_fill_f64(st, (double*)c->d_rejected, 1, 0.0);
    if (hipStreamSynchronize(st) != hipSuccess) return bail(ESVIO_FE_EHIP);
  }
  // ... and a burst of launches chained across the streams by events, nothing of it awaited until the end.  What it
  // is for: in one cold bench process of ten ONE track call around frame 20 of the first timed pass took 1.6-4.7 ms,
  // inside a HIP launch call of the calling thread or in its wait for the launch thread — never a lost CPU, never an
  // allocation of ours (profiles/r05_stall_hunt.txt).  Read as the runtime growing a pool (completion signals,
  // command records) inside whichever launch needs one more than it has; the replay schedule keeps ~60 launches and
  // ~25 cross-stream waits in flight, this puts 4 x 96 launches and as many waits in flight at once — and 30 cold
  // processes then ran without one call above 0.6 ms (profiles/r05_stall_hunt_after.txt).
  {
    const hipStream_t st[4] = {c->stream, c->stream2, c->stream3, c->stream4};
    hipEvent_t ev[4] = {};  // (events of the warm-up's own: the handle's stay unrecorded until a frame records them)
    bool ok = true;
    for (int i = 0; i < 4; i++) ok = ok && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
    for (int r = 0; ok && r < 96; r++)
      for (int i = 0; ok && i < 4; i++) {
        launch_spin(st[i], r == 0 ? 20000 : 0);  // (the first round's kernels hold everything behind them for 200 us)
        ok = hipEventRecord(ev[i], st[i]) == hipSuccess && hipStreamWaitEvent(st[(i + 1) & 3], ev[i], 0) == hipSuccess;
      }
    for (int i = 0; i < 4; i++) ok = (hipStreamSynchronize(st[i]) == hipSuccess) && ok;
    for (int i = 0; i < 4; i++)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (!ok) return bail(ESVIO_FE_EHIP);
  }
  if ((rc = stereo_split_prepare(c))) return bail(rc);  // (ESVIO_FE_STEREO_SPLIT=1)
  // Which of these streams share a hardware queue?  The runtime hands out at most GPU_MAX_HW_QUEUES (4) queues per
  // priority level and process, then doubles up — and two streams on one queue run their kernels one after the
  // other (tools/queue_probe.hip), which for this schedule means a frame's prefetch behind another frame's waiting
  // LK launch.  Counted once per handle (a 100 us spin on one stream, an empty kernel on the other: ~1.2 ms for
  // the ten pairs), reported by esvio_fe_debug_counters and, with ESVIO_FE_QUEUE_PROBE=1, on stderr.
  if (getenv("ESVIO_FE_QUEUE_PROBE")) {
    const hipStream_t st[5] = {c->stream, c->stream2, c->stream3, c->stream4, c->stream6};  // (stream6: only where the handle splits)
    const char* nm[5] = {"main", "prefetch", "speculative", "stereo", "stereo-unpublished"};
    const int ns = c->stream6 ? 5 : 4;
    for (int a = 0; a < ns; a++)
      for (int b = 0; b < ns; b++) {
        if (a == b || st[a] == st[b]) continue;
        launch_spin(st[a], 10000);
        const auto t0 = std::chrono::steady_clock::now();
        launch_fill_f64(st[b], (double*)c->d_rejected, 1, 0.0);
        (void)hipStreamSynchronize(st[b]);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        (void)hipStreamSynchronize(st[a]);
        if (us > 60.0) {
          c->n_queue_conflicts++;
          fprintf(stderr, "[esvio_fe] streams '%s' and '%s' share a hardware queue (%.0f us behind a 100 us kernel)\n", nm[b], nm[a], us);
        }
      }
    if (!c->n_queue_conflicts) fprintf(stderr, "[esvio_fe] the handle's %d streams have a hardware queue each\n", ns);
  }
  // The greedy selections (Event_FeaturesToTrack, goodFeaturesToTrack's min-distance pass) keep
  // their one-bit-per-pixel map in LDS; above ~1.3 M pixels (the frame cameras of the shipped ESVIO
  // configs go up to 1920x1200) it lives in device memory instead (k_select_gbm)
  c->select_ok = select_lds_bytes(c) <= 160 * 1024;
  *out = c;
  return 0;
}

int esvio_fe_reset(esvio_fe_handle c) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  (void)launcher_drain(c);  // (whatever it was still issuing is discarded with the batches below)
  launcher_clear_error(c);
  HIPCHK(c, hipStreamSynchronize(c->stream3));
  HIPCHK(c, hipStreamSynchronize(c->stream4));
  if (c->stream6) HIPCHK(c, hipStreamSynchronize(c->stream6));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  // (the main stream as well: a call that returned early in lazy mode, or one that failed half way,
  // may have kernels there that still raise the error flag or write into the pinned result words
  // cleared below)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->announced.clear();
  c->inflight.clear();
  stager_drain(c);
  c->cur_stage = -1;
  // (a call that failed with ESVIO_FE_EINTERNAL: the expired wait's flag, the sort's scratch words)
  pin_of(c).counts[3] = 0;
  if (c->hist) HIPCHK(c, hipMemsetAsync(c->hist, 0, c->hist_cap * 4, cur_stream(c)));
  std::memset(c->h_spec, 0, 2 * c->spec_bytes);
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
  c->ext_sae_pending = false;  // (the planes move on: a committed time-sliced batch is not "the next frame's" any more)
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
  const McParams mc = make_mc_params(motion);
  c->ext_sae_pending = false;
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
  // A rank that never tracks: the previous batch's commit simply IS its planes by now.  On a rank that
  // tracks, a committed batch must get its track call before the next batch's slices start — the call
  // would otherwise apply the batch to the planes a second time.
  if (c->ext_sae_pending && c->frame_no > 0)
    return fail(c, ESVIO_FE_EINVAL, "the committed batch has not been tracked yet (slice_last of the next batch "
                                    "comes after esvio_fe_track_event on a rank that tracks)");
  c->ext_sae_pending = false;
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
  if (!c->inflight.empty() || !c->announced.empty())
    return fail(c, ESVIO_FE_EINVAL, "time-sliced SAE update cannot be mixed with esvio_fe_set_next_batch");
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
  if (!c->inflight.empty() || !c->announced.empty())
    return fail(c, ESVIO_FE_EINVAL, "time-sliced SAE update cannot be mixed with esvio_fe_set_next_batch");
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
  if (space == ESVIO_FE_HOST) return copy_level0_out(c, d, dst);  // (staged: no 2-D copy over PCIe)
  const int stride = d.stride[0];
  HIPCHK(c, hipMemcpy2DAsync(dst, c->W, d.img[0] + (size_t)kPad * stride + kPad, stride, c->W, c->H,
                             hipMemcpyDeviceToDevice, cur_stream(c)));
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
  if (space == ESVIO_FE_HOST) {  // (staged through pinned memory: src is free again on return)
    if (int rc = copy_level0_in(c, d, src)) return rc;
  } else {
    HIPCHK(c, hipMemcpy2DAsync(d.img[0] + (size_t)kPad * stride + kPad, stride, src, c->W, c->W, c->H,
                               hipMemcpyDeviceToDevice, cur_stream(c)));
  }
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

int esvio_fe_host_hypot(const double* x, const double* y, int n, double* out) {
  if (n < 0 || (n && (!x || !y || !out))) return ESVIO_FE_EINVAL;
  host::host_hypot(x, y, n, out);
  return ESVIO_FE_OK;
}

int esvio_fe_host_stage_copy(void* dst, const void* src, size_t len) {
  if (len && (!dst || !src)) return ESVIO_FE_EINVAL;
  stager_copy_bytes((uint8_t*)dst, (const uint8_t*)src, len);
  return ESVIO_FE_OK;
}

int esvio_fe_host_stage_pack(void* dst, const void* src, size_t len, uint32_t* base_sec) {
  uint32_t b = 0;
  if (len && (!dst || !src)) return ESVIO_FE_EINVAL;
  const bool ok = stager_pack_bytes((uint8_t*)dst, (const uint8_t*)src, len, &b);
  if (base_sec) *base_sec = b;
  return ok ? 1 : 0;
}

int esvio_fe_staging_counters(esvio_fe_handle c, uint64_t out4[4]) {
  if (!c || !out4) return ESVIO_FE_EINVAL;
  stager_counters(c, out4);
  return ESVIO_FE_OK;
}

int esvio_fe_host_nullspace(const double* systems, int n, int lanes, double* f12, int32_t* redone) {
  if (n < 0 || (n && (!systems || !f12))) return ESVIO_FE_EINVAL;
  const int r = host::host_nullspace(systems, n, lanes, f12);
  if (redone) *redone = r;
  return ESVIO_FE_OK;
}

int esvio_fe_ransac_stats(uint64_t* out6, int reset) {
  if (!out6) return ESVIO_FE_EINVAL;
  const host::RansacStats r = host::ransac_stats(reset != 0);
  out6[0] = r.calls;
  out6[1] = r.iterations;
  out6[2] = r.points;
  out6[3] = r.ns;
  out6[4] = r.lmeds_calls;
  out6[5] = r.lmeds_ns;
  return ESVIO_FE_OK;
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

int esvio_fe_find_fundamental_mat_held(const float* p1, const float* p2, int n, double thr, double conf,
                                       int threads, int hold_mask, uint8_t* status, int32_t* n_inliers) {
  if (n < 0 || (n && (!p1 || !p2 || !status)) || threads < 2 || threads > 16 || hold_mask < 0 || hold_mask > 3)
    return ESVIO_FE_EINVAL;
  host::RansacPool* pool = host::ransac_pool_create(threads - 1);
  host::ransac_pool_hold(pool, hold_mask, true);
  const int k = host::find_fundamental_mat(p1, p2, n, thr, conf, status, pool);
  host::ransac_pool_hold(pool, hold_mask, false);
  host::ransac_pool_destroy(pool);
  if (n_inliers) *n_inliers = k;
  return 0;
}

namespace {
struct IdleTap {
  std::atomic<int> pending{0};
  std::atomic<uint64_t> calls{0}, done{0};
};
bool idle_tap_fn(void* arg) {  // what the staging hands the helpers: one unit of work, if there is one
  IdleTap* t = (IdleTap*)arg;
  t->calls.fetch_add(1, std::memory_order_relaxed);
  int v = t->pending.load(std::memory_order_acquire);
  while (v > 0)
    if (t->pending.compare_exchange_weak(v, v - 1, std::memory_order_acq_rel)) {
      const auto t0 = std::chrono::steady_clock::now();  // (a unit: ~5 us, a staging chunk's order of magnitude)
      while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(5)) {}
      t->done.fetch_add(1, std::memory_order_relaxed);
      return true;
    }
  return false;
}
}  // namespace

int esvio_fe_find_fundamental_mat_idle(const float* p1, const float* p2, int n, double thr, double conf,
                                       int threads, int repeats, int idle_units, uint8_t* status, int32_t* n_inliers,
                                       uint64_t out3[3]) {
  if (n < 0 || (n && (!p1 || !p2 || !status)) || threads < 2 || threads > 16 || repeats < 1 || idle_units < 0 || !out3)
    return ESVIO_FE_EINVAL;
  host::RansacPool* pool = host::ransac_pool_create(threads - 1);
  IdleTap tap;
  host::ransac_pool_set_idle_work(pool, &tap.pending, idle_tap_fn, &tap);
  int k = 0;
  for (int r = 0; r < repeats; r++) {
    tap.pending.fetch_add(idle_units, std::memory_order_acq_rel);  // (work arrives while the helpers spin)
    k = host::find_fundamental_mat(p1, p2, n, thr, conf, status, pool);
  }
  const auto t0 = std::chrono::steady_clock::now();  // what is left is taken while the helpers idle (bounded wait)
  while (tap.pending.load(std::memory_order_acquire) > 0 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2)) {}
  host::ransac_pool_set_idle_work(pool, nullptr, nullptr, nullptr);  // (returns once nobody is inside the hook)
  out3[0] = tap.calls.load();
  out3[1] = tap.done.load();
  out3[2] = (uint64_t)std::max(0, tap.pending.load());
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
    // (an empty vector's data() may be null, and memcpy's source must not be, whatever the size)
    auto put = [](void* dst, const void* src, size_t bytes) {
      if (dst && bytes) std::memcpy(dst, src, bytes);
    };
    put(out->ids, c->ids.data(), nl * 4);
    put(out->track_cnt, c->track_cnt.data(), nl * 4);
    put(out->cur_pts, c->cur_pts.data(), nl * 8);
    put(out->cur_un_pts, c->cur_un_pts.data(), nl * 8);
    put(out->pts_velocity, c->pts_velocity.data(), nl * 8);
    put(out->ids_right, c->ids_right.data(), nr * 4);
    put(out->cur_right_pts, c->cur_right_pts.data(), nr * 8);
    put(out->cur_un_right_pts, c->cur_un_right_pts.data(), nr * 8);
    put(out->right_pts_velocity, c->right_pts_velocity.data(), nr * 8);
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
    // (in this order: the previous published frame's new corners — which that frame's call may have left to "the next
    // call" while their stereo LK was running — extend the map this frame's right-camera velocities read)
    if (int rc = finalize_pending(c)) return rc;
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
int esvio_fe_exchange_tracks(esvio_fe_handle c, void* nccl_comm, int world, float* gathered) {
  if (!c || !nccl_comm || world < 1 || !gathered) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  nccl_allgather_fn all_gather = rccl_all_gather();
  if (!all_gather) return fail(c, ESVIO_FE_ENOTIMPL, "librccl.so not found (dlopen): %s", dlerror());
  const size_t cnt = (size_t)2 * std::max(c->cfg.max_cnt, 1) * 8;
  // the send / receive areas are shared with the asynchronous exchange of the handle's own
  // communicator: whatever that one still has packed or in flight goes first and is waited for
  if (int rc = exchange_flush(c)) return rc;
  if (c->x_pending) HIPCHK(c, hipEventSynchronize(c->x_done));
  if (c->x_pending && (size_t)world * cnt > c->x_recv_cap)
    c->x_pending = false;  // (its gathered block is dropped with the buffer that is about to grow)
  if (int rc = exchange_buffers(c, world)) return rc;
  if (int rc = esvio_fe_pack_track_records(c, c->x_pin, nullptr)) return rc;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->x_send, c->x_pin, cnt * 4, hipMemcpyHostToDevice, st));
  const int nrc = all_gather(c->x_send, c->x_recv, cnt, 7 /* ncclFloat32 */, nccl_comm, st);
  if (nrc != 0) return fail(c, ESVIO_FE_EHIP, "ncclAllGather failed: %d", nrc);
  HIPCHK(c, hipMemcpyAsync(gathered, c->x_recv, (size_t)world * cnt * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return 0;
}

int esvio_fe_comm_unique_id(uint8_t id[128]) {
  if (!id) return ESVIO_FE_EINVAL;
  nccl_get_id_fn get_id = rccl_sym<nccl_get_id_fn>("ncclGetUniqueId");
  if (!get_id) return ESVIO_FE_ENOTIMPL;
  NcclId u;
  if (get_id(&u) != 0) return ESVIO_FE_EHIP;
  std::memcpy(id, u.internal, 128);
  return 0;
}

int esvio_fe_comm_init(esvio_fe_handle c, const uint8_t id[128], int rank, int world) {
  if (!c || !id || world < 1 || rank < 0 || rank >= world) return ESVIO_FE_EINVAL;
  if (c->x_comm) return fail(c, ESVIO_FE_EINVAL, "the handle has a communicator already");
  HIPCHK(c, hipSetDevice(c->dev));
  nccl_init_rank_fn init_rank = rccl_sym<nccl_init_rank_fn>("ncclCommInitRank");
  if (!init_rank || !rccl_all_gather()) return fail(c, ESVIO_FE_ENOTIMPL, "librccl.so not found (dlopen)");
  NcclId u;
  std::memcpy(u.internal, id, 128);
  void* comm = nullptr;
  const int nrc = init_rank(&comm, world, u, rank);
  if (nrc != 0 || !comm) return fail(c, ESVIO_FE_EHIP, "ncclCommInitRank failed: %d", nrc);
  c->x_comm = comm;
  c->x_world = world;
  if (!c->x_stream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->x_stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->x_done, hipEventDisableTiming));
  }
  return exchange_buffers(c, world);
}

}  // extern "C"
namespace esvio {
namespace fe {
int exchange_pack(esvio_fe_ctx* c) {
  // (the previous exchange has to be through with the pinned areas)
  if (c->x_pending) HIPCHK(c, hipEventSynchronize(c->x_done));
  c->x_pending = false;
  if (int rc = esvio_fe_pack_track_records(c, c->x_pin, nullptr)) return rc;
  c->x_deferred = true;
  return 0;
}
int exchange_flush(esvio_fe_ctx* c) {
  if (!c->x_deferred) return 0;
  c->x_deferred = false;
  const size_t cnt = (size_t)2 * std::max(c->cfg.max_cnt, 1) * 8;
  HIPCHK(c, hipMemcpyAsync(c->x_send, c->x_pin, cnt * 4, hipMemcpyHostToDevice, c->x_stream));
  const int nrc = rccl_all_gather()(c->x_send, c->x_recv, cnt, 7 /* ncclFloat32 */, c->x_comm, c->x_stream);
  if (nrc != 0) return fail(c, ESVIO_FE_EHIP, "ncclAllGather failed: %d", nrc);
  HIPCHK(c, hipMemcpyAsync(c->x_pin_recv, c->x_recv, (size_t)c->x_world * cnt * 4, hipMemcpyDeviceToHost,
                           c->x_stream));
  HIPCHK(c, hipEventRecord(c->x_done, c->x_stream));
  c->x_pending = true;
  return 0;
}
}  // namespace fe
}  // namespace esvio
extern "C" {

int esvio_fe_exchange_begin(esvio_fe_handle c) {
  if (!c) return ESVIO_FE_EINVAL;
  if (!c->x_comm) return fail(c, ESVIO_FE_EINVAL, "esvio_fe_comm_init has not been called");
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = exchange_flush(c)) return rc;  // (an automatic one that is still waiting goes first)
  if (int rc = exchange_pack(c)) return rc;
  return exchange_flush(c);
}

int esvio_fe_set_auto_exchange(esvio_fe_handle c, int on) {
  if (!c) return ESVIO_FE_EINVAL;
  if (on && !c->x_comm) return fail(c, ESVIO_FE_EINVAL, "esvio_fe_comm_init has not been called");
  c->x_auto = on != 0;
  return 0;
}

int esvio_fe_exchange_end(esvio_fe_handle c, float* gathered) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = exchange_flush(c)) return rc;
  if (!c->x_pending) return fail(c, ESVIO_FE_EINVAL, "no exchange in flight");
  HIPCHK(c, hipEventSynchronize(c->x_done));
  c->x_pending = false;
  if (gathered)
    std::memcpy(gathered, c->x_pin_recv, (size_t)c->x_world * 2 * std::max(c->cfg.max_cnt, 1) * 8 * sizeof(float));
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
  stager_share_pool(c);
  return 0;
}

int esvio_fe_finish(esvio_fe_handle c, esvio_fe_tracks* out) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = finalize_pending(c)) return rc;
  if (int rc = finalize_right(c)) return rc;
  return fill_tracks(c, out);
}

static int set_next_batch_impl(esvio_fe_handle c, double next_cur_time, const esvio_fe_event* left,
                               size_t nL, const esvio_fe_event* right, size_t nR, int space,
                               int pub_hint, const esvio_fe_motion* motion) {
  if (!c) return ESVIO_FE_EINVAL;
  if (nL == 0 || !left || (nR && !right)) return fail(c, ESVIO_FE_EINVAL, "bad next batch");
  if (space != ESVIO_FE_HOST && space != ESVIO_FE_DEVICE) return ESVIO_FE_EINVAL;
  if (c->ext_right_pending) return fail(c, ESVIO_FE_EINVAL, "not with an imported right image");
  if ((int)(c->announced.size() + c->inflight.size()) >= 2 * kPrefetchDepth)
    return fail(c, ESVIO_FE_EINVAL, "at most %d batches can be announced and not yet tracked", 2 * kPrefetchDepth);
  Batch b;
  b.time = next_cur_time;
  b.left = left;
  b.nL = nL;
  b.right = right;
  b.nR = nR;
  b.space = space;
  b.pub = pub_hint != 0;
  if (motion) {
    b.has_motion = true;
    b.motion = *motion;
  }
  if (space == ESVIO_FE_HOST && stager_enabled(c)) {
    // the batch starts on its way to the device now: pinned chunks + DMA by the helper threads, under
    // the frames tracked before it
    HIPCHK(c, hipSetDevice(c->dev));
    // (one DMA for the whole batch.  Round 4 measured two to four, and the odd ones on a second copy stream:
    // no gain in the bench's configuration — 0.136-0.146 ms/step either way — and a loss without the RANSAC
    // helpers' share of the copying, 0.140 -> 0.165-0.18; two DMA engines at once: 0.197)
    if (int rc = stager_begin(c, left, nL, right, nR, 1, &b.stage)) return rc;
  } else if (space == ESVIO_FE_DEVICE && !c->stream6) {
    // (the first announcement of a batch that is already on the device: the second stereo stream, if this handle
    // is going to use it — here and not in esvio_fe_set_launch_thread, because a handle that is fed host batches
    // never splits and a stream it does not use still takes a hardware queue from the process's pool)
    HIPCHK(c, hipSetDevice(c->dev));
    if (int rc = stereo_split_prepare(c)) return rc;
  }
  c->announced.push_back(b);
  return 0;
}

int esvio_fe_set_next_batch(esvio_fe_handle c, double next_cur_time, const esvio_fe_event* left,
                            size_t nL, const esvio_fe_event* right, size_t nR, int space,
                            int pub_hint) {
  return set_next_batch_impl(c, next_cur_time, left, nL, right, nR, space, pub_hint, nullptr);
}

int esvio_fe_set_next_batch_mc(esvio_fe_handle c, double next_cur_time, const esvio_fe_event* left,
                               size_t nL, const esvio_fe_event* right, size_t nR, int space,
                               int pub_hint, const esvio_fe_motion* motion) {
  if (!motion) return ESVIO_FE_EINVAL;
  return set_next_batch_impl(c, next_cur_time, left, nL, right, nR, space, pub_hint, motion);
}

int esvio_fe_mem_alloc(int space, size_t bytes, void** out) {
  if (!out || (space != ESVIO_FE_HOST && space != ESVIO_FE_DEVICE)) return ESVIO_FE_EINVAL;
  *out = nullptr;
  const size_t b = std::max<size_t>(bytes, 16);
  const hipError_t e = space == ESVIO_FE_HOST ? hipHostMalloc(out, b, hipHostMallocDefault) : hipMalloc(out, b);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return ESVIO_FE_EHIP;
  }
  return 0;
}

int esvio_fe_mem_free(int space, void* p) {
  if (space != ESVIO_FE_HOST && space != ESVIO_FE_DEVICE) return ESVIO_FE_EINVAL;
  if (!p) return 0;
  return (space == ESVIO_FE_HOST ? hipHostFree(p) : hipFree(p)) == hipSuccess ? 0 : ESVIO_FE_EHIP;
}

int esvio_fe_mem_upload(void* dst_device, const void* src_host, size_t bytes) {
  if (bytes && (!dst_device || !src_host)) return ESVIO_FE_EINVAL;
  if (!bytes) return 0;
  return hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : ESVIO_FE_EHIP;
}

int esvio_fe_register_host_buffer(void* p, size_t bytes) {
  if (!p || !bytes) return ESVIO_FE_EINVAL;
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return ESVIO_FE_EHIP;
  }
  return 0;
}

int esvio_fe_unregister_host_buffer(void* p) {
  if (!p) return ESVIO_FE_EINVAL;
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return ESVIO_FE_EHIP;
  }
  return 0;
}

int esvio_fe_debug_inject(esvio_fe_handle c, int mask) {
  if (!c || mask < 0 || mask > 31) return ESVIO_FE_EINVAL;
  c->lim = esvio_fe_ctx::WaitLimits();
  c->lazy_late = (mask & ESVIO_FE_FAULT_LAZY_LATE) != 0;
  if (mask & ESVIO_FE_FAULT_TICKET) c->lim.ticket = 0;
  if (mask & ESVIO_FE_FAULT_LOOKBACK) c->lim.lookback = 0;
  if (mask & ESVIO_FE_FAULT_SPECULATIVE) c->lim.poll = 0;
  if (mask & ESVIO_FE_FAULT_CHAINED) c->lim.chain = 0;
  return 0;
}

int esvio_fe_debug_counters(esvio_fe_handle c, uint64_t out4[4]) {
  if (!c || !out4) return ESVIO_FE_EINVAL;
  out4[0] = c->n_spec_expired;
  out4[1] = c->n_chain_expired;
  out4[2] = c->tr_chain_launch;
  out4[3] = c->tr_chain_used;
  return 0;
}

int esvio_fe_plain_call_counters(esvio_fe_handle c, uint64_t out4[4]) {
  if (!c || !out4) return ESVIO_FE_EINVAL;
  out4[0] = c->n_plain_calls;
  out4[1] = c->n_cam_split;
  out4[2] = c->n_stereo_chained;
  out4[3] = c->n_chain_expired;
  return 0;
}

int esvio_fe_reserve(esvio_fe_handle c, size_t max_left, size_t max_right, int host_batches) {
  if (!c) return ESVIO_FE_EINVAL;
  const size_t n = max_left + max_right;
  if (n >= (1ull << 31)) return fail(c, ESVIO_FE_EINVAL, "batch too large");
  if (!c->inflight.empty() || !c->announced.empty())
    return fail(c, ESVIO_FE_EINVAL, "esvio_fe_reserve while batches are announced");
  HIPCHK(c, hipSetDevice(c->dev));
  // (growing frees the old buffers: nothing may still be using them)
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream3));
  HIPCHK(c, hipStreamSynchronize(c->stream4));
  if (c->stream6) HIPCHK(c, hipStreamSynchronize(c->stream6));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->tiled) {
    if (int rc = ensure_part_capacity(c, n, false)) return rc;
  } else if (int rc = ensure_sort_capacity(c, n)) {
    return rc;
  }
  for (int k = 0; k < kRightSlots; k++)
    if (int rc = ensure_arc_capacity(c, max_left, k)) return rc;
  if (host_batches) {
    if (int rc = ensure_event_capacity(c, n)) return rc;
    if (stager_enabled(c)) {
      if (int rc = stager_reserve(c, n)) return rc;
    } else {
      for (int lane = 0; lane < kPrefetchDepth; lane++)
        if (n > c->evp_cap[lane]) {
          if (c->d_evp[lane]) (void)hipFree(c->d_evp[lane]);
          c->d_evp[lane] = nullptr;
          c->evp_cap[lane] = 0;
          if (int rc = dev_alloc(c, &c->d_evp[lane], n + n / 4)) return rc;
          c->evp_cap[lane] = n + n / 4;
        }
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

const char* esvio_fe_latency_phase_name(int i) {
  static const char* nm[ESVIO_FE_LATENCY_PHASES] = {
      "enqueue sae+ts+pyr", "enqueue temporal LK", "wait temporal LK", "host filter", "host ransac",
      "host mask + enqueue detect/stereo", "wait stereo LK", "host tail",
      "pub: Event_setMask", "pub: points + k_select launch", "pub: speculative + chained LK launches",
      "pub: previous frame's right tail", "pub: stereo LK of new corners launch", "pub: next batch's prefetch launches",
      "check + take-up of a late batch", "sae: staging the host batch (part of enqueue sae+ts+pyr)"};
  return (i >= 0 && i < ESVIO_FE_LATENCY_PHASES) ? nm[i] : "";
}

int esvio_fe_latency_recent(esvio_fe_handle c, int back, esvio_fe_latency_call* out) {
  if (!c || !out || back < 0) return ESVIO_FE_EINVAL;
  const esvio_fe_ctx::Latency& L = c->lat;
  if ((uint64_t)back >= L.total_calls || back >= esvio_fe_ctx::Latency::kRecent) return ESVIO_FE_EINVAL;
  *out = L.recent[(L.total_calls - 1 - (uint64_t)back) % esvio_fe_ctx::Latency::kRecent];
  return 0;
}

int esvio_fe_latency_stats(esvio_fe_handle c, esvio_fe_latency* out, int reset) {
  if (!c || !out) return ESVIO_FE_EINVAL;
  const esvio_fe_ctx::Latency& L = c->lat;
  std::memset(out, 0, sizeof(*out));
  out->calls = L.calls;
  const size_t n = (size_t)std::min<uint64_t>(L.calls, esvio_fe_ctx::Latency::kRing);
  if (n) {
    std::vector<float> v(L.ring, L.ring + n);
    std::sort(v.begin(), v.end());
    out->mean_ms = L.sum_ms / (double)L.calls;
    out->p50_ms = v[n / 2];
    out->p99_ms = v[std::min(n - 1, (size_t)((double)n * 0.99))];
    out->max_ms = L.max_ms;
    out->max_call = L.max_call;
    out->max_published = L.max_pub;
    out->max_cpu_begin = L.max_cpu0;
    out->max_cpu_end = L.max_cpu1;
    out->max_invol_switches = L.max_nivcsw;
    out->max_allocs = L.max_allocs;
    std::memcpy(out->max_phase_ms, L.max_phase, sizeof(out->max_phase_ms));
  }
  out->allocs = L.allocs;
  out->invol_switches = L.nivcsw;
  if (reset) c->lat = esvio_fe_ctx::Latency();
  return 0;
}

int esvio_fe_ransac_tail(uint64_t out6[6], int reset) {
  if (!out6) return ESVIO_FE_EINVAL;
  host::ransac_tail(out6, reset != 0);
  return 0;
}

int esvio_fe_set_launch_thread(esvio_fe_handle c, int on) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  return launcher_set(c, on != 0);
}

int esvio_fe_set_profiling(esvio_fe_handle c, int on) {
  if (!c) return ESVIO_FE_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (int rc = launcher_drain(c)) return rc;
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
  if (int rc = launcher_drain(c)) return rc;
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
