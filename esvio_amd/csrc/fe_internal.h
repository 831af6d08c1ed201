// fe_internal.h — what fe_stages.cpp, fe_track.cpp, fe_image.cpp and fe_api.cpp share.
#pragma once
#include "fe_ctx.h"

namespace esvio {
namespace fe {

// ---------------------------------------------------------------- small templates / layouts
template <class T>
int dev_alloc(esvio_fe_ctx* c, T** p, size_t count) {
  HIPCHK(c, hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
  c->n_allocs++;
  return 0;
}

template <class T>
void reduce_vector(std::vector<T>& v, const std::vector<uint8_t>& status) {  // :56-81
  int j = 0;
  for (int i = 0; i < int(v.size()); i++)
    if (status[i]) v[j++] = v[i];
  v.resize(j);
}

inline uint8_t* px00(const PyrDesc& d) { return d.img[0] + (size_t)kPad * d.stride[0] + kPad; }

// device result block (and its pinned mirror): set 1 = temporal LK, then stereo LK of the temporal
// survivors; set 2 = stereo LK of the newly selected corners
struct ResLayout {
  size_t B1[2], C1[2], SA1[2], SB1[2], A[2], CNT, NEW, B2, C2, SA2, SB2, total;
};

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

// device-side address of a location inside the pinned block
template <typename T>
T* zdev(esvio_fe_ctx* c, T* host) {
  return (T*)(c->z_res + ((uint8_t*)host - c->h_pin));
}

// ---------------------------------------------------------------- fe_stages.cpp
int ensure_event_capacity(esvio_fe_ctx* c, size_t n);
int ensure_sort_capacity(esvio_fe_ctx* c, size_t n);
int ensure_part_capacity(esvio_fe_ctx* c, size_t n, bool mc);
int ensure_cand_capacity(esvio_fe_ctx* c, int set, size_t n);
int ensure_arc_capacity(esvio_fe_ctx* c, size_t n, int set);
int pyr_alloc(esvio_fe_ctx* c, PyrStore& ps, int w, int h, int max_level);
void pyr_build(esvio_fe_ctx* c, const PyrDesc* p, int nimg);
McParams make_mc_params(const esvio_fe_motion* m);
// arc_set >= 0: this batch's Arc* pass will run into candidate set arc_set; *arc_marked tells
// whether the update has set that set's touched flags on its way (else run_arc does it)
int sae_update(esvio_fe_ctx* c, const EventRec* evL, uint32_t nL, const EventRec* evR, uint32_t nR,
               const McParams* mc = nullptr, double2* L2 = nullptr, double2* S2 = nullptr, int arc_set = -1,
               bool* arc_marked = nullptr);
int stage_events(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL, const esvio_fe_event* right,
                 size_t nR, int space, const EventRec** dL, const EventRec** dR, int lane = -1);
void render_ts(esvio_fe_ctx* c, double t_sync, uint8_t* dst0, uint8_t* dst1, int ncam, const double2* S2);
void render_lk_images(esvio_fe_ctx* c, double t_sync, int cams, int slotL, int slotR, int rawbuf);
void render_and_build(esvio_fe_ctx* c, double t_sync, int slotL, int slotR, int rawbuf);
bool render_cam_ok(const esvio_fe_ctx* c);
void render_and_build_cam(esvio_fe_ctx* c, double t_sync, int cam, int slot);
const PyrDesc& raw_ts_desc(const esvio_fe_ctx* c, int cam);
LkArgs make_lk(const PyrDesc& P, const PyrDesc& N, const float2* prev, const float2* init, float2* next,
               uint8_t* status, const int* n_ptr, int n_max, int max_level, int max_count, double eps,
               int flags);
void run_lk(esvio_fe_ctx* c, const LkArgs& f, const LkArgs* b, float2* back_pts, uint8_t* back_status);
int copy_level0_out(esvio_fe_ctx* c, const PyrDesc& d, uint8_t* out);
int copy_level0_in(esvio_fe_ctx* c, const PyrDesc& d, const uint8_t* in);
bool in_border_event(const esvio_fe_ctx* c, const P2f& pt);
double pt_distance(const P2f& a, const P2f& b);
void event_set_mask(esvio_fe_ctx* c);
std::vector<P2f> undistorted_pts(const std::vector<P2f>& pts, const esvio_fe_camera& cam);
std::vector<P2f> pts_velocity_fn(std::vector<int>& ids, std::vector<P2f>& pts, IdMap& cur_id_pts,
                                 IdMap& prev_id_pts, double dt, size_t n_left);
void reject_with_f_event(esvio_fe_ctx* c);
ResLayout res_layout(size_t M);
Pin pin_of(esvio_fe_ctx* c, int set = 0);
size_t pin_bytes(const esvio_fe_config& cfg);
void clear_tracker_state(esvio_fe_ctx* c);
SelectArgs make_select_args(esvio_fe_ctx* c, int set, int max_corners, float2* out_pts, int out_base,
                            int32_t* out_idx);
size_t select_lds_bytes(const esvio_fe_ctx* c);
size_t select_tables_lds_bytes(const esvio_fe_ctx* c);
void run_compact(esvio_fe_ctx* c, uint32_t n_events, int set);
void run_select(esvio_fe_ctx* c, int set, int max_corners, float2* out_pts, int out_base, int32_t* out_idx,
                const uint32_t* mask_bits = nullptr, int* host_counts = nullptr, bool publish = false,
                const float2* stamp_pts = nullptr, int n_stamp = 0);
void run_arc(esvio_fe_ctx* c, const EventRec* ev, uint32_t n, const PyrDesc* ts, bool use_mask,
             bool want_flags, bool want_cand, int set, bool marked = false);
hipError_t sync_main(esvio_fe_ctx* c);
hipError_t sync_event(hipEvent_t ev);

// ---------------------------------------------------------------- fe_evstage.cpp (host-resident batches)
int stager_threads_from_env();  // ESVIO_FE_STAGE_THREADS (default 2; 0: plain hipMemcpyAsync from the caller's memory)
inline bool stager_enabled(const esvio_fe_ctx* c) { return c->stage_threads > 0; }
int stager_begin(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL, const esvio_fe_event* right, size_t nR,
                 int dma_groups, int* slot_out, bool by_camera = false);
int stager_attach_left(esvio_fe_ctx* c, int slot, hipStream_t s, const EventRec** dL);
int stager_mark_read_aux(esvio_fe_ctx* c, int slot, hipStream_t s);
bool stager_ready(esvio_fe_ctx* c, int slot);
int stager_attach(esvio_fe_ctx* c, int slot, size_t nL, hipStream_t s, const EventRec** dL, const EventRec** dR);
int stager_mark_read(esvio_fe_ctx* c, int slot, hipStream_t s, bool main_stream);
int stager_release(esvio_fe_ctx* c, int slot);
void stager_abandon(esvio_fe_ctx* c, int slot);
void stager_share_pool(esvio_fe_ctx* c);  // (re)connect the RANSAC helpers to the staging queue
int stager_reserve(esvio_fe_ctx* c, size_t n_events);
void stager_ptrs(esvio_fe_ctx* c, int slot, size_t nL, const EventRec** dL, const EventRec** dR);
void stager_drain(esvio_fe_ctx* c);
void stager_destroy(esvio_fe_ctx* c);
void stager_copy_bytes(uint8_t* dst, const uint8_t* src, size_t len);  // (test tap: a chunk's copy)
bool stager_pack_bytes(uint8_t* dst, const uint8_t* src, size_t len, uint32_t* base_sec);  // (test tap: a chunk packed to 8 B per event)
void stager_counters(esvio_fe_ctx* c, uint64_t out4[4]);  // {batches, bytes, chunks sent packed, chunks of packing batches sent raw}

// ---------------------------------------------------------------- fe_track.cpp
int prefetch_next(esvio_fe_ctx* c, bool wait_planes, bool must_take_first = false);
int launcher_set(esvio_fe_ctx* c, bool on);   // start / stop the launch thread
int stereo_split_prepare(esvio_fe_ctx* c);   // the second stereo stream, if this handle may use it (fe_track.cpp)
int launcher_drain(esvio_fe_ctx* c);          // every job handed over has been issued (returns the first job error)
int launcher_wait_lane(esvio_fe_ctx* c, int lane);  // ... the job that records this lane's events
void launcher_clear_error(esvio_fe_ctx* c);   // esvio_fe_reset: a failed job.s sticky error is dropped with the batches
int cancel_chain(esvio_fe_ctx* c);
int finalize_right(esvio_fe_ctx* c);
int finalize_pending(esvio_fe_ctx* c);
int track_event_impl(esvio_fe_ctx* c, double cur_time, const esvio_fe_event* left, size_t nL,
                     const esvio_fe_event* right, size_t nR, int space, bool PUB_THIS_FRAME,
                     const esvio_fe_motion* motion = nullptr);

// ---------------------------------------------------------------- fe_api.cpp (track exchange)
// pack the current frame's PointCloud records into the pinned send area (waits for the previous
// exchange to be through with it) and mark them "to be enqueued"; enqueue what is marked: upload +
// ncclAllGather + download on the exchange stream (no-op when nothing is marked)
int exchange_pack(esvio_fe_ctx* c);
int exchange_flush(esvio_fe_ctx* c);

// ---------------------------------------------------------------- fe_image.cpp
void euclid_halfwidths(double md, int8_t* hw /*[kMaxDiscR+1]*/, int* radius);
int gftt_run(esvio_fe_ctx* c, const PyrDesc& d, int max_corners, double quality, double min_distance,
             bool use_mask, float2* out_pts, int out_base, int* host_counts);
int track_image_impl(esvio_fe_ctx* c, double cur_time, const uint8_t* img_left, const uint8_t* img_right,
                     bool PUB_THIS_FRAME);

}  // namespace fe
}  // namespace esvio
