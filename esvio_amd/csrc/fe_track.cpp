// fe_track.cpp — FeatureTracker::trackEvent (reference: feature_tracker/src/feature_tracker.cpp:340-603)
// as a per-frame sequence over the stage launches of fe_stages.cpp, and the replay-mode scheduler
// around it: next-batch prefetch on a second stream, speculative and chained temporal LK, lazy
// right-camera tails.  Results are those of the plain sequence in every mode.
#include "fe_internal.h"

#include <sched.h>
#include <sys/resource.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace esvio {
namespace fe {

// ---------------------------------------------------------------- next-batch prefetch
// Enqueue the SAE update, time surfaces and pyramids of the batch announced with
// esvio_fe_set_next_batch on the second stream; they overlap the rest of the current frame (stereo
// LK, selection) and the host work between calls.  Waits for ev_planes_free (recorded on the main
// stream once the current frame has finished reading the SAE planes) when `wait_planes`; a frame
// that itself came from the prefetch stream and runs no Arc* on the main stream reads neither the
// planes nor the raw surfaces there, so the next prefetch only has to follow its own stream.
// With the caller's PUB hint the Arc* pass of the batch runs here too (into the other candidate
// set), which takes it off the main stream's per-frame chain.
// One batch's prefetch sequence = bookkeeping (which lane / pyramid slots / candidate set it gets, buffer
// capacities, where its events will be on the device: prefetch_next, always on the calling thread) + the
// HIP calls (prefetch_issue: waits, ~10 launches, event records — 35-45 us of host time).  With
// esvio_fe_set_launch_thread the second part is handed to the handle's launch thread.
struct PrefetchJob {
  Inflight b;
  bool wait_planes = false;
  int lks_wait = -1;  // copy of set 1 whose stereo LK (stream4) the batch's right pyramid slot has to follow, -1: none
};

static int prefetch_issue(esvio_fe_ctx* c, const PrefetchJob& j) {
  const Inflight& b = j.b;
  StreamScope on_prefetch_stream(c->stream2);
  if (j.wait_planes) HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_planes_free, 0));
  // the right-camera pyramid slot this batch gets may be the one an earlier frame's stereo LK
  // (stream4) still reads — in lazy mode nobody has waited for that launch yet
  if (j.lks_wait >= 0) HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_lks_done[j.lks_wait], 0));
  const EventRec *dL = b.dL, *dR = b.dR;
  if (b.stage >= 0) {
    if (int rc = stager_attach(c, b.stage, b.nL, c->stream2, &dL, &dR)) return rc;
  } else if (int rc = stage_events(c, b.left, b.nL, b.right, b.nR, b.space, &dL, &dR, b.lane)) {
    return rc;
  }
  bool arc_marked = false;
  McParams mcp;
  if (b.has_motion) mcp = make_mc_params(&b.motion);
  int rc = sae_update(c, dL, (uint32_t)b.nL, dR, (uint32_t)b.nR, b.has_motion ? &mcp : nullptr, nullptr, nullptr,
                      b.pub && b.nL ? b.cand : -1, &arc_marked);
  if (!rc) {
    render_and_build(c, b.time, b.slotL, b.slotR, b.raw);
    if (hipEventRecord(c->ev_lane_done[b.lane], c->stream2) != hipSuccess)
      rc = fail(c, ESVIO_FE_EHIP, "hipEventRecord failed");
    // (a sequence issued by the launch thread: the word a chained LK launch made before this point waits for)
    if (!rc && b.gate) launch_set_u32(c->stream2, c->d_lane_gate + b.lane, b.gate);
  }
  if (rc) return rc;
  if (b.arc_done) {  // (decided with the bookkeeping: the PUB hint says the frame will publish)
    const PyrDesc& ts = c->cfg.equalize ? c->raw[b.raw][0].d : c->pyr[b.slotL].d;
    run_arc(c, dL, (uint32_t)b.nL, &ts, false, false, true, b.cand, arc_marked);
    run_compact(c, (uint32_t)b.nL, b.cand);
    HIPCHK(c, hipEventRecord(c->ev_lane_arc[b.lane], c->stream2));
  }
  // (the last kernels on this stream that read the batch's events)
  if (b.stage >= 0)
    if (int rc2 = stager_mark_read(c, b.stage, c->stream2, false)) return rc2;
  // Nobody ever synchronises with this stream (the frames wait for its events): a query per sequence lets the runtime
  // retire the commands that have completed, a few at a time.  (Put in during the hunt for the 1.6-4.7 ms calls of a
  // cold process's first pass — profiles/r05_stall_hunt.txt; alone it did not remove them, the pool warm-up in
  // esvio_fe_create did; it was part of the configuration that then ran 30 cold processes without one, and stays.)
  (void)hipStreamQuery(c->stream2);
  (void)hipGetLastError();  // (hipErrorNotReady is the normal answer)
  return 0;
}

// ---- the launch thread: a single-producer single-consumer ring of jobs.  It spins while frames keep
// coming (a wake-up costs more than a job) and blocks after 2 ms without one.
struct Launcher {
  static constexpr int kRing = 8;  // (> kPrefetchDepth jobs can never be outstanding)
  esvio_fe_ctx* c = nullptr;
  std::thread th;
  PrefetchJob ring[kRing];
  alignas(64) std::atomic<uint64_t> submitted{0};
  alignas(64) std::atomic<uint64_t> done{0};
  // the first failure of a job: sticky until esvio_fe_reset (launcher_clear_error) or the thread is stopped — a failed
  // job has left its batch half applied, so every later wait on this launcher fails too, and nothing more is issued
  std::atomic<int> first_rc{0};
  std::string err_text;  // written by this thread before first_rc (release), read by the caller after it (acquire)
  std::atomic<bool> quit{false};
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> sleeping{false};

  void run() {
    (void)hipSetDevice(c->dev);
    fail_sink() = &err_text;
    auto idle_since = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
      const uint64_t d = done.load(std::memory_order_relaxed);
      if (submitted.load(std::memory_order_acquire) == d) {
        if (quit.load(std::memory_order_acquire)) return;
        cpu_relax();
        if ((++spins & 1023) == 0 &&
            std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(2000)) {
          // (seq_cst on both sides: either the submitter sees `sleeping` and notifies under the mutex, or the
          // predicate below sees its job — a store followed by a load of another word may otherwise pass
          // each other in the store buffer and both sides miss)
          std::unique_lock<std::mutex> lk(mu);
          sleeping.store(true, std::memory_order_seq_cst);
          cv.wait(lk, [&] {
            return submitted.load(std::memory_order_seq_cst) != done.load(std::memory_order_relaxed) ||
                   quit.load(std::memory_order_seq_cst);
          });
          sleeping.store(false, std::memory_order_seq_cst);
          idle_since = std::chrono::steady_clock::now();
        }
        continue;
      }
      const PrefetchJob& job = ring[d % kRing];
      if (first_rc.load(std::memory_order_relaxed) == 0) {  // (after a failure the jobs are only counted off)
        const int rc = prefetch_issue(c, job);
        if (rc) first_rc.store(rc, std::memory_order_release);
      }
      // a failed or counted-off job never runs the launch that writes its gate word: a chained LK already in flight
      // for it would sit out its whole bound (40 ms) on the stereo stream, and esvio_fe_reset / _destroy behind it.
      // Its waves are let go (the call that reads their results fails with this launcher's error anyway).
      if (first_rc.load(std::memory_order_relaxed) != 0 && job.b.gate) {
        launch_set_u32(c->stream2, c->d_lane_gate + job.b.lane, job.b.gate);
        (void)hipGetLastError();
      }
      done.store(d + 1, std::memory_order_release);
      idle_since = std::chrono::steady_clock::now();
    }
  }
};

// The second stereo stream (esvio_fe_ctx::stream6), made when it is going to be used — at the first announcement of
// a device-resident batch with the launch thread on and the float-order LK mode, or at creation when
// ESVIO_FE_STEREO_SPLIT=1 asks for it — never inside a track call (a stream's hardware queue comes with its first
// launch: 2 ms).  Normal priority: the runtime has four hardware queues per priority
// level and process, and with a third least-priority stream per handle the second handle of a process finds its
// prefetch stream on its stereo stream's queue (tools/queue_probe.hip); a handle that never splits keeps the four
// streams it had.
int stereo_split_prepare(esvio_fe_ctx* c) {
  if (c->stream6) return 0;
  const bool wanted = c->stereo_split_env >= 0 ? c->stereo_split_env != 0 : (c->cfg.lk_accum == 2 && c->launcher != nullptr);
  if (!wanted) return 0;
  HIPCHK(c, hipStreamCreateWithFlags(&c->stream6, hipStreamNonBlocking));
  launch_spin(c->stream6, 0);
  HIPCHK(c, hipStreamSynchronize(c->stream6));
  return 0;
}

int launcher_set(esvio_fe_ctx* c, bool on) {
  if (on == (c->launcher != nullptr)) return 0;
  // (job numbers are per launcher: a lane whose events were recorded before — by the calling thread, or by a
  // launcher that has been drained and is gone — has nothing outstanding)
  if (on) {
    for (uint64_t& j : c->lane_job) j = 0;
    c->lks_wait_job[0] = c->lks_wait_job[1] = 0;
    Launcher* l = new Launcher();
    l->c = c;
    l->th = std::thread([l] { l->run(); });
    c->launcher = l;
    return 0;
  }
  const int rc = launcher_drain(c);
  // (a failed job has left its batch half applied: the handle stays failed until esvio_fe_reset — switching the launch
  // thread off and on again must not make the error disappear)
  if (rc && !c->launch_err) c->launch_err = rc;
  for (uint64_t& j : c->lane_job) j = 0;
  c->lks_wait_job[0] = c->lks_wait_job[1] = 0;
  Launcher* l = c->launcher;
  l->quit.store(true, std::memory_order_seq_cst);
  { std::lock_guard<std::mutex> g(l->mu); }
  l->cv.notify_all();
  l->th.join();
  delete l;
  c->launcher = nullptr;
  return rc;
}

static int launcher_result(esvio_fe_ctx* c) {
  const int rc = c->launcher->first_rc.load(std::memory_order_acquire);
  if (rc) c->err = c->launcher->err_text.empty() ? "a prefetch job of the launch thread failed" : c->launcher->err_text;
  return rc;
}

// esvio_fe_reset: the handle starts from a clean slate, so does its launch thread (drained by the caller)
void launcher_clear_error(esvio_fe_ctx* c) {
  if (c->launcher) c->launcher->first_rc.store(0, std::memory_order_release);
  c->launch_err = 0;
}

int launcher_drain(esvio_fe_ctx* c) {
  Launcher* l = c->launcher;
  if (!l) return 0;
  while (l->done.load(std::memory_order_acquire) != l->submitted.load(std::memory_order_relaxed)) cpu_relax();
  return launcher_result(c);
}

int launcher_wait_lane(esvio_fe_ctx* c, int lane) {
  Launcher* l = c->launcher;
  if (!l) return 0;
  while (l->done.load(std::memory_order_acquire) < c->lane_job[lane]) cpu_relax();
  return launcher_result(c);
}

// Record "the stereo LK that writes copy `set` of set 1 is done" on the frame's stereo stream.  A prefetch job handed to
// the launch thread waits on this very event object for the record that existed when the job was made
// (PrefetchJob::lks_wait); a stream wait binds to the event's LATEST record at the time it is enqueued, so the job must
// have been issued before the event is recorded again — else a lagging launch thread would make the prefetch stream
// wait for THIS launch, which may sit behind a chained LK whose waves wait for the word that prefetch sequence writes:
// a cycle only the chained launch's 40 ms bound would break.  The job was submitted a call or more ago: normally no wait.
int record_lks_done(esvio_fe_ctx* c, int set) {
  if (Launcher* l = c->launcher) {
    while (l->done.load(std::memory_order_acquire) < c->lks_wait_job[set]) cpu_relax();
    if (int rc = launcher_result(c)) return rc;
  }
  HIPCHK(c, hipEventRecord(c->ev_lks_done[set], stereo_stream(c)));
  c->lks_last = set;
  return 0;
}

static int launcher_submit(esvio_fe_ctx* c, const PrefetchJob& j) {
  Launcher* l = c->launcher;
  const uint64_t s = l->submitted.load(std::memory_order_relaxed);
  l->ring[s % Launcher::kRing] = j;
  c->lane_job[j.b.lane] = s + 1;
  if (j.lks_wait >= 0) c->lks_wait_job[j.lks_wait] = s + 1;
  l->submitted.store(s + 1, std::memory_order_seq_cst);
  if (l->sleeping.load(std::memory_order_seq_cst)) {
    { std::lock_guard<std::mutex> g(l->mu); }
    l->cv.notify_all();
  }
  return 0;
}

int prefetch_next(esvio_fe_ctx* c, bool wait_planes, bool must_take_first) {
  int rc = 0;
  const bool only_first = must_take_first;
  // per-kernel timers and the trace's counters are the calling thread's
  const bool async_ok = c->launcher && !c->prof_on && !c->trace && !must_take_first;
  if (c->launcher && !async_ok)
    if ((rc = launcher_drain(c))) return rc;  // (this call's own HIP calls go behind the jobs handed over before)
  while (!rc && !c->announced.empty() && (int)c->inflight.size() < kPrefetchDepth) {
    PrefetchJob job;
    Inflight& b = job.b;
    static_cast<Batch&>(b) = c->announced.front();
    // host events still on their way through the staging slot: this call does not wait for them, the
    // batch is taken up by the next one (unless it is the very batch the caller is about to track)
    if (b.stage >= 0 && !must_take_first && !stager_ready(c, b.stage)) break;
    must_take_first = false;
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
    b.arc_done = b.pub && b.nL;
    job.wait_planes = wait_planes;
    wait_planes = false;  // later batches simply follow on the same stream
    job.lks_wait = c->lks_last;
    // where the events will be; a batch in pageable memory without the stager is copied by the issuing
    // thread into a lane buffer that may have to grow: that one stays on the calling thread
    bool async = async_ok;
    if (b.stage >= 0) {
      stager_ptrs(c, b.stage, b.nL, &b.dL, &b.dR);
    } else if (b.space == ESVIO_FE_DEVICE) {
      b.dL = (const EventRec*)b.left;
      b.dR = (const EventRec*)b.right;
    } else {
      async = false;
    }
    // capacities are the calling thread's business (growing frees buffers: nothing handed over may still
    // be about to use them)
    const size_t n = b.nL + b.nR;
    const bool grows = (c->tiled ? n > c->part_cap || (b.has_motion && !c->d_warp) : n > c->sort_cap) ||
                       (b.arc_done && b.nL > c->cand[b.cand].cap);
    if ((grows || !async) && c->launcher)
      if ((rc = launcher_drain(c))) break;
    {
      StreamScope on_prefetch_stream(c->stream2);  // (a grown buffer's initialisation precedes its first use there)
      if (c->tiled ? (rc = ensure_part_capacity(c, n, b.has_motion)) : (rc = ensure_sort_capacity(c, n))) break;
      if (b.arc_done && (rc = ensure_cand_capacity(c, b.cand, b.nL))) break;
    }
    if (async) {
      if (!++c->gate_seq) c->gate_seq = 1;
      b.gate = c->gate_seq;
      rc = launcher_submit(c, job);
    } else {
      rc = prefetch_issue(c, job);
      if (!rc && b.stage < 0 && b.space == ESVIO_FE_HOST) {  // (the lane buffer stage_events copied into)
        b.dL = c->d_evp[b.lane];
        b.dR = c->d_evp[b.lane] + b.nL;
      }
    }
    if (rc) break;
    c->inflight.push_back(b);
    c->announced.pop_front();
    // the late take-up of the batch about to be tracked takes that batch only: a second one applied to the
    // planes behind it would make this frame's PUB hint binding ("more than one batch in flight") although
    // the caller never had more than one announced ahead of its call
    if (only_first) break;
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
  if (int rc = launcher_wait_lane(c, nxt.lane)) return rc;
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
    f.poll_ticks = c->lim.poll;
  }
  // the frame after next, chained to this launch point by point (see esvio_fe_ctx::d_chain)
  const Inflight* nxt2 = nullptr;
  if (c->chain_enabled && c->waits_fit_chain && !nxt.pub && c->inflight.size() >= 2 && c->inflight[0].lane == nxt.lane &&
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
    // The frame after next's pyramids: built by a prefetch sequence the launch thread may still be issuing — waiting
    // for that here (to enqueue a stream wait on its event) put 25 us into this call whenever the thread was behind,
    // and the whole cycle into a slower regime (profiles/r05_slow_regime_timeline.txt).  The launch's waves wait on
    // the device instead, for the word the sequence's last launch writes.
    const bool gated = c->launcher && nxt2->gate != 0;
    if (!gated)
      if (int rc = launcher_wait_lane(c, nxt2->lane)) return rc;
    HIPCHK(c, hipStreamWaitEvent(c->stream4, c->ev_lane_done[nxt.lane], 0));
    if (!gated) HIPCHK(c, hipStreamWaitEvent(c->stream4, c->ev_lane_done[nxt2->lane], 0));
    uint8_t* zc = c->z_spec + c->spec_bytes;
    const PyrDesc& N2 = c->pyr[nxt2->slotL].d;
    LkArgs f2 = make_lk(N, N2, nullptr, nullptr, (float2*)zc, zc + M * 16, nullptr, n_max, 3, 30, 0.01, 0);
    LkArgs b2 = make_lk(N2, N, nullptr, nullptr, nullptr, nullptr, nullptr, n_max, 1, 30, 0.01,
                        ESVIO_FE_LK_USE_INITIAL_FLOW);
    f2.chain_in = c->d_chain;
    f2.chain_seq = c->chain_seq;
    f2.chain_ticks = c->lim.chain;
    f2.poll_err = (int*)(zc + M * 16 + 2 * stM);
    if (gated) {
      f2.gate_ptr = c->d_lane_gate + nxt2->lane;
      f2.gate_val = nxt2->gate;
    }
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
// One call of FeatureTracker::trackEvent (feature_tracker.cpp:340-603), phase by phase:
//   check()      what can refuse the call (a refused call leaves the handle as it was)
//   take_batch() the batch's SAE update / images / pyramids: taken from the prefetch stream, or enqueued now
//   temporal()   :405-440  temporal LK forward + backward (speculative / chained / plain launch), filters
//   survivors_stereo()  the stereo LK of every temporal survivor, on its own stream (+ the next frame's
//                speculative temporal LK on frames that publish nothing)
//   publish()    :442-469  rejectWithF_event, Event_setMask, k_select, stereo LK of the new corners
//   tails()      :463-603  new ids, undistortion / velocities, the right-camera tail, carried state
// The members are what the phases share; the handle (c) holds what outlives the call.
namespace {
using clk = std::chrono::steady_clock;

struct TrackCall {
  esvio_fe_ctx* c;
  const double time;
  const esvio_fe_event *left, *right;
  const size_t nL, nR;
  const int space;
  const bool PUB_THIS_FRAME;
  const esvio_fe_motion* motion;
  const int M;
  Pin pin{};
  const EventRec *dL = nullptr, *dR = nullptr;
  bool first = false;
  bool arc_done = false, arc_prefetched = false, arc_marked_main = false;
  int arc_lane = 0;
  bool main_reads_events = false;  // this call enqueues main-stream kernels that read the batch's events
  // what THIS frame enqueues on the main stream that reads the SAE planes / raw time surfaces: its
  // own SAE update + rendering unless prefetched, and Arc* unless that ran with the prefetch
  bool main_reads_planes = false;
  // the next frame's batch, if it is already in flight (two announced ahead), else once this
  // frame's early_work has put it there
  bool have_next = false, had_announced = false;
  Inflight next_b;
  bool early_done = false, defer_late = false;
  bool use_spec = false, use_chain = false, chain_covers_next = false;
  bool will_spec = false, detect = false;
  // a plain call (nothing announced, not lazy): Arc* runs on the prefetch stream beside the temporal LK,
  // and the stereo LK is launched WITH the temporal one, chained to it point by point on the device
  bool plain = false, arc_side = false, stereo_chained = false;
  bool split_right = false;  // ... and the right camera's update + image run on the stereo stream beside the left one's
  Pin pin_st{};               // where the frame's stereo LK results of the kept points land
  std::vector<int> surv_src;  // chained stereo: survivor i was the temporal launch's point surv_src[i]
  int n_surv = 0, n_kept = 0;
  clk::time_point tp;

  TrackCall(esvio_fe_ctx* ctx, double t, const esvio_fe_event* l, size_t nl, const esvio_fe_event* r, size_t nr,
            int sp, bool pub, const esvio_fe_motion* mo)
      : c(ctx), time(t), left(l), right(r), nL(nl), nR(nr), space(sp), PUB_THIS_FRAME(pub), motion(mo),
        M(ctx->cfg.max_cnt) {}

  void lap(int i) {  // (always on: esvio_fe_latency_stats names the slowest call's phases)
    const auto now = clk::now();
    const double ms = std::chrono::duration<double, std::milli>(now - tp).count();
    c->lat.cur_phase[i] += ms;
    if (c->trace) c->phase_ms[PUB_THIS_FRAME ? 1 : 0][i] += ms;
    tp = now;
  }

  bool same_motion(const Batch& a) const {  // (field by field: the struct has padding)
    if ((motion != nullptr) != a.has_motion) return false;
    if (!motion) return true;
    const esvio_fe_motion &x = *motion, &y = a.motion;
    bool eq = x.t1 == y.t1 && x.fx == y.fx && x.fy == y.fy && x.cx == y.cx && x.cy == y.cy;
    for (int i = 0; i < 3; i++)
      eq = eq && x.v[i] == y.v[i] && x.v_pre[i] == y.v_pre[i] && x.accel[i] == y.accel[i] && x.omega[i] == y.omega[i];
    return eq;
  }

  int check() {
    if (c->launch_err)
      return fail(c, c->launch_err, "a prefetch job of the launch thread failed earlier: esvio_fe_reset the handle");
    if (c->inflight.empty() && !c->announced.empty()) {
      // announced, but an earlier call left it where it was because its host events were still on
      // their way to the device: it is needed now
      const Batch& a = c->announced.front();
      if (left == a.left && nL == a.nL && right == a.right && nR == a.nR && space == a.space) {
        if (time != a.time || !same_motion(a))
          return fail(c, ESVIO_FE_EINVAL, "batch differs from the one given to esvio_fe_set_next_batch");
        // the take-up runs on the prefetch stream and waits for ev_planes_free; early_work only records
        // that event for batches announced DURING the previous call, so with announce-after-return
        // (track(k) returns, set_next_batch(k+1), track(k+1)) it would be stale: whatever frame k left
        // on the main stream (a first, unpublished or lazily returned frame is not synchronised) still
        // reads the planes and the single partition / CLAHE scratch — order the take-up behind it
        HIPCHK(c, hipEventRecord(c->ev_planes_free, c->stream));
        if (int rc = prefetch_next(c, true, true)) return rc;
      }
    }
    if (!c->inflight.empty()) {
      const Inflight& b = c->inflight.front();
      if (left != b.left || nL != b.nL || right != b.right || nR != b.nR || space != b.space ||
          time != b.time || !same_motion(b))
        return fail(c, ESVIO_FE_EINVAL, "batch differs from the one given to esvio_fe_set_next_batch");
      if (PUB_THIS_FRAME && !b.arc_done && c->inflight.size() > 1)
        return fail(c, ESVIO_FE_EINVAL,
                    "PUB hint was 0 for a published frame and a later batch is already applied to "
                    "the SAE: with more than one batch announced the hint must be exact");
    }
    return 0;
  }

  int take_batch() {
    first = !c->have_img;
    c->cur_stage = -1;
    if (!c->inflight.empty()) {
      // this batch was announced with esvio_fe_set_next_batch and its SAE update, images and
      // pyramids were enqueued on the prefetch stream during an earlier call
      const Inflight b = c->inflight.front();
      c->inflight.pop_front();
      // (launch thread: the lane's events must have been recorded by its job before anything waits on them)
      if (int rc = launcher_wait_lane(c, b.lane)) return rc;
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_lane_done[b.lane], 0));
      c->tr_lane = b.lane;
      c->cur_stage = b.stage;
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
      main_reads_events = true;
      const bool staged = space == ESVIO_FE_HOST && stager_enabled(c) && (nL + nR) * 16 >= (256u << 10);
      // A plain call (the reference's pattern) in the plain configuration is split by camera: the temporal
      // LK needs the LEFT image only, so the left camera's update and image go first on the main stream and
      // the right camera's follow on the stereo stream, under the temporal LK — and, for a batch in
      // pageable memory, the left array is a DMA of its own: the left chain starts while the right array is
      // still crossing PCIe.
      const bool split = c->cam_split_enabled && c->tiled && render_cam_ok(c) && !motion && !c->ext_sae_pending && !c->ext_right_pending &&
                         nL && nR && c->announced.empty() && c->inflight.empty() && !c->lazy_new && !c->chain_valid && !c->spec_valid &&
                         (staged || space == ESVIO_FE_DEVICE);
      if (staged) {
        // not announced: the helpers and this thread stage the chunks together, the DMA of group k runs
        // under the memcpy of group k+1
        const auto ts0 = clk::now();
        if (int rc = stager_begin(c, left, nL, right, nR, 4, &c->cur_stage, split)) return rc;
        if (split) {
          if (int rc = stager_attach_left(c, c->cur_stage, c->stream, &dL)) return rc;
        } else if (int rc = stager_attach(c, c->cur_stage, nL, c->stream, &dL, &dR)) {
          return rc;
        }
        c->lat.cur_phase[15] = std::chrono::duration<double, std::milli>(clk::now() - ts0).count();  // (part of phase 0)
      } else if (int rc = stage_events(c, left, nL, right, nR, space, &dL, &dR)) {
        return rc;
      }
      // SAEtoTimeSurface_left/right(cur_time) (:367-368) -> cur images; slot rotation replaces the
      // cv::Mat header swaps of :390-403,:585.  Left slots 0..2: {prev, cur, free}.
      int sl = 0;
      while (!first && (sl == c->slot_prevL || sl == c->slot_curL)) sl++;
      const int new_curL = sl;
      if (split) {
        if (int rc = sae_update(c, dL, (uint32_t)nL, nullptr, 0, nullptr, nullptr, nullptr,
                                PUB_THIS_FRAME ? c->cand_cur : -1, &arc_marked_main))
          return rc;
        HIPCHK(c, hipEventRecord(c->ev_sae_left, c->stream));
        c->slot_curL = new_curL;
        c->slot_curR = c->slot_curR == kLeftSlots ? kLeftSlots + 1 : kLeftSlots;
        c->raw_cur = (c->raw_cur + 1) % kRightSlots;
        render_and_build_cam(c, c->cur_time, 0, c->slot_curL);
        HIPCHK(c, hipEventRecord(c->ev_imgs_ready, c->stream));
        // the right camera, on the stereo stream
        if (staged) {
          const auto ts1 = clk::now();
          const EventRec* dl2 = nullptr;
          if (int rc = stager_attach(c, c->cur_stage, nL, stereo_stream(c), &dl2, &dR)) return rc;
          c->lat.cur_phase[15] += std::chrono::duration<double, std::milli>(clk::now() - ts1).count();
        }
        {
          StreamScope on_stereo_stream(stereo_stream(c));
          HIPCHK(c, hipStreamWaitEvent(stereo_stream(c), c->ev_sae_left, 0));  // (the partition scratch is the left chain's until then)
          if (int rc = sae_update(c, nullptr, 0, dR, (uint32_t)nR)) return rc;
          render_and_build_cam(c, c->cur_time, 1, c->slot_curR);
          HIPCHK(c, hipEventRecord(c->ev_right_ready, stereo_stream(c)));
        }
        split_right = true;
        c->n_cam_split++;
      } else {
      // createSAE_left / createSAE_right loops (:356-362), or their motion-compensated forms (:627-641)
      if (c->ext_sae_pending) {
        // esvio_fe_sae_slice_commit has put this batch into the planes already (its SAE update ran
        // time-sliced over several GPUs); the events are still needed below for Arc*
        if (motion) return fail(c, ESVIO_FE_EINVAL, "time-sliced SAE update has no motion-compensated form");
        c->ext_sae_pending = false;
      } else if (motion) {
        const McParams mc = make_mc_params(motion);
        if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR, &mc)) return rc;
      } else if (int rc = sae_update(c, dL, (uint32_t)nL, dR, (uint32_t)nR, nullptr, nullptr, nullptr,
                                     PUB_THIS_FRAME ? c->cand_cur : -1, &arc_marked_main)) {
        return rc;
      }
      c->slot_curL = new_curL;
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
      HIPCHK(c, hipEventRecord(c->ev_imgs_ready, c->stream));
      }
    }
    have_next = !c->inflight.empty();
    next_b = have_next ? c->inflight.front() : Inflight();
    had_announced = !c->announced.empty();
    if (first) c->slot_prevL = c->slot_curL;  // prev_img_left = cur_img_left = img_left (:391)
    c->have_img = true;
    main_reads_planes = !c->cur_prefetched;
    plain = !c->cur_prefetched && !had_announced && c->inflight.empty() && !c->lazy_new && !c->chain_valid &&
            !c->spec_valid;
    if (plain) c->n_plain_calls++;
    pin_st = pin;
    c->cur_pts.clear();
    c->cur_right_pts.clear();
    return 0;
  }

  const Inflight* next_batch() {
    if (!have_next && had_announced) {
      if (!c->inflight.empty()) {
        next_b = c->inflight.front();
        have_next = true;
      }
    }
    return have_next ? &next_b : nullptr;
  }

  // Arc* for every left event does not depend on the tracks: on published frames it is enqueued
  // now (behind the temporal LK) without the blocked-pixel mask, so it runs under the host-side
  // filtering / RANSAC / Event_setMask; the mask becomes k_select's initial bitmap.  After it
  // nothing of this frame reads the planes on the main stream, so the announced next batch is
  // started on the prefetch stream.
  int early_work() {
    if (early_done) return 0;
    early_done = true;
    if (PUB_THIS_FRAME && !arc_done) {
      if (int rc = ensure_arc_capacity(c, nL, c->cand_cur)) return rc;
      const PyrDesc ts = raw_ts_desc(c, 0);
      if (plain) {
        // on the prefetch stream (idle in a plain call), behind the frame's images: it runs beside the
        // temporal LK instead of behind it, and the host's wait for that LK no longer includes it
        StreamScope on_side_stream(c->stream2);
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_imgs_ready, 0));
        run_arc(c, dL, (uint32_t)nL, &ts, false, false, true, c->cand_cur, arc_marked_main);
        run_compact(c, (uint32_t)nL, c->cand_cur);
        HIPCHK(c, hipEventRecord(c->ev_arc_side, c->stream2));
        arc_side = true;
      } else {
        run_arc(c, dL, (uint32_t)nL, &ts, false, false, true, c->cand_cur, arc_marked_main);
        run_compact(c, (uint32_t)nL, c->cand_cur);
      }
      arc_done = true;
      main_reads_planes = true;
      main_reads_events = true;
    }
    if (!had_announced) return 0;
    if (main_reads_planes) HIPCHK(c, hipEventRecord(c->ev_planes_free, c->stream));
    return prefetch_next(c, main_reads_planes);
  }

  // which launch this frame's temporal LK results come from, and when the announced batch's prefetch
  // is enqueued
  int plan_temporal() {
    // a speculative launch of this very temporal LK may have been made by the previous call
    if (c->spec_valid) {
      c->spec_valid = false;
      use_spec = c->cur_prefetched && (int)c->prev_pts.size() == c->spec_n;
      if (!use_spec) HIPCHK(c, hipStreamSynchronize(c->stream3));
    }
    // ... or a chained one by the call before that; if it was made for the NEXT frame, this frame
    // is the one in between: it must publish nothing and track with the speculative results
    if (c->chain_valid && c->chain_for == c->frame_no) {
      use_chain = !use_spec && c->cur_prefetched && c->chain_map_ok &&
                  c->chain_map.size() == c->prev_pts.size();
      if (!use_chain)
        if (int rc = cancel_chain(c)) return rc;
      c->chain_valid = false;
    } else if (c->chain_valid && (c->chain_for != c->frame_no + 1 || PUB_THIS_FRAME || !use_spec)) {
      if (int rc = cancel_chain(c)) return rc;
    }
    chain_covers_next = c->chain_valid;  // (then: for frame_no + 1)
    const bool early_results = use_spec || use_chain;
    // When to enqueue the ~12 launches of the announced batch's prefetch (early_work):
    //  * before the wait for this frame's temporal LK when that is a speculative / chained launch
    //    still running and the frame publishes nothing: the host would only wait there;
    //  * late — a published frame whose successor is already in flight: after everything else of the
    //    frame, while the corner selection runs (RANSAC + mask + selection sit behind the temporal
    //    LK wait, so nothing is put in front of them);
    //  * else right away (Arc* still has to run on the main stream, or nothing to overlap with).
    // (With esvio_fe_set_launch_thread the HIP calls of an announced batch's prefetch are issued by the
    // handle's launch thread instead — Launcher above; the decision WHEN stays here, on the caller.)
    const bool before_sync = early_results && !PUB_THIS_FRAME;
    defer_late = !(PUB_THIS_FRAME && !arc_done) && have_next && !before_sync;
    return 0;
  }

  int temporal() {  // :405-440
    if (int rc = plan_temporal()) return rc;
    const esvio_fe_config& cfg = c->cfg;
    // rejectWithF_event lifts prev_pts and cur_pts through the camera model before its RANSAC; prev_pts are known
    // now, and the host is about to wait for the temporal LK anyway: their half of the lifts is done under that wait
    // (a published frame's RANSAC sits on the replay cycle's critical chain)
    c->pre_lift_valid = false;
    auto pre_lift = [&]() {
      const size_t n = c->prev_pts.size();
      if (!PUB_THIS_FRAME || !cfg.f_ransac || n < 8 || c->pre_lift_valid) return;
      c->pre_lx.resize(n);
      c->pre_ly.resize(n);
      host::lift_projective_batch(cfg.cam[0], &c->prev_pts[0].x, (int)n, c->pre_lx.data(), c->pre_ly.data());
      c->pre_lift_valid = true;
    };
    const PyrDesc& prevL = c->pyr[c->slot_prevL].d;
    const PyrDesc& curL = c->pyr[c->slot_curL].d;
    if (c->prev_pts.size() > 0) {
      const int n = (int)c->prev_pts.size();
      const uint8_t *t_stA, *t_stB;
      const P2f *t_ptsB, *t_ptsC;
      bool spec_ok = false;
      if (use_spec) {
        if (!defer_late)
          if (int rc = early_work()) return rc;
        pre_lift();
        lap(1);
        if (int rc = exchange_flush(c)) return rc;  // (host time that would be spent waiting)
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
        if (!spec_ok) {
          c->n_spec_expired++;
          // its waves have ended without results; nothing of it may still be running when the same
          // points go through the plain launch below
          HIPCHK(c, hipStreamSynchronize(c->stream3));
        }
      }
      std::vector<P2f> g_ptsB, g_ptsC;
      std::vector<uint8_t> g_stA, g_stB;
      if (use_chain) {
        if (!defer_late)
          if (int rc = early_work()) return rc;
        pre_lift();
        lap(1);
        if (int rc = exchange_flush(c)) return rc;
        HIPCHK(c, sync_event(c->ev_chain_done));
        lap(2);
        const size_t Mx = (size_t)std::max(M, 1), stM = (Mx + 63) / 64 * 64;
        const uint8_t* hc = c->h_spec + c->spec_bytes;
        int* wait_expired = (int*)(hc + Mx * 16 + 2 * stM);
        spec_ok = *wait_expired == 0;
        *wait_expired = 0;
        c->tr_chain_used += spec_ok;
        if (!spec_ok) c->n_chain_expired++;
        if (c->trace && spec_ok) trace_chain_intervals();
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
        stereo_chained = plain && c->chain_enabled && c->waits_fit_chain;
        if (stereo_chained) {
          c->n_stereo_chained++;
          c->chain_seq = (c->chain_seq + 1) & 0x3fffffffu;
          if (!c->chain_seq) c->chain_seq = 1;
          f.chain_out = c->d_chain;
          f.chain_seq = c->chain_seq;
        }
        run_lk(c, f, cfg.flow_back ? &b : nullptr, zdev(c, pin.ptsC), zdev(c, pin.stB));
        if (stereo_chained) {
          // cv::calcOpticalFlowPyrLK(curL, curR, cur_pts, ...) (:490) and its reverse (:495) start from the
          // temporal FORWARD results, point by point: launched now, on the stereo stream, every wave
          // waiting (bounded) for its point's forward result instead of for the host's round trip; the
          // points the host's filters drop below are simply not gathered.  Results: the other copy of
          // set 1 (the temporal launch is still writing this one).
          const PyrDesc& curR = c->pyr[c->slot_curR].d;
          pin_st = pin_of(c, c->res_set ^ 1);
          const size_t Mx = (size_t)std::max(M, 1), stM = (Mx + 63) / 64 * 64;
          LkArgs f2 = make_lk(curL, curR, nullptr, nullptr, zdev(c, pin_st.ptsB), zdev(c, pin_st.stA), nullptr, n, 3, 30,
                              0.01, 0);
          LkArgs b2 = make_lk(curR, curL, nullptr, nullptr, nullptr, nullptr, nullptr, n, 3, 30, 0.01, 0);
          f2.chain_in = c->d_chain;
          f2.chain_seq = c->chain_seq;
          f2.chain_ticks = c->lim.chain;
          f2.poll_err = (int*)(c->z_spec + c->spec_bytes + Mx * 16 + 2 * stM);
          StreamScope on_stereo_stream(stereo_stream(c));
          HIPCHK(c, hipStreamWaitEvent(stereo_stream(c), c->ev_imgs_ready, 0));
          run_lk(c, f2, cfg.flow_back ? &b2 : nullptr, zdev(c, pin_st.ptsC), zdev(c, pin_st.stB));
          if (int rc = record_lks_done(c, c->res_set)) return rc;
        }
        if (int rc = early_work()) return rc;
        if (int rc = exchange_flush(c)) return rc;
        pre_lift();
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
      if (stereo_chained) {
        surv_src.clear();
        for (int i = 0; i < n; i++)
          if (status[i]) surv_src.push_back(i);
      }
      if (c->pre_lift_valid) {
        reduce_vector(c->pre_lx, status);
        reduce_vector(c->pre_ly, status);
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
    return 0;
  }

  // (trace only) device-side intervals of the published frame's chain, from timing events
  void trace_chain_intervals() {
    float a = 0, b = 0, d = 0;
    if (hipEventElapsedTime(&a, c->ev_dbg_sel_start, c->ev_sel_host) == hipSuccess &&
        hipEventElapsedTime(&b, c->ev_sel_host, c->ev_spec_done) == hipSuccess &&
        hipEventElapsedTime(&d, c->ev_sel_host, c->ev_chain_done) == hipSuccess) {
      c->tr_gpu_sel += a;
      c->tr_gpu_spec += b;
      c->tr_gpu_chain += d;
      float e2 = 0;
      if (c->tr_lane >= 0 && hipEventElapsedTime(&e2, c->ev_sel_host, c->ev_lane_done[c->tr_lane]) == hipSuccess)
        c->tr_gpu_pyr += e2;
      else
        (void)hipGetLastError();
      c->tr_host_chain += std::chrono::duration<double, std::milli>(clk::now() - c->tr_sel_launch).count();
      c->tr_gpu_n++;
    } else {
      (void)hipGetLastError();
    }
  }

  int upload_kept() {
    if (!will_spec || !n_kept) return 0;
    // pin.news is a single buffer and the previous published frame's lazy stereo LK of its new
    // corners reads its points from there (z_new + its n_kept) in place.  Up to ~1000 points every
    // wave of that launch is resident from the start and has loaded its point long before the host
    // gets here (it had to wait for this frame's temporal LK first); a larger launch runs in
    // several rounds of blocks, so its completion is awaited before the slots are overwritten.
    if (c->pend.active && M > 1024) HIPCHK(c, sync_event(c->ev_lknew_done));
    std::memcpy(pin.news, c->cur_pts.data(), (size_t)n_kept * 8);  // read in place by the LK
    return 0;
  }

  // ---- speculative stereo LK of every temporal survivor (a superset of the points that survive
  // rejectWithF_event / Event_setMask): per-point results do not depend on the other points, so
  // this is exactly cv::calcOpticalFlowPyrLK(curL, curR, cur_pts, ...) (:490) and its reverse (:495)
  // for the kept points — launched now so that it overlaps the host-side RANSAC + mask.
  int survivors_stereo() {
    const esvio_fe_config& cfg = c->cfg;
    const PyrDesc& curL = c->pyr[c->slot_curL].d;
    const PyrDesc& curR = c->pyr[c->slot_curR].d;
    n_surv = (int)c->cur_pts.size();
    c->src_idx.resize(n_surv);
    for (int i = 0; i < n_surv; i++) c->src_idx[i] = stereo_chained ? surv_src[i] : i;
    lap(3);
    n_kept = n_surv;
    // the next batch's pyramids are in flight on the prefetch stream: next frame's temporal LK can be
    // launched as soon as this frame's points are final
    will_spec = (have_next || had_announced) && c->waits_fit_spec;
    if (!PUB_THIS_FRAME) {  // (ahead of the stereo LK so that the two launches overlap)
      if (int rc = upload_kept()) return rc;
      // (c->chain_valid here: the next frame's temporal LK is already running, chained to this one's)
      if (will_spec && n_kept && !c->chain_valid)
        if (const Inflight* nb = next_batch())
          if (int rc = enqueue_spec_temporal(c, *nb, n_kept, false)) return rc;
    }
    if (n_surv && !stereo_chained) {
      std::memcpy(pin.A, c->cur_pts.data(), (size_t)n_surv * 8);
      LkArgs f = make_lk(curL, curR, zdev(c, pin.A), nullptr, zdev(c, pin.ptsB), zdev(c, pin.stA), nullptr, n_surv, 3, 30,
                         0.01, 0);
      LkArgs b = make_lk(curR, curL, nullptr, nullptr, nullptr, nullptr, nullptr, n_surv, 3, 30, 0.01, 0);
      {
        // on its own stream.  Its inputs are complete without a device-side wait: the host has just
        // read this frame's temporal LK results, and that launch ran behind the frame's pyramids.
        StreamScope on_stereo_stream(stereo_stream(c));
        run_lk(c, f, cfg.flow_back ? &b : nullptr, zdev(c, pin.ptsC), zdev(c, pin.stB));
        if (int rc = record_lks_done(c, c->res_set)) return rc;
      }
    }
    return 0;
  }

  int publish() {  // :442-469
    const esvio_fe_config& cfg = c->cfg;
    const PyrDesc& curL = c->pyr[c->slot_curL].d;
    const PyrDesc& curR = c->pyr[c->slot_curR].d;
    if (PUB_THIS_FRAME) {
      if (cfg.f_ransac) reject_with_f_event(c);
      lap(4);
      auto tq = clk::now();
      auto sub = [&](int i) {
        const auto now = clk::now();
        const double ms = std::chrono::duration<double, std::milli>(now - tq).count();
        c->lat.cur_phase[8 + i] += ms;
        if (c->trace) c->pub_ms[i] += ms;
        tq = now;
      };
      event_set_mask(c);
      sub(0);
      // (a plain call's Arc* pass ran on the prefetch stream: the selection, and whatever the next call
      // puts on the main stream, follow it)
      if (arc_side) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_arc_side, 0));
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
        // (idle time: k_select is running — but only if the previous frame's stereo LK is over; else tails() runs
        // that frame's right-camera tail, after this frame's own left-camera bookkeeping: waiting HERE was 37-77 us
        // of every published call once the unpublished frame's call had become short, bench.py --call-phases)
        if (!c->pend_right.active || c->pend_right.left.empty() ||
            (!c->lazy_late && hipEventQuery(c->ev_lks_done[c->pend_right.set]) != hipErrorNotReady))
          if (int rc = finalize_right(c)) return rc;
        (void)hipGetLastError();
        sub(3);
        // stereo LK of the new corners only (count known on the device)
        if (split_right) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_right_ready, 0));  // (the right image)
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
    return 0;
  }

  int tails() {  // :463-603
    const esvio_fe_config& cfg = c->cfg;
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
    auto tt = clk::now();
    auto tail_lap = [&](int i) {
      if (!c->trace) return;
      const auto now = clk::now();
      c->tail_ms[PUB_THIS_FRAME ? 1 : 0][i] += std::chrono::duration<double, std::milli>(now - tt).count();
      tt = now;
    };
    c->cur_un_pts = undistorted_pts(c->cur_pts, cfg.cam[0]);  // :470-473
    c->pts_velocity = pts_velocity_fn(c->ids, c->cur_un_pts, c->cur_un_pts_map, c->prev_un_pts_map,
                                      c->cur_time - c->prev_time, c->cur_pts.size());
    lap(7);
    tail_lap(0);
    const bool lazy = c->lazy_new && detect;      // leave this frame's new corners to the next call
    const bool defer_right = c->lazy_new && !PUB_THIS_FRAME;  // ... or its whole right-camera tail
    // The previous published frame's new corners are completed here — unless this frame itself returns without its
    // right-camera tail and their stereo LK (main stream, behind that frame's selection: ~80 us) is still running:
    // the next call completes them before it runs this frame's tail, in the same order (their map entries are what
    // that tail's velocities read).  Waiting here held the call 40 us on every cycle whose speculative temporal LK
    // had finished early, and the published call that follows starts when this one ends (bench.py --call-phases,
    // profiles/r06_replay_cycle_floor.md).
    bool pending_later = false;
    if (defer_right && c->pend.active && !c->pend_right.active) {
      pending_later = c->lazy_late || hipEventQuery(c->ev_lknew_done) == hipErrorNotReady;
      (void)hipGetLastError();
    }
    if (!pending_later) {
      if (int rc = finalize_pending(c)) return rc;  // (the previous published frame's new corners,
      if (int rc = finalize_right(c)) return rc;    //  or the previous unpublished frame's whole tail)
    }
    tail_lap(1);
    if (defer_right) {
      // (returns with the stereo LK in flight)
    } else {
      if (!lazy) HIPCHK(c, sync_main(c));  // stereo LK results of the new corners
      if (n_surv || stereo_chained) HIPCHK(c, sync_event(c->ev_lks_done[c->res_set]));  // ... of the kept points
      if (stereo_chained) {
        const size_t Mx = (size_t)std::max(M, 1), stM = (Mx + 63) / 64 * 64;
        int* wait_expired = (int*)(c->h_spec + c->spec_bytes + Mx * 16 + 2 * stM);
        if (*wait_expired != 0) {
          // a wave of the chained stereo launch gave up waiting for its point: the same launch the plain
          // way, from the temporal forward results (still in this frame's copy of set 1; a point the
          // filters dropped is tracked for nothing, as in the chained launch)
          *wait_expired = 0;
          c->n_chain_expired++;
          const int n = (int)surv_src.size() ? surv_src.back() + 1 : 0;
          if (n) {
            const PyrDesc& curL = c->pyr[c->slot_curL].d;
            const PyrDesc& curR = c->pyr[c->slot_curR].d;
            LkArgs f2 = make_lk(curL, curR, zdev(c, pin.ptsB), nullptr, zdev(c, pin_st.ptsB), zdev(c, pin_st.stA), nullptr, n,
                                3, 30, 0.01, 0);
            LkArgs b2 = make_lk(curR, curL, nullptr, nullptr, nullptr, nullptr, nullptr, n, 3, 30, 0.01, 0);
            StreamScope on_stereo_stream(stereo_stream(c));
            run_lk(c, f2, cfg.flow_back ? &b2 : nullptr, zdev(c, pin_st.ptsC), zdev(c, pin_st.stB));
            HIPCHK(c, hipStreamSynchronize(stereo_stream(c)));
          }
        }
      }
    }
    lap(6);
    tt = clk::now();
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
      right_tail(c, pin_st, c->cur_pts.data(), c->ids.data(), c->src_idx.data(),
                 lazy ? n_kept : (int)c->cur_pts.size(), n_kept, c->cur_time - c->prev_time,
                 c->cur_pts.size());
    }
    tail_lap(2);
    c->slot_prevL = c->slot_curL;  // prev_img_left = cur_img_left (:585)
    c->prev_pts = c->cur_pts;
    c->prev_un_pts_map.swap(c->cur_un_pts_map);
    c->prev_time = c->cur_time;
    c->spec_n = (int)c->prev_pts.size();
    lap(7);
    if (c->cur_stage >= 0) {  // the staging slot of this batch's host events is free for another batch
      if (main_reads_events)
        if (int rc = stager_mark_read(c, c->cur_stage, c->stream, true)) return rc;
      if (arc_side)  // (its k_arc_ev read the batch's events on the prefetch stream)
        if (int rc = stager_mark_read(c, c->cur_stage, c->stream2, false)) return rc;
      if (split_right)  // (the right camera's update read its events on the stereo stream)
        if (int rc = stager_mark_read_aux(c, c->cur_stage, stereo_stream(c))) return rc;
      if (int rc = stager_release(c, c->cur_stage)) return rc;
      c->cur_stage = -1;
    }
    // whatever comes next on the main stream — the taps, the next call's update, which shares the partition
    // scratch — follows the right camera's chain
    if (split_right) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_right_ready, 0));
    c->phase_frames++;
    c->phase_count[PUB_THIS_FRAME ? 1 : 0]++;
    c->tr_surv += (uint64_t)n_surv;
    if (c->prof_on) resolve_profile(c);
    // esvio_fe_set_auto_exchange: what an earlier published frame left to enqueue goes out now at the
    // latest (normally it went out above, where this call waited for its temporal LK anyway); this
    // frame's records, if it publishes, are packed now and enqueued by the next call
    if (int rc = exchange_flush(c)) return rc;
    if (c->x_auto && c->x_comm && PUB_THIS_FRAME)
      if (int rc = exchange_pack(c)) return rc;
    tail_lap(3);
    return 0;
  }
};
}  // namespace

namespace {
long thread_invol_switches() {
#if defined(__linux__)
  struct rusage ru;
  if (getrusage(RUSAGE_THREAD, &ru) == 0) return ru.ru_nivcsw;
#endif
  return 0;
}

// the call's wall time, phases, CPUs, preemptions and allocations into the handle's latency record
void latency_commit(esvio_fe_ctx* c, bool pub, clk::time_point t0, int cpu0, long sw0, uint64_t allocs0) {
  esvio_fe_ctx::Latency& L = c->lat;
  const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
  const long sw = thread_invol_switches() - sw0;
  const uint64_t al = c->n_allocs - allocs0;
  {
    if (!L.have_t0) {
      L.have_t0 = true;
      L.t0 = t0;
    }
    esvio_fe_latency_call& r = L.recent[L.total_calls % esvio_fe_ctx::Latency::kRecent];
    r.call = L.total_calls++;
    r.published = pub ? 1 : 0;
    r.begin_ms = std::chrono::duration<double, std::milli>(t0 - L.t0).count();
    r.ms = ms;
    std::memcpy(r.phase_ms, L.cur_phase, sizeof(r.phase_ms));
  }
  L.ring[L.calls % esvio_fe_ctx::Latency::kRing] = (float)ms;
  L.sum_ms += ms;
  L.allocs += al;
  L.nivcsw += (uint64_t)std::max(0l, sw);
  if (ms > L.max_ms) {
    L.max_ms = ms;
    L.max_call = L.calls;
    L.max_pub = pub;
    L.max_cpu0 = cpu0;
    L.max_cpu1 = sched_getcpu();
    L.max_nivcsw = sw;
    L.max_allocs = (long)al;
    std::memcpy(L.max_phase, L.cur_phase, sizeof(L.max_phase));
  }
  if (c->slow_call_ms > 0 && ms > c->slow_call_ms) {
    fprintf(stderr, "[esvio_fe slow call] #%llu (frame %llu, %s) %.3f ms; cpu %d -> %d, %ld involuntary switches, %llu allocations;",
            (unsigned long long)L.calls, (unsigned long long)c->frame_no, pub ? "published" : "unpublished", ms, cpu0,
            sched_getcpu(), sw, (unsigned long long)al);
    for (int i = 0; i < ESVIO_FE_LATENCY_PHASES; i++)
      if (L.cur_phase[i] > 0.02) fprintf(stderr, " %s=%.3f", esvio_fe_latency_phase_name(i), L.cur_phase[i]);
    fprintf(stderr, "\n");
  }
  L.calls++;
}
}  // namespace

int track_event_impl(esvio_fe_ctx* c, double _cur_time, const esvio_fe_event* left, size_t nL,
                     const esvio_fe_event* right, size_t nR, int space, bool PUB_THIS_FRAME,
                     const esvio_fe_motion* motion) {
  TrackCall t(c, _cur_time, left, nL, right, nR, space, PUB_THIS_FRAME, motion);
  const auto t0 = clk::now();
  const int cpu0 = sched_getcpu();
  const long sw0 = thread_invol_switches();
  const uint64_t allocs0 = c->n_allocs;
  std::memset(c->lat.cur_phase, 0, sizeof(c->lat.cur_phase));
  if (int rc = t.check()) return rc;
  // set 1 alternates between its two copies: the previous frame's stereo LK may still be in flight
  // (lazy mode, pend_right) while this frame's kernels are enqueued
  c->res_set ^= 1;
  // (measured, KERNELS.md: the second stereo stream pays where the device chain is the bound — the float-order LK's
  // launches, 1.6x the exact mode's, with the launch thread issuing the prefetch and the events already on the
  // device; the exact mode's cycle gets 5 % longer with it, and so does a caller that issues every launch itself
  // or hands over host batches)
  c->stereo_split = c->stream6 && (c->stereo_split_env >= 0 ? c->stereo_split_env != 0
                                                            : (c->cfg.lk_accum == 2 && c->launcher != nullptr && c->stager == nullptr));
  c->stereo_unpub = !PUB_THIS_FRAME && c->stereo_split;
  c->frame_no++;
  t.pin = pin_of(c, c->res_set);
  if (PUB_THIS_FRAME && c->pool) host::ransac_pool_wake(c->pool);
  c->cur_time = _cur_time;
  t.tp = clk::now();
  c->lat.cur_phase[14] = std::chrono::duration<double, std::milli>(t.tp - t0).count();
  int rc = t.take_batch();
  if (!rc) {
    t.lap(0);
    rc = t.temporal();
  }
  if (!rc) rc = t.survivors_stereo();
  if (!rc) rc = t.publish();
  if (!rc) rc = t.tails();
  if (rc && c->cur_stage >= 0) {
    // a failed call: its host batch's staging slot goes back (else eight such failures use them all
    // up).  Kernels of this call may still read the slot's device buffer, and chunks of the batch may
    // still be on their way out of the caller's memory: both are waited for
    (void)launcher_drain(c);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamSynchronize(c->stream4);
    if (c->stream6) (void)hipStreamSynchronize(c->stream6);
    (void)hipGetLastError();
    stager_abandon(c, c->cur_stage);
    c->cur_stage = -1;
  }
  if (!rc) latency_commit(c, PUB_THIS_FRAME, t0, cpu0, sw0, allocs0);
  return rc;
}


}  // namespace fe
}  // namespace esvio
