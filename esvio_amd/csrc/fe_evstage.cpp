// fe_evstage.cpp — host-resident event batches (ESVIO_FE_HOST, what the reference's
// `const dvs_msgs::EventArray&` interface hands over: feature_tracker/src/feature_tracker.h:51-52)
// on their way to the device.
//
// A hipMemcpyAsync from pageable memory is not asynchronous: the runtime stages it through its own
// bounce buffers on the calling thread (measured: 15 GB/s, 0.35 ms for the 5.3 MB of a C3 batch —
// almost three times the rest of the frame).  Here the batch is cut into chunks of 256 KiB; helper
// threads copy them into pinned memory and, when the last chunk of a group is in, enqueue the
// group's DMA on a copy stream of its own (a DMA costs ~20 us whatever its size, so groups are
// large: the whole batch when it was announced with esvio_fe_set_next_batch — all of it then
// happens while the previous frames are tracked — and a few groups when the caller waits for it, so
// that group k's DMA runs under group k+1's memcpy); the compute streams only wait for an event.
// A source that already is pinned (hipHostMalloc / hipHostRegister) skips the memcpy: one DMA.
//
// Slots: one pinned + one device buffer per batch the handle knows about (announced, prefetched or
// being tracked).  A slot's device buffer is overwritten only after the kernels that read its
// previous batch (SAE update / Arc* on the prefetch stream, Arc* on the main stream) are done: the
// copy stream waits for the events recorded behind them.
#include "fe_internal.h"

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

namespace esvio {
namespace fe {

namespace {
constexpr size_t kChunkBytes = 256 * 1024;
constexpr int kMaxGroups = 8;

struct Group {
  size_t off = 0, len = 0;  // byte range of the slot's buffers that one DMA moves
  std::atomic<uint32_t> chunks_left{0};
  std::atomic<bool> dma_enq{false};  // the group's DMA has been enqueued (at least once)
};

// one memcpy into a slot's pinned buffer.  st: 0 not taken, 1 taken, 2 done — whoever moves it from 1 to 2
// accounts for the chunk (a helper, or the calling thread redoing the copy of a helper that went away)
struct Chunk {
  int group = 0;
  const uint8_t* src = nullptr;
  size_t off = 0, len = 0;  // byte offset inside the slot's buffers
  std::atomic<uint8_t> st{0};
};

struct Slot {
  std::unique_ptr<Chunk[]> chunk;
  size_t chunk_cap = 0;
  std::atomic<uint32_t> n_chunks{0};  // of the batch being staged
  std::atomic<uint32_t> next{0};      // the next chunk nobody has taken (>= n_chunks: none left)
  int n_groups = 0;
  uint8_t* pin = nullptr;
  EventRec* dev = nullptr;
  size_t cap = 0;  // events
  hipEvent_t copied = nullptr, pf_done = nullptr, main_done = nullptr, aux_done = nullptr;
  bool pf_rec = false, main_rec = false, aux_rec = false;
  // by_camera staging (a plain call that starts on the left camera while the right one is still on its
  // way): the left array is a DMA of its own, `copiedL` is recorded behind it
  hipEvent_t copiedL = nullptr;
  int left_group = -1;               // group that is the left array, -1: none (not by camera, or a pinned source)
  std::atomic<bool> left_enq{false}; // the left array's DMA is enqueued and copiedL recorded
  bool in_use = false;
  std::atomic<int> state{0};  // 0 idle, 1 staging, 2 every DMA enqueued and `copied` recorded, -1 failed
  Group grp[kMaxGroups];
  std::atomic<uint32_t> groups_left{0};
  std::atomic<int> busy{0};  // threads inside this slot's chunks (or about to take one)
};
}  // namespace

// Work is handed out without a lock (a thread that loses its CPU while holding one would stop every
// other thread for a scheduler quantum: profiles/r04_stall_forensics.md): a slot's chunks are numbered,
// `next` is an atomic counter.  Everything a thread does for a chunk may be done twice — the copy writes
// the same bytes, a group's DMA moves the same bytes, `copied` recorded again only moves the event
// later — so the calling thread, when it waits for a batch (stager_attach), finishes whatever a helper
// has started and not finished instead of waiting for that helper.
struct EventStager {
  esvio_fe_ctx* c = nullptr;
  hipStream_t stream = nullptr;
  Slot slot[kStageSlots];
  std::vector<std::thread> threads;
  std::mutex mu;  // (only the sleeping helpers' condition variable)
  std::condition_variable cv;
  std::atomic<int>* pending = nullptr;  // -> esvio_fe_ctx::stage_pending: chunks nobody has taken yet, all slots
  std::atomic<bool> stop{false};
  // (ESVIO_FE_TRACE) batches, bytes, time the calling thread waited for a batch's staging, chunks it
  // took itself meanwhile, batches a call left for the next one because they had not arrived yet
  uint64_t batches = 0, bytes_staged = 0, wait_ns = 0, caller_chunks = 0, skipped = 0, redone = 0;

  void enqueue_dma(Slot& s, Group& g) {
    if (hipMemcpyAsync((uint8_t*)s.dev + g.off, s.pin + g.off, g.len, hipMemcpyHostToDevice, stream) != hipSuccess) {
      (void)hipGetLastError();
      s.state.store(-1, std::memory_order_release);
    }
    if ((int)(&g - s.grp) == s.left_group) {
      if (hipEventRecord(s.copiedL, stream) != hipSuccess) s.state.store(-1, std::memory_order_release);
      s.left_enq.store(true, std::memory_order_release);
    }
    g.dma_enq.store(true, std::memory_order_release);
  }
  void finish_batch(Slot& s) {  // every group's DMA is enqueued: the event the compute streams wait for
    const bool ok = hipEventRecord(s.copied, stream) == hipSuccess;
    int expect = 1;
    if (!ok)
      s.state.store(-1, std::memory_order_release);
    else
      (void)s.state.compare_exchange_strong(expect, 2, std::memory_order_acq_rel);  // (2 already: a redone batch)
  }

  // the chunk is in the pinned buffer: its group's DMA if it was the group's last, the batch's event if
  // that was the last group
  void chunk_done(Slot& s, const Chunk& ch) {
    Group& g = s.grp[ch.group];
    if (g.chunks_left.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    enqueue_dma(s, g);
    if (s.groups_left.fetch_sub(1, std::memory_order_acq_rel) == 1) finish_batch(s);
  }

  void run_chunk(Slot& s, uint32_t idx) {
    Chunk& ch = s.chunk[idx];
    uint8_t q = 0;
    if (!ch.st.compare_exchange_strong(q, 1, std::memory_order_acq_rel)) return;  // (somebody else's already)
    std::memcpy(s.pin + ch.off, ch.src, ch.len);
    uint8_t taken = 1;
    if (ch.st.compare_exchange_strong(taken, 2, std::memory_order_acq_rel)) chunk_done(s, ch);
  }

  // any thread: take one chunk of any batch being staged, if there is one
  bool try_one() {
    if (pending->load(std::memory_order_acquire) <= 0) return false;
    for (Slot& s : slot) {
      if (s.state.load(std::memory_order_acquire) != 1) continue;
      s.busy.fetch_add(1, std::memory_order_acq_rel);  // (before the claim: stager_abandon / release wait for it)
      bool did = false;
      if (s.state.load(std::memory_order_acquire) == 1 &&
          s.next.load(std::memory_order_relaxed) < s.n_chunks.load(std::memory_order_acquire)) {
        const uint32_t idx = s.next.fetch_add(1, std::memory_order_acq_rel);
        if (idx < s.n_chunks.load(std::memory_order_acquire)) {
          pending->fetch_sub(1, std::memory_order_acq_rel);
          run_chunk(s, idx);
          did = true;
        }
      }
      s.busy.fetch_sub(1, std::memory_order_acq_rel);
      if (did) return true;
    }
    return false;
  }

  // The calling thread, waiting for slot k with nothing left to take: whatever some helper has started
  // and not finished is done again here — a chunk taken and not copied, a complete group whose DMA is
  // not enqueued, a complete batch whose event is not recorded.  Returns whether the batch is complete.
  void finish_for(int k) {
    Slot& s = slot[k];
    const uint32_t n = s.n_chunks.load(std::memory_order_acquire);
    for (uint32_t i = 0; i < n; i++) {
      Chunk& ch = s.chunk[i];
      if (ch.st.load(std::memory_order_acquire) != 1) continue;
      std::memcpy(s.pin + ch.off, ch.src, ch.len);
      uint8_t taken = 1;
      if (ch.st.compare_exchange_strong(taken, 2, std::memory_order_acq_rel)) {
        chunk_done(s, ch);
        redone++;
      }
    }
    bool all = true;
    for (int g = 0; g < s.n_groups; g++) {
      Group& gr = s.grp[g];
      if (gr.chunks_left.load(std::memory_order_acquire) != 0) {
        all = false;
        continue;
      }
      if (!gr.dma_enq.load(std::memory_order_acquire)) {  // (its last chunk's thread has not got to it)
        enqueue_dma(s, gr);
        redone++;
      }
    }
    if (all && s.state.load(std::memory_order_acquire) == 1) finish_batch(s);
  }

  void worker() {
    (void)hipSetDevice(c->dev);
    for (;;) {
      if (try_one()) continue;
      std::unique_lock<std::mutex> g(mu);
      cv.wait(g, [&] { return stop.load(std::memory_order_acquire) || pending->load(std::memory_order_acquire) > 0; });
      if (stop.load(std::memory_order_acquire)) return;
    }
  }
};

// what a spinning RANSAC helper does between jobs (host::ransac_pool_set_idle_work): one chunk
static bool stager_idle_work(void* arg) {
  EventStager* st = (EventStager*)arg;
  static thread_local int dev_set = -1;
  if (dev_set != st->c->dev) {  // (the helper enqueues a group's DMA when it copies the group's last chunk)
    (void)hipSetDevice(st->c->dev);
    dev_set = st->c->dev;
  }
  return st->try_one();
}

void stager_share_pool(esvio_fe_ctx* c) {
  if (!c->pool) return;
  if (c->stager)
    host::ransac_pool_set_idle_work(c->pool, &c->stage_pending, stager_idle_work, c->stager);
  else
    host::ransac_pool_set_idle_work(c->pool, nullptr, nullptr, nullptr);
}

int stager_threads_from_env() {
  if (const char* v = getenv("ESVIO_FE_STAGE_THREADS")) return std::max(0, std::min(8, atoi(v)));
  return 2;
}

static int stager_get(esvio_fe_ctx* c, EventStager** out) {
  if (!c->stager) {
    EventStager* st = new EventStager();
    st->c = c;
    st->pending = &c->stage_pending;
    c->stage_pending.store(0, std::memory_order_release);
    if (hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking) != hipSuccess) {
      delete st;
      return fail(c, ESVIO_FE_EHIP, "hipStreamCreate (event staging) failed");
    }
    for (Slot& s : st->slot)
      if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s.copiedL, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s.pf_done, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s.aux_done, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s.main_done, hipEventDisableTiming) != hipSuccess) {
        c->stager = st;
        stager_destroy(c);
        return fail(c, ESVIO_FE_EHIP, "hipEventCreate (event staging) failed");
      }
    for (int i = 0; i < c->stage_threads; i++) st->threads.emplace_back([st] { st->worker(); });
    c->stager = st;
    stager_share_pool(c);  // the RANSAC helpers, spinning between jobs anyway, take chunks as well
  }
  *out = c->stager;
  return 0;
}

static bool host_pointer_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (an ordinary malloc'ed pointer is "invalid value" to the runtime)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

static int slot_capacity(esvio_fe_ctx* c, Slot& s, size_t n) {
  if (n <= s.cap) return 0;
  // (the device buffer's previous readers: hipFree waits for the device)
  if (s.dev) (void)hipFree(s.dev);
  if (s.pin) (void)hipHostFree(s.pin);
  s.dev = nullptr;
  s.pin = nullptr;
  s.cap = 0;
  const size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
  if (int rc = dev_alloc(c, &s.dev, cap)) return rc;
  HIPCHK(c, hipHostMalloc((void**)&s.pin, cap * 16, hipHostMallocDefault));
  c->n_allocs++;
  s.cap = cap;
  return 0;
}

// esvio_fe_reserve: every idle slot sized for batches of n events
int stager_reserve(esvio_fe_ctx* c, size_t n_events) {
  if (!stager_enabled(c)) return 0;
  EventStager* st = nullptr;
  if (int rc = stager_get(c, &st)) return rc;
  for (Slot& s : st->slot)
    if (!s.in_use)
      if (int rc = slot_capacity(c, s, n_events)) return rc;
  return 0;
}

// Start staging [left; right] into a free slot: returns at once, the helpers do the work.
// dma_groups: DMAs the pageable part of the batch is moved with (1: the caller does not wait for it)
// by_camera: the left array and the right array are one DMA each (dma_groups is ignored), and the left one's
// completion can be waited for on its own (stager_attach_left)
int stager_begin(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL, const esvio_fe_event* right, size_t nR,
                 int dma_groups, int* slot_out, bool by_camera) {
  EventStager* st = nullptr;
  if (int rc = stager_get(c, &st)) return rc;
  int k = 0;
  // (a slot a straggling helper is still inside — its chunk was redone by the calling thread — is passed over)
  while (k < kStageSlots && (st->slot[k].in_use || st->slot[k].busy.load(std::memory_order_acquire) != 0)) k++;
  if (k == kStageSlots) return fail(c, ESVIO_FE_EINTERNAL, "no free event staging slot");
  Slot& s = st->slot[k];
  const size_t n = nL + nR;
  const bool pinL = nL && host_pointer_is_pinned(left), pinR = nR && host_pointer_is_pinned(right);
  if (int rc = slot_capacity(c, s, n)) return rc;
  // the DMA overwrites the slot's device buffer: behind the kernels that read its previous batch
  if (s.pf_rec) HIPCHK(c, hipStreamWaitEvent(st->stream, s.pf_done, 0));
  if (s.main_rec) HIPCHK(c, hipStreamWaitEvent(st->stream, s.main_done, 0));
  if (s.aux_rec) HIPCHK(c, hipStreamWaitEvent(st->stream, s.aux_done, 0));
  s.pf_rec = s.main_rec = s.aux_rec = false;
  s.left_group = -1;
  s.left_enq.store(false, std::memory_order_release);
  st->bytes_staged += n * 16;
  st->batches++;
  // a pinned source: one DMA straight from it, now (the slot is taken only once nothing below can fail
  // before its state is set)
  if (pinL) HIPCHK(c, hipMemcpyAsync(s.dev, left, nL * 16, hipMemcpyHostToDevice, st->stream));
  if (by_camera && (pinL || !nL)) {  // (the left array is on its way already, or there is none)
    HIPCHK(c, hipEventRecord(s.copiedL, st->stream));
    s.left_enq.store(true, std::memory_order_release);
  }
  if (pinR) HIPCHK(c, hipMemcpyAsync(s.dev + nL, right, nR * 16, hipMemcpyHostToDevice, st->stream));
  s.in_use = true;
  // the pageable part: destination byte range [lo, hi) of the slot's buffers
  const size_t lo = pinL ? nL * 16 : 0, hi = pinR ? nL * 16 : n * 16;
  int ng = 0;
  uint32_t nch = 0;
  s.n_chunks.store(0, std::memory_order_release);
  s.next.store(0, std::memory_order_release);
  if (hi > lo) {
    ng = std::max(1, std::min(dma_groups, kMaxGroups));
    const size_t chunks = (hi - lo + kChunkBytes - 1) / kChunkBytes;
    ng = (int)std::min<size_t>(ng, chunks);
    size_t per = (chunks + ng - 1) / ng;  // chunks per group
    ng = (int)((chunks + per - 1) / per);
    const bool cam_groups = by_camera && !pinL && !pinR && nL && nR;  // both arrays pageable: one group each
    if (cam_groups) ng = 2;
    if (by_camera && !pinL && nL) s.left_group = 0;  // (else: a single group of the right array, or of either one)
    if (by_camera && !cam_groups) {
      ng = 1;
      per = chunks;
    }
    if (chunks + 2 > s.chunk_cap) {  // (+1: a chunk never straddles the two source arrays)
      s.chunk_cap = chunks + 2 + chunks / 4;
      s.chunk.reset(new Chunk[s.chunk_cap]);
    }
    for (int g = 0; g < ng; g++) {
      size_t a = lo + (size_t)g * per * kChunkBytes, b = std::min(hi, a + per * kChunkBytes);
      if (cam_groups) {
        a = g ? nL * 16 : 0;
        b = g ? n * 16 : nL * 16;
      }
      s.grp[g].off = a;
      s.grp[g].len = b - a;
      s.grp[g].dma_enq.store(false, std::memory_order_relaxed);
      uint32_t cnt = 0;
      for (size_t o = a; o < b;) {
        // a chunk never straddles the boundary between the two source arrays
        const bool in_left = o < nL * 16;
        const size_t end = std::min(std::min(o + kChunkBytes, b), in_left ? nL * 16 : b);
        const uint8_t* src = in_left ? (const uint8_t*)left + o : (const uint8_t*)right + (o - nL * 16);
        Chunk& ch = s.chunk[nch];
        ch.group = g;
        ch.src = src;
        ch.off = o;
        ch.len = end - o;
        ch.st.store(0, std::memory_order_relaxed);
        nch++;
        cnt++;
        o = end;
      }
      s.grp[g].chunks_left.store(cnt, std::memory_order_relaxed);
    }
  }
  s.n_groups = ng;
  s.groups_left.store((uint32_t)ng, std::memory_order_relaxed);
  if (!nch && hipEventRecord(s.copied, st->stream) != hipSuccess) {
    s.in_use = false;
    return fail(c, ESVIO_FE_EHIP, "hipEventRecord (event staging) failed");
  }
  s.n_chunks.store(nch, std::memory_order_release);
  s.state.store(nch ? 1 : 2, std::memory_order_release);  // (open for taking)
  if (nch) {
    st->pending->fetch_add((int)nch, std::memory_order_acq_rel);
    { std::lock_guard<std::mutex> g(st->mu); }  // (a helper between its predicate and its sleep sees the count)
    st->cv.notify_all();
  }
  *slot_out = k;
  return 0;
}

bool stager_ready(esvio_fe_ctx* c, int slot) {
  const bool r = c->stager->slot[slot].state.load(std::memory_order_acquire) != 1;
  if (!r) c->stager->skipped++;
  return r;
}

// wait until every DMA of the slot's batch is enqueued (the calling thread takes chunks itself
// meanwhile), then make stream `s` wait for them; device pointers out
int stager_attach(esvio_fe_ctx* c, int slot, size_t nL, hipStream_t s, const EventRec** dL, const EventRec** dR) {
  EventStager* st = c->stager;
  Slot& sl = st->slot[slot];
  if (sl.state.load(std::memory_order_acquire) == 1) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned idle = 0;
    while (sl.state.load(std::memory_order_acquire) == 1) {
      if (st->try_one()) {
        st->caller_chunks++;
        idle = 0;
      } else if (++idle > 1000) {  // (~25 us with nothing to take: some helper has started something and is not finishing it)
        st->finish_for(slot);
        idle = 0;
      } else {
        __builtin_ia32_pause();
      }
    }
    st->wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  if (sl.state.load(std::memory_order_acquire) != 2) {
    // a DMA could not be enqueued: chunks of this slot may still be queued or being copied from the
    // caller's memory, which is the caller's again once this call has returned — finish them first
    stager_abandon(c, slot);
    return fail(c, ESVIO_FE_EHIP, "staging the event batch failed");
  }
  HIPCHK(c, hipStreamWaitEvent(s, sl.copied, 0));
  *dL = sl.dev;
  *dR = sl.dev + nL;
  return 0;
}

// by_camera staging: wait (taking chunks meanwhile) until the LEFT array's DMA is enqueued and make stream
// `s` wait for it; the right array may still be on its way (stager_attach follows for it)
int stager_attach_left(esvio_fe_ctx* c, int slot, hipStream_t s, const EventRec** dL) {
  EventStager* st = c->stager;
  Slot& sl = st->slot[slot];
  unsigned idle = 0;
  while (!sl.left_enq.load(std::memory_order_acquire) && sl.state.load(std::memory_order_acquire) == 1) {
    if (st->try_one()) {
      st->caller_chunks++;
      idle = 0;
    } else if (++idle > 1000) {
      st->finish_for(slot);
      idle = 0;
    } else {
      __builtin_ia32_pause();
    }
  }
  if (!sl.left_enq.load(std::memory_order_acquire) || sl.state.load(std::memory_order_acquire) < 0) {
    stager_abandon(c, slot);
    return fail(c, ESVIO_FE_EHIP, "staging the event batch failed");
  }
  HIPCHK(c, hipStreamWaitEvent(s, sl.copiedL, 0));
  *dL = sl.dev;
  return 0;
}

// give a slot back whose batch will not be tracked (a failed call): nothing of the caller's memory is
// read after this returns
void stager_abandon(esvio_fe_ctx* c, int slot) {
  if (slot < 0 || !c->stager) return;
  EventStager* st = c->stager;
  Slot& sl = st->slot[slot];
  {  // chunks of this slot nobody has taken: never will be
    const uint32_t n = sl.n_chunks.load(std::memory_order_acquire);
    const uint32_t nx = sl.next.exchange(n, std::memory_order_acq_rel);
    if (nx < n) st->pending->fetch_sub((int)(n - nx), std::memory_order_acq_rel);
    sl.state.store(-1, std::memory_order_release);  // (closed for taking)
  }
  while (sl.busy.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();  // chunks a thread holds: waited for
  (void)hipStreamSynchronize(st->stream);  // (DMAs straight from a pinned source, groups already enqueued)
  for (int g2 = 0; g2 < kMaxGroups; g2++) sl.grp[g2].chunks_left.store(0, std::memory_order_relaxed);
  sl.groups_left.store(0, std::memory_order_relaxed);
  sl.state.store(0, std::memory_order_release);
  sl.in_use = false;
}

// where the slot's batch will be on the device (known as soon as the slot is taken)
void stager_ptrs(esvio_fe_ctx* c, int slot, size_t nL, const EventRec** dL, const EventRec** dR) {
  Slot& sl = c->stager->slot[slot];
  *dL = sl.dev;
  *dR = sl.dev + nL;
}

// the kernels enqueued on `s` so far are the last ones on that stream to read the slot's device buffer
int stager_mark_read(esvio_fe_ctx* c, int slot, hipStream_t s, bool main_stream) {
  Slot& sl = c->stager->slot[slot];
  HIPCHK(c, hipEventRecord(main_stream ? sl.main_done : sl.pf_done, s));
  (main_stream ? sl.main_rec : sl.pf_rec) = true;
  return 0;
}
// ... and on a third stream (the stereo stream, where a plain call runs the right camera's update)
int stager_mark_read_aux(esvio_fe_ctx* c, int slot, hipStream_t s) {
  Slot& sl = c->stager->slot[slot];
  HIPCHK(c, hipEventRecord(sl.aux_done, s));
  sl.aux_rec = true;
  return 0;
}

// the batch has been tracked.  The caller's memory is free again when its track call returns: a
// DMA straight from a pinned source has to be over by then (it long is; the wait costs ~1 us)
int stager_release(esvio_fe_ctx* c, int slot) {
  if (slot < 0 || !c->stager) return 0;
  Slot& sl = c->stager->slot[slot];
  sl.in_use = false;
  if (sl.state.load(std::memory_order_acquire) == 2) HIPCHK(c, hipEventSynchronize(sl.copied));
  // (a helper whose chunk was redone may still be inside its own copy of it: it reads the caller's memory)
  while (sl.busy.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  return 0;
}

// every queued chunk done, every slot free (esvio_fe_reset)
void stager_drain(esvio_fe_ctx* c) {
  EventStager* st = c->stager;
  if (!st) return;
  for (int k = 0; k < kStageSlots; k++) {
    Slot& s = st->slot[k];
    unsigned idle = 0;
    while (s.state.load(std::memory_order_acquire) == 1) {
      if (st->try_one()) {
        idle = 0;
      } else if (++idle > 1000) {
        st->finish_for(k);
        idle = 0;
      } else {
        __builtin_ia32_pause();
      }
    }
    while (s.busy.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    s.in_use = false;
  }
  (void)hipStreamSynchronize(st->stream);
}

void stager_destroy(esvio_fe_ctx* c) {
  EventStager* st = c->stager;
  if (!st) return;
  if (c->pool) host::ransac_pool_set_idle_work(c->pool, nullptr, nullptr, nullptr);
  st->stop.store(true, std::memory_order_release);
  { std::lock_guard<std::mutex> g(st->mu); }
  st->cv.notify_all();
  for (std::thread& t : st->threads) t.join();
  if (c->trace && st->batches)
    fprintf(stderr, "[esvio_fe trace] host-event staging: %llu batches, %.1f MB, %d helper threads; calling thread waited "
            "%.3f ms per batch (took %llu chunks itself, redid %llu a helper had not finished), %llu take-ups postponed to the next call\n",
            (unsigned long long)st->batches, st->bytes_staged / 1e6, (int)st->threads.size(),
            st->wait_ns / 1e6 / st->batches, (unsigned long long)st->caller_chunks, (unsigned long long)st->redone,
            (unsigned long long)st->skipped);
  if (st->stream) {
    (void)hipStreamSynchronize(st->stream);
    (void)hipStreamDestroy(st->stream);
  }
  for (Slot& s : st->slot) {
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.copiedL) (void)hipEventDestroy(s.copiedL);
    if (s.aux_done) (void)hipEventDestroy(s.aux_done);
    if (s.pf_done) (void)hipEventDestroy(s.pf_done);
    if (s.main_done) (void)hipEventDestroy(s.main_done);
    if (s.dev) (void)hipFree(s.dev);
    if (s.pin) (void)hipHostFree(s.pin);
  }
  delete st;
  c->stager = nullptr;
}

}  // namespace fe
}  // namespace esvio
