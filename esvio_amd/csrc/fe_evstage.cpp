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
};

// one memcpy into a slot's pinned buffer.  st: 0 queued, 1 taken, 2 done — whoever moves it from 1 to 2
// accounts for the chunk (a helper, or the calling thread redoing the copy of a helper that went away)
struct Chunk {
  int group = 0;
  const uint8_t* src = nullptr;
  size_t off = 0, len = 0;  // byte offset inside the slot's buffers
  std::atomic<uint8_t> st{0};
};

struct Slot {
  std::unique_ptr<Chunk[]> chunk;
  size_t n_chunks = 0, chunk_cap = 0;
  uint8_t* pin = nullptr;
  EventRec* dev = nullptr;
  size_t cap = 0;  // events
  hipEvent_t copied = nullptr, pf_done = nullptr, main_done = nullptr;
  bool pf_rec = false, main_rec = false;
  bool in_use = false;
  std::atomic<int> state{0};  // 0 idle, 1 staging, 2 every DMA enqueued and `copied` recorded, -1 failed
  Group grp[kMaxGroups];
  std::atomic<uint32_t> groups_left{0};
  std::atomic<int> busy{0};  // chunks of this slot a thread has taken from the queue and not finished
};

struct Task {  // chunk `idx` of slot `slot`
  int slot;
  uint32_t idx;
};
}  // namespace

struct EventStager {
  esvio_fe_ctx* c = nullptr;
  hipStream_t stream = nullptr;
  Slot slot[kStageSlots];
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Task> q;
  bool stop = false;
  // (ESVIO_FE_TRACE) batches, bytes, time the calling thread waited for a batch's staging, chunks it
  // took itself meanwhile, batches a call left for the next one because they had not arrived yet
  uint64_t batches = 0, bytes_staged = 0, wait_ns = 0, caller_chunks = 0, skipped = 0, redone = 0;

  // the chunk is in the pinned buffer: its group's DMA if it was the group's last, the batch's event if
  // that was the last group
  void chunk_done(Slot& s, const Chunk& ch) {
    Group& g = s.grp[ch.group];
    if (g.chunks_left.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    if (hipMemcpyAsync((uint8_t*)s.dev + g.off, s.pin + g.off, g.len, hipMemcpyHostToDevice, stream) != hipSuccess) {
      (void)hipGetLastError();
      s.state.store(-1, std::memory_order_release);
    }
    if (s.groups_left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
      // the last group: every DMA of the batch has been enqueued (by whichever thread) before this record
      const bool ok = hipEventRecord(s.copied, stream) == hipSuccess;
      int expect = 1;
      if (!ok || !s.state.compare_exchange_strong(expect, 2, std::memory_order_acq_rel))
        s.state.store(-1, std::memory_order_release);
    }
  }

  void run_task(const Task& t) {
    Slot& s = slot[t.slot];
    Chunk& ch = s.chunk[t.idx];
    uint8_t q = 0;
    if (!ch.st.compare_exchange_strong(q, 1, std::memory_order_acq_rel)) return;  // (somebody else's already)
    std::memcpy(s.pin + ch.off, ch.src, ch.len);
    uint8_t taken = 1;
    if (ch.st.compare_exchange_strong(taken, 2, std::memory_order_acq_rel)) chunk_done(s, ch);
  }

  // The calling thread, with nothing left in the queue and the batch still not complete: a chunk some
  // helper has taken and not finished after `patience` polls is copied again here (same bytes to the same
  // place) and accounted for by whoever finishes first — a helper that has lost its CPU in the middle of
  // a 10 us memcpy must not cost the call a scheduler quantum (profiles/r04_stall_forensics.md).
  bool redo_stuck(int k) {
    Slot& s = slot[k];
    bool any = false;
    for (size_t i = 0; i < s.n_chunks; i++) {
      Chunk& ch = s.chunk[i];
      if (ch.st.load(std::memory_order_acquire) != 1) continue;
      std::memcpy(s.pin + ch.off, ch.src, ch.len);
      uint8_t taken = 1;
      if (ch.st.compare_exchange_strong(taken, 2, std::memory_order_acq_rel)) {
        chunk_done(s, ch);
        redone++;
        any = true;
      }
    }
    return any;
  }

  bool try_one() {  // any thread: take one chunk if there is one
    Task t;
    {
      std::lock_guard<std::mutex> g(mu);
      if (q.empty()) return false;
      t = q.front();
      q.pop_front();
      slot[t.slot].busy.fetch_add(1, std::memory_order_acq_rel);
    }
    run_task(t);
    slot[t.slot].busy.fetch_sub(1, std::memory_order_acq_rel);
    return true;
  }

  void worker() {
    (void)hipSetDevice(c->dev);
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return stop || !q.empty(); });
        if (stop && q.empty()) return;
        t = q.front();
        q.pop_front();
        slot[t.slot].busy.fetch_add(1, std::memory_order_acq_rel);
      }
      run_task(t);
      slot[t.slot].busy.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
};

int stager_threads_from_env() {
  if (const char* v = getenv("ESVIO_FE_STAGE_THREADS")) return std::max(0, std::min(8, atoi(v)));
  return 2;
}

static int stager_get(esvio_fe_ctx* c, EventStager** out) {
  if (!c->stager) {
    EventStager* st = new EventStager();
    st->c = c;
    if (hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking) != hipSuccess) {
      delete st;
      return fail(c, ESVIO_FE_EHIP, "hipStreamCreate (event staging) failed");
    }
    for (Slot& s : st->slot)
      if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s.pf_done, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s.main_done, hipEventDisableTiming) != hipSuccess) {
        c->stager = st;
        stager_destroy(c);
        return fail(c, ESVIO_FE_EHIP, "hipEventCreate (event staging) failed");
      }
    for (int i = 0; i < c->stage_threads; i++) st->threads.emplace_back([st] { st->worker(); });
    c->stager = st;
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
int stager_begin(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL, const esvio_fe_event* right, size_t nR,
                 int dma_groups, int* slot_out) {
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
  s.pf_rec = s.main_rec = false;
  st->bytes_staged += n * 16;
  st->batches++;
  // a pinned source: one DMA straight from it, now (the slot is taken only once nothing below can fail
  // before its state is set)
  if (pinL) HIPCHK(c, hipMemcpyAsync(s.dev, left, nL * 16, hipMemcpyHostToDevice, st->stream));
  if (pinR) HIPCHK(c, hipMemcpyAsync(s.dev + nL, right, nR * 16, hipMemcpyHostToDevice, st->stream));
  s.in_use = true;
  // the pageable part: destination byte range [lo, hi) of the slot's buffers
  const size_t lo = pinL ? nL * 16 : 0, hi = pinR ? nL * 16 : n * 16;
  std::vector<Task> tasks;
  int ng = 0;
  s.n_chunks = 0;
  if (hi > lo) {
    ng = std::max(1, std::min(dma_groups, kMaxGroups));
    const size_t chunks = (hi - lo + kChunkBytes - 1) / kChunkBytes;
    ng = (int)std::min<size_t>(ng, chunks);
    const size_t per = (chunks + ng - 1) / ng;  // chunks per group
    ng = (int)((chunks + per - 1) / per);
    if (chunks + 2 > s.chunk_cap) {  // (+1: a chunk never straddles the two source arrays)
      s.chunk_cap = chunks + 2 + chunks / 4;
      s.chunk.reset(new Chunk[s.chunk_cap]);
    }
    for (int g = 0; g < ng; g++) {
      const size_t a = lo + (size_t)g * per * kChunkBytes, b = std::min(hi, a + per * kChunkBytes);
      s.grp[g].off = a;
      s.grp[g].len = b - a;
      uint32_t cnt = 0;
      for (size_t o = a; o < b;) {
        // a chunk never straddles the boundary between the two source arrays
        const bool in_left = o < nL * 16;
        const size_t end = std::min(std::min(o + kChunkBytes, b), in_left ? nL * 16 : b);
        const uint8_t* src = in_left ? (const uint8_t*)left + o : (const uint8_t*)right + (o - nL * 16);
        Chunk& ch = s.chunk[s.n_chunks];
        ch.group = g;
        ch.src = src;
        ch.off = o;
        ch.len = end - o;
        ch.st.store(0, std::memory_order_relaxed);
        tasks.push_back(Task{k, (uint32_t)s.n_chunks});
        s.n_chunks++;
        cnt++;
        o = end;
      }
      s.grp[g].chunks_left.store(cnt, std::memory_order_relaxed);
    }
  }
  s.groups_left.store((uint32_t)ng, std::memory_order_relaxed);
  s.state.store(tasks.empty() ? 2 : 1, std::memory_order_release);
  if (tasks.empty() && hipEventRecord(s.copied, st->stream) != hipSuccess) {
    s.state.store(0, std::memory_order_release);
    s.in_use = false;
    return fail(c, ESVIO_FE_EHIP, "hipEventRecord (event staging) failed");
  }
  {
    std::lock_guard<std::mutex> g(st->mu);
    for (const Task& t : tasks) st->q.push_back(t);
  }
  st->cv.notify_all();
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
      } else if (++idle > 2000) {  // (~50 us with nothing to take: some helper holds a chunk and is not finishing it)
        st->redo_stuck(slot);
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

// give a slot back whose batch will not be tracked (a failed call): nothing of the caller's memory is
// read after this returns
void stager_abandon(esvio_fe_ctx* c, int slot) {
  if (slot < 0 || !c->stager) return;
  EventStager* st = c->stager;
  Slot& sl = st->slot[slot];
  {  // queued chunks of this slot: dropped
    std::lock_guard<std::mutex> g(st->mu);
    for (auto it = st->q.begin(); it != st->q.end();) it = it->slot == slot ? st->q.erase(it) : it + 1;
  }
  while (sl.busy.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();  // chunks a thread holds: waited for
  (void)hipStreamSynchronize(st->stream);  // (DMAs straight from a pinned source, groups already enqueued)
  for (int g2 = 0; g2 < kMaxGroups; g2++) sl.grp[g2].chunks_left.store(0, std::memory_order_relaxed);
  sl.groups_left.store(0, std::memory_order_relaxed);
  sl.state.store(0, std::memory_order_release);
  sl.in_use = false;
}

// the kernels enqueued on `s` so far are the last ones on that stream to read the slot's device buffer
int stager_mark_read(esvio_fe_ctx* c, int slot, hipStream_t s, bool main_stream) {
  Slot& sl = c->stager->slot[slot];
  HIPCHK(c, hipEventRecord(main_stream ? sl.main_done : sl.pf_done, s));
  (main_stream ? sl.main_rec : sl.pf_rec) = true;
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
  for (Slot& s : st->slot) {
    while (s.state.load(std::memory_order_acquire) == 1)
      if (!st->try_one()) __builtin_ia32_pause();
    s.in_use = false;
  }
  (void)hipStreamSynchronize(st->stream);
}

void stager_destroy(esvio_fe_ctx* c) {
  EventStager* st = c->stager;
  if (!st) return;
  {
    std::lock_guard<std::mutex> g(st->mu);
    st->stop = true;
  }
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
