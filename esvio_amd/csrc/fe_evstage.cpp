// fe_evstage.cpp — host-resident event batches (ESVIO_FE_HOST, what the reference's
// `const dvs_msgs::EventArray&` interface hands over: feature_tracker/src/feature_tracker.h:51-52)
// on their way to the device.
//
// A hipMemcpyAsync from pageable memory is not asynchronous: the runtime stages it through its own
// bounce buffers on the calling thread (measured: 15 GB/s, 0.35 ms for the 5.3 MB of a C3 batch —
// almost three times the rest of the frame).  Here the batch is cut into chunks of 256 KiB (64 KiB when
// the calling thread waits for it); helper threads copy them into pinned memory with streaming stores and,
// when the last chunk of a group is in, send the group to the device on a copy stream of its own: an
// announced batch (esvio_fe_set_next_batch: all of it happens while the previous frames are tracked) as ONE
// DMA; a plain call's batch in a few groups, each pulled out of the pinned buffer by a small kernel
// (k_stage_pull) under the memcpy of the next one — lower latency than a copy engine at these sizes, and
// several DMAs in flight at once make the runtime bring up further copy engines in the middle of a stream
// (7-9 ms inside whichever hipMemcpyAsync does it, measured round 4; with more engines up the same calls
// then run 0.08 ms slower).  The compute streams only wait for an event.
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

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <emmintrin.h>
#define ESVIO_STAGE_NT 1
#endif

namespace esvio {
namespace fe {

namespace {
// pageable source -> pinned chunk.  The destination is read next by the DMA engine, never by this core: streaming
// stores (no read-for-ownership of the destination lines, no cache filled with them); dst is 16-byte aligned (the
// slot is page-aligned, offsets are multiples of the 16-byte event record)
inline void stage_copy(uint8_t* dst, const uint8_t* src, size_t len) {
#ifdef ESVIO_STAGE_NT
  if (((uintptr_t)dst & 15u) == 0) {
    size_t i = 0;
    for (; i + 64 <= len; i += 64) {
      const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16)),
                    c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
      _mm_stream_si128((__m128i*)(dst + i), a);
      _mm_stream_si128((__m128i*)(dst + i + 16), b);
      _mm_stream_si128((__m128i*)(dst + i + 32), c);
      _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    if (i < len) std::memcpy(dst + i, src + i, len - i);
    _mm_sfence();  // (before the chunk is marked done and its group's DMA enqueued)
    return;
  }
#endif
  std::memcpy(dst, src, len);
}

// The same chunk as 8 bytes per event, when its events allow it: x | y << 16 as they are, nsec (30 bits) | polarity
// (!= 0, as every kernel reads it) << 30 | (sec - base) << 31 with base = the chunk's first event's second — so a
// chunk may cross ONE second boundary forwards; anything else (stamps going backwards across a second, nsec >= 2^30)
// and the caller copies the chunk raw instead.  Returns whether the packed form was written (dst: the first len / 2
// bytes of the chunk's place in the pinned buffer); *base_sec: the second the offsets count from.
inline bool stage_pack(uint8_t* dst, const uint8_t* src, size_t len, uint32_t* base_sec) {
  const size_t n = len / 16;
  if (!n || (len & 15u) || ((uintptr_t)dst & 15u)) return false;
  uint64_t a0;
  std::memcpy(&a0, src, 8);
  const uint32_t base = (uint32_t)(a0 >> 32);
  uint64_t bad = 0;
  size_t i = 0;
  auto pack1 = [&](const uint8_t* p) -> uint64_t {
    uint64_t a, b;  // a = x | y << 16 | sec << 32;  b = nsec | polarity byte << 32 | padding
    std::memcpy(&a, p, 8);
    std::memcpy(&b, p + 8, 8);
    const uint32_t nsec = (uint32_t)b, off = (uint32_t)(a >> 32) - base;
    bad |= (uint64_t)(nsec >> 30) | (uint64_t)(off >> 1);
    const uint32_t hi = nsec | (((b >> 32) & 0xffu) ? 1u << 30 : 0u) | (off << 31);
    return (a & 0xffffffffull) | ((uint64_t)hi << 32);
  };
#ifdef ESVIO_STAGE_NT
  for (; i + 2 <= n; i += 2) {
    const uint64_t lo = pack1(src + 16 * i), hi = pack1(src + 16 * i + 16);
    _mm_stream_si128((__m128i*)(dst + 8 * i), _mm_set_epi64x((long long)hi, (long long)lo));
  }
#endif
  for (; i < n; i++) {
    const uint64_t v = pack1(src + 16 * i);
    std::memcpy(dst + 8 * i, &v, 8);
  }
#ifdef ESVIO_STAGE_NT
  _mm_sfence();
#endif
  *base_sec = base;
  return bad == 0;
}
constexpr size_t kChunkBytes = 256 * 1024;
constexpr int kMaxGroups = 8;

struct Group {
  size_t off = 0, len = 0;  // byte range of the slot's buffers that one DMA moves
  uint32_t first_chunk = 0; // its first chunk (Slot::chunk / Slot::desc index)
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
  int n_left_groups = 0;             // groups [0, n_left_groups) are the left array's (0: not by camera, or a pinned source)
  std::atomic<bool> left_enq{false}; // the left array's DMA is enqueued and copiedL recorded
  uint32_t n_left_chunks = 0;        // chunks [0, n_left_chunks) are the left array's (by_camera)
  bool pull = false;                 // the groups go to the device by k_stage_pull instead of by DMA (a plain call's batch)
  // ... with the chunks packed to 8 bytes per event where their events allow it (by_camera staging: chunks of one
  // size, a group = whole chunks of one array): one {base second, packed?} pair per chunk, pinned, read by the kernel
  bool pack = false;
  uint32_t pack_epc = 0;             // events per chunk
  uint32_t* desc = nullptr;          // [2 * desc_cap]
  size_t desc_cap = 0;
  std::chrono::steady_clock::time_point t_begin;  // (trace) when stager_begin opened the batch for taking
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
  // A thread's first HIP calls cost milliseconds (measured: 7 ms inside the call whose DMA a RANSAC helper
  // was the first to enqueue): every thread that may enqueue a DMA does one dummy copy + event record on the
  // copy stream before it takes its first chunk
  hipEvent_t gate_ev[4] = {};                    // behind the last four DMAs
  std::atomic<uint32_t> gate_n{0}, gate_rec[4];  // DMAs issued; gate_rec[t & 3] == t + 1: DMA t's event is recorded
  std::atomic<uint64_t> gate_expired{0};
  uint8_t* warm_pin = nullptr;
  void* warm_dev = nullptr;
  hipEvent_t warm_ev = nullptr;
  void warm_thread() {
    (void)hipSetDevice(c->dev);
    if (warm_pin && warm_dev && hipMemcpyAsync(warm_dev, warm_pin, 16, hipMemcpyHostToDevice, stream) != hipSuccess) (void)hipGetLastError();
    if (warm_ev && hipEventRecord(warm_ev, stream) != hipSuccess) (void)hipGetLastError();
  }
  std::atomic<uint64_t> chunks_packed{0}, chunks_raw{0};  // chunks of packing slots that went out packed / raw (esvio_fe_staging_counters)
  // (trace) time inside stage_chunk, chunks, from a batch's opening to its chunk 0 being taken / its left array's last
  // group / its last group being on its way
  std::atomic<uint64_t> chunk_ns{0}, chunk_cnt{0}, first_take_ns{0}, left_sent_ns{0}, all_sent_ns{0};
  bool pack_enabled = true;  // (ESVIO_FE_STAGE_PACK=0, A/B and tests: every chunk raw)
  uint64_t begin_ns = 0, left_ns = 0, left_calls = 0, pin_ns = 0;  // (trace) by-camera staging: stager_begin, stager_attach_left

  // one DMA on the copy stream.  Never more than two in flight: the third makes the runtime bring up another
  // copy engine inside this very call (7-9 ms) — the stream runs them one after the other anyway, so DMA
  // number t waits (bounded) for number t - 2 to be over
  bool dma(void* dst, const void* src, size_t len) {
    const uint32_t t = gate_n.fetch_add(1, std::memory_order_acq_rel);
    if (t >= 2) {
      const auto tg0 = std::chrono::steady_clock::now();
      const uint32_t w = (t - 2) & 3u;
      for (unsigned spin = 0;; spin++) {
        if (gate_rec[w].load(std::memory_order_acquire) == t - 1 && hipEventQuery(gate_ev[w]) == hipSuccess) break;
        (void)hipGetLastError();  // (hipErrorNotReady)
        if ((spin & 15) == 15 && std::chrono::steady_clock::now() - tg0 > std::chrono::microseconds(400)) {
          gate_expired.fetch_add(1, std::memory_order_relaxed);
          break;
        }
        cpu_relax();
      }
    }
    const bool ok = hipMemcpyAsync(dst, src, len, hipMemcpyHostToDevice, stream) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    if (hipEventRecord(gate_ev[t & 3u], stream) != hipSuccess) (void)hipGetLastError();
    gate_rec[t & 3u].store(t + 1, std::memory_order_release);
    return ok;
  }

  void enqueue_dma(Slot& s, Group& g) {
    if (s.pull) {
      // a batch the calling thread waits for: the device pulls the group out of the pinned buffer itself
      // (k_stage_pull: lower latency than a copy engine, and the copy engines' state stays out of the call)
      (void)hipGetLastError();  // (whatever this thread's earlier calls left behind is not this launch's)
      if (s.pack)
        launch_stage_pull_packed(stream, s.pin + g.off, (uint8_t*)s.dev + g.off, g.len, s.desc + 2 * (size_t)g.first_chunk, s.pack_epc);
      else
        launch_stage_pull(stream, s.pin + g.off, (uint8_t*)s.dev + g.off, g.len);
      if (hipGetLastError() != hipSuccess) s.state.store(-1, std::memory_order_release);
    } else {
      // an announced batch, staged whole frames ahead: one DMA (a 5 MB pull kernel on the copy stream costs the
      // compute streams 0.025 ms/step in replay mode, the copy engine nothing)
      if (!dma((uint8_t*)s.dev + g.off, s.pin + g.off, g.len)) s.state.store(-1, std::memory_order_release);
    }
    g.dma_enq.store(true, std::memory_order_release);
    if ((int)(&g - s.grp) < s.n_left_groups) {
      // the left array's last DMA to be enqueued (whoever sees them all enqueued; twice does no harm: every
      // DMA whose flag is set is in the stream already)
      bool all = true;
      for (int i = 0; i < s.n_left_groups; i++) all = all && s.grp[i].dma_enq.load(std::memory_order_acquire);
      if (all) {
        if (c->trace) left_sent_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - s.t_begin).count());
        if (hipEventRecord(s.copiedL, stream) != hipSuccess) s.state.store(-1, std::memory_order_release);
        s.left_enq.store(true, std::memory_order_release);
      }
    }
  }
  void finish_batch(Slot& s) {  // every group's DMA is enqueued: the event the compute streams wait for
    if (c->trace) all_sent_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - s.t_begin).count());
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
    if (c->trace) {
      const auto t0 = std::chrono::steady_clock::now();
      if (idx == 0) first_take_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t0 - s.t_begin).count());
      stage_chunk(s, idx);
      chunk_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
      chunk_cnt.fetch_add(1);
    } else {
      stage_chunk(s, idx);
    }
    uint8_t taken = 1;
    if (ch.st.compare_exchange_strong(taken, 2, std::memory_order_acq_rel)) chunk_done(s, ch);
  }

  // (idempotent like everything a chunk goes through: a chunk done twice writes the same bytes and the same descriptor)
  void stage_chunk(Slot& s, uint32_t idx) {
    Chunk& ch = s.chunk[idx];
    if (s.pack) {
      uint32_t base = 0;
      const bool packed = stage_pack(s.pin + ch.off, ch.src, ch.len, &base);
      if (!packed) stage_copy(s.pin + ch.off, ch.src, ch.len);
      s.desc[2 * idx] = base;
      s.desc[2 * idx + 1] = packed ? 1u : 0u;
      (packed ? chunks_packed : chunks_raw).fetch_add(1, std::memory_order_relaxed);
      return;
    }
    stage_copy(s.pin + ch.off, ch.src, ch.len);
  }

  // the calling thread while it waits for the LEFT array of slot k (by_camera): only that array's chunks — a
  // right-array chunk taken now would keep it busy past the moment the left DMA is on its way
  bool try_one_left(int k) {
    Slot& s = slot[k];
    if (s.state.load(std::memory_order_acquire) != 1 || s.next.load(std::memory_order_relaxed) >= s.n_left_chunks) return false;
    s.busy.fetch_add(1, std::memory_order_acq_rel);
    bool did = false;
    if (s.state.load(std::memory_order_acquire) == 1 && s.next.load(std::memory_order_relaxed) < s.n_left_chunks) {
      const uint32_t idx = s.next.fetch_add(1, std::memory_order_acq_rel);  // (may be a right chunk after all: run it)
      if (idx < s.n_chunks.load(std::memory_order_acquire)) {
        pending->fetch_sub(1, std::memory_order_acq_rel);
        run_chunk(s, idx);
        did = true;
      }
    }
    s.busy.fetch_sub(1, std::memory_order_acq_rel);
    return did;
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
      stage_chunk(s, i);
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
    warm_thread();
    for (;;) {
      if (try_one()) continue;
      std::unique_lock<std::mutex> g(mu);
      cv.wait(g, [&] { return stop.load(std::memory_order_acquire) || pending->load(std::memory_order_acquire) > 0; });
      if (stop.load(std::memory_order_acquire)) return;
    }
  }
};

void stager_copy_bytes(uint8_t* dst, const uint8_t* src, size_t len) { stage_copy(dst, src, len); }
bool stager_pack_bytes(uint8_t* dst, const uint8_t* src, size_t len, uint32_t* base_sec) { return stage_pack(dst, src, len, base_sec); }
void stager_counters(esvio_fe_ctx* c, uint64_t out4[4]) {
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  if (EventStager* st = c->stager) {
    out4[0] = st->batches;
    out4[1] = st->bytes_staged;
    out4[2] = st->chunks_packed.load(std::memory_order_relaxed);
    out4[3] = st->chunks_raw.load(std::memory_order_relaxed);
  }
}

// what a spinning RANSAC helper does between jobs (host::ransac_pool_set_idle_work): one chunk
static bool stager_idle_work(void* arg) {
  EventStager* st = (EventStager*)arg;
  static thread_local int dev_set = -1;
  if (dev_set != st->c->dev) {  // (the helper enqueues a group's DMA when it copies the group's last chunk)
    st->warm_thread();
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
    constexpr size_t kWarmBytes = 64;
    if (hipHostMalloc((void**)&st->warm_pin, kWarmBytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&st->warm_dev, kWarmBytes) != hipSuccess ||
        hipEventCreateWithFlags(&st->warm_ev, hipEventDisableTiming) != hipSuccess) {
      c->stager = st;
      stager_destroy(c);
      return fail(c, ESVIO_FE_EHIP, "allocation (event staging) failed");
    }
    std::memset(st->warm_pin, 0, kWarmBytes);
    for (int i = 0; i < 4; i++) {
      st->gate_rec[i].store(0, std::memory_order_relaxed);
      if (hipEventCreateWithFlags(&st->gate_ev[i], hipEventDisableTiming) != hipSuccess) {
        c->stager = st;
        stager_destroy(c);
        return fail(c, ESVIO_FE_EHIP, "hipEventCreate (event staging) failed");
      }
    }
    if (const char* v = getenv("ESVIO_FE_STAGE_PACK")) st->pack_enabled = atoi(v) != 0;
    for (int i = 0; i < c->stage_threads; i++) st->threads.emplace_back([st] { st->worker(); });
    c->stager = st;
    stager_share_pool(c);  // the RANSAC helpers, spinning between jobs anyway, take chunks as well
  }
  *out = c->stager;
  return 0;
}

// [p, p + len) is page-locked memory the runtime knows: first AND last byte (a batch that begins inside a registered
// range and runs past its end is staged like pageable memory — k_stage_pull reading beyond the mapping would be a fatal
// queue error, not a return code)
static bool host_range_is_pinned(const void* p, size_t len) {
  auto pinned = [](const void* q) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) {
      (void)hipGetLastError();  // (an ordinary malloc'ed pointer is "invalid value" to the runtime)
      return false;
    }
    return a.type == hipMemoryTypeHost;
  };
  return len && pinned(p) && pinned((const uint8_t*)p + len - 1);
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

// the slot's chunk table (and, beside it, the pinned descriptor pairs of a packed batch) for `chunks` chunks
static int slot_chunks(esvio_fe_ctx* c, Slot& s, size_t chunks) {
  if (chunks + 4 > s.chunk_cap) {  // (+: a chunk never straddles the two source arrays)
    s.chunk_cap = chunks + 4 + chunks / 4;
    s.chunk.reset(new Chunk[s.chunk_cap]);
  }
  if (s.chunk_cap > s.desc_cap) {
    if (s.desc) (void)hipHostFree(s.desc);
    s.desc = nullptr;
    s.desc_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&s.desc, s.chunk_cap * 8, hipHostMallocDefault));
    c->n_allocs++;
    s.desc_cap = s.chunk_cap;
  }
  return 0;
}

// esvio_fe_reserve: every idle slot sized for batches of n events
int stager_reserve(esvio_fe_ctx* c, size_t n_events) {
  if (!stager_enabled(c)) return 0;
  EventStager* st = nullptr;
  if (int rc = stager_get(c, &st)) return rc;
  for (Slot& s : st->slot)
    if (!s.in_use) {
      if (int rc = slot_capacity(c, s, n_events)) return rc;
      const size_t chunks = (n_events * 16 + kChunkBytes / 4 - 1) / (kChunkBytes / 4);  // (the by-camera chunk size)
      if (int rc = slot_chunks(c, s, chunks)) return rc;
    }
  return 0;
}

// Start staging [left; right] into a free slot: returns at once, the helpers do the work.
// dma_groups: pieces the pageable part of the batch is moved in (1: an announced batch, nobody waits for it: one
// DMA; more: a plain call's batch, piece k goes to the device — by k_stage_pull — under the memcpy of piece k+1)
// by_camera: the left array and the right array are two pieces each (dma_groups is ignored), and the left
// one's arrival can be waited for on its own (stager_attach_left)
static int stager_begin_impl(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL, const esvio_fe_event* right, size_t nR,
                             int dma_groups, int* slot_out, bool by_camera);
int stager_begin(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL, const esvio_fe_event* right, size_t nR,
                 int dma_groups, int* slot_out, bool by_camera) {
  if (!c->trace) return stager_begin_impl(c, left, nL, right, nR, dma_groups, slot_out, by_camera);
  {  // (the stager's one-time set-up — stream, events, threads: ~9 ms — is not part of a batch's figure)
    EventStager* st = nullptr;
    if (int rc = stager_get(c, &st)) return rc;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = stager_begin_impl(c, left, nL, right, nR, dma_groups, slot_out, by_camera);
  if (c->stager)
    c->stager->begin_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
static int stager_begin_impl(esvio_fe_ctx* c, const esvio_fe_event* left, size_t nL, const esvio_fe_event* right, size_t nR,
                             int dma_groups, int* slot_out, bool by_camera) {
  EventStager* st = nullptr;
  if (int rc = stager_get(c, &st)) return rc;
  int k = 0;
  // (a slot a straggling helper is still inside — its chunk was redone by the calling thread — is passed over)
  while (k < kStageSlots && (st->slot[k].in_use || st->slot[k].busy.load(std::memory_order_acquire) != 0)) k++;
  if (k == kStageSlots) return fail(c, ESVIO_FE_EINTERNAL, "no free event staging slot");
  Slot& s = st->slot[k];
  const size_t n = nL + nR;
  const auto tp0 = std::chrono::steady_clock::now();
  const bool pinL = host_range_is_pinned(left, nL * 16), pinR = host_range_is_pinned(right, nR * 16);
  if (c->trace) st->pin_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tp0).count();
  if (int rc = slot_capacity(c, s, n)) return rc;
  // the DMA overwrites the slot's device buffer: behind the kernels that read its previous batch
  if (s.pf_rec) HIPCHK(c, hipStreamWaitEvent(st->stream, s.pf_done, 0));
  if (s.main_rec) HIPCHK(c, hipStreamWaitEvent(st->stream, s.main_done, 0));
  if (s.aux_rec) HIPCHK(c, hipStreamWaitEvent(st->stream, s.aux_done, 0));
  s.pf_rec = s.main_rec = s.aux_rec = false;
  s.n_left_groups = 0;
  s.n_left_chunks = 0;
  s.pull = by_camera || dma_groups > 1;  // (a batch the calling thread waits for)
  s.pack = false;
  s.left_enq.store(false, std::memory_order_release);
  st->bytes_staged += n * 16;
  st->batches++;
  // a pinned source: one DMA straight from it, now (the slot is taken only once nothing below can fail
  // before its state is set)
  // (a batch the calling thread waits for: pulled by a kernel, as its pageable groups are — hipMemcpyAsync out of
  // hipHostRegister'ed memory ran at 10 GB/s on this stack: the plain call took 0.77 ms against 0.40 with the pull
  // kernel and 0.45 from pageable memory; an announced batch: the copy engine, which costs the compute streams nothing)
  auto from_pinned = [&](void* dst, const void* src, size_t len) {
    if (!s.pull) return st->dma(dst, src, len);
    // (hipHostMalloc memory is mapped at its host address; hipHostRegister'ed memory need not be: ask)
    // The kernel reads uint4s: a source that is not 16-byte aligned (an EventArray inside a deserialisation buffer is
    // only 8-byte aligned in general), or whose first and last byte are mapped by two different registrations that do
    // not continue each other on the device side, goes through the copy engine instead.
    void *dp = nullptr, *dp_last = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<void*>(src), 0) != hipSuccess || !dp || ((uintptr_t)dp & 15) != 0 ||
        ((uintptr_t)dst & 15) != 0 ||
        hipHostGetDevicePointer(&dp_last, (uint8_t*)const_cast<void*>(src) + len - 1, 0) != hipSuccess ||
        (uint8_t*)dp_last != (uint8_t*)dp + len - 1) {
      (void)hipGetLastError();
      return st->dma(dst, src, len);
    }
    (void)hipGetLastError();
    launch_stage_pull(st->stream, dp, dst, len);
    return hipGetLastError() == hipSuccess;
  };
  if (pinL && !from_pinned(s.dev, left, nL * 16)) return fail(c, ESVIO_FE_EHIP, "event staging from pinned memory failed");
  if (by_camera && (pinL || !nL)) {  // (the left array is on its way already, or there is none)
    HIPCHK(c, hipEventRecord(s.copiedL, st->stream));
    s.left_enq.store(true, std::memory_order_release);
  }
  if (pinR && !from_pinned(s.dev + nL, right, nR * 16)) return fail(c, ESVIO_FE_EHIP, "event staging from pinned memory failed");
  s.in_use = true;
  // the pageable part: destination byte range [lo, hi) of the slot's buffers
  const size_t lo = pinL ? nL * 16 : 0, hi = pinR ? nL * 16 : n * 16;
  int ng = 0;
  uint32_t nch = 0;
  s.n_chunks.store(0, std::memory_order_release);
  s.next.store(0, std::memory_order_release);
  if (hi > lo) {
    // (by camera: the calling thread waits for the left array: smaller pieces, more threads on it at once)
    const size_t cb = by_camera ? kChunkBytes / 4 : kChunkBytes;
    const size_t chunks = (hi - lo + cb - 1) / cb;
    // the groups (one DMA each): destination byte ranges [a, b)
    size_t ga[kMaxGroups], gb[kMaxGroups];
    auto split = [&](size_t a0, size_t b0, int want) {  // [a0, b0) into <= want groups of whole chunks
      const size_t nc = (b0 - a0 + cb - 1) / cb;
      const int k = (int)std::min<size_t>((size_t)std::max(1, want), nc);
      const size_t per = (nc + k - 1) / k;
      for (size_t o = a0; o < b0 && ng < kMaxGroups; o += per * cb) {
        ga[ng] = o;
        gb[ng] = ng == kMaxGroups - 1 ? b0 : std::min(b0, o + per * cb);
        ng++;
      }
    };
    if (by_camera) {
      // the left array's DMAs can be waited for on their own; group k's DMA runs under the memcpy of group k+1
      if (!pinL && nL) {
        split(0, nL * 16, 2);
        s.n_left_groups = ng;
      }
      if (!pinR && nR) split(nL * 16, n * 16, 2);
    } else {
      split(lo, hi, std::min(dma_groups, kMaxGroups));
    }
    if (int rc = slot_chunks(c, s, chunks)) {
      s.in_use = false;
      return rc;
    }
    // by-camera staging of a batch the call waits for: chunks of one size, every group whole chunks of one array —
    // what the packed form needs (k_stage_pull_packed finds an event's chunk by division)
    s.pack = st->pack_enabled && by_camera && s.pull;
    s.pack_epc = (uint32_t)(cb / 16);
    for (int g = 0; g < ng; g++) {
      const size_t a = ga[g], b = gb[g];
      s.grp[g].off = a;
      s.grp[g].len = b - a;
      s.grp[g].first_chunk = nch;
      s.grp[g].dma_enq.store(false, std::memory_order_relaxed);
      uint32_t cnt = 0;
      for (size_t o = a; o < b;) {
        // a chunk never straddles the boundary between the two source arrays
        const bool in_left = o < nL * 16;
        const size_t end = std::min(std::min(o + cb, b), in_left ? nL * 16 : b);
        const uint8_t* src = in_left ? (const uint8_t*)left + o : (const uint8_t*)right + (o - nL * 16);
        Chunk& ch = s.chunk[nch];
        ch.group = g;
        ch.src = src;
        ch.off = o;
        ch.len = end - o;
        ch.st.store(0, std::memory_order_relaxed);
        nch++;
        cnt++;
        if (g < s.n_left_groups) s.n_left_chunks = nch;
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
  s.t_begin = std::chrono::steady_clock::now();
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
        cpu_relax();
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
  const auto tl0 = std::chrono::steady_clock::now();
  struct Lap {
    EventStager* st; std::chrono::steady_clock::time_point t0;
    ~Lap() { st->left_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); st->left_calls++; }
  } lap{st, tl0};
  while (!sl.left_enq.load(std::memory_order_acquire) && sl.state.load(std::memory_order_acquire) == 1) {
    if (st->try_one_left(slot)) {
      st->caller_chunks++;
      idle = 0;
    } else if (++idle > 1000) {
      st->finish_for(slot);
      idle = 0;
    } else {
      cpu_relax();
    }
  }
  const int stt = sl.state.load(std::memory_order_acquire);
  const bool left = sl.left_enq.load(std::memory_order_acquire);
  if (stt < 0 || (!left && stt != 2)) {
    stager_abandon(c, slot);
    return fail(c, ESVIO_FE_EHIP, "staging the event batch failed (left array)");
  }
  // (the whole batch can be complete — finish_for above closed it — while the thread that sent the last left
  // group has not got to recording copiedL yet: the batch's own event covers the left array as well)
  HIPCHK(c, hipStreamWaitEvent(s, left ? sl.copiedL : sl.copied, 0));
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
  while (sl.busy.load(std::memory_order_acquire) != 0) cpu_relax();  // chunks a thread holds: waited for
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
  while (sl.busy.load(std::memory_order_acquire) != 0) cpu_relax();
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
        cpu_relax();
      }
    }
    while (s.busy.load(std::memory_order_acquire) != 0) cpu_relax();
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
  if (c->trace && st->batches)
    fprintf(stderr, "[esvio_fe trace] host-event staging, calling thread per batch: stager_begin %.1f us (pinned-or-not query %.1f), "
            "wait for the left array %.1f us (%llu by-camera batches), wait for the whole batch %.1f us; %llu DMAs went out without waiting for the one two before\n",
            st->begin_ns / 1e3 / st->batches, st->pin_ns / 1e3 / st->batches, st->left_calls ? st->left_ns / 1e3 / st->left_calls : 0.0,
            (unsigned long long)st->left_calls, st->wait_ns / 1e3 / st->batches, (unsigned long long)st->gate_expired.load());
  if (c->trace && st->chunk_cnt.load())
    fprintf(stderr, "[esvio_fe trace] host-event staging: %.2f us inside a chunk's copy / packing on average (%llu chunks), chunk 0 taken %.1f us after "
            "the batch's opening, the left array's last group enqueued after %.1f us, the batch's after %.1f us\n",
            st->chunk_ns.load() / 1e3 / st->chunk_cnt.load(), (unsigned long long)st->chunk_cnt.load(), st->first_take_ns.load() / 1e3 / st->batches,
            st->left_sent_ns.load() / 1e3 / std::max<uint64_t>(1, st->left_calls), st->all_sent_ns.load() / 1e3 / st->batches);
  if (st->stream) {
    (void)hipStreamSynchronize(st->stream);
    (void)hipStreamDestroy(st->stream);
  }
  if (st->warm_ev) (void)hipEventDestroy(st->warm_ev);
  for (hipEvent_t ev : st->gate_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (st->warm_dev) (void)hipFree(st->warm_dev);
  if (st->warm_pin) (void)hipHostFree(st->warm_pin);
  for (Slot& s : st->slot) {
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.copiedL) (void)hipEventDestroy(s.copiedL);
    if (s.aux_done) (void)hipEventDestroy(s.aux_done);
    if (s.pf_done) (void)hipEventDestroy(s.pf_done);
    if (s.main_done) (void)hipEventDestroy(s.main_done);
    if (s.dev) (void)hipFree(s.dev);
    if (s.pin) (void)hipHostFree(s.pin);
    if (s.desc) (void)hipHostFree(s.desc);
  }
  delete st;
  c->stager = nullptr;
}

}  // namespace fe
}  // namespace esvio
