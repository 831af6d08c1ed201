// fe_kernels.hip — hand-written CDNA4 (gfx950) kernels of the ESVIO event front-end.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (bit-exactness with the reference's
// x86-64 -O3 build, which has no FMA contraction, needs every fp64/fp32 mul+add kept separate).
// wave = 64 lanes everywhere; no MFMA (scatter / stencil / reduction kernels, HBM- or
// latency-bound).  Reference semantics are cited per kernel (paths relative to the reference
// tree); OpenCV-internal arithmetic is marked [OpenCV].
#include "fe_kernels.h"
#include "fe_mc.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>

namespace esvio {

// ============================================================================ launches
namespace {
// every launch of this file goes through here
template <typename... P, typename... A>
void launch_k(void (*kernel)(P...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, A... args) {
  static_assert(sizeof...(P) == sizeof...(A), "argument count");
  hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
}
}  // namespace

// ============================================================================ helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ros::Time::toSec(): (double)sec + 1e-9*(double)nsec, two roundings
__device__ __forceinline__ double ev_time(uint32_t sec, uint32_t nsec) {
  return __dadd_rn((double)sec, __dmul_rn(1e-9, (double)nsec));
}

// Barrier for data exchanged through LDS only.  __syncthreads() fences every address space, i.e. it
// waits for the wave's outstanding global loads and stores too (s_waitcnt vmcnt(0)): a kernel that
// prefetches its next tile from HBM, or lets its stores drain while it goes on, must not use it.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int reflect101(int p, int len) {
  // cv::borderInterpolate(p, len, BORDER_REFLECT_101) for -len < p < 2*len-1
  p = p < 0 ? -p : p;
  return p >= len ? 2 * len - 2 - p : p;
}

// trackEvent's per-event gate (feature_tracker.cpp:627-641) + createSAE_* with
// Motion_correction_value (event_detector.cc:102-147): the pixel an in-sensor event is written at.
// t0 = left.events[0].ts.toSec() (:621) is read from the batch itself, dt = header stamp - t0 (:623).
struct McBatch {
  double t0, dt_batch;
};
__device__ __forceinline__ McBatch mc_batch(const McParams& mc, const uint4* __restrict__ evL) {
  McBatch b;
  const uint4 e0 = evL[0];
  b.t0 = ev_time(e0.y, e0.z);
  b.dt_batch = mc.t1 - b.t0;
  return b;
}
__device__ __forceinline__ uint32_t mc_pixel(const McParams& mc, const McBatch& b, int W, int H, const uint4& e) {
  uint32_t x = e.x & 0xffffu, y = e.x >> 16;
  const double et = ev_time(e.y, e.z);
  if (b.dt_batch > 0 && (et - b.t0) / b.dt_batch < 1 && mc.active) {
    int ox, oy;
    mc_warp(mc, W, H, (int)x, (int)y, et - b.t0, &ox, &oy);
    x = (uint32_t)ox;
    y = (uint32_t)oy;
  }
  return x | (y << 16);
}

// ============================================================================ SAE keys
// Coalesced 16 B/lane read of the raw AoS stream; one u32 key + one u32 index out per event.
// Also: per-pass global digit histograms of the keys (LDS pre-aggregation, one global atomic per
// non-empty bin per block) and clearing of the look-back words the sort passes use.
template <bool MC>
__global__ __launch_bounds__(256) void k_sae_keys(const uint4* __restrict__ evL, uint32_t nL,
                                                  const uint4* __restrict__ evR, uint32_t nR,
                                                  int W, int H, uint32_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals,
                                                  uint32_t invalid_key,
                                                  unsigned long long* n_rejected, int passes,
                                                  int bits, uint32_t* __restrict__ ghist,
                                                  uint32_t* __restrict__ lookback,
                                                  uint32_t lookback_words, McParams mc) {
  __shared__ uint32_t h[kRadixMaxPasses << kRadixMaxBits];
  const int bins = 1 << bits;
  for (int i = threadIdx.x; i < passes * bins; i += 256) h[i] = 0;
  __syncthreads();
  const uint32_t n = nL + nR;
  const uint32_t P = (uint32_t)W * (uint32_t)H;
  McBatch mb{};
  if (MC) mb = mc_batch(mc, evL);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool right = i >= nL;
    const uint4 e = right ? evR[i - nL] : evL[i];
    uint32_t x = e.x & 0xffffu, y = e.x >> 16;
    const bool ok = x < (uint32_t)W && y < (uint32_t)H;
    if (MC && ok) {  // the event is written at the warped pixel
      const uint32_t xy = mc_pixel(mc, mb, W, H, e);
      x = xy & 0xffffu;
      y = xy >> 16;
    }
    const uint32_t key = ok ? (right ? P : 0u) + y * (uint32_t)W + x : invalid_key;
    keys[i] = key;
    vals[i] = i;
    if (!ok) atomicAdd(n_rejected, 1ull);
    for (int p = 0; p < passes; p++) atomicAdd(&h[p * bins + ((key >> (p * bits)) & (bins - 1))], 1u);
  }
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < lookback_words;
       i += gridDim.x * blockDim.x)
    lookback[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < passes * bins; i += 256)
    if (h[i]) atomicAdd(&ghist[i], h[i]);
}

void launch_sae_keys(hipStream_t s, const EventRec* evL, uint32_t nL, const EventRec* evR,
                     uint32_t nR, int W, int H, uint32_t* keys, uint32_t* vals,
                     uint32_t invalid_key, unsigned long long* n_rejected, int passes, int bits,
                     uint32_t* ghist, uint32_t* lookback, uint32_t lookback_words,
                     const McParams* mc) {
  const uint32_t n = nL + nR;
  if (!n) return;
  uint32_t grid = (n + 1023) / 1024;  // ~4 events per thread: keeps the global atomics few
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  McParams m = McParams();  // value-initialised: enabled = 0
  if (mc) m = *mc;
  if (m.enabled)
    launch_k(k_sae_keys<true>, dim3(grid), dim3(256), 0, s, (const uint4*)evL, nL,
                       (const uint4*)evR, nR, W, H, keys, vals, invalid_key, n_rejected, passes, bits,
                       ghist, lookback, lookback_words, m);
  else
    launch_k(k_sae_keys<false>, dim3(grid), dim3(256), 0, s, (const uint4*)evL, nL,
                       (const uint4*)evR, nR, W, H, keys, vals, invalid_key, n_rejected, passes, bits,
                       ghist, lookback, lookback_words, m);
}

// ============================================================================ stable radix sort
// LSD radix sort, one kernel per 7/8-bit digit ("onesweep" form): a block ranks its 2048-key tile
// stably (wave-level match-any by ballots keeps stream order), publishes its per-digit counts and
// obtains the counts of all earlier tiles by decoupled look-back instead of a separate
// histogram + scan pass.  Stability (stream order inside a pixel) is what makes the parallel SAE
// update equal to the reference's sequential loop.
//
// Inter-workgroup protocol (placement independent): one 32-bit word per (tile, digit) holds
// {status:2, count:30}; it is written and read with relaxed AGENT-scope atomics (sc1, L2-served),
// data and flag travel in the same word so no fence is needed; tiles are numbered by an atomic
// ticket so a tile only ever waits for tiles that have already started; every spin is bounded and
// raises *err instead of hanging.
constexpr uint32_t kLbAgg = 1u << 30, kLbPrefix = 2u << 30, kLbMask = (1u << 30) - 1u;

__global__ __launch_bounds__(256) void k_radix_pass(const uint32_t* __restrict__ keys_in,
                                                    const uint32_t* __restrict__ vals_in,
                                                    uint32_t n, int shift, int bits,
                                                    const uint32_t* __restrict__ ghist,
                                                    uint32_t* __restrict__ lookback,
                                                    uint32_t* __restrict__ ticket,
                                                    uint32_t* __restrict__ keys_out,
                                                    uint32_t* __restrict__ vals_out,
                                                    int* __restrict__ err, uint32_t spin_limit) {
  __shared__ uint32_t wave_cnt_s[4][1 << kRadixMaxBits];
  __shared__ uint32_t bin_base[1 << kRadixMaxBits];
  __shared__ uint32_t s_tile;
  // NOTE: plain LDS accesses (a volatile generic pointer here turns every access into a
  // flat_load sc0 sc1 + vmcnt(0)); same-wave ordering between rounds is given by the in-order
  // LDS queue plus the wavefront-scope LDS fence below.
  uint32_t(*wave_cnt)[1 << kRadixMaxBits] = wave_cnt_s;
  const int bins = 1 << bits;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  for (int i = threadIdx.x; i < 4 * (1 << kRadixMaxBits); i += 256) (&wave_cnt_s[0][0])[i] = 0;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  if ((int)threadIdx.x < bins) bin_base[threadIdx.x] = ghist[threadIdx.x];
  __syncthreads();
  const uint32_t tile = s_tile;

  constexpr int ROUNDS = kRadixTile / 256;  // 8 rounds of 64 consecutive keys per wave
  uint32_t key[ROUNDS], val[ROUNDS], rank[ROUNDS];
  const uint32_t wbase = tile * kRadixTile + wave * (kRadixTile / 4);
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t i = wbase + r * 64 + lane;
    const bool ok = i < n;
    key[r] = ok ? keys_in[i] : 0xffffffffu;
    val[r] = ok ? vals_in[i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t i = wbase + r * 64 + lane;
    const bool ok = i < n;
    const uint32_t d = (key[r] >> shift) & (bins - 1);
    // match-any on the digit: lanes holding the same digit, in lane (= stream) order
    unsigned long long m = __ballot(ok);
    for (int b = 0; b < bits; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = __popcll(m & lt);
    const uint32_t cnt = __popcll(m);
    uint32_t base = 0;
    if (ok) base = wave_cnt[wave][d];
    rank[r] = base + before;
    if (ok && before == 0) wave_cnt[wave][d] = base + cnt;  // one leader per digit
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  }
  __syncthreads();
  // exclusive scan of the global digit totals (serial, <= 256 bins, one thread)
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < bins; i++) {
      const uint32_t t = bin_base[i];
      bin_base[i] = run;
      run += t;
    }
  }
  uint32_t excl = 0;
  if ((int)threadIdx.x < bins) {
    const int d = threadIdx.x;
    uint32_t cnt = 0;
    for (int w = 0; w < 4; w++) cnt += wave_cnt[w][d];
    uint32_t* my = lookback + (size_t)tile * bins + d;
    if (tile == 0) {
      __hip_atomic_store(my, kLbPrefix | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_store(my, kLbAgg | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int j = (int)tile - 1;
      uint32_t spins = 0;
      while (j >= 0) {
        const uint32_t w = __hip_atomic_load(lookback + (size_t)j * bins + d, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t st = w & ~kLbMask;
        if (st == 0) {
          if (++spins > spin_limit) {  // bounded (kSpinLookback polls): never hang the GPU
            *err = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        excl += w & kLbMask;
        if (st == kLbPrefix) break;
        j--;
      }
      __hip_atomic_store(my, kLbPrefix | (excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < bins) {
    uint32_t run = bin_base[threadIdx.x] + excl;
    for (int w = 0; w < 4; w++) {
      const uint32_t t = wave_cnt[w][threadIdx.x];
      wave_cnt[w][threadIdx.x] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t i = wbase + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (key[r] >> shift) & (bins - 1);
      const uint32_t pos = wave_cnt[wave][d] + rank[r];
      keys_out[pos] = key[r];
      vals_out[pos] = val[r];
    }
  }
}

void launch_radix_pass(hipStream_t s, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t n,
                       int shift, int bits, const uint32_t* ghist, uint32_t* lookback,
                       uint32_t* ticket, uint32_t* keys_out, uint32_t* vals_out, int* err, uint32_t spin_limit) {
  launch_k(k_radix_pass, dim3(radix_blocks(n)), dim3(256), 0, s, keys_in, vals_in, n, shift,
                     bits, ghist, lookback, ticket, keys_out, vals_out, err, spin_limit);
}

// ============================================================================ SAE apply
// createSAE_left/right (event_detector.cc:149-166, :212-228) for a whole batch.  After the stable
// sort every pixel's events are one contiguous segment in stream order; the segment's head lane
// walks it with the four plane values in registers:
//     pass = (t > L[p] + thr) || (L[!p] > L[p]);  L[p] = t;  if (pass) S[p] = t;
__global__ __launch_bounds__(256) void k_sae_apply(const uint32_t* __restrict__ keys,
                                                   const uint32_t* __restrict__ vals, uint32_t n,
                                                   const uint4* __restrict__ evL, uint32_t nL,
                                                   const uint4* __restrict__ evR,
                                                   double2* __restrict__ L2,
                                                   double2* __restrict__ S2, double thr,
                                                   uint32_t invalid_key,
                                                   uint32_t* __restrict__ sort_scratch,
                                                   uint32_t sort_scratch_words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  // the sort of this batch is finished: clear its digit histograms and tickets for the next one
  if (i < sort_scratch_words) sort_scratch[i] = 0;
  if (i >= n) return;
  const uint32_t k = keys[i];
  if (k == invalid_key) return;
  if (i > 0 && keys[i - 1] == k) return;  // not a segment head
  double2 Lv = L2[k], Sv = S2[k];
  double L[2] = {Lv.x, Lv.y}, S[2] = {Sv.x, Sv.y};
  // Which events belong to the segment does not depend on the state, only the four values do: the
  // keys / indices / events of kAhead positions are fetched together (two dependent round trips
  // per group instead of two per event), then applied in order.
  constexpr int kAhead = 8;
  for (uint32_t j0 = i;; j0 += kAhead) {
    uint32_t kk[kAhead], idx[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      const uint32_t j = min(j0 + u, n - 1);
      kk[u] = j0 + u < n ? keys[j] : ~k;
      idx[u] = vals[j];
    }
    uint4 e[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) e[u] = idx[u] >= nL ? evR[idx[u] - nL] : evL[idx[u]];
    bool more = true;
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      more = more && kk[u] == k;
      if (more) {
        const double t = ev_time(e[u].y, e[u].z);
        const bool p = (e[u].w & 0xffu) != 0;
        const double t_last = p ? L[1] : L[0];
        const double t_last_inv = p ? L[0] : L[1];
        const bool pass = (t > __dadd_rn(t_last, thr)) || (t_last_inv > t_last);
        if (p) {
          L[1] = t;
          if (pass) S[1] = t;
        } else {
          L[0] = t;
          if (pass) S[0] = t;
        }
      }
    }
    if (!more) break;
  }
  L2[k] = make_double2(L[0], L[1]);
  S2[k] = make_double2(S[0], S[1]);
}

void launch_sae_apply(hipStream_t s, const uint32_t* keys, const uint32_t* vals, uint32_t n,
                      const EventRec* evL, uint32_t nL, const EventRec* evR, double2* L2,
                      double2* S2, double filter_threshold, uint32_t invalid_key,
                      uint32_t* sort_scratch, uint32_t sort_scratch_words) {
  if (!n) return;
  const uint32_t need = n > sort_scratch_words ? n : sort_scratch_words;
  launch_k(k_sae_apply, dim3((need + 255) / 256), dim3(256), 0, s, keys, vals, n,
                     (const uint4*)evL, nL, (const uint4*)evR, L2, S2, filter_threshold,
                     invalid_key, sort_scratch, sort_scratch_words);
}

// ---- per-event form of the same update -------------------------------------------------------
// Every sorted position gets a lane.  What an event needs from its pixel's history is only the time
// of the nearest earlier event of its own and of the other polarity (that IS L[p] / L[!p] when it is
// processed), so pass flags are computed for all events at once; L[q] ends up as the time of the
// pixel's last event of polarity q, S[q] as that of its last PASSING event of polarity q; the lane
// that finds itself to be that event marks it, and a second launch (k_sae_apply_ev_write) stores
// the marked events' times — the lanes that fall back to the stored planes (no earlier event of
// that polarity in the batch) and the lanes that replace them can sit in different waves, so reads
// and writes of the planes are kept in different launches.  Inside a wave "nearest earlier" /
// "any later" are ballots restricted to the lane's run of equal keys plus a lane shuffle; only the
// first run of a wave can have history before the wave and only the last run a future after it —
// those two are resolved by the whole wave scanning 64 positions at a time backwards (until both
// polarities are found or the segment starts) and forwards (re-evaluating the pass rule with the
// carried times, until what this wave's candidates need is decided or the segment ends).
__global__ __launch_bounds__(256) void k_sae_apply_ev(const uint32_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ vals, uint32_t n,
                                                      const uint4* __restrict__ evL, uint32_t nL,
                                                      const uint4* __restrict__ evR,
                                                      double2* __restrict__ L2,
                                                      double2* __restrict__ S2, double thr,
                                                      uint32_t invalid_key,
                                                      uint32_t* __restrict__ sort_scratch,
                                                      uint32_t sort_scratch_words,
                                                      uint8_t* __restrict__ marks) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < sort_scratch_words) sort_scratch[i] = 0;
  const int lane = lane_id();
  const uint32_t base = i - (uint32_t)lane;
  if (base >= n) return;  // (wave-uniform)
  const bool in = i < n;
  const uint32_t k = in ? keys[i] : 0xffffffffu;
  const bool valid = in && k != invalid_key;
  const uint32_t idx = vals[in ? i : n - 1];
  const uint4 ev = idx >= nL ? evR[idx - nL] : evL[idx];
  const double t = ev_time(ev.y, ev.z);
  const bool p = (ev.w & 0xffu) != 0;
  auto msb = [](unsigned long long m) { return 63 - __clzll((long long)m); };
  const unsigned long long self = 1ull << lane, below = self - 1ull, above = ~(below | self);
  // this lane's run of equal keys inside the wave: lanes startL..endL
  const uint32_t k_up = __shfl_up(k, 1);
  const unsigned long long heads = __ballot(lane == 0 || k_up != k);
  const int startL = msb(heads & (below | self));
  const unsigned long long h_above = heads & above;
  const int endL = h_above ? __builtin_ctzll(h_above) - 1 : 63;
  const unsigned long long seg = (endL == 63 ? ~0ull : ((2ull << endL) - 1ull)) & ~((1ull << startL) - 1ull);
  const unsigned long long m1 = __ballot(valid && p), m0 = __ballot(valid && !p);
  const unsigned long long b1 = m1 & seg & below, b0 = m0 & seg & below;
  const double q1 = __shfl(t, b1 ? msb(b1) : lane), q0 = __shfl(t, b0 ? msb(b0) : lane);
  // history before the wave (first run only)
  const uint32_t k_first = __builtin_amdgcn_readfirstlane(k);
  const bool cont_before = base > 0 && k_first != invalid_key && keys[base - 1] == k_first;
  double c1 = 0, c0 = 0;
  bool f1 = false, f0 = false;
  if (cont_before) {
    for (uint32_t end = base;;) {  // positions [end-64, end)
      const bool ok = end >= 64u - (uint32_t)lane;  // end - 64 + lane >= 0
      const uint32_t j = end - 64u + (uint32_t)lane;
      const bool inb = ok && keys[ok ? j : 0] == k_first;  // (a suffix of the lanes: sorted keys)
      const uint32_t jd = vals[inb ? j : base];
      const uint4 e2 = jd >= nL ? evR[jd - nL] : evL[jd];
      const double t2 = ev_time(e2.y, e2.z);
      const bool p2 = (e2.w & 0xffu) != 0;
      const unsigned long long mm1 = __ballot(inb && p2), mm0 = __ballot(inb && !p2);
      const double x1 = __shfl(t2, mm1 ? msb(mm1) : 0), x0 = __shfl(t2, mm0 ? msb(mm0) : 0);
      if (!f1 && mm1) {
        c1 = x1;
        f1 = true;
      }
      if (!f0 && mm0) {
        c0 = x0;
        f0 = true;
      }
      if ((f1 && f0) || __ballot(inb) != ~0ull) break;
      end -= 64;
    }
  }
  const bool first_run = startL == 0 && cont_before;
  // the stored planes, where the batch holds no earlier event of that polarity for this pixel
  double2 Lst = make_double2(0, 0);
  const bool need1 = !b1 && !(first_run && f1), need0 = !b0 && !(first_run && f0);
  if (valid && (need1 || need0)) Lst = L2[k];
  const double prev1 = b1 ? q1 : (first_run && f1 ? c1 : Lst.y);
  const double prev0 = b0 ? q0 : (first_run && f0 ? c0 : Lst.x);
  const double t_last = p ? prev1 : prev0, t_last_inv = p ? prev0 : prev1;
  const bool pass = valid && ((t > __dadd_rn(t_last, thr)) || (t_last_inv > t_last));
  const unsigned long long s1 = __ballot(pass && p), s0 = __ballot(pass && !p);
  bool later_any = ((p ? m1 : m0) & seg & above) != 0;    // a later event of my polarity exists
  bool later_pass = ((p ? s1 : s0) & seg & above) != 0;   // ... a later passing one
  // future after the wave (last run only)
  const uint32_t k_last = __shfl(k, 63);
  const bool cont_after = base + 64 < n && k_last != invalid_key && keys[base + 64 < n ? base + 64 : 0] == k_last;
  if (cont_after) {
    // the last run's lanes: [lastStart, 63]
    const int lastStart = msb(heads);
    const unsigned long long lastseg = ~((1ull << lastStart) - 1ull);
    const bool whole = lastStart == 0 && cont_before;  // the run has history before this wave too
    // times of the nearest events of either polarity as of the end of this wave
    const unsigned long long l1 = m1 & lastseg, l0 = m0 & lastseg;
    const double2 Ll = L2[k_last];
    double E1 = l1 ? __shfl(t, msb(l1)) : (whole && f1 ? c1 : Ll.y);
    double E0 = l0 ? __shfl(t, msb(l0)) : (whole && f0 ? c0 : Ll.x);
    // what this wave's candidates still need to know
    bool want_any1 = (l1 != 0), want_any0 = (l0 != 0);              // "is my last p=q event the pixel's last?"
    bool want_pass1 = (s1 & lastseg) != 0, want_pass0 = (s0 & lastseg) != 0;
    bool any1 = false, any0 = false, ps1 = false, ps0 = false;
    for (uint32_t pos = base + 64;; pos += 64) {
      const uint32_t j = pos + (uint32_t)lane;
      const bool inb = j < n && keys[j < n ? j : n - 1] == k_last;  // (a prefix of the lanes)
      const unsigned long long m_in = __ballot(inb);
      if (!m_in) break;
      const uint32_t jd = vals[inb ? j : base];
      const uint4 e2 = jd >= nL ? evR[jd - nL] : evL[jd];
      const double t2 = ev_time(e2.y, e2.z);
      const bool p2 = (e2.w & 0xffu) != 0;
      const unsigned long long mm1 = __ballot(inb && p2), mm0 = __ballot(inb && !p2);
      const unsigned long long bb1 = mm1 & below, bb0 = mm0 & below;
      const double y1 = __shfl(t2, bb1 ? msb(bb1) : lane), y0 = __shfl(t2, bb0 ? msb(bb0) : lane);
      const double pv1 = bb1 ? y1 : E1, pv0 = bb0 ? y0 : E0;
      const double tl = p2 ? pv1 : pv0, ti = p2 ? pv0 : pv1;
      const bool pass2 = inb && ((t2 > __dadd_rn(tl, thr)) || (ti > tl));
      const unsigned long long ss1 = __ballot(pass2 && p2), ss0 = __ballot(pass2 && !p2);
      any1 = any1 || mm1;
      any0 = any0 || mm0;
      ps1 = ps1 || ss1;
      ps0 = ps0 || ss0;
      const double z1 = __shfl(t2, mm1 ? msb(mm1) : 0), z0 = __shfl(t2, mm0 ? msb(mm0) : 0);
      if (mm1) E1 = z1;
      if (mm0) E0 = z0;
      const bool done = (!want_any1 || any1) && (!want_any0 || any0) && (!want_pass1 || ps1) &&
                        (!want_pass0 || ps0);
      if (done || m_in != ~0ull) break;
    }
    if ((lastseg >> lane) & 1ull) {
      later_any = later_any || (p ? any1 : any0);
      later_pass = later_pass || (p ? ps1 : ps0);
    }
  }
  if (in) marks[i] = (uint8_t)((valid && !later_any ? 1 : 0) | (pass && !later_pass ? 2 : 0));
}

__global__ __launch_bounds__(256) void k_sae_apply_ev_write(const uint32_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ vals, uint32_t n,
                                                            const uint4* __restrict__ evL, uint32_t nL,
                                                            const uint4* __restrict__ evR,
                                                            double2* __restrict__ L2,
                                                            double2* __restrict__ S2,
                                                            const uint8_t* __restrict__ marks) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = marks[i];
  if (!f) return;
  const uint32_t k = keys[i], idx = vals[i];
  const uint4 ev = idx >= nL ? evR[idx - nL] : evL[idx];
  const double t = ev_time(ev.y, ev.z);
  const int q = (ev.w & 0xffu) != 0 ? 1 : 0;
  if (f & 1u) ((double*)&L2[k])[q] = t;
  if (f & 2u) ((double*)&S2[k])[q] = t;
}

void launch_sae_apply_ev(hipStream_t s, const uint32_t* keys, const uint32_t* vals, uint32_t n,
                         const EventRec* evL, uint32_t nL, const EventRec* evR, double2* L2,
                         double2* S2, double filter_threshold, uint32_t invalid_key,
                         uint32_t* sort_scratch, uint32_t sort_scratch_words, uint8_t* marks) {
  if (!n) return;
  const uint32_t need = n > sort_scratch_words ? n : sort_scratch_words;
  launch_k(k_sae_apply_ev, dim3((need + 255) / 256), dim3(256), 0, s, keys, vals, n, (const uint4*)evL,
           nL, (const uint4*)evR, L2, S2, filter_threshold, invalid_key, sort_scratch, sort_scratch_words,
           marks);
  launch_k(k_sae_apply_ev_write, dim3((n + 255) / 256), dim3(256), 0, s, keys, vals, n, (const uint4*)evL,
           nL, (const uint4*)evR, L2, S2, (const uint8_t*)marks);
}

// ============================================================================ SAE update, tiled
// See fe_kernels.h (TileGeom).  Bucket of an event = camera * nt_cam + (y / th) * tiles_x + x / tw,
// or the last bin for an out-of-sensor event.
// The partitioned records.  Wide: the raw 16-byte dvs_msgs::Event.  Compact (8 bytes; the default
// whenever the batch allows it, k_tile_scan): x = tile-local pixel (11 bits) | polarity << 11 |
// (sec - sec_base) << 12 (20 bits), y = nsec (30 bits) — everything k_tile_apply needs, at half
// the partition's write and the apply's read volume; the event time is rebuilt exactly
// (ros::Time::toSec() of the same two integers).
__device__ __forceinline__ uint2 tile_rec_pack(const uint4& e, int tw, int th, int twsh, uint32_t sec_base) {
  const uint32_t x = e.x & 0xffffu, y = e.x >> 16;
  const uint32_t pix = ((y & (uint32_t)(th - 1)) << twsh) | (x & (uint32_t)(tw - 1));
  return make_uint2(pix | ((e.w & 0xffu) ? 1u << 11 : 0u) | ((e.y - sec_base) << 12), e.z);
}
template <bool COMPACT>
struct TileRec;
template <>
struct TileRec<false> {
  typedef uint4 T;
  static __device__ __forceinline__ T load(const void* part, uint32_t i) { return ((const uint4*)part)[i]; }
  static __device__ __forceinline__ uint32_t pix(const T& e, int x0, int y0, int twsh) {
    return (((e.x >> 16) - (uint32_t)y0) << twsh) + ((e.x & 0xffffu) - (uint32_t)x0);
  }
  static __device__ __forceinline__ bool pol(const T& e) { return (e.w & 0xffu) != 0; }
  static __device__ __forceinline__ double time(const T& e, uint32_t) { return ev_time(e.y, e.z); }
};
template <>
struct TileRec<true> {
  typedef uint2 T;
  static __device__ __forceinline__ T load(const void* part, uint32_t i) { return ((const uint2*)part)[i]; }
  static __device__ __forceinline__ uint32_t pix(const T& e, int, int, int) { return e.x & 0x7ffu; }
  static __device__ __forceinline__ bool pol(const T& e) { return (e.x >> 11) & 1u; }
  static __device__ __forceinline__ double time(const T& e, uint32_t sec_base) { return ev_time(sec_base + (e.x >> 12), e.y); }
};

// bucket of an event INSIDE its camera: tile index, or nt_cam for an out-of-sensor event
__device__ __forceinline__ uint32_t tile_bin(const TileGeom& g, uint32_t xy) {
  const uint32_t x = xy & 0xffffu, y = xy >> 16;
  if (x >= (uint32_t)g.W || y >= (uint32_t)g.H) return (uint32_t)g.nt_cam;
  // tw is 32 or 64, th 16 or 32: shifts
  const uint32_t tx = x >> (g.tw == 64 ? 6 : 5), ty = y >> (g.th == 32 ? 5 : 4);
  return ty * (uint32_t)g.tiles_x + tx;
}

// Per-block bucket counts without global atomics or look-back: block `seg` of k_tile_hist walks `group`
// consecutive scatter blocks of ONE camera one after the other (TileSplit, fe_kernels.h) and writes, per
// scatter block, the exclusive prefix of the bucket counts inside its group (P[b][bucket], one coalesced row)
// and the group totals (T[seg][bucket]); k_tile_scan turns the group totals into exclusive prefixes over
// the camera's groups (C[seg][bucket]) and bucket totals.  k_tile_scatter then knows where every event goes:
// bucket start + C[group][bucket] + P[block][bucket] + rank inside the block.  Rows have nt_cam + 1 columns.
// MC (the motion-compensated overload, feature_tracker.cpp:605-641): the bucket is that of the WARPED
// pixel, computed by k_mc_warp (one lane per event: the per-event Matrix3f::exp + LU solve is a long
// dependent chain, so it wants every SIMD full — inside k_tile_hist, whose 1024-thread blocks take
// 2-4 events per thread one after the other, it cost 28 us at C3, on its own 8) into warp_xy[i]
// (4 B per event); k_tile_scatter puts the warped pixel into the partitioned records — nothing after
// the partition knows about the warp.
__global__ __launch_bounds__(256) void k_mc_warp(const uint4* __restrict__ evL, uint32_t nL,
                                                 const uint4* __restrict__ evR, uint32_t nR, int W, int H,
                                                 McParams mc, uint32_t* __restrict__ warp_xy) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nL + nR) return;
  const McBatch mb = mc_batch(mc, evL);
  const uint4 e = i >= nL ? evR[i - nL] : evL[i];
  uint32_t xy = e.x;
  if ((xy & 0xffffu) < (uint32_t)W && (xy >> 16) < (uint32_t)H) xy = mc_pixel(mc, mb, W, H, e);
  warp_xy[i] = xy;
}

constexpr int kTileHistThreads = 1024;
template <bool MC>
__global__ __launch_bounds__(kTileHistThreads) void k_tile_hist(const uint4* __restrict__ evL, uint32_t nL,
                                                   const uint4* __restrict__ evR, uint32_t nR, TileGeom g,
                                                   TileSplit sp, uint32_t* __restrict__ Pm, uint32_t* __restrict__ Tm,
                                                   const uint32_t* __restrict__ warp_xy,
                                                   uint4* __restrict__ ranges) {
  constexpr int UE = kTileScatterEvents / kTileHistThreads;  // a thread's events per scatter block
  // two histograms, by lane parity: the events of a moving edge come in runs of one bucket, and the LDS
  // takes same-address atomics of a wave one after the other (scene stream at C5: 33.7 -> 28.6 us;
  // four copies: 29.1; uniform events: unchanged)
  constexpr int HC = 2;
  __shared__ uint32_t h[HC * kTileMaxBins], run[kTileMaxBins];  // 24 KiB
  __shared__ uint32_t s_range[3];
  const int nbc = g.nt_cam + 1;  // this camera's buckets + the out-of-sensor bin
  uint32_t* const hme = h + (threadIdx.x & (HC - 1)) * kTileMaxBins;
  for (int i = threadIdx.x; i < kTileMaxBins; i += kTileHistThreads) {
#pragma unroll
    for (int q = 0; q < HC; q++) h[q * kTileMaxBins + i] = 0;
    run[i] = 0;
  }
  if (threadIdx.x == 0) {
    s_range[0] = 0xffffffffu;
    s_range[1] = 0;
    s_range[2] = 0;
  }
  // the batch's range of seconds and whether every nsec fits 30 bits (in-sensor events): decides
  // whether the partitioned records can take the 8-byte form (TileScratch::ranges, k_tile_scan)
  uint32_t tmin = 0xffffffffu, tmax = 0, tor = 0;
  // this block's camera, its array, its scatter blocks [b0, b0 + group) (clipped to the camera's)
  const bool right = blockIdx.x >= sp.nsegL;
  const uint4* __restrict__ ev = right ? evR : evL;
  const uint32_t ncam = right ? nR : nL, ioff = right ? nL : 0u;  // (ioff: the array's place in the whole stream, for warp_xy)
  const uint32_t lb0 = (blockIdx.x - (right ? sp.nsegL : 0u)) * sp.group, nblk_cam = right ? sp.nblkR : sp.nblkL;
  const uint32_t brow0 = (right ? sp.nblkL : 0u) + lb0;  // row of P of the first one
  const uint32_t te = sp.te, group = sp.group;
  // One scatter block
  // per step: its records are counted into the LDS histogram, then its row of P is written from the
  // running prefix (and the histogram cleared); the NEXT scatter block's records (UE per thread)
  // are requested before that, and the barriers order LDS traffic
  // only (lds_barrier) — so the requests and the stores of P are in flight while a step computes.
  uint4 ne[UE];
  uint32_t nw[UE];
  auto request = [&](uint32_t k) {
    const uint32_t lb = lb0 + k;
    const uint32_t lo = lb * te, hi = (k < group && lb < nblk_cam) ? min(lo + te, ncam) : lo;
#pragma unroll
    for (int j = 0; j < UE; j++) {
      const uint32_t i = lo + threadIdx.x + j * kTileHistThreads;
      ne[j] = make_uint4(0, 0, 0, 0);
      nw[j] = 0;
      if (i < hi) {
        ne[j] = ev[i];  // (whole records: one coalesced 1 KiB request per wave)
        if (MC) nw[j] = warp_xy[ioff + i];
      }
    }
  };
  request(0);
  __syncthreads();
  for (uint32_t k = 0; k < group; k++) {
    const uint32_t lb = lb0 + k;
    const bool live = lb < nblk_cam;
    const uint32_t lo = lb * te, hi = live ? min(lo + te, ncam) : lo;
    uint32_t bins[UE];
#pragma unroll
    for (int j = 0; j < UE; j++) {
      const uint32_t i = lo + threadIdx.x + j * kTileHistThreads;
      bins[j] = 0xffffffffu;
      if (i < hi) {
        bins[j] = tile_bin(g, MC ? nw[j] : ne[j].x);
        if (bins[j] != (uint32_t)g.nt_cam) {
          tmin = min(tmin, ne[j].y);
          tmax = max(tmax, ne[j].y);
          tor |= ne[j].z;
        }
      }
    }
    if (k + 1 < group) request(k + 1);
#ifndef ESVIO_ABL_HIST_NOCOUNT  // (measurement build: the kernel as a pure read of the records)
#pragma unroll
    for (int j = 0; j < UE; j++)
      if (bins[j] != 0xffffffffu) atomicAdd(&hme[bins[j]], 1u);
#else
    if (bins[0] == 0xfffffff0u) hme[0] = bins[UE - 1];
#endif
    lds_barrier();
    if (live)
      for (int i = threadIdx.x; i < nbc; i += kTileHistThreads) {
        const uint32_t r = run[i];
        Pm[(size_t)(brow0 + k) * nbc + i] = r;
        uint32_t cnt = 0;
#pragma unroll
        for (int q = 0; q < HC; q++) {
          cnt += h[q * kTileMaxBins + i];
          h[q * kTileMaxBins + i] = 0;  // (for the next step)
        }
        run[i] = r + cnt;
      }
    lds_barrier();
  }
  for (int i = threadIdx.x; i < nbc; i += kTileHistThreads) Tm[(size_t)blockIdx.x * nbc + i] = run[i];
  // wave-level reduction first, then one LDS atomic per wave, one slot per block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tmin = min(tmin, (uint32_t)__shfl_xor((int)tmin, o));
    tmax = max(tmax, (uint32_t)__shfl_xor((int)tmax, o));
    tor |= (uint32_t)__shfl_xor((int)tor, o);
  }
  if (lane_id() == 0) {
    atomicMin(&s_range[0], tmin);
    atomicMax(&s_range[1], tmax);
    atomicOr(&s_range[2], tor);
  }
  __syncthreads();
  if (threadIdx.x == 0) ranges[blockIdx.x] = make_uint4(s_range[0], s_range[1], s_range[2], 0);
}

// exclusive prefix of the group totals over the camera's groups, per bucket (the out-of-sensor bin: over all groups,
// the left camera's first).  Thread = (one of the block's 64 consecutive buckets, one of 16 consecutive ranges of
// groups): a wave reads 64 consecutive words of a row (the first version took 16 buckets per block: 64-byte pieces
// of every row, 13-15 us for 3 MB).  Buckets are numbered over both cameras here (camera * nt_cam + tile, then the
// out-of-sensor bin: TileGeom::nbins of them) — the numbering of totals, tile_off, tile_order and k_tile_apply.
constexpr int kTileScanBins = 64, kTileScanRanges = 16, kTileScanThreads = kTileScanBins * kTileScanRanges;
constexpr int kTileScanPer = (kTileMaxGroups + kTileScanRanges - 1) / kTileScanRanges;  // groups per thread, at most
__global__ __launch_bounds__(kTileScanThreads) void k_tile_scan(const uint32_t* __restrict__ Tm, TileSplit sp,
                                                                int nb, int nt_cam, uint32_t* __restrict__ Cm,
                                                                uint32_t* __restrict__ totals,
                                                                unsigned long long* n_rejected,
                                                                const uint4* __restrict__ ranges,
                                                                uint32_t* __restrict__ meta, int force_wide) {
  const uint32_t nseg = sp.nsegL + sp.nsegR;
  if (blockIdx.x == gridDim.x - 1) {  // (a block of its own, beside the scan: one wave)
    if (threadIdx.x >= 64) return;
    // 8-byte partitioned records (kTileRec*) when the batch's seconds span < 2^20 and every nsec < 2^30
    // (any ros::Time has nsec < 1e9); else the raw 16-byte records travel.
    uint32_t mn = 0xffffffffu, mx = 0, orr = 0;
    for (uint32_t i = threadIdx.x; i < nseg; i += 64) {
      const uint4 r = ranges[i];
      mn = min(mn, r.x);
      mx = max(mx, r.y);
      orr |= r.z;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
      orr |= (uint32_t)__shfl_xor((int)orr, o);
    }
    if (threadIdx.x == 0) {
      const bool compact = !force_wide && (mn > mx || (mx - mn < (1u << kTileRecSecBits) && (orr >> 30) == 0));
      meta[kTileMetaCompact] = compact ? 1u : 0u;
      meta[kTileMetaSecBase] = mn <= mx ? mn : 0u;
    }
    return;
  }
  __shared__ uint32_t part[kTileScanRanges][kTileScanBins];
  const int bl = threadIdx.x % kTileScanBins, r = threadIdx.x / kTileScanBins;
  const int bin = blockIdx.x * kTileScanBins + bl;
  const bool ok = bin < nb;
  // the bucket's column in its camera's rows, and the rows (groups) that have it
  const int nbc = nt_cam + 1;
  const bool reject = bin == nb - 1, right = !reject && bin >= nt_cam;
  const int col = reject ? nt_cam : bin - (right ? nt_cam : 0);
  const uint32_t sbeg = right ? sp.nsegL : 0u, scnt = reject ? nseg : (right ? sp.nsegR : sp.nsegL);
  const uint32_t per = (scnt + kTileScanRanges - 1) / kTileScanRanges;
  const uint32_t s0 = min((uint32_t)r * per, scnt);
  uint32_t v[kTileScanPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kTileScanPer; k++) {  // (all of a thread's words requested together)
    const uint32_t sg = s0 + (uint32_t)k;
    v[k] = (ok && (uint32_t)k < per && sg < scnt) ? Tm[(size_t)(sbeg + sg) * nbc + col] : 0u;
    sum += v[k];
  }
  part[r][bl] = sum;
  __syncthreads();
  uint32_t carry = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kTileScanRanges; w++) {
    const uint32_t x = part[w][bl];
    if (w < r) carry += x;
    tot += x;
  }
  if (!ok) return;
#pragma unroll
  for (int k = 0; k < kTileScanPer; k++) {
    const uint32_t sg = s0 + (uint32_t)k;
    if ((uint32_t)k < per && sg < scnt) {
      Cm[(size_t)(sbeg + sg) * nbc + col] = carry;
      carry += v[k];
    }
  }
  if (r == 0) {
    totals[bin] = tot;
    if (reject && tot) atomicAdd(n_rejected, (unsigned long long)tot);  // out-of-sensor events
  }
}

void launch_tile_hist(hipStream_t s, const EventRec* evL, uint32_t nL, const EventRec* evR, uint32_t nR,
                      const TileGeom& g, const TileScratch& sc, unsigned long long* n_rejected, const McParams* mc,
                      uint32_t* warp_xy) {
  const uint32_t n = nL + nR;
  if (!n) return;
  const TileSplit sp = tile_split(nL, nR);
  if (mc && mc->enabled) {
    launch_k(k_mc_warp, dim3((n + 255) / 256), dim3(256), 0, s, (const uint4*)evL, nL, (const uint4*)evR, nR, g.W, g.H,
             *mc, warp_xy);
    launch_k(k_tile_hist<true>, dim3(sp.nseg()), dim3(kTileHistThreads), 0, s, (const uint4*)evL, nL, (const uint4*)evR, nR, g,
             sp, sc.P, sc.T, (const uint32_t*)warp_xy, (uint4*)sc.ranges);
  } else {
    launch_k(k_tile_hist<false>, dim3(sp.nseg()), dim3(kTileHistThreads), 0, s, (const uint4*)evL, nL, (const uint4*)evR, nR, g,
             sp, sc.P, sc.T, (const uint32_t*)nullptr, (uint4*)sc.ranges);
  }
}

void launch_tile_scan(hipStream_t s, uint32_t nL, uint32_t nR, const TileGeom& g, const TileScratch& sc,
                      unsigned long long* n_rejected) {
  if (!(nL + nR)) return;
  const TileSplit sp = tile_split(nL, nR);
  static const int force_wide = getenv("ESVIO_FE_WIDE_RECORDS") ? 1 : 0;  // (A/B and tests)
  static_assert(kTileMaxGroups <= (uint32_t)(kTileScanPer * kTileScanRanges), "k_tile_scan covers every group");
  launch_k(k_tile_scan, dim3((g.nbins + kTileScanBins - 1) / kTileScanBins + 1), dim3(kTileScanThreads), 0, s, (const uint32_t*)sc.T, sp,
           g.nbins, g.nt_cam, sc.C, sc.totals, n_rejected, (const uint4*)sc.ranges, sc.meta, force_wide);
}

// Stable partition by bucket.  A scatter block of 4 waves owns TE = 256 * ROUNDS consecutive events of one camera;
// wave w takes them in rounds of 64 consecutive events and ranks every event inside its (wave, bucket)
// pair: rank = events of the bucket this wave has seen in earlier rounds + lanes BELOW this one that
// hold the same bucket in this round — stream order is kept.  "Which lanes hold my bucket" used to be
// a match-any by ballots over the 11 bucket bits: ~190 VALU instructions per round, and with 6.7 M
// events that alone is 32 us of the whole device's issue slots (the kernel took 85-97 us with 3 waves
// per SIMD; in-kernel timers: 20-29 of a block's 27-36 us in the ranking).  Now the lanes tell each
// other through LDS: every lane ORs its bit into mask[wave][bucket] (one ds_or per half-wave — the
// order in which the hardware applies them does not matter, OR commutes), reads the word back (the
// LDS executes a wave's instructions in order) and has the set of lanes that share its bucket; the
// lowest of them adds the group to the wave's running count and clears the mask again.  ~25
// instructions per round.  32-bit masks, so a round is two half-rounds of 32 lanes.
// LDS: one 8-byte entry per (wave, bucket) — {mask, running count (low half) | the wave's offset inside the scatter
// block's run of the bucket (high half)} — the half-round reads it with one ds_read_b64 and the group's lowest lane
// writes it back with one ds_write_b64 — + the bases: 36 B per bucket of ONE camera (33 KiB at 921: round 6; both
// cameras' buckets, 65 KiB, before).  Nothing is cleared between scatter blocks: the masks are zero again after
// every round, and the step that turns the four waves' counts into their offsets puts the counts back to zero
// (round 6: before, a pass over the whole table and a barrier per scatter block).  Each workgroup walks over scatter
// blocks of one camera; the bucket starts are computed once per workgroup.
constexpr int kTileScatterMaxGrid = 768;  // (3 resident blocks per CU: 128 VGPRs, 33 KiB of LDS at 921 buckets; 512 / 640 / 896 / 1024 measured slower)
template <int ROUNDS, bool MC>
__global__ __launch_bounds__(kTileScatterThreads) void k_tile_scatter(
    const uint4* __restrict__ evL, uint32_t nL, const uint4* __restrict__ evR, uint32_t nR, TileGeom g,
    TileSplit sp, uint32_t gridL, const uint32_t* __restrict__ Pm, const uint32_t* __restrict__ Cm,
    const uint32_t* __restrict__ totals, uint4* __restrict__ part, uint32_t* __restrict__ tile_off,
    uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ warp_xy, const uint32_t* __restrict__ meta) {
  // LDS, sized by one camera's bucket count (nbcp = nt_cam + 1 rounded up to 64): the (wave, bucket) entries
  // [4][nbcp] 8 B | bases [nbcp] u32 (first position of the current scatter block per bucket); the entries' space
  // first serves the scan of ALL the batch's bucket totals (nb words <= 8 nbcp)
  extern __shared__ __attribute__((aligned(16))) uint32_t scatter_lds[];
  const int nb = g.nbins, nbc = g.nt_cam + 1, nbcp = (nbc + 63) & ~63;
  uint2* mc_s = (uint2*)scatter_lds;                    // [wave * nbcp + bucket]
  uint32_t* bin_base = scatter_lds + 8 * nbcp;
  __shared__ uint32_t wave_tot[4];
  constexpr int TE = kTileScatterThreads * ROUNDS;
  constexpr int KB = kTileMaxBins / kTileScatterThreads;  // bins per thread (8)
  const int wave = threadIdx.x >> 6, lane = lane_id();
  // this workgroup's camera: the first gridL workgroups take the left camera's scatter blocks
  const bool right = blockIdx.x >= gridL;
  const uint4* __restrict__ ev = right ? evR : evL;
  const uint32_t ncam = right ? nR : nL, ioff = right ? nL : 0u;
  const uint32_t brow0 = right ? sp.nblkL : 0u, nblk_cam = right ? sp.nblkR : sp.nblkL, seg0 = right ? sp.nsegL : 0u;
  uint32_t bstart[KB];  // offsets into `part` of this camera's buckets threadIdx.x + k * 256 (bucket nt_cam: the out-of-sensor bin)
  // ---- once per workgroup: exclusive scan of the batch's bucket totals (thread t owns bins [t*KB, t*KB+KB))
  {
    uint32_t* all_off = scatter_lds;  // [nb], before the entries live there
    uint32_t loc[KB], sum = 0;
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int b = threadIdx.x * KB + k;
      loc[k] = b < nb ? totals[b] : 0u;
      sum += loc[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; w++) run += wave_tot[w];
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int b = threadIdx.x * KB + k;
      if (b < nb) all_off[b] = run;
      if (blockIdx.x == 0 && b <= nb) tile_off[b] = run;  // (b == nb: the total)
      run += loc[k];
    }
    if (blockIdx.x == 0) {
      // the order k_tile_apply takes the buckets in: largest size class (floor(log2(events))) first,
      // so that a bucket with many times the average number of events starts at once and the
      // launch does not end with it (counting sort by class; the order inside a class is arbitrary)
      __shared__ uint32_t cls_cnt[33];
      if (threadIdx.x < 33) cls_cnt[threadIdx.x] = 0;
      __syncthreads();
      const int nt = nb - 1;
#pragma unroll
      for (int k = 0; k < KB; k++) {
        const int b = threadIdx.x * KB + k;
        if (b < nt) atomicAdd(&cls_cnt[loc[k] ? 32 - __clz(loc[k]) : 0], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {  // exclusive prefix, largest class first
        uint32_t acc = 0;
        for (int c = 32; c >= 0; c--) {
          const uint32_t t = cls_cnt[c];
          cls_cnt[c] = acc;
          acc += t;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < KB; k++) {
        const int b = threadIdx.x * KB + k;
        if (b < nt) tile_order[atomicAdd(&cls_cnt[loc[k] ? 32 - __clz(loc[k]) : 0], 1u)] = (uint32_t)b;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int d = threadIdx.x + k * kTileScatterThreads;
      bstart[k] = d < nbc ? all_off[d < g.nt_cam ? (right ? g.nt_cam : 0) + d : nb - 1] : 0u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nbcp; i += kTileScatterThreads) ((uint4*)mc_s)[i] = make_uint4(0, 0, 0, 0);
  }
  // ---- the workgroup's scatter blocks.  Workgroups go to the 8 XCDs round-robin and each XCD has its own
  // L2; the runs two consecutive scatter blocks write into a bucket are neighbours in memory (a few
  // records each, less than a cache line), so neighbours should meet in ONE L2: XCD x takes the x-th
  // contiguous eighth of the camera's scatter blocks, its workgroups take them interleaved.  (From 16 workgroups on
  // gridL is a multiple of 8, so a workgroup's index inside its camera keeps its XCD; below that the split is still a
  // partition of the blocks, only without the affinity.)
  const uint32_t wg = blockIdx.x - (right ? gridL : 0u), nwg = right ? gridDim.x - gridL : gridL;
  uint32_t t_first, t_end, t_step;
  if (nwg < 8) {
    t_first = wg;
    t_end = nblk_cam;
    t_step = nwg;
  } else {
    const uint32_t x = wg & 7u, j = wg >> 3, q = nblk_cam >> 3, r = nblk_cam & 7u;
    t_step = (nwg - x + 7u) >> 3;  // workgroups of this camera on this XCD
    t_first = x * q + min(x, r) + j;
    t_end = (x + 1u) * q + min(x + 1u, r);
  }
  const bool compact = meta[kTileMetaCompact] != 0;  // (uniform)
  const uint32_t sec_base = meta[kTileMetaSecBase];
  const int twsh = g.tw == 64 ? 6 : 5;
  const uint32_t hl = (uint32_t)lane & 31u, hbit = 1u << hl, hlt = hbit - 1u;
  // The records (and count-matrix rows) of the workgroup's NEXT scatter block are requested while the
  // current one is ranked and stored; the barriers inside the loop order LDS traffic only
  // (lds_barrier), so nothing in it waits for HBM except the first use of a record.  (In-kernel
  // timers before: 3.5-7 us of a scatter block's 11-23 us were the wait for its own loads at the first
  // __syncthreads.)
  uint4 nrec[ROUNDS];
  uint32_t npc[KB];
  auto request = [&](uint32_t tile) {  // (tile: scatter block inside the camera)
    const uint32_t wb = tile * TE + wave * (TE / 4);
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const uint32_t i = wb + r * 64 + lane;
      nrec[r] = i < ncam ? ev[i] : make_uint4(0xffffffffu, 0, 0, 0);
      if (MC && i < ncam) nrec[r].x = warp_xy[ioff + i];  // (the pixel k_mc_warp warped the event to)
    }
    const uint32_t* Prow = Pm + (size_t)(brow0 + tile) * nbc;
    const uint32_t* Crow = Cm + (size_t)(seg0 + tile / sp.group) * nbc;
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int d = threadIdx.x + k * kTileScatterThreads;
      npc[k] = d < nbc ? Crow[d] + Prow[d] : 0u;
    }
  };
  if (t_first < t_end) request(t_first);
  __syncthreads();  // (the set-up above)
  for (uint32_t tile = t_first; tile < t_end; tile += t_step) {
    uint4 rec[ROUNDS];
    uint32_t dr[ROUNDS];  // bucket | rank inside (wave, bucket) << 16
    uint32_t pc[KB];
    const uint32_t wbase = tile * TE + wave * (TE / 4);
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) rec[r] = nrec[r];
#pragma unroll
    for (int k = 0; k < KB; k++) pc[k] = npc[k];
    if (tile + t_step < t_end) request(tile + t_step);
    // (this wave's entries: counts zero since the previous scatter block's offsets step, the offsets in their high
    // halves — which this wave's stores of that block have read by now, program order — ride along untouched)
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const uint32_t i = wbase + r * 64 + lane;
      const bool ok = i < ncam;
      const uint32_t d = tile_bin(g, rec[r].x);
      uint32_t rank = 0;
#ifndef ESVIO_ABL_NORANK  // (measurement builds, tools/build_variant.sh + tools/scatter_ablation.sh: what the kernel costs without its parts)
#pragma unroll
      for (int half = 0; half < 2; half++) {
        if (ok && (lane >> 5) == half) {
          uint2* q = mc_s + wave * nbcp + d;
          atomicOr(&q->x, hbit);  // (no return value: ds_or_b32; the LDS takes a wave's instructions in
          const uint2 v = *q;      //  order: the read sees every lane's bit)
          rank = (v.y & 0xffffu) + __popc(v.x & hlt);
          // the group's lowest lane: count it, clear the mask for the next round
          if ((v.x & hlt) == 0) *q = make_uint2(0u, v.y + __popc(v.x));
        }
        // (the other half's ds_or must be ISSUED after these writes — the LDS then performs them in that
        // order; nothing has to wait for them)
        __asm__ volatile("" ::: "memory");
      }
#endif
      dr[r] = d | (rank << 16);
    }
    lds_barrier();  // every wave's counts are final — and every wave's stores of the previous scatter block have read bin_base
    // first position of (this block, wave, bucket): bucket start + earlier groups + earlier blocks of
    // the group (bin_base) + earlier waves of the block (the entry's high half); the counts go back to zero
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int d = threadIdx.x + k * kTileScatterThreads;
      if (d < nbc) {
        bin_base[d] = bstart[k] + pc[k];
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const uint32_t t = mc_s[w * nbcp + d].y & 0xffffu;
          mc_s[w * nbcp + d].y = acc << 16;
          acc += t;
        }
      }
    }
    lds_barrier();
    if (compact) {
      uint2* __restrict__ part2 = (uint2*)part;
#pragma unroll
      for (int r = 0; r < ROUNDS; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < ncam) {
          const uint32_t d = dr[r] & 0xffffu;
          const uint32_t pos = bin_base[d] + (mc_s[wave * nbcp + d].y >> 16) + (dr[r] >> 16);
#if defined(ESVIO_ABL_NOWRITE)
          if (pos == 0xffffffffu)  // (never: the address is still computed)
#endif
#if defined(ESVIO_ABL_COALESCED)
          part2[ioff + i + (pos >> 31)] = tile_rec_pack(rec[r], g.tw, g.th, twsh, sec_base);
#else
          part2[pos] = tile_rec_pack(rec[r], g.tw, g.th, twsh, sec_base);
#endif
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < ROUNDS; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < ncam) {
          const uint32_t d = dr[r] & 0xffffu;
          part[bin_base[d] + (mc_s[wave * nbcp + d].y >> 16) + (dr[r] >> 16)] = rec[r];
        }
      }
    }
  }
}

void launch_tile_scatter(hipStream_t s, const EventRec* evL, uint32_t nL, const EventRec* evR, uint32_t nR,
                         const TileGeom& g, const TileScratch& sc, EventRec* part, const uint32_t* warp_xy) {
  const uint32_t n = nL + nR;
  if (!n) return;
  const TileSplit sp = tile_split(nL, nR);
  // workgroups per camera in proportion to its scatter blocks; the left camera's share a multiple of 8 when both have
  // some (the hardware deals workgroups to the 8 XCDs round-robin by index: a workgroup's index inside its camera
  // then names the same XCD)
  const uint32_t grid = std::min<uint32_t>(sp.nblk(), kTileScatterMaxGrid);
  uint32_t gridL = sp.nblkR == 0 ? grid : sp.nblkL == 0 ? 0u : (uint32_t)((uint64_t)grid * sp.nblkL / sp.nblk());
  if (sp.nblkL && sp.nblkR) {
    if (grid >= 16) gridL = std::min(std::max(8u, (gridL + 4u) & ~7u), grid - 8u);
    else gridL = std::min(std::max(1u, gridL), grid - 1u);
  }
  const unsigned lds = (unsigned)(((g.nt_cam + 1 + 63) & ~63) * 36);
#define ESVIO_TILE_SCATTER(R, M)                                                                          \
  launch_k(k_tile_scatter<R, M>, dim3(grid), dim3(kTileScatterThreads), lds, s, (const uint4*)evL, nL,    \
           (const uint4*)evR, nR, g, sp, gridL, (const uint32_t*)sc.P, (const uint32_t*)sc.C,             \
           (const uint32_t*)sc.totals, (uint4*)part, sc.tile_off, sc.tile_order, warp_xy, (const uint32_t*)sc.meta)
  static_assert(kTileScatterEvents == 8 * kTileScatterThreads, "8 rounds of 64 events per wave");
  if (warp_xy) ESVIO_TILE_SCATTER(8, true);
  else ESVIO_TILE_SCATTER(8, false);
#undef ESVIO_TILE_SCATTER
}

// One block per bucket.  The tile's {L[0],L[1]} sit in LDS; the bucket's events are taken in turns
// of kTileTurn chunks of 64, one wave per turn, waves taking the turns round-robin.  What an event
// needs is the time of the nearest earlier event of its own and of the other polarity at its pixel
// (that IS L[p] / L[!p] when the sequential loop reaches it): inside the chunk by a match-any on the
// pixel (ballots) and a lane shuffle, before the chunk from LDS.  The LDS reads and the L[p] = t
// writes of a chunk's last event per (pixel, polarity) must follow those of all earlier chunks: a
// ticket in LDS orders the turns — a wave does the reads and writes of its kTileTurn chunks back to
// back while it holds the ticket (LDS executes a wave's accesses in order), everything else
// (loads, ballots, the pass rule) runs unordered.  S is the time of the pixel's last PASSING event,
// kept as the largest passing position (LDS atomic max, any order) and turned into a time at the end.
// Only touched pixels are written back.
constexpr int kTileTurn = 4;
template <int kTileApplyThreads, int kPixBits, bool COMPACT>
__device__ __forceinline__ void tile_apply_body(
    const void* __restrict__ part, const uint32_t* __restrict__ tile_off,
    const uint32_t* __restrict__ tile_order, const TileGeom& g, double2* __restrict__ L2, double2* __restrict__ S2,
    double thr, uint8_t* __restrict__ arc_touched, int* __restrict__ err, uint32_t spin_limit, uint32_t sec_base) {
  typedef TileRec<COMPACT> Rec;
  // LDS (dynamic, sized by the tile): Ls[npx] double2 | Sidx[2 npx], per (pixel, polarity): 0 = no event of the
  // bucket, 1 = events but none that passed, 2 + position of the last passing event = 24 B per pixel: 12 KiB for
  // 32x16, 48 KiB for 64x32.  (Round 6: "touched" was a bitmap of its own, set by a second LDS atomic per event —
  // the ablation, profiles/r06_sae_chain_ablations.md, put the two atomics at 12 us of the kernel's 55-72.)
  extern __shared__ double2 tile_lds[];
  const int npx = g.tw * g.th;
  double2* Ls = tile_lds;
  uint32_t* Sidx = (uint32_t*)(Ls + npx);
  __shared__ uint32_t s_done;
  const uint32_t tile = tile_order[blockIdx.x];
  const uint32_t beg = tile_off[tile], end = tile_off[tile + 1];
  if (beg == end) return;
  const uint32_t cam = tile >= (uint32_t)g.nt_cam ? 1u : 0u;
  const uint32_t tt = tile - cam * (uint32_t)g.nt_cam;
  const int ty = (int)(tt / (uint32_t)g.tiles_x), tx = (int)(tt - (uint32_t)ty * (uint32_t)g.tiles_x);
  const int x0 = tx * g.tw, y0 = ty * g.th;
  const int twsh = g.tw == 64 ? 6 : 5;
  const size_t P = (size_t)g.W * g.H;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();  // (wave: scalar, so are the turn loop and the ticket poll)
  // (a wave's next turn is requested while it works on the current one; the first one together with
  // the tile's planes)
  typename Rec::T nxt[kTileTurn];
#pragma unroll
  for (int k = 0; k < kTileTurn; k++) {
    const uint32_t i = beg + ((uint32_t)wave * kTileTurn + k) * 64u + (uint32_t)lane;
    nxt[k] = Rec::load(part, i < end ? i : beg);
  }
  for (int p = threadIdx.x; p < npx; p += kTileApplyThreads) {
    const int gx = x0 + (p & (g.tw - 1)), gy = y0 + (p >> twsh);
#ifdef ESVIO_ABL_APPLY_NOLOAD
    Ls[p] = make_double2((double)gx, (double)gy);
#else
    Ls[p] = (gx < g.W && gy < g.H) ? L2[cam * P + (size_t)gy * g.W + gx] : make_double2(0, 0);
#endif
    Sidx[2 * p] = 0;
    Sidx[2 * p + 1] = 0;
  }
  if (threadIdx.x == 0) s_done = 0;
  __syncthreads();
  constexpr int NW = kTileApplyThreads / 64;
  const uint32_t nchunks = (end - beg + 63u) / 64u;
  const uint32_t nturns = (nchunks + kTileTurn - 1) / kTileTurn;
  const unsigned long long self = 1ull << lane, below = self - 1ull, above = ~(below | self);
  auto msb = [](unsigned long long m) { return 63 - __clzll((long long)m); };
  for (uint32_t turn = wave; turn < nturns; turn += NW) {
    uint32_t pixk[kTileTurn];
    double tk[kTileTurn], ts_in[kTileTurn], to_in[kTileTurn];
    uint32_t flg[kTileTurn];  // 1 valid, 2 polarity, 4 has same-polarity predecessor in the chunk, 8 other, 16 last
    typename Rec::T cur[kTileTurn];
#pragma unroll
    for (int k = 0; k < kTileTurn; k++) {
      cur[k] = nxt[k];
      const uint32_t i = beg + ((turn + NW) * kTileTurn + k) * 64u + (uint32_t)lane;
      nxt[k] = Rec::load(part, i < end ? i : beg);
    }
#pragma unroll
    for (int k = 0; k < kTileTurn; k++) {
      const uint32_t i = beg + (turn * kTileTurn + k) * 64u + (uint32_t)lane;
      const bool valid = i < end;
      const uint32_t pix = Rec::pix(cur[k], x0, y0, twsh);
      const bool pol = Rec::pol(cur[k]);
      const double t = Rec::time(cur[k], sec_base);
      unsigned long long m = __ballot(valid);
#ifdef ESVIO_ABL_APPLY_NOMATCH  // (measurement builds, WRONG results)
      m &= self;
      if (false)
#endif
#pragma unroll
      for (int b = 0; b < kPixBits; b++) {
        const int sx = ((int)(pix << (31 - b))) >> 31;  // -1 where the bit is set, else 0
        const unsigned long long bal = __ballot(sx != 0);
        m &= ~(bal ^ (unsigned long long)(long long)sx);  // lanes with the bit: bal, the others: ~bal
      }
      const unsigned long long m1 = __ballot(valid && pol);
      const unsigned long long same = m & (pol ? m1 : ~m1), opp = m & (pol ? ~m1 : m1);
      const unsigned long long bs = same & below, bo = opp & below;
      ts_in[k] = __shfl(t, bs ? msb(bs) : lane);
      to_in[k] = __shfl(t, bo ? msb(bo) : lane);
      pixk[k] = valid ? pix : 0u;
      tk[k] = t;
      flg[k] = (valid ? 1u : 0u) | (pol ? 2u : 0u) | (bs ? 4u : 0u) | (bo ? 8u : 0u) |
               ((valid && (same & above) == 0) ? 16u : 0u);
    }
    // ---- ordered part: after every earlier turn's.  Everything it needs is in registers before the
    // wait (exec masks of the reads and writes, LDS addresses: the compiler otherwise computes
    // addresses and predicates while the wave holds the ticket — in-kernel timers on the scene stream's
    // largest buckets, 35 k events: 26 of a wave's 62 us were waiting for the ticket, ~0.45 us per
    // hand-off), and the poll is wave-uniform: every lane reads the word, scalar branch.
    static_assert(kTileTurn == 4, "the ordered section is written for 4 chunks");
    typedef uint32_t lds_v4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) char lds_char;
    const uint32_t ls0 = (uint32_t)(size_t)(lds_char*)(char*)Ls, sd0 = (uint32_t)(size_t)(lds_char*)(char*)&s_done;
    unsigned long long rm[kTileTurn], wm[kTileTurn], svx;
    uint32_t ra[kTileTurn], wa[kTileTurn];
#pragma unroll
    for (int k = 0; k < kTileTurn; k++) {
      rm[k] = __ballot((flg[k] & 1u) && (flg[k] & 12u) != 12u);
      wm[k] = __ballot((flg[k] & 16u) != 0);
      ra[k] = ls0 + (pixk[k] << 4);
      wa[k] = ra[k] + ((flg[k] & 2u) ? 8u : 0u);
      __asm__ volatile("" : "+s"(rm[k]), "+s"(wm[k]), "+v"(ra[k]), "+v"(wa[k]));
    }
    bool gave_up = false;
#ifdef ESVIO_ABL_APPLY_NOTICKET  // (measurement build, WRONG results: the turns' LDS sections in no particular order)
    if (false)
#endif
    for (uint32_t spins = 0;;) {
      const uint32_t d =
          __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (d == turn) break;
      // (bounded: never hang the GPU.  A wave that gives up poisons the ticket, so that the block's
      // other waves give up at once instead of timing out one turn after the other; the host sees
      // *err and fails the call)
      if (d == 0xffffffffu || ++spins > spin_limit) {  // (kSpinTicket polls)
        gave_up = true;
        if (lane == 0) {
          *err = 2;
          __hip_atomic_store(&s_done, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        break;
      }
      // (waves whose turn is not the next one poll slowly: the LDS pipe is the ticket holder's)
      if (turn - d > 1u) __builtin_amdgcn_s_sleep(1);
    }
    // While the wave holds the ticket it issues nothing but the LDS instructions themselves: masks
    // and addresses are operands, exec is switched by hand (reads: lanes that need the value from
    // before the chunk; writes: the chunk's last event per (pixel, polarity)); the LDS executes a
    // wave's instructions in order, so the ticket (last) becomes visible after the accesses, and the
    // wait for the values read comes after it.
    lds_v4 fv[kTileTurn];
#pragma unroll
    for (int k = 0; k < kTileTurn; k++) fv[k] = (lds_v4)(0u);
    const unsigned long long l0m = __ballot(lane == 0 && !gave_up);  // (exec for the ticket store)
    const uint32_t nxt_turn = turn + 1u;
    __asm__ volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[r0]\n\tds_read_b128 %[f0], %[ra0]\n\t"
        "s_mov_b64 exec, %[w0]\n\tds_write_b64 %[wa0], %[t0]\n\t"
        "s_mov_b64 exec, %[r1]\n\tds_read_b128 %[f1], %[ra1]\n\t"
        "s_mov_b64 exec, %[w1]\n\tds_write_b64 %[wa1], %[t1]\n\t"
        "s_mov_b64 exec, %[r2]\n\tds_read_b128 %[f2], %[ra2]\n\t"
        "s_mov_b64 exec, %[w2]\n\tds_write_b64 %[wa2], %[t2]\n\t"
        "s_mov_b64 exec, %[r3]\n\tds_read_b128 %[f3], %[ra3]\n\t"
        "s_mov_b64 exec, %[w3]\n\tds_write_b64 %[wa3], %[t3]\n\t"
        "s_mov_b64 exec, %[l0]\n\tds_write_b32 %[sd], %[nx]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_nop 4\n\t"  // (exec written by the SALU: 5 wait states before a DPP instruction may follow)
        "s_waitcnt lgkmcnt(0)"
        : [sv] "=&s"(svx), [f0] "+v"(fv[0]), [f1] "+v"(fv[1]), [f2] "+v"(fv[2]), [f3] "+v"(fv[3])
        : [r0] "s"(rm[0]), [r1] "s"(rm[1]), [r2] "s"(rm[2]), [r3] "s"(rm[3]), [w0] "s"(wm[0]), [w1] "s"(wm[1]),
          [w2] "s"(wm[2]), [w3] "s"(wm[3]), [l0] "s"(l0m), [ra0] "v"(ra[0]), [ra1] "v"(ra[1]), [ra2] "v"(ra[2]),
          [ra3] "v"(ra[3]), [wa0] "v"(wa[0]), [wa1] "v"(wa[1]), [wa2] "v"(wa[2]), [wa3] "v"(wa[3]), [t0] "v"(tk[0]),
          [t1] "v"(tk[1]), [t2] "v"(tk[2]), [t3] "v"(tk[3]), [sd] "v"(sd0), [nx] "v"(nxt_turn)
        : "memory");
    double2 fb[kTileTurn];
#pragma unroll
    for (int k = 0; k < kTileTurn; k++) {
      fb[k].x = __hiloint2double((int)fv[k].y, (int)fv[k].x);
      fb[k].y = __hiloint2double((int)fv[k].w, (int)fv[k].z);
    }
    // ---- unordered again
#pragma unroll
    for (int k = 0; k < kTileTurn; k++) {
      const bool pol = (flg[k] & 2u) != 0;
      const double prev_same = (flg[k] & 4u) ? ts_in[k] : (pol ? fb[k].y : fb[k].x);
      const double prev_opp = (flg[k] & 8u) ? to_in[k] : (pol ? fb[k].x : fb[k].y);
      const bool pass = (flg[k] & 1u) && ((tk[k] > __dadd_rn(prev_same, thr)) || (prev_opp > prev_same));
      // one LDS atomic per event that matters: the chunk's last event of a (pixel, polarity) marks the pair as
      // touched (1), a passing event leaves 2 + its position; the maximum keeps the last passing one
      const uint32_t pos = (turn * kTileTurn + k) * 64u + (uint32_t)lane;
#ifdef ESVIO_ABL_APPLY_NOPOST
      if ((pass || (flg[k] & 16u)) && pos == 0xffffffffu)
#else
      if (pass || (flg[k] & 16u))
#endif
        atomicMax(&Sidx[2 * pixk[k] + (pol ? 1 : 0)], pass ? pos + 2u : 1u);
    }
  }
  __syncthreads();
  // write-back of the touched pixels.  The records of the last passing events (for their times) of all
  // of a thread's pixels are requested together (a loop with a run-time bound fetched them one pixel
  // per round trip), and S is written per polarity: nothing is read from the planes here.
  constexpr int kPer = kTileMaxPx / kTileApplyThreads;  // pixels per thread, at most
  typename Rec::T r0[kPer], r1[kPer];
  uint32_t sb[kPer];  // touched bits | (has S[0]) << 2 | (has S[1]) << 3
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int p = threadIdx.x + q * kTileApplyThreads;
    sb[q] = 0;
    if (p < npx) {
      const uint32_t s0 = Sidx[2 * p], s1 = Sidx[2 * p + 1];
      sb[q] = (s0 ? 1u : 0u) | (s1 ? 2u : 0u) | (s0 >= 2u ? 4u : 0u) | (s1 >= 2u ? 8u : 0u);
      r0[q] = Rec::load(part, (sb[q] & 4u) ? beg + s0 - 2u : beg);
      r1[q] = Rec::load(part, (sb[q] & 8u) ? beg + s1 - 2u : beg);
    }
  }
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int p = threadIdx.x + q * kTileApplyThreads;
    const uint32_t tb = sb[q] & 3u;
#ifdef ESVIO_ABL_APPLY_NOWB
    if (p >= npx || !tb || tb != 7u) continue;
#else
    if (p >= npx || !tb) continue;
#endif
    const int gx = x0 + (p & (g.tw - 1)), gy = y0 + (p >> twsh);
    const size_t k = cam * P + (size_t)gy * g.W + gx;
    L2[k] = Ls[p];
    // k_arc_mark's job for a batch whose Arc* pass is coming: the LEFT camera's touched
    // (pixel, polarity) flags, here from the tile's own bookkeeping instead of one scattered byte
    // store per event
    if (arc_touched && cam == 0)
      ((uint16_t*)arc_touched)[k] = (uint16_t)((tb & 1u) | ((tb & 2u) << 7));
    if (sb[q] & 4u) ((double*)&S2[k])[0] = Rec::time(r0[q], sec_base);
    if (sb[q] & 8u) ((double*)&S2[k])[1] = Rec::time(r1[q], sec_base);
  }
}

template <int kTileApplyThreads, int kPixBits>
__global__ __launch_bounds__(kTileApplyThreads) void k_tile_apply(
    const uint4* __restrict__ part, const uint32_t* __restrict__ tile_off,
    const uint32_t* __restrict__ tile_order, TileGeom g, double2* __restrict__ L2, double2* __restrict__ S2,
    double thr, uint8_t* __restrict__ arc_touched, int* __restrict__ err, uint32_t spin_limit,
    const uint32_t* __restrict__ meta) {
  // (the record format of this batch's partition, decided by k_tile_scan: uniform)
  if (meta[kTileMetaCompact])
    tile_apply_body<kTileApplyThreads, kPixBits, true>(part, tile_off, tile_order, g, L2, S2, thr, arc_touched, err,
                                                       spin_limit, meta[kTileMetaSecBase]);
  else
    tile_apply_body<kTileApplyThreads, kPixBits, false>(part, tile_off, tile_order, g, L2, S2, thr, arc_touched, err,
                                                        spin_limit, 0u);
}

void launch_tile_apply(hipStream_t s, const EventRec* part, uint32_t n, const TileGeom& g, const TileScratch& sc,
                       double2* L2, double2* S2, double filter_threshold, uint8_t* arc_touched, int* err,
                       uint32_t spin_limit) {
  const int npx = g.tw * g.th;
  const unsigned lds = (unsigned)(npx * 24 + 16);
  // a turn is 256 events: with a few turns per bucket 4 waves are plenty and many blocks fit a CU;
  // with thousands of events per bucket 8 waves (measured at 6.7 M events, 1841 buckets: 73 us with
  // 512 threads, 82 with 256, 112 with 1024; turns of 2 / 4 / 8 chunks at 512 threads: 90 / 73 / 111)
  const bool big = n / (uint32_t)(2 * g.nt_cam) >= 2048u;
#define ESVIO_TILE_APPLY(T, B)                                                                            \
  launch_k(k_tile_apply<T, B>, dim3(2 * g.nt_cam), dim3(T), lds, s, (const uint4*)part,                  \
           (const uint32_t*)sc.tile_off, (const uint32_t*)sc.tile_order, g, L2, S2, filter_threshold,     \
           arc_touched, err, spin_limit, (const uint32_t*)sc.meta)
#define ESVIO_TILE_APPLY_B(B)           \
  if (big) ESVIO_TILE_APPLY(512, B);    \
  else ESVIO_TILE_APPLY(256, B)
  if (g.pix_bits == 9) {
    ESVIO_TILE_APPLY_B(9);
  } else if (g.pix_bits == 10) {
    ESVIO_TILE_APPLY_B(10);
  } else {
    ESVIO_TILE_APPLY_B(11);
  }
#undef ESVIO_TILE_APPLY_B
#undef ESVIO_TILE_APPLY
}

// ============================================================================ time-slice composition
// One stream's batch cut into N time slices, one per GPU.  A slice's effect on
// the planes is exchanged as planes holding kSliceNone where the slice wrote nothing; the planes
// after the whole batch are the pre-batch planes overlaid with the slices in stream order.
__global__ __launch_bounds__(256) void k_fill_f64(double* __restrict__ p, size_t n, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ __launch_bounds__(256) void k_overlay_f64(double* __restrict__ dst, const double* __restrict__ src,
                                                     size_t n, double none) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double v = src[i];
    if (v != none) dst[i] = v;
  }
}

// one wave that does nothing for `ticks` of the 100 MHz wall clock (esvio_fe_create's look at which of the handle's
// streams the runtime has put on one hardware queue)
__global__ void k_spin(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
void launch_spin(hipStream_t s, unsigned long long ticks) { launch_k(k_spin, dim3(1), dim3(64), 0, s, ticks); }
__global__ void k_set_u32(uint32_t* p, uint32_t v) {
  if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_set_u32(hipStream_t s, uint32_t* p, uint32_t v) { launch_k(k_set_u32, dim3(1), dim3(64), 0, s, p, v); }

void launch_fill_f64(hipStream_t s, double* p, size_t n, double v) {
  if (!n) return;
  launch_k(k_fill_f64, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, p, n, v);
}

void launch_overlay_f64(hipStream_t s, double* dst, const double* src, size_t n, double none) {
  if (!n) return;
  launch_k(k_overlay_f64, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, dst, src, n,
           none);
}

// ============================================================================ time surface
// SAEtoTimeSurface_left/right (event_detector.cc:230-305): one 16 B {S0,S1} read and one u8 write
// per pixel.  u8 = saturate_cast<uchar>(cvRound(v*127.5+127.5)) (ignore_polarity: v*255+0).
__device__ __forceinline__ uint8_t ts_pixel(double2 s, double t_sync, double decay_sec,
                                            int ignore_polarity) {
  const bool pos = s.y > s.x;
  const double m = pos ? s.y : s.x;
  double v = 0.0;
  if (m > 0) {
    const double dt = __dsub_rn(t_sync, m);
#if defined(ESVIO_ABL_TS_NOEXP)  // (measurement build: the render without its exp and its division)
    v = dt * decay_sec;
#elif defined(ESVIO_ABL_TS_NODIV)  // (... with a multiplication by 1 / decay instead of the division: NOT the reference's rounding)
    v = exp(-dt * (1.0 / decay_sec));
#else
    v = exp(-dt / decay_sec);
#endif
    if (!ignore_polarity) v = pos ? v : -v;
  }
  const double sc = ignore_polarity ? __dadd_rn(__dmul_rn(v, 255.0), 0.0)
                                    : __dadd_rn(__dmul_rn(v, 127.5), 127.5);
  // cvRound: SSE2 cvtsd2si -> "integer indefinite" (INT_MIN) when out of int32 range / NaN,
  // which saturate_cast<uchar> then maps to 0 [OpenCV]
  int iv;
  if (!(sc > -2147483648.5 && sc < 2147483647.5)) {
    iv = INT_MIN;
  } else {
    const double r = rint(sc);
    iv = (r > 2147483647.0 || r < -2147483648.0) ? INT_MIN : (int)r;
  }
  return (uint8_t)((unsigned)iv <= 255u ? iv : iv > 0 ? 255 : 0);
}

__global__ __launch_bounds__(256) void k_time_surface(const double2* __restrict__ S2, int W,
                                                      int H, double t_sync, double decay_sec,
                                                      int ignore_polarity,
                                                      uint8_t* __restrict__ dst0,
                                                      uint8_t* __restrict__ dst1, int stride) {
  const int cam = blockIdx.y;
  const uint32_t P = (uint32_t)W * H;
  const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= P) return;
  uint8_t* dst = cam ? dst1 : dst0;
  const double2 s = S2[(size_t)cam * P + px];
  const uint32_t y = px / (uint32_t)W, x = px - y * (uint32_t)W;
  dst[(size_t)(y + kPad) * stride + x + kPad] = ts_pixel(s, t_sync, decay_sec, ignore_polarity);
}

// four consecutive pixels of a row per thread (W a multiple of 4): 64 B of planes in, one dword out — the render
// at its roof: every S2 word read once, every pixel written once, nothing else
__global__ __launch_bounds__(256) void k_time_surface4(const double2* __restrict__ S2, int W, int H, double t_sync,
                                                       double decay_sec, int ignore_polarity,
                                                       uint8_t* __restrict__ dst0, uint8_t* __restrict__ dst1, int stride) {
  const int cam = blockIdx.y;
  const uint32_t P = (uint32_t)W * H;
  const uint32_t px = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  if (px >= P) return;
  uint8_t* dst = cam ? dst1 : dst0;
  const double2* src = S2 + (size_t)cam * P + px;
  double2 sv[4];
#pragma unroll
  for (int k = 0; k < 4; k++) sv[k] = src[k];
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) out |= (uint32_t)ts_pixel(sv[k], t_sync, decay_sec, ignore_polarity) << (8 * k);
  const uint32_t y = px / (uint32_t)W, x = px - y * (uint32_t)W;
  *(uint32_t*)(dst + (size_t)(y + kPad) * stride + x + kPad) = out;  // (kPad and the stride are multiples of 4)
}

KernelId launch_time_surface(hipStream_t s, const double2* S2, int W, int H, double t_sync,
                             double decay_sec, int ignore_polarity, uint8_t* dst0, uint8_t* dst1,
                             int dst_stride, int ncam) {
  const uint32_t P = (uint32_t)W * H;
  static_assert(kPad % 4 == 0, "dword stores into the padded level");
  if (W % 4 == 0 && dst_stride % 4 == 0 && ((uintptr_t)dst0 & 3) == 0 && (ncam < 2 || ((uintptr_t)dst1 & 3) == 0)) {
    launch_k(k_time_surface4, dim3((P / 4 + 255) / 256, ncam), dim3(256), 0, s, S2, W, H, t_sync, decay_sec,
             ignore_polarity, dst0, dst1, dst_stride);
    return K_TIME_SURFACE4;
  }
  launch_k(k_time_surface, dim3((P + 255) / 256, ncam), dim3(256), 0, s, S2, W, H,
                   t_sync, decay_sec, ignore_polarity, dst0, dst1, dst_stride);
  return K_TIME_SURFACE;
}

// ============================================================================ CLAHE + normalize
// cv::createCLAHE()->apply (clipLimit 40, 8x8 tiles) then cv::normalize(.,0,255,NORM_MINMAX) —
// the `equalize: 1` branch of trackEvent (feature_tracker.cpp:375-382) [OpenCV imgproc/clahe.cpp,
// core/norm.cpp].  One block per tile builds the clipped-histogram LUT; one thread per pixel blends
// the four neighbouring LUTs and tracks the image min/max; a third pass rescales to 0..255.
constexpr int kClaheTiles = 8;

struct ClaheArgs {
  const uint8_t* raw[2];  // pixel (0,0) of the raw time surfaces
  int raw_stride;
  uint8_t* dst[2];        // pixel (0,0) of the LK level-0 images
  int dst_stride;
  int W, H;
  uint8_t* lut;           // [nimg][64][256]
  int* minmax;            // [nimg][2]
};

__global__ __launch_bounds__(256) void k_clahe_lut(ClaheArgs a) {
  __shared__ int hist[256];
  __shared__ int part[4];
  const int img = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int W = a.W, H = a.H;
  int EW = W, EH = H;
  if (!(W % kClaheTiles == 0 && H % kClaheTiles == 0)) {
    EW = W + (kClaheTiles - (W % kClaheTiles));
    EH = H + (kClaheTiles - (H % kClaheTiles));
  }
  const int tw = EW / kClaheTiles, th = EH / kClaheTiles, area = tw * th;
  const int ty = tile / kClaheTiles, tx = tile - ty * kClaheTiles;
  hist[tid] = 0;
  if (tile == 0 && tid == 0) {
    a.minmax[2 * img] = 255;
    a.minmax[2 * img + 1] = 0;
  }
  __syncthreads();
  const uint8_t* src = a.raw[img];
  for (int p = tid; p < area; p += 256) {
    const int py = p / tw, px = p - py * tw;
    const int y = reflect101(ty * th + py, H), x = reflect101(tx * tw + px, W);
    atomicAdd(&hist[src[(size_t)y * a.raw_stride + x]], 1);
  }
  __syncthreads();
  int clip = (int)(40.0 * area / 256);
  clip = max(clip, 1);
  int h = hist[tid];
  int over = h > clip ? h - clip : 0;
  h = min(h, clip);
  int v = over;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((tid & 63) == 0) part[tid >> 6] = v;
  __syncthreads();
  const int clipped = part[0] + part[1] + part[2] + part[3];
  const int redistBatch = clipped / 256;
  const int residual = clipped - redistBatch * 256;
  h += redistBatch;
  if (residual != 0) {
    const int step = max(256 / residual, 1);
    if (tid % step == 0 && tid / step < residual) h++;
  }
  // inclusive scan over the 256 bins
  int sum = h;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(sum, o);
    if ((tid & 63) >= o) sum += t;
  }
  __syncthreads();
  if ((tid & 63) == 63) part[tid >> 6] = sum;
  __syncthreads();
  for (int w = 0; w < (tid >> 6); w++) sum += part[w];
  const float lutScale = (float)255 / (float)area;
  const int r = __float2int_rn((float)sum * lutScale);
  a.lut[((size_t)img * 64 + tile) * 256 + tid] = (uint8_t)((unsigned)r <= 255u ? r : r > 0 ? 255 : 0);
}

__global__ __launch_bounds__(256) void k_clahe_interp(ClaheArgs a) {
  const int img = blockIdx.y;
  const int W = a.W, H = a.H;
  int EW = W, EH = H;
  if (!(W % kClaheTiles == 0 && H % kClaheTiles == 0)) {
    EW = W + (kClaheTiles - (W % kClaheTiles));
    EH = H + (kClaheTiles - (H % kClaheTiles));
  }
  const int tw = EW / kClaheTiles, th = EH / kClaheTiles;
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  const uint8_t* lut = a.lut + (size_t)img * 64 * 256;
  int mn = 255, mx = 0;
  // (a bounded grid walks the image: the launch ends with one atomic pair per block on the image's
  // two extreme words, and a few hundred of those are cheap where thousands serialise at L2)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < W * H; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    const float tyf = y * inv_th - 0.5f;
    const float fty = floorf(tyf);
    int ty1 = (int)fty, ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0);
    ty2 = min(ty2, kClaheTiles - 1);
    const float txf = x * inv_tw - 0.5f;
    const float ftx = floorf(txf);
    int tx1 = (int)ftx, tx2 = tx1 + 1;
    const float xa = txf - tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0);
    tx2 = min(tx2, kClaheTiles - 1);
    const int v = a.raw[img][(size_t)y * a.raw_stride + x];
    const float l11 = lut[(ty1 * 8 + tx1) * 256 + v], l12 = lut[(ty1 * 8 + tx2) * 256 + v];
    const float l21 = lut[(ty2 * 8 + tx1) * 256 + v], l22 = lut[(ty2 * 8 + tx2) * 256 + v];
    const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    const int r = __float2int_rn(res);
    const int res8 = (unsigned)r <= 255u ? r : r > 0 ? 255 : 0;
    a.dst[img][(size_t)y * a.dst_stride + x] = (uint8_t)res8;
    mn = min(mn, res8);
    mx = max(mx, res8);
  }
  // block min/max -> at most one atomic pair per block, and none once the image's extremes are
  // known (thousands of atomics on the same two words serialise at L2: 220 us of a 5 us kernel;
  // a stale read of the current extremes only costs a redundant atomic)
  __shared__ int s_mn[4], s_mx[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = min(mn, __shfl_xor(mn, o));
    mx = max(mx, __shfl_xor(mx, o));
  }
  if ((threadIdx.x & 63) == 0) {
    s_mn[threadIdx.x >> 6] = mn;
    s_mx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
    mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    if (mn < __hip_atomic_load(&a.minmax[2 * img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(&a.minmax[2 * img], mn);
    if (mx > __hip_atomic_load(&a.minmax[2 * img + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(&a.minmax[2 * img + 1], mx);
  }
}

__global__ __launch_bounds__(256) void k_normalize(ClaheArgs a) {
  const int img = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.W * a.H) return;
  const double smin = a.minmax[2 * img], smax = a.minmax[2 * img + 1];
  const double scale = 255.0 * (__dsub_rn(smax, smin) > 2.2204460492503131e-16 ? 1. / __dsub_rn(smax, smin) : 0);
  const double shift = __dsub_rn(0.0, __dmul_rn(smin, scale));
  const float fa = (float)scale, fb = (float)shift;
  const int y = i / a.W, x = i - y * a.W;
  uint8_t* p = a.dst[img] + (size_t)y * a.dst_stride + x;
  const int r = __float2int_rn(__fadd_rn(__fmul_rn((float)*p, fa), fb));
  *p = (uint8_t)((unsigned)r <= 255u ? r : r > 0 ? 255 : 0);
}

void launch_clahe(hipStream_t s, const uint8_t* raw0, const uint8_t* raw1, int raw_stride,
                  uint8_t* dst0, uint8_t* dst1, int dst_stride, int W, int H, uint8_t* lut,
                  int* minmax, int nimg, int stage) {
  ClaheArgs a;
  a.raw[0] = raw0;
  a.raw[1] = raw1;
  a.raw_stride = raw_stride;
  a.dst[0] = dst0;
  a.dst[1] = dst1;
  a.dst_stride = dst_stride;
  a.W = W;
  a.H = H;
  a.lut = lut;
  a.minmax = minmax;
  const int nb = (W * H + 255) / 256;
  if (stage == 0)
    launch_k(k_clahe_lut, dim3(64, nimg), dim3(256), 0, s, a);
  else if (stage == 1)
    launch_k(k_clahe_interp, dim3(nb < 256 ? nb : 256, nimg), dim3(256), 0, s, a);
  else
    launch_k(k_normalize, dim3(nb, nimg), dim3(256), 0, s, a);
}

// ============================================================================ pyramid
struct PyrPack {
  PyrDesc p[3];
};

// cv::pyrDown u8 [OpenCV imgproc/pyramids.cpp]: [1 4 6 4 1]x[1 4 6 4 1], (sum+128)>>8,
// BORDER_REFLECT_101, dst size ((w+1)/2,(h+1)/2). Reads the source interior only.
__global__ __launch_bounds__(256) void k_pyr_down(PyrPack pk, int src_level) {
  const PyrDesc& p = pk.p[blockIdx.z];
  const int sw = p.w[src_level], sh = p.h[src_level];
  const int dw = p.w[src_level + 1], dh = p.h[src_level + 1];
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
  const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= dw || y >= dh) return;
  const int sstride = p.stride[src_level], dstride = p.stride[src_level + 1];
  const uint8_t* src = p.img[src_level] + (size_t)kPad * sstride + kPad;
  const int wk[5] = {1, 4, 6, 4, 1};
  int xs[5];
#pragma unroll
  for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x + k - 2, sw);
  int acc = 0;
#pragma unroll
  for (int ky = 0; ky < 5; ky++) {
    const uint8_t* row = src + (size_t)reflect101(2 * y + ky - 2, sh) * sstride;
    int r = 0;
#pragma unroll
    for (int kx = 0; kx < 5; kx++) r += wk[kx] * (int)row[xs[kx]];
    acc += wk[ky] * r;
  }
  p.img[src_level + 1][(size_t)(y + kPad) * dstride + x + kPad] = (uint8_t)((acc + 128) >> 8);
}

static PyrPack make_pack(const PyrDesc* p, int nimg);

// Time surface + the three pyrDown levels of both cameras in ONE launch (instead of
// k_time_surface + 3 x k_pyr_down: each of those is a ~5 us kernel behind a ~3 us launch, all on the
// prefetch stream's dependent chain).  A block owns an 8x4 tile of level 3 = 16x8 of level 2 =
// 32x16 of level 1 = 64x32 of level 0 and recomputes the halo the 5x5 kernels need (85x53 level-0
// pixels rendered per block, x2.2 the owned ones — fp64 exp is cheap next to three launches), level
// by level through LDS.  Per pixel the arithmetic is ts_pixel resp. k_pyr_down's, reflect-101 taken at
// each level's own size, so every byte equals the unfused kernels' output.
constexpr int kFt3x = 8, kFt3y = 4;
constexpr int kFt2x = 2 * kFt3x + 3, kFt2y = 2 * kFt3y + 3;  // 19 x 11
constexpr int kFt1x = 2 * kFt2x + 3, kFt1y = 2 * kFt2y + 3;  // 41 x 25
constexpr int kFt0x = 2 * kFt1x + 3, kFt0y = 2 * kFt1y + 3;  // 85 x 53
constexpr int kFs0 = 88, kFs1 = 44, kFs2 = 20;               // LDS row strides
constexpr int kFtThreads = 1024;  // 16 waves per block: the fp64 exp chains need the latency hiding

template <int SRX, int SS, int DRX, int DRY, int DS>
__device__ __forceinline__ void fused_down(const uint8_t* __restrict__ src_l, uint8_t* __restrict__ dst_l,
                                           int sox, int soy, int sw, int sh, int dox, int doy, int dw,
                                           int dh, uint8_t* __restrict__ dst_img, int dstride, int ownx0,
                                           int owny0, int ownw, int ownh) {
  for (int i = threadIdx.x; i < DRX * DRY; i += kFtThreads) {
    const int ry = i / DRX, rx = i - ry * DRX;
    const int x = dox + rx, y = doy + ry;
    if (x < 0 || y < 0 || x >= dw || y >= dh) continue;
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x + k - 2, sw) - sox;
    const int wk[5] = {1, 4, 6, 4, 1};
    int acc = 0;
#pragma unroll
    for (int ky = 0; ky < 5; ky++) {
      const uint8_t* row = src_l + (reflect101(2 * y + ky - 2, sh) - soy) * SS;
      int r = 0;
#pragma unroll
      for (int kx = 0; kx < 5; kx++) r += wk[kx] * (int)row[xs[kx]];
      acc += wk[ky] * r;
    }
    const uint8_t v = (uint8_t)((acc + 128) >> 8);
    if (dst_l) dst_l[ry * DS + rx] = v;
    if (x >= ownx0 && x < ownx0 + ownw && y >= owny0 && y < owny0 + ownh)
      dst_img[(size_t)(y + kPad) * dstride + x + kPad] = v;
  }
}

// FROM_IMG: level 0 is not rendered from the SAE but is cv::normalize(., 0, 255, NORM_MINMAX) of an
// existing u8 image (the CLAHE output of the `equalize: 1` branch, k_clahe_interp) whose extremes are
// in minmax[2 cam .. 2 cam + 1] — k_normalize's arithmetic per pixel — so that branch, too, goes from
// its level-0 source to the four pyramid levels in one launch.
struct EqSrc {
  const uint8_t* img[2];  // pixel (0,0) of the two un-normalised images
  int stride;
  const int* minmax;
};

// MODE 1 (FROM_IMG): level 0 normalised from an image; 2: level 0 is in the pyramid already (k_time_surface4
// wrote it) — only the three pyrDown levels.  (MODE 0, the surface rendered here from the SAE planes — one launch,
// but every tile re-rendered its halo: 2.2x the planes' bytes and exps — lost to the split form at 1280x720 and
// tied with it at 640x480 in rounds 4 and 5 and is gone.)
template <int MODE>
__device__ __forceinline__ void pyr3_body(const EqSrc& eq, const PyrPack& pk) {
  __shared__ uint8_t l0[kFt0y * kFs0];
  __shared__ uint8_t l1[kFt1y * kFs1];
  __shared__ uint8_t l2[kFt2y * kFs2];
  const int cam = blockIdx.z;
  const PyrDesc& p = pk.p[cam];
  const int W = p.w[0], H = p.h[0];
  const int o3x = blockIdx.x * kFt3x, o3y = blockIdx.y * kFt3y;
  const int o2x = 2 * o3x - 2, o2y = 2 * o3y - 2;
  const int o1x = 2 * o2x - 2, o1y = 2 * o2y - 2;
  const int o0x = 2 * o1x - 2, o0y = 2 * o1y - 2;
  constexpr int kN0 = kFt0x * kFt0y;
  constexpr bool FROM_IMG = MODE == 1;
  if (MODE == 2) {
    const uint8_t* src = p.img[0] + (size_t)kPad * p.stride[0] + kPad;
    for (int i = threadIdx.x; i < kN0; i += kFtThreads) {
      const int ry = i / kFt0x, rx = i - ry * kFt0x;
      const int x = o0x + rx, y = o0y + ry;
      if (x < 0 || y < 0 || x >= W || y >= H) continue;
      l0[ry * kFs0 + rx] = src[(size_t)y * p.stride[0] + x];
    }
  }
  if (FROM_IMG) {
    const double smin = eq.minmax[2 * cam], smax = eq.minmax[2 * cam + 1];
    const double scale = 255.0 * (__dsub_rn(smax, smin) > 2.2204460492503131e-16 ? 1. / __dsub_rn(smax, smin) : 0);
    const float fa = (float)scale, fb = (float)__dsub_rn(0.0, __dmul_rn(smin, scale));
    const uint8_t* src = eq.img[cam];
    for (int i = threadIdx.x; i < kN0; i += kFtThreads) {
      const int ry = i / kFt0x, rx = i - ry * kFt0x;
      const int x = o0x + rx, y = o0y + ry;
      if (x < 0 || y < 0 || x >= W || y >= H) continue;
      const int r = __float2int_rn(__fadd_rn(__fmul_rn((float)src[(size_t)y * eq.stride + x], fa), fb));
      const uint8_t v = (uint8_t)((unsigned)r <= 255u ? r : r > 0 ? 255 : 0);
      l0[ry * kFs0 + rx] = v;
      if (x >= 8 * o3x && x < 8 * o3x + 8 * kFt3x && y >= 8 * o3y && y < 8 * o3y + 8 * kFt3y)
        p.img[0][(size_t)(y + kPad) * p.stride[0] + x + kPad] = v;
    }
  }
  __syncthreads();
  fused_down<kFt0x, kFs0, kFt1x, kFt1y, kFs1>(l0, l1, o0x, o0y, W, H, o1x, o1y, p.w[1], p.h[1], p.img[1],
                                               p.stride[1], 4 * o3x, 4 * o3y, 4 * kFt3x, 4 * kFt3y);
  __syncthreads();
  fused_down<kFt1x, kFs1, kFt2x, kFt2y, kFs2>(l1, l2, o1x, o1y, p.w[1], p.h[1], o2x, o2y, p.w[2], p.h[2],
                                               p.img[2], p.stride[2], 2 * o3x, 2 * o3y, 2 * kFt3x,
                                               2 * kFt3y);
  __syncthreads();
  fused_down<kFt2x, kFs2, kFt3x, kFt3y, kFt3x>(l2, nullptr, o2x, o2y, p.w[2], p.h[2], o3x, o3y, p.w[3],
                                                p.h[3], p.img[3], p.stride[3], o3x, o3y, kFt3x, kFt3y);
}

__global__ __launch_bounds__(kFtThreads) void k_norm_pyr(EqSrc eq, PyrPack pk) {
  pyr3_body<1>(eq, pk);
}
// the three pyrDown levels of images whose level 0 is in place (the split render: k_time_surface4, then this —
// every S2 word is read once and every level-0 pixel rendered once; the tiles' halos re-read bytes, not planes)
__global__ __launch_bounds__(kFtThreads) void k_pyr3(PyrPack pk) {
  pyr3_body<2>(EqSrc{}, pk);
}

void launch_pyr3(hipStream_t s, const PyrDesc* p, int nimg) {
  const int w3 = p[0].w[3], h3 = p[0].h[3];
  launch_k(k_pyr3, dim3((w3 + kFt3x - 1) / kFt3x, (h3 + kFt3y - 1) / kFt3y, nimg), dim3(kFtThreads), 0, s,
           make_pack(p, nimg));
}

// the `equalize: 1` branch: normalize + the three pyrDown levels of both cameras (see EqSrc)
void launch_norm_pyr(hipStream_t s, const uint8_t* src0, const uint8_t* src1, int src_stride, const int* minmax,
                     const PyrDesc* p) {
  const int w3 = p[0].w[3], h3 = p[0].h[3];
  EqSrc eq;
  eq.img[0] = src0;
  eq.img[1] = src1;
  eq.stride = src_stride;
  eq.minmax = minmax;
  launch_k(k_norm_pyr, dim3((w3 + kFt3x - 1) / kFt3x, (h3 + kFt3y - 1) / kFt3y, 2), dim3(kFtThreads), 0, s, eq,
           make_pack(p, 2));
}

// copyMakeBorder(level, BORDER_REFLECT_101) for every level [OpenCV buildOpticalFlowPyramid]
__global__ __launch_bounds__(256) void k_pyr_pad(PyrPack pk) {
  const PyrDesc& p = pk.p[blockIdx.z];
  const int level = blockIdx.y;
  if (level > p.levels) return;
  const int w = p.w[level], h = p.h[level];
  const int pw = w + 2 * kPad, ph = h + 2 * kPad, stride = p.stride[level];
  // enumerate only the border ring: top+bottom bands (pw*kPad each), then left+right bands
  const int band = pw * kPad;
  const int side = kPad * h;
  const int total = 2 * band + 2 * side;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int x, y;
    if (i < band) {
      y = i / pw;
      x = i - y * pw;
    } else if (i < 2 * band) {
      const int j = i - band;
      y = j / pw;
      x = j - y * pw;
      y += kPad + h;
    } else if (i < 2 * band + side) {
      const int j = i - 2 * band;
      y = j / kPad;
      x = j - y * kPad;
      y += kPad;
    } else {
      const int j = i - 2 * band - side;
      y = j / kPad;
      x = j - y * kPad;
      y += kPad;
      x += kPad + w;
    }
    const int sx = reflect101(x - kPad, w), sy = reflect101(y - kPad, h);
    uint8_t* img = p.img[level];
    img[(size_t)y * stride + x] = img[(size_t)(sy + kPad) * stride + sx + kPad];
  }
  (void)ph;
}

// calcSharrDeriv [OpenCV video/lkpyramid.cpp]: Ix = [3 10 3]^T (x) [-1 0 1],
// Iy = [-1 0 1]^T (x) [3 10 3], REFLECT_101 (read from the padded image), int16 interleaved.
__global__ __launch_bounds__(256) void k_scharr(PyrPack pk) {
  const PyrDesc& p = pk.p[blockIdx.z];
  const int level = blockIdx.y;
  if (level > p.levels) return;
  const int w = p.w[level], h = p.h[level];
  const int stride = p.stride[level];
  const uint8_t* img = p.img[level] + (size_t)kPad * stride + kPad;
  int* deriv = (int*)(p.deriv[level]) + (size_t)kPad * stride + kPad;
  const int total = w * h;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int y = i / w, x = i - y * w;
    const uint8_t* r0 = img + (ptrdiff_t)(y - 1) * stride + x;
    const uint8_t* r1 = r0 + stride;
    const uint8_t* r2 = r1 + stride;
    const int a0 = r0[-1], a1 = r0[0], a2 = r0[1];
    const int b0 = r1[-1], b2 = r1[1];
    const int c0 = r2[-1], c1 = r2[0], c2 = r2[1];
    const int ix = ((a2 + c2) * 3 + b2 * 10) - ((a0 + c0) * 3 + b0 * 10);
    const int iy = ((c2 - a2) + (c0 - a0)) * 3 + (c1 - a1) * 10;
    deriv[(size_t)y * stride + x] = (int)(((unsigned)(uint16_t)(int16_t)iy << 16) | (uint16_t)(int16_t)ix);
  }
}

// k_pyr_pad and k_scharr in one launch: the derivative of an edge pixel takes its neighbours by
// reflect-101 from the interior (= the values the border ring is about to receive), so the two
// parts do not depend on each other; blocks [0, n_scharr) differentiate, the rest fill the ring.
__global__ __launch_bounds__(256) void k_pad_scharr(PyrPack pk, int n_scharr) {
  const PyrDesc& p = pk.p[blockIdx.z];
  const int level = blockIdx.y;
  if (level > p.levels) return;
  const int w = p.w[level], h = p.h[level], stride = p.stride[level];
  if ((int)blockIdx.x < n_scharr) {
    const uint8_t* img = p.img[level] + (size_t)kPad * stride + kPad;
    int* deriv = (int*)(p.deriv[level]) + (size_t)kPad * stride + kPad;
    const int total = w * h;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += n_scharr * 256) {
      const int y = i / w, x = i - y * w;
      const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * stride;
      const uint8_t* r1 = img + (size_t)y * stride;
      const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * stride;
      const int a0 = r0[xm], a1 = r0[x], a2 = r0[xp];
      const int b0 = r1[xm], b2 = r1[xp];
      const int c0 = r2[xm], c1 = r2[x], c2 = r2[xp];
      const int ix = ((a2 + c2) * 3 + b2 * 10) - ((a0 + c0) * 3 + b0 * 10);
      const int iy = ((c2 - a2) + (c0 - a0)) * 3 + (c1 - a1) * 10;
      deriv[(size_t)y * stride + x] = (int)(((unsigned)(uint16_t)(int16_t)iy << 16) | (uint16_t)(int16_t)ix);
    }
    return;
  }
  const int n_pad = gridDim.x - n_scharr;
  const int pw = w + 2 * kPad;
  const int band = pw * kPad, side = kPad * h, total = 2 * band + 2 * side;
  for (int i = ((int)blockIdx.x - n_scharr) * 256 + threadIdx.x; i < total; i += n_pad * 256) {
    int x, y;
    if (i < band) {
      y = i / pw;
      x = i - y * pw;
    } else if (i < 2 * band) {
      const int j = i - band;
      y = j / pw;
      x = j - y * pw;
      y += kPad + h;
    } else if (i < 2 * band + side) {
      const int j = i - 2 * band;
      y = j / kPad;
      x = j - y * kPad;
      y += kPad;
    } else {
      const int j = i - 2 * band - side;
      y = j / kPad;
      x = j - y * kPad;
      y += kPad;
      x += kPad + w;
    }
    const int sx = reflect101(x - kPad, w), sy = reflect101(y - kPad, h);
    uint8_t* img = p.img[level];
    img[(size_t)y * stride + x] = img[(size_t)(sy + kPad) * stride + sx + kPad];
  }
}

static PyrPack make_pack(const PyrDesc* p, int nimg) {
  PyrPack pk;
  for (int i = 0; i < nimg; i++) pk.p[i] = p[i];
  for (int i = nimg; i < 3; i++) pk.p[i] = p[0];
  return pk;
}

void launch_pyr_down(hipStream_t s, const PyrDesc* p, int nimg, int src_level) {
  const int dw = p[0].w[src_level + 1], dh = p[0].h[src_level + 1];
  launch_k(k_pyr_down, dim3((dw + 31) / 32, (dh + 7) / 8, nimg), dim3(256), 0, s,
                     make_pack(p, nimg), src_level);
}
void launch_pyr_pad(hipStream_t s, const PyrDesc* p, int nimg) {
  const int pw = p[0].w[0] + 2 * kPad;
  const int total = 2 * pw * kPad + 2 * kPad * p[0].h[0];
  launch_k(k_pyr_pad, dim3((total + 255) / 256, p[0].levels + 1, nimg), dim3(256), 0, s,
                     make_pack(p, nimg));
}
void launch_pad_scharr(hipStream_t s, const PyrDesc* p, int nimg) {
  const int n_scharr = (p[0].w[0] * p[0].h[0] + 255) / 256;
  const int pw = p[0].w[0] + 2 * kPad;
  const int n_pad = (2 * pw * kPad + 2 * kPad * p[0].h[0] + 255) / 256;
  launch_k(k_pad_scharr, dim3(n_scharr + n_pad, p[0].levels + 1, nimg), dim3(256), 0, s,
           make_pack(p, nimg), n_scharr);
}
void launch_scharr(hipStream_t s, const PyrDesc* p, int nimg) {
  const int total = p[0].w[0] * p[0].h[0];
  int gx = (total + 255) / 256;
  launch_k(k_scharr, dim3(gx, p[0].levels + 1, nimg), dim3(256), 0, s,
                     make_pack(p, nimg));
}

// ============================================================================ pyramidal LK
// points (= waves) per block.  (Four: 122 KB of LDS per block in the float-order mode, one block per CU.  One —
// 40 KB, a straggler holds a SIMD, not a CU — measured slower on average in rounds 5 and 6 (+4 %), two +1 %:
// tools/build_variant.sh lk2 -DESVIO_LK_WAVES=2, KERNELS.md.)
constexpr int kLkWaves = kLkPointsPerBlock;  // (ESVIO_LK_WAVES, fe_kernels.h)
// cv::calcOpticalFlowPyrLK's LKTrackerInvoker [OpenCV video/lkpyramid.cpp], one wave64 per point,
// all levels — and optionally the forward AND the backward call of a forward/backward check — in
// one launch.  The loop is instruction-latency bound (one wave per SIMD, <=30 dependent
// iterations per level), so everything here is about a short dependent chain:
//  * lane = (window row, 7-px run): 63 lanes cover the 21x21 window, each lane owns 7 horizontally
//    adjacent pixels, so the 4-tap bilinear needs 8 bytes from each of two rows (not 28 taps);
//  * the patch (I, Ix, Iy) lives in registers across the <=30 iterations of a level;
//  * the search region of the next image is staged once per level into a 34x40 B LDS tile
//    (window + 6 px margin, dword-aligned origin), read back as aligned dwords + v_alignbyte, and
//    re-staged only if the window drifts out;
//  * 24-bit integer multiplies (v_mul_i32_i24 / v_mad_i32_i24) everywhere: all operands fit;
//  * A11/A12/A22 and b1/b2 are exact integer sums (OpenCV's int64-accumulator build of the same
//    loop): 3 DPP butterflies in int32, 8 v_readlane + scalar adds, one exact i64->f64->f32
//    conversion, so the result does not depend on reduction order and equals the oracle's bits.
#define CV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

constexpr int kLkMargin = 6;
constexpr int kLkRegW = 40;                              // staged bytes per row (10 dwords)
constexpr int kLkRegH = kLkWin + 1 + 2 * kLkMargin;      // 34 rows
constexpr int kLkRegDw = kLkRegW / 4 * kLkRegH;          // 340 dwords per wave
constexpr int kLkRegStore = (kLkRegDw + 63) / 64 * 64;   // (every lane stores its six dwords; an 8-column lane of the last row reads up to 4 B past the region: unused bytes)

// Exact wave64 sums of two per-lane integers with |v| < 2^28, result as float (one rounding, equal
// to (float)(int64 sum)) in EVERY lane, VALU only (no readlane / scalar hop):
//   3 DPP butterflies in int32 (8-lane sums < 2^31), split into 12 low bits and the rest so the
//   remaining xor-8/16/32 steps stay in int32 (row_mirror DPP, v_permlane16_swap, v_permlane32_swap),
//   then hi*4096+lo exactly in fp64 and one cvt to fp32.
template <int CTRL>
__device__ __forceinline__ int xor_add_dpp_t(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
#define xor_add_dpp(v, ctrl) xor_add_dpp_t<ctrl>(v)
__device__ __forceinline__ int xor16_add(int v) {
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return (int)r[0] + (int)r[1];
}
__device__ __forceinline__ int xor32_add(int v) {
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return (int)r[0] + (int)r[1];
}
// Per-lane constants of the packed reduction below: after the 8-lane butterflies every lane of a
// quad keeps ONE of the four pieces {a.lo12, a.hi20, b.lo12, b.hi20} (piece = lane & 3).
struct RedLane {
  bool take_b;       // piece belongs to b
  uint32_t off, wid; // bit field of the biased 8-lane sum
  float scale, corr; // piece -> its exact contribution to sum * 2^-20
};
__device__ __forceinline__ RedLane red_lane(int lane) {
  RedLane r;
  r.take_b = (lane & 2) != 0;
  const bool hi = (lane & 1) != 0;
  r.off = hi ? 12u : 0u;
  r.wid = hi ? 20u : 12u;
  r.scale = hi ? 0x1p-8f : 0x1p-20f;          // hi piece counts 4096 * 2^-20
  r.corr = hi ? -(float)(1 << 22) * 0x1p-8f : 0.f;  // eight biases of 2^31 = 2^22 * 4096
  return r;
}
template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));  // (every lane has a source)
}
// Exact wave64 sums Sa, Sb of two per-lane integers (|v| < 2^28), each passed in with kRedSeed
// (2^28) added so that the 8-lane sums carry a bias of 2^31; returns RN(Sa * 2^-20) and
// RN(Sb * 2^-20) (= (float)(int64 sum) * 2^-20, one rounding) in EVERY lane, VALU only:
//   3 DPP butterflies in uint32 (biased 8-lane sums in (0, 2^32)); keep one 12/20-bit piece per
//   lane, so the xor-8/16/32 steps (row_ror:8, v_permlane16_swap, v_permlane32_swap) run on a single
//   register; each piece total (< 2^23) converts to fp32 exactly, is scaled exactly, and lo + hi is
//   ONE fp32 addition of two exact terms, i.e. the correctly rounded sum.
constexpr int kRedSeed = 1 << 28;
__device__ __forceinline__ void wave_sum2_exact(int a, int b, const RedLane& rl, float& fa, float& fb) {
  a = xor_add_dpp(a, 0xB1);  // quad_perm [1,0,3,2]
  b = xor_add_dpp(b, 0xB1);
  a = xor_add_dpp(a, 0x4E);  // quad_perm [2,3,0,1]
  b = xor_add_dpp(b, 0x4E);
  a = xor_add_dpp(a, 0x141);  // row_half_mirror -> 8-lane sums
  b = xor_add_dpp(b, 0x141);
  const uint32_t v = (uint32_t)(rl.take_b ? b : a);
  int x = (int)__builtin_amdgcn_ubfe(v, rl.off, rl.wid);
  x = xor_add_dpp(x, 0x128);  // row_ror:8 -> 16-lane sums (lane & 3 preserved)
  x = xor16_add(x);
  x = xor32_add(x);
  const float t = __fmaf_rn((float)x, rl.scale, rl.corr);  // exact
  const float u = t + quad_bcast<0xB1>(t);  // lanes 0, 1 of a quad: a's lo + hi; lanes 2, 3: b's
  fa = quad_bcast<0x00>(u);
  fb = quad_bcast<0xAA>(u);
}

typedef short ss2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack16(int lo, int hi) {  // {(int16)lo, (int16)hi}
  return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c) {  // v_dot2_i32_i16
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b), c, false);
}

// every lane of the wave holds the same point state, so loop exits are uniform: turning the
// condition into a ballot makes the compiler emit scalar branches instead of exec-mask bookkeeping
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
// ... and combining several tests as ballot masks keeps the combination on the scalar unit
typedef unsigned long long lanemask_t;
__device__ __forceinline__ lanemask_t bal(bool c) { return __builtin_amdgcn_ballot_w64(c); }

struct LkCall {
  PyrDesc P;  // prev pyramid (+ derivatives)
  PyrDesc N;  // next pyramid (images only)
  int max_level;
  int max_count;
  double eps2;
  int flags;
};

// dword-aligned 34x40 B region of image J around window origin (wx,wy) -> registers
struct LkRegionLane {  // where this lane's dwords sit inside a region: row and byte column (the same at every level)
  int ry[(kLkRegDw + 63) / 64], rb[(kLkRegDw + 63) / 64];
};
__device__ __forceinline__ LkRegionLane lk_region_lane(int lane) {
  LkRegionLane r;
#pragma unroll
  for (int k = 0; k < (kLkRegDw + 63) / 64; k++) {
    const int d = min(k * 64 + lane, kLkRegDw - 1);  // (the lanes past the region's end repeat its last dword)
    r.ry[k] = d / (kLkRegW / 4);
    r.rb[k] = 4 * (d - r.ry[k] * (kLkRegW / 4));
  }
  return r;
}
__device__ __forceinline__ void lk_region_load(const uint8_t* J, int stride, int rows, int wx, int wy,
                                               int lane, const LkRegionLane& rl, int& rx0, int& ry0,
                                               uint32_t (&reg)[(kLkRegDw + 63) / 64]) {
  rx0 = (wx - kLkMargin) & ~3;
  ry0 = wy - kLkMargin;
  // (wx, wy) is the same in every lane: a region that lies inside the padded buffer — all but the windows at the
  // image's edge — is ONE scalar base address and a per-lane offset, row * stride + column
  const int sx = __builtin_amdgcn_readfirstlane(rx0), sy = __builtin_amdgcn_readfirstlane(ry0);
  if (sx >= -kPad && sx + kLkRegW <= stride - kPad && sy >= -kPad && sy + kLkRegH <= rows + kPad) {
    const uint8_t* base = J + (ptrdiff_t)sy * stride + sx;
#pragma unroll
    for (int k = 0; k < (kLkRegDw + 63) / 64; k++) {
      uint32_t off;
      asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(rl.ry[k]), "s"(stride), "v"(rl.rb[k]));
      reg[k] = *(const uint32_t*)(base + off);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < (kLkRegDw + 63) / 64; k++) {
    int gy = ry0 + rl.ry[k], gx = rx0 + rl.rb[k];
    // clamp into the padded buffer; clamped bytes are never used by a valid window
    gy = min(max(gy, -kPad), rows + kPad - 1);
    gx = min(max(gx, -kPad), stride - kPad - 4);
    reg[k] = *(const uint32_t*)(J + (ptrdiff_t)gy * stride + gx);
  }
}
__device__ __forceinline__ void lk_region_store(uint32_t* regJ, int lane,
                                                const uint32_t (&reg)[(kLkRegDw + 63) / 64]) {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
  for (int k = 0; k < (kLkRegDw + 63) / 64; k++) {
    regJ[k * 64 + lane] = reg[k];  // (past the region's end: spare words, any value)
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// ---- lk_accum 2: the sums of the LK normal equations in float, in the order of OpenCV 4.2's x86
// SIMD128 build (typedef float acctype; oracle/esvio_oracle.cpp calc_lk, accum == 2, restated from
// recall — unpinned): per window row the vector loop takes columns 0..15, lane k of a float32x4
// accumulator getting columns k, k+4, k+8, k+12 (A matrix: one product per add; b vector: the
// products of columns x, x+4 of a step of 8 added in int32 first, converted, then one add), the
// columns 16..20 go to a scalar accumulator; at the end scalar += horizontal sum of the lanes.
// A float sum in a prescribed order is a chain of dependent adds, so it cannot be spread over the
// wave as it is: the pixel lanes compute the per-pixel integer terms, put them into LDS in chain
// order, and walking lanes add them up in float (lk_float_sums_b below).
// In this mode a lane owns EIGHT adjacent columns of its window row (runs 0..7, 8..15, 16..20 + three
// columns that do not exist and contribute nothing), so both columns of every pmaddwd pair (x, x+4) sit
// in one lane and the tail's five columns in another.
//
// A matrix (once per call, every pyramid level at the same time): fifteen chains per level — type t = A11,
// A12, A22: chain 5 t + k for the vector lane k = column mod 4 (slot = column / 4, columns 0..15), chain
// 5 t + 4 for the tail (columns 16..20).  Every chain takes five terms per window row (a vector lane four
// and a zero, the tail its five columns), kept as floats in LDS, T[level][chain][row][5]: slots a chain does
// not use hold +0.f (adding it is exact), so that all chain lanes run one loop in lock step — 105 dependent
// adds, a row fetched with two LDS instructions while the previous one is added.  The pixel lanes write
// their own products (< 2^24: exact in float) straight into T, each column at an address the lane worked
// out once per launch (type and level are the instruction's immediate offset); the three columns a lane of
// run 2 owns beyond the window write their zeros into zero slots.  The walker of level L's chain c is lane
// 16 L + c, so a level is a DPP row and the combination tail + ((q0 + q2) + (q1 + q3)) — v_reduce_sum's
// order —, the eigenvalue test and 1 / D are done once for all levels, a level per row.
// (These sums grow monotonically past 2^24 on any textured patch, so there is nothing to guess here.)
// one chain: 21 rows of 5 terms, T -> its first term
__device__ __forceinline__ float lk_chain_walk(const float* T) {
  float acc = 0.f;
  float c[5], nx[5];
#pragma unroll
  for (int q = 0; q < 5; q++) c[q] = T[q];
#pragma unroll
  for (int y = 0; y < kLkWin; y++) {
    const int yn = y + 1 < kLkWin ? y + 1 : y;
#pragma unroll
    for (int q = 0; q < 5; q++) nx[q] = T[yn * 5 + q];
#pragma unroll
    for (int q = 0; q < 5; q++) acc = __fadd_rn(acc, c[q]);
#pragma unroll
    for (int q = 0; q < 5; q++) c[q] = nx[q];
  }
  return acc;
}

constexpr int kLkChainsA = 15;
constexpr int kLkChainWordsA = kLkWin * 5;                 // one chain: [row][5]
constexpr int kLkTypeWordsA = 5 * kLkChainWordsA;          // from a type's chains to the next type's
constexpr int kLkTermLevelA = kLkChainsA * kLkChainWordsA;  // T[chain][row][5] floats of one pyramid level
constexpr int kLkTermWordsA = kMaxLevels * kLkTermLevelA;  // all levels side by side: ONE walk sums them all
static_assert(kMaxLevels * 16 <= 64, "one DPP row of walkers per level");

typedef __attribute__((address_space(3))) float lds_f32;
// word offset (inside a level's table) of column x's A11 term in window row `row`; x >= kLkWin: a zero slot
__device__ __forceinline__ int lk_term_word_A(int row, int x) {
  const int chain = x < 16 ? (x & 3) : x < kLkWin ? 4 : (x & 3);
  const int slot = x < 16 ? (x >> 2) : x < kLkWin ? x - 16 : 4;
  return (chain * kLkWin + row) * 5 + slot;
}
// the pixel lanes' products of one level into that level's table: ta[k] = LDS byte address of column k's A11 term
// in level 0's table
template <int NP>
__device__ __forceinline__ void lk_float_terms_A(const uint32_t (&ta)[NP], int L, const int* pIx, const int* pIy) {
#pragma unroll
  for (int k = 0; k < NP; k++) {
    lds_f32* d = (lds_f32*)(uintptr_t)ta[k] + L * kLkTermLevelA;
    const float fx = (float)pIx[k], fy = (float)pIy[k];  // (|.| <= 4080: the products are exact)
    d[0] = __fmul_rn(fx, fx);
    d[kLkTypeWordsA] = __fmul_rn(fx, fy);
    d[2 * kLkTypeWordsA] = __fmul_rn(fy, fy);
  }
}
// every level's fifteen chains at once: lane 16 L + c walks the 105 terms of level L's chain c (round 4 walked level
// after level, fifteen lanes at a time: 1.1 us per level and call, 8.6 us of a 91 us launch)
__device__ __forceinline__ float lk_float_walk_A(const float* TA, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  const int c = (lane & 15) < kLkChainsA ? (lane & 15) : 0;
  return lk_chain_walk(TA + ((lane >> 4) * kLkChainsA + c) * kLkChainWordsA);
}
template <int CTRL>
__device__ __forceinline__ float dpp_row_f32(float v) {  // the value CTRL brings in (0 where it has no source lane)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// acc of lane 16 L + c -> level L's sums in lanes 16 L (A11), 16 L + 5 (A12), 16 L + 10 (A22), already scaled by
// FLT_SCALE; a11 / a12 / a22: all three brought to lane 16 L
__device__ __forceinline__ float lk_float_sums_A(float acc, float& a11, float& a12, float& a22) {
  const float kScale = 1.f / (float)(1 << 20);
  const float s1 = __fadd_rn(acc, dpp_row_f32<0x102>(acc));  // row_shl:2: chain 5 t: q0 + q2, chain 5 t + 1: q1 + q3
  const float s2 = __fadd_rn(s1, dpp_row_f32<0x101>(s1));    // chain 5 t: (q0 + q2) + (q1 + q3)
  const float s3 = __fadd_rn(dpp_row_f32<0x104>(acc), s2);   // tail + ...
  const float out = __fmul_rn(s3, kScale);
  a11 = out;
  a12 = dpp_row_f32<0x105>(out);
  a22 = dpp_row_f32<0x10A>(out);
  return out;
}

// The b vector is summed once per LK iteration.  Ten chains: vector chains c = 4 comp + k (comp 0: b1,
// 1: b2; k = column mod 4) take per window row the pair sums of columns (k, k + 4) and (8 + k, 12 + k) —
// added exactly in int32 (pmaddwd), converted, then ONE float add each, 42 terms —; the tails (one per
// comp) take columns 16..20 one by one, 105 terms.  Round 3 had ten lanes walk their chains row by row out
// of LDS (0.73 us of a 1.23 us iteration), round 4 38 lanes walking 16-word segments (0.70 us per
// iteration).  Now:
//  * the pixel lanes write the INTEGER terms into a table in CHAIN order, cut into segments: a vector
//    chain = 4 segments of <= 11 terms, a tail = 12 segments of <= 9 terms, each stored in 12 words (the
//    words a segment does not use stay 0).  Every pixel lane writes five (b1, b2) word pairs with five
//    ds_write2_b32 at per-lane addresses — lanes of runs 0 / 1 their four pair sums (the fifth pair goes
//    to a dump word of the lane), lanes of run 2 their five tail products — so one instruction stream
//    serves both kinds of lane: the (comp 0, k) -> (comp 1, k) distance of the vector chains equals the
//    tail's comp 0 -> comp 1 distance (kF32CompStride);
//  * one lane per segment — ALL 64 lanes minus 8: row 0 of the wave = tail b1 (segments 0..11), rows 1 / 2
//    = vector chains of b1 / b2 (lane = 4 g + k inside its row: segment g of chain k), row 3 = tail b2 —
//    reads its 12 words (3 ds_read_b128) and they all walk at the same time, each from a GUESS of what
//    the chain has summed to before its segment: the exact integer sum of the earlier segments (a DPP
//    scan inside the row: row_shr 1/2/4/8 for the tails, row_shr 4/8 under bank masks for the vector
//    rows) — which IS the float chain's value as long as no add before it rounded, i.e. while the
//    running sum stays below 2^24: 87 % of the iterations on the bench stream for ALL ten chains
//    (counted with the oracle); then every segment compares its guess with its left neighbour's end
//    value and, if one differs, all walk again from the corrected starts.  Segment 0 always starts right,
//    so after r rounds the first r segments are final: at most 12 rounds = the sequential walk, one round
//    of 11 adds in the usual case.  The result is the sequential chain's for any input — the guess only
//    decides how many rounds it takes;
//  * the four vector chains of a comp end in one quad ((k0 + k2) + (k1 + k3) = two quad_perm adds).
constexpr int kF32SegStore = 12;
constexpr int kF32VecSegs = 4, kF32VecTerms = 11;    // 4 x 11 >= 42
constexpr int kF32TailSegs = 12, kF32TailTerms = 9;  // 12 x 9 >= 105
constexpr int kF32VecWords = kF32VecSegs * kF32SegStore;                // 48 words per vector chain
constexpr int kF32CompStride = 4 * kF32VecWords;                        // 192
constexpr int kF32TailBase = 8 * kF32VecWords;                          // 384: tail b1; tail b2 at + kF32CompStride
constexpr int kF32ZeroSeg = kF32TailBase + kF32TailSegs * kF32SegStore;  // 528: twelve words nobody writes
constexpr int kF32Dump = kF32TailBase + kF32CompStride + kF32TailSegs * kF32SegStore;  // 720: one word per lane (+ kF32CompStride)
constexpr int kF32TermWords = kF32Dump + kF32CompStride + 64;           // 976
static_assert(kF32ZeroSeg + kF32SegStore <= kF32TailBase + kF32CompStride, "zero segment inside tail b1's padding");
constexpr unsigned long long kF32Walkers = 0x0FFFFFFFFFFF0FFFull;       // lanes 0..11, 16..47, 48..59

struct LkLaneF32 {  // per-lane constants of the float-order sums
  uint32_t wb[5];   // byte offsets of the five (b1, b2) word pairs this pixel lane writes
  uint32_t rb;      // byte offset of the segment this lane walks
  uint32_t ta[8];   // LDS byte addresses of the lane's eight columns' A11 terms (level 0; lk_float_terms_A)
  bool run2;        // pixel lane of columns 16..20
};
__device__ __forceinline__ LkLaneF32 lk_lane_f32(int lane_in, uint32_t ta_base /* LDS byte address of the wave's A table */) {
  // (lane 63 has no pixels of its own: it is a second lane 62 — same values to the same addresses)
  const int lane = lane_in < 63 ? lane_in : 62;
  LkLaneF32 r;
  const int row = lane / 3;
  const int run = lane - row * 3;
  r.run2 = run == 2;
#pragma unroll
  for (int k = 0; k < 8; k++) r.ta[k] = ta_base + 4u * (uint32_t)lk_term_word_A(row, run * 8 + k);
#pragma unroll
  for (int j = 0; j < 5; j++) {
    int w = kF32Dump + lane;
    if (run < 2 && j < 4) {
      const int idx = 2 * row + run;
      w = j * kF32VecWords + (idx / kF32VecTerms) * kF32SegStore + idx % kF32VecTerms;
    } else if (run == 2) {
      const int idx = 5 * row + j;
      w = kF32TailBase + (idx / kF32TailTerms) * kF32SegStore + idx % kF32TailTerms;
    }
    r.wb[j] = (uint32_t)w * 4u;
  }
  const int wrow = lane_in >> 4, l16 = lane_in & 15;
  int rw = kF32ZeroSeg;
  if (wrow == 0 || wrow == 3) {
    if (l16 < kF32TailSegs) rw = kF32TailBase + (wrow == 3 ? kF32CompStride : 0) + l16 * kF32SegStore;
  } else {
    rw = ((wrow - 1) * 4 + (l16 & 3)) * kF32VecWords + (l16 >> 2) * kF32SegStore;
  }
  r.rb = (uint32_t)rw * 4u;
  return r;
}
// v (+)= the value CTRL brings in, in the rows / banks the masks enable; everywhere else (and in lanes
// CTRL has no source lane for) v stays
template <int CTRL, int ROWS, int BANKS>
__device__ __forceinline__ int dpp_add_masked(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, BANKS, false);
}

typedef __attribute__((address_space(3))) int lds_int;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i32x4 lds_int4;

// vx / vy: the five (b1, b2) words this pixel lane contributes — runs 0 / 1: the four pmaddwd pair sums
// diff_j Ix_j + diff_(j+4) Ix_(j+4) (word 4 goes to the lane's dump word); run 2: the products of columns
// 16..20 — all exact in int32; tb: LDS byte address of the wave's term table
__device__ __forceinline__ void lk_float_sums_b(const int (&vx)[5], const int (&vy)[5], uint32_t tb, const LkLaneF32& ln,
                                                float& b1, float& b2) {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");  // (the previous iteration's walkers have read)
#pragma unroll
  for (int j = 0; j < 5; j++) {
    lds_int* d = (lds_int*)(uintptr_t)(tb + ln.wb[j]);
    d[0] = vx[j];
    d[kF32CompStride] = vy[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  // ---- the walking lanes
  int t[kF32SegStore];
  {
    const lds_int4* src = (const lds_int4*)(uintptr_t)(tb + ln.rb);
#pragma unroll
    for (int q = 0; q < kF32SegStore / 4; q++) {
      const i32x4 v = src[q];
      t[4 * q] = v.x;
      t[4 * q + 1] = v.y;
      t[4 * q + 2] = v.z;
      t[4 * q + 3] = v.w;
    }
  }
  int isum = 0;
  float f[kF32VecTerms];
#pragma unroll
  for (int i = 0; i < kF32VecTerms; i++) {  // (word 11 of a segment is never used)
    isum += t[i];
    f[i] = (float)t[i];  // v_cvt_f32_i32 of the int32 term
  }
  // what the chain's earlier segments add up to, exactly: inclusive scans inside the rows, minus the own sum
  int xa = isum, xb = isum;
  xa = dpp_add_masked<0x111, 0x9, 0xf>(xa);  // row_shr:1, rows 0 and 3 (tails)
  xb = dpp_add_masked<0x114, 0x6, 0xe>(xb);  // row_shr:4, rows 1 and 2, segments >= 1 (vector chains)
  xa = dpp_add_masked<0x112, 0x9, 0xf>(xa);
  xb = dpp_add_masked<0x118, 0x6, 0xc>(xb);  // row_shr:8, segments >= 2
  xa = dpp_add_masked<0x114, 0x9, 0xf>(xa);
  xa = dpp_add_masked<0x118, 0x9, 0xf>(xa);
  // rows 1 / 2 never changed xa, rows 0 / 3 never changed xb: (xa - isum) + (xb - isum) is the lane's own row's
  float guess = (float)(xa + xb - 2 * isum), end;
  for (;;) {
    float acc = guess;
#pragma unroll
    for (int i = 0; i < kF32VecTerms; i++) acc = __fadd_rn(acc, f[i]);
    end = acc;
    // the left neighbour's end; +0.f for segment 0
    int s = __builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x111, 0x9, 0xf, false);
    s = __builtin_amdgcn_update_dpp(s, __float_as_int(acc), 0x114, 0x6, 0xe, false);
    const lanemask_t changed = bal((uint32_t)s != __float_as_uint(guess)) & kF32Walkers;
    guess = __int_as_float(s);
    if (!changed) break;
  }
  // chain ends: tails in lanes 11 / 59, the four vector chains of a comp in lanes 28..31 / 44..47
  int e = __float_as_int(end);
  float q = __fadd_rn(end, __int_as_float(__builtin_amdgcn_update_dpp(0, e, 0x4E, 0xf, 0xf, false)));  // k0 + k2 | k1 + k3
  q = __fadd_rn(q, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0xB1, 0xf, 0xf, false)));
  const int qi = __float_as_int(q);
  const float kScale = 1.f / (float)(1 << 20);
  const float t1 = __int_as_float(__builtin_amdgcn_readlane(e, 11)), t2 = __int_as_float(__builtin_amdgcn_readlane(e, 59));
  const float v1 = __int_as_float(__builtin_amdgcn_readlane(qi, 28)), v2 = __int_as_float(__builtin_amdgcn_readlane(qi, 44));
  // fb += (s0 + 0.f) + (s2 + 0.f): the two + 0.f only turn a -0.f into +0.f, and no sum here is -0.f (every
  // chain starts from +0.f and adds converted integers, none of which is -0.f)
  b1 = __fmul_rn(__fadd_rn(t1, v1), kScale);
  b2 = __fmul_rn(__fadd_rn(t2, v2), kScale);
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;

// v_dot2_i32_i16 with the accumulator seed in an SGPR (VOP3P form): the compiler's own choice for a constant
// seed is v_mov + v_dot2c (the VOP2 form accumulates in place), one more instruction per pixel on a wave
// that issues one instruction every four cycles whatever it is
// oy * kLkRegW + ox as ONE v_mad_u32_u24 (left to itself the compiler re-associates the window offsets and
// ends up with a quarter-rate v_mul_lo_u32)
__device__ __forceinline__ uint32_t mad_u24_regw(uint32_t oy, uint32_t ox) {
  uint32_t r;
  static_assert(kLkRegW <= 64, "inline constant");
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(oy), "n"(kLkRegW), "v"(ox));
  return r;
}
__device__ __forceinline__ int sdot2_zero(uint32_t a, uint32_t b) {  // a.lo * b.lo + a.hi * b.hi, no accumulator
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int sdot2_seed(uint32_t a, uint32_t b, int seed) {
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(seed));
  return r;
}

// one calcOpticalFlowPyrLK call for one point; returns nextPts[pt] and status
template <int ACCUM>
__device__ __forceinline__ void lk_point(const LkCall& c, const float2 prev0, const float2 init,
                                         uint32_t* regJ, uint32_t term_base, float* acc_ta, const LkLaneF32* lnf, int lane,
                                         float2& np_out, int& st_out) {
  // pixels per lane: the exact mode's lanes own 7 adjacent columns (3 x 7 = 21); the float-order mode's 8
  // (0..7, 8..15, 16..23), see lk_float_sums_b
  constexpr int WIN = kLkWin, NP = ACCUM == 2 ? 8 : 7, NPP = 4, NL = kMaxLevels;
  const float halfWin = (WIN - 1) * 0.5f;
  const int W_BITS = 14;
  const RedLane rl = red_lane(lane);
  // exact mode: lane 63 is off (its terms are zeros in the wave's sums); float-order mode: lane 63 is a second lane
  // 62 (every sum goes through tables, where writing the same value to the same address twice is harmless)
  const bool on = ACCUM == 2 || lane < 63;
  const int plane = ACCUM == 2 ? min(lane, 62) : lane;
  const int row = on ? plane / 3 : 0;
  const int x0 = on ? (plane - row * 3) * NP : 0;
  const int lane_bo = row * kLkRegW + x0;  // this lane's byte offset inside a staged window
  const LkRegionLane rgl = lk_region_lane(lane);

  // ---- phase A: the previous-image side of EVERY level depends only on prevPts, so all
  // levels' patches (I, Ix, Iy in registers) and 2x2 matrices are built up front with the
  // global loads of all levels in flight together (one memory round trip instead of one per level)
  // the patch is kept as packed int16 pairs {px 2m, px 2m+1} (halves of columns that do not exist are
  // zero), the form the v_dot2 / v_pk iteration below consumes
  uint32_t pIp[NL][NPP], pIxp[NL][NPP], pIyp[NL][NPP];
  uint32_t pIxB[NL], pIyB[NL];  // (float-order mode: the fifth word's derivatives, see lk_float_sums_b)
  bool win_ok[NL], eig_ok[NL];
  float A11[NL], A12[NL], A22[NL], Dinv[NL];
  {
    // the window's image bytes as (unaligned) dwords — NP + 1 bytes of each of the lane's two rows — and its
    // derivative pairs {Ix, Iy}; the bilinear taps are the iteration's: a column's values of the two rows as an int16
    // pair (v_perm_b32) against the packed weight pairs (v_dot2_i32_i16), rounding constant as the accumulator seed
    constexpr int NW = (NP + 1 + 3) / 4;
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    uint32_t t0[NL][NW], t1[NL][NW];
    int g0[NL][NP + 1], g1[NL][NP + 1];
    uint32_t W0p[NL], W1p[NL];
    if (ACCUM == 2) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");  // (the previous call's walk has read its table)
#pragma unroll
    for (int L = 0; L < NL; L++) {
      win_ok[L] = false;
      eig_ok[L] = false;
      if (L > c.max_level) continue;
      const int cols = c.P.w[L], rows = c.P.h[L], stride = c.P.stride[L];
      const float sc = 1.f / (float)(1 << L);
      const float prevX = prev0.x * sc - halfWin, prevY = prev0.y * sc - halfWin;
      const float flx = floorf(prevX), fly = floorf(prevY);
      const int iprevX = (int)flx, iprevY = (int)fly;
      win_ok[L] = !(iprevX < -WIN || iprevX >= cols || iprevY < -WIN || iprevY >= rows);
      const float fa = prevX - flx, fb = prevY - fly;
      const int w00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
      const int w01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
      const int w10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
      const int w11 = (1 << W_BITS) - w00 - w01 - w10;
      W0p[L] = pack16(w00, w10);  // the LEFT column's weights for the rows (y, y + 1), (<= 2^14 each)
      W1p[L] = pack16(w01, w11);  // the right column's
      // clamp the window origin for the loads of an out-of-range window (values unused)
      const int lx = min(max(iprevX, -WIN), cols - 1), ly = min(max(iprevY, -WIN), rows - 1);
      const ptrdiff_t o = (ptrdiff_t)(kPad + row + ly) * stride + kPad + lx + x0;
      const uint8_t* s0 = c.P.img[L] + o;
      const uint8_t* s1 = s0 + stride;
      const int* d0 = (const int*)c.P.deriv[L] + o;
      const int* d1 = d0 + stride;
#pragma unroll
      for (int q = 0; q < NW; q++) {
        t0[L][q] = ((const u32_unaligned*)s0)[q];
        t1[L][q] = ((const u32_unaligned*)s1)[q];
      }
#pragma unroll
      for (int k = 0; k <= NP; k++) {
        g0[L][k] = d0[k];
        g1[L][k] = d1[k];
      }
    }
#pragma unroll
    for (int L = 0; L < NL; L++) {
      if (L > c.max_level) continue;
      int sA11 = kRedSeed, sA12 = kRedSeed, sA22 = kRedSeed;  // + 7 terms <= 2^24 each
      int pI[2 * NPP], pIx[2 * NPP], pIy[2 * NPP];
#pragma unroll
      for (int k = NP; k < 2 * NPP; k++) pI[k] = pIx[k] = pIy[k] = 0;
      // column k's two rows as an int16 pair — {row y, row y + 1} —: a pixel is its own column against the left
      // weights plus the next column against the right ones, so NP + 1 v_perm serve NP pixels
      uint32_t vI[NP + 1], vX[NP + 1], vY[NP + 1];
#pragma unroll
      for (int k = 0; k <= NP; k++) {
        const uint32_t sel = 0x0c000c00u | ((uint32_t)(4 + (k & 3)) << 16) | (uint32_t)(k & 3);
        vI[k] = __builtin_amdgcn_perm(t1[L][k >> 2], t0[L][k >> 2], sel);
        vX[k] = __builtin_amdgcn_perm((uint32_t)g1[L][k], (uint32_t)g0[L][k], 0x05040100u);  // {Ix y, Ix y+1}
        vY[k] = __builtin_amdgcn_perm((uint32_t)g1[L][k], (uint32_t)g0[L][k], 0x07060302u);  // {Iy y, Iy y+1}
      }
#pragma unroll
      for (int k = 0; k < NP; k++) {
        // CV_DESCALE(x, n) = (x + (1 << (n - 1))) >> n: the rounding constant is the first dot's accumulator
        const int ival = sdot2(vI[k + 1], W1p[L], sdot2_seed(vI[k], W0p[L], 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
        const int ixval = sdot2(vX[k + 1], W1p[L], sdot2_seed(vX[k], W0p[L], 1 << (W_BITS - 1))) >> W_BITS;
        const int iyval = sdot2(vY[k + 1], W1p[L], sdot2_seed(vY[k], W0p[L], 1 << (W_BITS - 1))) >> W_BITS;
        // (exact mode: lane 63; float-order mode: columns 21..23 of run 2)
        const bool col_ok = ACCUM == 2 ? (k < WIN - 16 || !lnf->run2) : on;
        pI[k] = ival;  // (0 .. 255 * 32; the derivatives: |.| <= 16 * 255 — all inside int16 as they are)
        pIx[k] = col_ok ? ixval : 0;
        pIy[k] = col_ok ? iyval : 0;
        if (ACCUM != 2) {
          sA11 += __mul24(pIx[k], pIx[k]);
          sA12 += __mul24(pIx[k], pIy[k]);
          sA22 += __mul24(pIy[k], pIy[k]);
        }
      }
#pragma unroll
      for (int m = 0; m < NPP; m++) {
        // exact mode: horizontally adjacent pixels share a register; float-order mode: the two columns of a
        // pmaddwd pair (j, j + 4), so that ONE v_dot2 is the pair sum the reference adds in int32
        const int a = ACCUM == 2 ? m : 2 * m, b = ACCUM == 2 ? m + 4 : 2 * m + 1;
        pIp[L][m] = pack16(pI[a], pI[b]);
        pIxp[L][m] = pack16(pIx[a], pIx[b]);
        pIyp[L][m] = pack16(pIy[a], pIy[b]);
      }
      if (ACCUM == 2) {
        // run 2 (columns 16..20): its pair 0 = columns 16 and 20, both real and both wanted on their own
        if (lnf->run2) {
          pIxp[L][0] = pack16(pIx[0], 0);
          pIyp[L][0] = pack16(pIy[0], 0);
        }
        pIxB[L] = pack16(0, pIx[4]);
        pIyB[L] = pack16(0, pIy[4]);
      }
      if (ACCUM == 2) {
        lk_float_terms_A<8>(lnf->ta, L, pIx, pIy);
      } else {
        float fdummy;
        wave_sum2_exact(sA11, sA12, rl, A11[L], A12[L]);  // already scaled by FLT_SCALE = 2^-20
        wave_sum2_exact(sA22, kRedSeed, rl, A22[L], fdummy);
      }
    }
    if (ACCUM == 2) {
      // the walk and what follows it once for all levels: level L's sums end up in the lanes of DPP row L
      float a11, a12, a22;
      const float out = lk_float_sums_A(lk_float_walk_A(acc_ta, lane), a11, a12, a22);
      const float D = a11 * a22 - a12 * a12;
      const float minEig = (a22 + a11 - sqrtf((a11 - a22) * (a11 - a22) + 4.f * a12 * a12)) / (float)(2 * WIN * WIN);
      const lanemask_t ok = bal(!(minEig < 1e-4f || D < 1.1920929e-07f /*FLT_EPSILON*/));
      const float dinv = 1.f / D;
#pragma unroll
      for (int L = 0; L < NL; L++) {
        if (L > c.max_level) continue;
        A11[L] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(out), 16 * L));
        A12[L] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(out), 16 * L + 5));
        A22[L] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(out), 16 * L + 10));
        Dinv[L] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dinv), 16 * L));
        eig_ok[L] = (ok >> (16 * L)) & 1;
      }
    } else {
#pragma unroll
      for (int L = 0; L < NL; L++) {
        if (L > c.max_level) continue;
        const float D = A11[L] * A22[L] - A12[L] * A12[L];
        const float minEig =
            (A22[L] + A11[L] - sqrtf((A11[L] - A22[L]) * (A11[L] - A22[L]) + 4.f * A12[L] * A12[L])) /
            (float)(2 * WIN * WIN);
        eig_ok[L] = !(minEig < 1e-4f || D < 1.1920929e-07f /*FLT_EPSILON*/);
        Dinv[L] = 1.f / D;
      }
    }
  }

  // ---- phase B: coarse-to-fine iterations
  const float eps_lo = (float)(c.eps2 * (1.0 - 0x1p-20)), eps_hi = (float)(c.eps2 * (1.0 + 0x1p-20));
  const uint32_t regJ_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)regJ;  // LDS byte address
  float2 np = (c.flags & 4) ? init : make_float2(0.f, 0.f);  // nextPts[ptidx]
  int st = 1;
#pragma unroll
  for (int L = NL - 1; L >= 0; L--) {
    if (L > c.max_level) continue;
    const int cols = c.P.w[L], rows = c.P.h[L];
    const int stride = c.P.stride[L];
    const uint8_t* J = c.N.img[L] + (size_t)kPad * stride + kPad;
    const float sc = 1.f / (float)(1 << L);
    float nextX, nextY;
    if (L == c.max_level) {
      if (c.flags & 4) {
        nextX = np.x * sc;
        nextY = np.y * sc;
      } else {
        nextX = prev0.x * sc;
        nextY = prev0.y * sc;
      }
    } else {
      nextX = np.x * 2.f;
      nextY = np.y * 2.f;
    }
    np = make_float2(nextX, nextY);
    if (wave_any(!win_ok[L] || !eig_ok[L])) {
      if (L == 0) st = 0;
      continue;
    }
    // The iteration, shaped for a lone wave (every instruction — scalar ones too — costs an issue slot of
    // four cycles): ONE loop with ONE exit test at its bottom.  The bottom of iteration j already has the
    // next window position, so it floors it, takes its offsets (ox, oy) inside the part of the staged
    // region that is also inside the image — one unsigned compare per axis covers OpenCV's "window left
    // the image" break and the re-staging — and folds that into the same exit mask as the convergence
    // pre-test, the oscillation test and the iteration count.  What a stop means is sorted out after the
    // loop, in OpenCV's order (converged, oscillating, count, outside), and the loop is re-entered after a
    // re-staging or an fp64 tie-break that said "not converged".
    f2 nxt = {nextX - halfWin, nextY - halfWin};
    f2 fl = {floorf(nxt.x), floorf(nxt.y)};
    int inx = (int)fl.x, iny = (int)fl.y;
    f2 d = {0.f, 0.f}, prevD = {__builtin_inff(), __builtin_inff()};  // (j == 0: |d + inf| <= 0.01 cannot hold)
    int j = 0;
    bool level_done = c.max_count <= 0;  // (a TermCriteria without iterations: the level passes its start on)
    if (wave_any(inx < -WIN || inx >= cols || iny < -WIN || iny >= rows)) {  // iteration 0's own check
      if (L == 0) st = 0;
      level_done = true;
    }
    while (!level_done) {
      // stage the search region around the current window: 34 x 40 B, dword-aligned origin
      int rx0, ry0;
      {
        uint32_t reg[(kLkRegDw + 63) / 64];
        lk_region_load(J, stride, rows, inx, iny, lane, rgl, rx0, ry0, reg);
        lk_region_store(regJ, lane, reg);
      }
      // window origins the region serves AND the image allows: [lox, lox + rngx] x [loy, loy + rngy]
      const int lox = max(rx0, -WIN), loy = max(ry0, -WIN);
      const uint32_t rngx = (uint32_t)(min(rx0 + (kLkRegW - (WIN + 1)), cols - 1) - lox);
      const uint32_t rngy = (uint32_t)(min(ry0 + (kLkRegH - (WIN + 1)), rows - 1) - loy);
      // LDS byte address of this lane's first tap for the window at (lox, loy)
      const uint32_t cbo = regJ_b + (uint32_t)lane_bo + mad_u24_regw((uint32_t)(loy - ry0), (uint32_t)(lox - rx0));
      uint32_t ox = (uint32_t)(inx - lox), oy = (uint32_t)(iny - loy);
      float d2 = 0.f;
      f2 osum = {0.f, 0.f};
      bool restage = false;
      for (;;) {  // re-entered without re-staging after an undecided tie-break
        for (;;) {
          // bilinear weights: rint(w * 2^14) for w in [0,1] is the low half of the bit pattern of
          // fma(w, 2^14, 2^23) (one RNE rounding to an integer), which v_perm packs straight into
          // the int16 pairs {iw00,iw01} {iw10,iw11}
          const f2 fr = nxt - fl, ofr = 1.f - fr;  // {fa, fb}, {1 - fa, 1 - fb}
          const float kMagic = 8388608.f, kW = (float)(1 << W_BITS);
          const f2 cross = fr * f2{ofr.y, ofr.x};  // {fa * ofb, fb * ofa}: one v_pk_mul_f32 (op_sel swaps the halves)
          const f2 bc = __builtin_elementwise_fma(cross, f2{kW, kW}, f2{kMagic, kMagic});  // v_pk_fma_f32: one rounding each
          const uint32_t b00 = __float_as_uint(__fmaf_rn(ofr.x * ofr.y, kW, kMagic));
          const uint32_t b01 = __float_as_uint(bc.x);
          const uint32_t b10 = __float_as_uint(bc.y);
          const uint32_t iw11 = ((1u << W_BITS) + 3u * 0x4B000000u) - (b00 + b01 + b10);
          const uint32_t W0 = __builtin_amdgcn_perm(b10, b00, 0x05040100u);   // {iw00, iw10}: the left column's rows
          const uint32_t W1 = __builtin_amdgcn_perm(iw11, b01, 0x05040100u);  // {iw01, iw11}: the right column's
          // NP + 1 bytes of two consecutive staged rows: 3 aligned dwords per row + funnel shifts
          // (v_alignbyte_b32 takes the byte count from the low two bits of its third operand)
          const uint32_t bo = mad_u24_regw(oy, ox) + cbo;
          lds_cu32* rp = (lds_cu32*)(uintptr_t)(bo & ~3u);
          const uint32_t a0 = rp[0], a1 = rp[1], a2 = rp[2];
          const uint32_t c0 = rp[kLkRegW / 4], c1 = rp[kLkRegW / 4 + 1], c2 = rp[kLkRegW / 4 + 2];
          uint32_t w0[3], w1[3];
          w0[0] = __builtin_amdgcn_alignbyte(a1, a0, bo);
          w0[1] = __builtin_amdgcn_alignbyte(a2, a1, bo);
          w1[0] = __builtin_amdgcn_alignbyte(c1, c0, bo);
          w1[1] = __builtin_amdgcn_alignbyte(c2, c1, bo);
          if (NP == 8) {  // the ninth byte
            w0[2] = __builtin_amdgcn_alignbyte(a2, a2, bo);
            w1[2] = __builtin_amdgcn_alignbyte(c2, c2, bo);
          }
          // per column: its bytes of the two rows as an int16 pair (v_perm_b32); per pixel: its own column against
          // the left weights plus the next column against the right ones (v_dot2_i32_i16), rounding constant as
          // the accumulator seed — NP + 1 v_perm for NP pixels
          uint32_t t[2 * NPP], vc[NP + 1];
#pragma unroll
          for (int k = 0; k <= NP; k++) {
            const uint32_t sel = 0x0c000c00u | ((uint32_t)(4 + (k & 3)) << 16) | (uint32_t)(k & 3);
            vc[k] = __builtin_amdgcn_perm(w1[k >> 2], w0[k >> 2], sel);
          }
#pragma unroll
          for (int k = 0; k < NP; k++)
            t[k] = (uint32_t)sdot2(vc[k + 1], W1, sdot2_seed(vc[k], W0, 1 << (W_BITS - 5 - 1)));  // in [1, 2^22]
#pragma unroll
          for (int k = NP; k < 2 * NPP; k++) t[k] = 0;
          // CV_DESCALE(.., 9) of two pixels at once: bytes 1..2 of each sum, packed shift by one more
          // bit, packed subtract of the patch, dot with the packed derivatives
          float b1, b2;
          if (ACCUM == 2) {
            // pair sums diff_j Ix_j + diff_(j+4) Ix_(j+4) (exact in int32: pmaddwd) -> the float chains
            int vx[5], vy[5];
#pragma unroll
            for (int m = 0; m < NPP; m++) {
              const uint32_t hi8 = __builtin_amdgcn_perm(t[m + 4], t[m], 0x06050201u);
              const us2 val = __builtin_bit_cast(us2, hi8) >> (unsigned short)1;
              const uint32_t diff = __builtin_bit_cast(uint32_t, __builtin_bit_cast(ss2, val) - __builtin_bit_cast(ss2, pIp[L][m]));
              vx[m] = sdot2_zero(diff, pIxp[L][m]);
              vy[m] = sdot2_zero(diff, pIyp[L][m]);
              if (m == 0) {
                vx[4] = sdot2_zero(diff, pIxB[L]);
                vy[4] = sdot2_zero(diff, pIyB[L]);
              }
            }
            lk_float_sums_b(vx, vy, term_base, *lnf, b1, b2);
          } else {
            int sb1 = 0, sb2 = 0;  // kRedSeed + 7 terms of <= 2^25 each
#pragma unroll
            for (int m = 0; m < NPP; m++) {
              const uint32_t hi8 = __builtin_amdgcn_perm(t[2 * m + 1], t[2 * m], 0x06050201u);
              const us2 val = __builtin_bit_cast(us2, hi8) >> (unsigned short)1;
              const uint32_t diff = __builtin_bit_cast(uint32_t, __builtin_bit_cast(ss2, val) - __builtin_bit_cast(ss2, pIp[L][m]));
              if (m == 0) {
                sb1 = sdot2_seed(diff, pIxp[L][m], kRedSeed);
                sb2 = sdot2_seed(diff, pIyp[L][m], kRedSeed);
              } else {
                sb1 = sdot2(diff, pIxp[L][m], sb1);
                sb2 = sdot2(diff, pIyp[L][m], sb2);
              }
            }
            wave_sum2_exact(sb1, sb2, rl, b1, b2);  // already scaled by FLT_SCALE
          }
          d = f2{(A12[L] * b2 - A22[L] * b1) * Dinv[L], (A12[L] * b1 - A11[L] * b2) * Dinv[L]};
          // |delta|^2 <= eps^2 is decided in fp64 by OpenCV; an fp32 evaluation (relative error
          // < 2^-22) settles it unless it lands within 2^-20 of the threshold
          d2 = __fmaf_rn(d.x, d.x, d.y * d.y);
          osum = d + prevD;
          prevD = d;
          nxt += d;
          fl = f2{floorf(nxt.x), floorf(nxt.y)};
          inx = (int)fl.x;
          iny = (int)fl.y;
          ox = (uint32_t)(inx - lox);
          oy = (uint32_t)(iny - loy);
          j++;
          // ONE exit test: "out of iterations" rides on the convergence pre-test's threshold (a scalar select; the
          // negated >= also stops on a NaN, which then runs the count down instead of spinning)
          const float thr = j >= c.max_count ? __builtin_inff() : eps_hi;
          // (double)|x| < 0.01  <=>  |x| <= 0.01f: 0.01f is the largest float below the real 0.01
          if (bal(!(d2 >= thr)) | bal(fmaxf(fabsf(osum.x), fabsf(osum.y)) <= 0.01f) | bal(ox > rngx) | bal(oy > rngy)) break;
        }
        // (what follows must not share its compares with the loop's exit test: the compiler would carry every
        // one of them out of the loop as a mask of its own, a dozen scalar instructions per iteration)
        asm volatile("" : "+v"(d2), "+v"(ox), "+v"(oy));
        asm volatile("" : "+v"(osum));
        np = make_float2(nxt.x + halfWin, nxt.y + halfWin);
        lanemask_t conv = bal(d2 <= eps_lo);
        if (__builtin_expect((bal(d2 > eps_lo) & bal(d2 < eps_hi)) != 0, 0)) {
          asm volatile("; fp64 tie-break" ::: "memory");  // keep this a branch, not a select
          conv = bal((double)d.x * (double)d.x + (double)d.y * (double)d.y <= c.eps2);
        }
        if (conv) {
          level_done = true;
          break;
        }
        if (bal(fabsf(osum.x) <= 0.01f) & bal(fabsf(osum.y) <= 0.01f)) {
          np.x -= d.x * 0.5f;
          np.y -= d.y * 0.5f;
          level_done = true;
          break;
        }
        if (j >= c.max_count) {
          level_done = true;
          break;
        }
        // the next iteration's own checks
        if (wave_any(inx < -WIN || inx >= cols || iny < -WIN || iny >= rows)) {
          if (L == 0) st = 0;
          level_done = true;
          break;
        }
        if (wave_any(ox > rngx || oy > rngy)) {  // the window drifted out of the staged region
          restage = true;
          break;
        }
      }
      (void)restage;
    }
    if (st && L == 0) {  // the `err` block of the tracker re-validates the final position
      const float fx = np.x - halfWin, fy = np.y - halfWin;
      const int ix = (int)floorf(fx), iy = (int)floorf(fy);
      if (ix < -WIN || ix >= cols || iy < -WIN || iy >= rows) st = 0;
    }
  }
  np_out = np;
  st_out = st;
}

struct LkKernelArgs {
  LkCall fwd;
  LkCall back;       // used when have_back: prevPts = fwd result, init = fwd prevPts
  int have_back;
  const float2* prev_pts;
  const float2* init_pts;  // fwd initial flow (USE_INITIAL_FLOW), may be NULL
  float2* next_pts;        // fwd result
  uint8_t* status;         // fwd status
  float2* back_pts;        // back result
  uint8_t* back_status;
  const int* n_ptr;        // device count (may be NULL -> n_max)
  int n_max;
  const unsigned long long* poll_slots;  // see LkArgs::poll_*
  const unsigned long long* poll_done;
  uint32_t poll_seq;
  int poll_from;
  int* poll_err;
  unsigned long long* chain_out;  // see LkArgs::chain_*
  const unsigned long long* chain_in;
  uint32_t chain_seq;
  unsigned long long poll_ticks, chain_ticks;  // bounds of the two waits (wall_clock64: 100 MHz)
  const uint32_t* gate_ptr;  // see LkArgs::gate_*
  uint32_t gate_val;
};

// producer side of a chained launch: point `pt`'s forward result (alive = its status), or alive = 0
// for a point that does not exist in this launch
__device__ __forceinline__ void chain_publish(const LkKernelArgs& a, int pt, int lane, float2 p, int alive) {
  if (!a.chain_out || lane != 0) return;
  const unsigned long long hi = (unsigned long long)((a.chain_seq << 2) | (uint32_t)(alive ? 1 : 0)) << 32;
  __hip_atomic_store(&a.chain_out[2 * pt], hi | __float_as_uint(p.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&a.chain_out[2 * pt + 1], hi | __float_as_uint(p.y), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

template <int ACCUM>
__device__ __forceinline__ void lk_kernel_body(const LkKernelArgs& a, uint32_t (*regJ_s)[kLkRegStore], uint32_t acc_t,
                                               float* acc_ta, const LkLaneF32* lnf) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int pt = blockIdx.x * kLkWaves + wave;
  const int n = a.n_ptr ? *a.n_ptr : a.n_max;
  if (pt >= a.n_max) return;
  if (pt >= n) {
    chain_publish(a, pt, lane, make_float2(0.f, 0.f), 0);
    return;
  }
  float2 prev0;
  if (a.chain_in) {
    // the previous frame's launch is still running (or about to): wait for this point's result
    const unsigned long long t0 = wall_clock64();
    unsigned long long vx, vy;
    if (!a.chain_ticks) {  // (fault injection: give up without looking, whether the result is there or not)
      if (lane == 0) *a.poll_err = 1;
      return;
    }
    for (;;) {
      vx = __hip_atomic_load(&a.chain_in[2 * pt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vy = __hip_atomic_load(&a.chain_in[2 * pt + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(vx >> 34) == a.chain_seq && (uint32_t)(vy >> 34) == a.chain_seq) break;
      if (wall_clock64() - t0 > a.chain_ticks) {  // (kTicksChain = 40 ms: the producer itself may wait 20 ms)
        if (lane == 0) *a.poll_err = 1;
        return;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!((uint32_t)(vx >> 32) & 1u)) {  // lost (or never existed) in the previous frame
      if (lane == 0) {
        a.status[pt] = 0;
        if (a.have_back) a.back_status[pt] = 0;
      }
      return;
    }
    prev0 = make_float2(__uint_as_float((uint32_t)vx), __uint_as_float((uint32_t)vy));
    if (a.gate_ptr) {
      // ... and for the `next` pyramid: its prefetch sequence may not even have been issued when this launch was made
      for (;;) {
        const uint32_t gv = __hip_atomic_load(a.gate_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(gv - a.gate_val) >= 0) break;
        if (wall_clock64() - t0 > a.chain_ticks) {
          if (lane == 0) *a.poll_err = 1;
          return;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (what the sequence's kernels wrote, not what this CU may have cached)
    }
  } else if (a.poll_slots && pt >= a.poll_from) {
    // this point is a corner k_select may still be about to accept: wait for its slot (or for the
    // final count to rule it out).  wall_clock64 ticks at 100 MHz: give up after kTicksPoll = 20 ms.
    const unsigned long long t0 = wall_clock64();
    unsigned long long v;
    for (;;) {
      v = __hip_atomic_load(&a.poll_slots[pt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(v >> 32) == a.poll_seq) break;
      const unsigned long long d = __hip_atomic_load(a.poll_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(d >> 32) == a.poll_seq && pt >= (int)(uint32_t)d) {  // not accepted
        chain_publish(a, pt, lane, make_float2(0.f, 0.f), 0);
        return;
      }
      if (wall_clock64() - t0 > a.poll_ticks) {
        if (lane == 0) *a.poll_err = 1;
        chain_publish(a, pt, lane, make_float2(0.f, 0.f), 0);
        return;
      }
      __builtin_amdgcn_s_sleep(16);
    }
    prev0 = make_float2((float)(uint32_t)(v & 0xffffu), (float)(uint32_t)((v >> 16) & 0xffffu));
  } else {
    prev0 = a.prev_pts[pt];
  }
  const float2 init = (a.fwd.flags & 4) ? a.init_pts[pt] : make_float2(0.f, 0.f);
  float2 np;
  int st;
  lk_point<ACCUM>(a.fwd, prev0, init, regJ_s[wave], acc_t, acc_ta, lnf, lane, np, st);
  chain_publish(a, pt, lane, np, st);
  if (lane == 0) {
    a.next_pts[pt] = np;
    a.status[pt] = (uint8_t)st;
  }
  if (a.have_back) {
    float2 bp;
    int bs;
    lk_point<ACCUM>(a.back, np, prev0, regJ_s[wave], acc_t, acc_ta, lnf, lane, bp, bs);
    if (lane == 0) {
      a.back_pts[pt] = bp;
      a.back_status[pt] = (uint8_t)bs;
    }
  }
}

__global__ __launch_bounds__(64 * kLkWaves) void k_lk(LkKernelArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t regJ_s[kLkWaves][kLkRegStore];
  lk_kernel_body<1>(a, regJ_s, 0u, nullptr, nullptr);
}
// lk_accum 2: float sums in the reference build's order (see lk_float_sums_A / _b)
__global__ __launch_bounds__(64 * kLkWaves) void k_lk_f32(LkKernelArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t regJ_s[kLkWaves][kLkRegStore];
  __shared__ __attribute__((aligned(16))) int term_all[kLkWaves][kF32TermWords];
  __shared__ __attribute__((aligned(16))) float term_a[kLkWaves][kLkTermWordsA];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < kF32TermWords; i += 64) term_all[wave][i] = 0;  // (the words a segment does not use stay 0)
  // (the A table: only the slots no pixel lane writes — the fifth slot of the vector chains' rows — need their zeros)
  {
    const int l = lane < 63 ? lane : 62, t = l / kLkWin, rw = l - t * kLkWin;  // a lane per (type, row)
    float* z = term_a[wave] + 5 * t * kLkChainWordsA + rw * 5 + 4;
#pragma unroll
    for (int L = 0; L < kMaxLevels; L++)
#pragma unroll
      for (int k = 0; k < 4; k++) z[L * kLkTermLevelA + k * kLkChainWordsA] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  const LkLaneF32 lnf =
      lk_lane_f32(lane, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)term_a[wave]);
  const uint32_t tb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)term_all[wave];
  lk_kernel_body<2>(a, regJ_s, tb, term_a[wave], &lnf);
}

void launch_lk(hipStream_t s, const LkArgs& f, const LkArgs* b, float2* back_pts,
               uint8_t* back_status) {
  if (f.n_max <= 0) return;
  LkKernelArgs a;
  a.fwd.P = f.P;
  a.fwd.N = f.N;
  a.fwd.max_level = f.max_level;
  a.fwd.max_count = f.max_count;
  a.fwd.eps2 = f.eps2;
  a.fwd.flags = f.flags;
  a.have_back = b ? 1 : 0;
  a.back = a.fwd;
  if (b) {
    a.back.P = b->P;
    a.back.N = b->N;
    a.back.max_level = b->max_level;
    a.back.max_count = b->max_count;
    a.back.eps2 = b->eps2;
    a.back.flags = b->flags;
  }
  a.prev_pts = f.prev_pts;
  a.init_pts = f.init_pts;
  a.next_pts = f.next_pts;
  a.status = f.status;
  a.back_pts = back_pts;
  a.back_status = back_status;
  a.n_ptr = f.n_ptr;
  a.n_max = f.n_max;
  a.poll_slots = f.poll_slots;
  a.poll_done = f.poll_done;
  a.poll_seq = f.poll_seq;
  a.poll_from = f.poll_from;
  a.poll_err = f.poll_err;
  a.chain_out = f.chain_out;
  a.chain_in = f.chain_in;
  a.chain_seq = f.chain_seq;
  a.poll_ticks = f.poll_ticks;
  a.chain_ticks = f.chain_ticks;
  a.gate_ptr = f.gate_ptr;
  a.gate_val = f.gate_val;
  if (f.accum == 2)
    launch_k(k_lk_f32, dim3((a.n_max + kLkWaves - 1) / kLkWaves), dim3(64 * kLkWaves), 0, s, a);
  else
    launch_k(k_lk, dim3((a.n_max + kLkWaves - 1) / kLkWaves), dim3(64 * kLkWaves), 0, s, a);
}

// ============================================================================ Arc*
// event_detector.cc:14-22
__constant__ int8_t c_small[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                      {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                      {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
__constant__ int8_t c_large[20][2] = {{0, 4},   {1, 4},   {2, 3},   {3, 2},  {4, 1},  {4, 0},  {4, -1},
                                      {3, -2},  {2, -3},  {1, -4},  {0, -4}, {-1, -4}, {-2, -3}, {-3, -2},
                                      {-4, -1}, {-4, 0},  {-4, 1},  {-3, 2}, {-2, 3},  {-1, 4}};

// One ring of isCorner (event_detector.cc:337-435 small / :438-541 large), on RANKS:
// isCorner only ever compares ring values with each other (>, >=, <
// and minima of them), so replacing every value by the number of ring values strictly below it
// (equal values get equal ranks) leaves every decision unchanged — and the ranks (4 or 5 bits) of a
// whole ring fit one or two 64-bit registers, which a lane can index with a variable shift: no LDS
// column, no LDS round trip per step of the arc walk.
template <int N>
struct RingRanks {
  unsigned long long lo, hi;  // N <= 16: 4 bits each in lo; N = 20: 5 bits each, 10 per word
  __device__ __forceinline__ int get(int i) const {
    if (N <= 16) return (int)((lo >> (4 * i)) & 15ull);
    return i < 10 ? (int)((lo >> (5 * i)) & 31ull) : (int)((hi >> (5 * (i - 10))) & 31ull);
  }
};

template <int N, int KMIN, int KMAX>
__device__ __forceinline__ bool arc_ring_ranks(const double (&v)[N]) {
  int r[N];
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = 0;
#pragma unroll
  for (int i = 1; i < N; i++)
#pragma unroll
    for (int j = 0; j < i; j++) {
      r[i] += v[j] < v[i] ? 1 : 0;
      r[j] += v[i] < v[j] ? 1 : 0;
    }
  RingRanks<N> rk;
  rk.lo = 0;
  rk.hi = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    if (N <= 16) rk.lo |= (unsigned long long)r[i] << (4 * i);
    else if (i < 10) rk.lo |= (unsigned long long)r[i] << (5 * i);
    else rk.hi |= (unsigned long long)r[i] << (5 * (i - 10));
  }
  int segment_new_min_t = r[0];
  int arc_right_idx = 0;
#pragma unroll
  for (int i = 1; i < N; i++)
    if (r[i] > segment_new_min_t) {
      segment_new_min_t = r[i];
      arc_right_idx = i;
    }
  int arc_left_idx = arc_right_idx == 0 ? N - 1 : arc_right_idx - 1;
  arc_right_idx = arc_right_idx == N - 1 ? 0 : arc_right_idx + 1;
  int arc_left_value = rk.get(arc_left_idx);
  int arc_right_value = rk.get(arc_right_idx);
  int arc_left_min_t = arc_left_value;
  int arc_right_min_t = arc_right_value;
  int newest_segment_size = KMIN;
#pragma unroll
  for (int iteration = 1; iteration < N; iteration++) {
    const bool right = arc_right_value > arc_left_value;
    const int val = right ? arc_right_value : arc_left_value;
    const int mn = right ? arc_right_min_t : arc_left_min_t;
    if (iteration < KMIN) {
      if (mn < segment_new_min_t) segment_new_min_t = mn;
    } else if (val >= segment_new_min_t) {
      newest_segment_size = iteration + 1;
      if (mn < segment_new_min_t) segment_new_min_t = mn;
    }
    if (right) {
      arc_right_idx = arc_right_idx == N - 1 ? 0 : arc_right_idx + 1;
      arc_right_value = rk.get(arc_right_idx);
      if (arc_right_value < arc_right_min_t) arc_right_min_t = arc_right_value;
    } else {
      arc_left_idx = arc_left_idx == 0 ? N - 1 : arc_left_idx - 1;
      arc_left_value = rk.get(arc_left_idx);
      if (arc_left_value < arc_left_min_t) arc_left_min_t = arc_left_value;
    }
  }
  return (newest_segment_size <= KMAX) ||
         ((newest_segment_size >= (N - KMAX)) && (newest_segment_size <= (N - KMIN)));
}

// EventDetector::isCorner (event_detector.cc:308-544) for a whole batch, in three kernels.
//
// isCorner is called after the whole batch is in the SAE (feature_tracker.cpp:356-368, then :458),
// and its two ring tests (:337-541) read nothing but the post-batch S[p] around (x,y): their result
// is a property of (pixel, polarity), not of the event.  Of the pre-check (:315)
//     if (et > L[p] + thr || L[!p] > L[p]) return false;
// only the first term looks at the event.  So
//   k_arc_mark  one lane per EVENT sets the flag byte of its (pixel, polarity) (plain stores: the
//               batch's 3-20 events per touched pixel collapse to one flag);
//   k_arc_map   a block owns kArcRegion consecutive pixels: it collects the region's flagged pairs
//               that pass  border / TS(y,x) != TS_LK_THRESHOLD / !(L[!p] > L[p])  into an LDS list
//               (clearing the flags for the next batch) and evaluates the small and the large ring
//               for those with all lanes busy, on the RANKS of the ring values packed into
//               registers (arc_ring_ranks: no LDS column, no LDS round trip per step of the arc
//               walk); the result is bit (pixel, polarity) of a bitmap;
//   k_arc_ev    one lane per EVENT streams the batch in order: flag = map bit && !(et > L[p] + thr)
//               (L is only fetched for events whose map bit is set), the blocked-mask test of
//               Event_FeaturesToTrack (feature_tracker.cpp:25) and the ordered in-block compaction
//               of the survivors (ballot + popcount) as before.
constexpr int kArcRegion = 256;  // pixels per k_arc_map block: 512 (pixel, polarity) flag bytes

__global__ __launch_bounds__(256) void k_arc_mark(const uint4* __restrict__ ev, uint32_t n, int W, int H,
                                                  uint8_t* __restrict__ touched) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t xy = ((const uint32_t*)ev)[4 * (size_t)i], pw = ((const uint32_t*)ev)[4 * (size_t)i + 3];
    const uint32_t x = xy & 0xffffu, y = xy >> 16;
    // (plain byte stores: every writer of a byte writes the same value)
    if (x < (uint32_t)W && y < (uint32_t)H) touched[2u * (y * (uint32_t)W + x) + ((pw & 0xffu) ? 1u : 0u)] = 1;
  }
}

__global__ __launch_bounds__(kArcBlock) void k_arc_map(ArcArgs a) {
  __shared__ uint16_t list_b[2 * kArcRegion];  // surviving pairs (2 * pixel + polarity, region-local)
  __shared__ uint32_t n_b;
  const uint32_t P = (uint32_t)a.W * (uint32_t)a.H;
  const uint32_t pair0 = blockIdx.x * (2u * kArcRegion);  // first (pixel, polarity) pair of the region
  if (threadIdx.x == 0) n_b = 0;
  __syncthreads();
  // thread t looks at pairs pair0 + 2t, 2t+1 (= the two polarities of pixel pair0/2 + t)
  {
    const uint32_t px = (pair0 >> 1) + threadIdx.x;
    uint32_t f = 0;
    if (px < P) {
      f = ((const uint16_t*)a.touched)[px];
      if (f) ((uint16_t*)a.touched)[px] = 0;  // (consumed: the next batch starts from cleared flags)
    }
    if (threadIdx.x < (2 * kArcRegion) / 32) {  // the region's result bits start out 0
      const uint32_t w = (pair0 >> 5) + threadIdx.x;
      if (w < (2u * P + 31u) / 32u) a.cmap[w] = 0;
    }
    if (f) {
      const int y = (int)(px / (uint32_t)a.W), x = (int)(px - (uint32_t)y * (uint32_t)a.W);
      bool ok = !(x < a.border || x >= a.W - a.border || y < a.border || y >= a.H - a.border);
      // (the rings reach 4 pixels out; the reference would index out of its matrices for MIN_DIST < 3)
      ok = ok && x >= 4 && y >= 4 && x < a.W - 4 && y < a.H - 4;
      if (ok && a.ts) ok = (double)a.ts[(size_t)(y + kPad) * a.ts_stride + x + kPad] != a.ts_lk_threshold;
      if (ok) {
        const double2 Lv = a.L2[px];
        // polarity p survives unless L[!p] > L[p]
        if ((f & 0x00ffu) && !(Lv.y > Lv.x)) list_b[atomicAdd(&n_b, 1u)] = (uint16_t)(2u * threadIdx.x);
        if ((f & 0xff00u) && !(Lv.x > Lv.y)) list_b[atomicAdd(&n_b, 1u)] = (uint16_t)(2u * threadIdx.x + 1u);
      }
    }
  }
  __syncthreads();
  const uint32_t nb = n_b;
  for (uint32_t i = threadIdx.x; i < nb; i += kArcBlock) {
    const uint32_t pair = pair0 + list_b[i], px = pair >> 1, pol = pair & 1u;
    const int y = (int)(px / (uint32_t)a.W), x = (int)(px - (uint32_t)y * (uint32_t)a.W);
    const double* S = (const double*)a.S2 + pol;
    double v16[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v16[k] = S[2 * ((size_t)(y + c_small[k][1]) * a.W + (x + c_small[k][0]))];
    if (!arc_ring_ranks<16, 4, 6>(v16)) continue;
    double v20[20];
#pragma unroll
    for (int k = 0; k < 20; k++) v20[k] = S[2 * ((size_t)(y + c_large[k][1]) * a.W + (x + c_large[k][0]))];
    if (arc_ring_ranks<20, 5, 8>(v20)) atomicOr(&a.cmap[pair >> 5], 1u << (pair & 31));
  }
}

__global__ __launch_bounds__(kArcBlock) void k_arc_ev(ArcArgs a) {
  __shared__ uint32_t wave_cnt[kArcBlock / 64];
  const uint32_t i = blockIdx.x * kArcBlock + threadIdx.x;
  bool corner = false;
  uint32_t x = 0, y = 0;
  if (i < a.n) {
    const uint4 e = ((const uint4*)a.ev)[i];
    x = e.x & 0xffffu;
    y = e.x >> 16;
    if (x < (uint32_t)a.W && y < (uint32_t)a.H) {
      const int pol = (e.w & 0xffu) ? 1 : 0;
      const uint32_t px = y * (uint32_t)a.W + x;
      const uint32_t pair = 2u * px + (uint32_t)pol;
      bool ok = (a.cmap[pair >> 5] >> (pair & 31)) & 1u;
      if (ok && a.mask_bits) ok = !((a.mask_bits[y * a.wpr + (x >> 5)] >> (x & 31)) & 1u);
      if (ok) {
        const double t_last = ((const double*)a.L2)[2 * (size_t)px + pol];
        if (ev_time(e.y, e.z) > __dadd_rn(t_last, a.filter_threshold)) ok = false;
      }
      corner = ok;
    }
    if (a.flags) a.flags[i] = corner ? 1 : 0;
  }
  if (a.cand_cnt) {
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const unsigned long long m = __ballot(corner);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wave_cnt[w];
    if (corner) {
      const uint32_t pos = blockIdx.x * kArcBlock + base + __popcll(m & ((1ull << lane) - 1ull));
      a.cand_xy[pos] = x | (y << 16);
      a.cand_idx[pos] = i;
      if (a.first_map) atomicMin(&a.first_map[y * (uint32_t)a.W + x], a.first_key | i);
    }
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int w = 0; w < kArcBlock / 64; w++) t += wave_cnt[w];
      a.cand_cnt[blockIdx.x] = t;
    }
  }
}

// Host batch on its way to the device (fe_evstage.cpp): the staged chunks sit in pinned host memory, which
// the device reads itself — a few dozen workgroups pulling 16 bytes per lane keep the PCIe link as busy as
// a copy engine does (tools/h2d_probe.hip on MI355X: 2.7 MB in 54 us = 52 GB/s against 65 us for
// hipMemcpyAsync, 64 KB in 6 us against 15) and, unlike hipMemcpyAsync, never make the runtime bring up
// another copy engine in the middle of a stream (7-9 ms inside the call that does, every other thread's copy
// calls waiting behind it).
__global__ __launch_bounds__(256) void k_stage_pull(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
// ... and with the chunks PACKED by the staging threads (fe_evstage.cpp stage_pack: 8 bytes per event — x | y << 16,
// nsec | polarity << 30 | (sec - the chunk's base second) << 31 — whenever a chunk's events allow it): half the bytes
// over PCIe, unpacked into the 16-byte records every kernel reads while they are written to the device buffer.  One
// descriptor per chunk of `epc` events says which form its slot of the pinned buffer holds; a packed chunk lies at the
// start of the 16 * epc bytes its raw form would take.  (Polarity as the reference reads it, != 0; the record's three
// padding bytes — which no kernel reads — come out as zero.)
// The descriptors are read ONCE per workgroup (into LDS): read per event they were a second, dependent trip over PCIe
// in every round of the loop (a plain call from pageable memory: 0.378-0.393 -> 0.362-0.370 ms).
constexpr int kStageDescLds = 1024;
__global__ __launch_bounds__(256) void k_stage_pull_packed(const uint8_t* __restrict__ src, uint4* __restrict__ dst, size_t n,
                                                           const uint2* __restrict__ desc, uint32_t epc) {
  __shared__ uint2 sd[kStageDescLds];
  const size_t nch = (n + epc - 1) / epc;
  const bool cached = nch <= (size_t)kStageDescLds;
  if (cached) {
    for (size_t k = threadIdx.x; k < nch; k += blockDim.x) sd[k] = desc[k];
    __syncthreads();
  }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t c = i / epc, j = i - c * epc;
    const uint2 d = cached ? sd[c] : desc[c];
    const uint8_t* base = src + c * (size_t)epc * 16;
    if (d.y) {
      const uint2 v = ((const uint2*)base)[j];
      dst[i] = make_uint4(v.x, d.x + (v.y >> 31), v.y & 0x3fffffffu, (v.y >> 30) & 1u);
    } else {
      dst[i] = ((const uint4*)base)[j];
    }
  }
}
void launch_stage_pull_packed(hipStream_t s, const void* pinned_src, void* dst, size_t bytes, const void* desc, uint32_t epc) {
  const size_t n = bytes / 16;
  if (!n) return;
  const unsigned grid = (unsigned)std::min<size_t>(64, (n + 255) / 256);
  launch_k(k_stage_pull_packed, dim3(grid), dim3(256), 0, s, (const uint8_t*)pinned_src, (uint4*)dst, n, (const uint2*)desc, epc);
}
void launch_stage_pull(hipStream_t s, const void* pinned_src, void* dst, size_t bytes) {
  const size_t n = bytes / 16;  // (event records: always whole 16-byte units)
  if (!n) return;
  const unsigned grid = (unsigned)std::min<size_t>(64, (n + 255) / 256);
  launch_k(k_stage_pull, dim3(grid), dim3(256), 0, s, (const uint4*)pinned_src, (uint4*)dst, n);
}

void launch_arc_mark(hipStream_t s, const ArcArgs& a) {
  if (!a.n) return;
  uint32_t grid = (a.n + 1023) / 1024;
  if (grid > 2048) grid = 2048;
  launch_k(k_arc_mark, dim3(grid), dim3(256), 0, s, (const uint4*)a.ev, a.n, a.W, a.H, a.touched);
}

void launch_arc_map(hipStream_t s, const ArcArgs& a) {
  const uint32_t P = (uint32_t)a.W * (uint32_t)a.H;
  launch_k(k_arc_map, dim3((P + kArcRegion - 1) / kArcRegion), dim3(kArcBlock), 0, s, a);
}

void launch_arc(hipStream_t s, const ArcArgs& a) {
  if (!a.n) return;
  launch_k(k_arc_ev, dim3((a.n + kArcBlock - 1) / kArcBlock), dim3(kArcBlock), 0, s, a);
}

// see launch_dedup (fe_kernels.h): one block per Arc* block, list rewritten in place
__global__ __launch_bounds__(kArcBlock) void k_dedup(uint32_t* __restrict__ cand_xy,
                                                     uint32_t* __restrict__ cand_idx,
                                                     uint32_t* __restrict__ cand_cnt,
                                                     const uint32_t* __restrict__ first_map,
                                                     uint32_t first_key, int W) {
  __shared__ uint32_t wave_cnt[kArcBlock / 64];
  const uint32_t b = blockIdx.x, c = cand_cnt[b];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  uint32_t xy = 0, idx = 0;
  bool keep = false;
  if (threadIdx.x < c) {
    xy = cand_xy[(size_t)b * kArcBlock + threadIdx.x];
    idx = cand_idx[(size_t)b * kArcBlock + threadIdx.x];
    keep = first_map[(xy >> 16) * (uint32_t)W + (xy & 0xffffu)] == (first_key | idx);
  }
  const unsigned long long m = __ballot(keep);
  if (lane == 0) wave_cnt[wave] = __popcll(m);
  __syncthreads();  // (also: every thread has read its entry before any is overwritten)
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < kArcBlock / 64; w++) {
    if (w < wave) base += wave_cnt[w];
    tot += wave_cnt[w];
  }
  if (keep) {
    const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
    cand_xy[(size_t)b * kArcBlock + pos] = xy;
    cand_idx[(size_t)b * kArcBlock + pos] = idx;
  }
  if (threadIdx.x == 0) cand_cnt[b] = tot;
}

void launch_dedup(hipStream_t s, uint32_t* cand_xy, uint32_t* cand_idx, uint32_t* cand_cnt,
                  uint32_t nblk, const uint32_t* first_map, uint32_t first_key, int W) {
  if (!nblk) return;
  launch_k(k_dedup, dim3(nblk), dim3(kArcBlock), 0, s, cand_xy, cand_idx, cand_cnt, first_map,
           first_key, W);
}

// ============================================================================ median blur
// exact median of the (2k+1)^2 neighbourhood, replicated borders: tile + halo in LDS, then an
// 8-step bisection on the value (the median is the smallest v with #(values < v+1) > n/2)
__global__ __launch_bounds__(256) void k_median(const uint8_t* src0, const uint8_t* src1, int src_stride,
                                                uint8_t* dst0, uint8_t* dst1, int dst_stride, int W,
                                                int H, int k) {
  constexpr int T = 16, MAXW = T + 2 * kMaxMedianK;
  __shared__ uint8_t tile[MAXW * MAXW];
  const uint8_t* src = blockIdx.z ? src1 : src0;
  uint8_t* dst = blockIdx.z ? dst1 : dst0;
  const int tw = T + 2 * k;
  const int x0 = blockIdx.x * T - k, y0 = blockIdx.y * T - k;
  for (int i = threadIdx.x; i < tw * tw; i += 256) {
    const int ty = i / tw, tx = i - ty * tw;
    const int yy = min(max(y0 + ty, 0), H - 1), xx = min(max(x0 + tx, 0), W - 1);
    tile[ty * MAXW + tx] = src[(size_t)yy * src_stride + xx];
  }
  __syncthreads();
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x = blockIdx.x * T + lx, y = blockIdx.y * T + ly;
  if (x >= W || y >= H) return;
  const int ks = 2 * k + 1, need = (ks * ks) / 2 + 1;
  int r = 0;
  for (int bit = 7; bit >= 0; bit--) {
    const int t = r | (1 << bit);
    int cnt = 0;
    for (int dy = 0; dy < ks; dy++)
      for (int dx = 0; dx < ks; dx++) cnt += tile[(ly + dy) * MAXW + lx + dx] < t;
    if (cnt < need) r = t;
  }
  dst[(size_t)y * dst_stride + x] = (uint8_t)r;
}

void launch_median(hipStream_t s, const uint8_t* src0, const uint8_t* src1, int src_stride,
                   uint8_t* dst0, uint8_t* dst1, int dst_stride, int W, int H, int k, int nimg) {
  launch_k(k_median, dim3((W + 15) / 16, (H + 15) / 16, nimg), dim3(256), 0, s, src0, src1,
                     src_stride, dst0, dst1, dst_stride, W, H, k);
}

// ============================================================================ goodFeaturesToTrack
// (see fe_kernels.h; every float operation in the order of the oracle's restatement)
__device__ __forceinline__ uint32_t f32_order_key(float v) {  // monotonic float -> uint
  const uint32_t b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

__global__ __launch_bounds__(256) void k_gftt_cov(GfttArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.max_key = 0u;
  if (x >= a.W || y >= a.H) return;
  double scale = (double)(1 << (3 - 1)) * 3;
  scale *= 255.0;
  scale = 1.0 / scale;
  const float fs = (float)scale, k0 = 2.f * fs, k1 = 1.f * fs;
  const uint8_t* p = a.img + (ptrdiff_t)y * a.stride + x;  // borders are materialised (reflect-101)
  const int s = a.stride;
  const int a00 = p[-s - 1], a01 = p[-s], a02 = p[-s + 1];
  const int a10 = p[-1], a12 = p[1];
  const int a20 = p[s - 1], a21 = p[s], a22 = p[s + 1];
  const int r0 = a02 - a00, r1 = a12 - a10, r2 = a22 - a20;
  const float dx = __fadd_rn(__fmul_rn((float)(r0 + r2), k1), __fmul_rn((float)r1, k0));
  const float R0 = __fadd_rn(__fmul_rn((float)(a00 + a02), k1), __fmul_rn((float)a01, k0));
  const float R2 = __fadd_rn(__fmul_rn((float)(a20 + a22), k1), __fmul_rn((float)a21, k0));
  const float dy = __fsub_rn(R2, R0);
  a.cov[(size_t)y * a.W + x] = make_float4(__fmul_rn(dx, dx), __fmul_rn(dx, dy), __fmul_rn(dy, dy), 0.f);
}

__global__ __launch_bounds__(256) void k_gftt_rowsum(GfttArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.W || y >= a.H) return;
  const float4* r = a.cov + (size_t)y * a.W;
  const float4 l = r[reflect101(x - 1, a.W)], c = r[x], rr = r[reflect101(x + 1, a.W)];
  a.rowsum[(size_t)y * a.W + x] =
      make_float4(__fadd_rn(__fadd_rn(l.x, c.x), rr.x), __fadd_rn(__fadd_rn(l.y, c.y), rr.y),
                  __fadd_rn(__fadd_rn(l.z, c.z), rr.z), 0.f);
}

// one thread per column; the running column sum of OpenCV's ColumnSum, rows -1 .. H in order
__global__ __launch_bounds__(64) void k_gftt_eig(GfttArgs a) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= a.W) return;
  const int W = a.W, H = a.H;
  auto row = [&](int y) { return a.rowsum[(size_t)reflect101(y, H) * W + x]; };
  float4 prev = row(-1), cur = row(0);
  float s0 = __fadd_rn(__fadd_rn(0.f, prev.x), cur.x), s1 = __fadd_rn(__fadd_rn(0.f, prev.y), cur.y),
        s2 = __fadd_rn(__fadd_rn(0.f, prev.z), cur.z);
  uint32_t best = 0u;
  constexpr int kAhead = 8;
  for (int y0 = 0; y0 < H; y0 += kAhead) {
    float4 nx[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; k++) nx[k] = row(min(y0 + k, H - 1) + 1);
#pragma unroll
    for (int k = 0; k < kAhead; k++) {
      const int y = y0 + k;
      if (y >= H) break;
      const float c0 = __fadd_rn(s0, nx[k].x), c1 = __fadd_rn(s1, nx[k].y), c2 = __fadd_rn(s2, nx[k].z);
      s0 = __fsub_rn(c0, prev.x);
      s1 = __fsub_rn(c1, prev.y);
      s2 = __fsub_rn(c2, prev.z);
      prev = cur;
      cur = nx[k];
      const float aa = __fmul_rn(c0, 0.5f), bb = c1, cc = __fmul_rn(c2, 0.5f);
      const float d = __fsub_rn(aa, cc);
      const float e = __fsub_rn(__fadd_rn(aa, cc), sqrtf(__fadd_rn(__fmul_rn(d, d), __fmul_rn(bb, bb))));
      a.eig[(size_t)y * W + x] = e;
      const bool allowed = !a.mask_bits || !((a.mask_bits[y * a.wpr + (x >> 5)] >> (x & 31)) & 1u);
      if (allowed) best = max(best, f32_order_key(e));
    }
  }
  // wave maximum, one atomic per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o));
  if (threadIdx.x == 0 && best) atomicMax(a.max_key, best);
}

void launch_gftt_response(hipStream_t s, const GfttArgs& a) {
  const dim3 grid((a.W + 63) / 64, (a.H + 3) / 4);
  launch_k(k_gftt_cov, grid, dim3(256), 0, s, a);
  launch_k(k_gftt_rowsum, grid, dim3(256), 0, s, a);
  launch_k(k_gftt_eig, dim3((a.W + 63) / 64), dim3(64), 0, s, a);
}

// threshold + 3x3 local maximum + mask; block b owns pixels [b*kArcBlock, (b+1)*kArcBlock) in
// row-major order and leaves its candidates in that order
__global__ __launch_bounds__(kArcBlock) void k_gftt_collect(GfttArgs a) {
  __shared__ uint32_t wave_cnt[kArcBlock / 64];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const uint32_t i = blockIdx.x * kArcBlock + threadIdx.x;
  const int W = a.W, H = a.H;
  const uint32_t mk = *a.max_key;
  bool take = false;
  float val = 0.f;
  int x = 0, y = 0;
  if (mk && i < (uint32_t)W * (uint32_t)H) {
    y = (int)(i / (uint32_t)W);
    x = (int)(i - (uint32_t)y * (uint32_t)W);
    const float thr = (float)__dmul_rn((double)f32_from_order_key(mk), a.quality);
    if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {
      const float* e = a.eig + (size_t)y * W + x;
      auto th = [&](float v) { return v > thr ? v : 0.f; };
      val = th(e[0]);
      if (val != 0.f) {
        float m = val;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
          for (int dx = -1; dx <= 1; dx++) m = fmaxf(m, th(e[dy * W + dx]));
        const bool allowed = !a.mask_bits || !((a.mask_bits[y * a.wpr + (x >> 5)] >> (x & 31)) & 1u);
        take = val == m && allowed;
      }
    }
  }
  const unsigned long long mb = __ballot(take);
  if (lane == 0) wave_cnt[wave] = __popcll(mb);
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; w++) base += wave_cnt[w];
  if (take) {
    const uint32_t pos = base + __popcll(mb & ((1ull << lane) - 1ull));
    a.cand_xy[(size_t)blockIdx.x * kArcBlock + pos] = (uint32_t)x | ((uint32_t)y << 16);
    a.cand_val[(size_t)blockIdx.x * kArcBlock + pos] = __float_as_uint(val);
  }
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kArcBlock / 64; w++) t += wave_cnt[w];
    a.cand_cnt[blockIdx.x] = t;
  }
}

void launch_gftt_collect(hipStream_t s, const GfttArgs& a) {
  const uint32_t nblk = ((uint32_t)a.W * a.H + kArcBlock - 1) / kArcBlock;
  launch_k(k_gftt_collect, dim3(nblk), dim3(kArcBlock), 0, s, a);
}

__global__ __launch_bounds__(256) void k_gftt_sortprep(const uint32_t* __restrict__ comp_xy,
                                                       const uint32_t* __restrict__ comp_val, uint32_t n,
                                                       uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals,
                                                       uint32_t* __restrict__ ghist,
                                                       uint32_t* __restrict__ lookback,
                                                       uint32_t lookback_words) {
  __shared__ uint32_t h[4 << 8];
  for (int i = threadIdx.x; i < (4 << 8); i += 256) h[i] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t src = n - 1u - i;
    const uint32_t key = ~comp_val[src];  // positive floats: larger value -> smaller key
    keys[i] = key;
    vals[i] = comp_xy[src];
    for (int p = 0; p < 4; p++) atomicAdd(&h[(p << 8) + ((key >> (8 * p)) & 255u)], 1u);
  }
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < lookback_words;
       i += gridDim.x * blockDim.x)
    lookback[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < (4 << 8); i += 256)
    if (h[i]) atomicAdd(&ghist[i], h[i]);
}

void launch_gftt_sortprep(hipStream_t s, const uint32_t* comp_xy, const uint32_t* comp_val, uint32_t n,
                          uint32_t* keys, uint32_t* vals, uint32_t* ghist, uint32_t* lookback,
                          uint32_t lookback_words) {
  uint32_t grid = (n + 1023) / 1024;
  grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
  launch_k(k_gftt_sortprep, dim3(grid), dim3(256), 0, s, comp_xy, comp_val, n, keys, vals,
                     ghist, lookback, lookback_words);
}

// ============================================================================ greedy selection
// Event_FeaturesToTrack (feature_tracker.cpp:13-38): candidates in stream order; accept iff the
// pixel is not blocked; stamp cv::circle(r = MIN_DIST, filled) [OpenCV midpoint disc]; stop at
// max_corners.  k_compact turns the per-block candidate lists into one ordered list (parallel),
// k_select is the inherently sequential greedy: ONE wave, bitmap of blocked pixels + new discs in
// LDS (seeded with Event_setMask's bitmap, so k_arc can run before that mask exists — the mask
// test of feature_tracker.cpp:25 commutes with the corner test), 64 candidates per sub-chunk.
// (many blocks: the counts of every 64 consecutive blocks are summed first, so that a block adds up
// nblk/64 + 64 numbers instead of nblk)
constexpr int kCompactGroup = 64;
__global__ __launch_bounds__(256) void k_compact_groups(const uint32_t* __restrict__ cand_cnt, uint32_t nblk,
                                                        uint32_t* __restrict__ grp) {
  const uint32_t gi = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lo = gi * kCompactGroup;
  if (lo >= nblk) return;
  uint32_t s = 0;
  for (uint32_t j = lo; j < min(lo + kCompactGroup, nblk); j++) s += cand_cnt[j];
  grp[gi] = s;
}

__global__ __launch_bounds__(kArcBlock) void k_compact(const uint32_t* __restrict__ cand_xy,
                                                       const uint32_t* __restrict__ cand_idx,
                                                       const uint32_t* __restrict__ cand_cnt,
                                                       uint32_t nblk, uint32_t* __restrict__ comp_xy,
                                                       uint32_t* __restrict__ comp_idx,
                                                       uint32_t* __restrict__ total,
                                                       const uint32_t* __restrict__ grp) {
  __shared__ uint32_t part[kArcBlock / 64];
  const uint32_t b = blockIdx.x;
  // exclusive prefix of the counts of all earlier blocks (every block recomputes its own)
  uint32_t s = 0;
  if (grp) {
    const uint32_t g = b / kCompactGroup;
    for (uint32_t j = threadIdx.x; j < g; j += kArcBlock) s += grp[j];
    for (uint32_t j = g * kCompactGroup + threadIdx.x; j < b; j += kArcBlock) s += cand_cnt[j];
  } else {
    for (uint32_t j = threadIdx.x; j < b; j += kArcBlock) s += cand_cnt[j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane_id() == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < kArcBlock / 64; w++) off += part[w];
  const uint32_t c = cand_cnt[b];
  if (threadIdx.x < c) {
    comp_xy[off + threadIdx.x] = cand_xy[(size_t)b * kArcBlock + threadIdx.x];
    comp_idx[off + threadIdx.x] = cand_idx[(size_t)b * kArcBlock + threadIdx.x];
  }
  if (b == nblk - 1 && threadIdx.x == 0) *total = off + c;
}

void launch_compact(hipStream_t s, const uint32_t* cand_xy, const uint32_t* cand_idx,
                    const uint32_t* cand_cnt, uint32_t nblk, uint32_t* comp_xy, uint32_t* comp_idx,
                    uint32_t* total, uint32_t* grp_scratch) {
  if (!nblk) return;
  const bool grouped = grp_scratch && nblk > 2048;  // (below that the extra launch costs more than it saves)
  if (grouped) {
    const uint32_t ngrp = (nblk + kCompactGroup - 1) / kCompactGroup;
    launch_k(k_compact_groups, dim3((ngrp + 255) / 256), dim3(256), 0, s, cand_cnt, nblk, grp_scratch);
  }
  launch_k(k_compact, dim3(nblk), dim3(kArcBlock), 0, s, cand_xy, cand_idx, cand_cnt, nblk,
                     comp_xy, comp_idx, total, (const uint32_t*)(grouped ? grp_scratch : nullptr));
}

// OR the bits of columns [xa,xb] that fall into word (w0+k) of row yy; rows/words outside the
// image contribute 0 to a clamped (valid) address, so there is no branch
// GBM: the bitmap lives in global memory (sensors whose bitmap does not fit LDS: the image front-end
// at the frame cameras' sizes) — its words are then read and OR-ed with device-scope atomics, which
// are performed at L2, so that a read never sees a stale line of the CU's vector cache.
template <bool GBM>
__device__ __forceinline__ void bm_or(uint32_t* p, uint32_t bits) {
  if (GBM)
    __hip_atomic_fetch_or(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_fetch_or(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <bool GBM>
__device__ __forceinline__ uint32_t bm_read(const uint32_t* p) {
  if (GBM) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool GBM>
__device__ __forceinline__ void bm_fence() {  // earlier ORs of this wave are visible to its later reads
  if (GBM) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  // (LDS: the unit takes a wave's instructions in order, nothing to do)
}
template <bool GBM>
__device__ __forceinline__ void stamp_words(uint32_t* lds, int wpr, int W, int H, int ax, int yy,
                                            int hw, int k0, int k1) {
  const bool valid = hw >= 0 && (unsigned)yy < (unsigned)H;
  const int xa = valid ? max(ax - hw, 0) : 1;
  const int xb = valid ? min(ax + hw, W - 1) : 0;
  const int w0 = xa >> 5;
  const int rowbase = min(max(yy, 0), H - 1) * wpr;
#pragma unroll
  for (int k = k0; k < k1; k++) {
    const int w = w0 + k;
    const int lo = max(xa - (w << 5), 0), hi = min(xb - (w << 5), 31);
    const uint32_t bits = hi >= lo ? (((2u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
    bm_or<GBM>(lds + rowbase + min(w, wpr - 1), bits);
  }
}
template <bool GBM>
__device__ __forceinline__ void stamp_row(uint32_t* lds, int wpr, int W, int H, int ax, int yy,
                                          int hw) {
  stamp_words<GBM>(lds, wpr, W, H, ax, yy, hw, 0, 3);  // 2*31+1 = 63 px -> at most 3 words
}
// radius <= 15: the row's span is at most 31 px, i.e. one 64-bit shifted mask = two words (the
// second OR is 0 when the span stays inside one word; the bitmap is followed by spare words)
template <bool GBM>
__device__ __forceinline__ void stamp_row_small(uint32_t* lds, int wpr, int W, int H, int ax, int yy,
                                                int hw) {
  const bool valid = hw >= 0 && (unsigned)yy < (unsigned)H;
  const int xa = max(ax - hw, 0), xb = min(ax + hw, W - 1);
  const int len = valid ? xb - xa + 1 : 0;  // <= 31
  const unsigned long long mm = (unsigned long long)((1u << len) - 1u) << (xa & 31);
  uint32_t* wp = lds + (min(max(yy, 0), H - 1) * wpr + (xa >> 5));
  bm_or<GBM>(wp, (uint32_t)mm);
  bm_or<GBM>(wp + 1, (uint32_t)(mm >> 32));
}
template <bool GBM>
__device__ __forceinline__ void stamp_row_tail(uint32_t* lds, int wpr, int W, int H, int ax, int yy,
                                               int hw) {
  stamp_words<GBM>(lds, wpr, W, H, ax, yy, hw, 3, 5);  // r <= 63: 127 px -> at most 5 words
}

// sub-chunks with more live candidates than this are resolved as a batch (see k_select)
constexpr int kSelectBatchMin = 4;
constexpr int kSelectGroup = 8;
constexpr int kSelectLaneStampMin = 12;  // accepted discs per sub-chunk above which each lane stamps its own

template <bool GBM>
__device__ __forceinline__ void select_body(const SelectArgs& a, uint32_t* lds_base) {
  // H * wpr words (+ spare), bit set = inside a disc stamped by this call
  uint32_t* const lds = GBM ? a.gbitmap : lds_base;  // (named for its usual home)
  uint32_t* const bitmap = lds;
  const int lane = lane_id();
  const int nwords = a.H * a.wpr;
  int* const hwtab = (int*)(GBM ? lds_base : lds_base + ((nwords + 3) & ~3));  // cv::circle half-widths by |dy|, -1 beyond r
  // the bitmap starts empty, or from the caller's blocked-pixel bitmap (Event_setMask): a candidate
  // on a blocked pixel is then skipped exactly like one inside an already stamped disc
  if (a.init_bits) {
    constexpr int kInFlight = 8;  // 8 x 16 B loads per lane in flight: the copy is latency-bound
    for (int i0 = lane * 4; i0 < nwords; i0 += 256 * kInFlight) {
      uint4 v[kInFlight];
#pragma unroll
      for (int k = 0; k < kInFlight; k++) {
        const int i = i0 + 256 * k;
        v[k] = i + 3 < nwords ? *(const uint4*)(a.init_bits + i) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < kInFlight; k++) {
        const int i = i0 + 256 * k;
        if (i + 3 < nwords) *(uint4*)(lds + i) = v[k];
      }
    }
    for (int k = (nwords & ~3) + lane; k < nwords; k += 64) lds[k] = a.init_bits[k];
  } else {
    for (int i = lane * 4; i < nwords; i += 256) {
      if (i + 3 < nwords) {
        *(uint4*)(lds + i) = make_uint4(0, 0, 0, 0);
      } else {
        for (int k = i; k < nwords; k++) lds[k] = 0;
      }
    }
    if (GBM)  // (the spare words behind the bitmap the two-word stamps may touch)
      for (int k = nwords + lane; k < nwords + 4; k += 64) lds[k] = 0;
  }
  const int r = a.radius;
  hwtab[lane] = lane <= r ? (int)a.hw[lane] : -1;
  uint32_t* const spts = (uint32_t*)(hwtab + 64);
  if (a.stamp_pts)  // (one round trip to the host-visible point list, all lanes at once)
    for (int k = lane; k < a.n_stamp; k += 64) {
      const float2 p = a.stamp_pts[k];
      spts[k] = (uint32_t)__float2int_rn(p.x) | ((uint32_t)__float2int_rn(p.y) << 16);  // cvRound
    }
  bm_fence<GBM>();
  if (a.stamp_pts) {
    const int rows = 2 * r + 1;
    for (int t = lane; t < a.n_stamp * rows; t += 64) {
      const int p = t / rows, row = t - p * rows;
      const uint32_t v = spts[p];
      const int x = (int)(v & 0xffffu), y = (int)(v >> 16);
      const int hwr = hwtab[row < r ? r - row : row - r];
      if (r <= 15) {
        stamp_row_small<GBM>(lds, a.wpr, a.W, a.H, x, y - r + row, hwr);
      } else {
        stamp_row<GBM>(lds, a.wpr, a.W, a.H, x, y - r + row, hwr);
        if (r > 31) stamp_row_tail<GBM>(lds, a.wpr, a.W, a.H, x, y - r + row, hwr);
      }
    }
    bm_fence<GBM>();
  }
  const uint32_t total = *a.total;
  int accepted = 0;
  // this lane's disc rows (row index lane and lane+64) and their half-widths
  const int row_a = lane, row_b = lane + 64;
  const int hw_a = row_a < 2 * r + 1 ? a.hw[row_a < r ? r - row_a : row_a - r] : -1;
  const int hw_b = row_b < 2 * r + 1 ? a.hw[row_b < r ? r - row_b : row_b - r] : -1;
  // 256 candidates per step (4 sub-chunks of 64 in stream order); the next step's loads are issued
  // before the current one is processed so the serial loop never waits on HBM/L2 latency
  constexpr int SUB = 4;
  uint32_t nxy[SUB], nci[SUB];
  // loads are unconditional (clamped index) so the compiler never has to wait for them right away
  const uint32_t last = total ? total - 1 : 0;
#pragma unroll
  for (int j = 0; j < SUB; j++) {
    const uint32_t i = min(j * 64 + lane, last);
    nxy[j] = a.comp_xy[i];
    nci[j] = a.comp_idx[i];
  }
  for (uint32_t base = 0; base < total && accepted < a.max_corners; base += 64 * SUB) {
    uint32_t cxy[SUB], cci[SUB];
#pragma unroll
    for (int j = 0; j < SUB; j++) {
      cxy[j] = nxy[j];
      cci[j] = nci[j];
      const uint32_t i = min(base + 64 * SUB + j * 64 + lane, last);
      nxy[j] = a.comp_xy[i];
      nci[j] = a.comp_idx[i];
    }
    {  // fast path: all 256 candidates of this step already blocked -> next step
      bm_fence<GBM>();
      bool any = false;
#pragma unroll
      for (int j = 0; j < SUB; j++) {
        const int x = cxy[j] & 0xffff, y = cxy[j] >> 16;
        any = any || ((base + j * 64 + lane < total) &&
                      !((bm_read<GBM>(bitmap + (y * a.wpr + (x >> 5))) >> (x & 31)) & 1u));
      }
      if (!__ballot(any)) continue;
    }
#pragma unroll
    for (int j = 0; j < SUB; j++) {
      const uint32_t i = base + j * 64 + lane;
      const bool have = i < total;
      const int x = cxy[j] & 0xffff, y = cxy[j] >> 16;
      // discs stamped while earlier sub-chunks were processed are visible through the bitmap
      bm_fence<GBM>();
      bool alive = have && !((bm_read<GBM>(bitmap + (y * a.wpr + (x >> 5))) >> (x & 31)) & 1u);
      const unsigned long long m_alive = __ballot(alive);
      if (__builtin_popcountll(m_alive) > kSelectBatchMin) {
        // Many live candidates in this sub-chunk (the usual case while the image is still empty):
        // resolve the whole sub-chunk at once.  B_j = lanes whose pixel lies inside candidate j's
        // disc (one readlane + a handful of VALU ops + a compare per j, independent across j);
        // the greedy itself then runs on the scalar unit over 64-bit lane masks, in stream order.
        unsigned long long dead = ~m_alive, acc = 0;
        unsigned long long rem = m_alive;  // live candidates not yet looked at, in stream order
        const int room = a.max_corners - accepted;
        while (rem && __builtin_popcountll(acc) < room) {
          // a group of live candidates, staged so that the table look-ups are in flight together
          // (candidates the previous groups' discs have already killed are not looked at again)
          rem &= ~dead;
          unsigned long long bit[kSelectGroup], Bdy[kSelectGroup];
          int dxs[kSelectGroup], hws[kSelectGroup];
          if (a.disc_c >= 0) {  // (wave-uniform) disc membership without the table
#pragma unroll
            for (int g = 0; g < kSelectGroup; g++) {
              const int q = rem ? __builtin_ctzll(rem) : 0;
              bit[g] = rem & (0ull - rem);
              rem &= rem - 1ull;
              const int ddx = x - __builtin_amdgcn_readlane(x, q), ddy = y - __builtin_amdgcn_readlane(y, q);
              Bdy[g] = bal(ddx * ddx + ddy * ddy <= a.disc_c);
            }
#pragma unroll
            for (int g = 0; g < kSelectGroup; g++) {
              const unsigned long long t = bit[g] & ~dead;  // candidate still free?
              acc |= t;
              dead |= t ? Bdy[g] : 0ull;
            }
            continue;
          }
#pragma unroll
          for (int g = 0; g < kSelectGroup; g++) {
            const int q = rem ? __builtin_ctzll(rem) : 0;
            bit[g] = rem & (0ull - rem);  // lowest live candidate (0 when none is left)
            rem &= rem - 1ull;            // (0 stays 0)
            const int ax = __builtin_amdgcn_readlane(x, q), ay = __builtin_amdgcn_readlane(y, q);
            const int dy = (int)__builtin_amdgcn_sad_u16((unsigned)y, (unsigned)ay, 0u);
            dxs[g] = (int)__builtin_amdgcn_sad_u16((unsigned)x, (unsigned)ax, 0u);
            Bdy[g] = bal(dy <= r);
            hws[g] = hwtab[min(dy, 63)];
          }
#pragma unroll
          for (int g = 0; g < kSelectGroup; g++) {
            const unsigned long long B = Bdy[g] & bal(dxs[g] <= hws[g]);  // lanes inside this disc
            const unsigned long long t = bit[g] & ~dead;                  // candidate still free?
            acc |= t;
            dead |= t ? B : 0ull;
          }
        }
        // only the first (max_corners - accepted) of them count
        const int before = __builtin_popcountll(acc & ((1ull << lane) - 1ull));
        const bool mine = ((acc >> lane) & 1ull) && before < room;
        if (mine) {
          a.out_pts[a.out_base + accepted + before] = make_float2((float)x, (float)y);
          if (a.pub_slots)
            __hip_atomic_store(&a.pub_slots[a.out_base + accepted + before],
                               ((unsigned long long)a.pub_seq << 32) | ((unsigned)y << 16) | (unsigned)x,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (a.out_idx) a.out_idx[accepted + before] = (int)cci[j];
        }
        accepted += min(__builtin_popcountll(acc), room);
        if (accepted >= a.max_corners) break;
        // stamp the accepted discs for the later sub-chunks
        if (__builtin_popcountll(acc) > kSelectLaneStampMin) {
          // many: one lane per disc, rows in a uniform loop
          for (int row = 0; row <= 2 * r; row++) {
            const int hwr = mine ? (int)a.hw[row < r ? r - row : row - r] : -1;
            if (r <= 15) {
              stamp_row_small<GBM>(lds, a.wpr, a.W, a.H, x, y - r + row, hwr);
            } else {
              stamp_row<GBM>(lds, a.wpr, a.W, a.H, x, y - r + row, hwr);
              if (r > 31) stamp_row_tail<GBM>(lds, a.wpr, a.W, a.H, x, y - r + row, hwr);
            }
          }
        } else {
          // few: one disc at a time, one lane per row
          for (unsigned long long todo = acc; todo; todo &= todo - 1ull) {
            const int q = __builtin_ctzll(todo);
            const int ax = __builtin_amdgcn_readlane(x, q), ay = __builtin_amdgcn_readlane(y, q);
            if (r <= 15) {
              stamp_row_small<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_a, hw_a);
              continue;
            }
            stamp_row<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_a, hw_a);
            if (r > 31) {
              stamp_row<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_b, hw_b);
              stamp_row_tail<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_a, hw_a);
              stamp_row_tail<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_b, hw_b);
            }
          }
        }
        continue;
      }
      unsigned long long m = m_alive;  // (live candidates of this sub-chunk, in stream order)
      while (m && accepted < a.max_corners) {
        const int first = __ffsll((long long)m) - 1;
        const int ax = __builtin_amdgcn_readlane(x, first);
        const int ay = __builtin_amdgcn_readlane(y, first);
        const int ai = __builtin_amdgcn_readlane((int)cci[j], first);
        if (lane == 0) {
          a.out_pts[a.out_base + accepted] = make_float2((float)ax, (float)ay);
          if (a.pub_slots)
            __hip_atomic_store(&a.pub_slots[a.out_base + accepted],
                               ((unsigned long long)a.pub_seq << 32) | ((unsigned)ay << 16) | (unsigned)ax,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (a.out_idx) a.out_idx[accepted] = ai;
        }
        accepted++;
        // stamp the disc for the LATER sub-chunks (LDS atomic OR, not waited for here): one lane
        // per row, branch-free: the row's span [xa,xb] touches at most 3 words for r <= 31
        if (r <= 15) {
          stamp_row_small<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_a, hw_a);
        } else {
          stamp_row<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_a, hw_a);
        }
        if (r > 31) {  // wave-uniform: rows 64.. of a large disc
          stamp_row<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_b, hw_b);
          stamp_row_tail<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_a, hw_a);
          stamp_row_tail<GBM>(lds, a.wpr, a.W, a.H, ax, ay - r + row_b, hw_b);
        }
        m &= m - 1ull;
        if (!m) break;  // (the usual case once the image has filled up: one live candidate)
        // ... and kill this sub-chunk's remaining candidates geometrically: inside the disc iff
        // |dx| <= hw[|dy|] (the same table the stamp uses), so no bitmap round trip per accept
        const int dy = y > ay ? y - ay : ay - y, dx = x > ax ? x - ax : ax - x;
        const bool inside = a.disc_c >= 0 ? dx * dx + dy * dy <= a.disc_c
                                          : (dy <= r && dx <= hwtab[min(dy, 63)]);
        alive = alive && lane > first && !inside;
        m = __ballot(alive);
      }
      if (accepted >= a.max_corners) break;
    }
  }
  if (lane == 0) {
    *a.n_out = accepted;
    if (a.n_total) *a.n_total = a.out_base + accepted;
    if (a.pub_done)
      __hip_atomic_store(a.pub_done, ((unsigned long long)a.pub_seq << 32) | (unsigned)(a.out_base + accepted),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.host_counts) {
      a.host_counts[0] = accepted;
      a.host_counts[1] = a.out_base + accepted;
      a.host_counts[2] = (int)total;
    }
  }
}

__global__ __launch_bounds__(64) void k_select(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  select_body<false>(a, lds);
}
// the same selection with the bitmap in global memory (a.gbitmap)
__global__ __launch_bounds__(64) void k_select_gbm(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  select_body<true>(a, lds);
}


// ---- the same greedy with 16 waves (default whenever bitmap + queue fit LDS).  What the sequential
// loop computes is the lexicographically first maximal independent set of the candidates under "lies
// inside the other's disc" (symmetric for cv::circle's table and for the Euclidean disc), seeded with
// the blocked pixels and cut off at max_corners: a candidate's fate depends on EARLIER candidates only.
// One wave walking the stream pays a serial LDS round trip per accepted corner (~0.19 us: k_select was
// 53-65 us for ~215 corners, on the frame's device chain).  Here
//   filter  (all waves): the not-yet-decided candidates — the queue's leftovers, then the stream —
//           are tested against the bitmap, the live ones compacted IN STREAM ORDER into the queue;
//           repeated until the queue holds 64 live candidates (or the stream ends);
//   resolve (wave 0): the queue's first <= 64 entries are all live (nothing was stamped since their
//           test), so their fate depends only on each other: the greedy among them as 64-bit mask
//           arithmetic (B_j = lanes inside candidate j's disc), accepted ones written out / published /
//           stamped; what is left in the queue is re-tested by the next filter.
// The accepted sequence is the sequential loop's: every candidate is decided after all earlier ones,
// against a bitmap holding exactly the discs accepted before it.  ~6 filter + resolve rounds for 215
// corners instead of ~200 serial sub-chunk steps; the bitmap clear and the kept points' discs are
// spread over 1024 threads (they were ~10 us of the one-wave kernel).
constexpr int kSelMwThreads = 1024;
constexpr int kSelMwQueue = kSelMwThreads + 64;
size_t select_mw_extra_lds_bytes() { return (size_t)(2 * kSelMwQueue + 32 + 16 + 128 + 64) * 4; }

__global__ __launch_bounds__(kSelMwThreads) void k_select_mw(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwords = a.H * a.wpr;
  uint32_t* const bitmap = lds;
  int* const hwtab = (int*)(lds + ((nwords + 3) & ~3));
  uint32_t* const spts = (uint32_t*)(hwtab + 64);
  uint32_t* const q_xy = spts + ((a.n_stamp + 3) & ~3);
  uint32_t* const q_ci = q_xy + kSelMwQueue;
  uint32_t* const wcnt = q_ci + kSelMwQueue;  // [16] live items per wave of the current filter pass
  uint32_t* const ctrl = wcnt + 16;           // [0] corners accepted so far, [1] by the latest resolve
  uint32_t* const adj = ctrl + 16;            // [64 x 2] entry j's neighbours among the 64 being resolved
  uint32_t* const acc_xy = adj + 128;         // [64] the latest resolve's accepted pixels
  const int r = a.radius;
  // ---- bitmap: empty or the caller's blocked pixels; the kept points' discs
  if (a.init_bits) {
    for (int i = tid * 4; i < nwords; i += kSelMwThreads * 4) {
      if (i + 3 < nwords) {
        *(uint4*)(bitmap + i) = *(const uint4*)(a.init_bits + i);
      } else {
        for (int k = i; k < nwords; k++) bitmap[k] = a.init_bits[k];
      }
    }
  } else {
    for (int i = tid * 4; i < nwords; i += kSelMwThreads * 4) {
      if (i + 3 < nwords) {
        *(uint4*)(bitmap + i) = make_uint4(0, 0, 0, 0);
      } else {
        for (int k = i; k < nwords; k++) bitmap[k] = 0;
      }
    }
  }
  if (tid < 64) hwtab[tid] = tid <= r ? (int)a.hw[tid] : -1;
  if (tid == 0) ctrl[0] = 0;
  if (a.stamp_pts)
    for (int k = tid; k < a.n_stamp; k += kSelMwThreads) {
      const float2 p = a.stamp_pts[k];
      spts[k] = (uint32_t)__float2int_rn(p.x) | ((uint32_t)__float2int_rn(p.y) << 16);  // cvRound
    }
  __syncthreads();
  if (a.stamp_pts) {
    const int rows = 2 * r + 1;
    for (int t = tid; t < a.n_stamp * rows; t += kSelMwThreads) {
      const int p = t / rows, row = t - p * rows;
      const uint32_t v = spts[p];
      const int x = (int)(v & 0xffffu), y = (int)(v >> 16);
      const int hwr = hwtab[row < r ? r - row : row - r];
      if (r <= 15) {
        stamp_row_small<false>(bitmap, a.wpr, a.W, a.H, x, y - r + row, hwr);
      } else {
        stamp_row<false>(bitmap, a.wpr, a.W, a.H, x, y - r + row, hwr);
        if (r > 31) stamp_row_tail<false>(bitmap, a.wpr, a.W, a.H, x, y - r + row, hwr);
      }
    }
  }
  const uint32_t total = *a.total;
  __syncthreads();
  // (block-uniform bookkeeping, every thread keeps its own copy)
  uint32_t pos = 0;      // next candidate of the stream nobody has looked at
  int qoff = 0, qn = 0;  // the queue's undecided entries: q_*[qoff .. qoff + qn)
  int accepted = 0;
  const uint32_t last = total ? total - 1 : 0;
  while (accepted < a.max_corners && (qn > 0 || pos < total)) {
    // ---- filter: thread t takes leftover t, or stream candidate pos + (t - qn)
    for (;;) {
      const bool from_q = tid < qn;
      const uint32_t si = pos + (uint32_t)(tid - qn);
      const bool have = from_q || si < total;
      uint32_t xy, ci;
      if (from_q) {
        xy = q_xy[qoff + tid];
        ci = q_ci[qoff + tid];
      } else {
        xy = a.comp_xy[min(si, last)];
        ci = a.comp_idx[min(si, last)];
      }
      const int x = xy & 0xffff, y = xy >> 16;
      const bool alive = have && !((bitmap[y * a.wpr + (x >> 5)] >> (x & 31)) & 1u);
      const lanemask_t m = bal(alive);
      if (lane == 0) wcnt[wave] = (uint32_t)__builtin_popcountll(m);
      lds_barrier();  // (every leftover has been read: the queue may be rewritten from its start)
      uint32_t before = 0, all = 0;
#pragma unroll
      for (int w = 0; w < kSelMwThreads / 64; w++) {
        const uint32_t cw = wcnt[w];
        before += w < wave ? cw : 0u;
        all += cw;
      }
      if (alive) {
        const uint32_t o = before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        q_xy[o] = xy;
        q_ci[o] = ci;
      }
      const uint32_t taken = (uint32_t)(kSelMwThreads - qn);  // stream candidates this pass has looked at
      pos = pos + taken < total ? pos + taken : total;
      qoff = 0;
      qn = (int)all;
      lds_barrier();
      if (qn >= 64 || pos >= total) break;
    }
    if (!qn) break;
    // ---- resolve: the first n entries, all live, against each other
    const int n = qn < 64 ? qn : 64;
    {
      // (a) who lies in whose disc: wave w tests every entry against the discs of entries 4w .. 4w+3
      const bool have = lane < n;
      const uint32_t xy = q_xy[have ? lane : 0], ci = q_ci[have ? lane : 0];
      const int x = xy & 0xffff, y = xy >> 16;
#pragma unroll
      for (int k = 0; k < 64 / (kSelMwThreads / 64); k++) {
        const int j = wave * (64 / (kSelMwThreads / 64)) + k;
        const int ax = __builtin_amdgcn_readlane(x, j), ay = __builtin_amdgcn_readlane(y, j);
        const int dy = (int)__builtin_amdgcn_sad_u16((unsigned)y, (unsigned)ay, 0u);
        const int dx = (int)__builtin_amdgcn_sad_u16((unsigned)x, (unsigned)ax, 0u);
        const bool inside = a.disc_c >= 0 ? dx * dx + dy * dy <= a.disc_c : (dy <= r && dx <= hwtab[min(dy, 63)]);
        const lanemask_t B = bal(have && inside);
        if (lane == 0) {
          adj[2 * j] = (uint32_t)B;
          adj[2 * j + 1] = (uint32_t)(B >> 32);
        }
      }
      lds_barrier();
      if (wave == 0) {
        // (b) the greedy among them, as rounds: an undecided entry with an accepted earlier neighbour is
        // refused, one whose earlier neighbours are all decided (none accepted) is accepted — the
        // lowest undecided entry is decided in every round, a typical batch takes 3-6 rounds
        const lanemask_t Bj = (lanemask_t)adj[2 * lane] | ((lanemask_t)adj[2 * lane + 1] << 32);
        const lanemask_t earlier = Bj & ((1ull << lane) - 1ull);
        lanemask_t U = bal(have), A = 0;
        while (U) {
          const bool und = (U >> lane) & 1ull;
          const bool rej = und && (earlier & A) != 0;
          const bool acc1 = und && !rej && (earlier & U) == 0;
          const lanemask_t ma = bal(acc1), mr = bal(rej);
          A |= ma;
          U &= ~(ma | mr);
        }
        // only the first (max_corners - accepted) of them count
        const int room = a.max_corners - accepted;
        const int before = __builtin_popcountll(A & ((1ull << lane) - 1ull));
        const bool mine = ((A >> lane) & 1ull) && before < room;
        if (mine) {
          a.out_pts[a.out_base + accepted + before] = make_float2((float)x, (float)y);
          if (a.pub_slots)
            __hip_atomic_store(&a.pub_slots[a.out_base + accepted + before],
                               ((unsigned long long)a.pub_seq << 32) | ((unsigned)y << 16) | (unsigned)x,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (a.out_idx) a.out_idx[accepted + before] = (int)ci;
          acc_xy[before] = xy;
        }
        if (lane == 0) {
          const int cnt = min(__builtin_popcountll(A), room);
          ctrl[0] = (uint32_t)(accepted + cnt);
          ctrl[1] = (uint32_t)cnt;
        }
      }
      lds_barrier();
      const int cnt = (int)ctrl[1];
      accepted = (int)ctrl[0];
      if (accepted < a.max_corners) {
        // (c) their discs, for everything that is still undecided: one thread per (disc, row)
        const int rows = 2 * r + 1;
        for (int t = tid; t < cnt * rows; t += kSelMwThreads) {
          const int d = t / rows, row = t - d * rows;
          const uint32_t v = acc_xy[d];
          const int hwr = hwtab[row < r ? r - row : row - r];
          if (r <= 15) {
            stamp_row_small<false>(bitmap, a.wpr, a.W, a.H, (int)(v & 0xffffu), (int)(v >> 16) - r + row, hwr);
          } else {
            stamp_row<false>(bitmap, a.wpr, a.W, a.H, (int)(v & 0xffffu), (int)(v >> 16) - r + row, hwr);
            if (r > 31) stamp_row_tail<false>(bitmap, a.wpr, a.W, a.H, (int)(v & 0xffffu), (int)(v >> 16) - r + row, hwr);
          }
        }
        lds_barrier();
      }
    }
    qoff = n;
    qn -= n;
  }
  if (tid == 0) {
    *a.n_out = accepted;
    if (a.n_total) *a.n_total = a.out_base + accepted;
    if (a.pub_done)
      __hip_atomic_store(a.pub_done, ((unsigned long long)a.pub_seq << 32) | (unsigned)(a.out_base + accepted),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.host_counts) {
      a.host_counts[0] = accepted;
      a.host_counts[1] = a.out_base + accepted;
      a.host_counts[2] = (int)total;
    }
  }
}

KernelId launch_select(hipStream_t s, const SelectArgs& a, size_t lds_bytes) {
  if (a.gbitmap) {
    launch_k(k_select_gbm, dim3(1), dim3(64), lds_bytes, s, a);
    return K_SELECT_GBM;
  }
  if (!a.one_wave && lds_bytes + select_mw_extra_lds_bytes() <= 160 * 1024) {
    launch_k(k_select_mw, dim3(1), dim3(kSelMwThreads), lds_bytes + select_mw_extra_lds_bytes(), s, a);
    return K_SELECT_MW;
  }
  launch_k(k_select, dim3(1), dim3(64), lds_bytes, s, a);
  return K_SELECT;
}

}  // namespace esvio
