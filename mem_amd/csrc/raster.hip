// Event stream -> [pos, tss, neg] uint8 voxel image (reference: mem/datasets.py:566-595).
//
// HBM-bound scatter: every event (32 B, the reference's (N,4) float64 row) is read exactly
// once with two 16-B loads per lane (a wave covers 2 KiB contiguous), binned with u32 atomics
// that resolve in the XCD L2 / memory side, and a second streaming pass folds the u32 bins to
// the reference's wrap-around uint8 counts.  Algorithmic bytes: 32*N + 3*H*W per sample.
#include "common.h"

static_assert(sizeof(memhip_event_aug_t) == 56, "memhip_event_aug_t ABI layout");


namespace {

constexpr int kThreads = 256;
constexpr int kEvPerThread = 8;

// order-preserving map double -> u64 (so integer atomic max == fp max)
__device__ __forceinline__ unsigned long long enc_f64(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(unsigned long long e) {
  unsigned long long b = (e & 0x8000000000000000ull) ? (e & 0x7fffffffffffffffull) : ~e;
  return __longlong_as_double((long long)b);
}

// workspace layout: bins u32 [B][3][HW] (0: pos count, 1: last event position + 1, 2: neg
// count), then tstat u64 [B][2] (max enc(t), max ~enc(t)).

struct Ev { double x, y, t, p; long long pos; bool keep; };

// One event through the reference's augmentation chain (datasets.py:464-549, 598-609), in
// float64 and in the reference's order.  `i` is the index inside the sample's window of n events.
// (the raw 32-byte row and the arithmetic are separate: the kernels issue the loads of ALL the events of a thread before
// the first event is looked at -- inside the per-event branches (out of range, filtered, bad index) hipcc waits for
// every load right where it is issued, i.e. one event's 32 bytes in flight per thread)
struct RawEv { double2 xy, tp; };
__device__ __forceinline__ RawEv load_raw(const double* __restrict__ ev, long long row) {
  return RawEv{reinterpret_cast<const double2*>(ev)[2 * row], reinterpret_cast<const double2*>(ev)[2 * row + 1]};
}
// The sample's augmentation as wave-uniform float64 constants (neutral when there is none: x * 1.0 and x + 0.0 are exact,
// the sign of a zero does not survive the truncation to a pixel index): the per-event arithmetic is branch-free and
// converts nothing -- the reference's order of operations is unchanged.
struct AugK { double sx, sy, fw1, shx, shy, fw, fh; bool tflip, flipx, filt; };
__device__ __forceinline__ AugK aug_k(const memhip_event_aug_t* __restrict__ a) {
  AugK k{1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, false, false, false};
  if (a) {
    k.sx = a->scale_x; k.sy = a->scale_y;
    k.tflip = a->time_flip != 0; k.flipx = a->flip_x != 0; k.filt = a->do_filter != 0;
    k.fw1 = (double)(a->flip_w - 1);
    k.shx = (double)a->shift_x; k.shy = (double)a->shift_y;
    k.fw = (double)a->filt_w; k.fh = (double)a->filt_h;
  }
  return k;
}
__device__ __forceinline__ Ev make_event_k(const RawEv& r, long long n, long long i, const AugK& k, double t_last) {
  Ev e{r.xy.x * k.sx, r.xy.y * k.sy, r.tp.x, r.tp.y, i, true};
  if (k.tflip) { e.pos = n - 1 - i; e.t = t_last - e.t; e.p = -e.p; }
  if (k.flipx) e.x = k.fw1 - e.x;
  e.x += k.shx;
  e.y += k.shy;
  if (k.filt) e.keep = (e.x >= 0.0) & (e.x < k.fw) & (e.y >= 0.0) & (e.y < k.fh);
  return e;
}
// unsigned division by a wave-uniform divisor d through its (multiplier, shift) pair (Granlund-Montgomery, n < 2^32):
//   l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1, n / d = (t + ((n - t) >> 1)) >> (l - 1) with t = mulhi(m, n)
struct UDiv { unsigned m; int l; };
__host__ __device__ inline UDiv udiv_prepare(unsigned d) {
  int l = 0;
  while ((1ull << l) < d) ++l;
  const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
  return UDiv{(unsigned)m, l};
}
__device__ __forceinline__ unsigned udiv_apply(unsigned n, const UDiv& u) {
  if (u.l == 0) return n;                                    // d == 1
  const unsigned t = __umulhi(u.m, n);
  return (t + ((n - t) >> 1)) >> (u.l - 1);
}

__device__ __forceinline__ Ev make_event(const RawEv& r, long long n, long long i, const memhip_event_aug_t* __restrict__ a,
                                         double t_last) {
  Ev e{r.xy.x, r.xy.y, r.tp.x, r.tp.y, i, true};
  if (a) {
    e.x *= a->scale_x;
    e.y *= a->scale_y;
    if (a->time_flip) { e.pos = n - 1 - i; e.t = t_last - e.t; e.p = -e.p; }
    if (a->flip_x) e.x = (double)(a->flip_w - 1) - e.x;
    e.x += (double)a->shift_x;
    e.y += (double)a->shift_y;
    if (a->do_filter)
      e.keep = (e.x >= 0.0) & (e.x < (double)a->filt_w) & (e.y >= 0.0) & (e.y < (double)a->filt_h);
  }
  return e;
}
__device__ __forceinline__ Ev load_event(const double* __restrict__ ev, long long beg, long long n,
                                         long long i, const memhip_event_aug_t* __restrict__ a,
                                         double t_last) {
  return make_event(load_raw(ev, beg + i), n, i, a, t_last);
}

__global__ __launch_bounds__(kThreads) void raster_count(
    const double* __restrict__ ev, const int64_t* __restrict__ offsets,
    const memhip_event_aug_t* __restrict__ augs, int H, int W,
    int time_surface, unsigned int* __restrict__ bins, unsigned long long* __restrict__ tstat,
    int32_t* __restrict__ status, const int32_t* __restrict__ dims, long long slot_px) {
  const int b = blockIdx.y;
  const long long beg = offsets[b], end = offsets[b + 1];
  if (dims) { H = dims[2 * b]; W = dims[2 * b + 1]; }       // per-sample canvas (memhip_rasterize_var_f64)
  const long long HW = (long long)H * W;
  if (HW <= 0) return;
  unsigned int* pos = bins + (size_t)b * 3 * (dims ? slot_px : HW);
  unsigned int* tss = pos + HW;
  unsigned int* neg = tss + HW;
  unsigned long long tmax = 0ull, tmin_inv = 0ull;
  int bad = 0;
  const long long n = end - beg;
  const memhip_event_aug_t* a = augs ? augs + b : nullptr;
  const double t_last = (a && a->time_flip && n > 0) ? ev[4 * (end - 1) + 2] : 0.0;
  const long long chunk = (long long)kThreads * kEvPerThread;
  for (long long base = (long long)blockIdx.x * chunk; base < n;
       base += (long long)gridDim.x * chunk) {
    RawEv raw[kEvPerThread];
#pragma unroll
    for (int k = 0; k < kEvPerThread; ++k) {
      const long long i = base + (long long)k * kThreads + threadIdx.x;
      raw[k] = load_raw(ev, beg + (i < n ? i : n - 1));          // (n > 0 inside this loop)
    }
#pragma unroll
    for (int k = 0; k < kEvPerThread; ++k) {
      const long long i = base + (long long)k * kThreads + threadIdx.x;
      if (i >= n) break;
      const Ev e = make_event(raw[k], n, i, a, t_last);
      if (!e.keep) continue;
      const long long xi = (long long)e.x;   // trunc toward zero == ndarray.astype(int)
      const long long yi = (long long)e.y;
      long long flat = xi + (long long)W * yi;
      if (flat < -HW || flat >= HW) { ++bad; continue; }   // reference: IndexError
      if (flat < 0) flat += HW;                             // NumPy negative index
      if (e.p == 1.0) atomicAdd(pos + flat, 1u);
      else if (e.p == -1.0) atomicAdd(neg + flat, 1u);
      if (time_surface) {
        atomicMax(tss + flat, (unsigned int)e.pos + 1u);    // last in (augmented) array order
        const unsigned long long q = enc_f64(e.t);
        tmax = q > tmax ? q : tmax;
        tmin_inv = (~q) > tmin_inv ? (~q) : tmin_inv;
      }
    }
  }
  if (time_surface) {
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long a = __shfl_xor(tmax, o), c = __shfl_xor(tmin_inv, o);
      tmax = a > tmax ? a : tmax;
      tmin_inv = c > tmin_inv ? c : tmin_inv;
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMax(tstat + 2 * b, tmax);
      atomicMax(tstat + 2 * b + 1, tmin_inv);
    }
  }
  if (bad) atomicAdd(status + b, bad);
}

__global__ __launch_bounds__(kThreads) void raster_finalize(
    const double* __restrict__ ev, const int64_t* __restrict__ offsets,
    const memhip_event_aug_t* __restrict__ augs, int H, int W,
    int time_surface, const unsigned int* __restrict__ bins,
    const unsigned long long* __restrict__ tstat, uint8_t* __restrict__ out,
    const int32_t* __restrict__ dims, long long slot_px) {
  const int b = blockIdx.y;
  if (dims) { H = dims[2 * b]; W = dims[2 * b + 1]; }
  const long long HW = (long long)H * W;
  const unsigned int* pos = bins + (size_t)b * 3 * (dims ? slot_px : HW);
  const unsigned int* tss = pos + HW;
  const unsigned int* neg = tss + HW;
  uint8_t* o = out + (size_t)b * 3 * (dims ? slot_px : HW);
  double tmin = 0.0, trange = 0.0;
  if (time_surface) {
    const double tmax = dec_f64(tstat[2 * b]);
    tmin = dec_f64(~tstat[2 * b + 1]);
    trange = tmax - tmin;            // == (ts - ts.min()).max()
  }
  const long long beg = offsets[b], n = offsets[b + 1] - beg;
  const bool tflip = augs && augs[b].time_flip;
  const double t_last = (tflip && n > 0) ? ev[4 * (beg + n - 1) + 2] : 0.0;
  for (long long p = (long long)blockIdx.x * kThreads + threadIdx.x; p < HW;
       p += (long long)gridDim.x * kThreads) {
    o[p] = (uint8_t)(pos[p] & 0xFFu);
    o[2 * HW + p] = (uint8_t)(neg[p] & 0xFFu);
    uint8_t tv = 0;
    if (time_surface) {
      const unsigned int last = tss[p];
      if (last) {
        const long long j = (long long)last - 1;           // position in augmented order
        double t = ev[4 * (beg + (tflip ? n - 1 - j : j)) + 2];
        if (tflip) t = t_last - t;
        const double v = (t - tmin) / trange * 255.0;   // same op order as the reference
        tv = (v == v) ? (uint8_t)v : (uint8_t)0;         // NaN (single timestamp) -> 0
      }
    }
    o[HW + p] = tv;
  }
}


// ---- small canvases without time surface: privatised counters.  Scattered u32 atomics resolve at the
// memory side at ~25 G adds/s chip-wide (one 64-byte request per lane), an order of magnitude below
// what the event stream could feed; LDS atomics do not have that limit.  One workgroup per (sample, band
// of rows): its u32 counters for [pos, neg] x band live in LDS, it scans ALL events of the sample (the
// re-reads of the other bands hit L2), and writes the wrapped uint8 counts straight to the output --
// no workspace, no memset, no second pass.  Used when a sample needs <= kMaxBands bands.
constexpr int kLdsThreads = 1024;
constexpr int kBandPixels = 14336;            // 2 polarities x 14336 x 4 B = 112 KiB of LDS
constexpr int kMaxBands = 6;

__global__ __launch_bounds__(kLdsThreads) void raster_lds_kernel(
    const double* __restrict__ ev, const int64_t* __restrict__ offsets,
    const memhip_event_aug_t* __restrict__ augs, int H, int W, int rows_per_band,
    uint8_t* __restrict__ out, int32_t* __restrict__ status) {
  extern __shared__ unsigned int cnt[];       // [2][band_px]
  const int b = blockIdx.y, band = blockIdx.x;
  const long long HW = (long long)H * W;
  const int row0 = band * rows_per_band;
  const int rows = (row0 + rows_per_band <= H) ? rows_per_band : H - row0;
  const int band_px = rows_per_band * W;
  const long long lo = (long long)row0 * W, hi = lo + (long long)rows * W;
  for (int i = threadIdx.x; i < 2 * band_px; i += kLdsThreads) cnt[i] = 0u;
  __syncthreads();
  const long long beg = offsets[b], n = offsets[b + 1] - beg;
  const memhip_event_aug_t* a = augs ? augs + b : nullptr;
  const double t_last = (a && a->time_flip && n > 0) ? ev[4 * (beg + n - 1) + 2] : 0.0;
  int bad = 0;
  constexpr int kLdsBatch = 4;
  for (long long i0 = threadIdx.x; i0 < n; i0 += (long long)kLdsBatch * kLdsThreads) {
    RawEv raw[kLdsBatch];
#pragma unroll
    for (int k = 0; k < kLdsBatch; ++k) {
      const long long i = i0 + (long long)k * kLdsThreads;
      raw[k] = load_raw(ev, beg + (i < n ? i : n - 1));
    }
#pragma unroll
    for (int k = 0; k < kLdsBatch; ++k) {
    const long long i = i0 + (long long)k * kLdsThreads;
    if (i >= n) break;
    const Ev e = make_event(raw[k], n, i, a, t_last);
    if (!e.keep) continue;
    const long long xi = (long long)e.x;   // trunc toward zero == ndarray.astype(int)
    const long long yi = (long long)e.y;
    long long flat = xi + (long long)W * yi;
    if (flat < -HW || flat >= HW) { ++bad; continue; }   // reference: IndexError
    if (flat < 0) flat += HW;                             // NumPy negative index
    if (flat < lo || flat >= hi) continue;
    const int l = (int)(flat - lo);
    if (e.p == 1.0) atomicAdd(cnt + l, 1u);
    else if (e.p == -1.0) atomicAdd(cnt + band_px + l, 1u);
    }
  }
  if (band == 0 && bad) atomicAdd(status + b, bad);
  __syncthreads();
  uint8_t* o = out + (size_t)b * 3 * HW + lo;
  const int npx = rows * W;
  for (int i = threadIdx.x * 4; i < npx; i += kLdsThreads * 4) {
    if (i + 4 <= npx && ((lo + i) & 3) == 0) {
      const unsigned int pv = (cnt[i] & 0xFFu) | ((cnt[i + 1] & 0xFFu) << 8) | ((cnt[i + 2] & 0xFFu) << 16) | (cnt[i + 3] << 24);
      const unsigned int nv = (cnt[band_px + i] & 0xFFu) | ((cnt[band_px + i + 1] & 0xFFu) << 8) |
                              ((cnt[band_px + i + 2] & 0xFFu) << 16) | (cnt[band_px + i + 3] << 24);
      *reinterpret_cast<unsigned int*>(o + i) = pv;
      *reinterpret_cast<unsigned int*>(o + HW + i) = 0u;            // time-surface channel: zeros
      *reinterpret_cast<unsigned int*>(o + 2 * HW + i) = nv;
    } else {
      for (int k = i; k < i + 4 && k < npx; ++k) {
        o[k] = (uint8_t)(cnt[k] & 0xFFu);
        o[HW + k] = 0;
        o[2 * HW + k] = (uint8_t)(cnt[band_px + k] & 0xFFu);
      }
    }
  }
}


// ---- long streams (N-ImageNet scale, ~1 M events per sample) on canvases whose counters do not fit one
// workgroup's LDS: two streaming passes instead of one global atomic per event.
//   pass 1 (raster_bin_keys): every event is read ONCE (32 B), reduced to a 2-byte key (its pixel inside its band) and the keys of
//     each chunk of kBinChunk events are written back sorted by (band, polarity) (counting sort in LDS).  Every segment starts at a
//     multiple of 8 keys (padded with 0xFFFF), so that pass 2 reads 16 bytes per lane; the segment boundaries of the chunk go to a
//     small header.  (Until round 6 the polarity was bit 15 of the key: 15-bit pixel indices, bands of <= 32 764 pixels, TEN bands on
//     480 x 640 -- 320 pass-2 workgroups for 32 samples, i.e. two rounds on 256 CUs; with the polarity in the segment a band is bounded
//     by the 160 KB of LDS: 40 000 pixels, EIGHT bands, 256 workgroups.)
//   pass 2 (raster_bin_accum): one workgroup per (sample, band) walks the headers, reads only its own two segments of every chunk
//     (they are neighbours: one 16-byte load per lane covers both on average; the next chunk's load is in flight while the current
//     keys are counted), counts in LDS and writes the wrapped uint8 planes.
// HBM traffic per event: 32 B read + ~2.3 B written + ~2.3 B read, against 32 B algorithmic.
// Counters: ONE u32 per pixel = [neg count : 16 | pos count : 16].  Only counts mod 256 are observable, so
// 16 bits are enough provided the carry out of the low half is undone: the low half changes only by +1, so
// exactly the add that sees 0xFFFF there carries, and that thread takes the carry back out of the high half.
constexpr int kBinThreads = 512;
constexpr int kBinEvPerThread = 8;
constexpr int kBinChunk = kBinThreads * kBinEvPerThread;   // 4096 events
constexpr int kBinParts = 2;                               // the next chunk is requested in this many parts (4: the same, 8: 5 % slower)
constexpr int kBinMaxBands = 64;
constexpr int kBinBandPixels = 40000;                      // pixels of a band: 16-bit key (0xFFFF is the pad key), 160 000 B of LDS counters in pass 2
constexpr int kBinSegs = 2 * kBinMaxBands;                 // a chunk's keys are sorted by (band, polarity): two segments per band
constexpr int kBinHdr = kBinSegs + 1;                      // segment boundaries of a chunk
constexpr int kBinSlots = kBinChunk + 8 * kBinSegs;        // key slots of a chunk: events + padding of every segment to 8
constexpr int kAccThreads = 1024;
constexpr int kBinKeyGrid = 4096;                           // pass-1 workgroups in all (rounded up to whole samples)
constexpr int kBinOverflow = 1 << 30;                      // status flag: n_events smaller than the offsets say

// The wave's 64 rows of one step as TWO fully contiguous 1-KiB wave-instructions (lane l loads 16 bytes at 16 l: half a row):
// an even lane 2e ends up with row e (its own (x, y) from the first KiB, (t, p) from its odd neighbour), an odd lane 2e + 1
// with row 32 + e (its own (t, p) from the second KiB, (x, y) from its even neighbour) -- one DPP exchange inside the lane pair.
// With every byte of a 128-byte line requested by ONE instruction the rows can be loaded nontemporal: a read-only sweep of
// the 2 GB stream of 64 x 1 M events runs at 6.1 TB/s with default-policy loads of either shape and at 6.85 TB/s with nt loads
// of this shape (tools/exp/r06_raster_read_probe.hip); a row-per-lane nt load fetches every line twice and is SLOWER than the
// default policy (round 3, re-measured in round 6: +35 us on 470).  `row0` = the wave's first row of the step, rows clamped
// to n - 1.
typedef double d2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ RawEv load_raw_pair(const double* __restrict__ ev, long long beg, long long n, long long row0, int lane) {
  const int half = lane & 1, e = lane >> 1;
  long long ra = row0 + e, rb = row0 + 32 + e;
  ra = ra < n ? ra : n - 1; rb = rb < n ? rb : n - 1;
  const d2v_t* pa = reinterpret_cast<const d2v_t*>(ev) + 2 * (beg + ra) + half;
  const d2v_t* pb = reinterpret_cast<const d2v_t*>(ev) + 2 * (beg + rb) + half;
  const d2v_t a = __builtin_nontemporal_load(pa), b = __builtin_nontemporal_load(pb);
  const d2v_t send = half ? a : b;                       // odd: (t, p) of row e; even: (x, y) of row 32 + e
  // (plain doubles first: __builtin_bit_cast of a vector ELEMENT expression reads element 0 with this clang)
  const double sx = send.x, sy = send.y;
  d2v_t recv;
  recv.x = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(sx), 0xB1, 0xF, 0xF, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(sx), 0xB1, 0xF, 0xF, true));
  recv.y = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(sy), 0xB1, 0xF, 0xF, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(sy), 0xB1, 0xF, 0xF, true));
  const d2v_t xy = half ? recv : a, tp = half ? b : recv;
  return RawEv{make_double2(xy.x, xy.y), make_double2(tp.x, tp.y)};
}

// workgroup barrier that orders LDS traffic only: the event rows of the NEXT chunk stay in flight across it (__syncthreads()
// carries a release fence that waits for every outstanding global load and store of the wave)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(kBinThreads) void raster_bin_keys(
    const double* __restrict__ ev, const int64_t* __restrict__ offsets,
    const memhip_event_aug_t* __restrict__ augs, int H, int W, int band_px, int nb, long long n_cap,
    unsigned short* __restrict__ keys, unsigned int* __restrict__ hdr, int32_t* __restrict__ bad_slots, UDiv band_div) {
  // (measured: a counter set per wave is slower for its extra LDS reads in the scatter, four sub-counters per band selected by
  // lane & 3 change nothing -- the returning atomics on these ~10 counters are not what bounds the kernel)
  __shared__ unsigned int cnt[kBinSegs];
  __shared__ unsigned int base[kBinSegs + 1];
  __shared__ __attribute__((aligned(16))) unsigned short sorted[kBinSlots];
  const int b = blockIdx.y, tid = threadIdx.x;
  const long long beg = offsets[b], n = offsets[b + 1] - beg, rel = beg - offsets[0];
  const long long HW = (long long)H * W;
  if (rel + n > n_cap) {
    if (tid == 0) bad_slots[(long long)b * gridDim.x + blockIdx.x] = blockIdx.x == 0 ? kBinOverflow : 0;
    return;
  }
  // events the reference would raise IndexError for: counted per workgroup, one plain store per workgroup; pass 2 adds the
  // slots of a sample up into status[b] (no memset node, no global atomics)
  __shared__ int bad_wg;
  if (tid == 0) bad_wg = 0;
  const memhip_event_aug_t* a = augs ? augs + b : nullptr;
  const double t_last = (a && a->time_flip && n > 0) ? ev[4 * (beg + n - 1) + 2] : 0.0;
  const AugK K = aug_k(a);
  const long long nchunks = (n + kBinChunk - 1) / kBinChunk;
  const long long hbase = rel / kBinChunk + b;
  int bad = 0;
  // The rows of a workgroup's next chunk are requested as soon as the current rows have been turned into keys (their registers
  // are free then) and arrive under the counting sort and the key write-out of the current chunk; the barriers inside the
  // loop order LDS traffic only.  (Two workgroups per CU at 125 registers: without the prefetch the pair-exchange form needs
  // 138 registers, one workgroup per CU, and runs at 0.51 instead of 0.62 of 8 TB/s.)
  RawEv raw[kBinEvPerThread];
  const int slot = (tid & ~63) + ((tid & 1) << 5) + ((tid & 63) >> 1);        // this thread's row inside a step (load_raw_pair)
  auto fetch_row = [&](long long cc, int k) -> RawEv {
    return load_raw_pair(ev, beg, n, cc * kBinChunk + (long long)k * kBinThreads + (tid & ~63), tid & 63);
  };
  if (blockIdx.x < nchunks) {
#pragma unroll
    for (int k = 0; k < kBinEvPerThread; ++k) raw[k] = fetch_row(blockIdx.x, k);       // (n > 0: nchunks > 0)
  }
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    if (tid < 2 * nb) cnt[tid] = 0u;
    for (int i = tid; i < kBinSlots / 8; i += kBinThreads)                     // pad key everywhere first
      reinterpret_cast<uint4*>(sorted)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    lds_barrier();
    unsigned int key[kBinEvPerThread], where[kBinEvPerThread];
    // two halves: the next chunk's first four rows are requested as soon as the current first four are keys, its last four behind the
    // current last four -- the workgroup always has rows in flight while it computes (requested all at once behind the whole chunk,
    // they had only the short sort + write-out to land in, and nothing of this workgroup was in flight during its ~3 us of key arithmetic)
#pragma unroll
    for (int hk = 0; hk < kBinParts; ++hk) {
#pragma unroll
    for (int k = hk * (kBinEvPerThread / kBinParts); k < (hk + 1) * (kBinEvPerThread / kBinParts); ++k) {
      const long long i = c * kBinChunk + (long long)k * kBinThreads + slot;
      where[k] = 0xFFFFFFFFu;
      if (i >= n) continue;
      const Ev e = make_event_k(raw[k], n, i, K, t_last);
      if (!e.keep) continue;
      // trunc toward zero == ndarray.astype(int).  Coordinates below 2^30 in magnitude (every real stream) convert with
      // one v_cvt_i32_f64 each; anything larger takes the 64-bit conversion (the reference indexes with int64)
      long long flat;
      if (fabs(e.x) < 1073741824.0 && fabs(e.y) < 1073741824.0) flat = (long long)(int)e.x + (long long)W * (int)e.y;
      else flat = (long long)e.x + (long long)W * (long long)e.y;
      if (flat < -HW || flat >= HW) { ++bad; continue; }   // reference: IndexError
      if (flat < 0) flat += HW;                             // NumPy negative index
      const bool isneg = e.p == -1.0;
      if (!(e.p == 1.0) && !isneg) continue;
      const unsigned int f32 = (unsigned int)flat;          // 0 <= flat < H * W <= 64 * 40000
      const unsigned int band = udiv_apply(f32, band_div);
      key[k] = f32 - band * (unsigned int)band_px;           // < band_px <= 40 000: never the pad key
      const unsigned int seg = 2u * band + (isneg ? 1u : 0u);
      where[k] = (seg << 16) | atomicAdd(&cnt[seg], 1u);     // slot inside the (band, polarity) segment
    }
    if (c + gridDim.x < nchunks) {
#pragma unroll
      for (int k = hk * (kBinEvPerThread / kBinParts); k < (hk + 1) * (kBinEvPerThread / kBinParts); ++k) raw[k] = fetch_row(c + gridDim.x, k);
    }
    }
    lds_barrier();
    if (tid < 64) {                          // exclusive scan of <= 128 segment counts (rounded up to 8) in one wave: lane = band
      const unsigned int v0 = tid < nb ? ((cnt[2 * tid] + 7u) & ~7u) : 0u, v1 = tid < nb ? ((cnt[2 * tid + 1] + 7u) & ~7u) : 0u;
      unsigned int inc = v0 + v1;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned int u = __shfl_up(inc, o);
        if (tid >= o) inc += u;
      }
      if (tid < nb) { base[2 * tid] = inc - v0 - v1; base[2 * tid + 1] = inc - v1; }
      if (tid == nb - 1) base[2 * nb] = inc;
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < kBinEvPerThread; ++k)
      if (where[k] != 0xFFFFFFFFu) sorted[base[where[k] >> 16] + (where[k] & 0xFFFFu)] = (unsigned short)key[k];
    lds_barrier();
    const unsigned int total8 = base[2 * nb] >> 3;                              // 16-byte groups to write
    uint4* kout = reinterpret_cast<uint4*>(keys + (hbase + c) * kBinSlots);
    for (unsigned int j = tid; j < total8; j += kBinThreads) kout[j] = reinterpret_cast<const uint4*>(sorted)[j];
    if (tid <= 2 * nb) hdr[(hbase + c) * kBinHdr + tid] = base[tid] >> 3;      // boundaries in 16-byte groups
    lds_barrier();
  }
  __syncthreads();
  if (bad) atomicAdd(&bad_wg, bad);
  __syncthreads();
  if (tid == 0) bad_slots[(long long)b * gridDim.x + blockIdx.x] = bad_wg;
}

__device__ __forceinline__ void bin_count8(unsigned int* cnt, const uint4& q, bool isneg) {
  const unsigned int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int hhalf = 0; hhalf < 2; ++hhalf) {
      const unsigned int l = hhalf ? (w[i] >> 16) : (w[i] & 0xFFFFu);
      if (l == 0xFFFFu) continue;                                               // segment padding
      if (isneg) {
        atomicAdd(cnt + l, 0x10000u);
      } else {
        const unsigned int old = atomicAdd(cnt + l, 1u);
        if ((old & 0xFFFFu) == 0xFFFFu) atomicSub(cnt + l, 0x10000u);           // undo the carry into the neg half
      }
    }
}

__global__ __launch_bounds__(kAccThreads) void raster_bin_accum(
    const int64_t* __restrict__ offsets, int H, int W, int band_px, long long n_cap,
    const unsigned short* __restrict__ keys, const unsigned int* __restrict__ hdr, uint8_t* __restrict__ out,
    const int32_t* __restrict__ bad_slots, int nslots, int32_t* __restrict__ status) {
  extern __shared__ unsigned int cnt[];       // [band_px]: neg << 16 | pos
  const int band = blockIdx.x, b = blockIdx.y;
  if (band == 0 && threadIdx.x < 64) {        // status[b] = what the sample's pass-1 workgroups counted (raster_bin_keys)
    int v = 0;
    for (int i = threadIdx.x; i < nslots; i += 64) v += bad_slots[(long long)b * nslots + i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (threadIdx.x == 0) status[b] = v;
  }
  const long long HW = (long long)H * W;
  const long long lo = (long long)band * band_px;
  const int npx = (int)((lo + band_px <= HW ? band_px : HW - lo));
  for (int i = threadIdx.x; i < band_px; i += kAccThreads) cnt[i] = 0u;
  __syncthreads();
  const long long beg = offsets[b], n = offsets[b + 1] - beg, rel = beg - offsets[0];
  if (rel + n <= n_cap) {
    const long long nchunks = (n + kBinChunk - 1) / kBinChunk;
    const long long hbase = rel / kBinChunk + b;
    const unsigned int* h = hdr + hbase * kBinHdr + 2 * band;     // [s, m) = the band's positive keys of a chunk, [m, e) its negative ones
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int kWaves = kAccThreads / 64;
    const uint4 kPad = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    // software pipeline over this wave's chunks: header two chunks ahead, first 16-byte group one chunk ahead
    long long c = wave;
    unsigned int s = 0, m = 0, e = 0, s1 = 0, m1 = 0, e1 = 0;
    if (c < nchunks) { s = h[c * kBinHdr]; m = h[c * kBinHdr + 1]; e = h[c * kBinHdr + 2]; }
    if (c + kWaves < nchunks) { s1 = h[(c + kWaves) * kBinHdr]; m1 = h[(c + kWaves) * kBinHdr + 1]; e1 = h[(c + kWaves) * kBinHdr + 2]; }
    uint4 q = kPad;
    if (c < nchunks && s + lane < e) q = reinterpret_cast<const uint4*>(keys + (hbase + c) * kBinSlots)[s + lane];
    while (c < nchunks) {
      const long long c1 = c + kWaves, c2 = c + 2 * kWaves;
      unsigned int s2 = 0, m2 = 0, e2 = 0;
      if (c2 < nchunks) { s2 = h[c2 * kBinHdr]; m2 = h[c2 * kBinHdr + 1]; e2 = h[c2 * kBinHdr + 2]; }
      uint4 qn = kPad;
      if (c1 < nchunks && s1 + lane < e1) qn = reinterpret_cast<const uint4*>(keys + (hbase + c1) * kBinSlots)[s1 + lane];
      bin_count8(cnt, q, s + lane >= m);
      const uint4* kin = reinterpret_cast<const uint4*>(keys + (hbase + c) * kBinSlots);
      for (unsigned int j = s + 64 + lane; j < e; j += 64) bin_count8(cnt, kin[j], j >= m);   // segments beyond 512 keys
      c = c1; s = s1; m = m1; e = e1; s1 = s2; m1 = m2; e1 = e2; q = qn;
    }
  }
  __syncthreads();
  uint8_t* o = out + (size_t)b * 3 * HW + lo;
  for (int i = threadIdx.x * 4; i < npx; i += kAccThreads * 4) {
    if (i + 4 <= npx && ((lo + i) & 3) == 0) {
      const unsigned int c0 = cnt[i], c1 = cnt[i + 1], c2 = cnt[i + 2], c3 = cnt[i + 3];
      const unsigned int pv = (c0 & 0xFFu) | ((c1 & 0xFFu) << 8) | ((c2 & 0xFFu) << 16) | ((c3 & 0xFFu) << 24);
      const unsigned int nv = ((c0 >> 16) & 0xFFu) | (((c1 >> 16) & 0xFFu) << 8) | (((c2 >> 16) & 0xFFu) << 16) |
                              (((c3 >> 16) & 0xFFu) << 24);
      *reinterpret_cast<unsigned int*>(o + i) = pv;
      *reinterpret_cast<unsigned int*>(o + HW + i) = 0u;            // time-surface channel: zeros
      *reinterpret_cast<unsigned int*>(o + 2 * HW + i) = nv;
    } else {
      for (int k = i; k < i + 4 && k < npx; ++k) {
        o[k] = (uint8_t)(cnt[k] & 0xFFu);
        o[HW + k] = 0;
        o[2 * HW + k] = (uint8_t)((cnt[k] >> 16) & 0xFFu);
      }
    }
  }
}

// bands of whole pixels (not rows): band_px pixels each, the last one shorter
inline bool bin_geometry(int H, int W, int* band_px, int* nb, int want_nb = 0) {
  const long long HW = (long long)H * W;
  long long n = (HW + kBinBandPixels - 1) / kBinBandPixels;
  if (want_nb > n && want_nb <= kBinMaxBands) n = want_nb;
  if (n > kBinMaxBands) return false;
  long long px = (HW + n - 1) / n;
  px = (px + 3) & ~3ll;                     // 4-pixel aligned bands keep the u32 output stores
  if (px > kBinBandPixels) { px = kBinBandPixels; n = (HW + px - 1) / px; if (n > kBinMaxBands) return false; }
  *band_px = (int)px;
  *nb = (int)n;
  return true;
}

// Bands per sample for a batch of B samples.  Pass 2 runs one workgroup per (sample, band); a workgroup with > 80 KB of counters has a
// CU to itself and takes ~0.22 us per 1000 pixels of its band (launch, clearing and writing its planes) + ~220 us per million events /
// bands (its keys), so what the pass costs is set by how many ROUNDS of workgroups the chip needs.  480 x 640 needs eight bands of
// 38 400 pixels: 32 samples are 256 workgroups = one round (36.9 us under rocprofv3; the ten 15-bit-key bands of the first half of
// round 6 were 320 workgroups = two rounds, 44.3 us), 64 samples two rounds (72 us; twelve bands = three rounds of smaller workgroups:
// 68 us -- the model calls them equal).  The candidates run from the fewest bands the LDS allows to twice that; more bands cost pass 1
// padding and shorter segments, so a larger count has to promise 10 % and batches that fit one round keep the fewest bands.
inline int choose_bands(int H, int W, int B, long long n_events) {
  const int forced = memhip::opt(memhip::OPT_RASTER_BANDS);
  if (forced > 0) return forced;
  const long long HW = (long long)H * W;
  const int nmin = (int)((HW + kBinBandPixels - 1) / kBinBandPixels);
  int cus = memhip::max_cus();
  if (cus <= 0) cus = 256;
  if ((long long)B * nmin <= cus) return nmin;                                  // one round either way
  int best = nmin;
  double best_cost = 0.0;
  for (int nbv = nmin; nbv <= 2 * nmin + 4 && nbv <= kBinMaxBands; ++nbv) {
    const long long px = ((HW + nbv - 1) / nbv + 3) & ~3ll;
    const int per_cu = (size_t)px * 4 * 2 + 1024 <= 160 * 1024 ? 2 : 1;       // 1024-thread workgroups: two per CU at most
    const long long rounds = ((long long)B * nbv + (long long)cus * per_cu - 1) / ((long long)cus * per_cu);
    const double cost = (double)rounds * (0.22e-3 * (double)px + 220.0e-6 * ((double)n_events / B) / nbv) * (per_cu == 2 ? 2.0 : 1.0);
    if (nbv == nmin || cost < best_cost * 0.9) { best = nbv; best_cost = cost; }
  }
  return best;
}

}  // namespace

extern "C" size_t memhip_rasterize_workspace(int B, int H, int W) {
  size_t bins = (size_t)B * 3 * H * W * sizeof(unsigned int);
  bins = (bins + 15) & ~(size_t)15;
  return bins + (size_t)B * 2 * sizeof(unsigned long long);
}

extern "C" int memhip_rasterize_aug_f64(const double* ev, const int64_t* offsets,
                                        const memhip_event_aug_t* aug, int B, int H, int W,
                                        int time_surface, uint8_t* out, int32_t* status,
                                        void* workspace, size_t workspace_bytes,
                                        memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0, "rasterize: bad shape B=%d H=%d W=%d", B, H, W);
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(ev && offsets && out && status && workspace, "rasterize: null pointer");
  MEMHIP_REQUIRE(((uintptr_t)ev & 15) == 0, "rasterize: ev must be 16-byte aligned");
  const size_t need = memhip_rasterize_workspace(B, H, W);
  if (workspace_bytes < need)
    return memhip::fail(MEMHIP_EWORKSPACE, "rasterize: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t s = memhip::as_stream(stream);
  {
    // privatised-counter path (see raster_lds_kernel): small canvas, no time surface
    int rpb = kBandPixels / W;
    const int bands = rpb > 0 ? memhip::cdiv(H, rpb) : kMaxBands + 1;
    const bool lds_on = memhip::opt(memhip::OPT_RASTER_LDS) != 0;
    if (lds_on && !time_surface && bands <= kMaxBands && ((uintptr_t)out & 3) == 0) {
      rpb = memhip::cdiv(H, bands);                        // even bands
      MEMHIP_HIP(hipMemsetAsync(status, 0, (size_t)B * sizeof(int32_t), s));
      static bool attr_done = false;
      if (!attr_done) {
        MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raster_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBandPixels * 4));
        attr_done = true;
      }
      hipLaunchKernelGGL(raster_lds_kernel, dim3(bands, B), dim3(kLdsThreads), (size_t)2 * rpb * W * 4, s, ev, offsets,
                         aug, H, W, rpb, out, status);
      return memhip::check_launch("rasterize(lds)");
    }
  }
  size_t bins_bytes = ((size_t)B * 3 * H * W * sizeof(unsigned int) + 15) & ~(size_t)15;
  unsigned int* bins = (unsigned int*)workspace;
  unsigned long long* tstat = (unsigned long long*)((char*)workspace + bins_bytes);
  MEMHIP_HIP(hipMemsetAsync(workspace, 0, need, s));
  MEMHIP_HIP(hipMemsetAsync(status, 0, (size_t)B * sizeof(int32_t), s));
  // enough blocks per sample to fill the chip at small B, grid-stride beyond
  int bx = B >= 256 ? 8 : (B >= 32 ? 32 : 256);
  dim3 g1(bx, B);
  hipLaunchKernelGGL(raster_count, g1, dim3(kThreads), 0, s, ev, offsets, aug, H, W, time_surface,
                     bins, tstat, status, (const int32_t*)nullptr, 0LL);
  long long HW = (long long)H * W;
  int fx = (int)((HW + kThreads - 1) / kThreads);
  if (fx > 64) fx = 64;
  dim3 g2(fx, B);
  hipLaunchKernelGGL(raster_finalize, g2, dim3(kThreads), 0, s, ev, offsets, aug, H, W,
                     time_surface, bins, tstat, out, (const int32_t*)nullptr, 0LL);
  return memhip::check_launch("rasterize");
}

// Per-sample canvases (the reference's data-dependent "W = xs.max() + 1", datasets.py:572-575): sample b is
// rasterized on dims[b] = (H_b, W_b) and stored DENSELY ([3, H_b, W_b], pitch W_b) at the start of its slot of
// 3 * Hmax * Wmax bytes.  Global-atomic form (any canvas, time surface supported).
extern "C" int memhip_rasterize_var_f64(const double* ev, const int64_t* offsets, const memhip_event_aug_t* aug,
                                        const int32_t* dims, int B, int Hmax, int Wmax, int time_surface, uint8_t* out,
                                        int32_t* status, void* workspace, size_t workspace_bytes,
                                        memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && Hmax > 0 && Wmax > 0, "rasterize_var: bad shape B=%d Hmax=%d Wmax=%d", B, Hmax, Wmax);
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(ev && offsets && dims && out && status && workspace, "rasterize_var: null pointer");
  MEMHIP_REQUIRE(((uintptr_t)ev & 15) == 0, "rasterize_var: ev must be 16-byte aligned");
  const size_t need = memhip_rasterize_workspace(B, Hmax, Wmax);
  if (workspace_bytes < need)
    return memhip::fail(MEMHIP_EWORKSPACE, "rasterize_var: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t s = memhip::as_stream(stream);
  const long long slot = (long long)Hmax * Wmax;
  size_t bins_bytes = ((size_t)B * 3 * slot * sizeof(unsigned int) + 15) & ~(size_t)15;
  unsigned int* bins = (unsigned int*)workspace;
  unsigned long long* tstat = (unsigned long long*)((char*)workspace + bins_bytes);
  MEMHIP_HIP(hipMemsetAsync(workspace, 0, need, s));
  MEMHIP_HIP(hipMemsetAsync(status, 0, (size_t)B * sizeof(int32_t), s));
  MEMHIP_HIP(hipMemsetAsync(out, 0, (size_t)B * 3 * slot, s));
  int bx = B >= 256 ? 8 : (B >= 32 ? 32 : 256);
  hipLaunchKernelGGL(raster_count, dim3(bx, B), dim3(kThreads), 0, s, ev, offsets, aug, Hmax, Wmax, time_surface,
                     bins, tstat, status, dims, slot);
  int fx = (int)((slot + kThreads - 1) / kThreads);
  if (fx > 64) fx = 64;
  hipLaunchKernelGGL(raster_finalize, dim3(fx, B), dim3(kThreads), 0, s, ev, offsets, aug, Hmax, Wmax,
                     time_surface, bins, tstat, out, dims, slot);
  return memhip::check_launch("rasterize_var");
}

namespace {
// Resolution of the reference's data-dependent widths / heights ON THE DEVICE (no host sync), from the extent of the
// events (memhip_events_extent): truncation toward zero like ndarray.astype(np.int64).
//   stage 0 (extent of the RAW window): Aug_FlipEvsAlongX W = int(max x) + 1 (datasets.py:516); Aug_RandomShiftEvs
//            W = int(max x') + 1 after the flip, H = int(max y) + 1 (:537-540) -> aug[b].flip_w / filt_w / filt_h for the
//            fields whose MEMHIP_AUG_INFER_* bit is set in aug[b].infer.
//   stage 1 (extent AFTER the whole chain incl. the bounds filter): EventArrToImg canvas W = xs.max() + 1,
//            H = ys.max() + 1 (:572-575) -> dims[b]; fixed_h / fixed_w > 0 override.  Empty samples (the reference raises
//            ValueError on max() of an empty array) and canvases beyond (hmax, wmax) get dims (0, 0) and status |= 1 << 29.
__global__ void aug_resolve_kernel(const double* __restrict__ ext, memhip_event_aug_t* __restrict__ aug, int B, int stage,
                                   int fixed_h, int fixed_w, int hmax, int wmax, int32_t* __restrict__ dims,
                                   int32_t* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double maxx = ext[4 * b], maxy = ext[4 * b + 1], minx = ext[4 * b + 2];
  const bool empty = !(maxx >= minx);
  if (stage == 0) {
    memhip_event_aug_t a = aug[b];
    if (empty) return;
    double mx = maxx;
    if (a.infer & MEMHIP_AUG_INFER_FLIP_W) a.flip_w = (long long)maxx + 1;
    if (a.flip_x) mx = (double)(a.flip_w - 1) - minx;                 // max of W - 1 - x
    if (a.infer & MEMHIP_AUG_INFER_FILT_W) a.filt_w = (int)((long long)mx + 1);
    if (a.infer & MEMHIP_AUG_INFER_FILT_H) a.filt_h = (int)((long long)maxy + 1);
    aug[b] = a;
  } else {
    long long h = fixed_h > 0 ? fixed_h : (empty ? 0 : (long long)maxy + 1);
    long long w = fixed_w > 0 ? fixed_w : (empty ? 0 : (long long)maxx + 1);
    if (h <= 0 || w <= 0 || h > hmax || w > wmax || h * w > (long long)hmax * wmax) {
      h = w = 0;
      if (status) atomicOr(status + b, 1 << 29);
    }
    dims[2 * b] = (int)h;
    dims[2 * b + 1] = (int)w;
  }
}
}  // namespace

extern "C" int memhip_aug_resolve(const double* extent, memhip_event_aug_t* aug, int B, int stage, int fixed_h, int fixed_w,
                                  int hmax, int wmax, int32_t* dims, int32_t* status, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && (stage == 0 || stage == 1), "aug_resolve: bad arguments");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(extent && (stage == 0 ? aug != nullptr : dims != nullptr), "aug_resolve: null pointer");
  hipLaunchKernelGGL(aug_resolve_kernel, dim3((B + 255) / 256), dim3(256), 0, memhip::as_stream(stream), extent, aug, B, stage,
                     fixed_h, fixed_w, hmax, wmax, dims, status);
  return memhip::check_launch("aug_resolve");
}

extern "C" int memhip_rasterize_f64(const double* ev, const int64_t* offsets, int B, int H, int W,
                                    int time_surface, uint8_t* out, int32_t* status,
                                    void* workspace, size_t workspace_bytes,
                                    memhip_stream_t stream) {
  return memhip_rasterize_aug_f64(ev, offsets, nullptr, B, H, W, time_surface, out, status,
                                  workspace, workspace_bytes, stream);
}

namespace {
__global__ __launch_bounds__(kThreads) void extent_init(double* __restrict__ ext, int B) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < 4 * B) reinterpret_cast<unsigned long long*>(ext)[i] = 0ull;
}
// ext holds, during accumulation, u64 maxima of enc(x), enc(y), ~enc(x), ~enc(y)
__global__ __launch_bounds__(kThreads) void extent_accum(
    const double* __restrict__ ev, const int64_t* __restrict__ offsets,
    const memhip_event_aug_t* __restrict__ augs, unsigned long long* __restrict__ ext) {
  const int b = blockIdx.y;
  const long long beg = offsets[b], n = offsets[b + 1] - beg;
  const memhip_event_aug_t* a = augs ? augs + b : nullptr;
  unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n;
       i += (long long)gridDim.x * kThreads) {
    const Ev e = load_event(ev, beg, n, i, a, 0.0);
    if (!e.keep) continue;
    const unsigned long long ex = enc_f64(e.x), ey = enc_f64(e.y);
    m[0] = ex > m[0] ? ex : m[0];
    m[1] = ey > m[1] ? ey : m[1];
    m[2] = (~ex) > m[2] ? (~ex) : m[2];
    m[3] = (~ey) > m[3] ? (~ey) : m[3];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long v = __shfl_xor(m[k], o);
      m[k] = v > m[k] ? v : m[k];
    }
    if ((threadIdx.x & 63) == 0 && m[k]) atomicMax(ext + 4 * b + k, m[k]);
  }
}
__global__ __launch_bounds__(kThreads) void extent_decode(double* __restrict__ ext, int B) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= 4 * B) return;
  const unsigned long long v = reinterpret_cast<unsigned long long*>(ext)[i];
  const int k = i & 3;
  double d;
  if (v == 0ull) d = (k < 2) ? -INFINITY : INFINITY;
  else d = dec_f64(k < 2 ? v : ~v);
  ext[i] = d;
}
}  // namespace

extern "C" int memhip_events_extent(const double* ev, const int64_t* offsets,
                                    const memhip_event_aug_t* aug, int B, double* extent,
                                    memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0, "events_extent: bad B");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(ev && offsets && extent, "events_extent: null pointer");
  hipStream_t s = memhip::as_stream(stream);
  const int nb = memhip::cdiv(4LL * B, kThreads);
  hipLaunchKernelGGL(extent_init, dim3(nb), dim3(kThreads), 0, s, extent, B);
  hipLaunchKernelGGL(extent_accum, dim3(B >= 64 ? 8 : 64, B), dim3(kThreads), 0, s, ev, offsets, aug,
                     reinterpret_cast<unsigned long long*>(extent));
  hipLaunchKernelGGL(extent_decode, dim3(nb), dim3(kThreads), 0, s, extent, B);
  return memhip::check_launch("events_extent");
}

extern "C" size_t memhip_rasterize_binned_workspace(int B, int H, int W, int64_t n_events) {
  (void)H; (void)W;
  if (B <= 0 || n_events < 0) return 0;
  const size_t chunks = (size_t)n_events / kBinChunk + (size_t)B + 1;
  const size_t keys = (chunks * kBinSlots * sizeof(unsigned short) + 15) & ~(size_t)15;
  const size_t hdr = chunks * kBinHdr * sizeof(unsigned int);
  return keys + hdr + ((size_t)kBinKeyGrid + (size_t)B) * sizeof(int32_t);     // + one slot per pass-1 workgroup
}

extern "C" int memhip_rasterize_binned_f64(const double* ev, const int64_t* offsets,
                                           const memhip_event_aug_t* aug, int B, int H, int W,
                                           int64_t n_events, uint8_t* out, int32_t* status,
                                           void* workspace, size_t workspace_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && n_events >= 0, "rasterize_binned: bad shape B=%d H=%d W=%d", B, H, W);
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(ev && offsets && out && status && workspace, "rasterize_binned: null pointer");
  MEMHIP_REQUIRE(((uintptr_t)ev & 15) == 0 && ((uintptr_t)out & 3) == 0, "rasterize_binned: ev must be 16-byte, out 4-byte aligned");
  int band_px = 0, nb = 0;
  if (!bin_geometry(H, W, &band_px, &nb, choose_bands(H, W, B, n_events)))
    return memhip::fail(MEMHIP_EUNSUPPORTED, "rasterize_binned: %dx%d canvas needs more than %d bands", H, W, kBinMaxBands);
  const size_t need = memhip_rasterize_binned_workspace(B, H, W, n_events);
  if (workspace_bytes < need)
    return memhip::fail(MEMHIP_EWORKSPACE, "rasterize_binned: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t s = memhip::as_stream(stream);
  unsigned short* keys = (unsigned short*)workspace;
  const size_t keys_bytes = ((((size_t)n_events / kBinChunk + (size_t)B + 1) * kBinSlots * sizeof(unsigned short)) + 15) & ~(size_t)15;
  unsigned int* hdr = (unsigned int*)((char*)workspace + keys_bytes);
  const size_t chunk_slots = (size_t)n_events / kBinChunk + (size_t)B + 1;
  int32_t* bad_slots = (int32_t*)((char*)hdr + chunk_slots * kBinHdr * sizeof(unsigned int));
  static bool attr_done = false;
  if (!attr_done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raster_bin_accum),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kBinBandPixels * 4));
    attr_done = true;
  }
  // chunks per sample are only known on the device: enough workgroups per sample to fill the chip at
  // small B, grid-stride beyond
  const long long avg_chunks = n_events / ((long long)B * kBinChunk) + 1;
  // (4096 workgroups in all: 64 x 1 M events 462 -> 454 us against 2048, 470 with 1024; 32 x 1 M: 241 / 241 / 246 / 249 us
  // with 4096 / 2048 / 1024 / 512)
  long long bx = (kBinKeyGrid + B - 1) / B;                                    // (bx * B <= kBinKeyGrid + B slots)
  if (bx > 4 * avg_chunks) bx = 4 * avg_chunks;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(raster_bin_keys, dim3((unsigned)bx, B), dim3(kBinThreads), 0, s, ev, offsets, aug, H, W, band_px, nb,
                     (long long)n_events, keys, hdr, bad_slots, udiv_prepare((unsigned)band_px));
  hipLaunchKernelGGL(raster_bin_accum, dim3(nb, B), dim3(kAccThreads), (size_t)band_px * 4, s, offsets, H, W, band_px,
                     (long long)n_events, keys, hdr, out, bad_slots, (int)bx, status);
  return memhip::check_launch("rasterize_binned");
}
