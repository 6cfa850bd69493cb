// Fused tensor-level event transforms (reference: mem/transforms.py:200-275 in the order of
// mem/datasets.py:637-653): ToTensor's /255, RemoveTimesurface, RemoveHotPixels (mean + k*std
// threshold over the two polarity channels, unbiased std -- or the top-k branch), Log/Gamma,
// NormalizeEvent (divide by the joint max).
//
// HBM-bound: per sample the polarity planes are read ONCE and the float32 output is written ONCE
// (3*H*W bytes in for the uint8 form -- 2*H*W when the time surface is dropped -- and 4*C*H*W out:
// 0.55 MB per 224 x 224 sample, 141 MB per batch of 256).  One 1024-thread workgroup per sample:
//   pass 1  16-byte loads of the two polarity planes; the raw bytes are parked in LDS (100 KB at 224^2) while the
//           float64 sums for mean / std are taken from the registers just loaded;
//   pass 2  (NormalizeEvent) the joint max of the transformed values, from LDS;
//   pass 3  transform, scale, 16-byte stores (a wave-instruction writes 1 KiB contiguous).
// The normalisation factor is applied before the only store, so nothing written is read back.  Samples whose planes
// do not fit LDS (480 x 640) and float32 inputs re-read their input (L2 hits) in passes 2 and 3 instead.
#include "common.h"

namespace {

constexpr int kT = 1024;
constexpr int kWaves = kT / 64;
constexpr size_t kStageMax = 128 * 1024;         // bytes of LDS for the two parked uint8 planes

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// torchvision ToTensor: uint8 -> float32, then true division by 255.  q = k * fl(1/255) corrected by one fused
// residual step is the correctly rounded quotient for every k in 0..255 (checked exhaustively against IEEE division;
// tests compare the kernel with torch's own division bit for bit): 3 instructions instead of the ~12 of a full divide.
__device__ __forceinline__ float u8_unit(unsigned k) {
  const float kf = (float)k, r = 1.0f / 255.0f;
  const float q = __fmul_rn(kf, r);
  const float rem = __builtin_fmaf(-q, 255.0f, kf);
  return __builtin_fmaf(rem, r, q);
}

// order-preserving map of a float onto an unsigned (for the top-k selection keys)
__device__ __forceinline__ unsigned f32_orderable(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long hot_key(float v, unsigned flat_idx) {
  return ((unsigned long long)f32_orderable(v) << 32) | flat_idx;
}

struct Xform {
  int flags;
  float thr, gamma;
  bool half_gamma, use_keys;
  unsigned long long key;
  unsigned HW;
  // a = pos, c = neg value of pixel p: hot-pixel zeroing (both polarities when either is hot), log, gamma
  __device__ __forceinline__ void operator()(float& a, float& c, unsigned p) const {
    if (flags & MEMHIP_EV_HOTPIX) {
      const bool hot = use_keys ? (hot_key(a, p) >= key || hot_key(c, HW + p) >= key) : (a > thr || c > thr);
      if (hot) { a = 0.f; c = 0.f; }
    }
    if (flags & MEMHIP_EV_LOG) { a = logf(a + 1.0f); c = logf(c + 1.0f); }
    if (flags & MEMHIP_EV_GAMMA) {
      a = half_gamma ? sqrtf(a) : powf(a, gamma);
      c = half_gamma ? sqrtf(c) : powf(c, gamma);
    }
  }
};

// V pixels of one plane starting at element idx (V = 16 / 4 / 1 for uint8, 4 / 1 for float32)
template <bool U8, int V>
__device__ __forceinline__ void load_px(const void* in, size_t idx, float (&v)[V]) {
  if constexpr (U8) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(in) + idx;
    if constexpr (V == 16) {
      const uint4 q = *reinterpret_cast<const uint4*>(p);
      const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = u8_unit((w[i >> 2] >> (8 * (i & 3))) & 255u);
    } else if constexpr (V == 4) {
      const unsigned w = *reinterpret_cast<const unsigned*>(p);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = u8_unit((w >> (8 * i)) & 255u);
    } else {
      v[0] = u8_unit(p[0]);
    }
  } else {
    const float* p = reinterpret_cast<const float*>(in) + idx;
    if constexpr (V == 4) {
      const float4 q = *reinterpret_cast<const float4*>(p);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      v[0] = p[0];
    }
  }
}

template <int V>
__device__ __forceinline__ void store_px(float* dst, const float (&v)[V]) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  else dst[0] = v[0];
}

// U8: uint8 [B,3,H,W] input (ToTensor fused) or float32.  V1 / V2: pixels per thread and iteration in pass 1 / in
// passes 2-3 (16,4: uint8 planes with H*W % 16 == 0; 4,4: float32 with H*W % 4 == 0; 1,1: anything).  STAGED: the two
// uint8 planes are parked in LDS by pass 1 (needs U8, V1 == 16, 2*H*W <= kStageMax).
template <bool U8, int V1, int V2, bool STAGED>
__global__ __launch_bounds__(kT) void event_norm_kernel(const void* __restrict__ in, int H, int W, int flags,
                                                        float num_stds, float gamma,
                                                        const unsigned long long* __restrict__ hot_keys,
                                                        float* __restrict__ out, int out_chans) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* red_d = reinterpret_cast<double*>(smem);                 // [2][kWaves]
  float* red_f = reinterpret_cast<float*>(smem + 2 * kWaves * sizeof(double));   // [kWaves]
  float* bc = red_f + kWaves;                                      // [2]
  unsigned char* stage = smem + 512;                               // [2][HW] raw bytes
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const size_t HW = (size_t)H * W;
  const size_t ibase = (size_t)b * 3 * HW;
  float* o = out + (size_t)b * out_chans * HW;
  float* o_pos = o;
  float* o_neg = o + (size_t)(out_chans - 1) * HW;

  Xform xf;
  xf.flags = flags; xf.gamma = gamma; xf.half_gamma = (gamma == 0.5f); xf.thr = 0.f; xf.HW = (unsigned)HW;
  xf.use_keys = hot_keys != nullptr;
  xf.key = hot_keys ? hot_keys[b] : 0ull;

  const bool need_stats = (flags & MEMHIP_EV_HOTPIX) && !hot_keys;
  if (need_stats || STAGED) {
    double s1 = 0.0, s2 = 0.0;
    for (size_t p = (size_t)tid * V1; p < HW; p += (size_t)kT * V1) {
      if constexpr (STAGED) {
        const uint4 qa = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + ibase + p);
        const uint4 qc = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + ibase + 2 * HW + p);
        *reinterpret_cast<uint4*>(stage + p) = qa;
        *reinterpret_cast<uint4*>(stage + HW + p) = qc;
        if (need_stats) {
          const unsigned wa[4] = {qa.x, qa.y, qa.z, qa.w}, wc[4] = {qc.x, qc.y, qc.z, qc.w};
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const double a = u8_unit((wa[i >> 2] >> (8 * (i & 3))) & 255u), c = u8_unit((wc[i >> 2] >> (8 * (i & 3))) & 255u);
            s1 += a + c;
            s2 += a * a + c * c;
          }
        }
      } else {
        float va[V1], vc[V1];
        load_px<U8, V1>(in, ibase + p, va);
        load_px<U8, V1>(in, ibase + 2 * HW + p, vc);
#pragma unroll
        for (int i = 0; i < V1; ++i) {
          const double a = va[i], c = vc[i];
          s1 += a + c;
          s2 += a * a + c * c;
        }
      }
    }
    if (need_stats) {
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) { red_d[wid] = s1; red_d[kWaves + wid] = s2; }
    }
    __syncthreads();                                   // also: the parked planes are complete
    if (need_stats) {
      if (tid == 0) {
        double t1 = 0, t2 = 0;
        for (int i = 0; i < kWaves; ++i) { t1 += red_d[i]; t2 += red_d[kWaves + i]; }
        const double n = 2.0 * (double)HW;
        const double mean = t1 / n;
        double var = (t2 - t1 * t1 / n) / (n - 1.0);   // torch.std: unbiased
        var = var > 0.0 ? var : 0.0;
        const float meanf = (float)mean, stdf = (float)sqrt(var);
        bc[0] = __fadd_rn(meanf, __fmul_rn(num_stds, stdf));   // fp32 mul then add, no fma
      }
      __syncthreads();
      xf.thr = bc[0];
    }
  }

  // the planes as passes 2 and 3 read them: V2 pixels of pos and neg at pixel p
  auto load2 = [&](size_t p, float (&va)[V2], float (&vc)[V2]) {
    if constexpr (STAGED) {
      const unsigned wa = *reinterpret_cast<const unsigned*>(stage + p), wc = *reinterpret_cast<const unsigned*>(stage + HW + p);
#pragma unroll
      for (int i = 0; i < 4; ++i) { va[i] = u8_unit((wa >> (8 * i)) & 255u); vc[i] = u8_unit((wc >> (8 * i)) & 255u); }
    } else {
      load_px<U8, V2>(in, ibase + p, va);
      load_px<U8, V2>(in, ibase + 2 * HW + p, vc);
    }
  };

  float factor = 1.0f;
  bool scale = false;
  if (flags & MEMHIP_EV_NORMALIZE) {
    float mx = -INFINITY;
    for (size_t p = (size_t)tid * V2; p < HW; p += (size_t)kT * V2) {
      float va[V2], vc[V2];
      load2(p, va, vc);
#pragma unroll
      for (int i = 0; i < V2; ++i) {
        xf(va[i], vc[i], (unsigned)p + i);
        mx = fmaxf(mx, fmaxf(va[i], vc[i]));
      }
    }
    mx = wave_max(mx);
    if (lane == 0) red_f[wid] = mx;
    __syncthreads();
    if (tid == 0) {
      float m = red_f[0];
      for (int i = 1; i < kWaves; ++i) m = fmaxf(m, red_f[i]);
      bc[1] = m;
    }
    __syncthreads();
    const float m = bc[1];
    if (m != 0.f) { factor = __fdiv_rn(1.0f, m); scale = true; }
  }

  for (size_t p = (size_t)tid * V2; p < HW; p += (size_t)kT * V2) {
    float va[V2], vc[V2];
    load2(p, va, vc);
#pragma unroll
    for (int i = 0; i < V2; ++i) {
      xf(va[i], vc[i], (unsigned)p + i);
      if (scale) { va[i] = __fmul_rn(va[i], factor); vc[i] = __fmul_rn(vc[i], factor); }
    }
    store_px<V2>(o_pos + p, va);
    store_px<V2>(o_neg + p, vc);
    if (out_chans == 3) {
      float vt[V2];
      if (flags & MEMHIP_EV_RM_TS) {
#pragma unroll
        for (int i = 0; i < V2; ++i) vt[i] = 0.f;
      } else {
        load_px<U8, V2>(in, ibase + HW + p, vt);
      }
      store_px<V2>(o + HW + p, vt);
    }
  }
}

// ---------------------------------------------------------------- RemoveHotPixels(num_hot_pixels = k)
// mem/transforms.py:257-263: the k largest entries of the flattened [pos, neg] planes are hot.  Per sample the kernel
// finds the k-th largest 64-bit key (orderable(value) << 32 | flat index) by an 8-pass byte-wise radix selection (LDS
// histograms); event_norm_kernel then zeroes every pixel with a key >= it.  Ties at the selection boundary: the
// reference's torch.argsort(stable=False) leaves the order of equal values to the sort implementation; here the
// larger flat index counts as larger (deterministic).  The reference's clamp (k >= sum / 4 -> k = sum / 4) is kept.
template <bool U8>
__global__ __launch_bounds__(kT) void hot_topk_kernel(const void* __restrict__ in, int H, int W, int num_hot_pixels,
                                                      unsigned long long* __restrict__ keys) {
  __shared__ double red_d[kWaves];
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ long long s_k;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const size_t HW = (size_t)H * W, n = 2 * HW;
  const size_t ibase = (size_t)b * 3 * HW;
  auto value = [&](size_t i) {                                     // entry i of x[0::2].flatten()
    const size_t idx = ibase + (i < HW ? i : i + HW);
    float v[1];
    load_px<U8, 1>(in, idx, v);
    return v[0];
  };
  double s = 0.0;
  for (size_t i = tid; i < n; i += kT) s += (double)value(i);
  s = wave_sum(s);
  if (lane == 0) red_d[wid] = s;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int i = 0; i < kWaves; ++i) t += red_d[i];
    const float quarter = __fdiv_rn((float)t, 4.0f);               // hist.sum() / 4 in float32
    long long k = num_hot_pixels;
    if ((float)num_hot_pixels >= quarter) k = (long long)quarter;  // int(tensor): truncation
    if (k > (long long)n) k = (long long)n;
    s_k = k;
    s_prefix = 0ull;
  }
  __syncthreads();
  if (s_k <= 0) {                                                  // argsort(...)[len - 0:] is empty: nothing is hot
    if (tid == 0) keys[b] = ~0ull;
    return;
  }
  for (int byte = 7; byte >= 0; --byte) {
    for (int i = tid; i < 256; i += kT) hist[i] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const int shift = 8 * byte;
    for (size_t i = tid; i < n; i += kT) {
      const unsigned long long key = hot_key(value(i), (unsigned)i);
      const bool match = byte == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8));
      if (match) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      long long k = s_k, acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + (long long)hist[d] >= k) break;
        acc += hist[d];
      }
      s_k = k - acc;                                               // rank inside the chosen digit's bucket
      s_prefix = prefix | ((unsigned long long)d << shift);
    }
    __syncthreads();
  }
  if (tid == 0) keys[b] = s_prefix;
}

template <bool U8, int V1, int V2, bool STAGED>
int launch_norm(const void* in, int B, int H, int W, int flags, float num_stds, float gamma,
                const unsigned long long* hot_keys, float* out, int out_chans, hipStream_t s) {
  const size_t lds = 512 + (STAGED ? 2 * (size_t)H * W : 0);
  auto kern = event_norm_kernel<U8, V1, V2, STAGED>;
  if (lds > 64 * 1024) {
    static bool attr_done = false;
    if (!attr_done) {
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(512 + kStageMax)));
      attr_done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(kT), lds, s, in, H, W, flags, num_stds, gamma, hot_keys, out, out_chans);
  return memhip::check_launch("event_norm");
}

int dispatch_norm(const void* in, int in_is_u8, int B, int H, int W, int flags, float num_stds, float gamma,
                  const unsigned long long* hot_keys, float* out, int out_chans, hipStream_t s) {
  const size_t HW = (size_t)H * W;
  MEMHIP_REQUIRE(2 * HW < (1ull << 32), "event_norm: canvas too large");
  const bool al16 = (reinterpret_cast<uintptr_t>(in) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  if (in_is_u8) {
    if (HW % 16 == 0 && al16) {
      if (2 * HW <= kStageMax) return launch_norm<true, 16, 4, true>(in, B, H, W, flags, num_stds, gamma, hot_keys, out, out_chans, s);
      return launch_norm<true, 16, 4, false>(in, B, H, W, flags, num_stds, gamma, hot_keys, out, out_chans, s);
    }
    return launch_norm<true, 1, 1, false>(in, B, H, W, flags, num_stds, gamma, hot_keys, out, out_chans, s);
  }
  if (HW % 4 == 0 && al16) return launch_norm<false, 4, 4, false>(in, B, H, W, flags, num_stds, gamma, hot_keys, out, out_chans, s);
  return launch_norm<false, 1, 1, false>(in, B, H, W, flags, num_stds, gamma, hot_keys, out, out_chans, s);
}

}  // namespace

extern "C" int memhip_event_norm(const void* in, int in_is_u8, int B, int H, int W, int flags,
                                 float num_stds, float gamma, float* out, int out_chans,
                                 memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0, "event_norm: bad shape");
  MEMHIP_REQUIRE(out_chans == 2 || out_chans == 3, "event_norm: out_chans must be 2 or 3");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out, "event_norm: null pointer");
  return dispatch_norm(in, in_is_u8, B, H, W, flags, num_stds, gamma, nullptr, out, out_chans, memhip::as_stream(stream));
}

extern "C" int memhip_event_norm_topk(const void* in, int in_is_u8, int B, int H, int W, int flags,
                                      int num_hot_pixels, float gamma, float* out, int out_chans,
                                      uint64_t* hot_keys, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && num_hot_pixels >= 0, "event_norm_topk: bad shape");
  MEMHIP_REQUIRE(out_chans == 2 || out_chans == 3, "event_norm_topk: out_chans must be 2 or 3");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out && hot_keys, "event_norm_topk: null pointer");
  MEMHIP_REQUIRE(2 * (size_t)H * W < (1ull << 32), "event_norm_topk: canvas too large");
  hipStream_t s = memhip::as_stream(stream);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(hot_keys);
  if (in_is_u8) hipLaunchKernelGGL(hot_topk_kernel<true>, dim3(B), dim3(kT), 0, s, in, H, W, num_hot_pixels, keys);
  else hipLaunchKernelGGL(hot_topk_kernel<false>, dim3(B), dim3(kT), 0, s, in, H, W, num_hot_pixels, keys);
  const int rc = memhip::check_launch("hot_topk");
  if (rc != MEMHIP_OK) return rc;
  return dispatch_norm(in, in_is_u8, B, H, W, flags | MEMHIP_EV_HOTPIX, 0.f, gamma, keys, out, out_chans, s);
}
