// Fused tensor-level event transforms (reference: mem/transforms.py:200-275 in the order of
// mem/datasets.py:637-653): ToTensor's /255, RemoveTimesurface, RemoveHotPixels (mean + k*std
// threshold over the two polarity channels, unbiased std), Log/Gamma, NormalizeEvent (divide by
// the joint max).  One 1024-thread workgroup per sample; the sample (<= 150 KB at 224^2) stays
// L2-resident across the three phases, so HBM sees it once in and once out.
#include "common.h"

namespace {

constexpr int kT = 1024;

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <bool U8>
__device__ __forceinline__ float load_px(const void* in, size_t idx) {
  if constexpr (U8) {
    // torchvision ToTensor: uint8 -> float32, then true division by 255
    return __fdiv_rn((float)reinterpret_cast<const uint8_t*>(in)[idx], 255.0f);
  } else {
    return reinterpret_cast<const float*>(in)[idx];
  }
}

template <bool U8>
__global__ __launch_bounds__(kT) void event_norm_kernel(const void* __restrict__ in, int H, int W,
                                                        int flags, float num_stds, float gamma,
                                                        float* __restrict__ out, int out_chans) {
  __shared__ double red_d[2][kT / 64];
  __shared__ float red_f[kT / 64];
  __shared__ float bc[2];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const size_t HW = (size_t)H * W;
  const size_t ibase = (size_t)b * 3 * HW;
  float* o = out + (size_t)b * out_chans * HW;
  float* o_pos = o;
  float* o_neg = o + (size_t)(out_chans - 1) * HW;

  float thr = 0.f;
  if (flags & MEMHIP_EV_HOTPIX) {
    double s1 = 0.0, s2 = 0.0;
    for (size_t p = tid; p < HW; p += kT) {
      const double a = load_px<U8>(in, ibase + p), c = load_px<U8>(in, ibase + 2 * HW + p);
      s1 += a + c;
      s2 += a * a + c * c;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { red_d[0][wid] = s1; red_d[1][wid] = s2; }
    __syncthreads();
    if (tid == 0) {
      double t1 = 0, t2 = 0;
      for (int i = 0; i < kT / 64; ++i) { t1 += red_d[0][i]; t2 += red_d[1][i]; }
      const double n = 2.0 * (double)HW;
      const double mean = t1 / n;
      double var = (t2 - t1 * t1 / n) / (n - 1.0);   // torch.std: unbiased
      var = var > 0.0 ? var : 0.0;
      const float meanf = (float)mean, stdf = (float)sqrt(var);
      bc[0] = __fadd_rn(meanf, __fmul_rn(num_stds, stdf));   // fp32 mul then add, no fma
    }
    __syncthreads();
    thr = bc[0];
  }

  float mx = -INFINITY;
  const bool half_gamma = (gamma == 0.5f);
  for (size_t p = tid; p < HW; p += kT) {
    float a = load_px<U8>(in, ibase + p), c = load_px<U8>(in, ibase + 2 * HW + p);
    if ((flags & MEMHIP_EV_HOTPIX) && (a > thr || c > thr)) { a = 0.f; c = 0.f; }
    if (flags & MEMHIP_EV_LOG) { a = logf(a + 1.0f); c = logf(c + 1.0f); }
    if (flags & MEMHIP_EV_GAMMA) {
      a = half_gamma ? sqrtf(a) : powf(a, gamma);
      c = half_gamma ? sqrtf(c) : powf(c, gamma);
    }
    o_pos[p] = a;
    o_neg[p] = c;
    if (out_chans == 3)
      o[HW + p] = (flags & MEMHIP_EV_RM_TS) ? 0.f : load_px<U8>(in, ibase + HW + p);
    mx = fmaxf(mx, fmaxf(a, c));
  }
  if (flags & MEMHIP_EV_NORMALIZE) {
    mx = wave_max(mx);
    if (lane == 0) red_f[wid] = mx;
    __syncthreads();
    if (tid == 0) {
      float m = red_f[0];
      for (int i = 1; i < kT / 64; ++i) m = fmaxf(m, red_f[i]);
      bc[1] = m;
    }
    __syncthreads();
    const float m = bc[1];
    if (m != 0.f) {
      const float factor = __fdiv_rn(1.0f, m);
      for (size_t p = tid; p < HW; p += kT) {   // each thread rescales what it wrote itself
        o_pos[p] = __fmul_rn(o_pos[p], factor);
        o_neg[p] = __fmul_rn(o_neg[p], factor);
      }
    }
  }
}

}  // namespace

extern "C" int memhip_event_norm(const void* in, int in_is_u8, int B, int H, int W, int flags,
                                 float num_stds, float gamma, float* out, int out_chans,
                                 memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0, "event_norm: bad shape");
  MEMHIP_REQUIRE(out_chans == 2 || out_chans == 3, "event_norm: out_chans must be 2 or 3");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out, "event_norm: null pointer");
  hipStream_t s = memhip::as_stream(stream);
  if (in_is_u8)
    hipLaunchKernelGGL(event_norm_kernel<true>, dim3(B), dim3(kT), 0, s, in, H, W, flags, num_stds,
                       gamma, out, out_chans);
  else
    hipLaunchKernelGGL(event_norm_kernel<false>, dim3(B), dim3(kT), 0, s, in, H, W, flags, num_stds,
                       gamma, out, out_chans);
  return memhip::check_launch("event_norm");
}
