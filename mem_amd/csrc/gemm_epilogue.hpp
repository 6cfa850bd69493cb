// Epilogues of the bf16 GEMM (see gemm.hip).  The rounding points mirror the reference under
// autocast: every Linear/Conv output is rounded to bf16 first (fp32 accumulate + fp32 bias), the
// residual stream / layer-scale / mask-token arithmetic that follows runs in fp32.
#pragma once
#include "common.h"

namespace memhip {

struct GemmArgs {   // == memhip_gemm_args_t
  const __bf16* A; const __bf16* B;
  long long lda, ldb;
  int M, N, K, epilogue;
  void* out0; long long ldo0;
  void* out1; long long ldo1;
  const float* bias;
  const float* vec1;
  float* resid; long long ldr;
  const void* aux; long long ldaux;
  const float* rowmask;
  float keep_prob;
  float colscale; int colscale_n;
  int rows_per_sample;
  int accumulate;
  float* colsum;   // optional: += column sums of the (rounded) primary output
};

__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }

// erf(x) by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-erff class accuracy) -- one
// rcp, one exp and five FMAs instead of libm's branchy erff: the GELU epilogues are VALU-bound
// otherwise.  The result feeds a bf16 rounding, 3 orders of magnitude coarser than the error.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float y = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(y, x);
}

// exact-erf GELU (nn.GELU default, modeling_finetune.py:57,62) and its derivative
__device__ __forceinline__ float gelu_f(float x) {
  return x * 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = __expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& p, int m, int n, float acc, float bias_n,
                                         float vec_n) {
  if constexpr (EPI == MEMHIP_EPI_BIAS_BF16) {
    float v = acc + bias_n;
    __bf16 y = (__bf16)v;
    if (n < p.colscale_n) y = (__bf16)((float)y * p.colscale);
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = y;
    if (p.colsum) atomicAdd(p.colsum + n, (float)y);
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU) {
    const __bf16 h = (__bf16)(acc + bias_n);
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = h;
    reinterpret_cast<__bf16*>(p.out1)[(long long)m * p.ldo1 + n] = (__bf16)gelu_f((float)h);
  } else if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    const __bf16 y = (__bf16)(acc + bias_n);
    if (p.out0) reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = y;
    float t = p.vec1 ? __fmul_rn(vec_n, (float)y) : (float)y;            // gamma * branch
    if (p.rowmask) t = __fmul_rn(__fdiv_rn(t, p.keep_prob), p.rowmask[m / p.rows_per_sample]);
    // residual input: aux (fp32, ldaux) when given, else in place
    const float xin = p.aux ? reinterpret_cast<const float*>(p.aux)[(long long)m * p.ldaux + n]
                            : p.resid[(long long)m * p.ldr + n];
    p.resid[(long long)m * p.ldr + n] = __fadd_rn(xin, t);
  } else if constexpr (EPI == MEMHIP_EPI_DGELU) {
    const float da = bf16_round(acc);
    const float h = (float)reinterpret_cast<const __bf16*>(p.aux)[(long long)m * p.ldaux + n];
    const __bf16 o = (__bf16)(da * gelu_grad_f(h));
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = o;
    if (p.colsum) atomicAdd(p.colsum + n, (float)o);
  } else if constexpr (EPI == MEMHIP_EPI_F32) {
    float* o = reinterpret_cast<float*>(p.out0) + (long long)m * p.ldo0 + n;
    *o = p.accumulate ? (*o + acc) : acc;
  } else if constexpr (EPI == MEMHIP_EPI_PATCH_EMBED) {
    // modeling_pretrain.py:101-108: x*(1-w) + mask_token*w, rows shifted by the cls token
    const float y = bf16_round(acc + bias_n);
    const int L = p.rows_per_sample;
    const int b = m / L, pi = m - b * L;
    const float w = (float)reinterpret_cast<const unsigned char*>(p.aux)[m];
    const float v = __fadd_rn(__fmul_rn(y, 1.0f - w), __fmul_rn(vec_n, w));
    p.resid[((long long)b * (L + 1) + 1 + pi) * p.ldr + n] = v;
  }
}


typedef __attribute__((ext_vector_type(8))) __bf16 ebf16x8;

__device__ __forceinline__ void ld8(const float* p, float* o) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float* o) {
  reinterpret_cast<float4*>(p)[0] = float4{o[0], o[1], o[2], o[3]};
  reinterpret_cast<float4*>(p)[1] = float4{o[4], o[5], o[6], o[7]};
}

// Row-vector form: 8 consecutive output columns n..n+7 of row m (n % 8 == 0, n + 8 <= N, every
// leading dimension a multiple of 8 elements): 16-byte global accesses only.
template <int EPI>
__device__ __forceinline__ void epilogue8(const GemmArgs& p, int m, int n, const float* acc, float* cs) {
  float bias[8];
  if (p.bias) ld8(p.bias + n, bias);
  else {
#pragma unroll
    for (int k = 0; k < 8; ++k) bias[k] = 0.f;
  }
  if constexpr (EPI == MEMHIP_EPI_BIAS_BF16) {
    ebf16x8 y;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      y[k] = (__bf16)(acc[k] + bias[k]);
      if (n + k < p.colscale_n) y[k] = (__bf16)((float)y[k] * p.colscale);
      cs[k] += (float)y[k];
    }
    *reinterpret_cast<ebf16x8*>(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n) = y;
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU) {
    ebf16x8 h, a;
#pragma unroll
    for (int k = 0; k < 8; ++k) { h[k] = (__bf16)(acc[k] + bias[k]); a[k] = (__bf16)gelu_f((float)h[k]); }
    *reinterpret_cast<ebf16x8*>(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n) = h;
    *reinterpret_cast<ebf16x8*>(reinterpret_cast<__bf16*>(p.out1) + (long long)m * p.ldo1 + n) = a;
  } else if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    ebf16x8 y;
    float g[8], x[8];
    if (p.vec1) ld8(p.vec1 + n, g);
    if (p.aux) ld8(reinterpret_cast<const float*>(p.aux) + (long long)m * p.ldaux + n, x);
    else ld8(p.resid + (long long)m * p.ldr + n, x);
    const float rm = p.rowmask ? p.rowmask[m / p.rows_per_sample] : 1.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      y[k] = (__bf16)(acc[k] + bias[k]);
      float t = p.vec1 ? __fmul_rn(g[k], (float)y[k]) : (float)y[k];
      if (p.rowmask) t = __fmul_rn(__fdiv_rn(t, p.keep_prob), rm);
      x[k] = __fadd_rn(x[k], t);
    }
    if (p.out0) *reinterpret_cast<ebf16x8*>(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n) = y;
    st8(p.resid + (long long)m * p.ldr + n, x);
  } else if constexpr (EPI == MEMHIP_EPI_DGELU) {
    const ebf16x8 h = *reinterpret_cast<const ebf16x8*>(reinterpret_cast<const __bf16*>(p.aux) + (long long)m * p.ldaux + n);
    ebf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = (__bf16)(bf16_round(acc[k]) * gelu_grad_f((float)h[k]));
      cs[k] += (float)o[k];
    }
    *reinterpret_cast<ebf16x8*>(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n) = o;
  } else if constexpr (EPI == MEMHIP_EPI_F32) {
    float* o = reinterpret_cast<float*>(p.out0) + (long long)m * p.ldo0 + n;
    float x[8];
    if (p.accumulate) {
      ld8(o, x);
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] += acc[k];
      st8(o, x);
    } else {
      st8(o, acc);
    }
  } else if constexpr (EPI == MEMHIP_EPI_PATCH_EMBED) {
    const int L = p.rows_per_sample;
    const int b = m / L, pi = m - b * L;
    const float w = (float)reinterpret_cast<const unsigned char*>(p.aux)[m];
    float mt[8], x[8];
    ld8(p.vec1 + n, mt);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      x[k] = __fadd_rn(__fmul_rn(bf16_round(acc[k] + bias[k]), 1.0f - w), __fmul_rn(mt[k], w));
    st8(p.resid + ((long long)b * (L + 1) + 1 + pi) * p.ldr + n, x);
  }
}

// Column sums of the rounded output (bias gradient of the producing Linear): cs[8] holds this lane's
// partial sums for columns n..n+7 over the rows it handled; lanes with equal (lane & 7) own the same
// columns.  One atomic per column per wave.
__device__ __forceinline__ void colsum_flush(const GemmArgs& p, int n, float* cs, int lane) {
  if (!p.colsum) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = cs[k];
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if ((lane >> 3) == 0) atomicAdd(p.colsum + n + k, v);
    cs[k] = 0.f;
  }
}

// all leading dimensions / pointers the vector epilogue touches are 16-byte friendly
__device__ __forceinline__ bool vec_ok(const GemmArgs& p) {
  return ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux) & 7) == 0 && (p.N & 7) == 0;
}

}  // namespace memhip
