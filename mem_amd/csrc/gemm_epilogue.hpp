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
};

__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }

// exact-erf GELU (nn.GELU default, modeling_finetune.py:57,62) and its derivative
__device__ __forceinline__ float gelu_f(float x) {
  return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
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
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = (__bf16)(da * gelu_grad_f(h));
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

}  // namespace memhip
