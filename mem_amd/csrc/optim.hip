// Flat-buffer optimizer step: global gradient L2 norm (two-stage, deterministic), gradient
// clipping folded into the update, decoupled-weight-decay Adam.  Reference:
// torch.nn.utils.clip_grad_norm_ / get_grad_norm_ (mem/utils.py:360-366,380-392) and
// optim.AdamW with betas (0.9, 0.95) (mem/optim_factory.py:121,132-133).
// Parameters, gradients and both moments live in four flat fp32 buffers (tensors padded to
// 1024 elements), so the whole model is one streaming pass: 16 B/lane, 28 B of traffic per
// parameter -- pure HBM roofline work.
#include "common.h"

namespace {

using namespace memhip;

constexpr int kT = 256;
constexpr int kChunk = 1024;   // elements per (tensor-aligned) chunk == one workgroup iteration

__global__ __launch_bounds__(kT) void sqnorm_partial_kernel(const float* __restrict__ g, long long n4,
                                                            double* __restrict__ partial) {
  __shared__ double sd[kT / 64];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n4; i += (long long)gridDim.x * kT) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sd[0] + sd[1]) + (sd[2] + sd[3]);
}

__global__ __launch_bounds__(kT) void sqnorm_final_kernel(const double* __restrict__ partial, int nb,
                                                          float* __restrict__ norm_out) {
  __shared__ double sd[kT / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += kT) s += partial[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = (float)sqrt((sd[0] + sd[1]) + (sd[2] + sd[3]));
}

// torch.optim.AdamW single-tensor arithmetic, in its order:
//   p *= 1 - lr*wd ; m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2)*g*g ;
//   denom = sqrt(v)/sqrt(bc2) + eps ; p -= (lr/bc1) * m / denom
__global__ __launch_bounds__(kT) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   long long nchunks, const unsigned char* __restrict__ wd_flag,
                                                   float decay_mul, float b1, float b2, float eps,
                                                   float step_size, float bc2_sqrt,
                                                   const float* __restrict__ gnorm, float max_norm) {
  // Non-finite guard on the device: a NaN/Inf loss gives a NaN/Inf gradient norm; the update is then skipped as a whole
  // (parameters and both moments untouched), so the host's deferred isfinite(loss) abort
  // (engine_for_pretraining.py:219-228 in the reference, checked here at the next meter flush) finds clean weights.
  if (gnorm && !isfinite(gnorm[0])) return;
  float coef = 1.0f;
  if (max_norm > 0.f) {                          // clip_grad_norm_: g *= clamp(max_norm/(norm+1e-6), max=1)
    coef = max_norm / (gnorm[0] + 1e-6f);
    coef = coef < 1.0f ? coef : 1.0f;
  }
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const float decay = wd_flag[c] ? decay_mul : 1.0f;
    const long long i = c * (kChunk / 4) + threadIdx.x;
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define UPD(f)                                                        \
    {                                                                 \
      const float gr = gg.f * coef;                                   \
      pp.f *= decay;                                                  \
      mm.f = mm.f + (gr - mm.f) * (1.0f - b1);                        \
      vv.f = vv.f * b2 + (1.0f - b2) * gr * gr;                       \
      const float denom = sqrtf(vv.f) / bc2_sqrt + eps;               \
      pp.f = pp.f - step_size * (mm.f / denom);                       \
    }
    UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

// The same update with PER-GROUP learning rates (layer-wise lr decay of finetuning: every parameter group has its
// own lr = schedule x lr_scale and weight decay, mem/optim_factory.py:31-53,56-100).  group_of_chunk[c] indexes a
// small device table of {1 - lr_g*wd_g, lr_g / bias_correction1}, both computed on the host in double like torch.
__global__ __launch_bounds__(kT) void adamw_groups_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          long long nchunks, const unsigned char* __restrict__ group_of_chunk,
                                                          const float2* __restrict__ group_table, float b1, float b2,
                                                          float eps, float bc2_sqrt, const float* __restrict__ gnorm,
                                                          float max_norm) {
  if (gnorm && !isfinite(gnorm[0])) return;      // see adamw_kernel
  float coef = 1.0f;
  if (max_norm > 0.f) {
    coef = max_norm / (gnorm[0] + 1e-6f);
    coef = coef < 1.0f ? coef : 1.0f;
  }
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const float2 gt = group_table[group_of_chunk[c]];
    const float decay = gt.x, step_size = gt.y;
    const long long i = c * (kChunk / 4) + threadIdx.x;
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define UPD(f)                                                        \
    {                                                                 \
      const float gr = gg.f * coef;                                   \
      pp.f *= decay;                                                  \
      mm.f = mm.f + (gr - mm.f) * (1.0f - b1);                        \
      vv.f = vv.f * b2 + (1.0f - b2) * gr * gr;                       \
      const float denom = sqrtf(vv.f) / bc2_sqrt + eps;               \
      pp.f = pp.f - step_size * (mm.f / denom);                       \
    }
    UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

}  // namespace

extern "C" size_t memhip_grad_norm_workspace(void) { return 1024 * sizeof(double); }

extern "C" int memhip_grad_norm(const float* g, int64_t n, float* norm_out, void* workspace,
                                size_t workspace_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && n % 4 == 0 && g && norm_out && workspace, "grad_norm: bad arguments");
  if (workspace_bytes < memhip_grad_norm_workspace())
    return fail(MEMHIP_EWORKSPACE, "grad_norm: workspace too small");
  hipStream_t s = as_stream(stream);
  const long long n4 = n / 4;
  int nb = (int)((n4 + kT - 1) / kT);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(kT), 0, s, g, n4, (double*)workspace);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(kT), 0, s, (const double*)workspace, nb, norm_out);
  return check_launch("grad_norm");
}

extern "C" int memhip_adamw(float* p, const float* g, float* m, float* v, int64_t n,
                            const uint8_t* wd_flag_per_chunk, double lr, double beta1, double beta2,
                            double eps, double weight_decay, int step, const float* gnorm, double max_norm,
                            memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && n % kChunk == 0, "adamw: n must be a multiple of %d", kChunk);
  if (n == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(p && g && m && v && wd_flag_per_chunk && step >= 1, "adamw: bad arguments");
  MEMHIP_REQUIRE(max_norm <= 0.0 || gnorm, "adamw: clipping needs gnorm");
  // host scalars in double like torch's Python-side arithmetic, then cast once
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const long long nchunks = n / kChunk;
  int blocks = (int)(nchunks < 8192 ? nchunks : 8192);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(kT), 0, as_stream(stream), p, g, m, v, nchunks,
                     wd_flag_per_chunk, (float)(1.0 - lr * weight_decay), (float)beta1, (float)beta2,
                     (float)eps, step_size, bc2_sqrt, gnorm, (float)max_norm);
  return check_launch("adamw");
}

extern "C" int memhip_adamw_groups(float* p, const float* g, float* m, float* v, int64_t n,
                                   const uint8_t* group_of_chunk, const float* group_table, int n_groups,
                                   double beta1, double beta2, double eps, int step, const float* gnorm,
                                   double max_norm, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && n % kChunk == 0, "adamw_groups: n must be a multiple of %d", kChunk);
  if (n == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(p && g && m && v && group_of_chunk && group_table && n_groups >= 1 && n_groups <= 256 && step >= 1,
                 "adamw_groups: bad arguments");
  MEMHIP_REQUIRE(max_norm <= 0.0 || gnorm, "adamw_groups: clipping needs gnorm");
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const long long nchunks = n / kChunk;
  int blocks = (int)(nchunks < 8192 ? nchunks : 8192);
  hipLaunchKernelGGL(adamw_groups_kernel, dim3(blocks), dim3(kT), 0, as_stream(stream), p, g, m, v, nchunks,
                     group_of_chunk, reinterpret_cast<const float2*>(group_table), (float)beta1, (float)beta2, (float)eps,
                     (float)sqrt(bc2), gnorm, (float)max_norm);
  return check_launch("adamw_groups");
}
