// Row-wise / column-reducing kernels of the ViT block that sit between the MFMA GEMMs.
// All of them are HBM-bound: one wave per token row with 16-byte lane accesses (LayerNorm,
// cross-entropy), or column-owning threads sweeping a band of rows (the reductions that produce
// layer-scale / bias / LayerNorm-affine gradients), finished with one fp32 atomic per column per
// workgroup.  Reference ops: nn.LayerNorm(eps=1e-6) (mem/modeling_pretrain.py:132), the layer-scale
// + DropPath residual (mem/modeling_finetune.py:187-188), nn.CrossEntropyLoss + argmax accuracy
// (mem/engine_for_pretraining.py:152,233).
#include "common.h"

namespace {

using namespace memhip;

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ float wsum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// a / b from r = rcp(b): q0 = a*r, one residual correction (correctly rounded away from denormals)
__device__ __forceinline__ float div_newton(float a, float b, float r) {
  const float q0 = a * r;
  return fmaf(fmaf(-q0, b, a), r, q0);
}

constexpr int kMaxChunks = 8;   // float4 chunks per lane: D <= 64*4*8 = 2048

// ---------------------------------------------------------------- LayerNorm forward
// one wave per output row; x fp32 (row via row_idx), y bf16, mean/rstd saved for backward
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                     const int* __restrict__ row_idx, int R, int D,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps,
                                                     __bf16* __restrict__ y, long long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const long long src = row_idx ? row_idx[r] : r;
  const float4* xr = reinterpret_cast<const float4*>(x + src * ldx);
  const int nch = D >> 2;
  float4 v[kMaxChunks];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int i = lane + c * 64;
    if (i < nch) { v[c] = xr[i]; s += (v[c].x + v[c].y) + (v[c].z + v[c].w); }
  }
  const float mu = wsum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int i = lane + c * 64;
    if (i < nch) {
      const float a = v[c].x - mu, b = v[c].y - mu, cc = v[c].z - mu, d = v[c].w - mu;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float var = wsum(q) / (float)D;          // biased, as nn.LayerNorm
  const float rs = 1.0f / sqrtf(var + eps);
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  bf16x4* yr = reinterpret_cast<bf16x4*>(y + (long long)r * ldy);
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int i = lane + c * 64;
    if (i < nch) {
      const float4 g = g4[i], b = b4[i];
      bf16x4 o;
      o[0] = (__bf16)((v[c].x - mu) * rs * g.x + b.x);
      o[1] = (__bf16)((v[c].y - mu) * rs * g.y + b.y);
      o[2] = (__bf16)((v[c].z - mu) * rs * g.z + b.z);
      o[3] = (__bf16)((v[c].w - mu) * rs * g.w + b.w);
      yr[i] = o;
    }
  }
}

// ---------------------------------------------------------------- LayerNorm backward
// workgroup = 4 waves x kRowsPerWave rows; dx per row (wave reductions), dgamma/dbeta per lane
// column accumulated in registers over the workgroup's rows, then LDS -> one atomic per column.

// NCH = float4 chunks per lane (D <= 256 * NCH): the per-lane arrays are sized for THIS D -- sized for the
// maximum they cost 198 VGPRs (2 waves per SIMD) and the kernel could not hide HBM latency.
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const __bf16* __restrict__ dy, long long lddy,
                                                     const float* __restrict__ x, long long ldx,
                                                     const int* __restrict__ row_idx, int R, int D,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     float* __restrict__ dres, long long lddres,
                                                     int accumulate, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta) {
  extern __shared__ float red[];   // [4][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  float4 ag[NCH], ab[NCH], gm[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    ag[c] = float4{0, 0, 0, 0};
    ab[c] = float4{0, 0, 0, 0};
    gm[c] = (lane + c * 64 < nch) ? g4[lane + c * 64] : float4{0, 0, 0, 0};
  }
  // rows are dealt to (workgroup, wave) round-robin: the column accumulators persist over ALL of a
  // workgroup's rows, so the number of same-address atomics is gridDim.x per column, not R/32
  for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
    const long long src = row_idx ? row_idx[r] : r;
    const float4* xr = reinterpret_cast<const float4*>(x + src * ldx);
    const bf16x4* dyr = reinterpret_cast<const bf16x4*>(dy + (long long)r * lddy);
    const float mu = mean[r], rs = rstd[r];
    float4 xh[NCH], gg[NCH], prev[NCH];
    float4* o = reinterpret_cast<float4*>(dres + src * lddres);
    if (accumulate) {                       // issued with the row, not after the reductions
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + c * 64 < nch) prev[c] = o[lane + c * 64];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane + c * 64;
      if (i < nch) {
        const float4 xv = xr[i];
        const bf16x4 d4 = dyr[i];
        const float4 g = gm[c];
        const float d0 = (float)d4[0], d1 = (float)d4[1], d2 = (float)d4[2], d3 = (float)d4[3];
        xh[c] = float4{(xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs};
        gg[c] = float4{d0 * g.x, d1 * g.y, d2 * g.z, d3 * g.w};
        s1 += (gg[c].x + gg[c].y) + (gg[c].z + gg[c].w);
        s2 += (gg[c].x * xh[c].x + gg[c].y * xh[c].y) + (gg[c].z * xh[c].z + gg[c].w * xh[c].w);
        ag[c].x += d0 * xh[c].x; ag[c].y += d1 * xh[c].y; ag[c].z += d2 * xh[c].z; ag[c].w += d3 * xh[c].w;
        ab[c].x += d0; ab[c].y += d1; ab[c].z += d2; ab[c].w += d3;
      }
    }
    const float m1 = wsum(s1) / (float)D, m2 = wsum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane + c * 64;
      if (i < nch) {
        float4 d{rs * (gg[c].x - m1 - xh[c].x * m2), rs * (gg[c].y - m1 - xh[c].y * m2),
                 rs * (gg[c].z - m1 - xh[c].z * m2), rs * (gg[c].w - m1 - xh[c].w * m2)};
        if (accumulate) { const float4 p = prev[c]; d.x += p.x; d.y += p.y; d.z += p.z; d.w += p.w; }
        o[i] = d;
      }
    }
  }
  float4* rg = reinterpret_cast<float4*>(red + (size_t)wave * 2 * D);
  float4* rb = reinterpret_cast<float4*>(red + (size_t)wave * 2 * D + D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = lane + c * 64;
    if (i < nch) { rg[i] = ag[c]; rb[i] = ab[c]; }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < D; n += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += red[(size_t)w * 2 * D + n]; b += red[(size_t)w * 2 * D + D + n]; }
    atomicAdd(dgamma + n, a);
    atomicAdd(dbeta + n, b);
  }
}

// ---------------------------------------------------------------- residual-branch backward
// forward (modeling_finetune.py:187-188):  x += (gamma * y / keep) * mask[b]
// backward: dt = (dx * mask[b]) / keep ; dgamma += sum_m dt * y ; dy = bf16(dt * gamma) ;
//           dbias += sum_m dy   (the Linear that produced y)
// one wave per token row (16-byte lane accesses), rows dealt round-robin to a fixed grid so that the
// column sums cost gridDim.x atomics per column (same-address atomics are the slow part)
constexpr int kBrMaxChunks = 8;   // float4 chunks per lane: D <= 2048

template <int NCH>
__global__ __launch_bounds__(256) void branch_bwd_kernel(const float* __restrict__ dx, long long lddx,
                                                         const __bf16* __restrict__ y, long long ldy,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ rowmask, float keep,
                                                         int rps, int M, int D, __bf16* __restrict__ dyo,
                                                         long long lddy, float* __restrict__ dgamma,
                                                         float* __restrict__ dbias, const int* __restrict__ out_map) {
  // out_map (work-skipping stochastic depth): sample -> index of the sample among the KEPT ones, or -1.  The rows of a
  // dropped sample are neither read nor written; kept rows land at their compact position and are scaled by 1 / keep.
  extern __shared__ float red[];   // [4][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  float4 g[NCH], ag[NCH], ab[NCH];
  const float rk = __frcp_rn(keep);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = lane + c * 64;
    g[c] = (gamma && i < nch) ? reinterpret_cast<const float4*>(gamma)[i] : float4{1.f, 1.f, 1.f, 1.f};
    ag[c] = float4{0, 0, 0, 0};
    ab[c] = float4{0, 0, 0, 0};
  }
  // two rows per iteration: both rows' loads are in flight before either is consumed
  const int stride = gridDim.x * 4;
  for (int m0 = blockIdx.x * 4 + wave; m0 < M; m0 += 2 * stride) {
    float4 dv[2][NCH];
    bf16x4 yv[2][NCH];
    float kk[2];
    long long mo[2];                                       // output row (-1: nothing to do for this row)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * stride;
      const int mc = m < M ? m : M - 1;
      const float4* dr = reinterpret_cast<const float4*>(dx + (long long)mc * lddx);
      kk[u] = rowmask ? rowmask[mc / rps] : 1.f;
      mo[u] = m < M ? m : -1;
      if (out_map && m < M) {
        const int smp = mc / rps, co = out_map[smp];
        mo[u] = co < 0 ? -1 : (long long)co * rps + (mc - smp * rps);
      }
      if (mo[u] < 0) continue;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int i = lane + c * 64;
        if (i < nch) {
          dv[u][c] = dr[i];
          if (y) yv[u][c] = reinterpret_cast<const bf16x4*>(y + (long long)mc * ldy)[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (mo[u] < 0) continue;
      bf16x4* orow = reinterpret_cast<bf16x4*>(dyo + mo[u] * lddy);
      const float k = kk[u];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int i = lane + c * 64;
        if (i < nch) {
          float4 d = dv[u][c];
          if (rowmask || out_map) {            // (dx * mask) / keep: reciprocal + one Newton step = the IEEE quotient
            d.x = div_newton(d.x * k, keep, rk); d.y = div_newton(d.y * k, keep, rk);
            d.z = div_newton(d.z * k, keep, rk); d.w = div_newton(d.w * k, keep, rk);
          }
          if (y) {
            const bf16x4 yy = yv[u][c];
            ag[c].x += d.x * (float)yy[0]; ag[c].y += d.y * (float)yy[1];
            ag[c].z += d.z * (float)yy[2]; ag[c].w += d.w * (float)yy[3];
          }
          bf16x4 o;
          o[0] = (__bf16)(d.x * g[c].x); o[1] = (__bf16)(d.y * g[c].y);
          o[2] = (__bf16)(d.z * g[c].z); o[3] = (__bf16)(d.w * g[c].w);
          orow[i] = o;
          ab[c].x += (float)o[0]; ab[c].y += (float)o[1]; ab[c].z += (float)o[2]; ab[c].w += (float)o[3];
        }
      }
    }
  }
  float4* rg = reinterpret_cast<float4*>(red + (size_t)wave * 2 * D);
  float4* rb = reinterpret_cast<float4*>(red + (size_t)wave * 2 * D + D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = lane + c * 64;
    if (i < nch) { rg[i] = ag[c]; rb[i] = ab[c]; }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < D; n += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += red[(size_t)w * 2 * D + n]; b += red[(size_t)w * 2 * D + D + n]; }
    if (dgamma) atomicAdd(dgamma + n, a);
    if (dbias) atomicAdd(dbias + n, b);
  }
}

// ---------------------------------------------------------------- fused LayerNorm backward + next branch backward
// In the backward of a block the LayerNorm gradient is added to the residual-stream gradient dx and the
// very next kernel (the backward of the previous residual branch) reads that dx back: fused, the row of
// dx is produced, stored and consumed in registers -- one pass over the fp32 gradient stream less.
//   dx[r] += LN'(dy[r]) ;  dt = dx[r] * mask[r / rps] / keep ;  dyb[r] = bf16(dt * gb) ;
//   dgamma_ln += sum dy*xhat ; dbeta_ln += sum dy ; dgb += sum dt*y ; dbias_b += sum dyb
template <int NCH, bool HAS_Y>
__global__ __launch_bounds__(256) void ln_bwd_branch_kernel(const __bf16* __restrict__ dy, long long lddy,
                                                            const float* __restrict__ x, long long ldx, int R, int D,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ dres, long long lddres,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            const __bf16* __restrict__ yb, long long ldyb,
                                                            const float* __restrict__ gb, const float* __restrict__ rowmask,
                                                            float keep, int rps, __bf16* __restrict__ dyo, long long lddyo,
                                                            float* __restrict__ dgb, float* __restrict__ dbiasb,
                                                            const int* __restrict__ in_map, const int* __restrict__ out_map) {
  // Work-skipping stochastic depth: r runs over the rows of the residual stream (x, dres).  in_map: sample -> its index
  // among the samples the LayerNorm'ed branch KEPT (dy, mean, rstd hold those samples only), -1: that branch skipped the
  // sample, its rows get no LayerNorm gradient.  out_map: the same for the branch whose output gradient is produced
  // (dyo holds the kept samples only, scaled by 1 / keep), -1: no output row.  NULL = identity.
  extern __shared__ float red[];   // [4][4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  float4 ag[NCH], ab[NCH], bg[HAS_Y ? NCH : 1], bb[NCH];      // the gamma vectors are re-read per row (L1): 24 VGPRs
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* gb4 = reinterpret_cast<const float4*>(gb);
  const float rk = __frcp_rn(keep);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    ag[c] = ab[c] = bb[c] = float4{0, 0, 0, 0};
    if (HAS_Y) bg[c] = float4{0, 0, 0, 0};
  }
  // software prefetch: the raw loads of the wave's NEXT row are issued before the current row is reduced
  const int rstride = gridDim.x * 4;
  float4 xn[NCH], pn[NCH];
  bf16x4 dn[NCH], yn[HAS_Y ? NCH : 1];
  long long rin_n = 0, rout_n = 0;                 // compact rows of the row whose loads are in flight (-1: not taking part)
  auto load_row = [&](int r) {
    rin_n = r; rout_n = r;
    if (in_map || out_map) {
      const int smp = r / rps, off = r - smp * rps;
      if (in_map) { const int ci = in_map[smp]; rin_n = ci < 0 ? -1 : (long long)ci * rps + off; }
      if (out_map) { const int co = out_map[smp]; rout_n = co < 0 ? -1 : (long long)co * rps + off; }
    }
    if (rin_n < 0 && rout_n < 0) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)r * ldx);
    const bf16x4* dyr = reinterpret_cast<const bf16x4*>(dy + (rin_n < 0 ? 0 : rin_n) * lddy);
    const float4* o = reinterpret_cast<const float4*>(dres + (long long)r * lddres);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane + c * 64;
      if (i < nch) {
        pn[c] = o[i];
        if (rin_n >= 0) {
          xn[c] = xr[i];
          dn[c] = dyr[i];
        }
        if constexpr (HAS_Y) yn[c] = reinterpret_cast<const bf16x4*>(yb + (long long)r * ldyb)[i];
      }
    }
  };
  int r = blockIdx.x * 4 + wave;
  if (r < R) load_row(r);
  for (; r < R; r += rstride) {
    const long long rin = rin_n, rout = rout_n;
    if (rin < 0 && rout < 0) {                     // dropped by both branches: the row of dres stays as it is
      if (r + rstride < R) load_row(r + rstride);
      continue;
    }
    float4* o = reinterpret_cast<float4*>(dres + (long long)r * lddres);
    const float mu = rin >= 0 ? mean[rin] : 0.f, rs = rin >= 0 ? rstd[rin] : 0.f;
    const float km = rowmask ? rowmask[r / rps] : 1.f;
    float4 xh[NCH], gg[NCH], prev[NCH];
    bf16x4 yv[HAS_Y ? NCH : 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane + c * 64;
      if (i < nch) {
        prev[c] = pn[c];
        if constexpr (HAS_Y) yv[c] = yn[c];
        if (rin < 0) {                             // no LayerNorm gradient for this row (rs = 0 below: d = prev)
          xh[c] = gg[c] = float4{0.f, 0.f, 0.f, 0.f};
          continue;
        }
        const float4 xv = xn[c];
        const bf16x4 d4 = dn[c];
        const float d0 = (float)d4[0], d1 = (float)d4[1], d2 = (float)d4[2], d3 = (float)d4[3];
        xh[c] = float4{(xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs};
        const float4 gmc = g4[i];
        gg[c] = float4{d0 * gmc.x, d1 * gmc.y, d2 * gmc.z, d3 * gmc.w};
        s1 += (gg[c].x + gg[c].y) + (gg[c].z + gg[c].w);
        s2 += (gg[c].x * xh[c].x + gg[c].y * xh[c].y) + (gg[c].z * xh[c].z + gg[c].w * xh[c].w);
        ag[c].x += d0 * xh[c].x; ag[c].y += d1 * xh[c].y; ag[c].z += d2 * xh[c].z; ag[c].w += d3 * xh[c].w;
        ab[c].x += d0; ab[c].y += d1; ab[c].z += d2; ab[c].w += d3;
      }
    }
    if (r + rstride < R) load_row(r + rstride);
    const float m1 = wsum(s1) / (float)D, m2 = wsum(s2) / (float)D;
    bf16x4* orow = reinterpret_cast<bf16x4*>(dyo + (rout < 0 ? 0 : rout) * lddyo);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane + c * 64;
      if (i < nch) {
        float4 d{prev[c].x + rs * (gg[c].x - m1 - xh[c].x * m2), prev[c].y + rs * (gg[c].y - m1 - xh[c].y * m2),
                 prev[c].z + rs * (gg[c].z - m1 - xh[c].z * m2), prev[c].w + rs * (gg[c].w - m1 - xh[c].w * m2)};
        if (rin >= 0) o[i] = d;
        if (rout < 0) continue;
        if (rowmask || out_map) {
          d.x = div_newton(d.x * km, keep, rk); d.y = div_newton(d.y * km, keep, rk);
          d.z = div_newton(d.z * km, keep, rk); d.w = div_newton(d.w * km, keep, rk);
        }
        if constexpr (HAS_Y) {
          bg[c].x += d.x * (float)yv[c][0]; bg[c].y += d.y * (float)yv[c][1];
          bg[c].z += d.z * (float)yv[c][2]; bg[c].w += d.w * (float)yv[c][3];
        }
        const float4 gbc = gb ? gb4[i] : float4{1.f, 1.f, 1.f, 1.f};
        bf16x4 q;
        q[0] = (__bf16)(d.x * gbc.x); q[1] = (__bf16)(d.y * gbc.y);
        q[2] = (__bf16)(d.z * gbc.z); q[3] = (__bf16)(d.w * gbc.w);
        orow[i] = q;
        bb[c].x += (float)q[0]; bb[c].y += (float)q[1]; bb[c].z += (float)q[2]; bb[c].w += (float)q[3];
      }
    }
  }
  float4* r0 = reinterpret_cast<float4*>(red + (size_t)wave * 4 * D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = lane + c * 64;
    if (i < nch) { r0[i] = ag[c]; r0[nch + i] = ab[c]; r0[2 * nch + i] = HAS_Y ? bg[HAS_Y ? c : 0] : float4{0, 0, 0, 0}; r0[3 * nch + i] = bb[c]; }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < D; n += 256) {
    float a = 0.f, b = 0.f, c2 = 0.f, d2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* rw = red + (size_t)w * 4 * D;
      a += rw[n]; b += rw[D + n]; c2 += rw[2 * D + n]; d2 += rw[3 * D + n];
    }
    atomicAdd(dgamma + n, a);
    atomicAdd(dbeta + n, b);
    if (dgb) atomicAdd(dgb + n, c2);
    if (dbiasb) atomicAdd(dbiasb + n, d2);
  }
}


// Layer-scale gradient without the branch output:  x += gamma * y,  y = A W^T + b  gives
//   dgamma_c = sum_m dt[m,c] y[m,c] = (sum_k W[c,k] dW[c,k] + b_c db_c) / gamma_c
// because dW[c,k] = sum_m dY[m,c] A[m,k], db_c = sum_m dY[m,c] and dY = gamma * dt: the forward does not
// have to store y (77 MB per branch at B=256) and the backward does not read it.  One wave per channel.
__global__ __launch_bounds__(256) void layerscale_grad_kernel(const __bf16* __restrict__ W, long long ldw,
                                                              const float* __restrict__ dW, long long lddw,
                                                              const float* __restrict__ b, const float* __restrict__ db,
                                                              const float* __restrict__ gamma, int N, int K,
                                                              float* __restrict__ dgamma) {
  // A workgroup takes four channels and ALL of its 256 threads walk each row (a wave per channel was latency-bound:
  // 1024 waves on the chip for the 1024 x 4096 fc2 weights of ViT-L, 24 MB in 44-54 us); the loads of the four rows are
  // independent, so up to 16 pieces per thread are in flight.
  __shared__ float red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4096) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + c < N ? n0 + c : N - 1;
      const __bf16* wr = W + (long long)n * ldw;
      const float* gr = dW + (long long)n * lddw;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 1024 + (int)threadIdx.x * 4;
        if (k < K) {
          const bf16x4 w = *reinterpret_cast<const bf16x4*>(wr + k);
          const float4 g = *reinterpret_cast<const float4*>(gr + k);
          s[c] += (float)w[0] * g.x + (float)w[1] * g.y + (float)w[2] * g.z + (float)w[3] * g.w;
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float t = wsum(s[c]);
    if (lane == 0) red[wave][c] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4 && n0 + (int)threadIdx.x < N) {
    const int n = n0 + threadIdx.x;
    float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (b && db) t += b[n] * db[n];
    const float g = gamma[n];
    dgamma[n] = g != 0.f ? t / g : 0.f;
  }
}

// ---------------------------------------------------------------- patch-embed / token backward
// forward (modeling_pretrain.py:101-108): row b*(L+1) = cls ; row b*(L+1)+1+p = y*(1-w) + mask_token*w
// backward: dcls += dx[cls rows]; dmask_token += sum dx*w; dy = bf16(dx*(1-w))
// A workgroup takes samples b, b + gridDim.x, ... and keeps its column sums in registers: one atomic per column and
// WORKGROUP (atomics on one address serialise at ~0.17 us each: with a workgroup per sample the 2 x 256 of them per column
// were most of the kernel's 142 us); seven rows of loads in flight per thread.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx, long long lddx,
                                                        const unsigned char* __restrict__ mask, int B,
                                                        int L, int D, __bf16* __restrict__ dy,
                                                        long long lddy, float* __restrict__ dcls,
                                                        float* __restrict__ dmask) {
  constexpr int U = 7;
  for (int c = threadIdx.x * 4; c < D; c += 256 * 4) {
    float4 cls{0, 0, 0, 0}, am{0, 0, 0, 0};
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
      const float* row0 = dx + (long long)b * (L + 1) * lddx + c;
      const float4 dc = *reinterpret_cast<const float4*>(row0);
      cls.x += dc.x; cls.y += dc.y; cls.z += dc.z; cls.w += dc.w;
      const unsigned char* mk = mask + (long long)b * L;
      __bf16* out = dy + (long long)b * L * lddy + c;
      int p = 0;
      for (; p + U <= L; p += U) {
        float4 d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = *reinterpret_cast<const float4*>(row0 + (long long)(1 + p + u) * lddx);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float w = (float)mk[p + u], q = 1.0f - w;
          am.x += d[u].x * w; am.y += d[u].y * w; am.z += d[u].z * w; am.w += d[u].w * w;
          bf16x4 o;
          o[0] = (__bf16)(d[u].x * q); o[1] = (__bf16)(d[u].y * q); o[2] = (__bf16)(d[u].z * q); o[3] = (__bf16)(d[u].w * q);
          *reinterpret_cast<bf16x4*>(out + (long long)(p + u) * lddy) = o;
        }
      }
      for (; p < L; ++p) {
        const float4 d = *reinterpret_cast<const float4*>(row0 + (long long)(1 + p) * lddx);
        const float w = (float)mk[p], q = 1.0f - w;
        am.x += d.x * w; am.y += d.y * w; am.z += d.z * w; am.w += d.w * w;
        bf16x4 o;
        o[0] = (__bf16)(d.x * q); o[1] = (__bf16)(d.y * q); o[2] = (__bf16)(d.z * q); o[3] = (__bf16)(d.w * q);
        *reinterpret_cast<bf16x4*>(out + (long long)p * lddy) = o;
      }
    }
    atomicAdd(dcls + c, cls.x); atomicAdd(dcls + c + 1, cls.y);
    atomicAdd(dcls + c + 2, cls.z); atomicAdd(dcls + c + 3, cls.w);
    atomicAdd(dmask + c, am.x); atomicAdd(dmask + c + 1, am.y);
    atomicAdd(dmask + c + 2, am.z); atomicAdd(dmask + c + 3, am.w);
  }
}

// ---------------------------------------------------------------- softmax cross-entropy
// one 256-thread workgroup per masked-token row; logits bf16 (the lm_head output under autocast),
// statistics in fp32.  Writes the row loss, "argmax == label" and (in place) dlogits = (softmax -
// onehot) * grad_scale as bf16.
template <int VPT>   // bf16x8 chunks per thread
__global__ __launch_bounds__(256) void ce_kernel(__bf16* __restrict__ logits, long long ld,
                                                 const long long* __restrict__ labels, int V,
                                                 float grad_scale, float* __restrict__ row_loss,
                                                 int* __restrict__ row_correct, int write_grad) {
  __shared__ float sm[4];
  __shared__ int si[4];
  __shared__ float bc[2];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16x8* row = reinterpret_cast<bf16x8*>(logits + (long long)r * ld);
  const int nch = V >> 3;
  float v[VPT][8];
  float mx = -INFINITY;
  int amax = 0x7fffffff;
#pragma unroll
  for (int c = 0; c < VPT; ++c) {
    const int i = tid + c * 256;
    if (i < nch) {
      const bf16x8 t = row[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[c][k] = (float)t[k];
        if (v[c][k] > mx) { mx = v[c][k]; amax = i * 8 + k; }   // first index of the max in this thread
      }
    }
  }
  // (max, smallest index) reduction == torch.max(-1) on CPU (first occurrence)
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o);
    const int oi = __shfl_xor(amax, o);
    if (om > mx || (om == mx && oi < amax)) { mx = om; amax = oi; }
  }
  if (lane == 0) { sm[wave] = mx; si[wave] = amax; }
  __syncthreads();
  if (tid == 0) {
    float m = sm[0]; int a = si[0];
    for (int w = 1; w < 4; ++w) if (sm[w] > m || (sm[w] == m && si[w] < a)) { m = sm[w]; a = si[w]; }
    bc[0] = m; si[0] = a;
  }
  __syncthreads();
  mx = bc[0];
  amax = si[0];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < VPT; ++c) {
    const int i = tid + c * 256;
    if (i < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[c][k] = __expf(v[c][k] - mx); s += v[c][k]; }
    }
  }
  s = wsum(s);
  __syncthreads();
  if (lane == 0) sm[wave] = s;
  __syncthreads();
  if (tid == 0) bc[1] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  const float sum = bc[1];
  const long long lab = labels[r];
  if (tid == 0) {
    // a label outside [0, V) (torch: "Target out of bounds") must not become an out-of-bounds read: the row's loss is
    // NaN, which the training loop's non-finite-loss abort reports (engine_for_pretraining.py:154-156)
    const bool lab_ok = lab >= 0 && lab < V;
    const float zl = lab_ok ? (float)logits[(long long)r * ld + lab] : __builtin_nanf("");
    row_loss[r] = (mx + __logf(sum)) - zl;        // -log_softmax[label]
    row_correct[r] = (amax == (int)lab) ? 1 : 0;
  }
  if (write_grad) {
    const float inv = 1.0f / sum;
    __syncthreads();                              // tid 0 has read logits[label] before it is overwritten
#pragma unroll
    for (int c = 0; c < VPT; ++c) {
      const int i = tid + c * 256;
      if (i < nch) {
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float p = v[c][k] * inv;
          if ((long long)(i * 8 + k) == lab) p -= 1.0f;
          o[k] = (__bf16)(p * grad_scale);
        }
        row[i] = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss,
                                                        const int* __restrict__ row_correct, int M,
                                                        float* __restrict__ out) {
  __shared__ double sd[4];
  __shared__ int sc[4];
  double s = 0.0;
  int c = 0;
  for (int i = threadIdx.x; i < M; i += 256) { s += (double)row_loss[i]; c += row_correct[i]; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
  if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (sd[0] + sd[1]) + (sd[2] + sd[3]);
    const int k = sc[0] + sc[1] + sc[2] + sc[3];
    out[0] = (float)(t / (double)M);          // nn.CrossEntropyLoss(reduction="mean")
    out[1] = (float)k / (float)M;             // mlm_acc
  }
}

}  // namespace

extern "C" int memhip_layernorm_fwd(const float* x, int64_t ldx, const int32_t* row_idx, int R, int D,
                                    const float* gamma, const float* beta, float eps, void* y,
                                    int64_t ldy, float* mean, float* rstd, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * kMaxChunks, "layernorm_fwd: bad R=%d D=%d", R, D);
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  MEMHIP_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "layernorm_fwd: ld must be a multiple of 4");
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, as_stream(stream), x, (long long)ldx,
                     row_idx, R, D, gamma, beta, eps, (__bf16*)y, (long long)ldy, mean, rstd);
  return check_launch("layernorm_fwd");
}

extern "C" int memhip_layernorm_bwd(const void* dy, int64_t lddy, const float* x, int64_t ldx,
                                    const int32_t* row_idx, int R, int D, const float* gamma,
                                    const float* mean, const float* rstd, float* dres, int64_t lddres,
                                    int accumulate, float* dgamma, float* dbeta, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * kMaxChunks, "layernorm_bwd: bad R=%d D=%d", R, D);
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dy && x && gamma && mean && rstd && dres && dgamma && dbeta, "layernorm_bwd: null pointer");
  MEMHIP_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddres % 4 == 0, "layernorm_bwd: ld must be a multiple of 4");
  int grid = cdiv(R, 4);
  const int cap = opt(OPT_LN_BWD_GRID) * 4 / 3;      // (4/3 of the fused kernel's grid: this one needs half the LDS per workgroup)
  if (grid > cap) grid = cap;
#define LNB_LAUNCH(N)                                                                                    \
  hipLaunchKernelGGL(ln_bwd_kernel<N>, dim3(grid), dim3(256), (size_t)8 * D * sizeof(float), as_stream(stream), \
                     (const __bf16*)dy, (long long)lddy, x, (long long)ldx, row_idx, R, D, gamma, mean, rstd, dres, \
                     (long long)lddres, accumulate, dgamma, dbeta)
  const int nchl = cdiv(D / 4, 64);
  if (nchl <= 1) LNB_LAUNCH(1);
  else if (nchl <= 2) LNB_LAUNCH(2);
  else if (nchl <= 3) LNB_LAUNCH(3);
  else if (nchl <= 4) LNB_LAUNCH(4);
  else LNB_LAUNCH(8);
#undef LNB_LAUNCH
  return check_launch("layernorm_bwd");
}

extern "C" int memhip_branch_bwd_map(const float* dx, int64_t lddx, const void* y, int64_t ldy,
                                     const float* gamma, const float* rowmask, float keep_prob,
                                     int rows_per_sample, int M, int D, void* dy, int64_t lddy,
                                     float* dgamma, float* dbias, const int32_t* out_map, memhip_stream_t stream) {
  MEMHIP_REQUIRE(M >= 0 && D > 0 && D % 4 == 0, "branch_bwd: bad M=%d D=%d", M, D);
  MEMHIP_REQUIRE(!out_map || (!rowmask && !y && rows_per_sample > 0), "branch_bwd: out_map excludes rowmask / y");
  if (M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dx && dy, "branch_bwd: null pointer");
  MEMHIP_REQUIRE(y || !dgamma, "branch_bwd: dgamma needs y (or use memhip_layerscale_grad)");
  MEMHIP_REQUIRE(lddx % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0, "branch_bwd: ld must be a multiple of 4");
  MEMHIP_REQUIRE(D <= 64 * 4 * kBrMaxChunks, "branch_bwd: D=%d too large", D);
  int grid = cdiv(M, 4);
  if (grid > 1024) grid = 1024;
#define BRB_LAUNCH(N)                                                                                    \
  hipLaunchKernelGGL(branch_bwd_kernel<N>, dim3(grid), dim3(256), (size_t)8 * D * sizeof(float), as_stream(stream), \
                     dx, (long long)lddx, (const __bf16*)y, (long long)ldy, gamma, rowmask, keep_prob,    \
                     rows_per_sample > 0 ? rows_per_sample : 1, M, D, (__bf16*)dy, (long long)lddy, dgamma, dbias, \
                     (const int*)out_map)
  const int nchl = cdiv(D / 4, 64);
  if (nchl <= 1) BRB_LAUNCH(1);
  else if (nchl <= 2) BRB_LAUNCH(2);
  else if (nchl <= 3) BRB_LAUNCH(3);
  else if (nchl <= 4) BRB_LAUNCH(4);
  else BRB_LAUNCH(8);
#undef BRB_LAUNCH
  return check_launch("branch_bwd");
}

extern "C" int memhip_branch_bwd(const float* dx, int64_t lddx, const void* y, int64_t ldy,
                                 const float* gamma, const float* rowmask, float keep_prob,
                                 int rows_per_sample, int M, int D, void* dy, int64_t lddy,
                                 float* dgamma, float* dbias, memhip_stream_t stream) {
  return memhip_branch_bwd_map(dx, lddx, y, ldy, gamma, rowmask, keep_prob, rows_per_sample, M, D, dy, lddy, dgamma, dbias,
                               nullptr, stream);
}

extern "C" int memhip_embed_bwd(const float* dx, int64_t lddx, const uint8_t* mask, int B, int L, int D,
                                void* dy, int64_t lddy, float* dcls, float* dmask_token,
                                memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && L > 0 && D > 0 && D % 4 == 0, "embed_bwd: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dx && mask && dy && dcls && dmask_token, "embed_bwd: null pointer");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(B < 128 ? B : 128), dim3(256), 0, as_stream(stream), dx, (long long)lddx, mask, B,
                     L, D, (__bf16*)dy, (long long)lddy, dcls, dmask_token);
  return check_launch("embed_bwd");
}

extern "C" int memhip_cross_entropy(void* logits, int64_t ld, const int64_t* labels, int M, int V,
                                    float grad_scale, float* row_loss, int32_t* row_correct,
                                    int write_grad, float* out2, memhip_stream_t stream) {
  MEMHIP_REQUIRE(M >= 0 && V > 0 && V % 8 == 0 && V <= 256 * 8 * 8, "cross_entropy: V=%d unsupported", V);
  if (M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(logits && labels && row_loss && row_correct && out2, "cross_entropy: null pointer");
  MEMHIP_REQUIRE(ld % 8 == 0, "cross_entropy: ld must be a multiple of 8");
  hipStream_t s = as_stream(stream);
  const int nch = V / 8, vpt = cdiv(nch, 256);
  __bf16* lg = (__bf16*)logits;
  const long long* lab = (const long long*)labels;
#define CE_LAUNCH(N) hipLaunchKernelGGL(ce_kernel<N>, dim3(M), dim3(256), 0, s, lg, (long long)ld, lab, V, \
                                        grad_scale, row_loss, row_correct, write_grad)
  if (vpt <= 1) CE_LAUNCH(1);
  else if (vpt <= 2) CE_LAUNCH(2);
  else if (vpt <= 4) CE_LAUNCH(4);
  else CE_LAUNCH(8);
#undef CE_LAUNCH
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss, row_correct, M, out2);
  return check_launch("cross_entropy");
}

extern "C" int memhip_layernorm_bwd_branch_map(const void* dy, int64_t lddy, const float* x, int64_t ldx, int R, int D,
                                               const float* gamma, const float* mean, const float* rstd, float* dres,
                                               int64_t lddres, float* dgamma, float* dbeta, const void* y_branch,
                                               int64_t ldyb, const float* gamma_branch, const float* rowmask,
                                               float keep_prob, int rows_per_sample, void* dy_branch, int64_t lddyb,
                                               float* dgamma_branch, float* dbias_branch, const int32_t* in_map,
                                               const int32_t* out_map, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * 4, "layernorm_bwd_branch: D=%d unsupported (<= 1024)", D);
  MEMHIP_REQUIRE(!(in_map || out_map) || (!rowmask && !y_branch && rows_per_sample > 0),
                 "layernorm_bwd_branch: sample maps exclude rowmask / y_branch");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dy && x && gamma && mean && rstd && dres && dgamma && dbeta && dy_branch,
                 "layernorm_bwd_branch: null pointer");
  MEMHIP_REQUIRE(y_branch || !dgamma_branch, "layernorm_bwd_branch: dgamma_branch needs y_branch");
  MEMHIP_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddres % 4 == 0 && ldyb % 4 == 0 && lddyb % 4 == 0,
                 "layernorm_bwd_branch: ld must be a multiple of 4");
  int grid = cdiv(R, 4);
  int cap = opt(OPT_LN_BWD_GRID);                      // 2048 (round 5).  768 = 3 resident workgroups per CU = one full round was the optimum of the
                                                       // kernel ALONE; inside the step it runs beside the weight-gradient workgroups of the other stream (128 KB of
                                                       // LDS each: a CU that holds one has no room for this kernel's 48 KB) and more, shorter workgroups find the
                                                       // free CUs sooner: 512 / 768 / 1024 / 1536 / 2048 / 3072 / 4096 = 35.0 / 34.64 / 34.6 / 34.45 / 34.35-34.43 /
                                                       // 34.5 / 34.65 ms per step (tools/exp/r05_run27.sh); ViT-L step 219 -> 216 ms
  if (D > 768 && cap > 512) cap = 512;                 // D = 1024: 64 KiB of LDS and 212 VGPRs per workgroup, two per CU (tools/ln_bwd_probe.py:
                                                       // 76 864 rows 301 -> 262 us, 19 216 rows 69 -> 64 us)
  if (grid > cap) grid = cap;
#define LBB_LAUNCH(N)                                                                                    \
  if (y_branch) LBB_LAUNCH2(N, true); else LBB_LAUNCH2(N, false)
#define LBB_LAUNCH2(N, Y)                                                                                \
  hipLaunchKernelGGL((ln_bwd_branch_kernel<N, Y>), dim3(grid), dim3(256), (size_t)16 * D * sizeof(float), \
                     as_stream(stream), (const __bf16*)dy, (long long)lddy, x, (long long)ldx, R, D, gamma, mean, \
                     rstd, dres, (long long)lddres, dgamma, dbeta, (const __bf16*)y_branch, (long long)ldyb, \
                     gamma_branch, rowmask, keep_prob, rows_per_sample > 0 ? rows_per_sample : 1,         \
                     (__bf16*)dy_branch, (long long)lddyb, dgamma_branch, dbias_branch, (const int*)in_map, \
                     (const int*)out_map)
  const int nchl = cdiv(D / 4, 64);
  if (nchl <= 1) LBB_LAUNCH(1);
  else if (nchl <= 2) LBB_LAUNCH(2);
  else if (nchl <= 3) LBB_LAUNCH(3);
  else LBB_LAUNCH(4);
#undef LBB_LAUNCH
#undef LBB_LAUNCH2
  return check_launch("layernorm_bwd_branch");
}

extern "C" int memhip_layernorm_bwd_branch(const void* dy, int64_t lddy, const float* x, int64_t ldx, int R, int D,
                                           const float* gamma, const float* mean, const float* rstd, float* dres,
                                           int64_t lddres, float* dgamma, float* dbeta, const void* y_branch,
                                           int64_t ldyb, const float* gamma_branch, const float* rowmask,
                                           float keep_prob, int rows_per_sample, void* dy_branch, int64_t lddyb,
                                           float* dgamma_branch, float* dbias_branch, memhip_stream_t stream) {
  return memhip_layernorm_bwd_branch_map(dy, lddy, x, ldx, R, D, gamma, mean, rstd, dres, lddres, dgamma, dbeta, y_branch,
                                         ldyb, gamma_branch, rowmask, keep_prob, rows_per_sample, dy_branch, lddyb,
                                         dgamma_branch, dbias_branch, nullptr, nullptr, stream);
}

extern "C" int memhip_layerscale_grad(const void* W_bf16, int64_t ldw, const float* dW, int64_t lddw, const float* bias,
                                      const float* dbias, const float* gamma, int N, int K, float* dgamma,
                                      memhip_stream_t stream) {
  MEMHIP_REQUIRE(N >= 0 && K > 0 && K % 4 == 0 && ldw % 4 == 0 && lddw % 4 == 0, "layerscale_grad: bad shape");
  if (N == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(W_bf16 && dW && gamma && dgamma && (!bias == !dbias), "layerscale_grad: null pointer");
  hipLaunchKernelGGL(layerscale_grad_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), (const __bf16*)W_bf16,
                     (long long)ldw, dW, (long long)lddw, bias, dbias, gamma, N, K, dgamma);
  return check_launch("layerscale_grad");
}
