// fp32 PARITY MODE of the ViT path (`--precision fp32`): the same operations as the bf16 product path with fp32 operands,
// fp32 accumulation and NO bf16 rounding points -- the reference's arithmetic without autocast
// (mem/modeling_finetune.py:56-189, mem/modeling_pretrain.py:97-126, mem/engine_for_pretraining.py:152).  Purpose: loss
// curves that can be held against the reference's fp32 CPU curve to 1e-5 (north star: step-100 loss within 1e-4).
// Speed is secondary: GEMMs run on v_mfma_f32_16x16x4_f32 (64 FLOP/clk/SIMD = 1/16 of the bf16 rate), everything else is a
// plain wave-per-row kernel.  No fusion tricks of the fast path are used here (no gradients derived from other gradients).
#include "common.h"
#include "gemm_epilogue.hpp"

namespace {

using namespace memhip;

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float wsumf(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wmaxf(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_exact(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  return cdf + x * expf(-0.5f * x * x) * 0.39894228040143267794f;
}

// ---------------------------------------------------------------- GEMM  C[M,N] = A[M,K] B[N,K]^T, fp32
constexpr int BM = 128, BN = 128, BK = 32, kThreads = 256, PITCH = BK + 2;
constexpr int kTileFloats = BM * PITCH, kStageFloats = 2 * kTileFloats;

template <int EPI>
__device__ __forceinline__ void epi_f32(const GemmArgs& p, int m, int n, float acc) {
  const float* A0 = nullptr; (void)A0;
  const float b = p.bias ? p.bias[n] : 0.f;
  if constexpr (EPI == MEMHIP_EPI_BIAS_BF16) {                       // "BIAS": plain fp32 output
    float v = acc + b;
    if (n < p.colscale_n) v *= p.colscale;
    reinterpret_cast<float*>(p.out0)[(long long)m * p.ldo0 + n] = v;
    if (p.colsum) atomicAdd(p.colsum + n, v);
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU) {
    const float h = acc + b;
    reinterpret_cast<float*>(p.out0)[(long long)m * p.ldo0 + n] = h;
    reinterpret_cast<float*>(p.out1)[(long long)m * p.ldo1 + n] = gelu_exact(h);
  } else if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    const float y = acc + b;
    if (p.out0) reinterpret_cast<float*>(p.out0)[(long long)m * p.ldo0 + n] = y;
    float t = p.vec1 ? p.vec1[n] * y : y;
    if (p.rowmask) t = t / p.keep_prob * p.rowmask[(m + p.m_base) / p.rows_per_sample];
    const float xin = p.aux ? reinterpret_cast<const float*>(p.aux)[(long long)m * p.ldaux + n] : p.resid[(long long)m * p.ldr + n];
    p.resid[(long long)m * p.ldr + n] = xin + t;
  } else if constexpr (EPI == MEMHIP_EPI_DGELU) {
    const float h = reinterpret_cast<const float*>(p.aux)[(long long)m * p.ldaux + n];
    const float o = acc * gelu_grad_exact(h);
    reinterpret_cast<float*>(p.out0)[(long long)m * p.ldo0 + n] = o;
    if (p.colsum) atomicAdd(p.colsum + n, o);
  } else if constexpr (EPI == MEMHIP_EPI_F32) {
    float* o = reinterpret_cast<float*>(p.out0) + (long long)m * p.ldo0 + n;
    *o = p.accumulate ? (*o + acc) : acc;
  } else if constexpr (EPI == MEMHIP_EPI_PATCH_EMBED) {
    const float y = acc + b;
    const int L = p.rows_per_sample;
    const int bb = m / L, pi = m - bb * L;
    const float w = (float)reinterpret_cast<const unsigned char*>(p.aux)[m];
    p.resid[((long long)bb * (L + 1) + 1 + pi) * p.ldr + n] = y * (1.0f - w) + p.vec1[n] * w;
  }
}

template <int EPI>
__global__ __launch_bounds__(kThreads, 2) void gemm_f32_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = ((int)blockIdx.x / ntn) * BM, n0 = ((int)blockIdx.x % ntn) * BN;
  const int chunk = lane & 7;
  long long abase[4], bbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + (lane >> 3);
    int m = m0 + row; m = m < p.M ? m : p.M - 1;
    int n = n0 + row; n = n < p.N ? n : p.N - 1;
    abase[j] = (long long)m * p.lda;
    bbase[j] = (long long)n * p.ldb;
  }
  float4 ra[4], rb[4];
  auto fetch = [&](int t) {
    const int k = t * BK + chunk * 4;
    const bool in = k < p.K;                               // K % 4 == 0: a 16-byte chunk is inside or outside
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = in ? *reinterpret_cast<const float4*>(A + abase[j] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[j] = in ? *reinterpret_cast<const float4*>(B + bbase[j] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 8 + (lane >> 3);
      float2* a = reinterpret_cast<float2*>(dst + row * PITCH + chunk * 4);
      a[0] = make_float2(ra[j].x, ra[j].y); a[1] = make_float2(ra[j].z, ra[j].w);
      float2* b = reinterpret_cast<float2*>(dst + kTileFloats + row * PITCH + chunk * 4);
      b[0] = make_float2(rb[j].x, rb[j].y); b[1] = make_float2(rb[j].z, rb[j].w);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = (p.K + BK - 1) / BK;
  fetch(0);
  commit(smem_f);
  __syncthreads();
  int cur = 0;
  const int frow = lane & 15, fk = lane >> 4;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) fetch(t + 1);
    const float* At = smem_f + cur * kStageFloats + (wr * 64 + frow) * PITCH + fk;
    const float* Bt = smem_f + cur * kStageFloats + kTileFloats + (wc * 64 + frow) * PITCH + fk;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      float af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = At[i * 16 * PITCH + kk * 4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = Bt[j * 16 * PITCH + kk * 4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) commit(smem_f + (cur ^ 1) * kStageFloats);
    __syncthreads();
    cur ^= 1;
  }
  // accumulator element (i, j, r) of this lane: row = wr*64 + i*16 + (lane>>4)*4 + r, col = wc*64 + j*16 + (lane&15)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + (lane & 15);
        if (n < p.N) epi_f32<EPI>(p, m, n, acc[i][j][r]);
      }
    }
}

template <int EPI>
int launch_f32(const GemmArgs& p, hipStream_t s) {
  const size_t lds = 2 * kStageFloats * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_nt_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_f32: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
  hipLaunchKernelGGL(gemm_f32_nt_kernel<EPI>, dim3(grid), dim3(kThreads), lds, s, p);
  return check_launch("gemm_f32_nt");
}

// out [Cc, ldout] = in [R, Cc]^T, columns [R, ldout) zero-filled
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, long long ldin, int R, int Cc,
                                                            float* __restrict__ out, long long ldout) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < Cc) ? in[(long long)r * ldin + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < Cc && r < ldout) out[(long long)c * ldout + r] = tile[tx][k];
  }
}

// ---------------------------------------------------------------- LayerNorm (wave per row)
__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* __restrict__ x, long long ldx, const int* __restrict__ row_idx,
                                                         int R, int D, const float* __restrict__ g, const float* __restrict__ b,
                                                         float eps, float* __restrict__ y, long long ldy, float* __restrict__ mean,
                                                         float* __restrict__ rstd) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + (long long)(row_idx ? row_idx[r] : r) * ldx;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c];
  const float mu = wsumf(s) / (float)D;
  float v = 0.f;
  for (int c = lane; c < D; c += 64) { const float d = xr[c] - mu; v += d * d; }
  const float rs = 1.0f / sqrtf(wsumf(v) / (float)D + eps);
  for (int c = lane; c < D; c += 64) y[(long long)r * ldy + c] = (xr[c] - mu) * rs * g[c] + b[c];
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}

__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                         long long ldx, const int* __restrict__ row_idx, int R, int D,
                                                         const float* __restrict__ g, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ dres,
                                                         long long lddres, int accumulate, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const long long xrow = row_idx ? row_idx[r] : r;
  const float* xr = x + xrow * ldx;
  const float* dyr = dy + (long long)r * lddy;
  const float mu = mean[r], rs = rstd[r];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float xh = (xr[c] - mu) * rs, dg = dyr[c] * g[c];
    s1 += dg; s2 += dg * xh;
    atomicAdd(dgamma + c, dyr[c] * xh);
    atomicAdd(dbeta + c, dyr[c]);
  }
  s1 = wsumf(s1) / (float)D; s2 = wsumf(s2) / (float)D;
  float* o = dres + xrow * lddres;
  for (int c = lane; c < D; c += 64) {
    const float xh = (xr[c] - mu) * rs;
    const float dx = rs * (dyr[c] * g[c] - s1 - xh * s2);
    o[c] = accumulate ? o[c] + dx : dx;
  }
}

// backward of x = x + drop_path(gamma * y): dy = dt * gamma, dgamma += sum dt * y, dbias += sum dy
__global__ __launch_bounds__(256) void branch_bwd_f32_kernel(const float* __restrict__ dx, long long lddx, const float* __restrict__ y,
                                                             long long ldy, const float* __restrict__ gamma,
                                                             const float* __restrict__ rowmask, float keep_prob, int rows_per_sample,
                                                             int M, int D, float* __restrict__ dy, long long lddy,
                                                             float* __restrict__ dgamma, float* __restrict__ dbias) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= M) return;
  const float rm = rowmask ? rowmask[r / rows_per_sample] / keep_prob : 1.0f;
  for (int c = lane; c < D; c += 64) {
    const float dt = dx[(long long)r * lddx + c] * rm;
    const float d = gamma ? dt * gamma[c] : dt;
    dy[(long long)r * lddy + c] = d;
    if (dgamma && y) atomicAdd(dgamma + c, dt * y[(long long)r * ldy + c]);
    if (dbias) atomicAdd(dbias + c, d);
  }
}

__global__ __launch_bounds__(256) void embed_bwd_f32_kernel(const float* __restrict__ dx, long long lddx, const unsigned char* __restrict__ mask,
                                                            int B, int L, int D, float* __restrict__ dy, long long lddy,
                                                            float* __restrict__ dcls, float* __restrict__ dmask) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dcls + c, dx[(long long)b * (L + 1) * lddx + c]);
    float am = 0.f;
    for (int p = 0; p < L; ++p) {
      const float d = dx[((long long)b * (L + 1) + 1 + p) * lddx + c];
      const float w = (float)mask[(long long)b * L + p];
      am += d * w;
      dy[((long long)b * L + p) * lddy + c] = d * (1.0f - w);
    }
    atomicAdd(dmask + c, am);
  }
}

// nn.CrossEntropyLoss (mean) + argmax accuracy on fp32 logits; dlogits in place
__global__ __launch_bounds__(256) void ce_f32_kernel(float* __restrict__ logits, long long ld, const long long* __restrict__ labels,
                                                     int M, int V, float grad_scale, float* __restrict__ row_loss,
                                                     int* __restrict__ row_ok, int write_grad) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= M) return;
  float* l = logits + (long long)r * ld;
  float mx = -INFINITY; int am = 0x7fffffff;
  for (int c = lane; c < V; c += 64) if (l[c] > mx) { mx = l[c]; am = c; }
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o); const int oi = __shfl_xor(am, o);
    if (om > mx || (om == mx && oi < am)) { mx = om; am = oi; }
  }
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += expf(l[c] - mx);
  s = wsumf(s);
  const int lab = (int)labels[r];
  const float lse = logf(s) + mx;
  if (lane == 0) { row_loss[r] = lse - l[lab]; row_ok[r] = am == lab; }
  if (write_grad)
    for (int c = lane; c < V; c += 64) l[c] = (expf(l[c] - lse) - (c == lab ? 1.0f : 0.0f)) * grad_scale;
}
__global__ __launch_bounds__(256) void ce_reduce_f32_kernel(const float* __restrict__ row_loss, const int* __restrict__ row_ok, int M,
                                                            float* __restrict__ out2) {
  __shared__ double sl[256]; __shared__ int so[256];
  double s = 0.0; int k = 0;
  for (int i = threadIdx.x; i < M; i += 256) { s += row_loss[i]; k += row_ok[i]; }
  sl[threadIdx.x] = s; so[threadIdx.x] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0; int q = 0;
    for (int i = 0; i < 256; ++i) { t += sl[i]; q += so[i]; }
    out2[0] = (float)(t / M); out2[1] = (float)q / (float)M;
  }
}

__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ in, long long ld, int R, int Cc, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Cc) return;
  float s = 0.f;
  for (int r = blockIdx.y; r < R; r += gridDim.y) s += in[(long long)r * ld + c];
  atomicAdd(out + c, s);
}

__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ x, int B, int C, int H, int W, int ph, int pw,
                                                         float* __restrict__ out) {
  const int gh = H / ph, gw = W / pw, K = C * ph * pw;
  const long long total = (long long)B * gh * gw * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % K);
    const long long row = i / K;
    const int px = k % pw, py = (k / pw) % ph, c = k / (pw * ph);
    const int gx = (int)(row % gw), gy = (int)((row / gw) % gh), b = (int)(row / ((long long)gw * gh));
    out[i] = x[(((long long)b * C + c) * H + gy * ph + py) * W + gx * pw + px];
  }
}

// ---------------------------------------------------------------- attention, generic (fp32, any head_dim <= 128, T <= 256)
// qkv [B*T, 3D] columns [q*scale | k | v]; bias[h, i, j] = table[index[i*T + j], h] (index NULL = no bias); one wave per
// (b, h, query row); lane j owns keys j, j + 64, ...
constexpr int kMaxKeys = 4;     // T <= 256

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, long long ld, int B, int T, int D, int heads,
                                                           const float* __restrict__ table, const int* __restrict__ index,
                                                           float* __restrict__ out, long long ldo, float* __restrict__ lse) {
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= T) return;
  const float* q = qkv + ((long long)b * T + i) * ld + h * HD;
  float s[kMaxKeys];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < kMaxKeys; ++u) {
    const int j = lane + u * 64;
    s[u] = -INFINITY;
    if (j < T) {
      const float* k = qkv + ((long long)b * T + j) * ld + D + h * HD;
      float a = 0.f;
      for (int d = 0; d < HD; ++d) a += q[d] * k[d];
      if (index) a += table[(long long)index[i * T + j] * heads + h];
      s[u] = a;
      mx = fmaxf(mx, a);
    }
  }
  mx = wmaxf(mx);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < kMaxKeys; ++u) { s[u] = (lane + u * 64 < T) ? expf(s[u] - mx) : 0.f; sum += s[u]; }
  sum = wsumf(sum);
  const float inv = 1.0f / sum;
  for (int d = 0; d < HD; ++d) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < kMaxKeys; ++u) {
      const int j = lane + u * 64;
      if (j < T) a += s[u] * inv * qkv[((long long)b * T + j) * ld + 2 * D + h * HD + d];
    }
    a = wsumf(a);
    if (lane == 0) out[((long long)b * T + i) * ldo + h * HD + d] = a;
  }
  if (lane == 0 && lse) lse[((long long)b * heads + h) * T + i] = logf(sum) + mx;
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ qkv, long long ld, const float* __restrict__ dout,
                                                           long long ldo, int B, int T, int D, int heads, float scale,
                                                           const float* __restrict__ table, const int* __restrict__ index,
                                                           float* __restrict__ dqkv, long long ldd, float* __restrict__ dtable) {
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= T) return;
  const float* q = qkv + ((long long)b * T + i) * ld + h * HD;
  const float* dO = dout + ((long long)b * T + i) * ldo + h * HD;
  float p[kMaxKeys], dp[kMaxKeys];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < kMaxKeys; ++u) {
    const int j = lane + u * 64;
    p[u] = -INFINITY; dp[u] = 0.f;
    if (j < T) {
      const float* k = qkv + ((long long)b * T + j) * ld + D + h * HD;
      const float* v = qkv + ((long long)b * T + j) * ld + 2 * D + h * HD;
      float a = 0.f, g = 0.f;
      for (int d = 0; d < HD; ++d) { a += q[d] * k[d]; g += dO[d] * v[d]; }
      if (index) a += table[(long long)index[i * T + j] * heads + h];
      p[u] = a; dp[u] = g;
      mx = fmaxf(mx, a);
    }
  }
  mx = wmaxf(mx);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < kMaxKeys; ++u) { p[u] = (lane + u * 64 < T) ? expf(p[u] - mx) : 0.f; sum += p[u]; }
  sum = wsumf(sum);
  float delta = 0.f;
#pragma unroll
  for (int u = 0; u < kMaxKeys; ++u) { p[u] /= sum; delta += p[u] * dp[u]; }
  delta = wsumf(delta);                                  // sum_j P dP (softmax backward)
  float ds[kMaxKeys];
#pragma unroll
  for (int u = 0; u < kMaxKeys; ++u) {
    const int j = lane + u * 64;
    ds[u] = p[u] * (dp[u] - delta);
    if (j < T) {
      if (dtable && index) atomicAdd(dtable + (long long)index[i * T + j] * heads + h, ds[u]);
      float* dk = dqkv + ((long long)b * T + j) * ldd + D + h * HD;
      float* dv = dqkv + ((long long)b * T + j) * ldd + 2 * D + h * HD;
      for (int d = 0; d < HD; ++d) { atomicAdd(dk + d, ds[u] * q[d]); atomicAdd(dv + d, p[u] * dO[d]); }
    }
  }
  for (int d = 0; d < HD; ++d) {                          // dq (w.r.t. the unscaled q: the stored q is q * scale)
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < kMaxKeys; ++u) {
      const int j = lane + u * 64;
      if (j < T) a += ds[u] * qkv[((long long)b * T + j) * ld + D + h * HD + d];
    }
    a = wsumf(a);
    if (lane == 0) dqkv[((long long)b * T + i) * ldd + h * HD + d] = a * scale;
  }
}

}  // namespace

extern "C" int memhip_f32_gemm_nt(const memhip_gemm_args_t* a, memhip_stream_t stream) {
  MEMHIP_REQUIRE(a, "gemm_f32: null args");
  GemmArgs p;
  __builtin_memset(&p, 0, sizeof(p));
  __builtin_memcpy(&p, a, sizeof(memhip_gemm_args_t));
  MEMHIP_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0, "gemm_f32: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
  if (p.M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(p.A && p.B && p.K % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 && ((uintptr_t)p.A & 15) == 0 &&
                     ((uintptr_t)p.B & 15) == 0,
                 "gemm_f32: operands must be 16-byte aligned with K and ld multiples of 4");
  hipStream_t s = as_stream(stream);
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: MEMHIP_REQUIRE(p.out0, "gemm_f32: out0"); return launch_f32<MEMHIP_EPI_BIAS_BF16>(p, s);
    case MEMHIP_EPI_BIAS_GELU: MEMHIP_REQUIRE(p.out0 && p.out1, "gemm_f32: out0/out1"); return launch_f32<MEMHIP_EPI_BIAS_GELU>(p, s);
    case MEMHIP_EPI_RESIDUAL: MEMHIP_REQUIRE(p.resid, "gemm_f32: residual args"); return launch_f32<MEMHIP_EPI_RESIDUAL>(p, s);
    case MEMHIP_EPI_DGELU: MEMHIP_REQUIRE(p.out0 && p.aux, "gemm_f32: dgelu args"); return launch_f32<MEMHIP_EPI_DGELU>(p, s);
    case MEMHIP_EPI_F32: MEMHIP_REQUIRE(p.out0, "gemm_f32: out0"); return launch_f32<MEMHIP_EPI_F32>(p, s);
    case MEMHIP_EPI_PATCH_EMBED: MEMHIP_REQUIRE(p.resid && p.vec1 && p.aux, "gemm_f32: patch args"); return launch_f32<MEMHIP_EPI_PATCH_EMBED>(p, s);
    default: return fail(MEMHIP_EUNSUPPORTED, "gemm_f32: epilogue %d is not part of the fp32 path", p.epilogue);
  }
}

extern "C" int memhip_f32_transpose(const float* in, int64_t ldin, int R, int Cc, float* out, int64_t ldout, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && Cc > 0 && ldout >= R, "transpose_f32: bad shape");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out, "transpose_f32: null pointer");
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(cdiv(Cc, 32), cdiv(ldout, 32)), dim3(256), 0, as_stream(stream), in, (long long)ldin,
                     R, Cc, out, (long long)ldout);
  return check_launch("transpose_f32");
}

extern "C" int memhip_f32_layernorm_fwd(const float* x, int64_t ldx, const int32_t* row_idx, int R, int D, const float* gamma,
                                        const float* beta, float eps, float* y, int64_t ldy, float* mean, float* rstd,
                                        memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && D > 0, "ln_fwd_f32: bad shape");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && gamma && beta && y && mean && rstd, "ln_fwd_f32: null pointer");
  hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3(cdiv(R, 4)), dim3(256), 0, as_stream(stream), x, (long long)ldx, row_idx, R, D, gamma,
                     beta, eps, y, (long long)ldy, mean, rstd);
  return check_launch("ln_fwd_f32");
}

extern "C" int memhip_f32_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int32_t* row_idx, int R,
                                        int D, const float* gamma, const float* mean, const float* rstd, float* dres,
                                        int64_t lddres, int accumulate, float* dgamma, float* dbeta, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && D > 0, "ln_bwd_f32: bad shape");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dy && x && gamma && mean && rstd && dres && dgamma && dbeta, "ln_bwd_f32: null pointer");
  hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3(cdiv(R, 4)), dim3(256), 0, as_stream(stream), dy, (long long)lddy, x, (long long)ldx,
                     row_idx, R, D, gamma, mean, rstd, dres, (long long)lddres, accumulate, dgamma, dbeta);
  return check_launch("ln_bwd_f32");
}

extern "C" int memhip_f32_branch_bwd(const float* dx, int64_t lddx, const float* y, int64_t ldy, const float* gamma,
                                     const float* rowmask, float keep_prob, int rows_per_sample, int M, int D, float* dy,
                                     int64_t lddy, float* dgamma, float* dbias, memhip_stream_t stream) {
  MEMHIP_REQUIRE(M >= 0 && D > 0 && rows_per_sample > 0, "branch_bwd_f32: bad shape");
  if (M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dx && dy, "branch_bwd_f32: null pointer");
  hipLaunchKernelGGL(branch_bwd_f32_kernel, dim3(cdiv(M, 4)), dim3(256), 0, as_stream(stream), dx, (long long)lddx, y, (long long)ldy,
                     gamma, rowmask, keep_prob, rows_per_sample, M, D, dy, (long long)lddy, dgamma, dbias);
  return check_launch("branch_bwd_f32");
}

extern "C" int memhip_f32_embed_bwd(const float* dx, int64_t lddx, const uint8_t* mask, int B, int L, int D, float* dy,
                                    int64_t lddy, float* dcls, float* dmask_token, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && L > 0 && D > 0, "embed_bwd_f32: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dx && mask && dy && dcls && dmask_token, "embed_bwd_f32: null pointer");
  hipLaunchKernelGGL(embed_bwd_f32_kernel, dim3(B), dim3(256), 0, as_stream(stream), dx, (long long)lddx, mask, B, L, D, dy,
                     (long long)lddy, dcls, dmask_token);
  return check_launch("embed_bwd_f32");
}

extern "C" int memhip_f32_cross_entropy(float* logits, int64_t ld, const int64_t* labels, int M, int V, float grad_scale,
                                        float* row_loss, int32_t* row_correct, int write_grad, float* out2,
                                        memhip_stream_t stream) {
  MEMHIP_REQUIRE(M > 0 && V > 0, "ce_f32: bad shape");
  MEMHIP_REQUIRE(logits && labels && row_loss && row_correct && out2, "ce_f32: null pointer");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(ce_f32_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, logits, (long long)ld, (const long long*)labels, M, V,
                     grad_scale, row_loss, row_correct, write_grad);
  hipLaunchKernelGGL(ce_reduce_f32_kernel, dim3(1), dim3(256), 0, s, row_loss, row_correct, M, out2);
  return check_launch("ce_f32");
}

extern "C" int memhip_f32_colsum(const float* in, int64_t ld, int R, int C, float* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && C > 0, "colsum_f32: bad shape");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out, "colsum_f32: null pointer");
  const int gy = R < 64 ? 1 : 64;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(cdiv(C, 256), gy), dim3(256), 0, as_stream(stream), in, (long long)ld, R, C, out);
  return check_launch("colsum_f32");
}

extern "C" int memhip_f32_im2col(const float* x, int B, int C, int H, int W, int ph, int pw, float* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && C > 0 && H % ph == 0 && W % pw == 0, "im2col_f32: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && out, "im2col_f32: null pointer");
  hipLaunchKernelGGL(im2col_f32_kernel, dim3(2048), dim3(256), 0, as_stream(stream), x, B, C, H, W, ph, pw, out);
  return check_launch("im2col_f32");
}

extern "C" int memhip_f32_attn_fwd(const float* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table,
                                   const int32_t* index, float* out, int64_t ldo, float* lse, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && T <= 64 * kMaxKeys && heads > 0 && D % heads == 0, "attn_f32: bad shape (T <= 256)");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && out && (!index || table), "attn_f32: null pointer");
  const int hd = D / heads;
  dim3 grid(B * heads, cdiv(T, 4));
  hipStream_t s = as_stream(stream);
  if (hd == 64) hipLaunchKernelGGL(attn_fwd_f32_kernel<64>, grid, dim3(256), 0, s, qkv, (long long)ldqkv, B, T, D, heads, table, index, out, (long long)ldo, lse);
  else if (hd == 32) hipLaunchKernelGGL(attn_fwd_f32_kernel<32>, grid, dim3(256), 0, s, qkv, (long long)ldqkv, B, T, D, heads, table, index, out, (long long)ldo, lse);
  else return fail(MEMHIP_EUNSUPPORTED, "attn_f32: head_dim %d (32 or 64)", hd);
  return check_launch("attn_fwd_f32");
}

extern "C" int memhip_f32_attn_bwd(const float* qkv, int64_t ldqkv, const float* dout, int64_t ldo, int B, int T, int D, int heads,
                                   float scale, const float* table, const int32_t* index, float* dqkv, int64_t lddqkv,
                                   float* dtable, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && T <= 64 * kMaxKeys && heads > 0 && D % heads == 0, "attn_bwd_f32: bad shape (T <= 256)");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && dout && dqkv && (!index || table), "attn_bwd_f32: null pointer");
  const int hd = D / heads;
  dim3 grid(B * heads, cdiv(T, 4));
  hipStream_t s = as_stream(stream);
  // dk / dv are accumulated with atomics over the query rows: zero the gradient first
  MEMHIP_HIP(hipMemset2DAsync(dqkv, (size_t)lddqkv * 4, 0, (size_t)3 * D * 4, (size_t)B * T, s));
  if (hd == 64) hipLaunchKernelGGL(attn_bwd_f32_kernel<64>, grid, dim3(256), 0, s, qkv, (long long)ldqkv, dout, (long long)ldo, B, T, D, heads, scale, table, index, dqkv, (long long)lddqkv, dtable);
  else if (hd == 32) hipLaunchKernelGGL(attn_bwd_f32_kernel<32>, grid, dim3(256), 0, s, qkv, (long long)ldqkv, dout, (long long)ldo, B, T, D, heads, scale, table, index, dqkv, (long long)lddqkv, dtable);
  else return fail(MEMHIP_EUNSUPPORTED, "attn_bwd_f32: head_dim %d (32 or 64)", hd);
  return check_launch("attn_bwd_f32");
}

// ---------------------------------------------------------------- MAE token plumbing + loss (mem/modeling_mae.py:204-292)
namespace {

// encoder input: row (b, 0) = cls + pos[0]; row (b, 1 + j) = xe[b, ids_keep[b, j]] + pos[1 + ids_keep[b, j]]
__global__ __launch_bounds__(256) void mae_enc_assemble_kernel(const float* __restrict__ xe, const float* __restrict__ pos,
                                                               const float* __restrict__ cls, const long long* __restrict__ ids_keep,
                                                               int B, int L, int K, int D, float* __restrict__ out) {
  const int row = blockIdx.x;                       // b * (K + 1) + t
  const int b = row / (K + 1), t = row - b * (K + 1);
  float* o = out + (long long)row * D;
  if (t == 0) {
    for (int c = threadIdx.x; c < D; c += 256) o[c] = cls[c] + pos[c];
  } else {
    const long long l = ids_keep[(long long)b * K + t - 1];
    const float* s = xe + ((long long)b * L + l) * D;
    const float* p = pos + (1 + l) * D;
    for (int c = threadIdx.x; c < D; c += 256) o[c] = s[c] + p[c];
  }
}
// backward: dxe[b, ids_keep[b, j]] = dx[b, 1 + j] (dxe pre-zeroed), dcls += dx[b, 0]
__global__ __launch_bounds__(256) void mae_enc_assemble_bwd_kernel(const float* __restrict__ dx, const long long* __restrict__ ids_keep,
                                                                   int B, int L, int K, int D, float* __restrict__ dxe,
                                                                   float* __restrict__ dcls) {
  const int row = blockIdx.x;
  const int b = row / (K + 1), t = row - b * (K + 1);
  const float* g = dx + (long long)row * D;
  if (t == 0) {
    for (int c = threadIdx.x; c < D; c += 256) atomicAdd(dcls + c, g[c]);
  } else {
    float* o = dxe + ((long long)b * L + ids_keep[(long long)b * K + t - 1]) * D;
    for (int c = threadIdx.x; c < D; c += 256) o[c] = g[c];
  }
}
// decoder input: row (b, 0) = y[b, 0] + dpos[0]; row (b, 1 + l) = (r < K ? y[b, 1 + r] : mask_token) + dpos[1 + l], r = ids_restore[b, l]
__global__ __launch_bounds__(256) void mae_dec_assemble_kernel(const float* __restrict__ y, const float* __restrict__ mask_token,
                                                               const float* __restrict__ dpos, const long long* __restrict__ ids_restore,
                                                               int B, int L, int K, int D, float* __restrict__ out) {
  const int row = blockIdx.x;                       // b * (L + 1) + t
  const int b = row / (L + 1), t = row - b * (L + 1);
  float* o = out + (long long)row * D;
  const float* p = dpos + (long long)t * D;
  const float* s;
  if (t == 0) s = y + (long long)b * (K + 1) * D;
  else {
    const long long r = ids_restore[(long long)b * L + t - 1];
    s = r < K ? y + ((long long)b * (K + 1) + 1 + r) * D : mask_token;
  }
  for (int c = threadIdx.x; c < D; c += 256) o[c] = s[c] + p[c];
}
// one workgroup per (sample, slice of its token rows): kept rows are copied, the rows of the mask token are summed in
// registers and leave as ONE atomic per column and workgroup (one atomic per row and column was 25 000 adds on each of
// the 512 addresses at B = 256: 0.6 ms)
__global__ __launch_bounds__(256) void mae_dec_assemble_bwd_kernel(const float* __restrict__ dxd, const long long* __restrict__ ids_restore,
                                                                   int B, int L, int K, int D, float* __restrict__ dy,
                                                                   float* __restrict__ dmask_token) {
  const int b = blockIdx.x, nsl = gridDim.y, sl = blockIdx.y;
  const int t0 = (int)((long long)(L + 1) * sl / nsl), t1 = (int)((long long)(L + 1) * (sl + 1) / nsl);
  for (int c0 = 0; c0 < D; c0 += 1024) {                     // 4 columns per thread and sweep
    const int c = c0 + threadIdx.x * 4;
    if (c >= D) continue;                                    // (D % 4 == 0)
    float4 acc{0.f, 0.f, 0.f, 0.f};
    for (int t = t0; t < t1; ++t) {
      const float4 g = *reinterpret_cast<const float4*>(dxd + ((long long)b * (L + 1) + t) * D + c);
      long long r = -1;                                      // cls row
      if (t > 0) r = ids_restore[(long long)b * L + t - 1];
      if (r < K) *reinterpret_cast<float4*>(dy + ((long long)b * (K + 1) + 1 + r) * D + c) = g;
      else { acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w; }
    }
    atomicAdd(dmask_token + c, acc.x); atomicAdd(dmask_token + c + 1, acc.y);
    atomicAdd(dmask_token + c + 2, acc.z); atomicAdd(dmask_token + c + 3, acc.w);
  }
}
// loss: pred [B*(L+1), P] (row (b,0) = cls, ignored), target = patchify(imgs) ('nchpwq->nhwpqc'); per-patch mean squared
// error; only_masked: sum(loss * mask) / sum(mask), else sum over all patches.  dpred written in place of nothing: separate buffer.
__global__ __launch_bounds__(256) void mae_loss_kernel(const float* __restrict__ pred, const float* __restrict__ img,
                                                       const float* __restrict__ mask, int B, int C, int H, int W, int p,
                                                       int only_masked, const float* __restrict__ mask_sum,
                                                       float* __restrict__ row_loss, float* __restrict__ dpred) {
  const int gw = W / p, L = (H / p) * gw, P = p * p * C;
  const int row = blockIdx.x;                       // b * (L + 1) + t
  const int b = row / (L + 1), t = row - b * (L + 1);
  float* d = dpred + (long long)row * P;
  if (t == 0) {
    for (int k = threadIdx.x; k < P; k += 256) d[k] = 0.f;
    return;
  }
  const int l = t - 1, gy = l / gw, gx = l - gy * gw;
  const float m = mask[(long long)b * L + l];
  const float wgt = only_masked ? m / mask_sum[0] : 1.0f;
  const float* pr = pred + (long long)row * P;
  float s = 0.f;
  for (int k = threadIdx.x; k < P; k += 256) {
    const int c = k % C, q = (k / C) % p, py = k / (C * p);
    const float tg = img[(((long long)b * C + c) * H + gy * p + py) * W + gx * p + q];
    const float e = pr[k] - tg;
    s += e * e;
    d[k] = 2.0f * e / (float)P * wgt;
  }
  __shared__ float sh[4];
  s = wsumf(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) row_loss[(long long)b * L + l] = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)P * wgt;
}
__global__ __launch_bounds__(256) void sum_f32_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) s += x[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < 256; ++i) t += sh[i]; out[0] = (float)t; }
}

}  // namespace

extern "C" int memhip_mae_enc_assemble(const float* xe, const float* pos, const float* cls, const int64_t* ids_keep, int B, int L,
                                       int K, int D, float* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B > 0 && L > 0 && K > 0 && K <= L && D > 0 && xe && pos && cls && ids_keep && out, "mae_enc_assemble: bad arguments");
  hipLaunchKernelGGL(mae_enc_assemble_kernel, dim3(B * (K + 1)), dim3(256), 0, as_stream(stream), xe, pos, cls,
                     (const long long*)ids_keep, B, L, K, D, out);
  return check_launch("mae_enc_assemble");
}
extern "C" int memhip_mae_enc_assemble_bwd(const float* dx, const int64_t* ids_keep, int B, int L, int K, int D, float* dxe,
                                           float* dcls, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B > 0 && L > 0 && K > 0 && D > 0 && dx && ids_keep && dxe && dcls, "mae_enc_assemble_bwd: bad arguments");
  hipStream_t s = as_stream(stream);
  MEMHIP_HIP(hipMemsetAsync(dxe, 0, (size_t)B * L * D * sizeof(float), s));
  hipLaunchKernelGGL(mae_enc_assemble_bwd_kernel, dim3(B * (K + 1)), dim3(256), 0, s, dx, (const long long*)ids_keep, B, L, K, D,
                     dxe, dcls);
  return check_launch("mae_enc_assemble_bwd");
}
extern "C" int memhip_mae_dec_assemble(const float* y, const float* mask_token, const float* dpos, const int64_t* ids_restore, int B,
                                       int L, int K, int D, float* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B > 0 && L > 0 && K > 0 && D > 0 && y && mask_token && dpos && ids_restore && out, "mae_dec_assemble: bad arguments");
  hipLaunchKernelGGL(mae_dec_assemble_kernel, dim3(B * (L + 1)), dim3(256), 0, as_stream(stream), y, mask_token, dpos,
                     (const long long*)ids_restore, B, L, K, D, out);
  return check_launch("mae_dec_assemble");
}
extern "C" int memhip_mae_dec_assemble_bwd(const float* dxd, const int64_t* ids_restore, int B, int L, int K, int D, float* dy,
                                           float* dmask_token, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B > 0 && L > 0 && K > 0 && D > 0 && D % 4 == 0 && dxd && ids_restore && dy && dmask_token,
                 "mae_dec_assemble_bwd: bad arguments");
  const int slices = B >= 1024 ? 1 : (1024 + B - 1) / B > L + 1 ? L + 1 : (1024 + B - 1) / B;   // ~1024 workgroups
  hipLaunchKernelGGL(mae_dec_assemble_bwd_kernel, dim3(B, slices), dim3(256), 0, as_stream(stream), dxd,
                     (const long long*)ids_restore, B, L, K, D, dy, dmask_token);
  return check_launch("mae_dec_assemble_bwd");
}
extern "C" int memhip_mae_loss(const float* pred, const float* img, const float* mask, int B, int C, int H, int W, int patch,
                               int only_masked, float* row_loss, float* dpred, float* scratch2, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B > 0 && C > 0 && H % patch == 0 && W % patch == 0 && pred && img && mask && row_loss && dpred && scratch2,
                 "mae_loss: bad arguments");
  hipStream_t s = as_stream(stream);
  const int L = (H / patch) * (W / patch);
  hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(256), 0, s, mask, (long long)B * L, scratch2);          // sum(mask)
  hipLaunchKernelGGL(mae_loss_kernel, dim3(B * (L + 1)), dim3(256), 0, s, pred, img, mask, B, C, H, W, patch, only_masked,
                     scratch2, row_loss, dpred);
  hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(256), 0, s, row_loss, (long long)B * L, scratch2 + 1);  // the loss
  return check_launch("mae_loss");
}
