// fp32 implicit-GEMM convolutions for the EXACT-label mode of the frozen dVAE tokenizer forward
// (reference: eventvae/vae/vae_model.py:29-42,86-101,153-158).  The reference runs the tokenizer in fp32 --
// mem/engine_for_pretraining.py:140-145 sits outside the autocast block at :147 -- and its output is an INTEGER
// (argmax ids), so the labels must come from fp32 arithmetic: v_mfma_f32_16x16x4_f32 (f32 operands, f32
// accumulate, bitwise an fmaf chain over k; 64 FLOP/clk/SIMD = the fp32 vector peak, 157 TFLOP/s chip-wide).
// The bf16 kernels of conv.hip remain as the fast approximate mode.
//
// Same layout idea as conv.hip: activations fp32 NHWC with a one-pixel ZERO border ([B, H+2, W+2, C]), weights
// [C_out][ky][kx][c] K-contiguous, A gathered through per-row base addresses + a per-chunk tap offset (any
// C_in % 4 == 0: a 16-byte chunk never straddles a tap).  128x128x32 tiles, 4 waves of 64x64 (4x4 MFMA blocks),
// register-staged double buffer; LDS rows have a pitch of 34 floats so that the fragment reads (16 rows x 2 k per
// 32-lane group) touch 32 distinct banks.  Epilogue: + bias, ReLU, residual add, fp32 float4 stores.
#include "common.h"

namespace {

using namespace memhip;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kThreads = 256;
constexpr int PITCH = BK + 2;                       // floats; 8-byte aligned rows, bank(row, k) = 2 row + k
constexpr int kTileFloats = BM * PITCH;
constexpr int kStageFloats = 2 * kTileFloats;       // A tile + B tile

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct ConvArgsF32 {
  const float* in;      // [B, Hp, Wp, Cin] padded
  const float* w;       // [Cout, K]
  const float* bias;    // [Cout] or null
  const float* add;     // residual, laid out like `out`, or null
  float* out;
  int B, Hp, Wp, Cin, Ho, Wo, Cout, kw, stride, off, K;
  int out_padded, relu;
  const int* n_active;  // device: samples actually present (<= B), or null = B (certified tokenizer: the count is made on the device)
  int dyn_lo, dyn_hi;   // dynamic batch: this launch works only when dyn_lo <= *n_active < dyn_hi (the 32-row and the 128-row form
                        // are both launched; the count picks one on the device)
};

__global__ __launch_bounds__(kThreads, 2) void conv_gemm_f32_kernel(ConvArgsF32 p) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int nb = p.B;
  if (p.n_active) {                                  // the grid covers the capacity B; tiles behind the live samples leave at once
    const int na = *p.n_active;
    if (na < p.dyn_lo || na >= p.dyn_hi) return;
    nb = na < nb ? (na > 0 ? na : 0) : nb;
  }
  const int M = nb * p.Ho * p.Wo;
  const int ntn = (p.Cout + BN - 1) / BN;
  // Static batch: one tile per workgroup (XCD-aware bijective remap, n fastest).  Dynamic batch (n_active): the grid is a
  // fixed number of workgroups that walk the LIVE tiles -- a grid that covered the capacity cost ~1 us per empty workgroup
  // (37 000 of them per layer at a capacity of 128 samples: 20-30 ms per call, tools/tok_cert_probe.py).
  const int ntiles = ((M + BM - 1) / BM) * ntn;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  int pid = tile;
  if (!p.n_active) {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (pid / ntn) * BM, n0 = (pid % ntn) * BN;

  // global -> register staging: load j of this thread covers row (wave*4 + j)*8 + lane/8, chunk lane%8 (4 floats)
  long long abase[4], bbase[4];
  const int chunk = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + (lane >> 3);
    int m = m0 + row;
    m = m < M ? m : M - 1;
    const int hw = p.Ho * p.Wo;
    const int b = m / hw, r = m - b * hw;
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    abase[j] = (((long long)b * p.Hp + oy * p.stride + p.off) * p.Wp + ox * p.stride + p.off) * p.Cin;
    int n = n0 + row;
    n = n < p.Cout ? n : p.Cout - 1;
    bbase[j] = (long long)n * p.K;
  }
  float4 ra[4], rb[4];
  auto fetch = [&](int t) {
    const int k = t * BK + chunk * 4;
    const int tap = k / p.Cin, c0 = k - tap * p.Cin;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const long long koff = ((long long)ky * p.Wp + kx) * p.Cin + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = *reinterpret_cast<const float4*>(p.in + abase[j] + koff);
      rb[j] = *reinterpret_cast<const float4*>(p.w + bbase[j] + k);
    }
  };
  auto commit = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 8 + (lane >> 3);
      float2* a = reinterpret_cast<float2*>(dst + row * PITCH + chunk * 4);
      a[0] = make_float2(ra[j].x, ra[j].y);
      a[1] = make_float2(ra[j].z, ra[j].w);
      float2* b = reinterpret_cast<float2*>(dst + kTileFloats + row * PITCH + chunk * 4);
      b[0] = make_float2(rb[j].x, rb[j].y);
      b[1] = make_float2(rb[j].z, rb[j].w);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  fetch(0);
  commit(smem_f);
  __syncthreads();
  int cur = 0;
  const int frow = lane & 15, fk = lane >> 4;        // fragment: row = lane % 16, k = lane / 16
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
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) commit(smem_f + (cur ^ 1) * kStageFloats);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: the wave's 64x64 sub-tile through LDS, 32 rows at a time; lane -> (row, 4 columns)
  const int mw = m0 + wr * 64, nw = n0 + wc * 64;
  constexpr int LS = 68;
  float* wreg = smem_f + wave * (32 * LS);
  const int c4 = (lane & 15) * 4;
  const int n = nw + c4;
  float bias[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bias[k] = (p.bias && n + k < p.Cout) ? p.bias[n + k] : 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          wreg[(ii * 16 + (lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] = acc[half * 2 + ii][j][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4);
      const int m = mw + half * 32 + row;
      if (m >= M || n >= p.Cout) continue;
      long long mo = m;
      if (p.out_padded) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, r = m - b * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        mo = ((long long)b * (p.Ho + 2) + oy + 1) * (p.Wo + 2) + ox + 1;
      }
      const float4 v0 = *reinterpret_cast<const float4*>(wreg + row * LS + c4);
      float v[4] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3]};
      if (p.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      if (p.add) {                                    // ResBlock: net(x) + x
        const float4 xv = *reinterpret_cast<const float4*>(p.add + mo * p.Cout + n);
        v[0] += xv.x; v[1] += xv.y; v[2] += xv.z; v[3] += xv.w;
      }
      *reinterpret_cast<float4*>(p.out + mo * p.Cout + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();                                   // the epilogue's LDS slots are free before the next tile is staged
  }
}

// The same convolution on 32 x 128 tiles for the DYNAMIC-batch calls of the certified tokenizer (a handful of live samples):
// a 128-row tile is a whole K loop on one CU (0.65 ms at K = 6144 on the fp32 matrix pipe) however few tiles the layer has,
// so four samples cost as much as sixty-four.  Identical arithmetic per output element -- the same BK = 32 steps, the same
// v_mfma_f32_16x16x4_f32 chain in the same order, the same epilogue -- so the results equal conv_gemm_f32_kernel's bit for bit
// (tests/test_tokenizer_gpu.py: certified ids == fp32 ids with samples recomputed here).  4 waves side by side along N,
// each 32 rows x 32 columns; persistent over the live tiles.
constexpr int SBM = 32;
__global__ __launch_bounds__(kThreads, 4) void conv_gemm_f32_m32_kernel(ConvArgsF32 p) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  constexpr int kATile = SBM * PITCH, kStage = kATile + kTileFloats;       // A tile (32 rows) + B tile (128 rows)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int nb = p.B;
  if (p.n_active) {
    const int na = *p.n_active;
    if (na < p.dyn_lo || na >= p.dyn_hi) return;
    nb = na < nb ? (na > 0 ? na : 0) : nb;
  }
  const int M = nb * p.Ho * p.Wo;
  const int ntn = (p.Cout + BN - 1) / BN;
  const int ntiles = ((M + SBM - 1) / SBM) * ntn;
  const int chunk = lane & 7;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = (tile / ntn) * SBM, n0 = (tile % ntn) * BN;
    long long abase, bbase[4];
    {
      const int row = wave * 8 + (lane >> 3);
      int m = m0 + row;
      m = m < M ? m : M - 1;
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, r = m - b * hw;
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      abase = (((long long)b * p.Hp + oy * p.stride + p.off) * p.Wp + ox * p.stride + p.off) * p.Cin;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + (wave * 4 + j) * 8 + (lane >> 3);
      n = n < p.Cout ? n : p.Cout - 1;
      bbase[j] = (long long)n * p.K;
    }
    float4 ra, rb[4];
    auto fetch = [&](int t) {
      const int k = t * BK + chunk * 4;
      const int tap = k / p.Cin, c0 = k - tap * p.Cin;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const long long koff = ((long long)ky * p.Wp + kx) * p.Cin + c0;
      ra = *reinterpret_cast<const float4*>(p.in + abase + koff);
#pragma unroll
      for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const float4*>(p.w + bbase[j] + k);
    };
    auto commit = [&](float* dst) {
      {
        float2* a = reinterpret_cast<float2*>(dst + (wave * 8 + (lane >> 3)) * PITCH + chunk * 4);
        a[0] = make_float2(ra.x, ra.y);
        a[1] = make_float2(ra.z, ra.w);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        float2* b = reinterpret_cast<float2*>(dst + kATile + row * PITCH + chunk * 4);
        b[0] = make_float2(rb[j].x, rb[j].y);
        b[1] = make_float2(rb[j].z, rb[j].w);
      }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / BK;
    fetch(0);
    commit(smem_f);
    __syncthreads();
    int cur = 0;
    const int frow = lane & 15, fk = lane >> 4;
    for (int t = 0; t < nk; ++t) {
      if (t + 1 < nk) fetch(t + 1);
      const float* At = smem_f + cur * kStage + frow * PITCH + fk;
      const float* Bt = smem_f + cur * kStage + kATile + (wave * 32 + frow) * PITCH + fk;
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = At[i * 16 * PITCH + kk * 4];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = Bt[j * 16 * PITCH + kk * 4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      if (t + 1 < nk) commit(smem_f + (cur ^ 1) * kStage);
      __syncthreads();
      cur ^= 1;
    }
    // ---- epilogue: the wave's 32 x 32 sub-tile through LDS; lane -> (row, 4 columns), 8 lanes per row
    constexpr int LS = 36;
    float* wreg = smem_f + wave * (32 * LS);
    const int nw = n0 + wave * 32;
    const int c4 = (lane & 7) * 4;
    const int n = nw + c4;
    float bias[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bias[k] = (p.bias && n + k < p.Cout) ? p.bias[n + k] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) wreg[(i * 16 + (lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3);
      const int m = m0 + row;
      if (m >= M || n >= p.Cout) continue;
      long long mo = m;
      if (p.out_padded) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, r = m - b * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        mo = ((long long)b * (p.Ho + 2) + oy + 1) * (p.Wo + 2) + ox + 1;
      }
      const float4 v0 = *reinterpret_cast<const float4*>(wreg + row * LS + c4);
      float v[4] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3]};
      if (p.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      if (p.add) {
        const float4 xv = *reinterpret_cast<const float4*>(p.add + mo * p.Cout + n);
        v[0] += xv.x; v[1] += xv.y; v[2] += xv.z; v[3] += xv.w;
      }
      *reinterpret_cast<float4*>(p.out + mo * p.Cout + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
  }
}

// images f32 NCHW [B, C<=4, H, W] -> fp32 padded NHWC4 interior, optional (x - mean) / std
__global__ __launch_bounds__(256) void nchw_to_padded_nhwc4_f32_kernel(const float* __restrict__ x, int B, int C, int H,
                                                                       int W, const float* __restrict__ mean,
                                                                       const float* __restrict__ stdv,
                                                                       float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * H * W) return;
  const int xw = (int)(i % W);
  const long long t = i / W;
  const int y = (int)(t % H), b = (int)(t / H);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    float u = x[(((long long)b * C + c) * H + y) * W + xw];
    if (mean) u = (u - mean[c]) / stdv[c];
    v[c] = u;
  }
  *reinterpret_cast<float4*>(out + (((long long)b * (H + 2) + y + 1) * (W + 2) + xw + 1) * 4) =
      make_float4(v[0], v[1], v[2], v[3]);
}

// torch.argmax's order: a NaN is greater than every number, ties (and several NaNs) go to the smallest index; so a row of
// all -inf gives 0 and a row with a NaN gives the first NaN's index -- always a valid id (it is used as a label next).
__device__ __forceinline__ bool argmax_better(float a, int ai, float b, int bi) {
  const bool an = a != a, bn = b != b;
  if (an || bn) return an && (!bn || ai < bi);
  return a > b || (a == b && ai < bi);
}

// ids[m] = argmax_n logits[m, n] (first maximum, NaN wins like torch.argmax), one wave per row
// (rms, optional: sqrt(mean_n logits[m, n]^2), the scale of the row the certified tokenizer compares the gap with;
//  rows_dyn, optional device int: only the first min(M, *rows_dyn) rows exist)
__global__ __launch_bounds__(256) void argmax_rows_f32_kernel(const float* __restrict__ logits, long long ld, int M, int N,
                                                              long long* __restrict__ ids, float* __restrict__ gap,
                                                              float* __restrict__ rms, const int* __restrict__ samples_dyn,
                                                              int rows_per_sample) {
  const int lane = threadIdx.x & 63;
  if (samples_dyn) {
    const long long live = (long long)(*samples_dyn) * rows_per_sample;
    M = live < M ? (int)live : M;
  }
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
  float best = -INFINITY, second = -INFINITY;
  float sq = 0.f;
  int bi = 0x7fffffff;
  for (int n = lane * 4; n < N; n += 256) {
    const float4 v = *reinterpret_cast<const float4*>(logits + (long long)m * ld + n);
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sq = fmaf(f[k], f[k], sq);
      if (argmax_better(f[k], n + k, best, bi)) { second = best; best = f[k]; bi = n + k; }
      else if (f[k] > second) second = f[k];
    }
  }
  if (rms) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o), os = __shfl_xor(second, o);
    const int oi = __shfl_xor(bi, o);
    if (argmax_better(ob, oi, best, bi)) {
      second = fmaxf(best, os);                       // the displaced maximum may be the runner-up (ties: gap 0)
      best = ob; bi = oi;
    } else {
      second = fmaxf(second, ob);
    }
  }
  if (lane == 0) {
    ids[m] = bi;
    if (gap) gap[m] = best - second;
    if (rms) rms[m] = sqrtf(sq / (float)N);
  }
  }
}

// ---- certified split-precision tokenizer (mem_amd/vae_model.py, precision "fp16x2"): which samples hold a token whose
// fp16x2 top-2 gap is NOT above kappa x the row's rms (the proven-safe margin: there the fp32 argmax may differ)?  One
// workgroup; list = the flagged sample indices in ascending order, count[0] = how many, stats[0] += count, stats[1] += 1.
// A NaN gap or rms flags its sample (the comparison is written so that NaN fails it).
__global__ __launch_bounds__(1024) void tok_flag_samples_kernel(const float* __restrict__ gap, const float* __restrict__ rms,
                                                                int B, int hw, float kappa, int* __restrict__ list,
                                                                int* __restrict__ count, long long* __restrict__ stats) {
  // (round 6: the margins of a group of samples are swept by all 1024 threads with independent, coalesced loads -- a wave that
  // decided one sample after the other made 64 dependent round trips to memory: 110 us for 256 x 196 tokens, now ~10)
  constexpr int kGroup = 1024;                 // samples per group (one flag word each)
  __shared__ int s_flag[kGroup];
  __shared__ int s_wcnt[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  for (int b0 = 0; b0 < B; b0 += kGroup) {
    const int nb = B - b0 < kGroup ? B - b0 : kGroup;
    s_flag[tid] = 0;
    __syncthreads();
    const long long first = (long long)b0 * hw, total = (long long)nb * hw;
    for (long long i = tid; i < total; i += 4 * kGroup) {
      float g[4], r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long j = i + (long long)u * kGroup;
        g[u] = j < total ? gap[first + j] : 1.f;
        r[u] = j < total ? rms[first + j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long j = i + (long long)u * kGroup;
        if (j < total && !(g[u] > kappa * r[u])) s_flag[(int)(j / hw)] = 1;       // (NaN fails the comparison: flagged)
      }
    }
    __syncthreads();
    // ordered compaction of the group's flags
    const int f = tid < nb ? s_flag[tid] : 0;
    const unsigned long long bal = __ballot(f);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += s_wcnt[w];
    if (f) list[before + __popcll(bal & ((1ull << lane) - 1ull))] = b0 + tid;
    __syncthreads();
    if (tid == 0) {
      int add = 0;
      for (int w = 0; w < 16; ++w) add += s_wcnt[w];
      s_base += add;
    }
    __syncthreads();
  }
  if (tid == 0) {
    count[0] = s_base;
    if (stats) { stats[0] += s_base; stats[1] += 1; }
  }
}

// images f32 NCHW [*, C<=4, H, W] -> fp32 padded NHWC4 interior of slot j, for the samples list[off + j], j < min(R, count - off)
__global__ __launch_bounds__(256) void gather_nchw_to_padded_nhwc4_f32_kernel(const float* __restrict__ x, int C, int H, int W,
                                                                              const float* __restrict__ mean,
                                                                              const float* __restrict__ stdv,
                                                                              const int* __restrict__ list,
                                                                              const int* __restrict__ count, int off, int R,
                                                                              float* __restrict__ out, int* __restrict__ n_round) {
  int live = *count - off;
  live = live < 0 ? 0 : (live > R ? R : live);
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_round = live;
  const long long total = (long long)live * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {   // live pixels only
    const int xw = (int)(i % W);
    const long long t = i / W;
    const int y = (int)(t % H), j = (int)(t / H);
    const int b = list[off + j];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      float u = x[(((long long)b * C + c) * H + y) * W + xw];
      if (mean) u = (u - mean[c]) / stdv[c];
      v[c] = u;
    }
    *reinterpret_cast<float4*>(out + (((long long)j * (H + 2) + y + 1) * (W + 2) + xw + 1) * 4) =
        make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ids_out[list[off + j] * hw + t] = ids_in[j * hw + t] for the live slots of the round
__global__ __launch_bounds__(256) void scatter_ids_kernel(const long long* __restrict__ ids_in, const int* __restrict__ list,
                                                          const int* __restrict__ n_round, int off, int hw,
                                                          long long* __restrict__ ids_out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)(*n_round) * hw) return;
  const int j = (int)(i / hw), t = (int)(i - (long long)j * hw);
  ids_out[(long long)list[off + j] * hw + t] = ids_in[i];
}

}  // namespace

static int conv2d_nhwc_f32_impl(const float* in, const float* weight, const float* bias, const float* add, float* out,
                                int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int relu,
                                int out_padded, const int* n_active, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv2d_f32: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && weight && out, "conv2d_f32: null pointer");
  MEMHIP_REQUIRE(ksize >= 1 && ksize <= 4 && stride >= 1 && pad >= 0 && pad <= 1,
                 "conv2d_f32: kernel size 1..4, padding 0 or 1 (one-pixel border layout)");
  MEMHIP_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "conv2d_f32: C_in and C_out must be multiples of 4");
  ConvArgsF32 p;
  p.in = in; p.w = weight; p.bias = bias; p.add = add; p.out = out;
  p.B = B; p.Hp = H + 2; p.Wp = W + 2; p.Cin = Cin;
  p.Ho = (H + 2 * pad - ksize) / stride + 1; p.Wo = (W + 2 * pad - ksize) / stride + 1;
  p.Cout = Cout; p.kw = ksize; p.stride = stride; p.off = 1 - pad; p.K = ksize * ksize * Cin;
  p.out_padded = out_padded; p.relu = relu; p.n_active = n_active;
  p.dyn_lo = 0; p.dyn_hi = 1 << 30;
  MEMHIP_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d_f32: empty output");
  MEMHIP_REQUIRE(p.K % BK == 0, "conv2d_f32: K = %d must be a multiple of %d", p.K, BK);
  const long long M = (long long)B * p.Ho * p.Wo;
  MEMHIP_REQUIRE(M < (1LL << 31), "conv2d_f32: too many output pixels");
  const size_t lds = 2 * kStageFloats * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  if (n_active) {
    // dynamic batch: BOTH tile forms are launched with fixed grids of persistent workgroups; the device-side count selects
    // one (fewer than kDynSwitch live samples: 32-row tiles, a layer is otherwise one under-filled round of 128-row tiles;
    // from kDynSwitch on: the 128-row tiles at their better rate per FLOP).  The other launch returns at its first instruction.
    // (per layer: the 128-row form pays once its live tiles fill the chip's 2 x 256 workgroup slots -- 7 samples at the 56 x 56
    // level, 111 at the 14 x 14 level with 384 output channels)
    const int ntn_ = cdiv(Cout, BN), hw_ = p.Ho * p.Wo;
    int kDynSwitch = (int)((512LL * BM + (long long)hw_ * ntn_ - 1) / ((long long)hw_ * ntn_));
    kDynSwitch = kDynSwitch < 1 ? 1 : kDynSwitch;
    const size_t lds_s = (size_t)2 * (SBM * PITCH + kTileFloats) * sizeof(float);
    int grid_s = cdiv(M, SBM) * cdiv(Cout, BN);
    grid_s = grid_s > 2048 ? 2048 : grid_s;
    static bool attr_s = false;
    if (!attr_s) {
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_m32_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
      attr_s = true;
    }
    p.dyn_lo = 0; p.dyn_hi = kDynSwitch;
    hipLaunchKernelGGL(conv_gemm_f32_m32_kernel, dim3(grid_s), dim3(kThreads), lds_s, as_stream(stream), p);
    int grid_b = cdiv(M, BM) * cdiv(Cout, BN);
    grid_b = grid_b > 1024 ? 1024 : grid_b;
    p.dyn_lo = kDynSwitch; p.dyn_hi = 1 << 30;
    hipLaunchKernelGGL(conv_gemm_f32_kernel, dim3(grid_b), dim3(kThreads), lds, as_stream(stream), p);
    return check_launch("conv2d_nhwc_f32(dyn)");
  }
  const int grid = cdiv(M, BM) * cdiv(Cout, BN);
  hipLaunchKernelGGL(conv_gemm_f32_kernel, dim3(grid), dim3(kThreads), lds, as_stream(stream), p);
  return check_launch("conv2d_nhwc_f32");
}

extern "C" int memhip_conv2d_nhwc_f32(const float* in, const float* weight, const float* bias, const float* add, float* out,
                                      int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int relu,
                                      int out_padded, memhip_stream_t stream) {
  return conv2d_nhwc_f32_impl(in, weight, bias, add, out, B, H, W, Cin, Cout, ksize, stride, pad, relu, out_padded, nullptr, stream);
}

extern "C" int memhip_conv2d_nhwc_f32_dyn(const float* in, const float* weight, const float* bias, const float* add, float* out,
                                          int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int relu,
                                          int out_padded, const int32_t* n_active, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n_active, "conv2d_f32_dyn: null n_active");
  return conv2d_nhwc_f32_impl(in, weight, bias, add, out, B, H, W, Cin, Cout, ksize, stride, pad, relu, out_padded, n_active, stream);
}

extern "C" int memhip_nchw_to_padded_nhwc4_f32(const float* x, int B, int C, int H, int W, const float* mean,
                                               const float* stdv, float* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && C >= 1 && C <= 4 && H > 0 && W > 0, "nchw_to_padded_nhwc4_f32: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && out && (!mean == !stdv), "nchw_to_padded_nhwc4_f32: null pointer");
  const long long n = (long long)B * H * W;
  hipLaunchKernelGGL(nchw_to_padded_nhwc4_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     x, B, C, H, W, mean, stdv, out);
  return check_launch("nchw_to_padded_nhwc4_f32");
}

extern "C" int memhip_argmax_rows_f32(const float* logits, int64_t ld, int M, int N, int64_t* ids, float* top2_gap,
                                      memhip_stream_t stream) {
  MEMHIP_REQUIRE(M >= 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "argmax_rows_f32: N and ld must be multiples of 4");
  if (M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(logits && ids, "argmax_rows_f32: null pointer");
  hipLaunchKernelGGL(argmax_rows_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), logits, (long long)ld, M, N,
                     (long long*)ids, top2_gap, (float*)nullptr, (const int*)nullptr, 0);
  return check_launch("argmax_rows_f32");
}

extern "C" int memhip_argmax_rows_f32_ex(const float* logits, int64_t ld, int M, int N, int64_t* ids, float* top2_gap,
                                         float* row_rms, const int32_t* n_samples, int rows_per_sample,
                                         memhip_stream_t stream) {
  MEMHIP_REQUIRE(M >= 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "argmax_rows_f32_ex: N and ld must be multiples of 4");
  MEMHIP_REQUIRE(!n_samples || rows_per_sample > 0, "argmax_rows_f32_ex: rows_per_sample");
  if (M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(logits && ids, "argmax_rows_f32_ex: null pointer");
  const int blocks = (M + 3) / 4;
  hipLaunchKernelGGL(argmax_rows_f32_kernel, dim3(n_samples && blocks > 2048 ? 2048 : blocks), dim3(256), 0, as_stream(stream),
                     logits, (long long)ld, M, N, (long long*)ids, top2_gap, row_rms, (const int*)n_samples, rows_per_sample);
  return check_launch("argmax_rows_f32_ex");
}

extern "C" int memhip_tok_flag_samples(const float* top2_gap, const float* row_rms, int B, int tokens_per_sample, float kappa,
                                       int32_t* list, int32_t* count, int64_t* stats, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && tokens_per_sample > 0 && kappa >= 0.f, "tok_flag_samples: bad shape");
  MEMHIP_REQUIRE(top2_gap && row_rms && list && count, "tok_flag_samples: null pointer");
  hipLaunchKernelGGL(tok_flag_samples_kernel, dim3(1), dim3(1024), 0, as_stream(stream), top2_gap, row_rms, B, tokens_per_sample,
                     kappa, (int*)list, (int*)count, (long long*)stats);
  return check_launch("tok_flag_samples");
}

extern "C" int memhip_tok_gather_images_f32(const float* x, int C, int H, int W, const float* mean, const float* stdv,
                                            const int32_t* list, const int32_t* count, int offset, int R, float* out,
                                            int32_t* n_round, memhip_stream_t stream) {
  MEMHIP_REQUIRE(C >= 1 && C <= 4 && H > 0 && W > 0 && offset >= 0 && R > 0, "tok_gather_images_f32: bad shape");
  MEMHIP_REQUIRE(x && out && list && count && n_round && (!mean == !stdv), "tok_gather_images_f32: null pointer");
  const long long n = (long long)R * H * W;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(gather_nchw_to_padded_nhwc4_f32_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, as_stream(stream),
                     x, C, H, W, mean, stdv, (const int*)list, (const int*)count, offset, R, out, (int*)n_round);
  return check_launch("tok_gather_images_f32");
}

extern "C" int memhip_tok_scatter_ids(const int64_t* ids_in, const int32_t* list, const int32_t* n_round, int offset, int R,
                                      int tokens_per_sample, int64_t* ids_out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(offset >= 0 && R > 0 && tokens_per_sample > 0, "tok_scatter_ids: bad shape");
  MEMHIP_REQUIRE(ids_in && list && n_round && ids_out, "tok_scatter_ids: null pointer");
  const long long n = (long long)R * tokens_per_sample;
  hipLaunchKernelGGL(scatter_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (const long long*)ids_in, (const int*)list, (const int*)n_round, offset, tokens_per_sample, (long long*)ids_out);
  return check_launch("tok_scatter_ids");
}
