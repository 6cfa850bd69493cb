// Implicit-GEMM convolutions + argmax for the frozen dVAE tokenizer forward
// (reference: eventvae/vae/vae_model.py:29-42 ResBlock, :86-101 encoder stack, :153-158
// get_codebook_indices; called every pretraining step at mem/engine_for_pretraining.py:144).
//
// Layout: every activation is bf16 NHWC with a one-pixel ZERO border ([B, H+2, W+2, C]); all three
// convolution shapes of the encoder (4x4 stride 2 pad 1, 3x3 stride 1 pad 1, 1x1) then read only
// in-bounds addresses and never branch on padding, and a GEMM row (b, oy, ox) of the output is the
// next layer's input pixel.  The convolution is the 128x128x64 MFMA GEMM of gemm.hip whose A-operand
// tile is GATHERED by the LDS-DMA source addresses:
//     A[m = (b,oy,ox)][k = (ky,kx,c)] = in[b][oy*s + off + ky][ox*s + off + kx][c],  off = 1 - pad
// -- with C_in a multiple of 64 a 64-deep K-tile lies inside one tap, so a lane's address is its row's
// base + a wave-uniform tap offset; weights are pre-packed [C_out][ky][kx][c] (K-contiguous).  The
// first layer (C_in = 3, stored as 4) is one K-tile: a 16-byte chunk = two horizontally adjacent
// pixels x 4 channels.  Epilogue (16-byte accesses after an LDS transposition): + bias, optional
// ReLU, optional residual add (ResBlock: net(x) + x), bf16, written into the interior of the padded
// output (or densely, for the token logits).
#include "common.h"

namespace {

using namespace memhip;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kTileBytes = BM * BK * 2;
constexpr int kStageBytes = 2 * kTileBytes;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct ConvArgs {
  const __bf16* in;     // [B, Hp, Wp, Cin] padded
  const __bf16* w;      // [Cout, K]
  const float* bias;    // [Cout] or null
  const __bf16* add;    // residual, laid out like `out`, or null
  __bf16* out;
  int B, Hp, Wp, Cin, Ho, Wo, Cout, kh, kw, stride, off, K;
  int out_padded, relu, cin4;
};

__device__ __forceinline__ int swz_slot(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ f32x2 unpack2(unsigned u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }

__global__ __launch_bounds__(kThreads, 2) void conv_gemm_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int M = p.B * p.Ho * p.Wo;
  // XCD-aware bijective remap of the block id, n fastest (an A row panel stays in one XCD's L2)
  const int nwg = gridDim.x;
  int pid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntn = (p.Cout + BN - 1) / BN;
  const int m0 = (pid / ntn) * BM, n0 = (pid % ntn) * BN;

  // ---- per-lane constants of the LDS-DMA issue: instruction j of this wave covers 8 rows
  long long abase[4];      // element offset of this lane's A row (top-left tap, channel 0)
  long long bbase[4];      // element offset of this lane's weight row
  int chunkg[4];           // global 16-byte chunk that lands at this lane's LDS position
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int inst = wave * 4 + j;
    const int row = inst * 8 + (lane >> 3);
    chunkg[j] = (lane & 7) ^ ((row >> 1) & 7);
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
  auto stage = [&](int t, char* dst) {
    const int k0 = t * BK;
    long long koff = 0;
    if (!p.cin4) {
      const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      koff = ((long long)ky * p.Wp + kx) * p.Cin + c0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int inst = wave * 4 + j;
      // first layer: chunk c = two adjacent pixels (kx = 2(c&1), 2(c&1)+1) of tap row ky = c >> 1
      const long long ao = p.cin4 ? ((long long)(chunkg[j] >> 1) * p.Wp + 2 * (chunkg[j] & 1)) * 4
                                  : koff + chunkg[j] * 8;
      glds16(p.in + abase[j] + ao, dst + inst * 1024);
      glds16(p.w + bbase[j] + k0 + chunkg[j] * 8, dst + kTileBytes + inst * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, smem);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) stage(t + 1, smem + (cur ^ 1) * kStageBytes);
    const char* At = smem + cur * kStageBytes;
    const char* Bt = At + kTileBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
      const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(At + swz_slot(wr * 64 + i * 16 + (lane & 15), chunk) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bt + swz_slot(wc * 64 + j * 16 + (lane & 15), chunk) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: transpose the wave's 64x64 sub-tile through LDS, 32 rows at a time
  const int mw = m0 + wr * 64, nw = n0 + wc * 64;
  constexpr int LS = 72;
  float* wreg = reinterpret_cast<float*>(smem + wave * 16384);
  const int c8 = (lane & 7) * 8;
  const int n = nw + c8;
  float bias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias[k] = (p.bias && n + k < p.Cout) ? p.bias[n + k] : 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          wreg[(ii * 16 + (lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] = acc[half * 2 + ii][j][r];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3);
      const int m = mw + half * 32 + row;
      if (m >= M || n >= p.Cout) continue;
      long long mo = m;
      if (p.out_padded) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, r = m - b * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        mo = ((long long)b * (p.Ho + 2) + oy + 1) * (p.Wo + 2) + ox + 1;
      }
      const float4 v0 = *reinterpret_cast<const float4*>(wreg + row * LS + c8);
      const float4 v1 = *reinterpret_cast<const float4*>(wreg + row * LS + c8 + 4);
      float v[8] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3],
                    v1.x + bias[4], v1.y + bias[5], v1.z + bias[6], v1.w + bias[7]};
      if (p.relu) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      unsigned y[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = pack2(v[2 * k], v[2 * k + 1]);
      if (p.add) {                                      // ResBlock: conv output (bf16) + x (bf16) -> bf16
        const uint4 xv = *reinterpret_cast<const uint4*>(p.add + mo * p.Cout + n);
        const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x2 s = unpack2(y[k]) + unpack2(xs[k]);
          y[k] = pack2(s.x, s.y);
        }
      }
      *reinterpret_cast<uint4*>(p.out + mo * p.Cout + n) = uint4{y[0], y[1], y[2], y[3]};
    }
  }
}

// images f32 NCHW [B, C, H, W] (C <= 4) -> bf16 padded NHWC4 [B, H+2, W+2, 4] interior, optional
// per-channel normalisation (x - mean) / std (DiscreteVAE.norm, vae_model.py:132-140)
__global__ __launch_bounds__(256) void nchw_to_padded_nhwc4_kernel(const float* __restrict__ x, int B, int C, int H,
                                                                   int W, const float* __restrict__ mean,
                                                                   const float* __restrict__ stdv,
                                                                   __bf16* __restrict__ out) {
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
  uint2 o{pack2(v[0], v[1]), pack2(v[2], v[3])};
  *reinterpret_cast<uint2*>(out + (((long long)b * (H + 2) + y + 1) * (W + 2) + xw + 1) * 4) = o;
}

// torch.argmax's order: NaN above every number, ties to the smallest index (all -inf -> 0): always a valid id
__device__ __forceinline__ bool argmax_better(float a, int ai, float b, int bi) {
  const bool an = a != a, bn = b != b;
  if (an || bn) return an && (!bn || ai < bi);
  return a > b || (a == b && ai < bi);
}

// ids[m] = argmax_n logits[m, n] (first maximum, NaN wins like torch.argmax), one wave per row, 16-byte loads
__global__ __launch_bounds__(256) void argmax_rows_kernel(const __bf16* __restrict__ logits, long long ld, int M, int N,
                                                          long long* __restrict__ ids) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int n = lane * 8; n < N; n += 512) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(logits + (long long)m * ld + n);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float f = (float)v[k];
      if (argmax_better(f, n + k, best, bi)) { best = f; bi = n + k; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (argmax_better(ob, oi, best, bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) ids[m] = bi;
}

}  // namespace

extern "C" int memhip_conv2d_nhwc_bf16(const void* in, const void* weight, const float* bias, const void* add,
                                       void* out, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                                       int pad, int relu, int out_padded, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv2d: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && weight && out, "conv2d: null pointer");
  MEMHIP_REQUIRE((ksize == 4 && stride == 2 && pad == 1) || (ksize == 3 && stride == 1 && pad == 1) ||
                     (ksize == 1 && stride == 1 && pad == 0),
                 "conv2d: only the encoder's shapes (4x4/s2/p1, 3x3/s1/p1, 1x1) are provided");
  const bool cin4 = Cin == 4;
  MEMHIP_REQUIRE(cin4 ? (ksize == 4) : (Cin % 64 == 0), "conv2d: C_in must be 4 (first layer, 4x4) or a multiple of 64");
  MEMHIP_REQUIRE(Cout % 8 == 0, "conv2d: C_out must be a multiple of 8");
  ConvArgs p;
  p.in = (const __bf16*)in; p.w = (const __bf16*)weight; p.bias = bias; p.add = (const __bf16*)add; p.out = (__bf16*)out;
  p.B = B; p.Hp = H + 2; p.Wp = W + 2; p.Cin = Cin;
  p.Ho = (H + 2 * pad - ksize) / stride + 1; p.Wo = (W + 2 * pad - ksize) / stride + 1;
  p.Cout = Cout; p.kh = ksize; p.kw = ksize; p.stride = stride; p.off = 1 - pad; p.K = ksize * ksize * Cin;
  p.out_padded = out_padded; p.relu = relu; p.cin4 = cin4 ? 1 : 0;
  MEMHIP_REQUIRE(p.K % BK == 0, "conv2d: K = %d must be a multiple of 64", p.K);
  const long long M = (long long)B * p.Ho * p.Wo;
  MEMHIP_REQUIRE(M < (1LL << 31), "conv2d: too many output pixels");
  const int grid = cdiv(M, BM) * cdiv(Cout, BN);
  static bool attr_done = false;
  if (!attr_done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_gemm_kernel, dim3(grid), dim3(kThreads), 2 * kStageBytes, as_stream(stream), p);
  return check_launch("conv2d_nhwc_bf16");
}

extern "C" int memhip_nchw_to_padded_nhwc4(const float* x, int B, int C, int H, int W, const float* mean,
                                           const float* stdv, void* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && C >= 1 && C <= 4 && H > 0 && W > 0, "nchw_to_padded_nhwc4: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && out && (!mean == !stdv), "nchw_to_padded_nhwc4: null pointer");
  const long long n = (long long)B * H * W;
  hipLaunchKernelGGL(nchw_to_padded_nhwc4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     x, B, C, H, W, mean, stdv, (__bf16*)out);
  return check_launch("nchw_to_padded_nhwc4");
}

extern "C" int memhip_argmax_rows_bf16(const void* logits, int64_t ld, int M, int N, int64_t* ids,
                                       memhip_stream_t stream) {
  MEMHIP_REQUIRE(M >= 0 && N > 0 && N % 8 == 0 && ld % 8 == 0, "argmax_rows: N and ld must be multiples of 8");
  if (M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(logits && ids, "argmax_rows: null pointer");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), (const __bf16*)logits,
                     (long long)ld, M, N, (long long*)ids);
  return check_launch("argmax_rows_bf16");
}
