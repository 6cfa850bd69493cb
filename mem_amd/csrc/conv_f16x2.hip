// "fp16 x 2" implicit-GEMM convolutions for the frozen dVAE tokenizer forward: fp32-class accuracy at fp16 MFMA speed
// (reference: eventvae/vae/vae_model.py:29-42,86-101,153-158; fp32 in the reference, mem/engine_for_pretraining.py:140-145).
//
// Every fp32 value v is carried as TWO fp16 numbers: hi = fp16(v) and lo = fp16((v - hi) * 2048) -- (v - hi) is exact in
// fp32 (hi keeps the top 11 bits of a 24-bit significand), so hi + lo / 2048 reproduces v to 2^-22 relative.  A product
// a * b = a_hi b_hi + (a_hi b_lo + a_lo b_hi) / 2048 + O(2^-22): THREE v_mfma_f32_16x16x32_f16 per tile step instead of one
// (fp16 x fp16 products are exact in the MFMA's fp32 accumulation), two accumulator sets (the cross terms are scaled once, in
// the epilogue).  Measured deviation of the 8192 logits from the fp32 module: <= 1.4e-5 at a logit spread of 1.77 (fp32
// summation-order noise is ~3e-6), ids equal to the reference's on both fixtures -- between the exact fp32 mode
// (conv_f32.hip, the default) and the bf16 mode (conv.hip, 2e-2) in error, ~2x faster than fp32.
// Layout: as conv.hip (NHWC with a one-pixel zero border, weights [C_out][ky][kx][c]) with two PLANES per tensor
// (hi plane, lo plane; plane stride passed in); the structure is conv.hip's 128x128x64 tile with an LDS-DMA-gathered A
// operand, doubled: {A_hi, A_lo, B_hi, B_lo} x 16 KiB per stage, two stages = 128 KiB (one workgroup per CU).
#include "common.h"

namespace {

using namespace memhip;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;              // one fp16 tile
constexpr int kStageBytes = 4 * kTileBytes;          // A_hi A_lo B_hi B_lo
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct ConvArgsH {
  const _Float16* in; long long in_plane;      // [2][B, Hp, Wp, Cin]
  const _Float16* w; long long w_plane;        // [2][Cout, K]
  const float* bias;
  const _Float16* add; long long add_plane;    // residual, laid out like `out` (two planes), or null
  void* out; long long out_plane;              // fp16 two planes, or fp32 dense when out_f32
  int B, Hp, Wp, Cin, Ho, Wo, Cout, kh, kw, stride, off, K;
  int out_padded, relu, cin4, out_f32;
};

__device__ __forceinline__ int swz_slot(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void split16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * kLoScale);
}

// NW = waves per workgroup: 4 (2 x 2 waves, 64 x 64 outputs each: ONE wave per SIMD -- nothing covers a wave's fragment reads, waits
// and barriers, the matrix pipe idles meanwhile) or 8 (4 x 2 waves, 32 x 64 each: two waves per SIMD; 12 instead of 16 fragment
// reads per 48 MFMAs and wave).  Same tile, staging and arithmetic: every accumulator sees the same sum order.
template <int NW>
__global__ __launch_bounds__(NW * 64) void conv_gemm_f16x2_kernel(ConvArgsH p) {
  constexpr int MI = 8 / NW * 2;                // 16-row fragments per wave: 4 (NW = 4) / 2 (NW = 8)
  constexpr int PW = 16 / NW;                   // LDS-DMA pieces per wave and tile: 4 / 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int M = p.B * p.Ho * p.Wo;
  const int nwg = gridDim.x;
  int pid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntn = (p.Cout + BN - 1) / BN;
  const int m0 = (pid / ntn) * BM, n0 = (pid % ntn) * BN;
  long long abase[PW], bbase[PW];
  int chunkg[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int inst = wave * PW + j;
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
    for (int j = 0; j < PW; ++j) {
      const int inst = wave * PW + j;
      const long long ao = p.cin4 ? ((long long)(chunkg[j] >> 1) * p.Wp + 2 * (chunkg[j] & 1)) * 4 : koff + chunkg[j] * 8;
      const long long bo = bbase[j] + k0 + chunkg[j] * 8;
      glds16(p.in + abase[j] + ao, dst + inst * 1024);
      glds16(p.in + p.in_plane + abase[j] + ao, dst + kTileBytes + inst * 1024);
      glds16(p.w + bo, dst + 2 * kTileBytes + inst * 1024);
      glds16(p.w + p.w_plane + bo, dst + 3 * kTileBytes + inst * 1024);
    }
  };
  f32x4 acch[MI][4], accx[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acch[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nk = p.K / BK;
  stage(0, smem);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) stage(t + 1, smem + (cur ^ 1) * kStageBytes);
    const char* Ah = smem + cur * kStageBytes;
    const char* Al = Ah + kTileBytes;
    const char* Bh = Ah + 2 * kTileBytes;
    const char* Bl = Ah + 3 * kTileBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 ah[MI], al[MI], bh[4], bl[4];
      const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int s = swz_slot(wr * (MI * 16) + i * 16 + (lane & 15), chunk) * 16;
        ah[i] = *reinterpret_cast<const half8*>(Ah + s);
        al[i] = *reinterpret_cast<const half8*>(Al + s);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = swz_slot(wc * 64 + j * 16 + (lane & 15), chunk) * 16;
        bh[j] = *reinterpret_cast<const half8*>(Bh + s);
        bl[j] = *reinterpret_cast<const half8*>(Bl + s);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acch[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acch[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: the wave's 64x64 sub-tile through LDS, 32 rows at a time (conv.hip's scheme); v = hh + x / 2048
  const int mw = m0 + wr * (MI * 16), nw = n0 + wc * 64;
  constexpr int LS = 72;
  float* wreg = reinterpret_cast<float*>(smem + wave * 16384);
  const int c8 = (lane & 7) * 8;
  const int n = nw + c8;
  float bias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias[k] = (p.bias && n + k < p.Cout) ? p.bias[n + k] : 0.f;
#pragma unroll
  for (int half = 0; half < MI / 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          wreg[(ii * 16 + (lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] =
              acch[half * 2 + ii][j][r] + accx[half * 2 + ii][j][r] * kLoInv;
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
      if (p.add) {                                      // ResBlock: net(x) + x, x reassembled from its two planes
        const half8 xh = *reinterpret_cast<const half8*>(p.add + mo * p.Cout + n);
        const half8 xl = *reinterpret_cast<const half8*>(p.add + p.add_plane + mo * p.Cout + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)xh[k] + (float)xl[k] * kLoInv;
      }
      if (p.out_f32) {
        float* o = reinterpret_cast<float*>(p.out) + mo * p.Cout + n;
        reinterpret_cast<float4*>(o)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(o)[1] = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        half8 oh, ol;
#pragma unroll
        for (int k = 0; k < 8; ++k) { _Float16 h, l; split16(v[k], h, l); oh[k] = h; ol[k] = l; }
        _Float16* o = reinterpret_cast<_Float16*>(p.out) + mo * p.Cout + n;
        *reinterpret_cast<half8*>(o) = oh;
        *reinterpret_cast<half8*>(o + p.out_plane) = ol;
      }
    }
  }
}

// ---- 256 x 128 x 32 tile, eight waves of 64 x 64 outputs (4 x 2), phase-interleaved.  In the 128 x 128 kernel above a wave reads
// 12 fragments for 24 MFMAs and all eight waves run in lockstep: they stage, read and compute together, so on every SIMD the
// matrix pipe idles while both of its waves load (~0.39 of the fp16 peak).  A 64 x 64 wave tile reads 16 fragments for 48 MFMAs;
// eight such waves need 256 output rows; with BK = 32 a stage is {A_hi, A_lo} x 16 KiB + {B_hi, B_lo} x 8 KiB = 48 KiB, three
// stages 144 KiB.  Measured on the 56 x 56 and 28 x 28 layers (B = 256): the wide tile alone 14.4 -> 12.9 ms (a third stage and
// conflict-free reads changed nothing: neither the loads nor the LDS were the limit), with the two waves of a SIMD half a phase
// apart -> 11.6 ms; the whole fp16x2 forward 21.45 -> 18.64 ms = 1.006 PFLOP/s of fp16 MFMA work, logits bit-equal.
// Rows are 64 bytes (4 chunks of 16 B): slot(row, chunk) = row * 4 + (chunk ^ wswz(row)).  A ds_read_b128 is served in four groups of
// 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, ...): a group holds rows 0-3 and 12-15 of the fragment at chunk c and rows 4-11 at
// chunk c ^ 1, and the four row quads must land on four different 16-byte columns of the 256-byte bank row: wswz = 0, 3, 2, 1 for
// quad 0, 1, 2, 3 gives c, c ^ 2, c ^ 3, c ^ 1.  Same arithmetic and the same k order per accumulator as the 128 x 128 kernel:
// bit-equal output.
constexpr int WBM = 256, WBN = 128, WBK = 32;
__device__ __forceinline__ int wswz(int row) { return (0 - (row >> 2)) & 3; }
constexpr int kWATile = WBM * WBK * 2, kWBTile = WBN * WBK * 2;     // bytes of one fp16 plane
constexpr int kWStage = 2 * kWATile + 2 * kWBTile;

__global__ __launch_bounds__(512) void conv_gemm_f16x2_wide_kernel(ConvArgsH p) {
  constexpr int MI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int M = p.B * p.Ho * p.Wo;
  const int nwg = gridDim.x;
  int pid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntn = (p.Cout + WBN - 1) / WBN;
  const int m0 = (pid / ntn) * WBM, n0 = (pid % ntn) * WBN;
  // LDS-DMA pieces of 1 KiB = 16 rows x 64 B: the wave stages A rows [32 wave, 32 wave + 32) and B rows [16 wave, 16 wave + 16)
  long long abase[2], bbase;
  int chunka[2], chunkb;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 16 + (lane >> 2);
    chunka[j] = (lane & 3) ^ wswz(row);
    int m = m0 + row;
    m = m < M ? m : M - 1;
    const int hw = p.Ho * p.Wo;
    const int b = m / hw, r = m - b * hw;
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    abase[j] = (((long long)b * p.Hp + oy * p.stride + p.off) * p.Wp + ox * p.stride + p.off) * p.Cin;
  }
  {
    const int row = wave * 16 + (lane >> 2);
    chunkb = (lane & 3) ^ wswz(row);
    int n = n0 + row;
    n = n < p.Cout ? n : p.Cout - 1;
    bbase = (long long)n * p.K;
  }
  // pieces of k-step t in two sets of three (one set per phase): set 0 = A rows of piece 0 (hi, lo) + B hi, set 1 = A piece 1 + B lo
  auto stage_set = [&](int t, char* dst, int set) {
    const int k0 = t * WBK;
    const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const long long koff = ((long long)ky * p.Wp + kx) * p.Cin + c0;
    const long long ao = abase[set] + koff + chunka[set] * 8;
    glds16(p.in + ao, dst + (wave * 2 + set) * 1024);
    glds16(p.in + p.in_plane + ao, dst + kWATile + (wave * 2 + set) * 1024);
    const long long bo = bbase + k0 + chunkb * 8;
    if (set == 0) glds16(p.w + bo, dst + 2 * kWATile + wave * 1024);
    else glds16(p.w + p.w_plane + bo, dst + 2 * kWATile + kWBTile + wave * 1024);
  };
  f32x4 acch[MI][4], accx[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acch[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // Phase-interleaved schedule (gemm_p8.hip's): a k-step is two PHASES (output rows 0-31 and 32-63 of the wave), a phase is a LOAD
  // segment (its fragment reads + three LDS-DMA pieces of k-step t + 2), s_barrier, a COMPUTE segment of 24 MFMAs, s_barrier.
  // Waves 4-7 (the second wave of every SIMD) pass one extra barrier first, so that on every SIMD one wave computes while the other
  // loads.  Three stage buffers.  RAW (LDS-DMA -> ds_read): the counted wait of a LOAD segment leaves only this and the previous
  // phase's pieces in flight; k-step t + 2's pieces are issued in the phases of k-step t and first read two phases after the wait
  // that retires them.  WAR (ds_read -> LDS-DMA): reads are retired (lgkmcnt(0)) before the barrier that ends their LOAD segment;
  // the buffer is restaged at least one barrier later.
#define CW_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  const int nk = p.K / WBK;
  const bool late = wave >= 4;
  stage_set(0, smem, 0); stage_set(0, smem, 1);
  if (nk > 1) { stage_set(1, smem + kWStage, 0); stage_set(1, smem + kWStage, 1); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CW_BARRIER();
  if (late) CW_BARRIER();
  int cur = 0;
  const int chunk = lane >> 4;
  half8 a_h[2], a_l[2], bh[4], bl[4];
  for (int t = 0; t < nk; ++t) {
    const char* Ah = smem + cur * kWStage;
    const char* Al = Ah + kWATile;
    const char* Bh = Ah + 2 * kWATile;
    const char* Bl = Bh + kWBTile;
    char* nxt = smem + (cur >= 1 ? cur - 1 : 2) * kWStage;
    const bool more = t + 2 < nk;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      // ---- LOAD segment
      if (ph == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wc * 64 + j * 16 + (lane & 15);
          const int sl = (row * 4 + (chunk ^ wswz(row))) * 16;
          bh[j] = *reinterpret_cast<const half8*>(Bh + sl);
          bl[j] = *reinterpret_cast<const half8*>(Bl + sl);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wr * 64 + (ph * 2 + i) * 16 + (lane & 15);
        const int sl = (row * 4 + (chunk ^ wswz(row))) * 16;
        a_h[i] = *reinterpret_cast<const half8*>(Ah + sl);
        a_l[i] = *reinterpret_cast<const half8*>(Al + sl);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        stage_set(t + 2, nxt, ph);
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      CW_BARRIER();
      // ---- COMPUTE segment: 24 MFMAs (the two products into accx[i][j] keep their order, eight MFMAs apart)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acch[ph * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[i], bh[j], acch[ph * 2 + i][j], 0, 0, 0);
          accx[ph * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[i], bl[j], accx[ph * 2 + i][j], 0, 0, 0);
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          accx[ph * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[i], bh[j], accx[ph * 2 + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      CW_BARRIER();
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  if (!late) CW_BARRIER();                      // waves 0-3 are one barrier ahead: realign before the epilogue reuses the LDS
#undef CW_BARRIER

  // ---- epilogue: as above, the wave's 64 x 64 sub-tile through 9 KiB of LDS, 32 rows at a time
  const int mw = m0 + wr * 64, nw = n0 + wc * 64;
  constexpr int LS = 72;
  float* wreg = reinterpret_cast<float*>(smem + wave * (32 * LS * 4));
  const int c8 = (lane & 7) * 8;
  const int n = nw + c8;
  float bias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias[k] = (p.bias && n + k < p.Cout) ? p.bias[n + k] : 0.f;
#pragma unroll
  for (int half = 0; half < MI / 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          wreg[(ii * 16 + (lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] =
              acch[half * 2 + ii][j][r] + accx[half * 2 + ii][j][r] * kLoInv;
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
      if (p.add) {
        const half8 xh = *reinterpret_cast<const half8*>(p.add + mo * p.Cout + n);
        const half8 xl = *reinterpret_cast<const half8*>(p.add + p.add_plane + mo * p.Cout + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)xh[k] + (float)xl[k] * kLoInv;
      }
      if (p.out_f32) {
        float* o = reinterpret_cast<float*>(p.out) + mo * p.Cout + n;
        reinterpret_cast<float4*>(o)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(o)[1] = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        half8 oh, ol;
#pragma unroll
        for (int k = 0; k < 8; ++k) { _Float16 h, l; split16(v[k], h, l); oh[k] = h; ol[k] = l; }
        _Float16* o = reinterpret_cast<_Float16*>(p.out) + mo * p.Cout + n;
        *reinterpret_cast<half8*>(o) = oh;
        *reinterpret_cast<half8*>(o + p.out_plane) = ol;
      }
    }
  }
}

// ---- the FIRST layer (C_in = 4: K = 64 is one k-tile; at batch 256 it writes 4.9 GB of planes).  As 75 264 workgroups of the
// 128 x 128 kernel it took 1.82 ms -- 1.43 ms of it with the stores removed: a workgroup's address arithmetic, LDS-DMA round trip,
// 48 MFMAs and epilogue run strictly one after the other, alone on the CU (tools/exp: two workgroups per CU gave -5 %).  Here ONE
// workgroup per CU is persistent: its column tile of the weights stays in LDS, the input rows of the NEXT row tile are requested before
// the current tile's epilogue (their LDS-DMA pieces are older than the epilogue's stores in the in-order vmcnt queue, so the wait
// for them leaves the eight stores per lane in flight: the stores of tile i drain under the MFMAs of tile i + 1).
// LDS: B hi / lo 32 KiB + A hi / lo x 2 buffers 64 KiB + 8 x 4.5 KiB epilogue staging (16 rows at a time) = 133 KiB.
// Same arithmetic, same k order: bit-equal to the 128 x 128 kernel.
__global__ __launch_bounds__(512) void conv_gemm_f16x2_first_kernel(ConvArgsH p) {
  constexpr int MI = 2, PW = 2;
  constexpr int kBBytes = 2 * kTileBytes, kABytes = 2 * kTileBytes, kWreg = 16 * 72 * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Bs = smem;                               // [B_hi | B_lo]
  char* As = smem + kBBytes;                     // two buffers of [A_hi | A_lo]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* wreg = reinterpret_cast<float*>(smem + kBBytes + 2 * kABytes + wave * kWreg);
  const unsigned wreg_lds = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem) + (unsigned)(kBBytes + 2 * kABytes + wave * kWreg);
  const int wr = wave >> 1, wc = wave & 1;
  const int M = p.B * p.Ho * p.Wo;
  const int ntn = (p.Cout + BN - 1) / BN;
  const int ncol = gridDim.x / ntn;              // workgroups per column tile (gridDim.x is a multiple of ntn)
  const int n0 = ((int)blockIdx.x % ntn) * BN;
  const int mt0 = (int)blockIdx.x / ntn;
  int chunkg[PW];
  long long abase[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int row = (wave * PW + j) * 8 + (lane >> 3);
    chunkg[j] = (lane & 7) ^ ((row >> 1) & 7);
  }
  auto a_setup = [&](int mt) {                   // gather bases of row tile mt (rows clamped to M - 1)
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int row = (wave * PW + j) * 8 + (lane >> 3);
      int m = mt * BM + row;
      m = m < M ? m : M - 1;
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, r = m - b * hw;
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      abase[j] = (((long long)b * p.Hp + oy * p.stride + p.off) * p.Wp + ox * p.stride + p.off) * p.Cin;
    }
  };
  auto a_stage = [&](char* dst) {                // C_in = 4: chunk c of a row = taps (ky, kx) = (c >> 1, 2 (c & 1) .. + 1), 4 channels each
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int inst = wave * PW + j;
      const long long ao = abase[j] + ((long long)(chunkg[j] >> 1) * p.Wp + 2 * (chunkg[j] & 1)) * 4;
      glds16(p.in + ao, dst + inst * 1024);
      glds16(p.in + p.in_plane + ao, dst + kTileBytes + inst * 1024);
    }
  };
  // the weights of this column tile: once
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int inst = wave * PW + j;
    const int row = inst * 8 + (lane >> 3);
    int n = n0 + row;
    n = n < p.Cout ? n : p.Cout - 1;
    const long long bo = (long long)n * p.K + chunkg[j] * 8;
    glds16(p.w + bo, Bs + inst * 1024);
    glds16(p.w + p.w_plane + bo, Bs + kTileBytes + inst * 1024);
  }
  const int c8 = (lane & 7) * 8;
  const int n = n0 + wc * 64 + c8;
  float bias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias[k] = p.bias ? p.bias[n + k] : 0.f;
  const int nmt = M / BM;                       // whole tiles only (the launcher checks): every lane issues every store
  if (mt0 < nmt) { a_setup(mt0); a_stage(As); }
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));          // weights, bias and the first rows are in
  int cur = 0;
  for (int mt = mt0; mt < nmt; mt += ncol) {
    // A(mt) has landed; the only younger vector-memory operations are the previous tile's 8 stores, which stay in flight.  Behind the barrier every wave has finished the previous tile, whose buffer takes the next rows.
    // (the waits go through the builtin so that the compiler's own bookkeeping sees them: behind the vmcnt(0) in front of the loop the
    // bias registers are known to be loaded, and it adds no draining wait of its own in front of their first use in the epilogue)
    __builtin_amdgcn_s_waitcnt(8 | (7 << 4) | (15 << 8));
    asm volatile("s_barrier" ::: "memory");
    if (mt + ncol < nmt) { a_setup(mt + ncol); a_stage(As + (cur ^ 1) * kABytes); }
    const char* Ah = As + cur * kABytes;
    const char* Al = Ah + kTileBytes;
    const char* Bh = Bs;
    const char* Bl = Bs + kTileBytes;
    f32x4 acch[MI][4], accx[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { acch[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 ah[MI], al[MI], bh[4], bl[4];
      const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int sl = swz_slot(wr * (MI * 16) + i * 16 + (lane & 15), chunk) * 16;
        ah[i] = *reinterpret_cast<const half8*>(Ah + sl);
        al[i] = *reinterpret_cast<const half8*>(Al + sl);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sl = swz_slot(wc * 64 + j * 16 + (lane & 15), chunk) * 16;
        bh[j] = *reinterpret_cast<const half8*>(Bh + sl);
        bl[j] = *reinterpret_cast<const half8*>(Bl + sl);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acch[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acch[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue, 16 rows at a time through the wave's own 4.5 KiB
    constexpr int LS = 72;
    const int mw = mt * BM + wr * 32;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) wreg[((lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] = acch[i][j][r] + accx[i][j][r] * kLoInv;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int m = mw + i * 16 + row;
        long long mo = m;
        if (p.out_padded) {
          const int hw = p.Ho * p.Wo;
          const int b = m / hw, r = m - b * hw;
          const int oy = r / p.Wo, ox = r - oy * p.Wo;
          mo = ((long long)b * (p.Ho + 2) + oy + 1) * (p.Wo + 2) + ox + 1;
        }
        // (read with inline asm: in front of a C++ LDS read hipcc drains the LDS-DMA stream -- s_waitcnt vmcnt(0) -- since it cannot
        // tell that the pieces in flight land in other LDS bytes)
        f32x4 v0, v1;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1) : "v"(wreg_lds + (unsigned)((row * LS + c8) * 4)) : "memory");
        float v[8] = {v0[0] + bias[0], v0[1] + bias[1], v0[2] + bias[2], v0[3] + bias[3],
                      v1[0] + bias[4], v1[1] + bias[5], v1[2] + bias[6], v1[3] + bias[7]};
        if (p.relu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        half8 oh, ol;
#pragma unroll
        for (int k = 0; k < 8; ++k) { _Float16 h, l; split16(v[k], h, l); oh[k] = h; ol[k] = l; }
        _Float16* o = reinterpret_cast<_Float16*>(p.out) + mo * p.Cout + n;
        *reinterpret_cast<half8*>(o) = oh;
        *reinterpret_cast<half8*>(o + p.out_plane) = ol;
      }
    }
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may outlive the workgroup
}

// images f32 NCHW [B, C<=4, H, W] -> two fp16 planes of padded NHWC4, optional (x - mean) / std
__global__ __launch_bounds__(256) void nchw_to_padded_nhwc4_f16x2_kernel(const float* __restrict__ x, int B, int C, int H,
                                                                         int W, const float* __restrict__ mean,
                                                                         const float* __restrict__ stdv,
                                                                         _Float16* __restrict__ out, long long plane) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * H * W) return;
  const int xw = (int)(i % W);
  const long long t = i / W;
  const int y = (int)(t % H), b = (int)(t / H);
  _Float16 h[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
  for (int c = 0; c < C; ++c) {
    float u = x[(((long long)b * C + c) * H + y) * W + xw];
    if (mean) u = (u - mean[c]) / stdv[c];
    split16(u, h[c], l[c]);
  }
  const long long o = (((long long)b * (H + 2) + y + 1) * (W + 2) + xw + 1) * 4;
#pragma unroll
  for (int c = 0; c < 4; ++c) { out[o + c] = h[c]; out[plane + o + c] = l[c]; }
}

}  // namespace

extern "C" int memhip_conv2d_nhwc_f16x2(const void* in, int64_t in_plane, const void* weight, int64_t w_plane, const float* bias,
                                        const void* add, int64_t add_plane, void* out, int64_t out_plane, int B, int H, int W,
                                        int Cin, int Cout, int ksize, int stride, int pad, int relu, int out_padded,
                                        int out_f32, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv2d_f16x2: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && weight && out, "conv2d_f16x2: null pointer");
  MEMHIP_REQUIRE((ksize == 4 && stride == 2 && pad == 1) || (ksize == 3 && stride == 1 && pad == 1) ||
                     (ksize == 1 && stride == 1 && pad == 0),
                 "conv2d_f16x2: only the encoder's shapes (4x4/s2/p1, 3x3/s1/p1, 1x1) are provided");
  const bool cin4 = Cin == 4;
  MEMHIP_REQUIRE(cin4 ? (ksize == 4) : (Cin % 64 == 0), "conv2d_f16x2: C_in must be 4 (first layer, 4x4) or a multiple of 64");
  MEMHIP_REQUIRE(Cout % 8 == 0, "conv2d_f16x2: C_out must be a multiple of 8");
  MEMHIP_REQUIRE(!(out_f32 && out_padded), "conv2d_f16x2: the fp32 output is the dense token-logit matrix");
  ConvArgsH p;
  p.in = (const _Float16*)in; p.in_plane = in_plane; p.w = (const _Float16*)weight; p.w_plane = w_plane; p.bias = bias;
  p.add = (const _Float16*)add; p.add_plane = add_plane; p.out = out; p.out_plane = out_plane;
  p.B = B; p.Hp = H + 2; p.Wp = W + 2; p.Cin = Cin;
  p.Ho = (H + 2 * pad - ksize) / stride + 1; p.Wo = (W + 2 * pad - ksize) / stride + 1;
  p.Cout = Cout; p.kh = ksize; p.kw = ksize; p.stride = stride; p.off = 1 - pad; p.K = ksize * ksize * Cin;
  p.out_padded = out_padded; p.relu = relu; p.cin4 = cin4 ? 1 : 0; p.out_f32 = out_f32;
  MEMHIP_REQUIRE(p.K % BK == 0, "conv2d_f16x2: K = %d must be a multiple of 64", p.K);
  const long long M = (long long)B * p.Ho * p.Wo;
  MEMHIP_REQUIRE(M < (1LL << 31), "conv2d_f16x2: too many output pixels");
  const int grid = cdiv(M, BM) * cdiv(Cout, BN);
  static bool attr_done = false;
  if (!attr_done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f16x2_kernel<4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f16x2_kernel<8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
    attr_done = true;
  }
  // the 256 x 128 tile where its grid is at least two full rounds of the chip: every layer of the encoder at batch 256 but the first
  // (the 14 x 14 layers are 588 workgroups = 2.3 rounds and still gain: forward 18.55 -> 18.30 ms against the finer 128 x 128 tiles)
  const int wgrid = cdiv(M, WBM) * cdiv(Cout, WBN);
  if (cin4 && opt(OPT_CONV_WAVES) >= 16 && p.K == BK && !add && !out_f32 && M % BM == 0 && Cout % BN == 0) {
    // the first layer, persistent (whole tiles only: every lane then issues every store, which the kernel's counted waits rely on)
    const int ntn = Cout / BN;
    int cols = max_cus() / ntn;
    cols = cols < 1 ? 1 : cols;
    const int nmt = (int)(M / BM);
    cols = cols > nmt ? nmt : cols;
    constexpr int kFirstLds = 2 * kTileBytes + 2 * 2 * kTileBytes + 8 * 16 * 72 * 4;
    static bool fattr_done = false;
    if (!fattr_done) {
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f16x2_first_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kFirstLds));
      fattr_done = true;
    }
    hipLaunchKernelGGL(conv_gemm_f16x2_first_kernel, dim3(cols * ntn), dim3(512), kFirstLds, as_stream(stream), p);
    return check_launch("conv2d_nhwc_f16x2(first)");
  }
  if (!cin4 && (opt(OPT_CONV_WAVES) == 32 || (opt(OPT_CONV_WAVES) == 16 && wgrid >= 2 * max_cus()))) {     // 32: the wide tile at any size (tests)
    static bool wattr_done = false;
    if (!wattr_done) {
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f16x2_wide_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kWStage));
      wattr_done = true;
    }
    hipLaunchKernelGGL(conv_gemm_f16x2_wide_kernel, dim3(wgrid), dim3(512), 3 * kWStage, as_stream(stream), p);
  } else if (opt(OPT_CONV_WAVES) == 4)
    hipLaunchKernelGGL(conv_gemm_f16x2_kernel<4>, dim3(grid), dim3(256), 2 * kStageBytes, as_stream(stream), p);
  else
    hipLaunchKernelGGL(conv_gemm_f16x2_kernel<8>, dim3(grid), dim3(512), 2 * kStageBytes, as_stream(stream), p);
  return check_launch("conv2d_nhwc_f16x2");
}

extern "C" int memhip_nchw_to_padded_nhwc4_f16x2(const float* x, int B, int C, int H, int W, const float* mean,
                                                 const float* stdv, void* out, int64_t out_plane, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && C >= 1 && C <= 4 && H > 0 && W > 0, "nchw_to_padded_nhwc4_f16x2: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && out && (!mean == !stdv), "nchw_to_padded_nhwc4_f16x2: null pointer");
  const long long n = (long long)B * H * W;
  hipLaunchKernelGGL(nchw_to_padded_nhwc4_f16x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     x, B, C, H, W, mean, stdv, (_Float16*)out, (long long)out_plane);
  return check_launch("nchw_to_padded_nhwc4_f16x2");
}
