// Fused multi-head attention with the shared additive relative-position bias, forward and
// backward (reference: Attention.forward, mem/modeling_finetune.py:137-154, with the bias of
// RelativePositionBias :242-247 broadcast over the batch).  head_dim = 64 (ViT-B and ViT-L).
//
// CDNA4 mapping.  The whole 197-token problem of one (sample, head) stays on chip; nothing of the
// [B,H,N,N] score tensor ever reaches HBM.  8-wave workgroups, a wave owns a 32-token block.
// Scores are computed TRANSPOSED (S^T = K Q^T, v_mfma_f32_32x32x16_bf16) so that a lane owns one
// query column and a softmax row reduction is in-lane + one cross-half shuffle; the fp32
// accumulator tile is then already the B operand of the next product (O^T = V^T P^T, and in
// backward dV^T, dK^T, dQ^T) -- no LDS round trip for P / dS.  K/V (forward), Q/dO (backward, kv
// kernel) and K/V (backward, q kernel) head slices are staged once per workgroup as row-major,
// XOR-swizzled LDS images with LDS-DMA; "row" fragments are ds_read_b128, the fragments that must
// be read down a column use the transposing LDS read ds_read_b64_tr_b16 -- no transposed copies.
// The relative-position-bias gradient is reduced on chip into the 732-bucket table with
// fixed-point INTEGER LDS atomics (float LDS atomics are ~30x slower on gfx950) and flushed once
// per workgroup.
//
// Rounding points follow the reference under autocast: q k^T and the P V / dO V^T / dS K products
// are rounded to bf16, bias add + softmax (+ its backward) run in fp32, P and dS are rounded to
// bf16 when they feed an MFMA.
#include "common.h"

namespace {

using namespace memhip;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int HD = 64;   // head dim

__device__ __forceinline__ float bfr(float v) { return (float)(__bf16)v; }

__device__ __forceinline__ bf16x8 ld16(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

__device__ __forceinline__ bf16x8 cat4(const __bf16* lo, const __bf16* hi) {
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(lo), b = *reinterpret_cast<const bf16x4*>(hi);
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// bf16 fragment (k-step s) of an fp32 accumulator tile, scaled: regs 8s..8s+7
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s, float mul) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)(x[8 * s + j] * mul);
  return r;
}

__device__ __attribute__((aligned(256))) unsigned char g_attn_zero_page[128];   // zero-initialised

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// LDS image of a [TP tokens][64] bf16 head slice: 128-B rows, 16-B chunk c of token t lives at
// chunk position c ^ ((t >> 1) & 7)  (same XOR as the GEMM tiles: conflict-free ds_read_b128 row
// fragments; the transposing reads below are 2-way at worst).
__device__ __forceinline__ int tok_slot(int tok, int chunk) { return tok * 8 + (chunk ^ ((tok >> 1) & 7)); }

// Stage src[tok*ld + 0..63] (tok < T, zero beyond) with LDS-DMA: one wave-instruction = 8 tokens.
__device__ __forceinline__ void stage_head(char* dst, const __bf16* src, long long ld, int T, int TP) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int inst = wave; inst < TP / 8; inst += nw) {
    const int tok = inst * 8 + (lane >> 3), cpos = lane & 7;
    const int chunk = cpos ^ ((tok >> 1) & 7);
    const void* g = tok < T ? (const void*)(src + (long long)tok * ld + chunk * 8)
                            : (const void*)(g_attn_zero_page + cpos * 16);
    glds16(g, dst + inst * 1024);
  }
}

// row fragment: 8 consecutive head-dim elements (16-B chunk) of one token -> MFMA operand whose
// lane-row is the token (lane r = tok, k = 16t + 8hh + j)
__device__ __forceinline__ bf16x8 row_frag(const char* img, int tok, int chunk) {
  return *reinterpret_cast<const bf16x8*>(img + tok_slot(tok, chunk) * 16);
}

// column fragment via ds_read_b64_tr_b16: lane (r = l&31 -> d = db*32 + r, hh = l>>5) receives the
// 8 tokens tokbase+{0..3} and tokbase+8+{0..3} of head-dim column d  (tokbase already holds +4hh)
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 col_frag(const char* img, int tokbase, int db, int lane) {
  const int rhalf = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int ch = db * 4 + rhalf * 2 + (p >> 1), h8 = (p & 1) * 8;
  const int t0 = tokbase + q, t1 = t0 + 8;
  const char* a0 = img + tok_slot(t0, ch) * 16 + h8;
  const char* a1 = img + tok_slot(t1, ch) * 16 + h8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lo;
  u.s.h = hi;
  return u.v;
}

// Per-lane byte offsets inside a head image for token block 0; token block kb adds the
// compile-time constant kb*4096 (the XOR term only depends on the token's low 5 bits), so the
// unrolled loops address LDS as base + immediate and carry 12 address registers instead of
// recomputing (or hoisting) one swizzled address per fragment.
struct LaneOffs {
  int row[4];        // row_frag(tok = kb*32 + r, chunk 2t + hh)
  int col[2][2][2];  // col_frag(tokbase = kb*32 + 16ss + 4hh, db): [ss][db][lo/hi]
};
__device__ __forceinline__ LaneOffs lane_offs(int lane) {
  LaneOffs o;
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 4; ++t) o.row[t] = tok_slot(r, 2 * t + hh) * 16;
  const int rhalf = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
  for (int ss = 0; ss < 2; ++ss)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int ch = db * 4 + rhalf * 2 + (p >> 1), h8 = (p & 1) * 8;
      const int t0 = 16 * ss + 4 * hh + q;
      o.col[ss][db][0] = tok_slot(t0, ch) * 16 + h8;
      o.col[ss][db][1] = tok_slot(t0 + 8, ch) * 16 + h8;
    }
  return o;
}
__device__ __forceinline__ bf16x8 row_frag_o(const char* img, const LaneOffs& o, int kb, int t) {
  return *reinterpret_cast<const bf16x8*>(img + o.row[t] + kb * 4096);
}
__device__ __forceinline__ bf16x8 col_frag_o(const char* img, const LaneOffs& o, int kb, int ss, int db) {
  const char* a0 = img + o.col[ss][db][0] + kb * 4096;
  const char* a1 = img + o.col[ss][db][1] + kb * 4096;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lo;
  u.s.h = hi;
  return u.v;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------- forward
template <int NKB>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const __bf16* __restrict__ qkv, long long ldq, int T,
                                                       int D, int H, const float* __restrict__ bias,
                                                       __bf16* __restrict__ out, long long ldo,
                                                       float* __restrict__ lse) {
  constexpr int TP = NKB * 32;
  __shared__ __attribute__((aligned(16))) char Ks[TP * 128];
  __shared__ __attribute__((aligned(16))) char Vs[TP * 128];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const __bf16* base = qkv + (long long)b * T * ldq + h * HD;      // q slice of this head
  stage_head(Ks, base + D, ldq, T, TP);
  stage_head(Vs, base + 2 * D, ldq, T, TP);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  for (int qb = wave; qb < NKB; qb += 8) {
    const int q = qb * 32 + r;
    const int qc = q < T ? q : T - 1;
    bf16x8 Qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Qf[t] = ld16(base + (long long)qc * ldq + 16 * t + 8 * hh);
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) s[kb] = MFMA32(row_frag_o(Ks, lo, kb, t), Qf[t], s[kb]);
    }
    // + bias, key mask, row max  (lane: query q; regs: keys)
    const float* brow = bias + ((long long)h * TP + q) * TP;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key0 = kb * 32 + 8 * g + 4 * hh;
        const float4 bv = *reinterpret_cast<const float4*>(brow + key0);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = bfr(s[kb][4 * g + e]) + bb[e];
          if (key0 + e >= T) v = -INFINITY;
          s[kb][4 * g + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __expf(s[kb][i] - mx);
        s[kb][i] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (hh == 0 && q < T) lse[((long long)b * H + h) * TP + q] = mx + __logf(sum);
    // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = acc_frag(s[kb], ss, inv);
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(col_frag_o(Vs, lo, kb, ss, db), pf, o[db]);
      }
    }
    if (q < T) {
      __bf16* orow = out + ((long long)b * T + q) * ldo + h * HD;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (__bf16)o[db][4 * g + e];
          *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * hh) = w;
        }
    }
  }
}

// ------------------------------------------------------------------------------- backward
// Two kernels (P and dS are recomputed in each, in the orientation that makes their fp32
// accumulator tile directly the next MFMA's B operand):
//   attn_bwd_kv : workgroup = (sample, head); a wave owns 32 keys and sweeps the queries with tiles
//                 [q rows][key cols] -> dV^T, dK^T.  Also produces delta = rowsum(dO*O) for (B).
//   attn_bwd_q  : workgroup = (head, SPB consecutive samples); a wave owns 32 queries and sweeps the
//                 keys with tiles [key rows][q cols] -> dQ^T per sample, while the bias gradient
//                 dS^T is summed over the workgroup's samples IN REGISTERS and only then bucketed
//                 (LDS float atomics are ~125 cycles per wave-instruction: once per SPB samples).
template <int NKB>
__global__ __launch_bounds__(512) void attn_bwd_kv_kernel(const __bf16* __restrict__ qkv, long long ldq,
                                                          const __bf16* __restrict__ dout,
                                                          const __bf16* __restrict__ out, long long ldo,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ biasT,
                                                          __bf16* __restrict__ dqkv, long long lddq,
                                                          float* __restrict__ delta, float* __restrict__ dvbias,
                                                          int T, int D, int H) {
  constexpr int TP = NKB * 32;
  __shared__ __attribute__((aligned(16))) char Qs[TP * 128];
  __shared__ __attribute__((aligned(16))) char dOs[TP * 128];
  __shared__ __attribute__((aligned(16))) float lseS[TP];
  __shared__ __attribute__((aligned(16))) float delS[TP];
  __shared__ float vsum[HD];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long long row0 = (long long)b * T;
  const __bf16* qb_ = qkv + row0 * ldq + h * HD;          // Q' slice (already scaled)
  const __bf16* kb_ = qb_ + D;
  const __bf16* vb_ = qb_ + 2 * D;
  const __bf16* dob = dout + row0 * ldo + h * HD;
  const __bf16* ob = out + row0 * ldo + h * HD;
  stage_head(Qs, qb_, ldq, T, TP);
  stage_head(dOs, dob, ldo, T, TP);
  for (int q = threadIdx.x; q < TP; q += blockDim.x) {
    float dl = 0.f, l = 0.f;
    if (q < T) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bf16x8 a = ld16(dob + (long long)q * ldo + 8 * c), o = ld16(ob + (long long)q * ldo + 8 * c);
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += (float)a[i] * (float)o[i];
      }
      l = lse[((long long)b * H + h) * TP + q];
    }
    delS[q] = dl;
    lseS[q] = l;
    delta[((long long)b * H + h) * TP + q] = dl;
  }
  if (threadIdx.x < HD) vsum[threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  for (int kb = wave; kb < NKB; kb += 8) {
    const int key = kb * 32 + r;
    const int kc = key < T ? key : T - 1;
    bf16x8 Kf[4], Vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Kf[t] = ld16(kb_ + (long long)kc * ldq + 16 * t + 8 * hh);
      Vf[t] = ld16(vb_ + (long long)kc * ldq + 16 * t + 8 * hh);
    }
    f32x16 dVt[2], dKt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dVt[db][i] = 0.f; dKt[db][i] = 0.f; }
    const float* bcol = biasT + ((long long)h * TP + key) * TP;     // biasT[h][key][q]
    for (int qb = 0; qb < NKB; ++qb) {
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        S = MFMA32(row_frag_o(Qs, lo, qb, t), Kf[t], S);
        dP = MFMA32(row_frag_o(dOs, lo, qb, t), Vf[t], dP);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int q0 = qb * 32 + 8 * g + 4 * hh;
        const float4 bv = *reinterpret_cast<const float4*>(bcol + q0);
        const float4 lv = *reinterpret_cast<const float4*>(lseS + q0);
        const float4 dv = *reinterpret_cast<const float4*>(delS + q0);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, ll[4] = {lv.x, lv.y, lv.z, lv.w},
                    dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          const float sv = bfr(S[i]) + bb[e];
          const float p = (q0 + e < T && key < T) ? __expf(sv - ll[e]) : 0.f;
          S[i] = p;
          dP[i] = p * (bfr(dP[i]) - dd[e]);
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = acc_frag(S, ss, 1.0f), dsf = acc_frag(dP, ss, 1.0f);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dVt[db] = MFMA32(col_frag_o(dOs, lo, qb, ss, db), pf, dVt[db]);
          dKt[db] = MFMA32(col_frag_o(Qs, lo, qb, ss, db), dsf, dKt[db]);
        }
      }
    }
    // lane col = key, regs -> d
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 wv, wk;
#pragma unroll
        for (int e = 0; e < 4; ++e) { wv[e] = (__bf16)dVt[db][4 * g + e]; wk[e] = (__bf16)dKt[db][4 * g + e]; }
        if (key < T) {
          __bf16* drow = dqkv + (row0 + key) * lddq + h * HD;
          *reinterpret_cast<bf16x4*>(drow + 2 * D + db * 32 + 8 * g + 4 * hh) = wv;
          *reinterpret_cast<bf16x4*>(drow + D + db * 32 + 8 * g + 4 * hh) = wk;
        }
        if (dvbias) {                       // v_bias gradient: column sums of the stored (bf16) dV
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = key < T ? (float)wv[e] : 0.f;
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (r == 0) atomicAdd(vsum + db * 32 + 8 * g + 4 * hh + e, v);
          }
        }
      }
  }
  if (dvbias) {
    __syncthreads();
    if (threadIdx.x < HD) atomicAdd(dvbias + h * HD + threadIdx.x, vsum[threadIdx.x]);
  }
}

// attn_bwd_q: the relative-position-bias gradient is bucketed with INTEGER LDS atomics (fixed
// point): ds_add_f32 costs ~190 cycles per wave-instruction on gfx950, ds_add_u32 ~6 (measured,
// tools/micro/lds_atomic.hip).  Per sample the scale is 2^24 / bound with
//   |dS| = p |dP - delta| <= max_q |dO_q| * max_key |V_key| + max_q |delta_q| =: bound,
// so a bucket (<= 196 terms) cannot overflow int32 and the quantisation step is 2^-24 of the bound;
// the integer buckets are folded into fp32 buckets once per sample.
template <int NKB>
__global__ __launch_bounds__(512) void attn_bwd_q_kernel(const __bf16* __restrict__ qkv, long long ldq,
                                                         const __bf16* __restrict__ dout, long long ldo,
                                                         const float* __restrict__ lse,
                                                         const float* __restrict__ delta,
                                                         const float* __restrict__ bias,
                                                         const int* __restrict__ relidx, int nrd,
                                                         __bf16* __restrict__ dqkv, long long lddq,
                                                         float* __restrict__ dtable, float* __restrict__ dqbias,
                                                         int B, int T, int D, int H, float scale, int spb) {
  constexpr int TP = NKB * 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int KVB = 2 * TP * 128;                           // one [K | V] image pair; two of them (double buffer)
  float* qsum = reinterpret_cast<float*>(smem_raw + 4 * TP * 128);   // [64]
  float* binsf = qsum + HD;                                   // [nrd] fp32 buckets (this workgroup)
  int* binsi = reinterpret_cast<int*>(binsf + nrd);           // [nrd] fixed-point buckets (this sample)
  int* red = binsi + nrd;                                     // [4] block maxima (float bits)
  const int h = blockIdx.x % H, b0 = (blockIdx.x / H) * spb;
  const int b1 = b0 + spb < B ? b0 + spb : B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  for (int i = threadIdx.x; i < HD + 2 * nrd + 4; i += blockDim.x) qsum[i] = 0.f;   // contiguous region
  const int qb = wave;                       // NKB <= 8: one query block per wave
  const bool active = qb < NKB;
  const LaneOffs lo = lane_offs(lane);
  const int q = qb * 32 + r;
  const int qc = q < T ? q : T - 1;
  const float* brow = bias + ((long long)h * TP + (active ? q : 0)) * TP;
  const int* irow = relidx + (long long)(active ? q : 0) * TP;
  if (b0 < b1) {
    const __bf16* s0 = qkv + (long long)b0 * T * ldq + h * HD;
    stage_head(smem_raw, s0 + D, ldq, T, TP);
    stage_head(smem_raw + TP * 128, s0 + 2 * D, ldq, T, TP);
  }
  for (int b = b0; b < b1; ++b) {
    const int cur = (b - b0) & 1;
    const char* Ks = smem_raw + cur * KVB;
    const char* Vs = Ks + TP * 128;
    const long long row0 = (long long)b * T;
    const __bf16* qb_ = qkv + row0 * ldq + h * HD;
    const __bf16* dob = dout + row0 * ldo + h * HD;
    __syncthreads();                      // K/V of sample b have landed; sample b-1 is fully consumed
    if (b + 1 < b1) {                     // prefetch the next sample behind this one's compute
      const __bf16* s1 = qkv + (row0 + T) * ldq + h * HD;
      stage_head(smem_raw + (cur ^ 1) * KVB, s1 + D, ldq, T, TP);
      stage_head(smem_raw + (cur ^ 1) * KVB + TP * 128, s1 + 2 * D, ldq, T, TP);
    }
    // ---- bound for the fixed-point scale
    float lq = 0.f, dq_ = 0.f;
    if (dtable) {
      if (threadIdx.x < T) {
        float vn = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const bf16x8 v = row_frag(Vs, threadIdx.x, c);
#pragma unroll
          for (int i = 0; i < 8; ++i) vn += (float)v[i] * (float)v[i];
        }
        atomicMax(red + 0, __float_as_int(vn));                       // max |V_key|^2
      }
    }
    bf16x8 Qf[4], dOf[4];
    if (active) {
      float dn = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Qf[t] = ld16(qb_ + (long long)qc * ldq + 16 * t + 8 * hh);
        dOf[t] = ld16(dob + (long long)qc * ldo + 16 * t + 8 * hh);
#pragma unroll
        for (int i = 0; i < 8; ++i) dn += (float)dOf[t][i] * (float)dOf[t][i];
      }
      lq = lse[((long long)b * H + h) * TP + q];
      dq_ = delta[((long long)b * H + h) * TP + q];
      if (dtable && q < T) {
        dn += __shfl_xor(dn, 32);                                     // the two lane halves hold half a row each
        atomicMax(red + 1, __float_as_int(dn));                       // max |dO_q|^2
        atomicMax(red + 2, __float_as_int(fabsf(dq_)));               // max |delta_q|
      }
    }
    float fx = 0.f;
    if (dtable) {
      __syncthreads();
      const float bound = sqrtf(__int_as_float(red[0])) * sqrtf(__int_as_float(red[1])) + __int_as_float(red[2]);
      fx = bound > 0.f ? 16777216.0f / bound : 0.f;
    }
    if (active) {
      f32x16 dQt[2];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQt[db][i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        f32x16 St, dPt;
#pragma unroll
        for (int i = 0; i < 16; ++i) { St[i] = 0.f; dPt[i] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          St = MFMA32(row_frag_o(Ks, lo, kb, t), Qf[t], St);
          dPt = MFMA32(row_frag_o(Vs, lo, kb, t), dOf[t], dPt);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key0 = kb * 32 + 8 * g + 4 * hh;
          const float4 bv = *reinterpret_cast<const float4*>(brow + key0);
          const int4 iv = *reinterpret_cast<const int4*>(irow + key0);
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
          const int ii[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            const bool ok = (q < T) && (key0 + e < T);
            const float sv = bfr(St[i]) + bb[e];
            const float p = ok ? __expf(sv - lq) : 0.f;
            const float ds = p * (bfr(dPt[i]) - dq_);
            dPt[i] = ds;
            if (ok && dtable) atomicAdd(binsi + ii[e], __float2int_rn(ds * fx));
          }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 dsf = acc_frag(dPt, ss, 1.0f);
#pragma unroll
          for (int db = 0; db < 2; ++db) dQt[db] = MFMA32(col_frag_o(Ks, lo, kb, ss, db), dsf, dQt[db]);
        }
      }
      // lane col = q, regs -> d ;  d(q_lin) = d(q') * scale
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (__bf16)(bfr(dQt[db][4 * g + e]) * scale);
          if (q < T) *reinterpret_cast<bf16x4*>(dqkv + (row0 + q) * lddq + h * HD + db * 32 + 8 * g + 4 * hh) = w;
          if (dqbias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = q < T ? (float)w[e] : 0.f;
              for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
              if (r == 0) atomicAdd(qsum + db * 32 + 8 * g + 4 * hh + e, v);
            }
          }
        }
    }
    if (dtable) {                          // fold this sample's fixed-point buckets into fp32
      __syncthreads();
      const float inv = fx > 0.f ? 1.0f / fx : 0.f;
      for (int i = threadIdx.x; i < nrd; i += blockDim.x) {
        binsf[i] += (float)binsi[i] * inv;
        binsi[i] = 0;
      }
      if (threadIdx.x < 4) red[threadIdx.x] = 0;
    }
  }
  __syncthreads();
  if (dtable)
    for (int i = threadIdx.x; i < nrd; i += blockDim.x) {
      const float v = binsf[i];
      if (v != 0.f) atomicAdd(dtable + (long long)i * H + h, v);
    }
  if (dqbias && threadIdx.x < HD) atomicAdd(dqbias + h * HD + threadIdx.x, qsum[threadIdx.x]);
}

}  // namespace

#define ATTN_DISPATCH(NKB_EXPR, MACRO)                                    \
  switch (NKB_EXPR) {                                                     \
    case 1: MACRO(1); break; case 2: MACRO(2); break; case 3: MACRO(3); break; \
    case 4: MACRO(4); break; case 5: MACRO(5); break; case 6: MACRO(6); break; \
    case 7: MACRO(7); break; case 8: MACRO(8); break;                     \
    default: return fail(MEMHIP_EUNSUPPORTED, "attention: %d tokens > 256 needs the streaming variant", T); \
  }

extern "C" int memhip_attn_tokens_padded(int T) { return ((T + 31) / 32) * 32; }

extern "C" int memhip_attn_fwd(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads,
                               const float* bias_pad, void* out, int64_t ldo, float* lse,
                               memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && heads > 0 && D == heads * HD, "attn_fwd: head_dim must be 64 (D=%d heads=%d)", D, heads);
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && bias_pad && out && lse, "attn_fwd: null pointer");
  MEMHIP_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0, "attn_fwd: ld must be a multiple of 8");
  hipStream_t s = as_stream(stream);
  const int nkb = (T + 31) / 32;
#define FWD(N) hipLaunchKernelGGL(attn_fwd_kernel<N>, dim3(B * heads), dim3(512), 0, s, (const __bf16*)qkv, \
                                  (long long)ldqkv, T, D, heads, bias_pad, (__bf16*)out, (long long)ldo, lse)
  ATTN_DISPATCH(nkb, FWD)
#undef FWD
  return check_launch("attn_fwd");
}

extern "C" int memhip_attn_bwd(const void* qkv, int64_t ldqkv, const void* dout, const void* out, int64_t ldo,
                               const float* lse, const float* bias_pad, const float* biasT_pad,
                               const int32_t* relidx_pad, int num_rel, int B, int T, int D, int heads,
                               float scale, void* dqkv, int64_t lddqkv, float* dtable, float* dq_bias,
                               float* dv_bias, float* delta_ws, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && heads > 0 && D == heads * HD, "attn_bwd: head_dim must be 64");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && dout && out && lse && bias_pad && biasT_pad && dqkv && delta_ws, "attn_bwd: null pointer");
  MEMHIP_REQUIRE(!dtable || (relidx_pad && num_rel > 0), "attn_bwd: dtable needs relidx_pad");
  MEMHIP_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0 && lddqkv % 8 == 0, "attn_bwd: ld must be a multiple of 8");
  hipStream_t s = as_stream(stream);
  const int nkb = (T + 31) / 32;
  const int nrd = dtable ? num_rel : 0;
#define BWD(N)                                                                                          \
  {                                                                                                     \
    hipLaunchKernelGGL(attn_bwd_kv_kernel<N>, dim3(B * heads), dim3(512), 0, s, (const __bf16*)qkv,     \
                       (long long)ldqkv, (const __bf16*)dout, (const __bf16*)out, (long long)ldo, lse,  \
                       biasT_pad, (__bf16*)dqkv, (long long)lddqkv, delta_ws, dv_bias, T, D, heads);    \
    const size_t sm = (size_t)4 * N * 32 * 128 + (size_t)(HD + 2 * nrd + 4) * 4 + 16;                   \
    int spb = (B * heads + 255) / 256;           /* one workgroup per CU when the batch allows */       \
    if (spb < 1) spb = 1;                                                                               \
    if (spb > 16) spb = 16;                                                                             \
    static bool attr_done = false;                                                                      \
    if (!attr_done) {                                                                                   \
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_q_kernel<N>),               \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));   \
      attr_done = true;                                                                                 \
    }                                                                                                   \
    hipLaunchKernelGGL(attn_bwd_q_kernel<N>, dim3(((B + spb - 1) / spb) * heads), dim3(512), sm, s,     \
                       (const __bf16*)qkv, (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse,  \
                       delta_ws, bias_pad, relidx_pad, nrd, (__bf16*)dqkv, (long long)lddqkv, dtable,   \
                       dq_bias, B, T, D, heads, scale, spb);                                            \
  }
  ATTN_DISPATCH(nkb, BWD)
#undef BWD
  return check_launch("attn_bwd");
}
