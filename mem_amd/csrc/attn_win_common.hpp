// Shared device helpers of the window-attention kernels (attn_win.hip: 8 waves x 32 resident tokens; attn_win2.hip: 4 waves x 64,
// one wave per SIMD): slot geometry, chunk staging, XCD-aware workgroup map, packed-bf16 idioms, table set-up, column sums.
// Internal linkage: each translation unit gets its own copy.
#pragma once
#include "attn_common.hpp"
#include <type_traits>

namespace {

template <int WW> struct WinGeo {
  static constexpr int WS = (WW + 7) & ~7;        // slots per grid row
  static constexpr int CT = 128;                   // slots per chunk
  static constexpr int RPC = CT / WS;              // grid rows per chunk
  static constexpr int PAD0 = RPC * WS;            // first padding slot of a chunk (the cls token in chunk 0)
  static constexpr int P = 2 * WW - 1;
  static_assert(PAD0 < CT && PAD0 % 8 == 0, "the window width needs at least one padding slot per chunk");
  static constexpr int CLS_KB = PAD0 / 32, CLS_G = (PAD0 % 32) / 8;
  static constexpr int CQ = ((RPC - 1) * P + WS + 8 + 3) & ~3;     // floats of the constant strip of the cls row
  // (slot s of a chunk, s a multiple of 4) -> constant part of the bucket index; s + 4 stays in the same grid row
  static constexpr int imm(int s) { return (s / WS) * P + (s % WS); }
  static constexpr bool valid(int s) { return s < PAD0 && (s % WS) < WW; }
  static constexpr int row(int s) { return s / WS; }
};

// the slots of chunk c of a head slice -> chunk image (rows indexed by the slot; zero rows for padding slots)
template <int WW>
__device__ __forceinline__ void stage_chunk_win(char* dst, const __bf16* src, long long ld, int c, int Wh) {
  using G = WinGeo<WW>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int inst = wave; inst < G::CT / 8; inst += nw) {
    const int lt = inst * 8 + (lane >> 3), cpos = lane & 7;
    const int chunk = cpos ^ img_key(lt);
    const int j = lt / G::WS, kx = lt - j * G::WS, ky = c * G::RPC + j;
    bool ok = lt < G::PAD0 && kx < WW && ky < Wh;
    int tok = 1 + ky * WW + kx;
    if (c == 0 && lt == G::PAD0) { ok = true; tok = 0; }
    const void* g = ok ? (const void*)(src + (long long)tok * ld + chunk * 8) : (const void*)(g_attn_zero_page + cpos * 16);
    glds16(g, dst + inst * 1024);
  }
}

// Workgroup -> (group of 8 resident blocks, head, first sample) with all GROUPS of one (head, sample slot) on ONE XCD
// (the hardware deals consecutive workgroup ids round-robin to the 8 XCDs): the groups of a pair stream the same K / V (or
// Q' / dO) chunks, and with the natural (group, head, sample) grid they sat on different XCDs -- every chunk came from HBM once
// per group (profiles/r05_final_vitl_traffic.json before the remap: forward 1 852 MB per launch for 630 MB of operands).
// Grid = 8 * ceil(pairs / 8) * groups; ids behind the last pair return.
struct WinWg { int group, h, bz; bool live; };
__device__ __forceinline__ WinWg win_wg(int groups, int heads, int nbz) {
  const int i = (int)blockIdx.x, xcd = i & 7, j = i >> 3;
  const int pair = (j / groups) * 8 + xcd;
  WinWg w;
  w.group = j % groups;
  w.h = pair % heads;
  w.bz = pair / heads;
  w.live = pair < heads * nbz;
  return w;
}

struct __attribute__((packed, aligned(4))) F2u { float a, b; };
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// (idioms of attn16.hip) two fp32 values rounded to one packed bf16 pair; c + (low / high half of the pair) in ONE
// instruction: v_dot2c_f32_bf16 with the selector pair (1, 0) / (0, 1) -- unpack + bias add (the fp32 sum is truncated, not
// rounded: 1 ulp of fp32, tools/micro/dot2_bf16.hip).  The low selector must live in a register (hipcc encodes 0x3f80 as the
// inline constant 1.0, which the instruction reads as the HIGH half).
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ unsigned sel_lo_reg() {
  unsigned v;
  asm volatile("s_mov_b32 %0, 0x3f80" : "=s"(v));
  return v;
}
__device__ __forceinline__ float add_lo(unsigned pk, float c, unsigned sel_lo) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk), __builtin_bit_cast(bf16x2_t, sel_lo), c, false);
}
__device__ __forceinline__ float add_hi(unsigned pk, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk), __builtin_bit_cast(bf16x2_t, 0x3F800000u), c, false);
}
// transposing column fragment with the block offset as an IMMEDIATE (col_frag_o adds kb * 4096 per read: 32 v_add per chunk)
struct ColAddr { unsigned a[2][2][2]; };           // absolute LDS addresses of block 0: [ss][db][lo / hi]
__device__ __forceinline__ ColAddr col_addr(const char* img, const LaneOffs& o) {
  ColAddr c;
  const unsigned b = lds_addr_of(img);
#pragma unroll
  for (int ss = 0; ss < 2; ++ss)
#pragma unroll
    for (int db = 0; db < 2; ++db) { c.a[ss][db][0] = b + o.col[ss][db][0]; c.a[ss][db][1] = b + o.col[ss][db][1]; }
  return c;
}
template <int OFF>
__device__ __forceinline__ bf16x8 col_frag_i(const ColAddr& c, int ss, int db) {
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.s.l) : "v"(c.a[ss][db][0]), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.s.h) : "v"(c.a[ss][db][1]), "n"(OFF) : "memory");
  return u.v;
}
// two consecutive floats at an ABSOLUTE LDS byte address + compile-time offset (4-byte aligned: ds_read2_b32 base offset0 offset1)
template <int OFF>
__device__ __forceinline__ void lds_pair(unsigned base, float& a, float& b) {
  const auto* p = reinterpret_cast<const __attribute__((address_space(3))) F2u*>(base + (unsigned)OFF);
  a = p->a;
  b = p->b;
}

// table in LDS for kernels whose REGISTERS run over streamed KEYS (forward, dQ): reversed, R[i] = table[NB - 1 - i] * mul,
// so that bucket(q, k) sits at A(q) + ky P + kx with A(q) = NB - 1 - (qy + Wh - 1) P - (qx + Ww - 1) >= 0;
// for kernels whose registers run over streamed QUERIES (dK / dV): forward order, bucket at Kp(k) + qy P + qx with
// Kp(k) = (Wh - 1 - ky) P + (Ww - 1 - kx) >= 0.  Cq = the constant strip a cls lane reads instead (stride 0).
template <int WW>
__device__ __forceinline__ void win_setup(float* R, float* Cq, const float* table, int nrd, int H, int h, int Wh, float mul,
                                          bool reversed, int cls_bucket) {
  using G = WinGeo<WW>;
  const int NB = (2 * Wh - 1) * G::P, NBP = (NB + 3) & ~3;
  for (int i = threadIdx.x; i < NB; i += blockDim.x) R[i] = table[(long long)(reversed ? NB - 1 - i : i) * H + h] * mul;
  // the alignment pad behind the table is READ (never used): the rows behind a ragged last chunk address up to
  // (RPC - 2) P + Ww + WS + 5 floats past the table, i.e. the pad and the strip.  The dK / dV kernel masks those slots through
  // -lse = -inf, which only yields a probability of 0 if the bias it adds is finite -- found by tools/stress_attn_win.py as
  // irreproducible NaNs in dK / dV of the three keys (ky = 0, kx = Ww - 3 ..) whose ragged-row buckets fall on the pad
  if (threadIdx.x < NBP - NB) R[NB + threadIdx.x] = 0.f;
  const float cv = table[(long long)cls_bucket * H + h] * mul;
  for (int i = threadIdx.x; i < G::CQ; i += blockDim.x) Cq[i] = cv;
}

// Column sums of an accumulator tile (q_bias / v_bias gradients) WITHOUT 32 registers that live through the chunk loop: per sample
// the 32 lanes of each half-wave are summed by DPP (row_shr 1 / 2 / 4 / 8: lane 15 of a row holds its total; row_bcast:15 into
// rows 1 and 3: lanes 31 / 63 hold the half's total) and lanes 31 / 63 add four columns at a time to this wave's private row
// of 64 floats in LDS (no atomics: one writer per word).  ~170 vector instructions per sample and wave.
__device__ __forceinline__ float half_sum_dpp(float v) {
#define WIN_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true))
  WIN_DPP_ADD(0x111, 0xf);
  WIN_DPP_ADD(0x112, 0xf);
  WIN_DPP_ADD(0x114, 0xf);
  WIN_DPP_ADD(0x118, 0xf);
  WIN_DPP_ADD(0x142, 0xa);
#undef WIN_DPP_ADD
  return v;
}
__device__ __forceinline__ void colsum_add4(float* row64, int r, int hh, int db, int g, float a0, float a1, float a2, float a3) {
  a0 = half_sum_dpp(a0); a1 = half_sum_dpp(a1); a2 = half_sum_dpp(a2); a3 = half_sum_dpp(a3);
  if (r == 31) {
    float4* p = reinterpret_cast<float4*>(row64 + db * 32 + 8 * g + 4 * hh);
    float4 t = *p;
    t.x += a0; t.y += a1; t.z += a2; t.w += a3;
    *p = t;
  }
}


template <typename K>
int set_lds_attr(K kernel, bool* done) {
  if (!*done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    *done = true;
  }
  return MEMHIP_OK;
}

}  // namespace
