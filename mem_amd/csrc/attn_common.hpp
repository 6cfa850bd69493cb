// Shared device helpers of the attention kernels (attn.hip: <= 256 tokens held on chip; attn_stream.hip: longer
// sequences streamed in chunks): LDS image layout + LDS-DMA staging, MFMA fragment reads, the extended
// relative-position table.  Everything has internal linkage (each translation unit gets its own copy).
#pragma once
#include "common.h"

namespace {

using namespace memhip;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int HD = 64;   // head dim
constexpr int kMaxLds = 160 * 1024;   // bytes of LDS one workgroup may use on gfx950

__device__ __forceinline__ float bfr(float v) { return (float)(__bf16)v; }
// round two fp32 values to bf16 precision with one packed convert (v_cvt_pk_bf16_f32 + shift + and)
template <typename V>
__device__ __forceinline__ void bfr2(V& x, int i) {             // rounds x[i], x[i+1]
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t pk = __builtin_convertvector(f32x2_t{x[i], x[i + 1]}, bf16x2_t);
  const unsigned u = __builtin_bit_cast(unsigned, pk);
  x[i] = __uint_as_float(u << 16);
  x[i + 1] = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }   // bare v_exp_f32
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }    // bare v_log_f32
__device__ __forceinline__ bf16x8 ld16(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// bf16 fragment (k-step s) of an fp32 accumulator tile, scaled: regs 8s..8s+7
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s, float mul) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)(x[8 * s + j] * mul);
  return r;
}

__device__ __attribute__((aligned(256))) unsigned char g_attn_zero_page[128];   // zero-initialised

// LDS-DMA (global_load_lds_dwordx4) as INLINE ASM: through the builtin, hipcc (ROCm 7.2) knows an
// LDS write is pending on the VM counter and puts s_waitcnt vmcnt(0) in front of the next LDS read
// it cannot disambiguate -- i.e. the first fragment read of the sample drained the prefetch of the
// next sample that had just been issued.  The asm form is invisible to the waitcnt pass; the kernels
// wait explicitly (ATTN_DMA_WAIT) in front of the barrier that publishes the images.
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
  const unsigned lds = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)lds_dst));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds) : "memory", "m0");
}
#define ATTN_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// LDS image of a [TP tokens][64] bf16 head slice: 128-B rows, 16-B chunk c of token t lives at chunk position
// c ^ img_key(t).  img_key(t) = the three bits of (t >> 1) rotated (bit 0 -> bit 2): any permutation of these bits keeps
// the ds_read_b128 row fragments conflict-free (16-lane groups, 8 tokens of one parity per group, 8 distinct keys); the
// rotation also spreads the transposing column reads (ds_read_b64_tr_b16: 32-lane groups = 4 consecutive tokens x 4 chunks
// x 2 halves) over all 64 banks -- with the plain key (t >> 1) & 7 tokens t and t + 2 of a group met chunks c and c ^ 1 in
// the same banks (2-way conflict on every column read: tools/attn_pmc.sh, SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int img_key(int tok) {
  const int k = (tok >> 1) & 7;
  return ((k & 1) << 2) | (k >> 1);
}
__device__ __forceinline__ int tok_slot(int tok, int chunk) { return tok * 8 + (chunk ^ img_key(tok)); }

// Stage src[tok*ld + 0..63] (tok < T, zero beyond) with LDS-DMA: one wave-instruction = 8 tokens.
__device__ __forceinline__ void stage_head(char* dst, const __bf16* src, long long ld, int T, int TP) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int inst = wave; inst < TP / 8; inst += nw) {
    const int tok = inst * 8 + (lane >> 3), cpos = lane & 7;
    const int chunk = cpos ^ img_key(tok);
    const void* g = tok < T ? (const void*)(src + (long long)tok * ld + chunk * 8)
                            : (const void*)(g_attn_zero_page + cpos * 16);
    glds16(g, dst + inst * 1024);
  }
}

// Per-lane byte offsets inside a head image for token block 0; token block kb adds the constant
// kb*4096 (the XOR term only depends on the token's low 5 bits): base + immediate addressing.
struct LaneOffs {
  int row[4];        // row fragment: token kb*32 + r, chunk 2t + hh
  int col[2][2][2];  // column fragment: tokens kb*32 + 16ss + 4hh + {0..3, 8..11}, d block db: [ss][db][lo/hi]
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
// Column fragments use the transposing LDS read as INLINE ASM: with the builtin, hipcc (ROCm 7.2)
// cannot tell these reads from the in-flight LDS-DMA of the next sample and drains it (s_waitcnt
// vmcnt(0)) in front of the first read -- the prefetch then overlaps nothing.  The asm form is
// invisible to the compiler's waitcnt pass: whoever consumes a fragment waits with LDS_TR_WAIT().
__device__ __forceinline__ s16x4 lds_tr16_b64(unsigned lds_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
  return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return (unsigned)(unsigned long long)((__attribute__((address_space(3))) const char*)p);
}
#define LDS_TR_WAIT()                                    \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
__device__ __forceinline__ bf16x8 col_frag_o(const char* img, const LaneOffs& o, int kb, int ss, int db) {
  const unsigned base = lds_addr_of(img) + kb * 4096;
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lds_tr16_b64(base + o.col[ss][db][0]);
  u.s.h = lds_tr16_b64(base + o.col[ss][db][1]);
  return u.v;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// ---- relative-position bias from the head's table in LDS (mem/modeling_finetune.py:224-240):
//   tokens 1.. are the Wh x Ww grid, token 0 is cls.  bucket(q,k) = (qy-ky+Wh-1)*(2Ww-1) + (qx-kx+Ww-1)
//   = (Kc(q) + off) - Kc(k) with Kc(t) = ty*(2Ww-1) + tx, off = (Wh-1)*(2Ww-1) + (Ww-1); cls row ->
//   nrd-3, cls column -> nrd-2, (cls,cls) -> nrd-1.
// To keep the per-element work at ONE subtraction + ONE LDS gather (no selects), the three cls
// buckets are reached arithmetically as well: codeQ(cls) = 4off+2 and codeK(cls) = -(off+1) push
// the difference codeQ(q) - codeK(k) into disjoint regions of an EXTENDED table
//   [0, 2off]           the (2Wh-1)(2Ww-1) grid buckets     (codeQ in [off, 2off], codeK in [0, off])
//   [2off+1, 3off+1]    key = cls      (all = table[nrd-2])
//   [3off+2, 4off+2]    query = cls    (all = table[nrd-3])
//   5off+3              both cls       (= table[nrd-1])
// Codes are stored pre-multiplied by 4 (byte offsets).  The table is stored times log2(e) so that
// the softmax runs on exp2.
struct RelGeom { int off, len; };
__device__ __host__ __forceinline__ RelGeom rel_geom(int Wh, int Ww) {
  const int off = (Wh - 1) * (2 * Ww - 1) + (Ww - 1);
  return RelGeom{off, (5 * off + 4 + 3) & ~3};   // padded to 16 bytes: the arrays laid out behind it are read as b128
}
// target bucket of extended index i
__device__ __forceinline__ int rel_target(int i, int off, int nrd) {
  if (i <= 2 * off) return i;
  if (i <= 3 * off + 1) return nrd - 2;
  if (i <= 4 * off + 2) return nrd - 3;
  return nrd - 1;                       // 4off+3 .. 5off+2 are never addressed; 5off+3 = (cls, cls)
}
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

__device__ __forceinline__ void rel_setup(float* tabX, int* codeQ, int* codeK, const float* table, int nrd, int H,
                                          int h, int T, int TP, int Wh, int Ww, float mul) {
  const RelGeom g = rel_geom(Wh, Ww);
  for (int i = threadIdx.x; i < g.len; i += blockDim.x) {
    const int t = rel_target(i, g.off, nrd);
    tabX[i] = table[(long long)t * H + h] * mul;
  }
  for (int t = threadIdx.x; t < TP; t += blockDim.x) {
    int cq = g.off, ck = 0;                                   // padding tokens: any in-range value
    if (t == 0) { cq = 4 * g.off + 2; ck = -(g.off + 1); }
    else if (t < T) { const int u = t - 1; ck = (u / Ww) * (2 * Ww - 1) + (u % Ww); cq = ck + g.off; }
    codeQ[t] = 4 * cq;
    codeK[t] = 4 * ck;
  }
}
// Gather from an ABSOLUTE LDS byte address.  The table's base is folded into the per-lane code once per sample;
// through a generic `base + offset` hipcc emits a separate v_add_u32 of the (link-time) LDS base in front of
// every ds_read_b32 -- one VALU instruction per score element in kernels that are VALU-issue bound.
__device__ __forceinline__ float lds_f32_abs(int lds_byte_addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) float*>((unsigned)lds_byte_addr);
}
// round(x * scale) to int with ONE fma + ONE integer subtract: adding 1.5 * 2^23 leaves the rounded (nearest-even)
// integer in the low mantissa bits for |x * scale| < 2^22 (v_mul + v_rndne + v_cvt_i32 are three instructions)
__device__ __forceinline__ int fx_round(float x, float scale) {
  return __float_as_int(fmaf(x, scale, 12582912.0f)) - 0x4B400000;
}
__device__ __forceinline__ void lds_add_i32_abs(int lds_byte_addr, int v) {       // ds_add_u32, no return
  __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) int*>((unsigned)lds_byte_addr), v,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float lds_f32_at(const float* base, int byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

}  // namespace
