// Fused multi-head attention with the shared additive relative-position bias, forward and
// backward (reference: Attention.forward, mem/modeling_finetune.py:137-154, with the bias of
// RelativePositionBias :242-247 broadcast over the batch).  head_dim = 64 (ViT-B and ViT-L).
//
// CDNA4 mapping.  One workgroup (4 waves) per (sample, head); the whole 197-token problem stays
// on chip, nothing of the [B,H,N,N] score tensor ever reaches HBM.  Scores are computed
// TRANSPOSED (S^T = K Q^T, v_mfma_f32_32x32x16_bf16) so that a lane owns one query column and a
// softmax row reduction is in-lane + one cross-half shuffle; the fp32 accumulator tile is then
// already the B operand of the next product (O^T = V^T P^T, and in backward dV^T, dK^T, dQ^T) --
// no LDS round trip for P / dS.  Q/K/V/dO fragments for the row-operand side are 16-byte loads
// straight from the token-major qkv buffer; the operands that must be read "down a column" come
// from transposed bf16 copies staged once per workgroup in LDS (row stride = keys + 4 elements:
// odd multiple of 8 B => conflict-free ds_read_b64).  The relative-position-bias gradient is
// reduced on chip into the 732-bucket table with LDS float atomics and flushed once.
//
// Rounding points follow the reference under autocast: q k^T and the P V / dO V^T / dS K products
// are rounded to bf16, bias add + softmax (+ its backward) run in fp32, P and dS are rounded to
// bf16 when they feed an MFMA.
#include <cstdlib>
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
      for (int t = 0; t < 4; ++t) s[kb] = MFMA32(row_frag(Ks, kb * 32 + r, 2 * t + hh), Qf[t], s[kb]);
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
        const int key0 = kb * 32 + 16 * ss + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(col_frag(Vs, key0, db, lane), pf, o[db]);
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
template <int NKB>
__global__ __launch_bounds__(512) void attn_bwd_kernel(const __bf16* __restrict__ qkv, long long ldq,
                                                       const __bf16* __restrict__ dout,
                                                       const __bf16* __restrict__ out, long long ldo,
                                                       const float* __restrict__ lse,
                                                       const float* __restrict__ bias,
                                                       const int* __restrict__ relidx, int nrd,
                                                       __bf16* __restrict__ dqkv, long long lddq,
                                                       float* __restrict__ dtable, int T, int D, int H,
                                                       float scale, int dbg) {
  constexpr int TP = NKB * 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* Qs = smem_raw;
  char* Ks = Qs + TP * 128;
  char* Vs = Ks + TP * 128;
  char* dOs = Vs + TP * 128;
  float* lseS = reinterpret_cast<float*>(dOs + TP * 128);
  float* delS = lseS + TP;
  float* bins = delS + TP;

  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long long row0 = (long long)b * T;
  const __bf16* qb_ = qkv + row0 * ldq + h * HD;          // Q' slice (already scaled)
  const __bf16* kb_ = qb_ + D;
  const __bf16* vb_ = qb_ + 2 * D;
  const __bf16* dob = dout + row0 * ldo + h * HD;
  const __bf16* ob = out + row0 * ldo + h * HD;

  stage_head(Qs, qb_, ldq, T, TP);
  stage_head(Ks, kb_, ldq, T, TP);
  stage_head(Vs, vb_, ldq, T, TP);
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
  }
  for (int i = threadIdx.x; i < nrd; i += blockDim.x) bins[i] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;

  // ---- phase A: a wave owns 32 keys, sweeps the queries: dV, dK   (tiles are [q rows][key cols])
  if (!(dbg & 1))
  for (int kb = wave; kb < NKB; kb += 8) {
    const int key = kb * 32 + r;
    bf16x8 Kf[4], Vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Kf[t] = row_frag(Ks, key, 2 * t + hh);
      Vf[t] = row_frag(Vs, key, 2 * t + hh);
    }
    f32x16 dVt[2], dKt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dVt[db][i] = 0.f; dKt[db][i] = 0.f; }
    for (int qb = 0; qb < NKB; ++qb) {
      const int qr = qb * 32 + r;
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        S = MFMA32(row_frag(Qs, qr, 2 * t + hh), Kf[t], S);
        dP = MFMA32(row_frag(dOs, qr, 2 * t + hh), Vf[t], dP);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int q = qb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float sv = bfr(S[i]) + bias[((long long)h * TP + q) * TP + key];
        const float p = (q < T && key < T) ? __expf(sv - lseS[q]) : 0.f;
        S[i] = p;
        dP[i] = p * (bfr(dP[i]) - delS[q]);
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = acc_frag(S, ss, 1.0f), dsf = acc_frag(dP, ss, 1.0f);
        const int q0 = qb * 32 + 16 * ss + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dVt[db] = MFMA32(col_frag(dOs, q0, db, lane), pf, dVt[db]);
          dKt[db] = MFMA32(col_frag(Qs, q0, db, lane), dsf, dKt[db]);
        }
      }
    }
    if (key < T) {
      __bf16* drow = dqkv + (row0 + key) * lddq + h * HD;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 wv, wk;
#pragma unroll
          for (int e = 0; e < 4; ++e) { wv[e] = (__bf16)dVt[db][4 * g + e]; wk[e] = (__bf16)dKt[db][4 * g + e]; }
          *reinterpret_cast<bf16x4*>(drow + 2 * D + db * 32 + 8 * g + 4 * hh) = wv;
          *reinterpret_cast<bf16x4*>(drow + D + db * 32 + 8 * g + 4 * hh) = wk;
        }
    }
  }

  // ---- phase B: a wave owns 32 queries, sweeps the keys: dQ, dBias   (tiles are [key rows][q cols])
  if (!(dbg & 2))
  for (int qb = wave; qb < NKB; qb += 8) {
    const int q = qb * 32 + r;
    bf16x8 Qf[4], dOf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Qf[t] = row_frag(Qs, q, 2 * t + hh);
      dOf[t] = row_frag(dOs, q, 2 * t + hh);
    }
    const float lq = lseS[q], dq_ = delS[q];
    f32x16 dQt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) dQt[db][i] = 0.f;
    const float* brow = bias + ((long long)h * TP + q) * TP;
    const int* irow = relidx + (long long)q * TP;
    for (int kb = 0; kb < NKB; ++kb) {
      const int kr = kb * 32 + r;
      f32x16 St, dPt;
#pragma unroll
      for (int i = 0; i < 16; ++i) { St[i] = 0.f; dPt[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        St = MFMA32(row_frag(Ks, kr, 2 * t + hh), Qf[t], St);
        dPt = MFMA32(row_frag(Vs, kr, 2 * t + hh), dOf[t], dPt);
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
          if (ok && dtable) atomicAdd(bins + ii[e], ds);
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 dsf = acc_frag(dPt, ss, 1.0f);
        const int k0 = kb * 32 + 16 * ss + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db) dQt[db] = MFMA32(col_frag(Ks, k0, db, lane), dsf, dQt[db]);
      }
    }
    if (q < T) {
      __bf16* drow = dqkv + (row0 + q) * lddq + h * HD;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (__bf16)(bfr(dQt[db][4 * g + e]) * scale);
          *reinterpret_cast<bf16x4*>(drow + db * 32 + 8 * g + 4 * hh) = w;
        }
    }
  }
  if (dtable) {
    __syncthreads();
    for (int i = threadIdx.x; i < nrd; i += blockDim.x) {
      const float v = bins[i];
      if (v != 0.f) atomicAdd(dtable + (long long)i * H + h, v);
    }
  }
}

template <int NKB>
size_t bwd_smem(int nrd) {
  constexpr int TP = NKB * 32;
  return (size_t)4 * TP * 128 + (size_t)2 * TP * 4 + (size_t)nrd * 4 + 16;
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
                               const float* lse, const float* bias_pad, const int32_t* relidx_pad,
                               int num_rel, int B, int T, int D, int heads, float scale, void* dqkv,
                               int64_t lddqkv, float* dtable, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && heads > 0 && D == heads * HD, "attn_bwd: head_dim must be 64");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && dout && out && lse && bias_pad && dqkv, "attn_bwd: null pointer");
  MEMHIP_REQUIRE(!dtable || (relidx_pad && num_rel > 0), "attn_bwd: dtable needs relidx_pad");
  MEMHIP_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0 && lddqkv % 8 == 0, "attn_bwd: ld must be a multiple of 8");
  hipStream_t s = as_stream(stream);
  const int nkb = (T + 31) / 32;
  const int nrd = dtable ? num_rel : 0;
  static const int dbg = getenv("MEMHIP_ATTN_DBG") ? atoi(getenv("MEMHIP_ATTN_DBG")) : 0;
#define BWD(N)                                                                                         \
  {                                                                                                    \
    const size_t sm = bwd_smem<N>(nrd);                                                                \
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<N>),                  \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));              \
    hipLaunchKernelGGL(attn_bwd_kernel<N>, dim3(B * heads), dim3(512), sm, s, (const __bf16*)qkv,      \
                       (long long)ldqkv, (const __bf16*)dout, (const __bf16*)out, (long long)ldo, lse, \
                       bias_pad, relidx_pad, nrd, (__bf16*)dqkv, (long long)lddqkv, dtable, T, D, heads, scale, dbg); \
  }
  ATTN_DISPATCH(nkb, BWD)
#undef BWD
  return check_launch("attn_bwd");
}
