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

// Stage the transposed copy dst[d][tok] (row stride VS) of a [T][64] head slice that sits at
// src + tok*ld (tokens >= T are zero-filled up to TP).
__device__ __forceinline__ void stage_transposed(__bf16* dst, int VS, const __bf16* src, long long ld, int T,
                                                 int TP) {
  for (int idx = threadIdx.x; idx < TP * 8; idx += blockDim.x) {
    const int tok = idx >> 3, dc = idx & 7;
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)0.f;
    if (tok < T) v = ld16(src + (long long)tok * ld + dc * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[(dc * 8 + i) * VS + tok] = v[i];
  }
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------- forward
template <int NKB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const __bf16* __restrict__ qkv, long long ldq, int T,
                                                       int D, int H, const float* __restrict__ bias,
                                                       __bf16* __restrict__ out, long long ldo,
                                                       float* __restrict__ lse) {
  constexpr int TP = NKB * 32, VS = TP + 4;
  __shared__ __attribute__((aligned(16))) __bf16 Vt[HD * VS];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const __bf16* base = qkv + (long long)b * T * ldq + h * HD;      // q slice of this head
  stage_transposed(Vt, VS, base + 2 * D, ldq, T, TP);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  for (int qb = wave; qb < NKB; qb += 4) {
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
      const int kr = kb * 32 + r;
      const int krc = kr < T ? kr : T - 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 Kf = ld16(base + D + (long long)krc * ldq + 16 * t + 8 * hh);
        s[kb] = MFMA32(Kf, Qf[t], s[kb]);
      }
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
        for (int db = 0; db < 2; ++db) {
          const __bf16* vrow = Vt + (db * 32 + r) * VS + key0;
          o[db] = MFMA32(cat4(vrow, vrow + 8), pf, o[db]);
        }
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
__global__ __launch_bounds__(256) void attn_bwd_kernel(const __bf16* __restrict__ qkv, long long ldq,
                                                       const __bf16* __restrict__ dout,
                                                       const __bf16* __restrict__ out, long long ldo,
                                                       const float* __restrict__ lse,
                                                       const float* __restrict__ bias,
                                                       const int* __restrict__ relidx, int nrd,
                                                       __bf16* __restrict__ dqkv, long long lddq,
                                                       float* __restrict__ dtable, int T, int D, int H,
                                                       float scale) {
  constexpr int TP = NKB * 32, VS = TP + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16* Qt = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* dOt = Qt + HD * VS;
  __bf16* Kt = dOt + HD * VS;
  float* lseS = reinterpret_cast<float*>(Kt + HD * VS);
  float* delS = lseS + TP;
  float* bins = delS + TP;

  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long long row0 = (long long)b * T;
  const __bf16* qb_ = qkv + row0 * ldq + h * HD;          // Q' slice (already scaled)
  const __bf16* kb_ = qb_ + D;
  const __bf16* vb_ = qb_ + 2 * D;
  const __bf16* dob = dout + row0 * ldo + h * HD;
  const __bf16* ob = out + row0 * ldo + h * HD;

  stage_transposed(Qt, VS, qb_, ldq, T, TP);
  stage_transposed(dOt, VS, dob, ldo, T, TP);
  stage_transposed(Kt, VS, kb_, ldq, T, TP);
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
  for (int kb = wave; kb < NKB; kb += 4) {
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
    for (int qb = 0; qb < NKB; ++qb) {
      const int qr = qb * 32 + r;
      const int qc = qr < T ? qr : T - 1;
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 Qf = ld16(qb_ + (long long)qc * ldq + 16 * t + 8 * hh);
        const bf16x8 dOf = ld16(dob + (long long)qc * ldo + 16 * t + 8 * hh);
        S = MFMA32(Qf, Kf[t], S);
        dP = MFMA32(dOf, Vf[t], dP);
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
          const __bf16* a1 = dOt + (db * 32 + r) * VS + q0;
          const __bf16* a2 = Qt + (db * 32 + r) * VS + q0;
          dVt[db] = MFMA32(cat4(a1, a1 + 8), pf, dVt[db]);
          dKt[db] = MFMA32(cat4(a2, a2 + 8), dsf, dKt[db]);
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
  for (int qb = wave; qb < NKB; qb += 4) {
    const int q = qb * 32 + r;
    const int qc = q < T ? q : T - 1;
    bf16x8 Qf[4], dOf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Qf[t] = ld16(qb_ + (long long)qc * ldq + 16 * t + 8 * hh);
      dOf[t] = ld16(dob + (long long)qc * ldo + 16 * t + 8 * hh);
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
      const int kc = kr < T ? kr : T - 1;
      f32x16 St, dPt;
#pragma unroll
      for (int i = 0; i < 16; ++i) { St[i] = 0.f; dPt[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 Kf = ld16(kb_ + (long long)kc * ldq + 16 * t + 8 * hh);
        const bf16x8 Vf = ld16(vb_ + (long long)kc * ldq + 16 * t + 8 * hh);
        St = MFMA32(Kf, Qf[t], St);
        dPt = MFMA32(Vf, dOf[t], dPt);
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
        for (int db = 0; db < 2; ++db) {
          const __bf16* a = Kt + (db * 32 + r) * VS + k0;
          dQt[db] = MFMA32(cat4(a, a + 8), dsf, dQt[db]);
        }
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
  constexpr int TP = NKB * 32, VS = TP + 4;
  return (size_t)3 * HD * VS * 2 + (size_t)2 * TP * 4 + (size_t)nrd * 4 + 16;
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
#define FWD(N) hipLaunchKernelGGL(attn_fwd_kernel<N>, dim3(B * heads), dim3(256), 0, s, (const __bf16*)qkv, \
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
#define BWD(N)                                                                                         \
  {                                                                                                    \
    const size_t sm = bwd_smem<N>(nrd);                                                                \
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<N>),                  \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));              \
    hipLaunchKernelGGL(attn_bwd_kernel<N>, dim3(B * heads), dim3(256), sm, s, (const __bf16*)qkv,      \
                       (long long)ldqkv, (const __bf16*)dout, (const __bf16*)out, (long long)ldo, lse, \
                       bias_pad, relidx_pad, nrd, (__bf16*)dqkv, (long long)lddqkv, dtable, T, D, heads, scale); \
  }
  ATTN_DISPATCH(nkb, BWD)
#undef BWD
  return check_launch("attn_bwd");
}
