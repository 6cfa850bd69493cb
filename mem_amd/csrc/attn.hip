// Fused multi-head attention with the shared additive relative-position bias, forward and
// backward (reference: Attention.forward, mem/modeling_finetune.py:137-154, with the bias of
// RelativePositionBias :213-247 broadcast over the batch).  head_dim = 64 (ViT-B and ViT-L),
// up to 256 tokens (longer sequences: attn_stream.hip).
//
// CDNA4 mapping.  The whole 197-token problem of one (sample, head) stays on chip; nothing of the
// [B,H,N,N] score tensor ever reaches HBM.  Scores are computed TRANSPOSED (S^T = K Q^T,
// v_mfma_f32_32x32x16_bf16) so that a lane owns one query column and a softmax row reduction is
// in-lane + one cross-half shuffle; the fp32 accumulator tile is then already the B operand of the
// next product (O^T = V^T P^T; in backward dV^T, dK^T, dQ^T) -- no LDS round trip for P / dS.
// Head slices are staged as row-major XOR-swizzled LDS images by LDS-DMA; "row" fragments are
// ds_read_b128, fragments that run down a column use the transposing read ds_read_b64_tr_b16.
//
// Latency structure (what the first version got wrong: 60 % of wave time was s_waitcnt):
//   * every kernel is PERSISTENT over `spb` consecutive samples of one head with double-buffered
//     LDS images: the LDS-DMA of sample b+1 and the few per-wave register fragments of sample b+1
//     are issued before sample b is computed;
//   * the compute phase contains NO global loads: the additive bias is not read from a [H,N,N]
//     tensor but gathered from the head's 732-entry table held in LDS, with the bucket index
//     computed arithmetically (index(q,k) = C(q) - Kc(k), three special cls buckets) -- so nothing
//     forces the in-flight prefetch to drain (vmcnt completes in order on CDNA4);
//   * delta = rowsum(dO * O) arrives precomputed (fused into the proj-dgrad GEMM epilogue).
// The bias-table gradient is bucketed with fixed-point INTEGER LDS atomics (ds_add_f32 costs ~190
// cycles per wave-instruction on gfx950, ds_add_u32 ~6: tools/micro/lds_atomic.hip).
//
// Rounding points follow the reference under autocast: q k^T and the P V / dO V^T / dS K products
// are rounded to bf16, bias add + softmax (+ its backward) run in fp32, P and dS are rounded to
// bf16 when they feed an MFMA.
#include "attn_common.hpp"

namespace {


// ------------------------------------------------------------------------------- forward
template <int NKB>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const __bf16* __restrict__ qkv, long long ldq, int B, int T,
                                                       int D, int H, const float* __restrict__ table, int nrd,
                                                       int Wh, int Ww, __bf16* __restrict__ out, long long ldo,
                                                       float* __restrict__ lse, int spb) {
  constexpr int TP = NKB * 32;
  constexpr int IMG = TP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RelGeom geo = rel_geom(Wh, Ww);
  float* tabX = reinterpret_cast<float*>(smem);                // extended bias table * log2(e), at LDS offset 0
  int* codeQ = reinterpret_cast<int*>(tabX + geo.len);
  int* codeK = codeQ + TP;
  char* imgs = smem + (((geo.len + 2 * TP) * 4 + 15) & ~15);
  const int h = blockIdx.x % H, b0 = (blockIdx.x / H) * spb;
  const int b1 = b0 + spb < B ? b0 + spb : B;
  if (b0 >= b1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  rel_setup(tabX, codeQ, codeK, table, nrd, H, h, T, TP, Wh, Ww, kLog2e);
  const int qb = wave;                         // NKB <= 8: one 32-query block per wave
  const bool active = qb < NKB;
  const int q = qb * 32 + r;
  const int qc = q < T ? q : T - 1;
  {
    const __bf16* s0 = qkv + (long long)b0 * T * ldq + h * HD;
    stage_head(imgs, s0 + D, ldq, T, TP);
    stage_head(imgs + IMG, s0 + 2 * D, ldq, T, TP);
  }
  bf16x8 Qn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) Qn[t] = ld16(qkv + ((long long)b0 * T + qc) * ldq + h * HD + 16 * t + 8 * hh);
  for (int b = b0; b < b1; ++b) {
    const int cur = (b - b0) & 1;
    const char* Ks = imgs + cur * 2 * IMG;
    const char* Vs = Ks + IMG;
    bf16x8 Qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Qf[t] = Qn[t];
    ATTN_DMA_WAIT();
    __syncthreads();                         // sample b's images landed; sample b-1 fully consumed
    if (b + 1 < b1) {                        // prefetch sample b+1: register fragments first, then LDS-DMA
      const __bf16* s1 = qkv + (long long)(b + 1) * T * ldq + h * HD;
#pragma unroll
      for (int t = 0; t < 4; ++t) Qn[t] = ld16(s1 + (long long)qc * ldq + 16 * t + 8 * hh);
      stage_head(imgs + (cur ^ 1) * 2 * IMG, s1 + D, ldq, T, TP);
      stage_head(imgs + (cur ^ 1) * 2 * IMG + IMG, s1 + 2 * D, ldq, T, TP);
    }
    if (!active) continue;
    const int cq4 = codeQ[qc] + (int)lds_addr_of(reinterpret_cast<const char*>(tabX));   // absolute LDS address of the lane's table window
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) s[kb] = MFMA32(row_frag_o(Ks, lo, kb, t), Qf[t], s[kb]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key0 = kb * 32 + 8 * g + 4 * hh;
        const int4 kc = *reinterpret_cast<const int4*>(codeK + key0);
        const int kcs[4] = {kc.x, kc.y, kc.z, kc.w};
        bfr2(s[kb], 4 * g);
        bfr2(s[kb], 4 * g + 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // log2-domain score: (q.k) * log2e + bias * log2e   (one gather, one fma)
          float v = fmaf(s[kb][4 * g + e], kLog2e, lds_f32_abs(cq4 - kcs[e]));
          if (kb == NKB - 1 && key0 + e >= T) v = -INFINITY;
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
        // groups of 8 keys that lie completely in the padding of the last block (27 of its 32 keys at 197 tokens):
        // wave-uniform skip of the exp work
        if (kb == NKB - 1 && kb * 32 + 8 * (i >> 2) >= T) { s[kb][i] = 0.f; continue; }
        const float p = fexp2(s[kb][i] - mx);
        s[kb][i] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (hh == 0 && q < T) lse[((long long)b * H + h) * TP + q] = (mx + flog2(sum)) * kLn2;
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      bf16x8 vf[2][2];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int db = 0; db < 2; ++db) vf[ss][db] = col_frag_o(Vs, lo, kb, ss, db);
      bf16x8 pf[2];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) pf[ss] = acc_frag(s[kb], ss, inv);
      LDS_TR_WAIT();
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(vf[ss][db], pf[ss], o[db]);
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

// ------------------------------------------------------------------------------- backward (dK, dV)
// a wave owns 32 keys and sweeps the queries with tiles [q rows][key cols]
template <int NKB, bool VB>
__global__ __launch_bounds__(512) void attn_bwd_kv_kernel(const __bf16* __restrict__ qkv, long long ldq,
                                                          const __bf16* __restrict__ dout, long long ldo,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta,
                                                          float* __restrict__ stats,
                                                          const float* __restrict__ table, int nrd, int Wh, int Ww,
                                                          __bf16* __restrict__ dqkv, long long lddq,
                                                          float* __restrict__ dvbias, int B, int T, int D, int H,
                                                          int spb) {
  constexpr int TP = NKB * 32;
  constexpr int IMG = TP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RelGeom geo = rel_geom(Wh, Ww);
  float* tabX = reinterpret_cast<float*>(smem);
  int* codeQ = reinterpret_cast<int*>(tabX + geo.len);
  int* codeK = codeQ + TP;
  float* lseS = reinterpret_cast<float*>(codeK + TP);       // [2][TP]  (log2 domain)
  float* delS = lseS + 2 * TP;                              // [2][TP]
  float* vsum = delS + 2 * TP;                              // [64]
  char* imgs = smem + (((geo.len + 6 * TP + HD) * 4 + 15) & ~15);
  const int h = blockIdx.x % H, b0 = (blockIdx.x / H) * spb;
  const int b1 = b0 + spb < B ? b0 + spb : B;
  if (b0 >= b1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  rel_setup(tabX, codeQ, codeK, table, nrd, H, h, T, TP, Wh, Ww, kLog2e);
  if (threadIdx.x < HD) vsum[threadIdx.x] = 0.f;
  const int kb = wave;
  const bool active = kb < NKB;
  const int key = kb * 32 + r;
  const int kc_tok = key < T ? key : T - 1;
  auto stage_sample = [&](int b, int buf) {
    const __bf16* s = qkv + (long long)b * T * ldq + h * HD;
    stage_head(imgs + buf * 2 * IMG, s, ldq, T, TP);                                        // Q'
    stage_head(imgs + buf * 2 * IMG + IMG, dout + (long long)b * T * ldo + h * HD, ldo, T, TP);   // dO
  };
  bf16x8 Kn[4], Vn[4];
  float bsum[VB ? 32 : 1];
#pragma unroll
  for (int i = 0; i < (VB ? 32 : 1); ++i) bsum[i] = 0.f;
  float vmax = 0.f, dmax = 0.f, nmax = 0.f;
  float lsen = 0.f, deln = 0.f;                             // this thread's element of the next lse/delta rows
  auto load_next = [&](int b) {
    const __bf16* s = qkv + ((long long)b * T + kc_tok) * ldq + h * HD;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Kn[t] = ld16(s + D + 16 * t + 8 * hh);
      Vn[t] = ld16(s + 2 * D + 16 * t + 8 * hh);
    }
    const int qq = threadIdx.x;
    lsen = (qq < T) ? lse[((long long)b * H + h) * TP + qq] * kLog2e : 0.f;
    deln = (qq < T) ? delta[((long long)b * T + qq) * H + h] : 0.f;
    if (stats && qq < T) nmax = fmaxf(nmax, delta[((long long)B * T + (long long)b * T + qq) * H + h]);   // |dO_q|^2
  };
  load_next(b0);
  stage_sample(b0, 0);
  for (int b = b0; b < b1; ++b) {
    const int cur = (b - b0) & 1;
    const char* Qs = imgs + cur * 2 * IMG;
    const char* dOs = Qs + IMG;
    bf16x8 Kf[4], Vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { Kf[t] = Kn[t]; Vf[t] = Vn[t]; }
    if (stats) {
      // bounds for the fixed-point table-gradient buckets of attn_bwd_q_kernel (runs after this kernel):
      // max_k |V_k|^2 (half a row per lane; padding keys repeat key T-1) and max_q |delta_q|
      float vn = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) vn = fmaf((float)Vf[t][i], (float)Vf[t][i], vn);
      vn += __shfl_xor(vn, 32);
      vmax = fmaxf(vmax, vn);
      dmax = fmaxf(dmax, fabsf(deln));
    }
    if ((int)threadIdx.x < TP) { lseS[cur * TP + threadIdx.x] = lsen; delS[cur * TP + threadIdx.x] = deln; }
    ATTN_DMA_WAIT();
    __syncthreads();
    if (b + 1 < b1) { load_next(b + 1); stage_sample(b + 1, cur ^ 1); }
    if (!active) continue;
    const int ck4 = codeK[kc_tok] - (int)lds_addr_of(reinterpret_cast<const char*>(tabX));   // (folds the table's LDS base)
    const float kmask = key < T ? 1.f : 0.f;
    const float* lseC = lseS + cur * TP;
    const float* delC = delS + cur * TP;
    f32x16 dVt[2], dKt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dVt[db][i] = 0.f; dKt[db][i] = 0.f; }
#pragma unroll
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
        if (qb == NKB - 1 && qb * 32 + 8 * g >= T) {          // 8 padded queries: P = dS = 0 (wave-uniform skip)
#pragma unroll
          for (int e = 0; e < 4; ++e) { S[4 * g + e] = 0.f; dP[4 * g + e] = 0.f; }
          continue;
        }
        const int4 qcv = *reinterpret_cast<const int4*>(codeQ + q0);
        const float4 lv = *reinterpret_cast<const float4*>(lseC + q0);
        const float4 dv = *reinterpret_cast<const float4*>(delC + q0);
        const int qcs[4] = {qcv.x, qcv.y, qcv.z, qcv.w};
        const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
        bfr2(S, 4 * g);
        bfr2(S, 4 * g + 2);
        bfr2(dP, 4 * g);
        bfr2(dP, 4 * g + 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          const float sv = fmaf(S[i], kLog2e, lds_f32_abs(qcs[e] - ck4));
          float p = fexp2(sv - ll[e]) * kmask;                 // kmask = 0 for padding keys
          if (qb == NKB - 1 && q0 + e >= T) p = 0.f;
          S[i] = p;
          dP[i] = p * (dP[i] - dd[e]);
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        bf16x8 cdo[2], cq[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          cdo[db] = col_frag_o(dOs, lo, qb, ss, db);
          cq[db] = col_frag_o(Qs, lo, qb, ss, db);
        }
        const bf16x8 pf = acc_frag(S, ss, 1.0f), dsf = acc_frag(dP, ss, 1.0f);
        LDS_TR_WAIT();
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dVt[db] = MFMA32(cdo[db], pf, dVt[db]);
          dKt[db] = MFMA32(cq[db], dsf, dKt[db]);
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
          __bf16* drow = dqkv + ((long long)b * T + key) * lddq + h * HD;
          *reinterpret_cast<bf16x4*>(drow + 2 * D + db * 32 + 8 * g + 4 * hh) = wv;
          *reinterpret_cast<bf16x4*>(drow + D + db * 32 + 8 * g + 4 * hh) = wk;
        }
        // v_bias gradient: column sums of the stored (bf16) dV, kept per lane until the end
        if constexpr (VB) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[db * 16 + 4 * g + e] += (float)wv[e] * kmask;
        }
      }
  }
  if (stats) {
    for (int o = 32; o > 0; o >>= 1) {
      vmax = fmaxf(vmax, __shfl_xor(vmax, o));
      dmax = fmaxf(dmax, __shfl_xor(dmax, o));
      nmax = fmaxf(nmax, __shfl_xor(nmax, o));
    }
    if (lane == 0) {
      atomicMax(reinterpret_cast<int*>(stats) + h * 4 + 0, __float_as_int(nmax));
      atomicMax(reinterpret_cast<int*>(stats) + h * 4 + 1, __float_as_int(dmax));
      atomicMax(reinterpret_cast<int*>(stats) + h * 4 + 2, __float_as_int(vmax));
    }
  }
  if (VB) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float v = bsum[i];
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (r == 0) atomicAdd(vsum + (i >> 4) * 32 + 8 * ((i >> 2) & 3) + 4 * hh + (i & 3), v);
      }
    }
    __syncthreads();
    if (threadIdx.x < HD) atomicAdd(dvbias + h * HD + threadIdx.x, vsum[threadIdx.x]);
  }
}

// ------------------------------------------------------------------------------- backward (dQ, dBias)
// a wave owns 32 queries and sweeps the keys with tiles [key rows][q cols].  The bias-table
// gradient is bucketed with fixed-point integer LDS atomics: per sample the scale is 2^24 / bound,
//   |dS| = p |dP - delta| <= max_q |dO_q| * max_key |V_key| + max_q |delta_q| =: bound,
// so a bucket (<= 196 terms) cannot overflow int32; the integer buckets are folded into fp32
// buckets once per sample and flushed to the table gradient once per workgroup.
template <int NKB, bool DT>
__global__ __launch_bounds__(512) void attn_bwd_q_kernel(const __bf16* __restrict__ qkv, long long ldq,
                                                         const __bf16* __restrict__ dout, long long ldo,
                                                         const float* __restrict__ lse,
                                                         const float* __restrict__ delta,
                                                         const float* __restrict__ stats,
                                                         const float* __restrict__ table, int nrd, int Wh, int Ww,
                                                         __bf16* __restrict__ dqkv, long long lddq,
                                                         float* __restrict__ dtable, float* __restrict__ dqbias,
                                                         int B, int T, int D, int H, float scale, int spb) {
  constexpr int TP = NKB * 32;
  constexpr int IMG = TP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RelGeom geo = rel_geom(Wh, Ww);
  float* tabX = reinterpret_cast<float*>(smem);
  int* binsi = reinterpret_cast<int*>(tabX + geo.len);      // [len] fixed-point buckets (this workgroup), extended index
  float* qsum = reinterpret_cast<float*>(binsi + geo.len);  // [64]
  int* codeQ = reinterpret_cast<int*>(qsum + HD);
  int* codeK = codeQ + TP;
  char* imgs = smem + (((2 * geo.len + HD + 2 * TP) * 4 + 15) & ~15);
  const int h = blockIdx.x % H, b0 = (blockIdx.x / H) * spb;
  const int b1 = b0 + spb < B ? b0 + spb : B;
  if (b0 >= b1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  rel_setup(tabX, codeQ, codeK, table, nrd, H, h, T, TP, Wh, Ww, kLog2e);
  for (int i = threadIdx.x; i < geo.len + HD; i += blockDim.x) binsi[i] = 0;   // binsi, qsum
  const int qb = wave;
  const bool active = qb < NKB;
  const int q = qb * 32 + r;
  const int qc = q < T ? q : T - 1;
  // fixed-point scale of the table-gradient buckets, from the head's bounds (see memhip_attn_delta):
  //   |dS| = p |dP - delta| <= max|dO_q| max|V_k| + max|delta_q| =: bound;  scale = 2^19 / bound, so that
  //   the <= 16 samples x 196 terms a bucket can collect stay below 2^31.
  float fx = 0.f;
  if (DT) {
    const float bound = sqrtf(stats[h * 4 + 0]) * sqrtf(stats[h * 4 + 2]) + stats[h * 4 + 1];
    fx = bound > 0.f ? 524288.0f / bound : 0.f;
  }
  bf16x8 Qn[4], dOn[4];
  float bsum[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) bsum[i] = 0.f;
  const float qmask = q < T ? 1.f : 0.f;
  float lqn = 0.f, dqn = 0.f;
  auto load_q = [&](int b) {
    const long long row = (long long)b * T + qc;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Qn[t] = ld16(qkv + row * ldq + h * HD + 16 * t + 8 * hh);
      dOn[t] = ld16(dout + row * ldo + h * HD + 16 * t + 8 * hh);
    }
    lqn = lse[((long long)b * H + h) * TP + qc] * kLog2e;
    dqn = delta[row * H + h];
  };
  auto stage_sample = [&](int b, int buf) {
    const __bf16* s = qkv + (long long)b * T * ldq + h * HD;
    stage_head(imgs + buf * 2 * IMG, s + D, ldq, T, TP);               // K
    stage_head(imgs + buf * 2 * IMG + IMG, s + 2 * D, ldq, T, TP);     // V
  };
  load_q(b0);
  stage_sample(b0, 0);
  for (int b = b0; b < b1; ++b) {
    const int cur = (b - b0) & 1;
    const char* Ks = imgs + cur * 2 * IMG;
    const char* Vs = Ks + IMG;
    bf16x8 Qf[4], dOf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { Qf[t] = Qn[t]; dOf[t] = dOn[t]; }
    const float lq = lqn, dq_ = dqn;
    ATTN_DMA_WAIT();
    __syncthreads();                      // K/V of sample b landed; sample b-1 fully consumed
    if (b + 1 < b1) { load_q(b + 1); stage_sample(b + 1, cur ^ 1); }
    if (active) {
      const int cq4 = codeQ[qc] + (int)lds_addr_of(reinterpret_cast<const char*>(tabX));   // absolute LDS address of the lane's table window
      const int bins_delta = (int)(lds_addr_of(reinterpret_cast<const char*>(binsi)) - lds_addr_of(reinterpret_cast<const char*>(tabX)));
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
          if (kb == NKB - 1 && kb * 32 + 8 * g >= T) {        // 8 padded keys: dS = 0, nothing for the buckets
#pragma unroll
            for (int e = 0; e < 4; ++e) dPt[4 * g + e] = 0.f;
            continue;
          }
          const int4 kc = *reinterpret_cast<const int4*>(codeK + key0);
          const int kcs[4] = {kc.x, kc.y, kc.z, kc.w};
          bfr2(St, 4 * g);
          bfr2(St, 4 * g + 2);
          bfr2(dPt, 4 * g);
          bfr2(dPt, 4 * g + 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            const int idx4 = cq4 - kcs[e];
            const float sv = fmaf(St[i], kLog2e, lds_f32_abs(idx4));
            float p = fexp2(sv - lq) * qmask;                   // qmask = 0 for padding queries
            if (kb == NKB - 1 && key0 + e >= T) p = 0.f;
            const float ds = p * (dPt[i] - dq_);
            dPt[i] = ds;
            // masked elements add 0 (their codes are valid): no divergent branch around the atomic
            if (DT) lds_add_i32_abs(idx4 + bins_delta, fx_round(ds, fx));
          }
        }
        bf16x8 ckf[2][2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int db = 0; db < 2; ++db) ckf[ss][db] = col_frag_o(Ks, lo, kb, ss, db);
        bf16x8 dsf[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) dsf[ss] = acc_frag(dPt, ss, 1.0f);
        LDS_TR_WAIT();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int db = 0; db < 2; ++db) dQt[db] = MFMA32(ckf[ss][db], dsf[ss], dQt[db]);
      }
      // lane col = q, regs -> d ;  d(q_lin) = d(q') * scale
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (__bf16)(bfr(dQt[db][4 * g + e]) * scale);
          if (q < T)
            *reinterpret_cast<bf16x4*>(dqkv + ((long long)b * T + q) * lddq + h * HD + db * 32 + 8 * g + 4 * hh) = w;
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[db * 16 + 4 * g + e] += (float)w[e] * qmask;   // q_bias gradient
        }
    }
  }
  __syncthreads();
  if (DT) {
    // grid buckets one to one; the two cls regions are summed on chip first (hundreds of atomics on
    // ONE address per workgroup would serialise in L2)
    const float inv = fx > 0.f ? 1.0f / fx : 0.f;
    for (int i = threadIdx.x; i <= 2 * geo.off; i += blockDim.x) {
      const int v = binsi[i];
      if (v != 0) atomicAdd(dtable + (long long)i * H + h, (float)v * inv);
    }
    if (wave < 2) {
      const int base = wave == 0 ? 2 * geo.off + 1 : 3 * geo.off + 2;
      float v = 0.f;
      for (int i = lane; i <= geo.off; i += 64) v += (float)binsi[base + i] * inv;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) atomicAdd(dtable + (long long)(wave == 0 ? nrd - 2 : nrd - 3) * H + h, v);
    } else if (threadIdx.x == 128) {
      atomicAdd(dtable + (long long)(nrd - 1) * H + h, (float)binsi[5 * geo.off + 3] * inv);
    }
  }
  if (dqbias) {
    if (active) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float v = bsum[i];
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (r == 0) atomicAdd(qsum + (i >> 4) * 32 + 8 * ((i >> 2) & 3) + 4 * hh + (i & 3), v);
      }
    }
    __syncthreads();
  }
  if (dqbias && threadIdx.x < HD) atomicAdd(dqbias + h * HD + threadIdx.x, qsum[threadIdx.x]);
}

// rowsum(dO * O) and |dO|^2 per (token, head)
__global__ __launch_bounds__(256) void attn_delta_kernel(const __bf16* __restrict__ dout, const __bf16* __restrict__ out,
                                                         long long ldo, long long rows, int H,
                                                         float* __restrict__ delta) {
  const long long i = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);     // (row, head) pair index
  if (i >= rows * H) return;
  const long long row = i / H;
  const int h = (int)(i - row * H);
  const int c = (threadIdx.x & 7) * 8;
  const bf16x8 a = ld16(dout + row * ldo + h * HD + c), o = ld16(out + row * ldo + h * HD + c);
  float s = 0.f, n2 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    s = fmaf((float)a[k], (float)o[k], s);
    n2 = fmaf((float)a[k], (float)a[k], n2);
  }
  s += __shfl_xor(s, 1);   n2 += __shfl_xor(n2, 1);
  s += __shfl_xor(s, 2);   n2 += __shfl_xor(n2, 2);
  s += __shfl_xor(s, 4);   n2 += __shfl_xor(n2, 4);
  if ((threadIdx.x & 7) == 0) {
    delta[i] = s;
    delta[rows * H + i] = n2;              // |dO_row,h|^2: bound of the table-gradient buckets (attn_bwd)
  }
}

__global__ void attn_stats_zero_kernel(float* stats, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) stats[i] = 0.f;
}

// Samples per workgroup.  One workgroup is resident per CU (LDS), so a grid of more than #CUs
// workgroups runs a second, nearly empty round: ceil(256*12/256) = 12 samples per workgroup gives
// 22 * 12 = 264 workgroups on 256 CUs, i.e. T(12) + T(4) -- 13 gives 240 workgroups and T(13)
// (measured: forward 156 -> 130 us, backward 603 -> 503 us per layer).  Dealing (head, sample) pairs
// perfectly evenly (12 per workgroup, runs crossing into the next head) was tried and is no faster:
// the crossing workgroups pay the per-head setup twice.  Smallest count for which the grid fits one
// round, capped at 16 (the table-gradient buckets are sized for that).
int pick_spb(int B, int heads, hipStream_t s) {
  int num_cu = memhip::usable_cus(s);
  if (num_cu <= 0) num_cu = 256;
  for (int spb = 1; spb <= 16; ++spb)
    if ((long long)((B + spb - 1) / spb) * heads <= num_cu) return spb;
  return 16;
}

}  // namespace

namespace memhip {   // attn_stream.hip: sequences longer than 256 tokens
int attn_fwd_stream(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table, int window_h,
                    int window_w, void* out, int64_t ldo, float* lse, hipStream_t s);
int attn_bwd_stream(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse, float* delta,
                    float* stats, const float* table, int window_h, int window_w, int B, int T, int D, int heads,
                    float scale, void* dqkv, int64_t lddqkv, float* dtable, float* dq_bias, float* dv_bias,
                    hipStream_t s);
// attn_win.hip: long windows 40 / 20 wide in the slot layout (round 5)
bool attn_win_fits(int T, int window_h, int window_w);
int attn_fwd_win(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table, int window_h, int window_w,
                 void* out, int64_t ldo, float* lse, hipStream_t s);
int attn_bwd_win(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse, float* delta, float* stats,
                 const float* table, int window_h, int window_w, int B, int T, int D, int heads, float scale, void* dqkv,
                 int64_t lddqkv, float* dtable, float* dq_bias, float* dv_bias, void* ws, int64_t ws_bytes, hipStream_t s);
int64_t attn_bwd_win_workspace(int B, int T, int heads, int window_h, int window_w);
// attn16.hip: the 14 x 14 window (197 tokens) -- key-slot layout, fused backward
bool attn16_fits(int T, int window_h, int window_w);
int attn16_fwd(const void* qkv, int64_t ldqkv, int B, int D, int heads, const float* table, void* out, int64_t ldo,
               float* lse, hipStream_t s);
int attn16_bwd(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const void* out, int64_t ldout, const float* lse,
               const float* delta, const float* table, int B, int D, int heads, float scale, void* dqkv, int64_t lddqkv,
               float* dtable, float* dq_bias, hipStream_t s);
}

#define ATTN_DISPATCH(NKB_EXPR, MACRO)                                    \
  switch (NKB_EXPR) {                                                     \
    case 1: MACRO(1); break; case 2: MACRO(2); break; case 3: MACRO(3); break; \
    case 4: MACRO(4); break; case 5: MACRO(5); break; case 6: MACRO(6); break; \
    case 7: MACRO(7); break; case 8: MACRO(8); break;                     \
    default: return fail(MEMHIP_EUNSUPPORTED, "attention: %d tokens: internal dispatch error", T); \
  }

extern "C" int memhip_attn_tokens_padded(int T) { return ((T + 31) / 32) * 32; }

extern "C" int memhip_attn_fwd(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table,
                               int window_h, int window_w, void* out, int64_t ldo, float* lse,
                               memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && heads > 0 && D == heads * HD, "attn_fwd: head_dim must be 64 (D=%d heads=%d)", D, heads);
  MEMHIP_REQUIRE(window_h > 0 && window_w > 0 && window_h * window_w + 1 == T, "attn_fwd: T must be window_h*window_w + 1");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && table && out && lse, "attn_fwd: null pointer");
  MEMHIP_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0, "attn_fwd: ld must be a multiple of 8");
  hipStream_t s = as_stream(stream);
  if (opt(OPT_ATTN16) && memhip::attn16_fits(T, window_h, window_w))
    return memhip::attn16_fwd(qkv, ldqkv, B, D, heads, table, out, ldo, lse, s);
  const int nkb = (T + 31) / 32;
  const int nrd = (2 * window_h - 1) * (2 * window_w - 1) + 3;
  if (nkb > 8) {
    if (opt(OPT_ATTN_WIN) && memhip::attn_win_fits(T, window_h, window_w)) {
      const int rc = memhip::attn_fwd_win(qkv, ldqkv, B, T, D, heads, table, window_h, window_w, out, ldo, lse, s);
      if (rc != MEMHIP_EUNSUPPORTED) return rc;
    }
    return memhip::attn_fwd_stream(qkv, ldqkv, B, T, D, heads, table, window_h, window_w, out, ldo, lse, s);
  }
  const int spb = pick_spb(B, heads, s);
#define FWD(N)                                                                                          \
  {                                                                                                     \
    const size_t sm = (size_t)4 * N * 32 * 128 + (size_t)(rel_geom(window_h, window_w).len + 2 * N * 32) * 4 + 32; \
    if (sm > (size_t)kMaxLds) return fail(MEMHIP_EUNSUPPORTED, "attn_fwd: LDS budget exceeded");         \
    static bool attr_done = false;                                                                      \
    if (!attr_done) {                                                                                   \
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<N>),                 \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));             \
      attr_done = true;                                                                                 \
    }                                                                                                   \
    hipLaunchKernelGGL(attn_fwd_kernel<N>, dim3(((B + spb - 1) / spb) * heads), dim3(512), sm, s,       \
                       (const __bf16*)qkv, (long long)ldqkv, B, T, D, heads, table, nrd, window_h,      \
                       window_w, (__bf16*)out, (long long)ldo, lse, spb);                               \
  }
  ATTN_DISPATCH(nkb, FWD)
#undef FWD
  return check_launch("attn_fwd");
}

extern "C" int memhip_attn_delta(const void* dout, const void* out, int64_t ldo, int64_t rows, int heads,
                                 float* delta, memhip_stream_t stream) {
  MEMHIP_REQUIRE(rows >= 0 && heads > 0 && ldo % 8 == 0, "attn_delta: bad arguments");
  if (rows == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(dout && out && delta, "attn_delta: null pointer");
  const long long pairs = rows * heads;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((pairs + 31) / 32)), dim3(256), 0, as_stream(stream),
                     (const __bf16*)dout, (const __bf16*)out, (long long)ldo, (long long)rows, heads, delta);
  return check_launch("attn_delta");
}

// Attention backward from the forward OUTPUT: rowsum(dout * out) is computed by the library -- inside the fused 14 x 14
// kernel when it applies (no separate pass), otherwise by memhip_attn_delta into `delta` in front of memhip_attn_bwd.
extern "C" int memhip_attn_bwd_out(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const void* out, int64_t ldout,
                                   const float* lse, float* delta, const float* table, int window_h, int window_w, int B,
                                   int T, int D, int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable,
                                   float* dq_bias, float* dv_bias, memhip_stream_t stream) {
  return memhip_attn_bwd_out_ws(qkv, ldqkv, dout, ldo, out, ldout, lse, delta, table, window_h, window_w, B, T, D, heads, scale, dqkv,
                                lddqkv, dtable, dq_bias, dv_bias, nullptr, 0, stream);
}

extern "C" int memhip_attn_bwd_out_ws(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const void* out, int64_t ldout,
                                      const float* lse, float* delta, const float* table, int window_h, int window_w, int B,
                                      int T, int D, int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable,
                                      float* dq_bias, float* dv_bias, void* ws, int64_t ws_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && heads > 0 && D == heads * HD, "attn_bwd: head_dim must be 64");
  MEMHIP_REQUIRE(window_h > 0 && window_w > 0 && window_h * window_w + 1 == T, "attn_bwd: T must be window_h*window_w + 1");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && dout && out && lse && delta && table && dqkv, "attn_bwd: null pointer");
  MEMHIP_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0 && ldout % 8 == 0 && lddqkv % 8 == 0, "attn_bwd: ld must be a multiple of 8");
  if (opt(OPT_ATTN16) && !dv_bias && memhip::attn16_fits(T, window_h, window_w))
    return memhip::attn16_bwd(qkv, ldqkv, dout, ldo, out, ldout, lse, delta, table, B, D, heads, scale, dqkv, lddqkv, dtable,
                              dq_bias, as_stream(stream));
  MEMHIP_REQUIRE(ldo == ldout, "attn_bwd: dout and out must share a leading dimension on this path");
  const int rc = memhip_attn_delta(dout, out, ldo, (int64_t)B * T, heads, delta, stream);
  if (rc != MEMHIP_OK) return rc;
  return memhip_attn_bwd_ws(qkv, ldqkv, dout, ldo, lse, delta, table, window_h, window_w, B, T, D, heads, scale, dqkv, lddqkv, dtable,
                            dq_bias, dv_bias, ws, ws_bytes, stream);
}

extern "C" int memhip_attn_bwd(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse,
                               float* delta, const float* table, int window_h, int window_w, int B,
                               int T, int D, int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable,
                               float* dq_bias, float* dv_bias, memhip_stream_t stream) {
  return memhip_attn_bwd_ws(qkv, ldqkv, dout, ldo, lse, delta, table, window_h, window_w, B, T, D, heads, scale, dqkv, lddqkv, dtable,
                            dq_bias, dv_bias, nullptr, 0, stream);
}

extern "C" int64_t memhip_attn_bwd_workspace(int B, int T, int heads, int window_h, int window_w) {
  if (B <= 0 || T <= 0 || heads <= 0) return 0;
  return memhip::attn_bwd_win_workspace(B, T, heads, window_h, window_w);
}

extern "C" int memhip_attn_bwd_ws(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse,
                                  float* delta, const float* table, int window_h, int window_w, int B,
                                  int T, int D, int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable,
                                  float* dq_bias, float* dv_bias, void* ws, int64_t ws_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && heads > 0 && D == heads * HD, "attn_bwd: head_dim must be 64");
  MEMHIP_REQUIRE(window_h > 0 && window_w > 0 && window_h * window_w + 1 == T, "attn_bwd: T must be window_h*window_w + 1");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(qkv && dout && lse && delta && table && dqkv, "attn_bwd: null pointer");
  MEMHIP_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0 && lddqkv % 8 == 0, "attn_bwd: ld must be a multiple of 8");
  hipStream_t s = as_stream(stream);
  // the fused kernel has no v_bias-gradient output (the engine derives it from the proj dgrad: vit_engine.py)
  if (opt(OPT_ATTN16) && !dv_bias && memhip::attn16_fits(T, window_h, window_w))
    return memhip::attn16_bwd(qkv, ldqkv, dout, ldo, nullptr, 0, lse, delta, table, B, D, heads, scale, dqkv, lddqkv, dtable,
                              dq_bias, s);
  const int nkb = (T + 31) / 32;
  const int nrd = (2 * window_h - 1) * (2 * window_w - 1) + 3;
  const int spb = pick_spb(B, heads, s);
  const int grid = ((B + spb - 1) / spb) * heads;
  const int glen = rel_geom(window_h, window_w).len;
  float* stats = delta + 2LL * B * T * heads;             // per-head bounds, written by the kv kernel for the q kernel
  if (dtable) hipLaunchKernelGGL(attn_stats_zero_kernel, dim3(1), dim3(64), 0, s, stats, 4 * heads);
  if (nkb > 8) {
    if (opt(OPT_ATTN_WIN) && memhip::attn_win_fits(T, window_h, window_w)) {
      const int rc = memhip::attn_bwd_win(qkv, ldqkv, dout, ldo, lse, delta, stats, table, window_h, window_w, B, T, D, heads,
                                          scale, dqkv, lddqkv, dtable, dq_bias, dv_bias, ws, ws_bytes, s);
      if (rc != MEMHIP_EUNSUPPORTED) return rc;
    }
    return memhip::attn_bwd_stream(qkv, ldqkv, dout, ldo, lse, delta, stats, table, window_h, window_w, B, T, D, heads,
                                   scale, dqkv, lddqkv, dtable, dq_bias, dv_bias, s);
  }
#define BWD(N)                                                                                          \
  {                                                                                                     \
    static bool attr_done = false;                                                                      \
    if (!attr_done) {                                                                                   \
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kv_kernel<N, true>),        \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));             \
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kv_kernel<N, false>),       \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));             \
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_q_kernel<N, true>),         \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));             \
      MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_q_kernel<N, false>),        \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));             \
      attr_done = true;                                                                                 \
    }                                                                                                   \
    const size_t sm_kv = (size_t)4 * N * 32 * 128 + (size_t)(glen + 6 * N * 32 + HD) * 4 + 32;          \
    const size_t sm_q = (size_t)4 * N * 32 * 128 + (size_t)(2 * glen + HD + 2 * N * 32) * 4 + 32;       \
    if (sm_kv > (size_t)kMaxLds || sm_q > (size_t)kMaxLds)                                              \
      return fail(MEMHIP_EUNSUPPORTED, "attn_bwd: %d tokens with a %dx%d window exceed the LDS budget", T, window_h, \
                  window_w);                                                                            \
    if (dv_bias)                                                                                        \
      hipLaunchKernelGGL((attn_bwd_kv_kernel<N, true>), dim3(grid), dim3(512), sm_kv, s, (const __bf16*)qkv, \
                         (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta,             \
                         dtable ? stats : (float*)nullptr, table, nrd, window_h, window_w, (__bf16*)dqkv, \
                         (long long)lddqkv, dv_bias, B, T, D, heads, spb);                              \
    else                                                                                                \
      hipLaunchKernelGGL((attn_bwd_kv_kernel<N, false>), dim3(grid), dim3(512), sm_kv, s, (const __bf16*)qkv, \
                         (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta,             \
                         dtable ? stats : (float*)nullptr, table, nrd, window_h, window_w, (__bf16*)dqkv, \
                         (long long)lddqkv, dv_bias, B, T, D, heads, spb);                              \
    if (dtable)                                                                                         \
      hipLaunchKernelGGL((attn_bwd_q_kernel<N, true>), dim3(grid), dim3(512), sm_q, s, (const __bf16*)qkv, \
                         (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta, stats,      \
                         table, nrd, window_h, window_w, (__bf16*)dqkv, (long long)lddqkv, dtable,      \
                         dq_bias, B, T, D, heads, scale, spb);                                          \
    else                                                                                                \
      hipLaunchKernelGGL((attn_bwd_q_kernel<N, false>), dim3(grid), dim3(512), sm_q, s, (const __bf16*)qkv, \
                         (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta, stats,      \
                         table, nrd, window_h, window_w, (__bf16*)dqkv, (long long)lddqkv, dtable,      \
                         dq_bias, B, T, D, heads, scale, spb);                                          \
  }
  ATTN_DISPATCH(nkb, BWD)
#undef BWD
  return check_launch("attn_bwd");
}
