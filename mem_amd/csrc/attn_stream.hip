// Fused attention with the shared relative-position bias for sequences LONGER than 256 tokens
// (BASELINE configs[4]: ViT-L/16 on 480 x 640 voxels = 30 x 40 + 1 = 1201 tokens), forward and backward.
// Same arithmetic, operand layouts, rounding points and outputs (lse, delta, table-gradient buckets) as
// attn.hip -- reference: Attention.forward, mem/modeling_finetune.py:137-154 + RelativePositionBias
// :213-247 -- but the sequence no longer fits one workgroup's LDS, so:
//   * a workgroup owns 8 x 32 "resident" tokens (one 32-token block per wave, fragments in registers) of
//     one (sample, head) and STREAMS the other operand pair through LDS in chunks of CKB x 32 tokens,
//     double-buffered by LDS-DMA (chunk c+1 is in flight while chunk c is computed);
//   * forward keeps a running row maximum / row sum per query (one lane = one query column of the
//     transposed score tile, so the rescale is in-lane) and normalises the fp32 accumulator at the end;
//   * backward recomputes P from the stored lse, exactly as the short-sequence kernels do.
// The head's extended bias table (5*off+4 floats, 46.6 KB for a 30 x 40 window) stays in LDS for the
// whole workgroup; bucket indices are code differences (attn_common.hpp), no [H,N,N] tensor exists.
#include "attn_common.hpp"
#include <type_traits>

namespace {

// tokens [t0, t0 + NT) of a head slice -> chunk image (rows indexed by the LOCAL token; zero beyond T)
__device__ __forceinline__ void stage_chunk(char* dst, const __bf16* src, long long ld, int t0, int T, int NT) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int inst = wave; inst < NT / 8; inst += nw) {
    const int lt = inst * 8 + (lane >> 3), cpos = lane & 7;
    const int chunk = cpos ^ img_key(lt);
    const int tok = t0 + lt;
    const void* g = tok < T ? (const void*)(src + (long long)tok * ld + chunk * 8)
                            : (const void*)(g_attn_zero_page + cpos * 16);
    glds16(g, dst + inst * 1024);
  }
}

// ------------------------------------------------------------------------------- forward
template <int CKB>
__global__ __launch_bounds__(512) void attn_fwd_stream_kernel(const __bf16* __restrict__ qkv, long long ldq, int T, int TP,
                                                              int TPc, int D, int H, const float* __restrict__ table,
                                                              int nrd, int Wh, int Ww, __bf16* __restrict__ out,
                                                              long long ldo, float* __restrict__ lse) {
  constexpr int CT = CKB * 32;
  constexpr int IMG = CT * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RelGeom geo = rel_geom(Wh, Ww);
  float* tabX = reinterpret_cast<float*>(smem);
  int* codeQ = reinterpret_cast<int*>(tabX + geo.len);
  int* codeK = codeQ + TPc;
  char* imgs = smem + (((geo.len + 2 * TPc) * 4 + 15) & ~15);
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  rel_setup(tabX, codeQ, codeK, table, nrd, H, h, T, TPc, Wh, Ww, kLog2e);
  const int qb = blockIdx.x * 8 + wave;
  const bool active = qb * 32 < T;
  const int q = qb * 32 + r;
  const int qc = q < T ? q : T - 1;
  const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
  bf16x8 Qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) Qf[t] = ld16(s0 + (long long)qc * ldq + 16 * t + 8 * hh);
  const int nchunks = (T + CT - 1) / CT;
  stage_chunk(imgs, s0 + D, ldq, 0, T, CT);
  stage_chunk(imgs + IMG, s0 + 2 * D, ldq, 0, T, CT);
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    const char* Ks = imgs + cur * 2 * IMG;
    const char* Vs = Ks + IMG;
    ATTN_DMA_WAIT();
    __syncthreads();                         // chunk c landed; chunk c-1 fully consumed
    if (c + 1 < nchunks) {
      stage_chunk(imgs + (cur ^ 1) * 2 * IMG, s0 + D, ldq, (c + 1) * CT, T, CT);
      stage_chunk(imgs + (cur ^ 1) * 2 * IMG + IMG, s0 + 2 * D, ldq, (c + 1) * CT, T, CT);
    }
    if (!active) continue;
    const int cq4 = codeQ[qc] + (int)lds_addr_of(reinterpret_cast<const char*>(tabX));   // absolute LDS address of the lane's table window
    f32x16 s[CKB];
#pragma unroll
    for (int kb = 0; kb < CKB; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) s[kb] = MFMA32(row_frag_o(Ks, lo, kb, t), Qf[t], s[kb]);
    }
    float cmax = -INFINITY;
    // keys >= T only exist in the last chunk (a wave-uniform fact): every other chunk runs without the per-element
    // compare + select (2 of ~13 vector instructions per score in a VALU-issue-bound loop)
    auto scores = [&](auto PAD) {
#pragma unroll
      for (int kb = 0; kb < CKB; ++kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key0 = c * CT + kb * 32 + 8 * g + 4 * hh;
          const int4 kc = *reinterpret_cast<const int4*>(codeK + key0);
          const int kcs[4] = {kc.x, kc.y, kc.z, kc.w};
          bfr2(s[kb], 4 * g);
          bfr2(s[kb], 4 * g + 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = fmaf(s[kb][4 * g + e], kLog2e, lds_f32_abs(cq4 - kcs[e]));
            if (decltype(PAD)::value && key0 + e >= T) v = -INFINITY;
            s[kb][4 * g + e] = v;
            cmax = fmaxf(cmax, v);
          }
        }
      }
    };
    if ((c + 1) * CT > T) scores(std::true_type{}); else scores(std::false_type{});
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
    const float mn = fmaxf(m, cmax);         // finite from the first chunk on (key 0 is never masked)
    const float alpha = fexp2(m - mn);
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < CKB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = fexp2(s[kb][i] - mn);
        s[kb][i] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    l = fmaf(l, alpha, sum);
    m = mn;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
#pragma unroll
    for (int kb = 0; kb < CKB; ++kb) {
      bf16x8 vf[2][2];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int db = 0; db < 2; ++db) vf[ss][db] = col_frag_o(Vs, lo, kb, ss, db);
      bf16x8 pf[2];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) pf[ss] = acc_frag(s[kb], ss, 1.0f);
      LDS_TR_WAIT();
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(vf[ss][db], pf[ss], o[db]);
    }
  }
  if (!active) return;
  const float inv = 1.0f / l;
  if (hh == 0 && q < T) lse[((long long)b * H + h) * TP + q] = (m + flog2(l)) * kLn2;
  if (q < T) {
    __bf16* orow = out + ((long long)b * T + q) * ldo + h * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (__bf16)(o[db][4 * g + e] * inv);
        *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * hh) = w;
      }
  }
}

// ------------------------------------------------------------------------------- backward (dK, dV)
// a wave owns 32 keys (K, V fragments in registers) and the workgroup streams Q' / dO chunks
template <int CKB, bool VB>
__global__ __launch_bounds__(512) void attn_bwd_kv_stream_kernel(
    const __bf16* __restrict__ qkv, long long ldq, const __bf16* __restrict__ dout, long long ldo,
    const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ stats,
    const float* __restrict__ table, int nrd, int Wh, int Ww, __bf16* __restrict__ dqkv, long long lddq,
    float* __restrict__ dvbias, int B, int T, int TP, int TPc, int D, int H) {
  constexpr int CT = CKB * 32;
  constexpr int IMG = CT * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RelGeom geo = rel_geom(Wh, Ww);
  float* tabX = reinterpret_cast<float*>(smem);
  int* codeQ = reinterpret_cast<int*>(tabX + geo.len);
  int* codeK = codeQ + TPc;
  float* lseS = reinterpret_cast<float*>(codeK + TPc);      // [2][CT]  (log2 domain)
  float* delS = lseS + 2 * CT;                              // [2][CT]
  float* vsum = delS + 2 * CT;                              // [64]
  char* imgs = smem + (((geo.len + 2 * TPc + 4 * CT + HD) * 4 + 15) & ~15);
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  rel_setup(tabX, codeQ, codeK, table, nrd, H, h, T, TPc, Wh, Ww, kLog2e);
  if (threadIdx.x < HD) vsum[threadIdx.x] = 0.f;
  const int kbg = blockIdx.x * 8 + wave;
  const bool active = kbg * 32 < T;
  const int key = kbg * 32 + r;
  const int kc_tok = key < T ? key : T - 1;
  const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
  const __bf16* d0 = dout + (long long)b * T * ldo + h * HD;
  bf16x8 Kf[4], Vf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    Kf[t] = ld16(s0 + (long long)kc_tok * ldq + D + 16 * t + 8 * hh);
    Vf[t] = ld16(s0 + (long long)kc_tok * ldq + 2 * D + 16 * t + 8 * hh);
  }
  float vmax = 0.f, dmax = 0.f, nmax = 0.f;
  if (stats) {
    float vn = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) vn = fmaf((float)Vf[t][i], (float)Vf[t][i], vn);
    vn += __shfl_xor(vn, 32);
    vmax = vn;
  }
  float lsen = 0.f, deln = 0.f;
  auto load_next = [&](int c) {
    const int qq = c * CT + (int)threadIdx.x;
    const bool ok = (int)threadIdx.x < CT && qq < T;
    lsen = ok ? lse[((long long)b * H + h) * TP + qq] * kLog2e : 0.f;
    deln = ok ? delta[((long long)b * T + qq) * H + h] : 0.f;
    if (stats && ok) nmax = fmaxf(nmax, delta[((long long)B * T + (long long)b * T + qq) * H + h]);   // |dO_q|^2
  };
  const int nchunks = (T + CT - 1) / CT;
  load_next(0);
  stage_chunk(imgs, s0, ldq, 0, T, CT);                     // Q'
  stage_chunk(imgs + IMG, d0, ldo, 0, T, CT);               // dO
  const float kmask = key < T ? 1.f : 0.f;
  f32x16 dVt[2], dKt[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dVt[db][i] = 0.f; dKt[db][i] = 0.f; }
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    const char* Qs = imgs + cur * 2 * IMG;
    const char* dOs = Qs + IMG;
    if (stats) dmax = fmaxf(dmax, fabsf(deln));
    if ((int)threadIdx.x < CT) { lseS[cur * CT + threadIdx.x] = lsen; delS[cur * CT + threadIdx.x] = deln; }
    ATTN_DMA_WAIT();
    __syncthreads();
    if (c + 1 < nchunks) {
      load_next(c + 1);
      stage_chunk(imgs + (cur ^ 1) * 2 * IMG, s0, ldq, (c + 1) * CT, T, CT);
      stage_chunk(imgs + (cur ^ 1) * 2 * IMG + IMG, d0, ldo, (c + 1) * CT, T, CT);
    }
    if (!active) continue;
    const bool qpad = (c + 1) * CT > T;                                     // this chunk holds queries >= T
    const bool kpad = __builtin_amdgcn_readfirstlane(kbg) * 32 + 32 > T;    // this wave holds keys >= T
    const int ck4 = codeK[kc_tok] - (int)lds_addr_of(reinterpret_cast<const char*>(tabX));   // (folds the table's LDS base)
    const float* lseC = lseS + cur * CT;
    const float* delC = delS + cur * CT;
#pragma unroll
    for (int qb = 0; qb < CKB; ++qb) {
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
        const int ql = qb * 32 + 8 * g + 4 * hh;            // inside the chunk
        const int q0 = c * CT + ql;
        const int4 qcv = *reinterpret_cast<const int4*>(codeQ + q0);
        const float4 lv = *reinterpret_cast<const float4*>(lseC + ql);
        const float4 dv = *reinterpret_cast<const float4*>(delC + ql);
        const int qcs[4] = {qcv.x, qcv.y, qcv.z, qcv.w};
        const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
        bfr2(S, 4 * g);
        bfr2(S, 4 * g + 2);
        bfr2(dP, 4 * g);
        bfr2(dP, 4 * g + 2);
        // (padding keys: only the wave that holds them multiplies; padding queries: only the last chunk selects -- both
        // conditions are wave-uniform, the common path has neither instruction)
        if (!kpad && !qpad) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            const float sv = fmaf(S[i], kLog2e, lds_f32_abs(qcs[e] - ck4));
            const float p = fexp2(sv - ll[e]);
            S[i] = p;
            dP[i] = p * (dP[i] - dd[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            const float sv = fmaf(S[i], kLog2e, lds_f32_abs(qcs[e] - ck4));
            float p = fexp2(sv - ll[e]) * kmask;               // kmask = 0 for padding keys
            if (q0 + e >= T) p = 0.f;
            S[i] = p;
            dP[i] = p * (dP[i] - dd[e]);
          }
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
  }
  float bsum[VB ? 32 : 1];
  if (active) {
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
        if constexpr (VB) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[db * 16 + 4 * g + e] = (float)wv[e] * kmask;
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
// a wave owns 32 queries (Q', dO fragments in registers) and the workgroup streams K / V chunks; the
// workgroup is persistent over `spb` samples so that the table-gradient buckets are flushed once.
// Fixed point: a bucket collects at most ONE term per resident query and sample (for a fixed query the
// bucket index is injective in the key), i.e. <= 256 * spb <= 4096 terms of magnitude <= 2^18.
template <int CKB, bool DT>
__global__ __launch_bounds__(512) void attn_bwd_q_stream_kernel(
    const __bf16* __restrict__ qkv, long long ldq, const __bf16* __restrict__ dout, long long ldo,
    const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ stats,
    const float* __restrict__ table, int nrd, int Wh, int Ww, __bf16* __restrict__ dqkv, long long lddq,
    float* __restrict__ dtable, float* __restrict__ dqbias, int B, int T, int TP, int TPc, int D, int H, float scale,
    int spb) {
  constexpr int CT = CKB * 32;
  constexpr int IMG = CT * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RelGeom geo = rel_geom(Wh, Ww);
  float* tabX = reinterpret_cast<float*>(smem);
  int* binsi = reinterpret_cast<int*>(tabX + geo.len);      // [len] fixed-point buckets, extended index
  float* qsum = reinterpret_cast<float*>(binsi + geo.len);  // [64]
  int* codeQ = reinterpret_cast<int*>(qsum + HD);
  int* codeK = codeQ + TPc;
  char* imgs = smem + (((2 * geo.len + HD + 2 * TPc) * 4 + 15) & ~15);
  const int h = blockIdx.y, b0 = blockIdx.z * spb;
  const int b1 = b0 + spb < B ? b0 + spb : B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  rel_setup(tabX, codeQ, codeK, table, nrd, H, h, T, TPc, Wh, Ww, kLog2e);
  for (int i = threadIdx.x; i < geo.len + HD; i += blockDim.x) binsi[i] = 0;   // binsi, qsum
  const int qb = blockIdx.x * 8 + wave;
  const bool active = qb * 32 < T;
  const int q = qb * 32 + r;
  const int qc = q < T ? q : T - 1;
  float fx = 0.f;
  if (DT) {
    const float bound = sqrtf(stats[h * 4 + 0]) * sqrtf(stats[h * 4 + 2]) + stats[h * 4 + 1];
    fx = bound > 0.f ? 262144.0f / bound : 0.f;
  }
  float bsum[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) bsum[i] = 0.f;
  const float qmask = q < T ? 1.f : 0.f;
  const int nchunks = (T + CT - 1) / CT;
  for (int b = b0; b < b1; ++b) {
    const long long row = (long long)b * T + qc;
    const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
    bf16x8 Qf[4], dOf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Qf[t] = ld16(qkv + row * ldq + h * HD + 16 * t + 8 * hh);
      dOf[t] = ld16(dout + row * ldo + h * HD + 16 * t + 8 * hh);
    }
    const float lq = lse[((long long)b * H + h) * TP + qc] * kLog2e;
    const float dq_ = delta[row * H + h];
    __syncthreads();                       // previous sample's last chunk fully consumed (and the setup done)
    stage_chunk(imgs, s0 + D, ldq, 0, T, CT);
    stage_chunk(imgs + IMG, s0 + 2 * D, ldq, 0, T, CT);
    f32x16 dQt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) dQt[db][i] = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      const int cur = c & 1;
      const char* Ks = imgs + cur * 2 * IMG;
      const char* Vs = Ks + IMG;
      ATTN_DMA_WAIT();
      __syncthreads();
      if (c + 1 < nchunks) {
        stage_chunk(imgs + (cur ^ 1) * 2 * IMG, s0 + D, ldq, (c + 1) * CT, T, CT);
        stage_chunk(imgs + (cur ^ 1) * 2 * IMG + IMG, s0 + 2 * D, ldq, (c + 1) * CT, T, CT);
      }
      if (!active) continue;
      const bool kpadc = (c + 1) * CT > T;                                   // this chunk holds keys >= T
      const bool qpadw = __builtin_amdgcn_readfirstlane(qb) * 32 + 32 > T;   // this wave holds queries >= T
      const int cq4 = codeQ[qc] + (int)lds_addr_of(reinterpret_cast<const char*>(tabX));   // absolute LDS address of the lane's table window
      const int bins_delta = (int)(lds_addr_of(reinterpret_cast<const char*>(binsi)) - lds_addr_of(reinterpret_cast<const char*>(tabX)));
#pragma unroll
      for (int kb = 0; kb < CKB; ++kb) {
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
          const int key0 = c * CT + kb * 32 + 8 * g + 4 * hh;
          const int4 kc = *reinterpret_cast<const int4*>(codeK + key0);
          const int kcs[4] = {kc.x, kc.y, kc.z, kc.w};
          bfr2(St, 4 * g);
          bfr2(St, 4 * g + 2);
          bfr2(dPt, 4 * g);
          bfr2(dPt, 4 * g + 2);
          if (!qpadw && !kpadc) {                                // (wave-uniform: no padding query in this wave, no padding key in this chunk)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = 4 * g + e;
              const int idx4 = cq4 - kcs[e];
              const float sv = fmaf(St[i], kLog2e, lds_f32_abs(idx4));
              const float ds = fexp2(sv - lq) * (dPt[i] - dq_);
              dPt[i] = ds;
              if (DT) lds_add_i32_abs(idx4 + bins_delta, fx_round(ds, fx));
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = 4 * g + e;
              const int idx4 = cq4 - kcs[e];
              const float sv = fmaf(St[i], kLog2e, lds_f32_abs(idx4));
              float p = fexp2(sv - lq) * qmask;                 // qmask = 0 for padding queries
              if (key0 + e >= T) p = 0.f;
              const float ds = p * (dPt[i] - dq_);
              dPt[i] = ds;
              if (DT) lds_add_i32_abs(idx4 + bins_delta, fx_round(ds, fx));
            }
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
    }
    if (active) {
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
    if (threadIdx.x < HD) atomicAdd(dqbias + h * HD + threadIdx.x, qsum[threadIdx.x]);
  }
}

constexpr int kFwdCKB = 4, kKvCKB = 4, kQCKB = 2;

template <typename K>
int set_lds_attr(K kernel, bool* done) {
  if (!*done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kMaxLds));
    *done = true;
  }
  return MEMHIP_OK;
}

}  // namespace

namespace memhip {

// called by memhip_attn_fwd / memhip_attn_bwd (attn.hip) when T > 256; arguments already validated
int attn_fwd_stream(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table, int window_h,
                    int window_w, void* out, int64_t ldo, float* lse, hipStream_t s) {
  constexpr int CT = kFwdCKB * 32;
  const int TP = ((T + 31) / 32) * 32, TPc = ((T + CT - 1) / CT) * CT;
  const int nrd = (2 * window_h - 1) * (2 * window_w - 1) + 3;
  const size_t sm = (size_t)4 * CT * 128 + (size_t)(rel_geom(window_h, window_w).len + 2 * TPc) * 4 + 32;
  if (sm > (size_t)kMaxLds)
    return fail(MEMHIP_EUNSUPPORTED, "attn_fwd: a %dx%d window needs %zu bytes of LDS", window_h, window_w, sm);
  static bool done = false;
  if (int rc = set_lds_attr(attn_fwd_stream_kernel<kFwdCKB>, &done)) return rc;
  hipLaunchKernelGGL(attn_fwd_stream_kernel<kFwdCKB>, dim3((TP / 32 + 7) / 8, heads, B), dim3(512), sm, s,
                     (const __bf16*)qkv, (long long)ldqkv, T, TP, TPc, D, heads, table, nrd, window_h, window_w,
                     (__bf16*)out, (long long)ldo, lse);
  return check_launch("attn_fwd(stream)");
}

int attn_bwd_stream(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse, float* delta,
                    float* stats, const float* table, int window_h, int window_w, int B, int T, int D, int heads,
                    float scale, void* dqkv, int64_t lddqkv, float* dtable, float* dq_bias, float* dv_bias,
                    hipStream_t s) {
  const int TP = ((T + 31) / 32) * 32;
  const int nrd = (2 * window_h - 1) * (2 * window_w - 1) + 3;
  const int glen = rel_geom(window_h, window_w).len;
  constexpr int CTK = kKvCKB * 32, CTQ = kQCKB * 32;
  const int TPcK = ((T + CTK - 1) / CTK) * CTK, TPcQ = ((T + CTQ - 1) / CTQ) * CTQ;
  const size_t sm_kv = (size_t)4 * CTK * 128 + (size_t)(glen + 2 * TPcK + 4 * CTK + HD) * 4 + 32;
  const size_t sm_q = (size_t)4 * CTQ * 128 + (size_t)(2 * glen + HD + 2 * TPcQ) * 4 + 32;
  if (sm_kv > (size_t)kMaxLds || sm_q > (size_t)kMaxLds)
    return fail(MEMHIP_EUNSUPPORTED, "attn_bwd: %d tokens with a %dx%d window exceed the LDS budget", T, window_h, window_w);
  static bool d0 = false, d1 = false, d2 = false, d3 = false;
  if (int rc = set_lds_attr(attn_bwd_kv_stream_kernel<kKvCKB, true>, &d0)) return rc;
  if (int rc = set_lds_attr(attn_bwd_kv_stream_kernel<kKvCKB, false>, &d1)) return rc;
  if (int rc = set_lds_attr(attn_bwd_q_stream_kernel<kQCKB, true>, &d2)) return rc;
  if (int rc = set_lds_attr(attn_bwd_q_stream_kernel<kQCKB, false>, &d3)) return rc;
  const int groups = (TP / 32 + 7) / 8;
  const dim3 gkv(groups, heads, B);
  if (dv_bias)
    hipLaunchKernelGGL((attn_bwd_kv_stream_kernel<kKvCKB, true>), gkv, dim3(512), sm_kv, s, (const __bf16*)qkv,
                       (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta, dtable ? stats : (float*)nullptr,
                       table, nrd, window_h, window_w, (__bf16*)dqkv, (long long)lddqkv, dv_bias, B, T, TP, TPcK, D, heads);
  else
    hipLaunchKernelGGL((attn_bwd_kv_stream_kernel<kKvCKB, false>), gkv, dim3(512), sm_kv, s, (const __bf16*)qkv,
                       (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta, dtable ? stats : (float*)nullptr,
                       table, nrd, window_h, window_w, (__bf16*)dqkv, (long long)lddqkv, dv_bias, B, T, TP, TPcK, D, heads);
  // samples per workgroup of the dQ kernel: amortise the bucket flush once the grid is a few rounds deep
  long long spb = (long long)B * heads * groups / 1024;
  spb = spb < 1 ? 1 : (spb > 16 ? 16 : spb);
  const dim3 gq(groups, heads, (unsigned)((B + spb - 1) / spb));
  if (dtable)
    hipLaunchKernelGGL((attn_bwd_q_stream_kernel<kQCKB, true>), gq, dim3(512), sm_q, s, (const __bf16*)qkv,
                       (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta, stats, table, nrd, window_h,
                       window_w, (__bf16*)dqkv, (long long)lddqkv, dtable, dq_bias, B, T, TP, TPcQ, D, heads, scale, (int)spb);
  else
    hipLaunchKernelGGL((attn_bwd_q_stream_kernel<kQCKB, false>), gq, dim3(512), sm_q, s, (const __bf16*)qkv,
                       (long long)ldqkv, (const __bf16*)dout, (long long)ldo, lse, delta, stats, table, nrd, window_h,
                       window_w, (__bf16*)dqkv, (long long)lddqkv, dtable, dq_bias, B, T, TP, TPcQ, D, heads, scale, (int)spb);
  return check_launch("attn_bwd(stream)");
}

}  // namespace memhip
