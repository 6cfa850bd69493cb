// Window attention with WIDE waves (round 6): four waves per workgroup, one per SIMD, the whole 512-register file each, 64
// resident tokens per wave (two 32-token blocks).  Same contract, arithmetic, rounding points and outputs as the kernels of
// attn_win.hip (8 waves x 32 resident tokens, 256 registers) -- reference: Attention.forward, mem/modeling_finetune.py:137-154
// + RelativePositionBias :213-247.
//
// Why.  The instruction streams of the 8-wave kernels say where their time goes (tools/isa_loops.py on hipcc -S): the dK / dV
// kernel issues 604 vector instructions and 64 MFMAs per 128-slot chunk and wave = ~3 200 cycles of vector issue + 2 050 of
// matrix pipe per wave, two waves per SIMD, and a chunk takes 12 500 cycles: the two waves run the same program in step, so
// vector and matrix work of a SIMD add up instead of overlapping, and a 256-register wave has no room to keep the S / dP
// products of the next tile in flight under the exponentials of this one.  Here
//   * the streamed Q' / dO row fragments, column fragments and per-slot scalars of a tile are read ONCE for two key blocks
//     (LDS traffic per tile 28 -> 16 KB);
//   * the long-lived dK^T / dV^T accumulators (128 registers) live in the ACCUMULATOR half of the register file: their MFMAs are
//     inline asm with "+a" operands (this file is compiled with -mllvm -amdgpu-mfma-vgpr-form=1: every builtin MFMA -- the
//     short-lived S / dP tiles the vector unit works on -- takes the VGPR form, no v_accvgpr copies);
//   * the S / dP products of tile q + 1 are issued in front of the vector work of tile q (two tile buffers);
//   * padded keys read their bias from a strip of -inf, so their probabilities are exactly 0 without a compare or a multiply
//     per element (the 8-wave kernel spent 128 of its 604 vector instructions per chunk on that mask).
#include "attn_win_common.hpp"
#ifndef WIN2_ORDER
#define WIN2_ORDER 0
#endif

namespace {

// d (16 accumulator registers) += a x b.  Accumulate chains of the same d need no wait states between them; the operands are
// VGPRs written by the vector unit or by an LDS read that has been waited for: `fresh` = 1 puts the two wait states a just-written
// VGPR needs in front of the MFMA that reads it (hipcc pads nothing inside an asm string).
template <int FRESH>
__device__ __forceinline__ void mfma_acc(f32x16& d, const bf16x8& a, const bf16x8& b) {
  if (FRESH)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
// the accumulators are read by compiler code (v_accvgpr_read) at the end of a sample: an 8-pass MFMA's result needs 12 wait
// states before any reader other than the next MFMA of its chain
template <typename... A>
__device__ __forceinline__ void mfma_acc_settle(A&... d) {
  asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");
  ((void)d, ...);
}

// ------------------------------------------------------------------------------- backward (dK, dV)
template <int WW, bool VB>
__global__ __launch_bounds__(256) void attn_bwd_kv_win2_kernel(
    const __bf16* __restrict__ qkv, long long ldq, const __bf16* __restrict__ dout, long long ldo,
    const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ stats,
    const float* __restrict__ table, int nrd, int Wh, __bf16* __restrict__ dqkv, long long lddq,
    float* __restrict__ dvbias, int B, int T, int TP, int D, int H, int groups, int nbz) {
  using G = WinGeo<WW>;
  constexpr int CT = G::CT, IMG = CT * 128, CKB = CT / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NB = (2 * Wh - 1) * G::P;
  float* R = reinterpret_cast<float*>(smem);
  float* Cq = R + ((NB + 3) & ~3);
  float* Cn = Cq + G::CQ;                                    // strip of -inf: the "bias" of a padded key
  float* nlS = Cn + G::CQ;                                   // [2][CT]  -lse * log2(e) by slot
  float* ndS = nlS + 2 * CT;                                 // [2][CT]  -delta by slot
  float* vsum = ndS + 2 * CT;                                // [4 waves][64]: v_bias gradient, a private row per wave
  char* imgs = reinterpret_cast<char*>(vsum + 4 * HD);
  const WinWg wg_ = win_wg(groups, H, nbz);
  if (!wg_.live) return;
  const int h = wg_.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  win_setup<WW>(R, Cq, table, nrd, H, h, Wh, 1.0f, false, nrd - 2);
  for (int i = threadIdx.x; i < G::CQ; i += blockDim.x) Cn[i] = -INFINITY;
  vsum[threadIdx.x] = 0.f;                                   // (256 threads = 4 x 64)
  const unsigned sel_lo = sel_lo_reg();
  const int kbg0 = wg_.group * 8 + wave * 2;                 // the wave's two key blocks: kbg0, kbg0 + 1
  const bool active = kbg0 * 32 < T;
  int key[2], kc_tok[2];
  unsigned base0[2], cstep[2];
  float bcls[2], kmask[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    key[kk] = (kbg0 + kk) * 32 + r;
    kc_tok[kk] = key[kk] < T ? key[kk] : T - 1;
    if (key[kk] >= T) {                                      // padded key: every bias it reads is -inf -> p = exp2(-inf) = 0
      base0[kk] = lds_addr_of(reinterpret_cast<const char*>(Cn));
      cstep[kk] = 0;
    } else if (key[kk] == 0) {
      base0[kk] = lds_addr_of(reinterpret_cast<const char*>(Cq));
      cstep[kk] = 0;
    } else {
      const int u = key[kk] - 1, ky = u / WW, kx = u - ky * WW;
      base0[kk] = lds_addr_of(reinterpret_cast<const char*>(R)) + 4u * (unsigned)((Wh - 1 - ky) * G::P + (WW - 1 - kx));
      cstep[kk] = 4u * G::RPC * G::P;
    }
    base0[kk] += 16u * hh;
    bcls[kk] = key[kk] >= T ? -INFINITY : table[(long long)(key[kk] == 0 ? nrd - 1 : nrd - 3) * H + h];   // bias from the cls query
    kmask[kk] = key[kk] < T ? 1.f : 0.f;
  }
  const int nch = (Wh + G::RPC - 1) / G::RPC;
  float vmax = 0.f, dmax = 0.f, nmax = 0.f;
  for (int b = wg_.bz; b < B; b += nbz) {
    const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
    const __bf16* d0 = dout + (long long)b * T * ldo + h * HD;
    bf16x8 Kf[2][4], Vf[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Kf[kk][t] = ld16(s0 + (long long)kc_tok[kk] * ldq + D + 16 * t + 8 * hh);
        Vf[kk][t] = ld16(s0 + (long long)kc_tok[kk] * ldq + 2 * D + 16 * t + 8 * hh);
      }
    if (stats) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float vn = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 8; ++i) vn = fmaf((float)Vf[kk][t][i], (float)Vf[kk][t][i], vn);
        vn += __shfl_xor(vn, 32);
        vmax = fmaxf(vmax, vn);
      }
    }
    float nln = 0.f, ndn = 0.f;
    auto load_next = [&](int c) {                            // this thread's slot of chunk c
      const int t = (int)threadIdx.x;
      if (t >= CT) return;
      const int j = t / G::WS, qx = t - j * G::WS, qy = c * G::RPC + j;
      bool ok = t < G::PAD0 && qx < WW && qy < Wh;
      int tok = 1 + qy * WW + qx;
      if (c == 0 && t == G::PAD0) { ok = true; tok = 0; }
      nln = ok ? -lse[((long long)b * H + h) * TP + tok] * kLog2e : -INFINITY;
      ndn = ok ? -delta[((long long)b * T + tok) * H + h] : 0.f;
      if (stats && ok) nmax = fmaxf(nmax, delta[((long long)B * T + (long long)b * T + tok) * H + h]);   // |dO_q|^2
    };
    load_next(0);
    __syncthreads();                                         // the previous sample's last chunk is consumed (and the setup done)
    stage_chunk_win<WW>(imgs, s0, ldq, 0, Wh);               // Q'
    stage_chunk_win<WW>(imgs + IMG, d0, ldo, 0, Wh);         // dO
    f32x16 dVt[2][2], dKt[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dVt[kk][db][i] = 0.f; dKt[kk][db][i] = 0.f; }
    for (int c = 0; c < nch; ++c) {
      const int cur = c & 1;
      const char* Qs = imgs + cur * 2 * IMG;
      const char* dOs = Qs + IMG;
      if (stats) dmax = fmaxf(dmax, fabsf(ndn));
      if ((int)threadIdx.x < CT) { nlS[cur * CT + threadIdx.x] = nln; ndS[cur * CT + threadIdx.x] = ndn; }
      ATTN_DMA_WAIT();
      __syncthreads();
      if (c + 1 < nch) {
        load_next(c + 1);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG, s0, ldq, c + 1, Wh);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG + IMG, d0, ldo, c + 1, Wh);
      }
      if (!active) continue;
      const unsigned base[2] = {base0[0] + (unsigned)c * cstep[0], base0[1] + (unsigned)c * cstep[1]};
      const float* nlC = nlS + cur * CT;
      const float* ndC = ndS + cur * CT;
      const ColAddr qa = col_addr(Qs, lo), da = col_addr(dOs, lo);
      // S^T = Q' K^T and dP^T = dO V^T of query block qb for both key blocks (lane = key, registers = query slots)
      auto sdp = [&](auto QB, f32x16 (&S)[2], f32x16 (&dP)[2]) {
        constexpr int qb = decltype(QB)::value;
        bf16x8 qf[4], df[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { qf[t] = row_frag_o(Qs, lo, qb, t); df[t] = row_frag_o(dOs, lo, qb, t); }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < 16; ++i) { S[kk][i] = 0.f; dP[kk][i] = 0.f; }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            S[kk] = MFMA32(qf[t], Kf[kk][t], S[kk]);
            dP[kk] = MFMA32(df[t], Vf[kk][t], dP[kk]);
          }
        }
      };
      // S -> P = exp2((bf16(S) + bias) log2e - lse log2e), dP -> dS = P (bf16(dP) - delta), in place
      auto soft = [&](auto QB, f32x16 (&S)[2], f32x16 (&dP)[2]) {
        constexpr int qb = decltype(QB)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int s0i = qb * 32 + 8 * g;                   // slot of (hh = 0, e = 0); hh = 1: + 4
          const int ql = s0i + 4 * hh;
          const float4 lv = *reinterpret_cast<const float4*>(nlC + ql);
          const float4 dv = *reinterpret_cast<const float4*>(ndC + ql);
          const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            float bz[4];
            if (G::valid(s0i) || G::valid(s0i + 4)) {
              const auto* p = reinterpret_cast<const __attribute__((address_space(3))) F2u*>(base[kk] + 4u * (unsigned)G::imm(s0i));
              bz[0] = p[0].a; bz[1] = p[0].b; bz[2] = p[1].a; bz[3] = p[1].b;
            } else {
              bz[0] = bz[1] = bz[2] = bz[3] = 0.f;
            }
            if (qb == G::CLS_KB && g == G::CLS_G && c == 0 && hh == 0) bz[0] = bcls[kk];      // the cls query's slot
            const unsigned s01 = pk_bf16(S[kk][4 * g], S[kk][4 * g + 1]), s23 = pk_bf16(S[kk][4 * g + 2], S[kk][4 * g + 3]);
            const unsigned d01 = pk_bf16(dP[kk][4 * g], dP[kk][4 * g + 1]), d23 = pk_bf16(dP[kk][4 * g + 2], dP[kk][4 * g + 3]);
            float sv[4], dq[4];
            sv[0] = add_lo(s01, bz[0], sel_lo); sv[1] = add_hi(s01, bz[1]); sv[2] = add_lo(s23, bz[2], sel_lo); sv[3] = add_hi(s23, bz[3]);
            dq[0] = add_lo(d01, dd[0], sel_lo); dq[1] = add_hi(d01, dd[1]); dq[2] = add_lo(d23, dd[2], sel_lo); dq[3] = add_hi(d23, dd[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float p = fexp2(fmaf(sv[e], kLog2e, ll[e]));     // ll = -lse log2(e): -inf for padding slots, sv = -inf for padded keys: p = 0
              S[kk][4 * g + e] = p;
              dP[kk][4 * g + e] = p * dq[e];
            }
          }
        }
      };
      // dV^T += dO^T P, dK^T += Q'^T dS (accumulators in the accumulator file)
      auto accum = [&](auto QB, f32x16 (&S)[2], f32x16 (&dP)[2]) {
        constexpr int qb = decltype(QB)::value;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          bf16x8 cdo[2], cq[2];
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            cdo[db] = col_frag_i<qb * 4096>(da, ss, db);
            cq[db] = col_frag_i<qb * 4096>(qa, ss, db);
          }
          bf16x8 pf[2], dsf[2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) { pf[kk] = acc_frag(S[kk], ss, 1.0f); dsf[kk] = acc_frag(dP[kk], ss, 1.0f); }
          LDS_TR_WAIT();
          mfma_acc<1>(dVt[0][0], cdo[0], pf[0]);
          mfma_acc<0>(dKt[0][0], cq[0], dsf[0]);
          mfma_acc<0>(dVt[0][1], cdo[1], pf[0]);
          mfma_acc<0>(dKt[0][1], cq[1], dsf[0]);
          mfma_acc<0>(dVt[1][0], cdo[0], pf[1]);
          mfma_acc<0>(dKt[1][0], cq[0], dsf[1]);
          mfma_acc<0>(dVt[1][1], cdo[1], pf[1]);
          mfma_acc<0>(dKt[1][1], cq[1], dsf[1]);
        }
      };
      static_assert(CKB == 4, "four 32-slot blocks per chunk");
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      f32x16 Sa[2], dPa[2], Sb[2], dPb[2];
#if WIN2_ORDER == 0
      sdp(I0{}, Sa, dPa);
      sdp(I1{}, Sb, dPb);
      soft(I0{}, Sa, dPa);
      accum(I0{}, Sa, dPa);
      __builtin_amdgcn_sched_barrier(0);
      sdp(I2{}, Sa, dPa);
      soft(I1{}, Sb, dPb);
      accum(I1{}, Sb, dPb);
      __builtin_amdgcn_sched_barrier(0);
      sdp(I3{}, Sb, dPb);
      soft(I2{}, Sa, dPa);
      accum(I2{}, Sa, dPa);
      __builtin_amdgcn_sched_barrier(0);
      soft(I3{}, Sb, dPb);
      accum(I3{}, Sb, dPb);
#elif WIN2_ORDER == 1       // no scheduling barriers between the blocks
      sdp(I0{}, Sa, dPa);
      sdp(I1{}, Sb, dPb);
      soft(I0{}, Sa, dPa);
      accum(I0{}, Sa, dPa);
      sdp(I2{}, Sa, dPa);
      soft(I1{}, Sb, dPb);
      accum(I1{}, Sb, dPb);
      sdp(I3{}, Sb, dPb);
      soft(I2{}, Sa, dPa);
      accum(I2{}, Sa, dPa);
      soft(I3{}, Sb, dPb);
      accum(I3{}, Sb, dPb);
#else                       // one tile buffer, no software pipeline
      sdp(I0{}, Sa, dPa); soft(I0{}, Sa, dPa); accum(I0{}, Sa, dPa);
      __builtin_amdgcn_sched_barrier(0);
      sdp(I1{}, Sa, dPa); soft(I1{}, Sa, dPa); accum(I1{}, Sa, dPa);
      __builtin_amdgcn_sched_barrier(0);
      sdp(I2{}, Sa, dPa); soft(I2{}, Sa, dPa); accum(I2{}, Sa, dPa);
      __builtin_amdgcn_sched_barrier(0);
      sdp(I3{}, Sa, dPa); soft(I3{}, Sa, dPa); accum(I3{}, Sa, dPa);
#endif
    }
    if (active) {
      mfma_acc_settle(dVt[0][0], dVt[0][1], dVt[1][0], dVt[1][1], dKt[0][0], dKt[0][1], dKt[1][0], dKt[1][1]);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 wv, wk;
#pragma unroll
            for (int e = 0; e < 4; ++e) { wv[e] = (__bf16)dVt[kk][db][4 * g + e]; wk[e] = (__bf16)dKt[kk][db][4 * g + e]; }
            if (key[kk] < T) {
              __bf16* drow = dqkv + ((long long)b * T + key[kk]) * lddq + h * HD;
              *reinterpret_cast<bf16x4*>(drow + 2 * D + db * 32 + 8 * g + 4 * hh) = wv;
              *reinterpret_cast<bf16x4*>(drow + D + db * 32 + 8 * g + 4 * hh) = wk;
            }
            if constexpr (VB)
              colsum_add4(vsum + wave * HD, r, hh, db, g, (float)wv[0] * kmask[kk], (float)wv[1] * kmask[kk],
                          (float)wv[2] * kmask[kk], (float)wv[3] * kmask[kk]);
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
    if (threadIdx.x < HD) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += vsum[w * HD + threadIdx.x];
      atomicAdd(dvbias + h * HD + threadIdx.x, v);
    }
  }
}

template <int WW>
int launch_kv2(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse, const float* delta, float* stats,
               const float* table, int Wh, int B, int T, int D, int heads, void* dqkv, int64_t lddqkv, float* dv_bias, int groups,
               int nbz, hipStream_t s) {
  using G = WinGeo<WW>;
  const int TP = ((T + 31) / 32) * 32;
  const int nrd = (2 * Wh - 1) * (2 * WW - 1) + 3;
  const int NBP = ((2 * Wh - 1) * G::P + 3) & ~3;
  const size_t sm = (size_t)(NBP + 2 * G::CQ + 4 * G::CT + 4 * HD) * 4 + (size_t)4 * G::CT * 128;
  if (sm > (size_t)kMaxLds) return MEMHIP_EUNSUPPORTED;
  static bool d0 = false, d1 = false;
  if (int rc = set_lds_attr(attn_bwd_kv_win2_kernel<WW, true>, &d0)) return rc;
  if (int rc = set_lds_attr(attn_bwd_kv_win2_kernel<WW, false>, &d1)) return rc;
  const dim3 grid(8 * ((heads * nbz + 7) / 8) * groups);
  if (dv_bias)
    hipLaunchKernelGGL((attn_bwd_kv_win2_kernel<WW, true>), grid, dim3(256), sm, s, (const __bf16*)qkv, (long long)ldqkv,
                       (const __bf16*)dout, (long long)ldo, lse, delta, stats, table, nrd, Wh, (__bf16*)dqkv, (long long)lddqkv,
                       dv_bias, B, T, TP, D, heads, groups, nbz);
  else
    hipLaunchKernelGGL((attn_bwd_kv_win2_kernel<WW, false>), grid, dim3(256), sm, s, (const __bf16*)qkv, (long long)ldqkv,
                       (const __bf16*)dout, (long long)ldo, lse, delta, stats, table, nrd, Wh, (__bf16*)dqkv, (long long)lddqkv,
                       dv_bias, B, T, TP, D, heads, groups, nbz);
  return check_launch("attn_bwd_kv(win2)");
}

}  // namespace

namespace memhip {

int attn_bwd_kv_win2(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse, const float* delta,
                     float* stats, const float* table, int window_h, int window_w, int B, int T, int D, int heads, void* dqkv,
                     int64_t lddqkv, float* dv_bias, int groups, int nbz, hipStream_t s) {
  if (window_w == 40)
    return launch_kv2<40>(qkv, ldqkv, dout, ldo, lse, delta, stats, table, window_h, B, T, D, heads, dqkv, lddqkv, dv_bias, groups, nbz, s);
  if (window_w == 20)
    return launch_kv2<20>(qkv, ldqkv, dout, ldo, lse, delta, stats, table, window_h, B, T, D, heads, dqkv, lddqkv, dv_bias, groups, nbz, s);
  return MEMHIP_EUNSUPPORTED;
}

}  // namespace memhip
