// Streaming attention for LONG windows with the key-slot layout of attn16.hip carried to wide grids (round 5;
// BASELINE configs[4]: ViT-L/16 on 480 x 640 voxels = a 30 x 40 window + cls = 1201 tokens).  Same contract, arithmetic,
// rounding points and outputs (lse, delta, table-gradient buckets) as attn_stream.hip / attn.hip -- reference:
// Attention.forward, mem/modeling_finetune.py:137-154 + RelativePositionBias :213-247.
//
// What attn_stream.hip pays per score element is two DEPENDENT LDS round trips (the key's code word, then the bias gathered
// at code(q) - code(k)) in front of one fma: 59 s_waitcnt per 128-key chunk and wave, which two waves per SIMD cannot hide
// (forward 0.147 of the MFMA peak, profiles/r04_vitl64_kernel_stats.csv).  Here the STREAMED operand is laid out by grid
// geometry instead of by token id:
//
//   * a chunk of CT = 128 slots holds RPC whole grid rows, each padded to WS = roundup(Ww, 8) slots (30 x 40: three rows of
//     40 = 120 slots + 8 padding slots; the cls token lives in the first padding slot of chunk 0);
//   * in the transposed score tile (lane = resident token, registers = streamed slots) the bucket of register (kb, g, e) is
//       lane part + chunk part + ((slot / WS) * (2 Ww - 1) + slot % WS)     -- the last term a COMPILE-TIME constant,
//     so with the head's table stored in the matching direction every bias is `ds_read2_b32 base offset:imm`: no index
//     arithmetic, no code tables, and -- the point -- no load that depends on another load: all bias reads of a 32 x 32
//     block are issued together, behind the K fragments of the next block;
//   * padding slots are known at compile time (static -inf / zero assignments, no compare per element); a ragged last chunk
//     (Wh % RPC != 0) takes a wave-uniform branch;
//   * the cls ROW reads a constant strip (stride 0), the cls COLUMN is one static register of chunk 0.
// Table in LDS: (2Wh-1)(2Ww-1) floats (18.6 KB for 30 x 40) instead of the 46.6 KB extended table.
// Instantiated for the window widths the engine meets (40: config #5; 20: the 16 x 20 parity fixture); every other
// long window stays on attn_stream.hip.
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

struct __attribute__((packed, aligned(4))) F2u { float a, b; };
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
  const int NB = (2 * Wh - 1) * G::P;
  for (int i = threadIdx.x; i < NB; i += blockDim.x) R[i] = table[(long long)(reversed ? NB - 1 - i : i) * H + h] * mul;
  const float cv = table[(long long)cls_bucket * H + h] * mul;
  for (int i = threadIdx.x; i < G::CQ; i += blockDim.x) Cq[i] = cv;
}

// ------------------------------------------------------------------------------- forward
// a wave owns 32 resident queries (token order), the workgroup streams K / V slot chunks; persistent over `nbz`-strided samples
template <int WW>
__global__ __launch_bounds__(512) void attn_fwd_win_kernel(const __bf16* __restrict__ qkv, long long ldq, int B, int T, int TP,
                                                           int D, int H, const float* __restrict__ table, int nrd, int Wh,
                                                           __bf16* __restrict__ out, long long ldo, float* __restrict__ lse) {
  using G = WinGeo<WW>;
  constexpr int CT = G::CT, IMG = CT * 128, CKB = CT / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NB = (2 * Wh - 1) * G::P;
  float* R = reinterpret_cast<float*>(smem);
  float* Cq = R + ((NB + 3) & ~3);
  char* imgs = reinterpret_cast<char*>(Cq + G::CQ);
  const int h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  win_setup<WW>(R, Cq, table, nrd, H, h, Wh, kLog2e, true, nrd - 3);
  const int qb = blockIdx.x * 8 + wave;
  const bool active = qb * 32 < T;
  const int q = qb * 32 + r;
  const int qc = q < T ? q : T - 1;
  // lane part of the bucket address (bytes, absolute LDS address) and its per-chunk step
  unsigned base0, cstep;
  if (q == 0 || q >= T) {
    base0 = lds_addr_of(reinterpret_cast<const char*>(Cq));
    cstep = 0;
  } else {
    const int u = q - 1, qy = u / WW, qx = u - qy * WW;
    base0 = lds_addr_of(reinterpret_cast<const char*>(R)) + 4u * (unsigned)(NB - 1 - (qy + Wh - 1) * G::P - (qx + WW - 1));
    cstep = 4u * G::RPC * G::P;
  }
  base0 += 16u * hh;
  const float bcls = table[(long long)(q == 0 ? nrd - 1 : nrd - 2) * H + h] * kLog2e;      // bias towards the cls key
  const int nch = (Wh + G::RPC - 1) / G::RPC;
  for (int b = blockIdx.z; b < B; b += gridDim.z) {
    const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
    bf16x8 Qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Qf[t] = ld16(s0 + (long long)qc * ldq + 16 * t + 8 * hh);
    __syncthreads();                          // the previous sample's last chunk is consumed (and the setup done)
    stage_chunk_win<WW>(imgs, s0 + D, ldq, 0, Wh);
    stage_chunk_win<WW>(imgs + IMG, s0 + 2 * D, ldq, 0, Wh);
    float m = -INFINITY, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    for (int c = 0; c < nch; ++c) {
      const int cur = c & 1;
      const char* Ks = imgs + cur * 2 * IMG;
      const char* Vs = Ks + IMG;
      ATTN_DMA_WAIT();
      __syncthreads();                         // chunk c landed; chunk c-1 fully consumed
      if (c + 1 < nch) {
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG, s0 + D, ldq, c + 1, Wh);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG + IMG, s0 + 2 * D, ldq, c + 1, Wh);
      }
      if (!active) continue;
      const unsigned base = base0 + (unsigned)c * cstep;
      const int rows_left = Wh - c * G::RPC;   // grid rows of this chunk that exist (wave-uniform)
      // bias prefetch: the reads of two 32-slot blocks are in flight while the score MFMAs run; block kb + 2 is issued
      // when block kb has been consumed (the loads depend on nothing but the lane's base)
      float bz[2][16];
      auto bias_issue = [&](int kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int s0i = kb * 32 + 8 * g;
          if (!G::valid(s0i) && !G::valid(s0i + 4)) continue;
          const auto* p = reinterpret_cast<const __attribute__((address_space(3))) F2u*>(base + 4u * (unsigned)G::imm(s0i));
#if defined(WIN_EXP) && WIN_EXP == 2     // timing experiment (wrong results): no bias reads
          (void)p;
          bz[kb & 1][4 * g] = bz[kb & 1][4 * g + 1] = bz[kb & 1][4 * g + 2] = bz[kb & 1][4 * g + 3] = bcls;
#else
          bz[kb & 1][4 * g] = p[0].a; bz[kb & 1][4 * g + 1] = p[0].b; bz[kb & 1][4 * g + 2] = p[1].a; bz[kb & 1][4 * g + 3] = p[1].b;
#endif
        }
      };
      bias_issue(0);
      bias_issue(1);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 s[CKB];
#pragma unroll
      for (int kb = 0; kb < CKB; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) s[kb] = MFMA32(row_frag_o(Ks, lo, kb, t), Qf[t], s[kb]);
      }
      float cmax = -INFINITY;
      float cls_raw = s[G::CLS_KB][4 * G::CLS_G];
      auto scores = [&](auto RAGGED) {
#pragma unroll
        for (int kb = 0; kb < CKB; ++kb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int s0i = kb * 32 + 8 * g;                   // slot of (hh = 0, e = 0); hh = 1: + 4
            const bool v0 = G::valid(s0i), v1 = G::valid(s0i + 4);
            if (!v0 && !v1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) s[kb][4 * g + e] = -INFINITY;
              continue;
            }
            bfr2(s[kb], 4 * g);
            bfr2(s[kb], 4 * g + 2);
            const bool rowdead = decltype(RAGGED)::value && G::row(s0i) >= rows_left;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = fmaf(s[kb][4 * g + e], kLog2e, bz[kb & 1][4 * g + e]);
              if (v0 != v1) v = (hh ? v1 : v0) ? v : -INFINITY;
              if (rowdead) v = -INFINITY;
              s[kb][4 * g + e] = v;
              cmax = fmaxf(cmax, v);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 2 < CKB) bias_issue(kb + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (rows_left < G::RPC) scores(std::true_type{}); else scores(std::false_type{});
      if (c == 0) {                            // the cls key: slot PAD0 = register (CLS_KB, CLS_G, e = 0) of the hh = 0 lanes
        const float v = hh == 0 ? fmaf(bfr(cls_raw), kLog2e, bcls) : -INFINITY;
        s[G::CLS_KB][4 * G::CLS_G] = v;
        cmax = fmaxf(cmax, v);
      }
      cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
      const float mn = fmaxf(m, cmax);         // finite from the first chunk on
      const float alpha = fexp2(m - mn);
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < CKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#if defined(WIN_EXP) && WIN_EXP == 1     // timing experiment (wrong results): no exp / sub / add per element
          const float p = s[kb][i];
#else
          const float p = fexp2(s[kb][i] - mn);
#endif
          s[kb][i] = p;
#if !(defined(WIN_EXP) && WIN_EXP == 1)
          sum += p;
#endif
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
#if defined(WIN_EXP) && WIN_EXP == 3     // timing experiment (wrong results): no PV MFMAs
        o[0][0] += (float)vf[0][0][0] + (float)vf[1][1][0] + (float)vf[0][1][0] + (float)vf[1][0][0] + (float)pf[0][0] + (float)pf[1][0];
#else
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int db = 0; db < 2; ++db) o[db] = MFMA32(vf[ss][db], pf[ss], o[db]);
#endif
      }
    }
    if (active) {
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

template <int WW>
size_t win_lds_fwd(int Wh) {
  using G = WinGeo<WW>;
  const int NB = (2 * Wh - 1) * G::P;
  return (size_t)(((NB + 3) & ~3) + G::CQ) * 4 + (size_t)4 * G::CT * 128;
}

template <int WW>
int launch_fwd(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table, int Wh, void* out,
               int64_t ldo, float* lse, hipStream_t s) {
  const int TP = ((T + 31) / 32) * 32;
  const int nrd = (2 * Wh - 1) * (2 * WW - 1) + 3;
  const size_t sm = win_lds_fwd<WW>(Wh);
  if (sm > (size_t)kMaxLds) return MEMHIP_EUNSUPPORTED;
  static bool done = false;
  if (int rc = set_lds_attr(attn_fwd_win_kernel<WW>, &done)) return rc;
  const int groups = (TP / 32 + 7) / 8;
  // samples per workgroup: the table set-up is paid once per workgroup; keep the grid a few rounds of the chip deep
  int nbz = B;
  const long long per = (long long)groups * heads;
  const int cus = usable_cus(s);
  while (nbz > 1 && per * nbz > 6LL * cus) nbz = (nbz + 1) / 2;
  hipLaunchKernelGGL(attn_fwd_win_kernel<WW>, dim3(groups, heads, nbz), dim3(512), sm, s, (const __bf16*)qkv, (long long)ldqkv,
                     B, T, TP, D, heads, table, nrd, Wh, (__bf16*)out, (long long)ldo, lse);
  return check_launch("attn_fwd(win)");
}

}  // namespace

namespace memhip {

bool attn_win_fits(int T, int window_h, int window_w) {
  return T > 256 && (window_w == 40 || window_w == 20) && T == window_h * window_w + 1;
}

int attn_fwd_win(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table, int window_h, int window_w,
                 void* out, int64_t ldo, float* lse, hipStream_t s) {
  if (window_w == 40) return launch_fwd<40>(qkv, ldqkv, B, T, D, heads, table, window_h, out, ldo, lse, s);
  if (window_w == 20) return launch_fwd<20>(qkv, ldqkv, B, T, D, heads, table, window_h, out, ldo, lse, s);
  return MEMHIP_EUNSUPPORTED;
}

}  // namespace memhip
