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
#include "attn_win_common.hpp"

namespace {

// ------------------------------------------------------------------------------- forward
// Phase A of a chunk: S^T = K_chunk Q^T (lane = resident query, registers = streamed slots), bf16 rounding, + bias (log2 domain),
// static / ragged masks, the cls key of chunk 0; returns the lane's maximum over the chunk's scores.
template <int WW>
__device__ __forceinline__ void fwd_phase_a(const char* Ks, const LaneOffs& lo, const bf16x8 (&Qf)[4], unsigned base, int rows_left,
                                            int c, int hh, float bcls, f32x16 (&s)[4], float& cmax_out) {
  using G = WinGeo<WW>;
  constexpr int CKB = G::CT / 32;
  const unsigned sel_lo = sel_lo_reg();
  // Software pipeline over the four 32-slot blocks: the score MFMAs of block kb + 1 are INTERLEAVED with the vector work of
  // block kb (one MFMA per ~9 vector instructions: a wave issues in order, so a run of 16 back-to-back MFMAs would hold it for
  // 512 cycles while the SIMD's vector issue idles -- both waves of a SIMD run this phase at about the same time); the K
  // fragments and bias words of block kb + 2 are read meanwhile (the bias reads depend on nothing but the lane's base).
  float bz[2][16];
  auto bias_issue = [&](int kb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int s0i = kb * 32 + 8 * g;
      if (!G::valid(s0i) && !G::valid(s0i + 4)) continue;
      const auto* p = reinterpret_cast<const __attribute__((address_space(3))) F2u*>(base + 4u * (unsigned)G::imm(s0i));
      bz[kb & 1][4 * g] = p[0].a; bz[kb & 1][4 * g + 1] = p[0].b; bz[kb & 1][4 * g + 2] = p[1].a; bz[kb & 1][4 * g + 3] = p[1].b;
    }
  };
  bf16x8 kf[2][4];
  auto kread = [&](int kb) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kf[kb & 1][t] = row_frag_o(Ks, lo, kb, t);
    }
  };
  auto chain = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s[kb] = MFMA32(kf[kb & 1][t], Qf[t], s[kb]);
  };
  float cmax = -INFINITY;
  float cls_raw = 0.f;
  auto process = [&](int kb, auto RAGGED) {
    if (kb == G::CLS_KB) cls_raw = s[kb][4 * G::CLS_G];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int s0i = kb * 32 + 8 * g;                   // slot of (hh = 0, e = 0); hh = 1: + 4
      const bool v0 = G::valid(s0i), v1 = G::valid(s0i + 4);
      if (!v0 && !v1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[kb][4 * g + e] = -INFINITY;
        continue;
      }
      const unsigned p0 = pk_bf16(s[kb][4 * g], s[kb][4 * g + 1]), p1 = pk_bf16(s[kb][4 * g + 2], s[kb][4 * g + 3]);
      float vv[4];
      vv[0] = add_lo(p0, bz[kb & 1][4 * g], sel_lo);
      vv[1] = add_hi(p0, bz[kb & 1][4 * g + 1]);
      vv[2] = add_lo(p1, bz[kb & 1][4 * g + 2], sel_lo);
      vv[3] = add_hi(p1, bz[kb & 1][4 * g + 3]);
      const bool rowdead = decltype(RAGGED)::value && G::row(s0i) >= rows_left;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = vv[e];
        if (v0 != v1) v = (hh ? v1 : v0) ? v : -INFINITY;
        if (rowdead) v = -INFINITY;
        s[kb][4 * g + e] = v;
        cmax = fmaxf(cmax, v);
      }
    }
  };
  auto run = [&](auto RAGGED) {
    kread(0);
    kread(1);
    bias_issue(0);
    bias_issue(1);
    chain(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < CKB; ++kb) {
      if (kb + 1 < CKB) chain(kb + 1);
      process(kb, RAGGED);
      if (kb + 2 < CKB) {
        kread(kb + 2);
        bias_issue(kb + 2);
      }
      if (kb + 1 < CKB) {
        // the LDS reads of block kb + 2 go FIRST (left to the scheduler they end up behind the region's last MFMA and the
        // next region opens with a wait for them: ~90 cycles per read exposed, 16 % of the kernel by removal)
        if (kb + 2 < CKB) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);     // nine vector instructions
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (rows_left < G::RPC) run(std::true_type{}); else run(std::false_type{});
  if (c == 0) {                            // the cls key: slot PAD0 = register (CLS_KB, CLS_G, e = 0) of the hh = 0 lanes
    const float v = hh == 0 ? bfr(cls_raw) + bcls : -INFINITY;
    s[G::CLS_KB][4 * G::CLS_G] = v;
    cmax = fmaxf(cmax, v);
  }
  cmax_out = cmax;
}

// Phase B: running maximum / sum (natural-log domain, exp2 of a packed fma), P = exp(S - m), O += P V.  Per 32-slot block:
// the exponentials of block kb + 1 are issued behind the PV MFMAs of block kb (the matrix pipe runs under the VALU work).
template <int WW>
__device__ __forceinline__ void fwd_phase_b(const char* Vs, const LaneOffs& lo, f32x16 (&s)[4], float cmax, float& m, float& l,
                                            f32x16 (&o)[2]) {
  using G = WinGeo<WW>;
  constexpr int CKB = G::CT / 32;
  cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
  const float mn = fmaxf(m, cmax);         // finite from the first chunk on
  const float alpha = fexp2((m - mn) * kLog2e);
  const f32x2_t mneg2 = {-mn * kLog2e, -mn * kLog2e}, l2e2 = {kLog2e, kLog2e};
  f32x2_t sum2 = {0.f, 0.f};
  m = mn;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
  const ColAddr vc = col_addr(Vs, lo);
  auto exps = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f32x2_t a = f32x2_t{s[kb][i], s[kb][i + 1]} * l2e2 + mneg2;
      const f32x2_t p = {fexp2(a[0]), fexp2(a[1])};
      sum2 += p;
      s[kb][i] = p[0];
      s[kb][i + 1] = p[1];
    }
  };
  // V fragments of block kb + 1 are read (transposing LDS reads, asm) BEFORE the exponentials of block kb + 1: the reads'
  // latency runs under that VALU work instead of in front of the block's MFMAs
  bf16x8 vf[2][2][2];
  auto vread = [&](auto KB) {
    constexpr int kb = decltype(KB)::value;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int db = 0; db < 2; ++db) vf[kb & 1][ss][db] = col_frag_i<kb * 4096>(vc, ss, db);
  };
  auto pv = [&](auto KB) {
    constexpr int kb = decltype(KB)::value;
    bf16x8 pf[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) pf[ss] = acc_frag(s[kb], ss, 1.0f);
    // the block's own 8 reads are complete; the 8 reads of the next block (issued behind them) may stay in flight
    // (the fragments are named as operands of the wait: an MFMA that reads them cannot be scheduled above it)
    if (kb + 1 < CKB)
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vf[kb & 1][0][0]), "+v"(vf[kb & 1][0][1]), "+v"(vf[kb & 1][1][0]), "+v"(vf[kb & 1][1][1])::"memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[kb & 1][0][0]), "+v"(vf[kb & 1][0][1]), "+v"(vf[kb & 1][1][0]), "+v"(vf[kb & 1][1][1])::"memory");
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int db = 0; db < 2; ++db) o[db] = MFMA32(vf[kb & 1][ss][db], pf[ss], o[db]);
  };
  static_assert(CKB == 4, "four 32-slot blocks per chunk");
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  auto hint = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);      // ten vector instructions (two of them exponentials)
    }
  };
  vread(I0{});
  exps(0);
  vread(I1{});
  __builtin_amdgcn_sched_barrier(0);
  pv(I0{});
  exps(1);
  hint();
  __builtin_amdgcn_sched_barrier(0);
  vread(I2{});
  pv(I1{});
  exps(2);
  hint();
  __builtin_amdgcn_sched_barrier(0);
  vread(I3{});
  pv(I2{});
  exps(3);
  hint();
  __builtin_amdgcn_sched_barrier(0);
  pv(I3{});
  float sum = sum2[0] + sum2[1];
  sum += __shfl_xor(sum, 32);
  l = fmaf(l, alpha, sum);
}

// A wave owns 32 resident queries (token order), the workgroup (8 waves, one per CU) streams K / V slot chunks, double-buffered,
// one barrier per chunk, and is persistent over the samples b = blockIdx.z, + gridDim.z, ...
// Measured and dropped (round 5, B = 64 x 16 heads x 1201 tokens; this kernel 815 us, attn_stream.hip 950-1070):
//   * two 4-wave workgroups per CU with single-buffered chunks (two barriers per chunk): 860 us;
//   * waves 4-7 half a chunk behind waves 0-3 (the gemm_p8 stagger; two barriers per chunk): 921 us;
//   * one barrier in the MIDDLE of the chunk + the next chunk's first K fragments / bias words read ahead of phase B: 885-908 us
//     (10 spilled registers);
//   * the LDS reads of block kb + 2 forced to the front of region kb: no change;
//   * THREE 4-wave workgroups per CU (<= 168 registers per lane: one block of K fragments / bias words at a time, K and V
//     chunks single-buffered, 51.5 KB of LDS each): 1495 us -- 57 spilled registers and a phase A with nothing in flight
//     cost far more than the third wave per SIMD hides.
// Removal experiments (-DWIN_EXP, profiles/r05_attn_win_fwd_removals.txt): without the LDS reads of phase A (16 K-fragment and
// 30 bias reads per chunk and wave) -28 %, without the score MFMAs -14 %, without the exponentials -12 %, without barrier and
// staging -15 %, with the barrier kept but no LDS-DMA in the chunk loop -17 % (a per-lane staging plan that cuts the ~30
// vector instructions per piece to a multiply-add + a scalar-base load changed nothing: it is the transfer, not its issue);
// counters (profiles/r05_attn_win_fwd_pmc.txt): a wave issues 34 % of its cycles, is parked at a wait or the
// barrier 33 % and is issue-stalled 33 %; the SIMD's vector unit is busy ~53 %, the matrix pipe 21 %.
#define WIN_Q_CKF_EARLY 0   // dQ kernel with the table gradient: 1 = the K column fragments of the dQ product are read in front of the block's bucket atomics as well
#define WIN_DMA_LATE 0   // 1: forward: the LDS-DMA of chunk c + 1 is issued between phase A and phase B of chunk c
#define WIN_PRIO 0     // 1: waves 4-7 (the second-dispatched, arbitration-losing half of the workgroup) run at priority 1 (measured: forward 834 vs 812 us, backward unchanged)
#ifdef WIN_STAMP
// diagnostic build (tools/build_variant_fast.sh ... -DWIN_STAMP): waves 0 and 4 of every workgroup accumulate shader cycles
// (s_memtime) per section of the chunk loop: [0] DMA wait + barrier + staging issue, [1] phase A, [2] phase B, [3] chunks
__device__ unsigned long long g_win_stamps[1024 * 8];
#define WIN_T(var) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); } while (0)
#else
#define WIN_T(var) do { } while (0)
#endif
template <int WW>
__global__ __launch_bounds__(512) void attn_fwd_win_kernel(
    const __bf16* __restrict__ qkv, long long ldq, int B, int T, int TP, int D, int H, const float* __restrict__ table, int nrd,
    int Wh, __bf16* __restrict__ out, long long ldo, float* __restrict__ lse, int groups, int nbz) {
  using G = WinGeo<WW>;
  constexpr int CT = G::CT, IMG = CT * 128, CKB = CT / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NB = (2 * Wh - 1) * G::P;
  float* R = reinterpret_cast<float*>(smem);
  float* Cq = R + ((NB + 3) & ~3);
  char* imgs = reinterpret_cast<char*>(Cq + G::CQ);
  const WinWg wg_ = win_wg(groups, H, nbz);
  if (!wg_.live) return;
  const int h = wg_.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  win_setup<WW>(R, Cq, table, nrd, H, h, Wh, 1.0f, true, nrd - 3);
  const int qb = wg_.group * 8 + wave;
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
  const float bcls = table[(long long)(q == 0 ? nrd - 1 : nrd - 2) * H + h];      // bias towards the cls key
  const int nch = (Wh + G::RPC - 1) / G::RPC;
#ifdef WIN_STAMP
  unsigned long long st_acc[4] = {0, 0, 0, 0};
#endif
  for (int b = wg_.bz; b < B; b += nbz) {
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
      const unsigned base = base0 + (unsigned)c * cstep;
      const int rows_left = Wh - c * G::RPC;   // grid rows of this chunk that exist (wave-uniform)
      f32x16 s[CKB];
      float cmax = -INFINITY;
      const int cur = c & 1;
      const char* Ks = imgs + cur * 2 * IMG;
      const char* Vs = Ks + IMG;
      unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      (void)t0; (void)t1; (void)t2; (void)t3;
      WIN_T(t0);
      ATTN_DMA_WAIT();
      __syncthreads();                         // chunk c landed; chunk c-1 fully consumed
      if (c + 1 < nch && !WIN_DMA_LATE) {
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG, s0 + D, ldq, c + 1, Wh);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG + IMG, s0 + 2 * D, ldq, c + 1, Wh);
      }
      WIN_T(t1);
      if (active) fwd_phase_a<WW>(Ks, lo, Qf, base, rows_left, c, hh, bcls, s, cmax);
      WIN_T(t2);
      if (WIN_DMA_LATE && c + 1 < nch) {           // the next chunk's transfer starts under phase B (few LDS reads) instead of phase A
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG, s0 + D, ldq, c + 1, Wh);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG + IMG, s0 + 2 * D, ldq, c + 1, Wh);
      }
      if (!active) continue;
      fwd_phase_b<WW>(Vs, lo, s, cmax, m, l, o);
      WIN_T(t3);
#ifdef WIN_STAMP
      st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += 1;
#endif
    }
    if (active) {
      const float inv = 1.0f / l;
      if (hh == 0 && q < T) lse[((long long)b * H + h) * TP + q] = m + flog2(l) * kLn2;
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
#ifdef WIN_STAMP
  if ((wave == 0 || wave == 4) && lane == 0) {
    const int wg = (int)blockIdx.x & 1023;
#pragma unroll
    for (int i = 0; i < 4; ++i) g_win_stamps[wg * 8 + (wave >> 2) * 4 + i] = st_acc[i];
  }
#endif
}

// ------------------------------------------------------------------------------- backward (dK, dV)
// A wave owns 32 resident keys (token order; K, V fragments in registers), the workgroup streams Q' / dO chunks in the SLOT
// layout over queries: lane = key, registers = query slots, bucket(q, k) at Kp(k) + qy P + qx with the table in forward order.
// Per-query scalars travel by slot as well: -lse log2(e) (= -inf for padding slots: their probabilities are exactly 0, no
// compare anywhere) and -delta.  Same outputs and rounding points as attn_bwd_kv_stream_kernel.
template <int WW, bool VB>
__global__ __launch_bounds__(512) void attn_bwd_kv_win_kernel(
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
  float* nlS = Cq + G::CQ;                                   // [2][CT]  -lse * log2(e) by slot
  float* ndS = nlS + 2 * CT;                                 // [2][CT]  -delta by slot
  float* vsum = ndS + 2 * CT;                                // [8 waves][64]: v_bias gradient, a private row per wave
  char* imgs = reinterpret_cast<char*>(vsum + 8 * HD);
  const WinWg wg_ = win_wg(groups, H, nbz);
  if (!wg_.live) return;
  const int h = wg_.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  win_setup<WW>(R, Cq, table, nrd, H, h, Wh, 1.0f, false, nrd - 2);
  vsum[threadIdx.x] = 0.f;                                   // (512 threads = 8 x 64)
  const unsigned sel_lo = sel_lo_reg();
  const int kbg = wg_.group * 8 + wave;
  const bool active = kbg * 32 < T;
  const int key = kbg * 32 + r;
  const int kc_tok = key < T ? key : T - 1;
  unsigned base0, cstep;
  if (key == 0 || key >= T) {
    base0 = lds_addr_of(reinterpret_cast<const char*>(Cq));
    cstep = 0;
  } else {
    const int u = key - 1, ky = u / WW, kx = u - ky * WW;
    base0 = lds_addr_of(reinterpret_cast<const char*>(R)) + 4u * (unsigned)((Wh - 1 - ky) * G::P + (WW - 1 - kx));
    cstep = 4u * G::RPC * G::P;
  }
  base0 += 16u * hh;
  const float bcls = table[(long long)(key == 0 ? nrd - 1 : nrd - 3) * H + h];       // bias from the cls query
  const float kmask = key < T ? 1.f : 0.f;
  const bool kpad = __builtin_amdgcn_readfirstlane(kbg) * 32 + 32 > T;               // this wave holds keys >= T
  const int nch = (Wh + G::RPC - 1) / G::RPC;
  float vmax = 0.f, dmax = 0.f, nmax = 0.f;
  for (int b = wg_.bz; b < B; b += nbz) {
    const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
    const __bf16* d0 = dout + (long long)b * T * ldo + h * HD;
    bf16x8 Kf[4], Vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Kf[t] = ld16(s0 + (long long)kc_tok * ldq + D + 16 * t + 8 * hh);
      Vf[t] = ld16(s0 + (long long)kc_tok * ldq + 2 * D + 16 * t + 8 * hh);
    }
    if (stats) {
      float vn = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) vn = fmaf((float)Vf[t][i], (float)Vf[t][i], vn);
      vn += __shfl_xor(vn, 32);
      vmax = fmaxf(vmax, vn);
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
    f32x16 dVt[2], dKt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dVt[db][i] = 0.f; dKt[db][i] = 0.f; }
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
      const unsigned base = base0 + (unsigned)c * cstep;
      const float* nlC = nlS + cur * CT;
      const float* ndC = ndS + cur * CT;
      const ColAddr qa = col_addr(Qs, lo), da = col_addr(dOs, lo);
      auto block = [&](auto QB) {
        constexpr int qb = decltype(QB)::value;
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
          const int s0i = qb * 32 + 8 * g;                   // slot of (hh = 0, e = 0); hh = 1: + 4
          const int ql = s0i + 4 * hh;
          const float4 lv = *reinterpret_cast<const float4*>(nlC + ql);
          const float4 dv = *reinterpret_cast<const float4*>(ndC + ql);
          const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
          float bz[4];
          if (G::valid(s0i) || G::valid(s0i + 4)) {
            const auto* p = reinterpret_cast<const __attribute__((address_space(3))) F2u*>(base + 4u * (unsigned)G::imm(s0i));
            bz[0] = p[0].a; bz[1] = p[0].b; bz[2] = p[1].a; bz[3] = p[1].b;
          } else {
            bz[0] = bz[1] = bz[2] = bz[3] = 0.f;
          }
          if (qb == G::CLS_KB && g == G::CLS_G && c == 0 && hh == 0) bz[0] = bcls;      // the cls query's slot
          const unsigned s01 = pk_bf16(S[4 * g], S[4 * g + 1]), s23 = pk_bf16(S[4 * g + 2], S[4 * g + 3]);
          const unsigned d01 = pk_bf16(dP[4 * g], dP[4 * g + 1]), d23 = pk_bf16(dP[4 * g + 2], dP[4 * g + 3]);
          float sv[4], dq[4];
          sv[0] = add_lo(s01, bz[0], sel_lo); sv[1] = add_hi(s01, bz[1]); sv[2] = add_lo(s23, bz[2], sel_lo); sv[3] = add_hi(s23, bz[3]);
          dq[0] = add_lo(d01, dd[0], sel_lo); dq[1] = add_hi(d01, dd[1]); dq[2] = add_lo(d23, dd[2], sel_lo); dq[3] = add_hi(d23, dd[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float p = fexp2(fmaf(sv[e], kLog2e, ll[e]));      // ll = -lse log2(e); -inf for padding slots: p = 0
            if (kpad) p *= kmask;                             // (wave-uniform: only the wave that holds padding keys)
            S[4 * g + e] = p;
            dP[4 * g + e] = p * dq[e];
          }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          bf16x8 cdo[2], cq[2];
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            cdo[db] = col_frag_i<qb * 4096>(da, ss, db);
            cq[db] = col_frag_i<qb * 4096>(qa, ss, db);
          }
          const bf16x8 pf = acc_frag(S, ss, 1.0f), dsf = acc_frag(dP, ss, 1.0f);
          LDS_TR_WAIT();
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dVt[db] = MFMA32(cdo[db], pf, dVt[db]);
            dKt[db] = MFMA32(cq[db], dsf, dKt[db]);
          }
        }
      };
      static_assert(CKB == 4, "four 32-slot blocks per chunk");
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
      block(std::integral_constant<int, 2>{});
      block(std::integral_constant<int, 3>{});
    }
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
          if constexpr (VB)
            colsum_add4(vsum + wave * HD, r, hh, db, g, (float)wv[0] * kmask, (float)wv[1] * kmask, (float)wv[2] * kmask,
                        (float)wv[3] * kmask);
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
      for (int w = 0; w < 8; ++w) v += vsum[w * HD + threadIdx.x];
      atomicAdd(dvbias + h * HD + threadIdx.x, v);
    }
  }
}

// ------------------------------------------------------------------------------- backward, dS-storing form (round 6)
// The two-kernel backward above computes the score tile twice: S, dP, the exponentials and dS once with the key on the lane
// (dK, dV) and once with the query on the lane (dQ, table gradient) -- 28 MFMAs and ~740 vector instructions per 32 x 32 tile.
// Here the dK / dV kernel keeps what it has already computed: dS (bf16, the very fragments it feeds to the dK product) goes to a
// workspace as dS^T[key token][query slot], and it owns the table gradient (its lanes are keys: bucket = Kp(k) + qy P + qx, the
// same base + immediate address as its bias reads).  The dQ kernel is then a plain product dQ = dS K that streams dS^T and K
// through LDS (transposing reads turn the [key][slot] image into fragments with the key as the contraction index): 4 MFMAs and
// a few vector instructions per tile, bound by the 2 B per score element it reads.  Same outputs and rounding points.
//   workspace: B * H * TP * QS bf16 in tiles of 32 keys x 16 slots, QS = 128 * ceil(Wh / RPC) query slots per key (3.2 GB at
//   B = 64, 16 heads, 30 x 40)
template <int WW, bool VB, bool DT>
__global__ __launch_bounds__(512) void attn_bwd_kvs_win_kernel(
    const __bf16* __restrict__ qkv, long long ldq, const __bf16* __restrict__ dout, long long ldo,
    const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ stats,
    const float* __restrict__ table, int nrd, int Wh, __bf16* __restrict__ dqkv, long long lddq,
    float* __restrict__ dvbias, float* __restrict__ dtable, __bf16* __restrict__ dS, int QS,
    int B, int T, int TP, int D, int H, int groups, int nbz) {
  using G = WinGeo<WW>;
  constexpr int CT = G::CT, IMG = CT * 128, CKB = CT / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NB = (2 * Wh - 1) * G::P, NBP = (NB + 3) & ~3;
  float* R = reinterpret_cast<float*>(smem);
  float* Cq = R + NBP;
  float* Cn = Cq + G::CQ;                                    // strip of -inf: the "bias" of a padded key (its probabilities are exactly 0:
                                                             // no compare and no multiply per element)
  int* binsR = reinterpret_cast<int*>(Cn + G::CQ);           // DT: fixed-point gradient buckets, an image of [R | Cq | Cn]
  constexpr int kRed = 16;
  float* red = reinterpret_cast<float*>(binsR + (DT ? NBP + 2 * G::CQ : 0));     // [16] workgroup reduction scratch
  float* nlS = red + kRed;                                   // [2][CT]  -lse * log2(e) by slot
  float* ndS = nlS + 2 * CT;                                 // [2][CT]  -delta by slot
  float* vsum = ndS + 2 * CT;                                // [8 waves][64]: v_bias gradient, a private row per wave
  char* imgs = reinterpret_cast<char*>(vsum + 8 * HD);
  const WinWg wg_ = win_wg(groups, H, nbz);
  if (!wg_.live) return;
  const int h = wg_.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  win_setup<WW>(R, Cq, table, nrd, H, h, Wh, 1.0f, false, nrd - 2);
  vsum[threadIdx.x] = 0.f;                                   // (512 threads = 8 x 64)
  for (int i = threadIdx.x; i < G::CQ; i += blockDim.x) Cn[i] = -INFINITY;
  if (DT)
    for (int i = threadIdx.x; i < NBP + 2 * G::CQ; i += blockDim.x) binsR[i] = 0;
  const unsigned bins_delta = (unsigned)(NBP + 2 * G::CQ) * 4u;
  const unsigned sel_lo = sel_lo_reg();
  const int kbg = wg_.group * 8 + wave;
  const bool active = kbg * 32 < T;
  const int key = kbg * 32 + r;
  const int kc_tok = key < T ? key : T - 1;
  unsigned base0, cstep;
  if (key >= T) {
    base0 = lds_addr_of(reinterpret_cast<const char*>(Cn));
    cstep = 0;
  } else if (key == 0) {
    base0 = lds_addr_of(reinterpret_cast<const char*>(Cq));
    cstep = 0;
  } else {
    const int u = key - 1, ky = u / WW, kx = u - ky * WW;
    base0 = lds_addr_of(reinterpret_cast<const char*>(R)) + 4u * (unsigned)((Wh - 1 - ky) * G::P + (WW - 1 - kx));
    cstep = 4u * G::RPC * G::P;
  }
  base0 += 16u * hh;
  const float bcls = key >= T ? -INFINITY : table[(long long)(key == 0 ? nrd - 1 : nrd - 3) * H + h];       // bias from the cls query
  const float kmask = key < T ? 1.f : 0.f;
  const int nch = (Wh + G::RPC - 1) / G::RPC;
  // DT: the fixed-point scale of the buckets.  |dS| <= P (|dO_q| |V_k| + |delta_q|): max |dO_q|^2 and max |delta_q| of the head come
  // from the statistics pass over the delta arrays (stats[h][0..1]), max |V_k|^2 is taken HERE over the keys this workgroup holds
  // in all of its samples (one extra read of its V rows, served again from L2 in the sample loop).  A bucket collects at most
  // one term per resident key and sample (the bucket index is injective in the query): <= 256 * 16 terms of magnitude <= 2^18.
  float fx = 0.f, gcls = 0.f;
  if (DT) {
    float vm = 0.f;
    for (int b = wg_.bz; b < B; b += nbz) {
      const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
      float vn = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 v = ld16(s0 + (long long)kc_tok * ldq + 2 * D + 16 * t + 8 * hh);
#pragma unroll
        for (int i = 0; i < 8; ++i) vn = fmaf((float)v[i], (float)v[i], vn);
      }
      vn += __shfl_xor(vn, 32);
      vm = fmaxf(vm, vn);
    }
    for (int o = 16; o > 0; o >>= 1) vm = fmaxf(vm, __shfl_xor(vm, o));
    if (lane == 0) red[wave] = vm;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 8; ++w) vm = fmaxf(vm, red[w]);
    const float bound = sqrtf(stats[h * 4 + 0]) * sqrtf(vm) + stats[h * 4 + 1];
    fx = bound > 0.f ? 262144.0f / bound : 0.f;
  }
  for (int b = wg_.bz; b < B; b += nbz) {
    const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
    const __bf16* d0 = dout + (long long)b * T * ldo + h * HD;
    bf16x8 Kf[4], Vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Kf[t] = ld16(s0 + (long long)kc_tok * ldq + D + 16 * t + 8 * hh);
      Vf[t] = ld16(s0 + (long long)kc_tok * ldq + 2 * D + 16 * t + 8 * hh);
    }
    // dS^T leaves in TILES of 32 keys x 16 query slots (1 KB, [key][slot]): tile (key block, slot block) of (b, h) at
    // ((b H + h) (TP / 32) + key block) (QS / 16) + slot block -- one store instruction of the wave (fixed slot block) writes
    // one whole tile, 1 KB contiguous (rows of a [key][slot] matrix took 32 lines of 32 B per instruction: +376 us)
    __bf16* dstile = dS + ((((long long)b * H + h) * (TP / 32) + kbg) * (QS / 16)) * 512 + r * 16 + hh * 8;
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
    };
    load_next(0);
    __syncthreads();                                         // the previous sample's last chunk is consumed (and the setup done)
    stage_chunk_win<WW>(imgs, s0, ldq, 0, Wh);               // Q'
    stage_chunk_win<WW>(imgs + IMG, d0, ldo, 0, Wh);         // dO
    f32x16 dVt[2], dKt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dVt[db][i] = 0.f; dKt[db][i] = 0.f; }
    for (int c = 0; c < nch; ++c) {
      const int cur = c & 1;
      const char* Qs = imgs + cur * 2 * IMG;
      const char* dOs = Qs + IMG;
      if ((int)threadIdx.x < CT) { nlS[cur * CT + threadIdx.x] = nln; ndS[cur * CT + threadIdx.x] = ndn; }
      // chunk c has landed when everything but this wave's 8 dS stores of chunk c - 1 (issued behind its LDS-DMA pieces; vmcnt
      // counts loads and stores together, in issue order) is done: a full drain would wait for the stores' acknowledgements
      // once per chunk (measured: dK / dV kernel 1 573 us with the drain against 1 197 without the stores)
      // (and the barrier is the bare s_barrier behind a wait for this wave's LDS writes: __syncthreads() carries a release
      // fence, for which hipcc drains the wave's outstanding global stores -- vmcnt(0) -- in front of every barrier)
      if (c > 0 && active) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else ATTN_DMA_WAIT();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < nch) {
        load_next(c + 1);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG, s0, ldq, c + 1, Wh);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG + IMG, d0, ldo, c + 1, Wh);
      }
      if (!active) continue;
      const unsigned base = base0 + (unsigned)c * cstep;
      const float* nlC = nlS + cur * CT;
      const float* ndC = ndS + cur * CT;
      const ColAddr qa = col_addr(Qs, lo), da = col_addr(dOs, lo);
      auto block = [&](auto QB) {
        constexpr int qb = decltype(QB)::value;
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
          const int s0i = qb * 32 + 8 * g;                   // slot of (hh = 0, e = 0); hh = 1: + 4
          const int ql = s0i + 4 * hh;
          const float4 lv = *reinterpret_cast<const float4*>(nlC + ql);
          const float4 dv = *reinterpret_cast<const float4*>(ndC + ql);
          const float ll[4] = {lv.x, lv.y, lv.z, lv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
          float bz[4];
          if (G::valid(s0i) || G::valid(s0i + 4)) {
            const auto* p = reinterpret_cast<const __attribute__((address_space(3))) F2u*>(base + 4u * (unsigned)G::imm(s0i));
            bz[0] = p[0].a; bz[1] = p[0].b; bz[2] = p[1].a; bz[3] = p[1].b;
          } else {
            bz[0] = bz[1] = bz[2] = bz[3] = 0.f;
          }
          if (qb == G::CLS_KB && g == G::CLS_G && c == 0 && hh == 0) bz[0] = bcls;      // the cls query's slot
          const unsigned s01 = pk_bf16(S[4 * g], S[4 * g + 1]), s23 = pk_bf16(S[4 * g + 2], S[4 * g + 3]);
          const unsigned d01 = pk_bf16(dP[4 * g], dP[4 * g + 1]), d23 = pk_bf16(dP[4 * g + 2], dP[4 * g + 3]);
          float sv[4], dq[4];
          sv[0] = add_lo(s01, bz[0], sel_lo); sv[1] = add_hi(s01, bz[1]); sv[2] = add_lo(s23, bz[2], sel_lo); sv[3] = add_hi(s23, bz[3]);
          dq[0] = add_lo(d01, dd[0], sel_lo); dq[1] = add_hi(d01, dd[1]); dq[2] = add_lo(d23, dd[2], sel_lo); dq[3] = add_hi(d23, dd[3]);
          const bool v0 = G::valid(s0i), v1 = G::valid(s0i + 4);
          const bool clsg = qb == G::CLS_KB && g == G::CLS_G;           // the group that holds the cls query's slot (chunk 0, hh = 0, e = 0)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = fexp2(fmaf(sv[e], kLog2e, ll[e]));      // ll = -lse log2(e): -inf for padding slots; sv = -inf for padded keys: p = 0
            const float ds = p * dq[e];
            S[4 * g + e] = p;
            dP[4 * g + e] = ds;
            if (DT) {
              // every dead element (padding slot, ragged row, padded key) has p = 0 exactly: it adds 0 to whatever word of the
              // bucket image its bias address maps to
              if (clsg) {
                if (e == 0) gcls += ds;                             // the bucket of (cls query, this key): a lane register
              } else if (v0 || v1) {
                __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) int*>(base + 4u * (unsigned)G::imm(s0i) + bins_delta + 4u * e),
                                       fx_round(ds, fx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          bf16x8 cdo[2], cq[2];
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            cdo[db] = col_frag_i<qb * 4096>(da, ss, db);
            cq[db] = col_frag_i<qb * 4096>(qa, ss, db);
          }
          const bf16x8 pf = acc_frag(S, ss, 1.0f), dsf = acc_frag(dP, ss, 1.0f);
          {
            // dS^T leaves for the dQ kernel as 16-byte pieces: the lane holds, for its key, the query slots 16 ss + 4 hh + {0..3}
            // (dwords 0, 1) and 16 ss + 8 + 4 hh + {0..3} (dwords 2, 3); the two halves of the wave exchange one pair
            // (v_permlane32_swap: the upper lanes' first operand <-> the lower lanes' second), after which a lower lane holds
            // slots 16 ss + 0..7 and an upper lane slots 16 ss + 8..15 of its key
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            const u32x4_t d = __builtin_bit_cast(u32x4_t, dsf);
            const auto x0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            const auto x1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            const u32x4_t w = {x0[0], x1[0], x0[1], x1[1]};
            *reinterpret_cast<u32x4_t*>(dstile + (long long)(c * (CT / 16) + qb * 2 + ss) * 512) = w;
          }
          LDS_TR_WAIT();
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dVt[db] = MFMA32(cdo[db], pf, dVt[db]);
            dKt[db] = MFMA32(cq[db], dsf, dKt[db]);
          }
        }
      };
      static_assert(CKB == 4, "four 32-slot blocks per chunk");
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
      block(std::integral_constant<int, 2>{});
      block(std::integral_constant<int, 3>{});
    }
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
          if constexpr (VB)
            colsum_add4(vsum + wave * HD, r, hh, db, g, (float)wv[0] * kmask, (float)wv[1] * kmask, (float)wv[2] * kmask,
                        (float)wv[3] * kmask);
        }
    }
  }
  if (DT) {
    __syncthreads();
    const float inv = fx > 0.f ? 1.0f / fx : 0.f;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) {
      const int v = binsR[i];
      if (v != 0) atomicAdd(dtable + (long long)i * H + h, (float)v * inv);
    }
    if (wave == 0) {                                         // the cls KEY: the strip behind the buckets
      float v = 0.f;
      for (int i = lane; i < G::CQ; i += 64) v += (float)binsR[NBP + i] * inv;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && v != 0.f) atomicAdd(dtable + (long long)(nrd - 2) * H + h, v);
    }
    if (active) {                                            // the cls QUERY: lane sums ((cls, cls) is the cls key's own term)
      float rowv = (key != 0 && key < T) ? gcls : 0.f;
      const float both = key == 0 ? gcls : 0.f;
      rowv += __shfl_xor(rowv, 32);                          // (only hh = 0 lanes hold terms)
      for (int o = 16; o > 0; o >>= 1) rowv += __shfl_xor(rowv, o);
      if (lane == 0 && rowv != 0.f) atomicAdd(dtable + (long long)(nrd - 3) * H + h, rowv);
      if (key == 0 && hh == 0 && both != 0.f) atomicAdd(dtable + (long long)(nrd - 1) * H + h, both);
    }
  }
  if (VB) {
    __syncthreads();
    if (threadIdx.x < HD) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += vsum[w * HD + threadIdx.x];
      atomicAdd(dvbias + h * HD + threadIdx.x, v);
    }
  }
}


// statistics of the head for the fixed-point table gradient, from the delta arrays (rowsum(dO * O) and |dO|^2 per (token, head)):
// stats[h][0] = max |dO_q|^2, stats[h][1] = max |delta_q| (bit patterns of non-negative floats order like ints)
__global__ __launch_bounds__(256) void attn_win_stats_kernel(const float* __restrict__ delta, long long rows, int H,
                                                             float* __restrict__ stats) {
  // block = 256 consecutive (row, head) pairs per step: thread t always meets head (t0 + t) % H when the step is a multiple of
  // H (the launcher makes gridDim.x a multiple of H); the block folds its 256 partial maxima per head in LDS and issues ONE
  // atomic pair per head (65 536 same-address atomics cost 160 us)
  __shared__ float sd[256], sn[256];
  const long long n = rows * H;
  float dm = 0.f, nm = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    dm = fmaxf(dm, fabsf(delta[i]));
    nm = fmaxf(nm, delta[n + i]);
  }
  sd[threadIdx.x] = dm; sn[threadIdx.x] = nm;
  __syncthreads();
  const int h0 = (int)(((long long)blockIdx.x * blockDim.x) % H);       // head of thread 0
  if ((int)threadIdx.x < H) {
    // threads t with (h0 + t) % H == h hold head h; this thread folds head (h0 + threadIdx.x) % H
    for (int t = threadIdx.x + H; t < 256; t += H) { dm = fmaxf(dm, sd[t]); nm = fmaxf(nm, sn[t]); }
    const int h = (h0 + threadIdx.x) % H;
    atomicMax(reinterpret_cast<int*>(stats) + h * 4 + 0, __float_as_int(nm));
    atomicMax(reinterpret_cast<int*>(stats) + h * 4 + 1, __float_as_int(dm));
  }
}

// dQ = dS K over the stored dS^T.  A workgroup owns 256 consecutive query SLOTS of one (sample, head) -- wave w the slots
// 32 w .. 32 w + 31, waves 2 j and 2 j + 1 share the 64-slot-wide (128-byte rows) image j -- and streams the keys in TOKEN order
// in chunks of 64: per chunk the K rows (8 KB) and four dS^T images of 64 keys x 64 slots (8 KB each), three chunks in flight.
template <int WW>
__global__ __launch_bounds__(512) void attn_bwd_qs_win_kernel(
    const __bf16* __restrict__ qkv, long long ldq, const __bf16* __restrict__ dS, int QS, int Wh, __bf16* __restrict__ dqkv,
    long long lddq, float* __restrict__ dqbias, int B, int T, int TP, int D, int H, float scale, int groups, int nbz) {
  using G = WinGeo<WW>;
  constexpr int CK = 64, IMGK = CK * 128, NBUF = 3, BUF = 5 * IMGK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qsum = reinterpret_cast<float*>(smem);              // [8 waves][64]: q_bias gradient, a private row per wave
  char* imgs = reinterpret_cast<char*>(qsum + 8 * HD);
  const WinWg wg_ = win_wg(groups, H, nbz);
  if (!wg_.live) return;
  const int h = wg_.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  qsum[threadIdx.x] = 0.f;
  const int slot0 = wg_.group * 256;
  const int slot = slot0 + wave * 32 + r;                    // this lane's query slot
  const bool active = slot0 + wave * 32 < QS;
  // slot -> token (the slot layout of the dK / dV kernel's streamed queries: chunk of 128 = RPC grid rows of WS slots)
  int tok = -1;
  {
    const int c = slot / G::CT, ls = slot - c * G::CT;
    const int j = ls / G::WS, qx = ls - j * G::WS, qy = c * G::RPC + j;
    if (ls < G::PAD0 && qx < WW && qy < Wh) tok = 1 + qy * WW + qx;
    if (c == 0 && ls == G::PAD0) tok = 0;
    if (slot >= QS) tok = -1;
  }
  const float qmask = tok >= 0 ? 1.f : 0.f;
  const int pair = wave >> 1, half = wave & 1;
  const int nck = (TP + CK - 1) / CK;
  // staging: per chunk 5 images x 8 pieces of 8 rows = 40 wave-instructions, 5 per wave (piece = wave + 8 i)
  auto stage = [&](int b, int j, int buf) {
    char* dst = imgs + buf * BUF;
    const __bf16* kbase = qkv + (long long)b * T * ldq + D + h * HD;
    const __bf16* sbase = dS + (((long long)b * H + h) * (TP / 32)) * (QS / 16) * 512;        // tiles of 32 keys x 16 slots, 1 KB each
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int piece = wave + 8 * i, img = piece >> 3, inst = piece & 7;      // img 0: K, 1..4: dS^T of wave pair img - 1
      const int lt = inst * 8 + (lane >> 3), cpos = lane & 7;
      const int chunk = cpos ^ img_key(lt);
      const int t = j * CK + lt;
      const void* g;
      if (img == 0) g = t < T ? (const void*)(kbase + (long long)t * ldq + chunk * 8) : (const void*)(g_attn_zero_page + cpos * 16);
      else {
        // 16-byte piece = slots 8 chunk .. 8 chunk + 7 of the pair's 64: slot block (slot0 + 64 (img - 1)) / 16 + chunk / 2, half chunk % 2
        const long long tile = (long long)(t >> 5) * (QS / 16) + ((slot0 + (img - 1) * 64) >> 4) + (chunk >> 1);
        g = (t < TP && slot0 + (img - 1) * 64 < QS) ? (const void*)(sbase + tile * 512 + (t & 31) * 16 + (chunk & 1) * 8)
                                                    : (const void*)(g_attn_zero_page + cpos * 16);
      }
      glds16(g, dst + img * IMGK + inst * 1024);
    }
  };
  for (int b = wg_.bz; b < B; b += nbz) {
    f32x16 dQt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) dQt[db][i] = 0.f;
    __syncthreads();                                         // the previous sample's last chunks are consumed
    stage(b, 0, 0);
    if (nck > 1) stage(b, 1, 1);
    for (int j = 0; j < nck; ++j) {
      // chunk j has landed when all but this wave's newest 5 LDS-DMA pieces (chunk j + 1) are done
      if (j + 1 < nck) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                       // ... for every wave; chunk j - 1 is consumed
      if (j + 2 < nck) stage(b, j + 2, (j + 2) % NBUF);
      if (!active) continue;
      const char* Ks = imgs + (j % NBUF) * BUF;
      const char* Ss = Ks + (1 + pair) * IMGK;
      const ColAddr ka = col_addr(Ks, lo);
      ColAddr sa = col_addr(Ss, lo);
      if (half) {                                            // this wave's 32 slots are the second column block of the pair's image
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) { sa.a[ss][0][0] = sa.a[ss][1][0]; sa.a[ss][0][1] = sa.a[ss][1][1]; }
      }
      auto block = [&](auto KB) {
        constexpr int kb = decltype(KB)::value;
        bf16x8 ckf[2][2], dsf[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          dsf[ss] = col_frag_i<kb * 4096>(sa, ss, 0);
#pragma unroll
          for (int db = 0; db < 2; ++db) ckf[ss][db] = col_frag_i<kb * 4096>(ka, ss, db);
        }
        LDS_TR_WAIT();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int db = 0; db < 2; ++db) dQt[db] = MFMA32(ckf[ss][db], dsf[ss], dQt[db]);
      };
      block(std::integral_constant<int, 0>{});
      block(std::integral_constant<int, 1>{});
    }
    if (active) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (__bf16)(bfr(dQt[db][4 * g + e]) * scale);
          if (tok >= 0)
            *reinterpret_cast<bf16x4*>(dqkv + ((long long)b * T + tok) * lddq + h * HD + db * 32 + 8 * g + 4 * hh) = w;
          if (dqbias)                                                                   // q_bias gradient
            colsum_add4(qsum + wave * HD, r, hh, db, g, (float)w[0] * qmask, (float)w[1] * qmask, (float)w[2] * qmask, (float)w[3] * qmask);
        }
    }
  }
  __syncthreads();
  if (dqbias && threadIdx.x < HD) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += qsum[w * HD + threadIdx.x];
    atomicAdd(dqbias + h * HD + threadIdx.x, v);
  }
}

// ------------------------------------------------------------------------------- backward (dQ, dBias)
// A wave owns 32 resident queries (Q', dO fragments, lse, delta in registers), the workgroup streams K / V slot chunks (the
// forward's orientation: reversed table, bucket at A(q) + ky P + kx) and is persistent over its samples, so the fixed-point
// gradient buckets -- an integer image of [R | Cq] -- are flushed once.  A bucket collects at most one term per resident
// query and sample (the bucket index is injective in the key): <= 256 * samples terms of magnitude <= 2^18.
// The cls KEY column is one static register of chunk 0: its gradient is summed in a lane register.
template <int WW, bool DT>
__global__ __launch_bounds__(512) void attn_bwd_q_win_kernel(
    const __bf16* __restrict__ qkv, long long ldq, const __bf16* __restrict__ dout, long long ldo,
    const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ stats,
    const float* __restrict__ table, int nrd, int Wh, __bf16* __restrict__ dqkv, long long lddq,
    float* __restrict__ dtable, float* __restrict__ dqbias, int B, int T, int TP, int D, int H, float scale, int groups, int nbz) {
  using G = WinGeo<WW>;
  constexpr int CT = G::CT, IMG = CT * 128, CKB = CT / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NB = (2 * Wh - 1) * G::P, NBP = (NB + 3) & ~3;
  float* R = reinterpret_cast<float*>(smem);
  float* Cq = R + NBP;
  int* binsR = reinterpret_cast<int*>(Cq + G::CQ);           // fixed-point buckets: image of [R | Cq]
  float* qsum = reinterpret_cast<float*>(binsR + NBP + G::CQ);   // [8 waves][64]: q_bias gradient, a private row per wave
  char* imgs = reinterpret_cast<char*>(qsum + 8 * HD);
  const WinWg wg_ = win_wg(groups, H, nbz);
  if (!wg_.live) return;
  const int h = wg_.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const LaneOffs lo = lane_offs(lane);
  win_setup<WW>(R, Cq, table, nrd, H, h, Wh, 1.0f, true, nrd - 3);
  for (int i = threadIdx.x; i < NBP + G::CQ + 8 * HD; i += blockDim.x) binsR[i] = 0;     // buckets, qsum
  const unsigned sel_lo = sel_lo_reg();
  const unsigned bins_delta = (unsigned)(NBP + G::CQ) * 4u;
  const int qb = wg_.group * 8 + wave;
  const bool active = qb * 32 < T;
  const int q = qb * 32 + r;
  const int qc = q < T ? q : T - 1;
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
  const float bcls = table[(long long)(q == 0 ? nrd - 1 : nrd - 2) * H + h];      // bias towards the cls key
  float fx = 0.f;
  if (DT) {
    const float bound = sqrtf(stats[h * 4 + 0]) * sqrtf(stats[h * 4 + 2]) + stats[h * 4 + 1];
    fx = bound > 0.f ? 262144.0f / bound : 0.f;
  }
  float gcls = 0.f;                                          // gradient of the cls-key bucket of this lane's query
  const float qmask = q < T ? 1.f : 0.f;
  const bool qpadw = __builtin_amdgcn_readfirstlane(qb) * 32 + 32 > T;            // this wave holds queries >= T
  const int nch = (Wh + G::RPC - 1) / G::RPC;
  for (int b = wg_.bz; b < B; b += nbz) {
    const long long row = (long long)b * T + qc;
    const __bf16* s0 = qkv + (long long)b * T * ldq + h * HD;
    bf16x8 Qf[4], dOf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Qf[t] = ld16(qkv + row * ldq + h * HD + 16 * t + 8 * hh);
      dOf[t] = ld16(dout + row * ldo + h * HD + 16 * t + 8 * hh);
    }
    const float nlq = -lse[((long long)b * H + h) * TP + qc] * kLog2e;
    const float ndq = -delta[row * H + h];
    __syncthreads();                       // previous sample's last chunk fully consumed (and the setup done)
    stage_chunk_win<WW>(imgs, s0 + D, ldq, 0, Wh);
    stage_chunk_win<WW>(imgs + IMG, s0 + 2 * D, ldq, 0, Wh);
    f32x16 dQt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) dQt[db][i] = 0.f;
    for (int c = 0; c < nch; ++c) {
      const int cur = c & 1;
      const char* Ks = imgs + cur * 2 * IMG;
      const char* Vs = Ks + IMG;
      ATTN_DMA_WAIT();
      __syncthreads();
      if (c + 1 < nch) {
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG, s0 + D, ldq, c + 1, Wh);
        stage_chunk_win<WW>(imgs + (cur ^ 1) * 2 * IMG + IMG, s0 + 2 * D, ldq, c + 1, Wh);
      }
      if (!active) continue;
      const unsigned base = base0 + (unsigned)c * cstep;
      const int rows_left = Wh - c * G::RPC;
      const ColAddr ka = col_addr(Ks, lo);
      auto block = [&](auto KB, auto RAGGED) {
        constexpr int kb = decltype(KB)::value;
        f32x16 St, dPt;
#pragma unroll
        for (int i = 0; i < 16; ++i) { St[i] = 0.f; dPt[i] = 0.f; }
        // every LDS read of the block is issued BEFORE its first bucket atomic: LDS operations return in order, so a bias read
        // (or a K column fragment of the dQ product) behind the four atomics of the group before it waits for all of them
        float bzs[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int s0i = kb * 32 + 8 * g;
#pragma unroll
          for (int e = 0; e < 4; ++e) bzs[g][e] = 0.f;
          if (G::valid(s0i) || G::valid(s0i + 4)) {
            const auto* p = reinterpret_cast<const __attribute__((address_space(3))) F2u*>(base + 4u * (unsigned)G::imm(s0i));
            bzs[g][0] = p[0].a; bzs[g][1] = p[0].b; bzs[g][2] = p[1].a; bzs[g][3] = p[1].b;
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          St = MFMA32(row_frag_o(Ks, lo, kb, t), Qf[t], St);
          dPt = MFMA32(row_frag_o(Vs, lo, kb, t), dOf[t], dPt);
        }
        bf16x8 ckf[2][2];
        if (WIN_Q_CKF_EARLY && DT) {
#pragma unroll
          for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int db = 0; db < 2; ++db) ckf[ss][db] = col_frag_i<kb * 4096>(ka, ss, db);
          LDS_TR_WAIT();
#pragma unroll
          for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int db = 0; db < 2; ++db) asm volatile("" : "+v"(ckf[ss][db]));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int s0i = kb * 32 + 8 * g;                   // slot of (hh = 0, e = 0); hh = 1: + 4
          const bool v0 = G::valid(s0i), v1 = G::valid(s0i + 4);
          const bool clsg = kb == G::CLS_KB && g == G::CLS_G;           // the group that holds the cls key's slot (chunk 0)
          if (!v0 && !v1 && !clsg) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dPt[4 * g + e] = 0.f;
            continue;
          }
          float bz[4] = {bzs[g][0], bzs[g][1], bzs[g][2], bzs[g][3]};
          const unsigned a = base + 4u * (unsigned)G::imm(s0i);
          if (clsg) bz[0] = bcls;
          const unsigned s01 = pk_bf16(St[4 * g], St[4 * g + 1]), s23 = pk_bf16(St[4 * g + 2], St[4 * g + 3]);
          const unsigned d01 = pk_bf16(dPt[4 * g], dPt[4 * g + 1]), d23 = pk_bf16(dPt[4 * g + 2], dPt[4 * g + 3]);
          float sv[4], dq[4];
          sv[0] = add_lo(s01, bz[0], sel_lo); sv[1] = add_hi(s01, bz[1]); sv[2] = add_lo(s23, bz[2], sel_lo); sv[3] = add_hi(s23, bz[3]);
          dq[0] = add_lo(d01, ndq, sel_lo); dq[1] = add_hi(d01, ndq); dq[2] = add_lo(d23, ndq, sel_lo); dq[3] = add_hi(d23, ndq);
          const bool rowdead = decltype(RAGGED)::value && G::row(s0i) >= rows_left;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float pe = fexp2(fmaf(sv[e], kLog2e, nlq));
            if (qpadw) pe *= qmask;                          // (wave-uniform: only the wave that holds padding queries)
            bool live = hh ? v1 : v0;                        // compile-time per half unless the two halves differ
            if (clsg) live = (e == 0) && (c == 0) && (hh == 0);
            if (rowdead) live = false;
            const float ds = live ? pe * dq[e] : 0.f;
            dPt[4 * g + e] = ds;
            if (DT) {
              if (clsg) {
                if (e == 0) gcls += ds;
              } else if (v0 || v1) {
                __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) int*>(a + bins_delta + 4u * e),
                                       fx_round(ds, fx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
        if (!(WIN_Q_CKF_EARLY && DT)) {
#pragma unroll
          for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int db = 0; db < 2; ++db) ckf[ss][db] = col_frag_i<kb * 4096>(ka, ss, db);
        }
        bf16x8 dsf[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) dsf[ss] = acc_frag(dPt, ss, 1.0f);
        if (!(WIN_Q_CKF_EARLY && DT)) LDS_TR_WAIT();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int db = 0; db < 2; ++db) dQt[db] = MFMA32(ckf[ss][db], dsf[ss], dQt[db]);
      };
      auto chunk = [&](auto RAGGED) {
        block(std::integral_constant<int, 0>{}, RAGGED);
        block(std::integral_constant<int, 1>{}, RAGGED);
        block(std::integral_constant<int, 2>{}, RAGGED);
        block(std::integral_constant<int, 3>{}, RAGGED);
      };
      static_assert(CKB == 4, "four 32-slot blocks per chunk");
      if (rows_left < G::RPC) chunk(std::true_type{}); else chunk(std::false_type{});
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
          if (dqbias)                                                                   // q_bias gradient
            colsum_add4(qsum + wave * HD, r, hh, db, g, (float)w[0] * qmask, (float)w[1] * qmask, (float)w[2] * qmask, (float)w[3] * qmask);
        }
    }
  }
  __syncthreads();
  if (DT) {
    const float inv = fx > 0.f ? 1.0f / fx : 0.f;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) {
      const int v = binsR[i];
      if (v != 0) atomicAdd(dtable + (long long)(NB - 1 - i) * H + h, (float)v * inv);
    }
    if (wave == 0) {                                         // the cls ROW: the strip behind the buckets
      float v = 0.f;
      for (int i = lane; i < G::CQ; i += 64) v += (float)binsR[NBP + i] * inv;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && v != 0.f) atomicAdd(dtable + (long long)(nrd - 3) * H + h, v);
    }
    // the cls COLUMN: lane sums (the cls query's own term is the (cls, cls) bucket)
    if (active) {
      float colv = (q != 0 && q < T) ? gcls : 0.f;
      const float both = q == 0 ? gcls : 0.f;
      colv += __shfl_xor(colv, 32);                         // (only hh = 0 lanes hold terms)
      for (int o = 16; o > 0; o >>= 1) colv += __shfl_xor(colv, o);
      if (lane == 0 && colv != 0.f) atomicAdd(dtable + (long long)(nrd - 2) * H + h, colv);
      if (q == 0 && hh == 0 && both != 0.f) atomicAdd(dtable + (long long)(nrd - 1) * H + h, both);
    }
  }
  if (dqbias && threadIdx.x < HD) {                          // (behind the barrier above)
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += qsum[w * HD + threadIdx.x];
    atomicAdd(dqbias + h * HD + threadIdx.x, v);
  }
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
  const int grid = 8 * ((heads * nbz + 7) / 8) * groups;
  hipLaunchKernelGGL((attn_fwd_win_kernel<WW>), dim3(grid), dim3(512), sm, s, (const __bf16*)qkv, (long long)ldqkv,
                     B, T, TP, D, heads, table, nrd, Wh, (__bf16*)out, (long long)ldo, lse, groups, nbz);
  return check_launch("attn_fwd(win)");
}

template <int WW>
int launch_bwd(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse, float* delta, float* stats,
               const float* table, int Wh, int B, int T, int D, int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable,
               float* dq_bias, float* dv_bias, void* ws, int64_t ws_bytes, hipStream_t s) {
  using G = WinGeo<WW>;
  const int TP = ((T + 31) / 32) * 32;
  const int nrd = (2 * Wh - 1) * (2 * WW - 1) + 3;
  const int NBP = ((2 * Wh - 1) * G::P + 3) & ~3;
  const int groups = (TP / 32 + 7) / 8;
  int nbz = B;
  const long long per = (long long)groups * heads;
  const int cus = usable_cus(s);
  while (nbz > 1 && per * nbz > 6LL * cus) nbz = (nbz + 1) / 2;
  // the kernel that owns the table gradient is persistent over at most 16 samples per workgroup (the fixed-point bound of the buckets)
  int nbq = nbz;
  while ((B + nbq - 1) / nbq > 16) ++nbq;
  // ---- dS-storing form: needs the caller's workspace (memhip_attn_bwd_workspace)
  const int QS = G::CT * ((Wh + G::RPC - 1) / G::RPC);
  const int64_t need = (int64_t)B * heads * TP * QS * 2;
  if (opt(OPT_ATTN_WIN) == 1 && ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0) {
    const size_t sm_kvs = (size_t)(2 * (NBP + 2 * G::CQ) + 16 + 4 * G::CT + 8 * HD) * 4 + (size_t)4 * G::CT * 128;
    const size_t sm_kvs0 = (size_t)((NBP + 2 * G::CQ) + 16 + 4 * G::CT + 8 * HD) * 4 + (size_t)4 * G::CT * 128;
    const size_t sm_qs = (size_t)8 * HD * 4 + (size_t)3 * 5 * 64 * 128;
    if (sm_kvs <= (size_t)kMaxLds) {
      static bool a0 = false, a1 = false, a2 = false, a3 = false, a4 = false;
      if (int rc = set_lds_attr(attn_bwd_kvs_win_kernel<WW, true, true>, &a0)) return rc;
      if (int rc = set_lds_attr(attn_bwd_kvs_win_kernel<WW, false, true>, &a1)) return rc;
      if (int rc = set_lds_attr(attn_bwd_kvs_win_kernel<WW, true, false>, &a2)) return rc;
      if (int rc = set_lds_attr(attn_bwd_kvs_win_kernel<WW, false, false>, &a3)) return rc;
      if (int rc = set_lds_attr(attn_bwd_qs_win_kernel<WW>, &a4)) return rc;
      if (dtable) {            // max |dO_q|^2, max |delta_q| per head (the caller has zeroed stats)
        const int sg = heads * ((256 + heads - 1) / heads);       // a multiple of the head count (see the kernel), ~256 blocks
        hipLaunchKernelGGL(attn_win_stats_kernel, dim3(sg), dim3(256), 0, s, (const float*)delta, (long long)B * T, heads, stats);
      }
      const int nb = dtable ? nbq : nbz;
      const dim3 gk(8 * ((heads * nb + 7) / 8) * groups);
#define KVS_LAUNCH(VBF, DTF, SM)                                                                                              \
      hipLaunchKernelGGL((attn_bwd_kvs_win_kernel<WW, VBF, DTF>), gk, dim3(512), SM, s, (const __bf16*)qkv, (long long)ldqkv,   \
                         (const __bf16*)dout, (long long)ldo, lse, (const float*)delta, (const float*)stats, table, nrd, Wh,   \
                         (__bf16*)dqkv, (long long)lddqkv, dv_bias, dtable, (__bf16*)ws, QS, B, T, TP, D, heads, groups, nb)
      if (dtable) { if (dv_bias) KVS_LAUNCH(true, true, sm_kvs); else KVS_LAUNCH(false, true, sm_kvs); }
      else { if (dv_bias) KVS_LAUNCH(true, false, sm_kvs0); else KVS_LAUNCH(false, false, sm_kvs0); }
#undef KVS_LAUNCH
      const int qgroups = (QS + 255) / 256;
      int nbs = B;
      while (nbs > 1 && (long long)qgroups * heads * nbs > 6LL * cus) nbs = (nbs + 1) / 2;
      const dim3 gq2(8 * ((heads * nbs + 7) / 8) * qgroups);
      hipLaunchKernelGGL((attn_bwd_qs_win_kernel<WW>), gq2, dim3(512), sm_qs, s, (const __bf16*)qkv, (long long)ldqkv,
                         (const __bf16*)ws, QS, Wh, (__bf16*)dqkv, (long long)lddqkv, dq_bias, B, T, TP, D, heads, scale, qgroups, nbs);
      return check_launch("attn_bwd(win, dS-storing)");
    }
  }
  // ---- recomputing form (no workspace)
  const size_t sm_kv = (size_t)(NBP + G::CQ + 4 * G::CT + 8 * HD) * 4 + (size_t)4 * G::CT * 128;
  const size_t sm_q = (size_t)(2 * (NBP + G::CQ) + 8 * HD) * 4 + (size_t)4 * G::CT * 128;
  if (sm_kv > (size_t)kMaxLds || sm_q > (size_t)kMaxLds) return MEMHIP_EUNSUPPORTED;
  static bool d0 = false, d1 = false, d2 = false, d3 = false;
  if (int rc = set_lds_attr(attn_bwd_kv_win_kernel<WW, true>, &d0)) return rc;
  if (int rc = set_lds_attr(attn_bwd_kv_win_kernel<WW, false>, &d1)) return rc;
  if (int rc = set_lds_attr(attn_bwd_q_win_kernel<WW, true>, &d2)) return rc;
  if (int rc = set_lds_attr(attn_bwd_q_win_kernel<WW, false>, &d3)) return rc;
  const dim3 grid(8 * ((heads * nbz + 7) / 8) * groups);
  if (dv_bias)
    hipLaunchKernelGGL((attn_bwd_kv_win_kernel<WW, true>), grid, dim3(512), sm_kv, s, (const __bf16*)qkv, (long long)ldqkv,
                       (const __bf16*)dout, (long long)ldo, lse, delta, dtable ? stats : (float*)nullptr, table, nrd, Wh,
                       (__bf16*)dqkv, (long long)lddqkv, dv_bias, B, T, TP, D, heads, groups, nbz);
  else
    hipLaunchKernelGGL((attn_bwd_kv_win_kernel<WW, false>), grid, dim3(512), sm_kv, s, (const __bf16*)qkv, (long long)ldqkv,
                       (const __bf16*)dout, (long long)ldo, lse, delta, dtable ? stats : (float*)nullptr, table, nrd, Wh,
                       (__bf16*)dqkv, (long long)lddqkv, dv_bias, B, T, TP, D, heads, groups, nbz);
  const dim3 gq(8 * ((heads * nbq + 7) / 8) * groups);
  if (dtable)
    hipLaunchKernelGGL((attn_bwd_q_win_kernel<WW, true>), gq, dim3(512), sm_q, s, (const __bf16*)qkv, (long long)ldqkv,
                       (const __bf16*)dout, (long long)ldo, lse, delta, stats, table, nrd, Wh, (__bf16*)dqkv, (long long)lddqkv,
                       dtable, dq_bias, B, T, TP, D, heads, scale, groups, nbq);
  else
    hipLaunchKernelGGL((attn_bwd_q_win_kernel<WW, false>), gq, dim3(512), sm_q, s, (const __bf16*)qkv, (long long)ldqkv,
                       (const __bf16*)dout, (long long)ldo, lse, delta, stats, table, nrd, Wh, (__bf16*)dqkv, (long long)lddqkv,
                       dtable, dq_bias, B, T, TP, D, heads, scale, groups, nbq);
  return check_launch("attn_bwd(win)");
}

template <int WW>
int64_t win_ws_bytes(int B, int T, int heads, int Wh) {
  using G = WinGeo<WW>;
  const int TP = ((T + 31) / 32) * 32;
  return (int64_t)B * heads * TP * (G::CT * ((Wh + G::RPC - 1) / G::RPC)) * 2;
}

}  // namespace

#ifdef WIN_STAMP

extern "C" int memhip_debug_win_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_win_stamps), sizeof(unsigned long long) * 1024 * 8) == hipSuccess ? 0 : -1;
}
#endif

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


int attn_bwd_win(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse, float* delta, float* stats,
                 const float* table, int window_h, int window_w, int B, int T, int D, int heads, float scale, void* dqkv,
                 int64_t lddqkv, float* dtable, float* dq_bias, float* dv_bias, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (window_w == 40)
    return launch_bwd<40>(qkv, ldqkv, dout, ldo, lse, delta, stats, table, window_h, B, T, D, heads, scale, dqkv, lddqkv, dtable,
                          dq_bias, dv_bias, ws, ws_bytes, s);
  if (window_w == 20)
    return launch_bwd<20>(qkv, ldqkv, dout, ldo, lse, delta, stats, table, window_h, B, T, D, heads, scale, dqkv, lddqkv, dtable,
                          dq_bias, dv_bias, ws, ws_bytes, s);
  return MEMHIP_EUNSUPPORTED;
}

// bytes of the dS workspace the dS-storing backward wants (0: no such form for this window)
int64_t attn_bwd_win_workspace(int B, int T, int heads, int window_h, int window_w) {
  if (!attn_win_fits(T, window_h, window_w)) return 0;
  if (window_w == 40) return win_ws_bytes<40>(B, T, heads, window_h);
  if (window_w == 20) return win_ws_bytes<20>(B, T, heads, window_h);
  return 0;
}

}  // namespace memhip
