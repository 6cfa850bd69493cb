// 256x128-tile form of the phase-interleaved persistent bf16 NT GEMM (gemm_p8.hip) with a STREAMED epilogue: the finished
// tile's accumulators are parked in 64 registers and leave -- bias, GELU, conversion, store -- one output row per phase
// inside the LOAD segments of the next tile's main loop, so the matrix pipe never waits for an epilogue and the stores
// never arrive as one burst.  (gemm_p8.hip: all eight waves stop for the epilogue, 26 % of a K = 768 bias tile and
// more with GELU; a second accumulator set does not fit beside 128 accumulators, it does beside 64.)
//
//   * a K-tile (64 deep) of the 256x128 tile is two 128x128 quadrants (A0, B0), (A1, B0) = two PHASES of 16 MFMAs per
//     wave; waves 4-7 run half a phase behind waves 0-3 as in gemm_p8 (one group computes while the other loads);
//   * three operand stages {A0, A1, B0} of 48 KiB; per phase and wave: phase 1 reads A0 (8 fragments) + B0 (4) and
//     issues the LDS-DMA of A1(c + 2); phase 2 reads A1 (8) and issues B0(c + 3), A0(c + 3): every half-tile is in flight
//     for five phases; the counted wait before a phase's first barrier is vmcnt(12) in both phases (the DMA
//     instructions issued behind the half-tile the next phase reads).  Epilogue stores and the per-tile bias DMA are extra
//     entries of the in-order queue: they only make the counted wait stricter, never unsafe;
//   * a slot is restaged ONE phase after the phase that read it: the fragment reads are retired (lgkmcnt(0)) BEFORE the
//     phase's first barrier, so no read is in flight when the other wave group issues the DMA in the next interval;
//   * parked rows: a lane holds 8 rows x 8 columns of the tile (q = A half, mf = 16-row fragment); row r = 4 q + mf leaves
//     in phase r of the next tile (K >= 256: 8 phases), the last tile of a workgroup is flushed behind the loop.
// Same contract as gemm_p8 for the epilogues it takes (bias -> bf16, bias + GELU); no column sums, no row guard
// (M % 256 == 0), N % 128 == 0, K % 64 == 0, K >= 256.
//
// RESULT (round 3, tools/p8s_check.py, M = 50 432): bit-equal to gemm_p8 on every shape, and SLOWER: N = 3072 K = 768 bias
// 284 vs 263 us, GELU 365 vs 329; N = 768 K = 3072 bias 264 vs 209 us -- the main loop takes 1 650 ticks per K-tile of
// 2 M MACs against 2 380 per 4 M MACs.  Common to both kernels: 27-29 bytes per clock and CU through global_load_lds (48 KiB
// per 1 650 clocks, 64 KiB per 2 380) -- the main loop runs at the LDS-DMA operand feed, and a 256x128 tile needs 1.5 x the
// operand bytes per flop (matrix pipe 62 % vs 87 % busy).  The streamed epilogue works (no stall at tile boundaries) but
// cannot win that back; only K = 256 shapes gain (35.6 vs 42.6 us).  Kept as the measured form of the "second accumulator
// set" item, OFF by default (option `gemm_p8s`); epilogue overlap has to keep the 256x256 tile's bytes per flop.
#include "common.h"
#include "gemm_epilogue.hpp"
#include <type_traits>

namespace {

using namespace memhip;

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kThreads = 512;
constexpr int kHalf = 128 * BK * 2;               // 16 KiB: 128 rows x 64 k
constexpr int kBOff = 2 * kHalf;                  // B0 behind A0, A1
constexpr int kBuf = 3 * kHalf;                   // 48 KiB
constexpr int kStages = 3;
constexpr int kLds = kStages * kBuf;              // 144 KiB
constexpr int kColsSlot = 1024;                   // 128 f32 bias values (read as 256: the upper half is a copy)
constexpr int kColsOff = kLds;
constexpr int kLdsAll = kLds + 2 * kColsSlot;
constexpr int MF = 4;                             // 16-row fragments per wave and A half
constexpr int kGroupM = 8;
constexpr int kWait = 12;                         // see the header

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ int key_a(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int key_b(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

#define P8S_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define P8S_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define P8S_BARRIER()                     \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

template <int EPI>
__global__ __launch_bounds__(kThreads) void gemm_p8s_kernel(GemmArgs p, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = p.K / BK;
  const int ntiles = ntm * ntn;
  const int per_xcd = (gridDim.x + 7) / 8;
  const int first = (gridDim.x % 8 == 0) ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;                      // K-tiles of this workgroup's stream
  if (total <= 0) return;

  // ---- LDS-DMA issue constants (gemm_p8.hip): a 128-row half-tile is 16 pieces of 8 rows x 128 B, this wave moves pieces
  // 2 wave and 2 wave + 1
  int prow[2];
  unsigned offA[2], offB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    prow[j] = (wave * 2 + j) * 8 + (lane >> 3);
    offB[j] = (unsigned)((long long)prow[j] * p.ldb * 2) + (unsigned)(((lane & 7) ^ key_b(prow[j])) * 16);
    offA[j] = (unsigned)((long long)prow[j] * p.lda * 2) + (unsigned)(((lane & 7) ^ key_a(prow[j])) * 16);
  }
  auto decode = [&](int id, int& tm, int& tn) {
    const int gsz = kGroupM * ntn;
    const int grp = id / gsz, rem = id - grp * gsz;
    const int rows = ntm - grp * kGroupM < kGroupM ? ntm - grp * kGroupM : kGroupM;
    tn = rem / rows;
    tm = grp * kGroupM + (rem - tn * rows);
  };
  // half-tile H (0 = A0, 1 = A1, 2 = B0) of K-tile kt of tile (tm, tn) into stage st
  auto stage = [&](int H, int st, int tm, int tn, int kt) {
    char* slot = smem + st * kBuf + H * kHalf + wave * 2048;
    if (H < 2) {
      const char* base = reinterpret_cast<const char*>(p.A) + ((long long)(tm * BM + H * 128) * p.lda + kt * BK) * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(base + offA[j], slot + j * 1024);
    } else {
      const char* base = reinterpret_cast<const char*>(p.B) + ((long long)(tn * BN) * p.ldb + kt * BK) * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(base + offB[j], slot + j * 1024);
    }
  };
  // stream cursors: K-tile g of the stream = (tile id, k); clamped at the end of the stream
  struct Cur { int g, id, k, tm, tn, st; };
  auto cur_init = [&](Cur& c) { c.g = 0; c.id = first; c.k = 0; c.st = 0; decode(c.id, c.tm, c.tn); };
  auto cur_next = [&](Cur& c) {
    c.st = c.st == kStages - 1 ? 0 : c.st + 1;            // (the stage always advances: a clamped cursor restages the last
    if (c.g + 1 < total) {                                //  K-tile into slots nobody reads any more)
      ++c.g;
      if (++c.k == nk) { c.k = 0; c.id += gridDim.x; decode(c.id, c.tm, c.tn); }
    }
  };
  Cur c2, c3;                                              // K-tiles c + 2 and c + 3 of the stream
  cur_init(c2);
  // prologue: K-tiles 0, 1 entirely, B0 and A0 of K-tile 2 (what the steady state has issued before phase 1 of K-tile 0)
  stage(2, c2.st, c2.tm, c2.tn, c2.k); stage(0, c2.st, c2.tm, c2.tn, c2.k); stage(1, c2.st, c2.tm, c2.tn, c2.k);
  cur_next(c2);
  stage(2, c2.st, c2.tm, c2.tn, c2.k); stage(0, c2.st, c2.tm, c2.tn, c2.k); stage(1, c2.st, c2.tm, c2.tn, c2.k);
  cur_next(c2);
  stage(2, c2.st, c2.tm, c2.tn, c2.k); stage(0, c2.st, c2.tm, c2.tn, c2.k);
  c3 = c2;
  cur_next(c3);
  P8S_WAIT_VM(10);                                         // K-tile 0 has landed: B0 A0 A1 of K-tile 1, B0 A0 of 2 may fly
  P8S_BARRIER();
  if (wr == 1) P8S_BARRIER();                              // waves 4-7 run half a phase behind

  // ---- fragment read addresses (gemm_p8.hip): row = 16 x + (lane & 15), chunk = 4 kh + (lane >> 4)
  const int sw = (lane >> 1) & 7;
  const int roff0 = (lane & 15) * 128 + ((((lane >> 4)) ^ sw) << 4);
  const unsigned ldsb = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem);
  const unsigned rdA[2] = {ldsb + wr * (MF * 2048) + roff0, ldsb + wr * (MF * 2048) + (roff0 ^ 64)};
  const int bi = lane & 15;
  const int roffb = (((bi >> 2) * 8 + (bi & 3)) * 128) + (((lane >> 4) ^ key_b((bi >> 2) * 8 + (bi & 3))) << 4);
  const unsigned rdB[2] = {ldsb + kBOff + wc * 4096 + roffb, ldsb + kBOff + wc * 4096 + (roffb ^ 64)};
  auto lds128 = [&](unsigned addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(addr);
  };

  f32x4 acc[2][MF][2], pk[2][MF][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) { acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; pk[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  bf16x8 a[MF][2], b[2][2];

  // ---- tile state
  int c_tile = first, c_k = 0, st = 0;                     // current K-tile: tile id, k index, stage
  int ctm, ctn;
  decode(c_tile, ctm, ctn);
  int tile_par = 0;
  bool have_pk = false;                                    // a parked tile is waiting to leave
  int ptm = 0, ptn = 0, ppar = 0;                          // its coordinates and the parity of its bias slot
  EpiCols pcols;                                           // its per-column operands (read from LDS with its first row)
  float cs_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  // the tile's 128 bias values into the tile parity's LDS slot (one 16-byte-per-lane DMA by wave 0; the upper 32 lanes
  // bring a second copy: a 256-value read would pass the end of the bias vector at the last column tile)
  auto bias_dma = [&]() {
    if (wave == 0) {
      int dl;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(dl));
      const char* src = (p.bias ? reinterpret_cast<const char*>(p.bias + ctn * BN) : reinterpret_cast<const char*>(g_epi_zero256)) +
                        (dl & 31) * 16;
      glds16(src, smem + kColsOff + tile_par * kColsSlot);
    }
  };
  auto cols_from_lds = [&](int par, int ncl, EpiCols& c) {
    const unsigned addr = ldsb + (unsigned)(kColsOff + par * kColsSlot + ncl * 4);
    f32x4 q0, q1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(addr) : "memory");
    c.bias[0] = ef32x2{q0[0], q0[1]}; c.bias[1] = ef32x2{q0[2], q0[3]};
    c.bias[2] = ef32x2{q1[0], q1[1]}; c.bias[3] = ef32x2{q1[2], q1[3]};
  };
  // row r = 4 q + mf of the parked tile: epilogue + store (16 bytes per lane and output)
  auto park_row = [&](auto R) {
    constexpr int r = decltype(R)::value, q = r >> 2, mf = r & 3;
    int el;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
    const int m = ptm * BM + q * 128 + wr * (MF * 16) + mf * 16 + (el & 15);
    const int ncl = wc * 32 + (el >> 4) * 8;
    if (r == 0) cols_from_lds(ppar, ncl, pcols);
    float v[8];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int c = 0; c < 4; ++c) v[nf * 4 + c] = pk[q][mf][nf][c];
    EpiRow<EPI> row;
    epilogue8<EPI, 2>(p, m, ptn * BN + ncl, v, cs_dummy, pcols, row);
  };

#define P8S_READ_A(half)                                                                                  \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)      \
      a[mf][kh] = lds128(rdA[kh] + bo + (half) * kHalf + mf * 2048)
#define P8S_READ_B()                                                                                      \
  _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)       \
      b[nf][kh] = lds128(rdB[kh] + bo + nf * 512)
#define P8S_MFMA(q)                                                                                       \
  do {                                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)     \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                                  \
            acc[q][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf][kh], a[mf][kh], acc[q][mf][nf], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                        \
  } while (0)
// one K-tile; R0 / R1: parked rows that leave in its two load segments (-1: none)
#define P8S_KTILE(R0, R1)                                                                                 \
  do {                                                                                                    \
    const unsigned bo = (unsigned)(st * kBuf);                                                            \
    if (c_k == 0) bias_dma();                                                                             \
    /* phase 1: quadrant (A0, B0) */                                                                      \
    P8S_READ_A(0);                                                                                        \
    P8S_READ_B();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    stage(1, c2.st, c2.tm, c2.tn, c2.k);                                                                  \
    if constexpr ((R0) >= 0) { if (have_pk) park_row(std::integral_constant<int, ((R0) >= 0 ? (R0) : 0)>{}); } \
    P8S_WAIT_VM(kWait);                                                                                   \
    P8S_WAIT_LDS();                                                                                       \
    P8S_BARRIER();                                                                                        \
    P8S_MFMA(0);                                                                                          \
    P8S_BARRIER();                                                                                        \
    /* phase 2: quadrant (A1, B0) */                                                                      \
    P8S_READ_A(1);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    stage(2, c3.st, c3.tm, c3.tn, c3.k);                                                                  \
    stage(0, c3.st, c3.tm, c3.tn, c3.k);                                                                  \
    if constexpr ((R1) >= 0) { if (have_pk) park_row(std::integral_constant<int, ((R1) >= 0 ? (R1) : 0)>{}); } \
    P8S_WAIT_VM(kWait);                                                                                   \
    P8S_WAIT_LDS();                                                                                       \
    P8S_BARRIER();                                                                                        \
    P8S_MFMA(1);                                                                                          \
    P8S_BARRIER();                                                                                        \
    cur_next(c2);                                                                                         \
    cur_next(c3);                                                                                         \
    st = st == kStages - 1 ? 0 : st + 1;                                                                  \
    ++c_k;                                                                                                \
  } while (0)

  for (int t = 0; t < my_tiles; ++t) {
    P8S_KTILE(0, 1);
    P8S_KTILE(2, 3);
    P8S_KTILE(4, 5);
    P8S_KTILE(6, 7);
    for (int k = 4; k < nk; ++k) P8S_KTILE(-1, -1);
    // the tile is complete: park it (its rows leave during the next tile), start the next one from zero
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { pk[q][i][j] = acc[q][i][j]; acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    have_pk = true;
    ptm = ctm; ptn = ctn; ppar = tile_par;
    c_k = 0;
    c_tile += gridDim.x;
    if (c_tile < ntiles) decode(c_tile, ctm, ctn);
    tile_par ^= 1;
  }
  // the last tile of this workgroup leaves here
  park_row(std::integral_constant<int, 0>{}); park_row(std::integral_constant<int, 1>{});
  park_row(std::integral_constant<int, 2>{}); park_row(std::integral_constant<int, 3>{});
  park_row(std::integral_constant<int, 4>{}); park_row(std::integral_constant<int, 5>{});
  park_row(std::integral_constant<int, 6>{}); park_row(std::integral_constant<int, 7>{});
  if (wr == 0) P8S_BARRIER();                              // balances the stagger barrier of waves 4-7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup
}

template <int EPI>
int launch_p8s(const GemmArgs& p, hipStream_t s, int num_cu) {
  const int ntm = p.M / BM, ntn = p.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8s_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAll);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_p8s: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int grid = ntm * ntn < num_cu ? ntm * ntn : num_cu;
  hipLaunchKernelGGL((gemm_p8s_kernel<EPI>), dim3(grid), dim3(kThreads), kLdsAll, s, p, ntm, ntn);
  return check_launch("gemm_bf16_nt(p8s)");
}

}  // namespace

namespace memhip {

// Returns MEMHIP_EUNSUPPORTED when the shape / epilogue is not this kernel's (caller falls back to gemm_p8).
int gemm_p8s_dispatch(const GemmArgs& p, hipStream_t s) {
  const bool vec = ((p.ldo0 | p.ldo1) & 7) == 0;
  if (p.M < 4096 || p.M % BM != 0 || p.N % BN != 0 || p.K % BK != 0 || p.K < 4 * BK || !vec) return MEMHIP_EUNSUPPORTED;
  if (p.colsum) return MEMHIP_EUNSUPPORTED;
  const int num_cu = usable_cus(s);
  if (!num_cu) return MEMHIP_EUNSUPPORTED;
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch_p8s<MEMHIP_EPI_BIAS_BF16>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch_p8s<MEMHIP_EPI_BIAS_GELU>(p, s, num_cu);
    default: return MEMHIP_EUNSUPPORTED;
  }
}

}  // namespace memhip
