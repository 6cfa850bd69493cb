// gemm_p8 (256x256x64 tiles, phase-interleaved persistent bf16 NT GEMM: see gemm_p8.hip for the main loop) with the
// tile's STORES deferred into the next tile's first K-tile.
//
// What gemm_p8 pays at every tile boundary (tools/p8_stamps*.py, round 3): the epilogue's own cycles, and ~5 200 more in
// the K-tiles behind it -- every workgroup of the chip stores its 128-256 KB at the same moment, and the LDS-DMA requests
// of the next tile queue behind those stores in the CU's memory pipeline.  A second accumulator set does not fit (128
// accumulators + 64 fragment registers + the stream's lane state = the whole 256-register budget), and 160 KB of LDS
// cannot park a tile beside 128 KB of operand stages.  But the registers ARE free when they are needed:
//
//   * quadrant q of the tile (128x128; phase q of every K-tile) is final after phase q of the LAST K-tile and is not
//     written again before phase q of the next tile's FIRST K-tile, whose MFMAs start from C = 0;
//   * so the epilogue only does the ARITHMETIC (epi8_math: bias / GELU / GELU' x aux / residual add, every load it needs),
//     all eight waves together as before (VALU at full rate), and leaves the finished values IN PLACE of the accumulators
//     (bf16 pairs or fp32: never more than the 128 registers they came from);
//   * the stores go out in the load segments of the next tile's first K-tile: quadrant q right before phase q computes
//     into its registers again -- 4 to 8 store instructions per wave and phase between the fragment reads and the LDS-DMA
//     issue, 128-256 KB per workgroup spread over ~2 400 cycles of MFMA work instead of one burst.  The first MFMA of an
//     accumulator in that K-tile takes the constant 0 as C, so no register is re-zeroed either;
//   * the counted waits of the eight phases behind a tile boundary are raised by exactly the stores issued since the
//     LDS-DMA they retire (table at P8D_KTILE_N / _M); the first tile of a workgroup stores to a scratch line (the counts
//     stay compile-time constants), the last tile is flushed behind the loop.
//
// Contract, operand staging, swizzles, tile order and hazards are gemm_p8.hip's (whole 256-row tiles only, K % 128 == 0).
#include "common.h"
#include "gemm_epilogue.hpp"
#include <type_traits>

namespace {

using namespace memhip;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 512;
constexpr int kHalf = 128 * BK * 2;     // 16 KiB: 128 rows x 64 k
constexpr int kBOff = 2 * kHalf;        // B0 behind A0, A1
constexpr int kBuf = 4 * kHalf;         // A0 A1 B0 B1
constexpr int kLds = 2 * kBuf;
constexpr int kColsSlot = 3072;         // [256 f32 bias | 256 f32 layer scale | 256 i32 sample map] per tile parity
constexpr int kColsOff = kLds, kTrashOff = kLds + 2 * kColsSlot;
constexpr int kLdsAll = kLds + 2 * kColsSlot + 8 * 1024;
constexpr int MF = 4;                   // 16-row fragments per wave and A half
constexpr int kWait = 10;               // LDS-DMA instructions of the five youngest half-tiles (gemm_p8.hip: kWaitA = kWaitB)
enum { HA0 = 0, HA1 = 1, HB0 = 2, HB1 = 3 };
constexpr int kGroupM = 8;
constexpr int kEpiAhead = 2;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
using QC0 = std::integral_constant<int, 0>; using QC1 = std::integral_constant<int, 1>;
using QC2 = std::integral_constant<int, 2>; using QC3 = std::integral_constant<int, 3>;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ int key_a(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int key_b(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

#define P8D_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define P8D_BARRIER()                     \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// deferred stores are global stores whatever the pointer's provenance (a select with the scratch line must not turn them
// into flat stores: a flat access waits for the LDS-DMA stream)
typedef eu32x4 __attribute__((address_space(1)))* gst16_ptr;
__device__ __attribute__((aligned(2048))) unsigned char g_p8d_trash[2048];
__device__ __forceinline__ void st16_nt(char* q, unsigned a, unsigned b, unsigned c, unsigned d) {
  __builtin_nontemporal_store(eu32x4{a, b, c, d}, (gst16_ptr)q);
}
__device__ __forceinline__ void st16(char* q, unsigned a, unsigned b, unsigned c, unsigned d) {
  *(gst16_ptr)q = eu32x4{a, b, c, d};
}

#ifdef P8D_STAMP
// diagnostic build (tools/build_variant.sh p8dstamp -DP8D_STAMP): waves 0 and 4 of every workgroup record s_memtime at six
// points of their first 8 tiles: 0 tile start, 1 first K-tile done, 2 second K-tile done, 3 main loop done, 4 arithmetic
// starts (behind the realignment barrier), 5 arithmetic done.  memhip_debug_p8d_stamps copies the table out.
__device__ unsigned long long g_p8d_stamps[256 * 2 * 8 * 6];
#define P8D_STAMP_AT(slot)                                                                                  \
  do {                                                                                                      \
    if ((wave & 3) == 0 && t < 8) {                                                                         \
      unsigned long long t_;                                                                                \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
      if (lane == 0) g_p8d_stamps[(((blockIdx.x & 255) * 2 + wr) * 8 + t) * 6 + (slot)] = t_;               \
    }                                                                                                       \
  } while (0)
#else
#define P8D_STAMP_AT(slot) do { } while (0)
#endif

template <int EPI, bool COPY>
__global__ __launch_bounds__(kThreads) void gemm_p8d_kernel(GemmArgs p, int ntm, int ntn, int stagger) {
  constexpr int PW = EpiPk<EPI>::W;                  // parked dwords per row piece (4: one 16-byte store, 8: two)
  constexpr int S = 4 * (PW / 4);                    // store instructions per wave and phase of the first K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = p.K / BK;
  const int ntiles = ntm * ntn;
  const int per_xcd = (gridDim.x + 7) / 8;
  const int first = (gridDim.x % 8 == 0) ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  if (total <= 0) return;
  // experiment (option gemm_stagger < 0): every workgroup starts late by a per-workgroup fraction of |stagger| cycles per
  // K-tile of a tile, so that the workgroups of the chip reach their tile boundaries at different times
  if (stagger < 0) {
    const unsigned frac = (((unsigned)blockIdx.x * 2654435761u) >> 22) & 1023u;
    const unsigned long long wait = ((unsigned long long)(-stagger) * (unsigned)nk * frac) >> 10;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
  }

  // ---- LDS-DMA issue constants (gemm_p8.hip)
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
  auto stage = [&](int H, int buf, int tm, int tn, int kt) {
    if (H == HA0 || H == HA1) {
      char* slot = smem + buf * kBuf + (H == HA1 ? kHalf : 0) + wave * 2048;
      const int r0 = tm * BM + (H == HA1 ? BM / 2 : 0);
      const char* base = reinterpret_cast<const char*>(p.A) + ((long long)r0 * p.lda + kt * BK) * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(base + offA[j], slot + j * 1024);
    } else {
      char* slot = smem + buf * kBuf + kBOff + (H == HB1 ? kHalf : 0) + wave * 2048;
      const int c0 = tn * BN + (H == HB1 ? 128 : 0);
      const char* base = reinterpret_cast<const char*>(p.B) + ((long long)c0 * p.ldb + kt * BK) * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(base + offB[j], slot + j * 1024);
    }
  };
  int g2 = 0, id2 = first, k2 = 0, tm2, tn2;            // becomes K-tile c+2
  decode(id2, tm2, tn2);
  auto advance2 = [&]() {
    if (g2 + 1 < total) {
      ++g2;
      if (++k2 == nk) { k2 = 0; id2 += gridDim.x; decode(id2, tm2, tn2); }
    }
  };
  // prologue: K-tile 0 entirely, B0 A0 B1 of K-tile 1
  stage(HB0, 0, tm2, tn2, k2); stage(HA0, 0, tm2, tn2, k2); stage(HB1, 0, tm2, tn2, k2); stage(HA1, 0, tm2, tn2, k2);
  advance2();
  int tm1 = tm2, tn1 = tn2, k1 = k2;                    // K-tile c+1
  stage(HB0, 1, tm1, tn1, k1); stage(HA0, 1, tm1, tn1, k1); stage(HB1, 1, tm1, tn1, k1);
  advance2();
  P8D_WAIT_VM(kWait);
  P8D_BARRIER();
  if (wr == 1) P8D_BARRIER();                            // waves 4-7 run half a phase behind

  // ---- fragment read addresses: row = 16*x + (lane & 15), chunk = 4*kh + (lane >> 4)
  const int sw = (lane >> 1) & 7;
  const int roff0 = (lane & 15) * 128 + ((((lane >> 4)) ^ sw) << 4);
  const char* rdA[2] = {smem + wr * (MF * 2048) + roff0, smem + wr * (MF * 2048) + (roff0 ^ 64)};
  const int bi = lane & 15;
  const int roffb = (((bi >> 2) * 8 + (bi & 3)) * 128) + (((lane >> 4) ^ key_b((bi >> 2) * 8 + (bi & 3))) << 4);
  const char* rdB[2] = {smem + kBOff + wc * 4096 + roffb, smem + kBOff + wc * 4096 + (roffb ^ 64)};

  // (no zero fill: the first K-tile of every tile starts its accumulators from C = 0, and the "parked values" the first
  // tile sends to the scratch line may be anything.  A zero SSA value shared by 256 registers is kept alive -- and
  // spilled -- across the whole loop by hipcc; an opaque register is not.)
  f32x4 acc[4][MF][2];
  unsigned pk[4][MF][PW];                                  // the previous tile's finished values (quadrant, row fragment)
  {
    unsigned any;
    asm volatile("v_mov_b32 %0, 0" : "=v"(any));
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < PW; ++j) pk[q][i][j] = any;
  }

  int c_tile = first;
  bf16x8 a[MF][2], bx[2][2], by[2][2];

#define P8D_READ_A(half)                                                                                  \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)      \
      a[mf][kh] = *reinterpret_cast<const bf16x8*>(rdA[kh] + bo + (half) * kHalf + mf * 2048)
#define P8D_READ_B(dst, boff, half)                                                                       \
  _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)       \
      dst[nf][kh] = *reinterpret_cast<const bf16x8*>(rdB[kh] + (boff) + (half) * kHalf + nf * 512)
#define P8D_MFMA_HALF(q, bsrc, kh)                                                                        \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)      \
      acc[q][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bsrc[nf][kh], a[mf][kh], acc[q][mf][nf], 0, 0, 0)
// first K-tile of an output tile: the accumulator starts from the constant 0 (its registers hold the previous tile's
// finished values until the stores of this phase's load segment have read them)
#define P8D_MFMA_HALF0(q, bsrc)                                                                           \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)      \
      acc[q][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bsrc[nf][0], a[mf][0], (f32x4{0.f, 0.f, 0.f, 0.f}), 0, 0, 0)
#define P8D_MFMA(q, bsrc, FIRST)                                                                          \
  do {                                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    if constexpr (FIRST) { P8D_MFMA_HALF0(q, bsrc); } else { P8D_MFMA_HALF(q, bsrc, 0); }                 \
    P8D_MFMA_HALF(q, bsrc, 1);                                                                            \
    __builtin_amdgcn_s_setprio(0);                                                                        \
  } while (0)
// One phase: fragment reads, [deferred stores of quadrant SQ], the phase's LDS-DMA issue, the counted wait, barrier,
// 16 MFMAs, barrier.  WAITN = vector-memory instructions issued behind the LDS-DMA this wait retires (the one the NEXT
// phase reads, issued five phases ago).
#define P8D_PHASE(READS, STORES, STAGE_CALL, WAITN, q, bsrc, FIRST)                                       \
    READS;                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    STAGE_CALL;                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    STORES;                                                                                               \
    P8D_WAIT_VM(WAITN);                                                                                   \
    P8D_BARRIER();                                                                                        \
    P8D_MFMA(q, bsrc, FIRST);                                                                             \
    P8D_BARRIER();
// One K-tile = four phases, quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0) = acc[0], acc[1], acc[3], acc[2]; B0 of this
// K-tile is already in `bq0`, phase 4 reads B0 of the next K-tile into `bq1` (gemm_p8.hip).
//   X1..X4: instructions besides the half-tile stream that the phase's wait must leave in flight as well
//   ST:     0 = plain K-tile; 1 = first K-tile of a tile: phase q stores the previous tile's quadrant and starts from C = 0
#define P8D_KTILE(bq0, bq1, X1, X2, X3, X4, ST)                                                           \
  do {                                                                                                    \
    const int bo = bc * kBuf;                                                                             \
    P8D_PHASE(P8D_READ_A(0), if constexpr (ST) store_quadrant(QC0{}),          \
              stage(HA1, bc ^ 1, tm1, tn1, k1), kWait + (X1), 0, bq0, ST)                                 \
    P8D_PHASE(P8D_READ_B(bq1, bo, 1), if constexpr (ST) store_quadrant(QC1{}), \
              stage(HB0, bc, tm2, tn2, k2), kWait + (X2), 1, bq1, ST)                                     \
    P8D_PHASE(P8D_READ_A(1), if constexpr (ST) store_quadrant(QC3{}),          \
              stage(HA0, bc, tm2, tn2, k2), kWait + (X3), 3, bq1, ST)                                     \
    P8D_PHASE(P8D_READ_B(bq1, (bc ^ 1) * kBuf, 0), if constexpr (ST) store_quadrant(QC2{}), \
              stage(HB1, bc, tm2, tn2, k2), kWait + (X4), 2, bq0, ST)                                     \
  } while (0)

  constexpr bool kHasBias = EPI == MEMHIP_EPI_BIAS_BF16 || EPI == MEMHIP_EPI_BIAS_GELU || EPI == MEMHIP_EPI_RESIDUAL ||
                            EPI == MEMHIP_EPI_BIAS_GELU_DG;
  int tile_par = 0;                                        // parity of the current output tile of this workgroup
  int ctm, ctn;                                            // coordinates of the current output tile
  decode(c_tile, ctm, ctn);
  // the parked tile: coordinates, the parity of its per-column LDS slot, whether it exists (first tile: scratch stores)
  int ptm = 0, ptn = 0, ppar = 0;
  unsigned pmask = 0;                                      // all ones once a tile is parked
  const int smask = p.sample_map ? -1 : 0;

  // One LDS-DMA per wave at the start of a tile (always issued: the counted waits are compile-time constants): wave 0
  // brings the tile's 256 bias values, wave 1 its layer-scale values, wave 2 the sample-map entries of its rows into the
  // tile parity's LDS slot (gemm_p8.hip: per-column operands from LDS); the other waves touch a landing slot.
  auto tile_dma = [&]() {
    const char* src = reinterpret_cast<const char*>(g_epi_zero256);
    char* dst = smem + kTrashOff + wave * 1024;
    int dlane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(dlane));
    if constexpr (kHasBias) {
      if (wave == 0) {
        src = p.bias ? reinterpret_cast<const char*>(p.bias + ctn * BN) : reinterpret_cast<const char*>(g_epi_zero256);
        dst = smem + kColsOff + tile_par * kColsSlot;
      }
    }
    if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
      if (wave == 1) {
        src = p.vec1 ? reinterpret_cast<const char*>(p.vec1 + ctn * BN) : reinterpret_cast<const char*>(g_epi_one256);
        dst = smem + kColsOff + tile_par * kColsSlot + 1024;
      }
      if (wave == 2) {     // sample-map entries from the tile's first compact sample on (the tile's rows span at most 4)
        const int s0 = (ctm * BM + p.m_base) / p.rows_per_sample;
        src = p.sample_map ? reinterpret_cast<const char*>(p.sample_map + s0) : reinterpret_cast<const char*>(g_epi_zero256);
        dst = smem + kColsOff + tile_par * kColsSlot + 2048;
      }
    }
    glds16(src + dlane * 16, dst);
  };
  const unsigned lds_base = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem);
  auto lds_read8 = [&](unsigned addr, float* o) {
    f32x4 q0, q1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(addr) : "memory");
    o[0] = q0[0]; o[1] = q0[1]; o[2] = q0[2]; o[3] = q0[3]; o[4] = q1[0]; o[5] = q1[1]; o[6] = q1[2]; o[7] = q1[3];
  };
  auto cols_from_lds = [&](int ncl, EpiCols& c) {
    const unsigned addr = lds_base + (unsigned)(kColsOff + tile_par * kColsSlot + ncl * 4);
    if constexpr (kHasBias) {
      float b[8];
      lds_read8(addr, b);
#pragma unroll
      for (int k = 0; k < 4; ++k) c.bias[k] = ef32x2{b[2 * k], b[2 * k + 1]};
    }
    if constexpr (EPI == MEMHIP_EPI_RESIDUAL) lds_read8(addr + 1024, c.g);
  };

  // residual rows of the parked tile (RESIDUAL with a sample map): recomputed at store time from the LDS copy of the map
  // that the parked tile's first K-tile brought in (slot ppar; the current tile writes the other slot)
  const float inv_rps = (EPI == MEMHIP_EPI_RESIDUAL && (p.rowmask || p.sample_map)) ? __frcp_rn((float)p.rows_per_sample) : 0.f;
  int ps0 = 0;                                             // first compact sample of the parked tile

  // ---- the deferred stores of quadrant Q of the parked tile (Q = 2 i + j: row half i, column half j)
  auto store_quadrant = [&](auto QC) {
    constexpr int Q = decltype(QC)::value, i = Q >> 1, j = Q & 1;
    int el;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
    const int m0 = ptm * BM + i * (BM / 2) + wr * (MF * 16) + (el & 15);
    const int n = ptn * BN + j * 128 + wc * 32 + (el >> 4) * 8;
    // first tile of the workgroup: nothing is parked, the stores go to a scratch line (same instruction count).  The
    // choice is ARITHMETIC on the address (real & mask | scratch & ~mask): a pointer select becomes a branch around
    // each store, and a phase's load segment must stay one basic block (counted waits).
    // (built from 32-bit halves: a zero-extended lane offset keeps a zero register alive across the whole loop)
    const unsigned long long tb = (unsigned long long)reinterpret_cast<char*>(g_p8d_trash);      // 2 KiB aligned
    const unsigned tlo = ((unsigned)tb | (unsigned)((el & 63) * 16)) & ~pmask, thi = (unsigned)(tb >> 32) & ~pmask;
    auto sel = [&](const void* real, unsigned trash_off) {
      const unsigned long long r = (unsigned long long)real;
      const unsigned lo = ((unsigned)r & pmask) | (tlo + (trash_off & ~pmask));
      const unsigned hi = ((unsigned)(r >> 32) & pmask) | thi;
      return reinterpret_cast<char*>(((unsigned long long)hi << 32) | lo);
    };
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int m = m0 + mf * 16;
      const unsigned* o = pk[Q][mf];
      if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
        // (no branch in a load segment: without a map the LDS slot holds zeros and the select keeps m)
        const int mm = m + p.m_base;
        const int smp = (int)(((float)mm + 0.5f) * inv_rps);
        int kd;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(kd)
                     : "v"(lds_base + (unsigned)(kColsOff + ppar * kColsSlot + 2048) + (unsigned)((smp - ps0) << 2))
                     : "memory");
        const int rmap = kd * p.rows_per_sample + (mm - smp * p.rows_per_sample);
        const int rr = (rmap & smask) | (m & ~smask);
        char* q = sel(p.resid + (long long)rr * p.ldr + n, 0);
        st16(q, o[0], o[1], o[2], o[3]);
        st16(q + 16, o[4], o[5], o[6], o[7]);
      } else if constexpr (PW == 4) {
        st16_nt(sel(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n, 0), o[0], o[1], o[2], o[3]);
      } else {
        st16_nt(sel(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n, 0), o[0], o[1], o[2], o[3]);
        st16_nt(sel(reinterpret_cast<__bf16*>(p.out1) + (long long)m * p.ldo1 + n, 1024), o[4], o[5], o[6], o[7]);
      }
    }
  };

  P8D_READ_B(bx, 0, 0);
  const int npairs = nk >> 1;                              // K-tiles go in pairs (K % 128 == 0): static buffer parity
  for (int t = 0; t < my_tiles; ++t) {
    // ---- first pair of K-tiles: the previous tile's stores ride in the first K-tile.  Waits (see the header): behind
    // the LDS-DMA that a phase retires lie the tile DMA (first five phases) and the stores issued since.
    P8D_STAMP_AT(0);
    tile_dma();
    {
      constexpr int bc = 0;
      P8D_KTILE(bx, by, 1 + S, 1 + 2 * S, 1 + 3 * S, 1 + 4 * S, true);
      tm1 = tm2; tn1 = tn2; k1 = k2;
      advance2();
    }
    P8D_STAMP_AT(1);
    {
      constexpr int bc = 1;
      P8D_KTILE(by, bx, 1 + 4 * S, 4 * S, 3 * S, 2 * S, false);
      tm1 = tm2; tn1 = tn2; k1 = k2;
      advance2();
    }
    P8D_STAMP_AT(2);
    for (int kp = 1; kp < npairs; ++kp) {
      {
        constexpr int bc = 0;
        P8D_KTILE(bx, by, 0, 0, 0, 0, false);
        tm1 = tm2; tn1 = tn2; k1 = k2;
        advance2();
      }
      {
        constexpr int bc = 1;
        P8D_KTILE(by, bx, 0, 0, 0, 0, false);
        tm1 = tm2; tn1 = tn2; k1 = k2;
        advance2();
      }
    }
    // ---- the tile is complete: its arithmetic, all eight waves together (waves 0-3 wait for the last compute segment of
    // waves 4-7; waves 4-7 fall half a phase behind again after it)
    P8D_STAMP_AT(3);
    if (wr == 0) P8D_BARRIER();
    P8D_STAMP_AT(4);
    {
      const int tm = ctm, tn = ctn;
      int elane;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
      const int mrow = tm * BM + wr * (MF * 16) + (elane & 15);
      const int ncol = tn * BN + wc * 32 + (elane >> 4) * 8;
      if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
        // row by row with the fp32 residual loads of the next two rows in flight (inline asm, hand-counted waits: see
        // gemm_p8.hip); the only vector-memory instructions in between are these loads and, with COPY, one store per row
        constexpr int kResidAhead = 2;
        constexpr int kStores = COPY ? 1 : 0;
        auto row_m = [&](int r) { return mrow + ((r >> 2) & 1) * (BM / 2) + (r & 3) * 16; };
        auto row_n = [&](int r) { return ncol + (r >> 3) * 128; };
        const float* xbase = p.aux ? reinterpret_cast<const float*>(p.aux) : p.resid;
        const long long xld = p.aux ? p.ldaux : p.ldr;
        const float* rmb = p.rowmask ? p.rowmask : &g_epi_one;
        const int s0 = (tm * BM + p.m_base) / p.rows_per_sample;
        const unsigned kid_lds = lds_base + (unsigned)(kColsOff + tile_par * kColsSlot + 2048);
        f32x4 xa[3], xb[3];
        float rmv[3];
        auto issue_row = [&](int r, int slot) {
          const int m = row_m(r);
          const int mm = m + p.m_base;
          const int smp = (int)(((float)mm + 0.5f) * inv_rps);                    // rows < 2^21 (p8d_fits)
          int kd;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(kd) : "v"(kid_lds + (unsigned)((smp - s0) << 2)) : "memory");
          const int rrow = p.sample_map ? kd * p.rows_per_sample + (mm - smp * p.rows_per_sample) : m;
          const float* src = xbase + (long long)rrow * xld + row_n(r);
          const float* rsrc = rmb + (p.rowmask ? smp : 0);
          asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dword %2, %4, off"
                       : "=&v"(xa[slot]), "=&v"(xb[slot]), "=&v"(rmv[slot]) : "v"(src), "v"(rsrc) : "memory");
        };
        issue_row(0, 0);
        issue_row(1, 1);
        float cs[8];
        EpiCols cols;
        auto do_row = [&](int r, int slot, auto waitc) {
          __builtin_amdgcn_sched_barrier(0);
          if ((r & 7) == 0) cols_from_lds(wc * 32 + (elane >> 4) * 8 + (r >> 3) * 128, cols);
          if (r + kResidAhead < 16) issue_row(r + kResidAhead, (slot + kResidAhead) % 3);
          const int q = ((r >> 2) & 1) * 2 + (r >> 3), mf = r & 3;
          float v[8];
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[nf * 4 + c] = acc[q][mf][nf][c];
          asm volatile("s_waitcnt vmcnt(%3)" : "+v"(xa[slot]), "+v"(xb[slot]), "+v"(rmv[slot]) : "n"(decltype(waitc)::value) : "memory");
          EpiRow<EPI> row;
#pragma unroll
          for (int c = 0; c < 4; ++c) { row.x[c] = xa[slot][c]; row.x[4 + c] = xb[slot][c]; }
          row.rm = rmv[slot];
          row.row = 0;
          epi8_math<EPI, COPY ? 2 : 0>(p, row_m(r), row_n(r), v, cs, cols, row, pk[q][mf]);
        };
        using W2 = std::integral_constant<int, 2 * kStores + 6>;     // S(r-2) L(r+1) S(r-1) L(r+2) behind L(r)
        using W1 = std::integral_constant<int, 2 * kStores + 3>;     // row 14: no L(16)
        using W0 = std::integral_constant<int, 2 * kStores>;         // row 15
        do_row(0, 0, std::integral_constant<int, 6>{});              // behind L(0): L(1) L(2)
        do_row(1, 1, std::integral_constant<int, kStores + 6>{});    // behind L(1): L(2) S(0) L(3)
#pragma unroll
        for (int r = 2; r < 14; ++r) do_row(r, r % 3, W2{});
        do_row(14, 14 % 3, W1{});
        do_row(15, 15 % 3, W0{});
        ps0 = s0;
      } else {
        constexpr bool kNeedRows = EPI == MEMHIP_EPI_DGELU || EPI == MEMHIP_EPI_MUL_AUX;
        constexpr int kAhead = kNeedRows ? kEpiAhead : 0;
        EpiRow<EPI> rows[4][MF];
        auto load_batch = [&](int b) {
          if constexpr (kNeedRows) {
            const int jj = b >> 1, ii = b & 1;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) epi_row_load<EPI>(p, mrow + ii * (BM / 2) + mf * 16, ncol + jj * 128, rows[b][mf]);
          }
        };
#pragma unroll
        for (int b = 0; b < kAhead && b < 4; ++b) load_batch(b);
        float cs[8];
        EpiCols cols;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = b >> 1, i = b & 1;
          const int n = ncol + j * 128;
          __builtin_amdgcn_sched_barrier(0);
          if (i == 0) {
            float z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
            for (int r = 0; r < 8; ++r) cs[r] = z;
            cols_from_lds(wc * 32 + (elane >> 4) * 8 + j * 128, cols);
          }
          if (b + kAhead < 4) load_batch(b + kAhead);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const int m = mrow + i * (BM / 2) + mf * 16;
            float v[8];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
              for (int r = 0; r < 4; ++r) v[nf * 4 + r] = acc[i * 2 + j][mf][nf][r];
            epi8_math<EPI, 0>(p, m, n, v, cs, cols, rows[b][mf], pk[i * 2 + j][mf]);
          }
          if (i == 1) colsum_flush16(p, n, cs, elane);
        }
      }
    }
    // the finished values exist HERE: their only readers are the next iteration's stores, and hipcc otherwise sinks the
    // arithmetic that produces them below the barrier into the loop latch (and spills its inputs to get there)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < PW; ++j) asm volatile("" : "+v"(pk[q][i][j]));
    P8D_STAMP_AT(5);
    pmask = ~0u;
    ptm = ctm; ptn = ctn; ppar = tile_par;
    c_tile += gridDim.x;
    if (c_tile < ntiles) decode(c_tile, ctm, ctn);
    tile_par ^= 1;
    if (wr == 1) P8D_BARRIER();
  }
  // ---- the last tile of this workgroup leaves here
  store_quadrant(QC0{});
  store_quadrant(QC1{});
  store_quadrant(QC2{});
  store_quadrant(QC3{});
  if (wr == 0) P8D_BARRIER();                              // balances the last stagger barrier of waves 4-7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup
}

template <int EPI, bool COPY>
int launch_p8d(const GemmArgs& p, hipStream_t s, int num_cu) {
  const int ntm = p.M / BM, ntn = p.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8d_kernel<EPI, COPY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAll);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_p8d: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int grid = ntm * ntn < num_cu ? ntm * ntn : num_cu;
  hipLaunchKernelGGL((gemm_p8d_kernel<EPI, COPY>), dim3(grid), dim3(kThreads), kLdsAll, s, p, ntm, ntn, opt(OPT_GEMM_STAGGER));
  return check_launch("gemm_bf16_nt(p8d)");
}

}  // namespace

#ifdef P8D_STAMP
extern "C" int memhip_debug_p8d_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_p8d_stamps), sizeof(unsigned long long) * 256 * 2 * 8 * 6) == hipSuccess ? 0 : -1;
}
#endif

namespace memhip {

// Whole 256-row tiles of the shapes gemm_p8 takes, for the epilogues with a deferred form; MEMHIP_EUNSUPPORTED otherwise
// (the caller falls back to gemm_p8).
int gemm_p8d_dispatch(const GemmArgs& p, hipStream_t s) {
  const bool vec = ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux | p.colscale_n) & 7) == 0;
  const bool rows_ok = (!(p.rowmask || p.sample_map) || (long long)p.M + p.m_base < (1 << 21)) &&
                       (!p.sample_map || p.rows_per_sample >= 86);
  if (p.M < 4096 || p.M % BM != 0 || p.N % BN != 0 || p.K % (2 * BK) != 0 || !vec || !rows_ok) return MEMHIP_EUNSUPPORTED;
  const int num_cu = usable_cus(s);
  if (!num_cu) return MEMHIP_EUNSUPPORTED;
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch_p8d<MEMHIP_EPI_BIAS_BF16, false>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch_p8d<MEMHIP_EPI_BIAS_GELU, false>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU_DG: return launch_p8d<MEMHIP_EPI_BIAS_GELU_DG, false>(p, s, num_cu);
    case MEMHIP_EPI_DGELU: return launch_p8d<MEMHIP_EPI_DGELU, false>(p, s, num_cu);
    case MEMHIP_EPI_MUL_AUX: return launch_p8d<MEMHIP_EPI_MUL_AUX, false>(p, s, num_cu);
    case MEMHIP_EPI_RESIDUAL:      // (with the bf16 branch copy the row loop spills: gemm_p8 takes those)
      return p.out0 ? MEMHIP_EUNSUPPORTED : launch_p8d<MEMHIP_EPI_RESIDUAL, false>(p, s, num_cu);
    default: return MEMHIP_EUNSUPPORTED;
  }
}

}  // namespace memhip
