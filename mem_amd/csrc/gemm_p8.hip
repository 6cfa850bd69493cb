// 256x256-tile, 64-deep, phase-interleaved persistent bf16 NT GEMM (same contract and epilogues
// as gemm.hip / gemm256.hip).
//
// What gemm256.hip leaves on the table: its eight waves run in lockstep -- all of them issue the
// LDS-DMA of the next stage, all of them read their fragments, all of them run their 32 MFMAs, so
// on every SIMD the matrix pipe idles while both of its waves load.  Here the two waves of a SIMD
// (waves w and w+4) run half a phase apart:
//
//   * a K-tile (64 deep) of the 256x256 output tile is four 128x128 quadrants; a PHASE is one
//     quadrant: a LOAD segment (this phase's fragment reads, 2 LDS-DMA instructions of prefetch),
//     s_barrier, a COMPUTE segment of 16 MFMAs (16x16x32), s_barrier;
//   * waves 4-7 execute one extra s_barrier up front, so their LOAD segments coincide with the
//     COMPUTE segments of waves 0-3 and vice versa: matrix work beside memory work on every SIMD;
//   * quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0): the A sub-tile (64 rows x 64 k per wave) and
//     both B sub-tiles (32 columns x 64 k) live in registers; phase 4 already reads B0 of the NEXT
//     K-tile into the register set B1 has left, so the phases read 8 / 4 / 8 / 4 fragments;
//   * operands move HBM -> LDS by LDS-DMA as 16 KiB half-tiles (128 rows x 64 k, 128-byte rows,
//     16-byte chunks XOR-swizzled with (row >> 1) & 7 on the SOURCE address), two buffers of
//     {A0, A1, B0, B1}; one half-tile is issued per phase, in the order the slots fall free:
//         phase 1 of K-tile c: A1(c+1)   phase 2: B0(c+2)   phase 3: A0(c+2)   phase 4: B1(c+2)
//     so every half-tile is in flight for at least five phases; s_waitcnt vmcnt(10) (five
//     half-tiles may stay in flight) before the first barrier of every phase retires exactly
//     what the NEXT phase reads.  The stream never stops at output-tile boundaries (persistent
//     workgroups, XCD-contiguous tile order) and idles on the last K-tile at the very end.
//
// Hazards (barrier intervals are global: waves 0-3 load in even intervals, waves 4-7 in odd):
//   RAW  LDS-DMA -> ds_read: the counted vmcnt sits before a phase's first barrier, the read in
//        the next phase -- every wave has waited and passed one more barrier by then.
//   WAR  ds_read -> LDS-DMA: a slot is restaged two phases after the phase that read it (the reads
//        are retired by lgkmcnt(0) right after that phase's first barrier, one full phase earlier).
// The MFMAs take (B fragment, A fragment), so an accumulator tile is C^T: a lane holds four
// consecutive output columns of one row, and the B rows are interleaved so that its two n-fragments
// are 8 consecutive columns: the epilogue runs straight out of the registers with 16-byte accesses
// (no LDS transposition, no barrier) while the prefetch stream keeps running under it.
#include "common.h"
#include "gemm_epilogue.hpp"
#include <type_traits>

#ifndef P8_EPI_AHEAD
#define P8_EPI_AHEAD 2
#endif
#ifndef P8_REALIGN
#define P8_REALIGN 1   // 1: all eight waves run a tile's epilogue together (waves 0-3 wait, waves 4-7 re-stagger after it);
                       // 0 (each group enters the epilogue when it is done) measured 8-12 % slower on the K = 768 shapes
#endif
#ifndef P8_EPI_RESID_LATE
#define P8_EPI_RESID_LATE 2
#endif
#ifndef P8_STAGE_MID
#define P8_STAGE_MID 0   // 1: a phase's LDS-DMA issue sits between the two halves of its MFMA block instead of in its load
                         // segment (the counted waits then allow that many fewer pieces in flight)
#endif

#ifndef P8_PRIO_MODE
#define P8_PRIO_MODE 0   // 0: s_setprio 1 around every MFMA block; 1: none; 2: none + waves 4-7 at priority 1 for the whole kernel
#endif
#if P8_PRIO_MODE == 0
#define P8_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define P8_PRIO(x) do { } while (0)
#endif

namespace {

using namespace memhip;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 512;
constexpr int kHalf = 128 * BK * 2;     // 16 KiB: 128 rows x 64 k (a B half-tile; an A half-tile when BMT = 256)
// The tile HEIGHT is a template parameter: BMT = 256 is the main kernel; BMT = 128 (A half-tiles of 64 rows,
// 8 MFMAs per phase) handles the rows a launch of 256-row tiles would leave to a poorly filled last round.
constexpr int kEpiAhead = P8_EPI_AHEAD;   // batches of epilogue row loads in flight ahead of the compute

template <int BMT> struct P8Geo {
  static constexpr int kAHalf = (BMT / 2) * BK * 2;        // bytes of an A half-tile
  static constexpr int kBOff = 2 * kAHalf;                  // B0 behind A0, A1
  static constexpr int kBuf = 2 * kAHalf + 2 * kHalf;       // A0 A1 B0 B1
  static constexpr int kLds = 2 * kBuf;
  static constexpr int MF = BMT / 64;                       // 16-row fragments per wave and A half
  static constexpr int kAPieces = BMT / 128;                // LDS-DMA instructions per wave for an A half-tile
  // glds in flight that a phase's wait must leave alone (= the five half-tiles issued after the one needed)
  static constexpr int kWaitA = 3 * 2 + 2 * kAPieces;       // phases 2 and 4 (prologue): B A B A B ... newest
  static constexpr int kWaitB = 2 * 2 + 3 * kAPieces;       // phases 1 and 3
};
enum { HA0 = 0, HA1 = 1, HB0 = 2, HB1 = 3 };
constexpr int kGroupM = 8;            // tile rows per group of the tile order

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// LDS image of a half-tile: 128-byte rows, 16-byte chunk c of row r at position c ^ key(r).  The 16
// lanes of a ds_read_b128 group read one chunk of 16 rows and must see 16 distinct (r & 1, key(r)):
//   A rows are 16x + i           -> key_a(r) = (r >> 1) & 7
//   B rows are 32x + 8(i >> 2) + 4 nf + (i & 3)  (i = lane & 15; this interleave makes the two
//   n-fragments of a lane 8 CONSECUTIVE output columns) -> key_b(r) = ((r >> 1) & 1) | 2 ((r >> 3) & 3)
__device__ __forceinline__ int key_a(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int key_b(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

#define P8_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define P8_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// GUARD: the last tile row may reach past M (rows clamped on load, guarded on store).  The 256-row kernel is only
// instantiated without it; the 128-row kernel in both forms (the rows it is handed are usually whole tiles too).
template <int EPI, int BMT, bool GUARD>
__global__ __launch_bounds__(kThreads) void gemm_p8_kernel(GemmArgs p, int ntm, int ntn, int stagger) {
  using G = P8Geo<BMT>;
  constexpr int kBuf = G::kBuf, MF = G::MF, AP = G::kAPieces, kAHalf = G::kAHalf;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = p.K / BK;
  const int ntiles = ntm * ntn;
  // every XCD (workgroups b, b+8, ...) takes a contiguous run of tile ids per round
  const int per_xcd = (gridDim.x + 7) / 8;
  const int first = (gridDim.x % 8 == 0) ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  if (total <= 0) return;
  // Phase stagger.  Every workgroup runs the same number of K-tiles per output tile, so all CUs reach their epilogues
  // together: 256 x (128 KiB store + operand rows) is an HBM burst the whole chip waits for while the main loops leave HBM
  // idle (tools/epi_probe.py: the fc2-dgrad epilogue costs 118 us of 395).  Workgroups that own one tile less than the
  // fullest have a tile time of slack: they start late by a per-workgroup fraction of `stagger` cycles per K-tile of a
  // tile (stagger < 0: every workgroup is delayed, |stagger| cycles per K-tile).
  if (stagger != 0) {
    const int max_tiles = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    if (stagger < 0 || my_tiles < max_tiles) {
      const unsigned frac = (((unsigned)blockIdx.x * 2654435761u) >> 22) & 1023u;
      const unsigned long long wait = ((unsigned long long)(stagger < 0 ? -stagger : stagger) * (unsigned)nk * frac) >> 10;
      const unsigned long long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
  }

  // ---- LDS-DMA issue constants: a 128-row half-tile is 16 pieces of 8 rows x 128 B, this wave moves pieces
  // 2*wave and 2*wave+1; a 64-row A half-tile (BMT = 128) is 8 pieces, one per wave
  int prow[2];
  unsigned offA[2], offB[2], pch[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    prow[j] = (wave * 2 + j) * 8 + (lane >> 3);
    offB[j] = (unsigned)((long long)prow[j] * p.ldb * 2) + (unsigned)(((lane & 7) ^ key_b(prow[j])) * 16);
  }
  int arow[2];                                            // rows of this wave's A pieces inside an A half-tile
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    arow[j] = (wave * AP + j) * 8 + (lane >> 3);
    pch[j] = (unsigned)(((lane & 7) ^ key_a(arow[j])) * 16);
    offA[j] = (unsigned)((long long)arow[j] * p.lda * 2) + pch[j];
  }
  // tile id -> (tm, tn): ids sweep groups of kGroupM tile rows column by column, so the 32
  // consecutive ids one XCD takes per round form an 8 x 4 block of tiles: its L2 is filled with
  // 8 A panels + 4 B panels instead of 1 + 32
  auto decode = [&](int id, int& tm, int& tn) {
    const int gsz = kGroupM * ntn;
    const int grp = id / gsz, rem = id - grp * gsz;
    const int rows = ntm - grp * kGroupM < kGroupM ? ntm - grp * kGroupM : kGroupM;
    tn = rem / rows;
    tm = grp * kGroupM + (rem - tn * rows);
  };
  auto stage = [&](int H, int buf, int tm, int tn, int kt) {
    if (H == HA0 || H == HA1) {
      char* slot = smem + buf * kBuf + (H == HA1 ? kAHalf : 0) + wave * AP * 1024;
      const int r0 = tm * BMT + (H == HA1 ? BMT / 2 : 0);
      const char* base = reinterpret_cast<const char*>(p.A) + ((long long)r0 * p.lda + kt * BK) * 2;
      if (GUARD && tm == ntm - 1 && r0 + BMT / 2 > p.M) {      // last M tile: clamp rows to M-1
#pragma unroll
        for (int j = 0; j < AP; ++j) {
          int gr = r0 + arow[j];
          gr = gr < p.M ? gr : p.M - 1;
          glds16(base + (long long)(gr - r0) * p.lda * 2 + pch[j], slot + j * 1024);
        }
      } else {
#pragma unroll
        for (int j = 0; j < AP; ++j) glds16(base + offA[j], slot + j * 1024);
      }
    } else {
      char* slot = smem + buf * kBuf + G::kBOff + (H == HB1 ? kHalf : 0) + wave * 2048;
      const int c0 = tn * BN + (H == HB1 ? 128 : 0);
      const char* base = reinterpret_cast<const char*>(p.B) + ((long long)c0 * p.ldb + kt * BK) * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(base + offB[j], slot + j * 1024);
    }
  };
  // K-tile cursors of the prefetch stream (clamped to the last K-tile at the end of the stream)
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
  P8_WAIT_VM(G::kWaitA);
  P8_BARRIER();
  if (wr == 1) P8_BARRIER();                            // waves 4-7 run half a phase behind
#if P8_PRIO_MODE == 2
  if (wr == 1) __builtin_amdgcn_s_setprio(1);
#endif

  // ---- fragment read addresses: row = 16*x + (lane & 15), chunk = 4*kh + (lane >> 4)
  const int sw = (lane >> 1) & 7;
  const int roff0 = (lane & 15) * 128 + ((((lane >> 4)) ^ sw) << 4);
  const char* rdA[2] = {smem + wr * (MF * 2048) + roff0, smem + wr * (MF * 2048) + (roff0 ^ 64)};
  const int bi = lane & 15;
  const int roffb = (((bi >> 2) * 8 + (bi & 3)) * 128) + (((lane >> 4) ^ key_b((bi >> 2) * 8 + (bi & 3))) << 4);
  const char* rdB[2] = {smem + G::kBOff + wc * 4096 + roffb, smem + G::kBOff + wc * 4096 + (roffb ^ 64)};

  f32x4 acc[4][MF][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int c_tile = first, c_k = 0;
  bf16x8 a[MF][2], bx[2][2], by[2][2];

#define P8_READ_A(half)                                                                                   \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)      \
      a[mf][kh] = *reinterpret_cast<const bf16x8*>(rdA[kh] + bo + (half) * kAHalf + mf * 2048)
#define P8_READ_B(dst, boff, half)                                                                        \
  _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)       \
      dst[nf][kh] = *reinterpret_cast<const bf16x8*>(rdB[kh] + (boff) + (half) * kHalf + nf * 512)
#define P8_MFMA_HALF(q, bsrc, kh)                                                                         \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)      \
      acc[q][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bsrc[nf][kh], a[mf][kh], acc[q][mf][nf], 0, 0, 0)
#define P8_MFMA(q, bsrc)                                                                                  \
  do {                                                                                                    \
    P8_PRIO(1);                                                                        \
    P8_MFMA_HALF(q, bsrc, 0);                                                                             \
    P8_MFMA_HALF(q, bsrc, 1);                                                                             \
    P8_PRIO(0);                                                                        \
  } while (0)
// compute segment with the phase's LDS-DMA issue in the middle (P8_STAGE_MID)
#define P8_MFMA_STAGE(q, bsrc, STAGE_CALL)                                                                \
  do {                                                                                                    \
    P8_PRIO(1);                                                                        \
    P8_MFMA_HALF(q, bsrc, 0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    STAGE_CALL;                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    P8_MFMA_HALF(q, bsrc, 1);                                                                             \
    P8_PRIO(0);                                                                        \
  } while (0)
// One K-tile = four phases.  B0 of this K-tile is already in `bq0` (read during phase 4 of the
// previous K-tile); phase 4 reads B0 of the next K-tile into `bq1`, so the two register sets swap
// roles from one K-tile to the next and the phases read 8 / 4 / 8 / 4 fragments.
#if P8_STAGE_MID
#define P8_PHASE(READS, STAGE_CALL, WAITN, PIECES, q, bsrc)                                               \
    READS;                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    P8_WAIT_VM((WAITN) - (PIECES));                                                                       \
    P8_BARRIER();                                                                                         \
    P8_MFMA_STAGE(q, bsrc, STAGE_CALL);                                                                   \
    P8_BARRIER();
#else
#define P8_PHASE(READS, STAGE_CALL, WAITN, PIECES, q, bsrc)                                               \
    READS;                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    STAGE_CALL;                                                                                           \
    P8_WAIT_VM(WAITN);                                                                                    \
    P8_BARRIER();                                                                                         \
    P8_MFMA(q, bsrc);                                                                                     \
    P8_BARRIER();
#endif
#define P8_KTILE(bq0, bq1)                                                                                  \
  do {                                                                                                    \
    const int bo = bc * kBuf;                                                                             \
    /* phase 1: quadrant (A0, B0) */                                                                      \
    P8_PHASE(P8_READ_A(0), stage(HA1, bc ^ 1, tm1, tn1, k1), G::kWaitB, AP, 0, bq0)                       \
    /* phase 2: quadrant (A0, B1) */                                                                      \
    P8_PHASE(P8_READ_B(bq1, bo, 1), stage(HB0, bc, tm2, tn2, k2), G::kWaitA, 2, 1, bq1)                   \
    /* phase 3: quadrant (A1, B1) */                                                                      \
    P8_PHASE(P8_READ_A(1), stage(HA0, bc, tm2, tn2, k2), G::kWaitB, AP, 3, bq1)                           \
    /* phase 4: quadrant (A1, B0); B0 of the next K-tile comes from the other buffer */                   \
    P8_PHASE(P8_READ_B(bq1, (bc ^ 1) * kBuf, 0), stage(HB1, bc, tm2, tn2, k2), G::kWaitA, 2, 2, bq0)      \
  } while (0)
#define P8_READ_B0_FIRST() P8_READ_B(bx, 0, 0)

  P8_READ_B0_FIRST();
  // K-tiles go in pairs (K % 128 == 0), so that buffer parity and the B register roles are static
  for (int c = 0; c < total; c += 2) {
    {
      constexpr int bc = 0;
      P8_KTILE(bx, by);
      tm1 = tm2; tn1 = tn2; k1 = k2;
      advance2();
    }
    {
      constexpr int bc = 1;
      P8_KTILE(by, bx);
      tm1 = tm2; tn1 = tn2; k1 = k2;
      advance2();
    }

    if (c_k == nk - 2) {
      // waves 0-3 wait for the last compute segment of waves 4-7, so that all eight waves run the
      // (VALU-bound, barrier-free) epilogue together; waves 4-7 fall half a phase behind again after it
      if (P8_REALIGN && wr == 0) P8_BARRIER();
      // ---- epilogue of tile c_tile straight out of the accumulators: the MFMAs ran with swapped
      // operands, so a lane holds 4 consecutive columns (registers) of one row (lane & 15)
      int tm, tn;
      decode(c_tile, tm, tn);
      const int mrow = tm * BMT + wr * (MF * 16) + (lane & 15);
      const int ncol = tn * BN + wc * 32 + (lane >> 4) * 8;
      // four batches (j = column half, i = row half) of MF rows each.  The per-row operand loads (GELU input /
      // residual) of batch b + kAhead are issued before batch b is computed: with one batch in flight
      // a CU has 32 KB outstanding, i.e. ~20 GB/s per CU at ~1.5 us latency -- the epilogue was latency
      // bound, not HBM bound.  Row indices are clamped instead of branched, so that nothing orders the loads.
      // (the residual epilogue carries 8 registers per row: no room for a second batch beside 128 accumulators)
#ifdef P8_RESID_AHEAD
      constexpr int kAhead = (EPI == MEMHIP_EPI_RESIDUAL || EPI == MEMHIP_EPI_PATCH_EMBED) ? P8_RESID_AHEAD : kEpiAhead;
#else
      constexpr int kAhead = (EPI == MEMHIP_EPI_RESIDUAL || EPI == MEMHIP_EPI_PATCH_EMBED) ? 0 : kEpiAhead;
#endif
      // residual epilogue: batch 0 alone; batches 1 and 2 go out together once batch 0 has released its
      // accumulators and row registers (the accumulators are re-zeroed after the loop, not inside it)
#ifdef P8_RESID_AHEAD
      constexpr bool kLate = false;
#else
      constexpr bool kLate = (EPI == MEMHIP_EPI_RESIDUAL || EPI == MEMHIP_EPI_PATCH_EMBED) && P8_EPI_RESID_LATE;
#endif
      // The row guard (m < M) is a per-lane branch: a basic block per row, and at every block entry hipcc's waitcnt pass
      // falls back to s_waitcnt vmcnt(0) in front of the first use of a loaded row -- which also waits for the STORE of
      // the previous row (one store round trip per row, 16 per tile: the GELU' epilogue spent 22 us per tile that way).
      // The 256-row kernel therefore only takes M % 256 == 0 (the launcher hands the remaining rows to the 128-row
      // kernel, which keeps the guard): its epilogue is one basic block with counted waits -- loads complete under the
      // arithmetic, stores are never waited for.
      if constexpr (EPI == MEMHIP_EPI_RESIDUAL && BMT == 256) {
        // Residual epilogue of the 256-row kernel.  A row carries 8 fp32 registers of residual input next to its 8
        // accumulators, and about 80 registers are free beside the 128 accumulators, the B fragment of the next tile and the
        // lane state of the prefetch stream: the generic form (batches of 4 rows) could keep a single batch of row loads in
        // flight, and spilled.  Here the 16 rows of a lane go one by one with the loads of the next kResidAhead rows in
        // flight (48 registers), in one basic block (counted waits, no store is ever waited for).
#ifndef P8_RESID_ROWS_AHEAD
#define P8_RESID_ROWS_AHEAD 2
#endif
        constexpr int kResidAhead = P8_RESID_ROWS_AHEAD;
        EpiRow<EPI> rows[16];
        auto row_m = [&](int r) { return mrow + ((r >> 2) & 1) * (BMT / 2) + (r & 3) * 16; };
        auto row_n = [&](int r) { return ncol + (r >> 3) * 128; };
#pragma unroll
        for (int r = 0; r < kResidAhead; ++r) epi_row_load<EPI>(p, row_m(r), row_n(r), rows[r]);
        float cs[8];
        EpiCols cols;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          __builtin_amdgcn_sched_barrier(0);
          if ((r & 7) == 0) epi_cols_load<EPI>(p, row_n(r), cols);
          if (r + kResidAhead < 16) epi_row_load<EPI>(p, row_m(r + kResidAhead), row_n(r + kResidAhead), rows[r + kResidAhead]);
          const int q = ((r >> 2) & 1) * 2 + (r >> 3), mf = r & 3;
          float v[8];
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[nf * 4 + c] = acc[q][mf][nf][c];
          epilogue8<EPI>(p, row_m(r), row_n(r), v, cs, cols, rows[r]);
        }
      } else {
        constexpr bool kEdge = GUARD;
        EpiRow<EPI> rows[4][MF];
        auto load_batch = [&](int b) {
          const int jj = b >> 1, ii = b & 1;
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const int m = mrow + ii * (BMT / 2) + mf * 16;
            epi_row_load<EPI>(p, kEdge ? (m < p.M ? m : p.M - 1) : m, ncol + jj * 128, rows[b][mf]);
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
          // (without the row branches the whole epilogue is one scheduling region: keep the loads of later batches from
          // being hoisted over this batch -- the row registers are budgeted per batch)
          __builtin_amdgcn_sched_barrier(0);
          if (i == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) cs[r] = 0.f;
            epi_cols_load<EPI>(p, n, cols);
          }
          if constexpr (kLate) {
            if (b == 0) load_batch(0);
            if (P8_EPI_RESID_LATE == 1) {
              if (b == 1) { load_batch(1); load_batch(2); }
              if (b == 2) load_batch(3);
            } else {
              if (b == 1) load_batch(1);
              if (b == 2) { load_batch(2); load_batch(3); }
            }
          } else {
            if (b + kAhead < 4) load_batch(b + kAhead);
          }
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const int m = mrow + i * (BMT / 2) + mf * 16;
            float v[8];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
              for (int r = 0; r < 4; ++r) v[nf * 4 + r] = acc[i * 2 + j][mf][nf][r];
            if (!kEdge || m < p.M) epilogue8<EPI>(p, m, n, v, cs, cols, rows[b][mf]);
            if (!kEdge) __builtin_amdgcn_sched_barrier(0);     // one row at a time (register budget)
          }
          if (i == 1) colsum_flush16(p, n, cs, lane);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][mf][nf][r] = 0.f;
      c_k = 0;
      c_tile += gridDim.x;
      if (P8_REALIGN && wr == 1) P8_BARRIER();
    } else {
      c_k += 2;
    }
  }
  if (wr == 0) P8_BARRIER();                               // balances the last stagger barrier of waves 4-7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup
}

template <int EPI, int BMT, bool GUARD>
int launch_p8g(const GemmArgs& p, hipStream_t s, int num_cu) {
  const int ntm = (p.M + BMT - 1) / BMT, ntn = p.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8_kernel<EPI, BMT, GUARD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, P8Geo<BMT>::kLds);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_p8: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int grid = ntm * ntn < num_cu ? ntm * ntn : num_cu;
  hipLaunchKernelGGL((gemm_p8_kernel<EPI, BMT, GUARD>), dim3(grid), dim3(kThreads), P8Geo<BMT>::kLds, s, p, ntm, ntn,
                     BMT == 256 ? opt(OPT_GEMM_STAGGER) : 0);
  return check_launch("gemm_bf16_nt(p8)");
}

template <int EPI, int BMT>
int launch_p8(const GemmArgs& p, hipStream_t s, int num_cu) {
  if constexpr (BMT == 256) return launch_p8g<EPI, 256, false>(p, s, num_cu);
  else return p.M % 128 == 0 ? launch_p8g<EPI, 128, false>(p, s, num_cu) : launch_p8g<EPI, 128, true>(p, s, num_cu);
}

}  // namespace

namespace memhip {

static int p8_num_cu() {
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return num_cu;
}

static bool p8_fits(const GemmArgs& p) {
  const bool vec = ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux | p.colscale_n) & 7) == 0;      // host twin of vec_ok()
  // Narrow outputs (N = 768: 591 tiles on 256 CUs) are taken too: the launcher hands the rows of a
  // poorly filled last round to the 128x128 kernel (gemm_p8_split_rows).  MEMHIP_GEMM_P8_MIN_N overrides.
  const int min_n = opt(OPT_GEMM_P8_MIN_N);
  const bool wide = p.N >= min_n;
  return p.M >= 4096 && wide && p.N % BN == 0 && p.K % (2 * BK) == 0 && vec;
}

// Rows the persistent launch should take when its last round of tiles would be less than half full
// (0 = take everything): full rounds only, the caller runs the remaining rows on finer tiles.
int gemm_p8_split_rows(const GemmArgs& p) {
  if (!p8_fits(p)) return 0;
  const int num_cu = p8_num_cu();
  if (!num_cu) return 0;
  const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
  const int tiles = ntm * ntn, rounds = tiles / num_cu, rem = tiles % num_cu;
  const int whole = (p.M / BM) * BM;                       // the 256-row kernel takes whole tiles only
  if (rounds < 1 || rem == 0 || rem * 2 > num_cu) return whole == p.M ? 0 : whole;
  return (rounds * num_cu / ntn) * BM;
}

// Returns MEMHIP_EUNSUPPORTED when the shape does not fit this structure (caller falls back).
int gemm_p8_dispatch(const GemmArgs& p, hipStream_t s) {
  if (!p8_fits(p) || p.M % BM != 0) return MEMHIP_EUNSUPPORTED;   // whole 256-row tiles only (no row guard in the epilogue)
  const int num_cu = p8_num_cu();
  if (!num_cu) return MEMHIP_EUNSUPPORTED;
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch_p8<MEMHIP_EPI_BIAS_BF16, 256>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch_p8<MEMHIP_EPI_BIAS_GELU, 256>(p, s, num_cu);
    case MEMHIP_EPI_RESIDUAL: return launch_p8<MEMHIP_EPI_RESIDUAL, 256>(p, s, num_cu);
    case MEMHIP_EPI_DGELU: return launch_p8<MEMHIP_EPI_DGELU, 256>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU_DG: return launch_p8<MEMHIP_EPI_BIAS_GELU_DG, 256>(p, s, num_cu);
    case MEMHIP_EPI_MUL_AUX: return launch_p8<MEMHIP_EPI_MUL_AUX, 256>(p, s, num_cu);
    case MEMHIP_EPI_F32: return launch_p8<MEMHIP_EPI_F32, 256>(p, s, num_cu);
    default: return MEMHIP_EUNSUPPORTED;
  }
}

// The 128-row-tile form for the rows that gemm_p8_split_rows leaves over (any M; same N / K rules).
int gemm_p8_half_dispatch(const GemmArgs& p, hipStream_t s) {
  const bool vec = ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux | p.colscale_n) & 7) == 0;
  if (p.M < 128 || p.N % BN != 0 || p.K % (2 * BK) != 0 || !vec) return MEMHIP_EUNSUPPORTED;
  const int num_cu = p8_num_cu();
  if (!num_cu) return MEMHIP_EUNSUPPORTED;
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch_p8<MEMHIP_EPI_BIAS_BF16, 128>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch_p8<MEMHIP_EPI_BIAS_GELU, 128>(p, s, num_cu);
    case MEMHIP_EPI_RESIDUAL: return launch_p8<MEMHIP_EPI_RESIDUAL, 128>(p, s, num_cu);
    case MEMHIP_EPI_DGELU: return launch_p8<MEMHIP_EPI_DGELU, 128>(p, s, num_cu);
    case MEMHIP_EPI_F32: return launch_p8<MEMHIP_EPI_F32, 128>(p, s, num_cu);
    default: return MEMHIP_EUNSUPPORTED;
  }
}

}  // namespace memhip
