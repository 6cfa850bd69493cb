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

#define P8_EPI_AHEAD 2
#define P8_REALIGN 1   // 1: all eight waves run a tile's epilogue together (waves 0-3 wait, waves 4-7 re-stagger after it);
                       // 0 (each group enters the epilogue when it is done) measured 8-12 % slower on the K = 768 shapes
#define P8_RESID_NT 1    // bit 0 = residual-input loads nontemporal (the row is read once here and again only in backward: it need not push
                         // 110-112 (loads) / 105-107 (stores) / 116 (both); in the step 33.58-33.64 -> 33.48-33.58 ms (loads), 33.66-33.75
                         // (stores: the LayerNorm behind it reads the row from HBM) -- profiles/r05_resid_nt_ab.txt
#define P8_EPI_RESID_LATE 2
#define P8_STAGE_MID 0   // 1: a phase's LDS-DMA issue sits between the two halves of its MFMA block instead of in its load
                         // segment (the counted waits then allow that many fewer pieces in flight)

#define P8_BAR_MID 0   // n > 0: 2 n MFMAs of a phase in front of its first barrier (see P8_PHASE; measured: n = 1 neutral, 2 and 4 slower)
#define P8_PRIO_MODE 0   // 0: s_setprio 1 around every MFMA block; 1: none; 2: none + waves 4-7 at priority 1 for the whole kernel
#define P8_PRIO(x) __builtin_amdgcn_s_setprio(x)

#ifdef P8_STAMP
// diagnostic build (tools/build_variant.sh stamp -DP8_STAMP): wave 0 of every workgroup records s_memtime at the main-loop
// start, epilogue start and epilogue end of its first 10 tiles; memhip_debug_p8_stamps copies the table out.  Never part of
// the shipped library (its fences forbid overlaps the real kernel has: read the SHARES, not the lengths).
__device__ unsigned long long g_p8_stamps[256 * 32];
__device__ unsigned long long g_p8_stamps_rt[256 * 32];     // s_memrealtime (constant 100 MHz) at the same points: the clock
#define P8_STAMP_AT(slot)                                                                                   \
  do {                                                                                                      \
    if (wave == 0 && stamp_tile < 10) {                                                                     \
      unsigned long long t_, r_;                                                                            \
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory"); \
      if (lane == 0) {                                                                                      \
        g_p8_stamps[(blockIdx.x & 255) * 32 + stamp_tile * 3 + (slot)] = t_;                                \
        g_p8_stamps_rt[(blockIdx.x & 255) * 32 + stamp_tile * 3 + (slot)] = r_;                             \
      }                                                                                                     \
    }                                                                                                       \
  } while (0)
#else
#define P8_STAMP_AT(slot) do { } while (0)
#endif
// -DP8_STAMP=2: instead, one stamp per loop iteration (two K-tiles; taken before the epilogue of a tile's last iteration)
// for the first 32 iterations of every workgroup: where inside a tile do the cycles beyond the steady state go?
#if defined(P8_STAMP) && P8_STAMP == 2
#undef P8_STAMP_AT
#define P8_STAMP_AT(slot) do { } while (0)
#define P8_STAMP_IT()                                                                                       \
  do {                                                                                                      \
    if (wave == 0 && stamp_it < 32) {                                                                       \
      unsigned long long t_;                                                                                \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
      if (lane == 0) g_p8_stamps[(blockIdx.x & 255) * 32 + stamp_it] = t_;                                  \
    }                                                                                                       \
    ++stamp_it;                                                                                             \
  } while (0)
#else
#define P8_STAMP_IT() do { } while (0)
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
  // behind the operand buffers: per tile parity the tile's bias and layer-scale columns and the sample-map entries of its
  // rows (work-skipping stochastic depth), then a 1 KiB landing slot per wave for the epilogue-operand prefetch (never read)
  static constexpr int kColsSlot = 3072;                    // [256 f32 bias | 256 f32 layer scale | 256 i32 sample map]
  static constexpr int kColsOff = kLds, kTrashOff = kLds + 2 * kColsSlot;
  static constexpr int kLdsAll = kLds + 2 * kColsSlot + 8 * 1024;
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
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

#define P8_SADDR 1     // 1: operand pieces as global_load_lds with a scalar base + 32-bit lane offset (inline asm) instead of a
                       // 64-bit address per lane (two VALU adds per piece and twice the address traffic): NT GEMMs +0.5-1 %,
                       // weight gradients +2-3 %, step -0.2 ms (tools/exp/r04_run19.sh)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, const void* lds_dst) {
  const unsigned lds = (unsigned)(unsigned long long)((const __attribute__((address_space(3))) char*)lds_dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
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
// COPY (residual epilogue only): the bf16 copy of the branch output (out0) is wanted.  Without it the epilogue has no
// such store at all (it used to go to a 1 KiB scratch line shared by the whole chip: 77 MB of stores per launch).
// (the kernel body as a device function of (block id, block count): gemm_p8_kernel runs it over the whole grid, gemm_p8_pair_kernel
// runs the 256-row form on the first workgroups of a launch and the 128-row form on the rest)
template <int EPI, int BMT, bool GUARD, bool COPY>
__device__ __forceinline__ void p8_body(const GemmArgs& p, int ntm, int ntn, int stagger, int prefetch, const int bid, const int nblk) {
  using G = P8Geo<BMT>;
  constexpr int kBuf = G::kBuf, MF = G::MF, AP = G::kAPieces, kAHalf = G::kAHalf;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = p.K / BK;
  const int ntiles = ntm * ntn;
  // every XCD (workgroups b, b+8, ...) takes a contiguous run of tile ids per round
  const int per_xcd = (nblk + 7) / 8;
  const int first = (nblk % 8 == 0) ? (bid % 8) * per_xcd + bid / 8 : bid;
  const int my_tiles = (ntiles - first + nblk - 1) / nblk;
  const int total = my_tiles * nk;
  if (total <= 0) return;
  // Phase stagger.  Every workgroup runs the same number of K-tiles per output tile, so all CUs reach their epilogues
  // together: 256 x (128 KiB store + operand rows) is an HBM burst the whole chip waits for while the main loops leave HBM
  // idle (tools/epi_probe.py: the fc2-dgrad epilogue costs 118 us of 395).  Workgroups that own one tile less than the
  // fullest have a tile time of slack: they start late by a per-workgroup fraction of `stagger` cycles per K-tile of a
  // tile (stagger < 0: every workgroup is delayed, |stagger| cycles per K-tile).
  if (stagger != 0) {
    const int max_tiles = (ntiles + nblk - 1) / nblk;
    if (stagger < 0 || my_tiles < max_tiles) {
      const unsigned frac = (((unsigned)bid * 2654435761u) >> 22) & 1023u;
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
    // row r of half-tile B_h holds output column 64 (r / 32) + 32 h + r % 32 of the tile: wave wc's 32 columns of B0 and
    // its 32 columns of B1 are 64 ADJACENT columns, i.e. one 128-byte line of a bf16 output row (full-line epilogue)
    offB[j] = (unsigned)((long long)((prow[j] >> 5) * 64 + (prow[j] & 31)) * p.ldb * 2) +
              (unsigned)(((lane & 7) ^ key_b(prow[j])) * 16);
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
        for (int j = 0; j < AP; ++j) {
          if (P8_SADDR) glds16s(base, offA[j], slot + j * 1024);
          else glds16(base + offA[j], slot + j * 1024);
        }
      }
    } else {
      char* slot = smem + buf * kBuf + G::kBOff + (H == HB1 ? kHalf : 0) + wave * 2048;
      const int c0 = tn * BN + (H == HB1 ? 32 : 0);
      const char* base = reinterpret_cast<const char*>(p.B) + ((long long)c0 * p.ldb + kt * BK) * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (P8_SADDR) glds16s(base, offB[j], slot + j * 1024);
        else glds16(base + offB[j], slot + j * 1024);
      }
    }
  };
  // K-tile cursors of the prefetch stream (clamped to the last K-tile at the end of the stream)
  int g2 = 0, id2 = first, k2 = 0, tm2, tn2;            // becomes K-tile c+2
  decode(id2, tm2, tn2);
  auto advance2 = [&]() {
    if (g2 + 1 < total) {
      ++g2;
      if (++k2 == nk) { k2 = 0; id2 += nblk; decode(id2, tm2, tn2); }
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
#define ACC_EL(q, mf, nf, r) acc[q][mf][nf][r]

  int c_tile = first, c_k = 0;
  bf16x8 a[MF][2], bx[2][2], by[2][2];

#define P8_READ_A(half)                                                                                   \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)      \
      a[mf][kh] = *reinterpret_cast<const bf16x8*>(rdA[kh] + bo + (half) * kAHalf + mf * 2048)
#define P8_READ_B(dst, boff, half)                                                                        \
  _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)       \
      dst[nf][kh] = *reinterpret_cast<const bf16x8*>(rdB[kh] + (boff) + (half) * kHalf + nf * 512)
// P8_EXP_NOREAD = 1: no fragment reads at all, 2: A fragments are not read, 3: B fragments are not read
#define P8_NR_A(h) (void)0
#define P8_NR_B(d, o, h) (void)0
#define P8_MFMA_HALF(q, bsrc, kh)                                                                         \
  _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)      \
      acc[q][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bsrc[nf][kh], a[mf][kh], acc[q][mf][nf], 0, 0, 0)
#define P8_MFMA_PART(q, bsrc, kh, m0, m1)                                                                 \
  _Pragma("unroll") for (int mf = (m0); mf < (m1) && mf < MF; ++mf) _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) \
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
#define P8_PHASE(READS, STAGE_CALL, WAITN, PIECES, q, bsrc)                                               \
    READS;                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    STAGE_CALL;                                                                                           \
    P8_WAIT_VM(WAITN);                                                                                    \
    P8_BARRIER();                                                                                         \
    P8_MFMA(q, bsrc);                                                                                     \
    P8_BARRIER();
// X1..X4: vector-memory instructions besides the half-tile stream that the phase's wait must leave in flight as well
// (the epilogue-operand prefetch, one LDS-DMA per wave and loop iteration: see `prefetch_aux`)
#define P8_KTILE(bq0, bq1, X1, X2, X3, X4)                                                                \
  do {                                                                                                    \
    const int bo = bc * kBuf;                                                                             \
    /* phase 1: quadrant (A0, B0) */                                                                      \
    P8_PHASE(P8_READ_A(0), stage(HA1, bc ^ 1, tm1, tn1, k1), G::kWaitB + (X1), AP, 0, bq0)                \
    /* phase 2: quadrant (A0, B1) */                                                                      \
    P8_PHASE(P8_READ_B(bq1, bo, 1), stage(HB0, bc, tm2, tn2, k2), G::kWaitA + (X2), 2, 1, bq1)            \
    /* phase 3: quadrant (A1, B1) */                                                                      \
    P8_PHASE(P8_READ_A(1), stage(HA0, bc, tm2, tn2, k2), G::kWaitB + (X3), AP, 3, bq1)                    \
    /* phase 4: quadrant (A1, B0); B0 of the next K-tile comes from the other buffer */                   \
    P8_PHASE(P8_READ_B(bq1, (bc ^ 1) * kBuf, 0), stage(HB1, bc, tm2, tn2, k2), G::kWaitA + (X4), 2, 2, bq0) \
  } while (0)
#define P8_READ_B0_FIRST() P8_READ_B(bx, 0, 0)

  // ---- one extra LDS-DMA per wave and loop iteration (two K-tiles), always issued, so that the counted waits stay
  // compile-time constants (+1 for the five phases behind it: X1..X4 of P8_KTILE):
  //  * iteration 0 of a tile: wave 0 brings the tile's 256 bias values (1 KiB = one 16-byte-per-lane instruction), wave 1
  //    its 256 layer-scale values into LDS (double-buffered by tile parity: a slow wave may still be reading the previous
  //    tile's copy in its epilogue).  The epilogue then takes its per-column operands from LDS: it used to load them from
  //    global memory at its start, and waiting for the YOUNGEST vector-memory operation is s_waitcnt vmcnt(0) -- every
  //    store issued so far and the whole prefetch stream of the next tile, two to three times per output tile.  Absent
  //    operands come from constant lines (bias 0, scale 1): no branch, no special case in the epilogue.
  //  * later iterations: EPILOGUE-OPERAND PREFETCH.  The GELU' / residual epilogues read a 256 x 256 tile of a second
  //    operand (the GELU input, bf16, 128 KB; the fp32 residual stream, 256 KB) straight from HBM, all workgroups at about
  //    the same time, while the main loops leave HBM idle.  Every wave touches 64 of that tile's 128-byte lines per
  //    iteration (per-lane source addresses; the data lands in a scratch slot and is never read): 512 lines per iteration
  //    and workgroup, the whole operand tile within 2 (bf16) or 4 (fp32) of a K = 768 tile's 6 iterations; the epilogue's
  //    own loads then hit L2.
  constexpr bool kHasAux = EPI == MEMHIP_EPI_DGELU || EPI == MEMHIP_EPI_MUL_AUX || EPI == MEMHIP_EPI_RESIDUAL;
  constexpr bool kHasBias = EPI == MEMHIP_EPI_BIAS_BF16 || EPI == MEMHIP_EPI_BIAS_GELU || EPI == MEMHIP_EPI_RESIDUAL ||
                            EPI == MEMHIP_EPI_BIAS_GELU_DG;
  constexpr int PX = 1;
  const char* pf_base = reinterpret_cast<const char*>(p.A);
  long long pf_ld = 0;                                     // bytes per row
  int pf_lpr = 1;                                          // 128-byte lines per tile row
  if constexpr (kHasAux) {
    if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
      pf_base = p.aux ? reinterpret_cast<const char*>(p.aux) : reinterpret_cast<const char*>(p.resid);
      pf_ld = (p.aux ? p.ldaux : p.ldr) * 4;
      pf_lpr = BN * 4 / 128;
    } else {
      pf_base = reinterpret_cast<const char*>(p.aux);
      pf_ld = p.ldaux * 2;
      pf_lpr = BN * 2 / 128;
    }
  }
  int tile_par = 0;                                        // parity of the current output tile of this workgroup
  int ctm, ctn;                                            // coordinates of the current output tile
  decode(c_tile, ctm, ctn);
  auto iter_dma = [&]() {
    const int it = c_k >> 1;
    const int tm = ctm, tn = ctn;
    const char* src = pf_base;
    char* dst = smem + G::kTrashOff + wave * 1024;
    // (lane id from a fresh mbcnt: no lane constant of this block may stay live across the main loop -- the loop uses
    // every register, the constant is spilled, and its reload waits with vmcnt(0) in every iteration)
    int dlane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(dlane));
    if constexpr (kHasAux) {
      int line = (it - 1) * 512 + wave * 64 + dlane;
      if (!prefetch || it < 1 || line >= BMT * pf_lpr) line = 0;
      int row = tm * BMT + line / pf_lpr;
      if (GUARD) row = row < p.M ? row : p.M - 1;
      src = pf_base + (long long)row * pf_ld + (long long)tn * (pf_lpr * 128) + (line % pf_lpr) * 128;
    }
    if (it == 0) {
      if constexpr (kHasBias) {
        if (wave == 0) {
          src = (p.bias ? reinterpret_cast<const char*>(p.bias + tn * BN) : reinterpret_cast<const char*>(g_epi_zero256)) + dlane * 16;
          dst = smem + G::kColsOff + tile_par * G::kColsSlot;
        }
      }
      if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
        if (wave == 1) {
          src = (p.vec1 ? reinterpret_cast<const char*>(p.vec1 + tn * BN) : reinterpret_cast<const char*>(g_epi_one256)) + dlane * 16;
          dst = smem + G::kColsOff + tile_par * G::kColsSlot + 1024;
        }
        if (wave == 2) {     // sample-map entries from the tile's first compact sample on (the tile's rows span at most 4)
          const int s0 = (tm * BMT + p.m_base) / p.rows_per_sample;
          src = (p.sample_map ? reinterpret_cast<const char*>(p.sample_map + s0) : reinterpret_cast<const char*>(g_epi_zero256)) +
                dlane * 16;
          dst = smem + G::kColsOff + tile_par * G::kColsSlot + 2048;
        }
      }
    }
    glds16(src, dst);
  };
  // the tile's per-column operands of a lane's 8 columns (ncl = first column inside the tile), from LDS
  // (read with inline asm: a C++ LDS read here makes hipcc drain the LDS-DMA stream -- s_waitcnt vmcnt(0) -- first, since
  // it cannot tell that the prefetch stream writes other LDS bytes; the slot itself was filled a whole tile ago, behind
  // the main loop's counted waits and barriers)
  auto lds_read8 = [&](unsigned addr, float* o) {
    f32x4 q0, q1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(addr) : "memory");
    o[0] = q0[0]; o[1] = q0[1]; o[2] = q0[2]; o[3] = q0[3]; o[4] = q1[0]; o[5] = q1[1]; o[6] = q1[2]; o[7] = q1[3];
  };
  auto cols_from_lds = [&](int ncl, EpiCols& c) {
    const unsigned addr = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem) + (unsigned)(G::kColsOff + tile_par * G::kColsSlot + ncl * 4);
    if constexpr (kHasBias) {
      float b[8];
      lds_read8(addr, b);
#pragma unroll
      for (int k = 0; k < 4; ++k) c.bias[k] = ef32x2{b[2 * k], b[2 * k + 1]};
    }
    if constexpr (EPI == MEMHIP_EPI_RESIDUAL) lds_read8(addr + 1024, c.g);
  };

  P8_READ_B0_FIRST();
  // K-tiles go in pairs (K % 128 == 0), so that buffer parity and the B register roles are static
#ifdef P8_STAMP
  int stamp_tile = 0;
  int stamp_it = 0;
  (void)stamp_it; (void)stamp_tile;
#endif
  for (int c = 0; c < total; c += 2) {
    if (c_k == 0) P8_STAMP_AT(0);
    iter_dma();
    {
      constexpr int bc = 0;
      P8_KTILE(bx, by, PX, PX, PX, PX);
      tm1 = tm2; tn1 = tn2; k1 = k2;
      advance2();
    }
    {
      constexpr int bc = 1;
      P8_KTILE(by, bx, PX, 0, 0, 0);
      tm1 = tm2; tn1 = tn2; k1 = k2;
      advance2();
    }

    P8_STAMP_IT();
    if (c_k == nk - 2) {
      // waves 0-3 wait for the last compute segment of waves 4-7, so that all eight waves run the
      // (VALU-bound, barrier-free) epilogue together; waves 4-7 fall half a phase behind again after it
      if (P8_REALIGN && wr == 0) P8_BARRIER();
      P8_STAMP_AT(1);
      // ---- epilogue of tile c_tile straight out of the accumulators: the MFMAs ran with swapped
      // operands, so a lane holds 4 consecutive columns (registers) of one row (lane & 15)
      const int tm = ctm, tn = ctn;
      // the lane's row / column inside the tile is recomputed here from a fresh lane id: taken from the kernel's entry
      // block these two lane constants are live across the whole main loop, hipcc spills them (the loop uses every
      // register), and the reload at the top of the epilogue comes with s_waitcnt vmcnt(0) -- the prefetch stream drained
      // (asm: hipcc knows that mbcnt is the lane id, folds it into the value it already has and spills that one)
      int elane;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
      const int mrow = tm * BMT + wr * (MF * 16) + (elane & 15);
      const int ncol = tn * BN + wc * 64 + (elane >> 4) * 8;     // + 32 for the second column half (B1)
      // four batches (j = column half, i = row half) of MF rows each.  The per-row operand loads (GELU input /
      // residual) of batch b + kAhead are issued before batch b is computed: with one batch in flight
      // a CU has 32 KB outstanding, i.e. ~20 GB/s per CU at ~1.5 us latency -- the epilogue was latency
      // bound, not HBM bound.  Row indices are clamped instead of branched, so that nothing orders the loads.
      // (the residual epilogue carries 8 registers per row: no room for a second batch beside 128 accumulators)
      constexpr int kAhead = (EPI == MEMHIP_EPI_RESIDUAL || EPI == MEMHIP_EPI_PATCH_EMBED) ? 0 : kEpiAhead;
      // residual epilogue: batch 0 alone; batches 1 and 2 go out together once batch 0 has released its
      // accumulators and row registers (the accumulators are re-zeroed after the loop, not inside it)
      constexpr bool kLate = (EPI == MEMHIP_EPI_RESIDUAL || EPI == MEMHIP_EPI_PATCH_EMBED) && P8_EPI_RESID_LATE;
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
#define P8_RESID_ROWS_AHEAD 2     // rows of residual loads in flight ahead of the row being computed.  3 / 4 measured (round 5): 14 / 21
        // The row loads are inline asm with HAND-COUNTED waits: left to hipcc this block's schedule is a matter of luck
        // (with the loads visible to it, one build ran the load-independent arithmetic of all rows first and consumed
        // each row's loads right after issuing them; another spilled and waited with vmcnt(0) per row).  hipcc never
        // waits for its own stores here (nothing depends on them) and does not see the asm loads, so it inserts no wait
        // at all; the only waits are the vmcnt(N) below, N = the vector-memory operations issued behind the row's loads:
        // per row section [3 loads of row r+2][arithmetic of row r][kStores stores of row r].
        constexpr int kResidAhead = P8_RESID_ROWS_AHEAD, kSlots = kResidAhead + 1;
        constexpr int kStores = 2 + (COPY ? 1 : 0);
        // FULL-LINE form (gemm_epilogue.hpp, pq_pack): a lane's 8 fp32 columns are 32 bytes of a 128-byte row segment; its
        // two 16-byte halves are exchanged with lane r ^ 8, so that the two loads (and the two stores) of a row-op cover
        // rows 0-7 [P] and 8-15 [Q] of the fragment in whole lines instead of 16 half-used lines each.  Memory side of
        // row-op r: rows row_mP(r) and + 8, four floats at column row_nP(r); arithmetic side: row row_m(r), 8 columns at row_n(r).
        const int er = elane & 15;
        const bool hi = (er & 8) != 0;
        auto row_m = [&](int r) { return mrow + ((r >> 2) & 1) * (BMT / 2) + (r & 3) * 16; };
        auto row_n = [&](int r) { return ncol + (r >> 3) * 32; };
        auto row_mP = [&](int r) { return row_m(r) - er + (er & 7); };
        auto row_nP = [&](int r) { return row_n(r) + (er >> 3) * 4; };
        const float* xbase = p.aux ? reinterpret_cast<const float*>(p.aux) : p.resid;
        const long long xld = p.aux ? p.ldaux : p.ldr;
        const bool per_sample = p.rowmask || p.sample_map;
        const float inv_rps = per_sample ? __frcp_rn((float)p.rows_per_sample) : 0.f;
        const float* rmb = p.rowmask ? p.rowmask : &g_epi_one;
        // work-skipping stochastic depth: the residual rows of the tile's compact samples come from the LDS copy of the
        // sample map that iteration 0 of this tile brought in (entries s0, s0 + 1, ...: zeros without a map)
        const int s0 = (tm * BMT + p.m_base) / p.rows_per_sample;
        const unsigned kid_lds = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem) +
                                 (unsigned)(G::kColsOff + tile_par * G::kColsSlot + 2048);
        f32x4 xa[kSlots], xb[kSlots];
        float rmv[kSlots];
        int rrP[kSlots], rrQ[kSlots];                           // residual rows of the slot's P / Q rows (32 bits: rows < 2^21)
        auto sample_of = [&](int mm) { return (int)(((float)mm + 0.5f) * inv_rps); };      // rows < 2^21 (p8_fits)
        auto resid_row = [&](int m) {                           // row m of this launch -> row of the residual stream
          const int mm = m + p.m_base;
          const int smp = sample_of(mm);
          // (one LDS word per lookup, read on the spot: more live values do not fit this epilogue's registers)
          int kd;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(kd) : "v"(kid_lds + (unsigned)((smp - s0) << 2)) : "memory");
          return p.sample_map ? kd * p.rows_per_sample + (mm - smp * p.rows_per_sample) : m;
        };
        auto issue_row = [&](int r, int slot) {
          const int mP = row_mP(r);
          rrP[slot] = resid_row(mP);
          rrQ[slot] = resid_row(mP + 8);
          const float* srcP = xbase + (long long)rrP[slot] * xld + row_nP(r);
          const float* srcQ = xbase + (long long)rrQ[slot] * xld + row_nP(r);
          const float* rsrc = rmb + (p.rowmask ? sample_of(row_m(r) + p.m_base) : 0);
          asm volatile("global_load_dwordx4 %0, %3, off nt\n\tglobal_load_dwordx4 %1, %4, off nt\n\tglobal_load_dword %2, %5, off"
                       : "=&v"(xa[slot]), "=&v"(xb[slot]), "=&v"(rmv[slot]) : "v"(srcP), "v"(srcQ), "v"(rsrc) : "memory");
        };
#pragma unroll
        for (int r = 0; r < kResidAhead; ++r) issue_row(r, r);
        float cs[8];
        EpiCols cols;
        auto do_row = [&](int r, int slot, auto waitc) {
          __builtin_amdgcn_sched_barrier(0);
          if ((r & 7) == 0) cols_from_lds(wc * 64 + (elane >> 4) * 8 + (r >> 3) * 32, cols);
          if (r + kResidAhead < 16) issue_row(r + kResidAhead, (slot + kResidAhead) % kSlots);
          const int q = ((r >> 2) & 1) * 2 + (r >> 3), mf = r & 3;
          float v[8];
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[nf * 4 + c] = ACC_EL(q, mf, nf, c);
          // the row's loads have landed (and its registers are named here, so nothing reads or reuses them earlier)
          asm volatile("s_waitcnt vmcnt(%3)" : "+v"(xa[slot]), "+v"(xb[slot]), "+v"(rmv[slot]) : "n"(decltype(waitc)::value) : "memory");
          EpiRow<EPI> row;
          {
            const unsigned P[4] = {__float_as_uint(xa[slot][0]), __float_as_uint(xa[slot][1]), __float_as_uint(xa[slot][2]), __float_as_uint(xa[slot][3])};
            const unsigned Q[4] = {__float_as_uint(xb[slot][0]), __float_as_uint(xb[slot][1]), __float_as_uint(xb[slot][2]), __float_as_uint(xb[slot][3])};
            unsigned lo[4], hh[4];
            pq_unpack(P, Q, hi, lo, hh);
#pragma unroll
            for (int c = 0; c < 4; ++c) { row.x[c] = __uint_as_float(lo[c]); row.x[4 + c] = __uint_as_float(hh[c]); }
          }
          row.rm = rmv[slot];
          row.row = 0;
          unsigned o[8], P[4], Q[4];
          epi8_math<EPI, COPY ? 2 : 0>(p, row_m(r), row_n(r), v, cs, cols, row, o);
          pq_pack(o, o + 4, hi, P, Q);
          float* dP = p.resid + (long long)rrP[slot] * p.ldr + row_nP(r);
          float* dQ = p.resid + (long long)rrQ[slot] * p.ldr + row_nP(r);
          *reinterpret_cast<float4*>(dP) = float4{__uint_as_float(P[0]), __uint_as_float(P[1]), __uint_as_float(P[2]), __uint_as_float(P[3])};
          *reinterpret_cast<float4*>(dQ) = float4{__uint_as_float(Q[0]), __uint_as_float(Q[1]), __uint_as_float(Q[2]), __uint_as_float(Q[3])};
        };
        // vector-memory operations issued behind L(r) when row r is consumed: the loads of the rows r + 1 .. r + kResidAhead
        // that exist (3 each) and the stores of the rows max(0, r - kResidAhead) .. r - 1 (kStores each)
        auto row_step = [&](auto R) {
          constexpr int r = decltype(R)::value;
          constexpr int loads = (r + kResidAhead < 16 ? kResidAhead : 15 - r), stores = r < kResidAhead ? r : kResidAhead;
          do_row(r, r % kSlots, std::integral_constant<int, 3 * loads + kStores * stores>{});
        };
        row_step(std::integral_constant<int, 0>{}); row_step(std::integral_constant<int, 1>{});
        row_step(std::integral_constant<int, 2>{}); row_step(std::integral_constant<int, 3>{});
        row_step(std::integral_constant<int, 4>{}); row_step(std::integral_constant<int, 5>{});
        row_step(std::integral_constant<int, 6>{}); row_step(std::integral_constant<int, 7>{});
        row_step(std::integral_constant<int, 8>{}); row_step(std::integral_constant<int, 9>{});
        row_step(std::integral_constant<int, 10>{}); row_step(std::integral_constant<int, 11>{});
        row_step(std::integral_constant<int, 12>{}); row_step(std::integral_constant<int, 13>{});
        row_step(std::integral_constant<int, 14>{}); row_step(std::integral_constant<int, 15>{});
      } else if constexpr (EPI == MEMHIP_EPI_BIAS_BF16 || EPI == MEMHIP_EPI_BIAS_GELU || EPI == MEMHIP_EPI_BIAS_GELU_DG ||
                           EPI == MEMHIP_EPI_DGELU || EPI == MEMHIP_EPI_MUL_AUX) {
        // bf16 outputs (and the bf16 second operand of GELU' / MUL_AUX) in FULL 128-byte lines: the two column halves of a
        // row fragment are computed together and exchanged between lanes r and r ^ 8 (pq_pack / pq_unpack,
        // gemm_epilogue.hpp): 2 instructions x 8 full lines instead of 2 x 16 half-used ones.  A unit = one 16-row fragment
        // (row half i, fragment mf), both column halves; the second-operand loads run kUnitsAhead units ahead.
        constexpr bool kEdge = GUARD;
        constexpr bool kNeedRows = EPI == MEMHIP_EPI_DGELU || EPI == MEMHIP_EPI_MUL_AUX;
        constexpr int NOUT = EpiPk<EPI>::W / 4;
        constexpr int kUnits = 2 * MF, kUnitsAhead = 3;
        const int er = elane & 15;
        const bool hi = (er & 8) != 0;
        const int nq = tn * BN + wc * 64 + (er >> 3) * 32 + (elane >> 4) * 8;         // memory side: 8 columns at nq,
        const int mq = tm * BMT + wr * (MF * 16) + (er & 7);                          // rows mq (+ unit) [P] and + 8 [Q]
        auto unit_mq = [&](int u) { return mq + (u / MF) * (BMT / 2) + (u % MF) * 16; };
        uint4 hP[kUnits], hQ[kUnits];
        auto load_unit = [&](int u) {
          if constexpr (kNeedRows) {
            int mP = unit_mq(u), mQ = mP + 8;
            if (kEdge) { mP = mP < p.M ? mP : p.M - 1; mQ = mQ < p.M ? mQ : p.M - 1; }
            const __bf16* ax = reinterpret_cast<const __bf16*>(p.aux);
            hP[u] = *reinterpret_cast<const uint4*>(ax + (long long)mP * p.ldaux + nq);
            hQ[u] = *reinterpret_cast<const uint4*>(ax + (long long)mQ * p.ldaux + nq);
          }
        };
#pragma unroll
        for (int u = 0; u < kUnitsAhead; ++u) load_unit(u);
        EpiCols cols[2];
        cols_from_lds(wc * 64 + (elane >> 4) * 8, cols[0]);
        cols_from_lds(wc * 64 + (elane >> 4) * 8 + 32, cols[1]);
        float cs[2][8];
        {
          float z;                                   // (an opaque zero: hipcc keeps a shared constant alive across the main loop)
          asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
          for (int r = 0; r < 8; ++r) { cs[0][r] = z; cs[1][r] = z; }
        }
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
          __builtin_amdgcn_sched_barrier(0);         // one unit at a time (register budget)
          if (u + kUnitsAhead < kUnits) load_unit(u + kUnitsAhead);
          const int i = u / MF, mf = u % MF;
          const int m = mrow + i * (BMT / 2) + mf * 16;
          EpiRow<EPI> ra, rb;
          if constexpr (kNeedRows) {
            const unsigned P[4] = {hP[u].x, hP[u].y, hP[u].z, hP[u].w}, Q[4] = {hQ[u].x, hQ[u].y, hQ[u].z, hQ[u].w};
            unsigned a[4], b[4];
            pq_unpack(P, Q, hi, a, b);
            ra.h = uint4{a[0], a[1], a[2], a[3]};
            rb.h = uint4{b[0], b[1], b[2], b[3]};
          }
          float v0[8], v1[8];
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[nf * 4 + r] = ACC_EL(i * 2, mf, nf, r); v1[nf * 4 + r] = ACC_EL(i * 2 + 1, mf, nf, r); }
          unsigned o0[4 * NOUT], o1[4 * NOUT];
          if constexpr (kEdge) {                     // rows past M (clamped loads) are computed, never stored, and not summed
            float c0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            epi8_math<EPI, 0>(p, m, ncol, v0, c0, cols[0], ra, o0);
            epi8_math<EPI, 0>(p, m, ncol + 32, v1, c1, cols[1], rb, o1);
            const bool in = m < p.M;
#pragma unroll
            for (int k = 0; k < 8; ++k) { cs[0][k] += in ? c0[k] : 0.f; cs[1][k] += in ? c1[k] : 0.f; }
          } else {
            epi8_math<EPI, 0>(p, m, ncol, v0, cs[0], cols[0], ra, o0);
            epi8_math<EPI, 0>(p, m, ncol + 32, v1, cs[1], cols[1], rb, o1);
          }
          const int mP = unit_mq(u);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            unsigned P[4], Q[4];
            pq_pack(o0 + 4 * o, o1 + 4 * o, hi, P, Q);
            void* base = o == 0 ? p.out0 : p.out1;
            const long long ld = o == 0 ? p.ldo0 : p.ldo1;
            if (!kEdge || mP < p.M) st_stream16(base, (long long)mP * ld + nq, P[0], P[1], P[2], P[3]);
            if (!kEdge || mP + 8 < p.M) st_stream16(base, (long long)(mP + 8) * ld + nq, Q[0], Q[1], Q[2], Q[3]);
          }
        }
        colsum_flush16(p, ncol, cs[0], elane);
        colsum_flush16(p, ncol + 32, cs[1], elane);
      } else {
        constexpr bool kEdge = GUARD;
        EpiRow<EPI> rows[4][MF];
        auto load_batch = [&](int b) {
          const int jj = b >> 1, ii = b & 1;
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const int m = mrow + ii * (BMT / 2) + mf * 16;
            epi_row_load<EPI>(p, kEdge ? (m < p.M ? m : p.M - 1) : m, ncol + jj * 32, rows[b][mf]);
          }
        };
#pragma unroll
        for (int b = 0; b < kAhead && b < 4; ++b) load_batch(b);
        float cs[8];
        EpiCols cols;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = b >> 1, i = b & 1;
          const int n = ncol + j * 32;
          // (without the row branches the whole epilogue is one scheduling region: keep the loads of later batches from
          // being hoisted over this batch -- the row registers are budgeted per batch)
          __builtin_amdgcn_sched_barrier(0);
          if (i == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) cs[r] = 0.f;
            cols_from_lds(wc * 64 + (elane >> 4) * 8 + j * 32, cols);
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
              for (int r = 0; r < 4; ++r) v[nf * 4 + r] = ACC_EL(i * 2 + j, mf, nf, r);
            if (!kEdge || m < p.M) epilogue8<EPI, COPY ? 2 : 0>(p, m, n, v, cs, cols, rows[b][mf]);
            if (!kEdge) __builtin_amdgcn_sched_barrier(0);     // one row at a time (register budget)
          }
          if (i == 1) colsum_flush16(p, n, cs, elane);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) ACC_EL(q, mf, nf, r) = 0.f;
      P8_STAMP_AT(2);
#ifdef P8_STAMP
      ++stamp_tile;
#endif
      c_k = 0;
      c_tile += nblk;
      if (c_tile < ntiles) decode(c_tile, ctm, ctn);
      tile_par ^= 1;
      if (P8_REALIGN && wr == 1) P8_BARRIER();
    } else {
      c_k += 2;
    }
  }
  if (wr == 0) P8_BARRIER();                               // balances the last stagger barrier of waves 4-7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup
}

template <int EPI, int BMT, bool GUARD, bool COPY>
__global__ __launch_bounds__(kThreads) void gemm_p8_kernel(GemmArgs p, int ntm, int ntn, int stagger, int prefetch) {
  p8_body<EPI, BMT, GUARD, COPY>(p, ntm, ntn, stagger, prefetch, (int)blockIdx.x, (int)gridDim.x);
}

// ONE launch for a product whose last round of 256-row tiles would be poorly filled (N = 768: 591 tiles on 256 CUs): workgroups
// 0 .. nmain-1 are the persistent 256-row workgroups over the rows of the full rounds (`head`), the remaining workgroups run the
// 128-row form over the left-over rows (`tail`).  Both kinds ask for the same LDS (one workgroup per CU), so the dispatcher
// starts a tail workgroup on every CU the moment its 256-row workgroup retires: no second launch, no chip-wide drain between
// the full rounds and the ragged one (round 5; the two-launch form stays behind MEMHIP option gemm_p8_pair = 0).
template <int EPI, bool GUARD_T, bool COPY>
__global__ __launch_bounds__(kThreads) void gemm_p8_pair_kernel(GemmArgs head, GemmArgs tail, int ntm_h, int ntm_t, int ntn, int nmain,
                                                                int stagger, int prefetch) {
  if ((int)blockIdx.x < nmain)
    p8_body<EPI, 256, false, COPY>(head, ntm_h, ntn, stagger, prefetch, (int)blockIdx.x, nmain);
  else
    p8_body<EPI, 128, GUARD_T, COPY>(tail, ntm_t, ntn, 0, prefetch, (int)blockIdx.x - nmain, (int)gridDim.x - nmain);
}

template <int EPI, int BMT, bool GUARD, bool COPY>
int launch_p8gc(const GemmArgs& p, hipStream_t s, int num_cu) {
  const int ntm = (p.M + BMT - 1) / BMT, ntn = p.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8_kernel<EPI, BMT, GUARD, COPY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, P8Geo<BMT>::kLdsAll);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_p8: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int grid = ntm * ntn < num_cu ? ntm * ntn : num_cu;
  hipLaunchKernelGGL((gemm_p8_kernel<EPI, BMT, GUARD, COPY>), dim3(grid), dim3(kThreads), P8Geo<BMT>::kLdsAll, s, p, ntm, ntn,
                     BMT == 256 ? opt(OPT_GEMM_STAGGER) : 0, opt(OPT_GEMM_PREFETCH));
  return check_launch("gemm_bf16_nt(p8)");
}

template <int EPI, int BMT, bool GUARD>
int launch_p8g(const GemmArgs& p, hipStream_t s, int num_cu) {
  if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    if (!p.out0) return launch_p8gc<EPI, BMT, GUARD, false>(p, s, num_cu);
    // (the bf16 copy of the branch output beside the full-line residual epilogue of the 256-row kernel spills; no engine
    // asks for it on the bf16 path: those calls run on the 128-row form)
    if constexpr (BMT == 256) return MEMHIP_EUNSUPPORTED;
    else return launch_p8gc<EPI, BMT, GUARD, true>(p, s, num_cu);
  } else {
    return launch_p8gc<EPI, BMT, GUARD, true>(p, s, num_cu);
  }
}

template <int EPI, int BMT>
int launch_p8(const GemmArgs& p, hipStream_t s, int num_cu) {
  if constexpr (BMT == 256) return launch_p8g<EPI, 256, false>(p, s, num_cu);
  else return p.M % 128 == 0 ? launch_p8g<EPI, 128, false>(p, s, num_cu) : launch_p8g<EPI, 128, true>(p, s, num_cu);
}

}  // namespace

#ifdef P8_STAMP
extern "C" int memhip_debug_p8_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_p8_stamps), sizeof(unsigned long long) * 256 * 32) == hipSuccess ? 0 : -1;
}
extern "C" int memhip_debug_p8_stamps_rt(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_p8_stamps_rt), sizeof(unsigned long long) * 256 * 32) == hipSuccess ? 0 : -1;
}
#endif

namespace memhip {

static int p8_num_cu(hipStream_t s) { return usable_cus(s); }

static bool p8_fits(const GemmArgs& p) {
  const bool vec = ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux | p.colscale_n) & 7) == 0;      // host twin of vec_ok()
  // Narrow outputs (N = 768: 591 tiles on 256 CUs) are taken too: the launcher hands the rows of a
  // poorly filled last round to the 128x128 kernel (gemm_p8_split_rows).  MEMHIP_GEMM_P8_MIN_N overrides.
  const int min_n = opt(OPT_GEMM_P8_MIN_N);
  const bool wide = p.N >= min_n;
  // (the residual epilogue of the 256-row kernel finds a row's sample without an integer division: rows below 2^21)
  // (a 256-row tile touches at most four samples of a sample map: rows_per_sample >= 86)
  const bool rows_ok = (!(p.rowmask || p.sample_map) || (long long)p.M + p.m_base < (1 << 21)) &&
                       (!p.sample_map || p.rows_per_sample >= 86);
  return p.M >= 4096 && wide && p.N % BN == 0 && p.K % (2 * BK) == 0 && vec && rows_ok;
}

// Rows the persistent launch should take when its last round of tiles would be less than half full
// (0 = take everything): full rounds only, the caller runs the remaining rows on finer tiles.
int gemm_p8_split_rows(const GemmArgs& p, hipStream_t s) {
  if (!p8_fits(p)) return 0;
  const int num_cu = p8_num_cu(s);
  if (!num_cu) return 0;
  const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
  const int tiles = ntm * ntn, rounds = tiles / num_cu, rem = tiles % num_cu;
  const int whole = (p.M / BM) * BM;                       // the 256-row kernel takes whole tiles only
  if (rounds < 1 || rem == 0 || rem * 2 > num_cu) return whole == p.M ? 0 : whole;
  return (rounds * num_cu / ntn) * BM;
}

// Returns MEMHIP_EUNSUPPORTED when the shape does not fit this structure (caller falls back).
int gemm_p8_half_dispatch(const GemmArgs& p, hipStream_t s);
int gemm_p8_dispatch(const GemmArgs& p, hipStream_t s) {
  if (!p8_fits(p) || p.M % BM != 0) return MEMHIP_EUNSUPPORTED;   // whole 256-row tiles only (no row guard in the epilogue)
  const int num_cu = p8_num_cu(s);
  if (!num_cu) return MEMHIP_EUNSUPPORTED;
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch_p8<MEMHIP_EPI_BIAS_BF16, 256>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch_p8<MEMHIP_EPI_BIAS_GELU, 256>(p, s, num_cu);
    case MEMHIP_EPI_RESIDUAL:
      if (p.out0) return gemm_p8_half_dispatch(p, s);
      return launch_p8<MEMHIP_EPI_RESIDUAL, 256>(p, s, num_cu);
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
  const int num_cu = p8_num_cu(s);
  if (!num_cu) return MEMHIP_EUNSUPPORTED;
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch_p8<MEMHIP_EPI_BIAS_BF16, 128>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch_p8<MEMHIP_EPI_BIAS_GELU, 128>(p, s, num_cu);
    case MEMHIP_EPI_RESIDUAL: return launch_p8<MEMHIP_EPI_RESIDUAL, 128>(p, s, num_cu);
    case MEMHIP_EPI_DGELU: return launch_p8<MEMHIP_EPI_DGELU, 128>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU_DG: return launch_p8<MEMHIP_EPI_BIAS_GELU_DG, 128>(p, s, num_cu);
    case MEMHIP_EPI_MUL_AUX: return launch_p8<MEMHIP_EPI_MUL_AUX, 128>(p, s, num_cu);
    case MEMHIP_EPI_F32: return launch_p8<MEMHIP_EPI_F32, 128>(p, s, num_cu);
    default: return MEMHIP_EUNSUPPORTED;
  }
}


template <int EPI, bool GUARD_T>
static int launch_pair(const GemmArgs& head, const GemmArgs& tail, hipStream_t s, int num_cu) {
  constexpr bool COPY = EPI != MEMHIP_EPI_RESIDUAL;
  const int ntn = head.N / BN, ntm_h = head.M / BM, ntm_t = (tail.M + 127) / 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8_pair_kernel<EPI, GUARD_T, COPY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, P8Geo<256>::kLdsAll);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_p8(pair): set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int nmain = ntm_h * ntn < num_cu ? ntm_h * ntn : num_cu;
  const int ntail = ntm_t * ntn < num_cu ? ntm_t * ntn : num_cu;
  hipLaunchKernelGGL((gemm_p8_pair_kernel<EPI, GUARD_T, COPY>), dim3(nmain + ntail), dim3(kThreads), P8Geo<256>::kLdsAll, s, head, tail,
                     ntm_h, ntm_t, ntn, nmain, opt(OPT_GEMM_STAGGER), opt(OPT_GEMM_PREFETCH));
  return check_launch("gemm_bf16_nt(p8 pair)");
}

// head: whole 256-row tiles of the full rounds; tail: the left-over rows (any count >= 128).  MEMHIP_EUNSUPPORTED when this
// epilogue / shape has no paired form (the caller launches the two kernels one after the other).
int gemm_p8_pair_dispatch(const GemmArgs& head, const GemmArgs& tail, hipStream_t s) {
  if (!p8_fits(head) || head.M % BM != 0 || tail.M < 128) return MEMHIP_EUNSUPPORTED;
  if (head.epilogue == MEMHIP_EPI_RESIDUAL && head.out0) return MEMHIP_EUNSUPPORTED;     // (the COPY form lives on the 128-row kernel only)
  const int num_cu = p8_num_cu(s);
  if (!num_cu || num_cu % 8 != 0) return MEMHIP_EUNSUPPORTED;
  const bool g = tail.M % 128 != 0;
#define P8_PAIR(E) return g ? launch_pair<E, true>(head, tail, s, num_cu) : launch_pair<E, false>(head, tail, s, num_cu)
  switch (head.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: P8_PAIR(MEMHIP_EPI_BIAS_BF16);
    case MEMHIP_EPI_RESIDUAL: P8_PAIR(MEMHIP_EPI_RESIDUAL);
    case MEMHIP_EPI_BIAS_GELU_DG: P8_PAIR(MEMHIP_EPI_BIAS_GELU_DG);
    case MEMHIP_EPI_MUL_AUX: P8_PAIR(MEMHIP_EPI_MUL_AUX);
    default: return MEMHIP_EUNSUPPORTED;
  }
#undef P8_PAIR
}

}  // namespace memhip
