// Phase-interleaved 256x256 weight-gradient GEMM  out[N,K] += sum_r A[r,N] * B[r,K]  (contract of
// gemm_tn.hip).  The structure of gemm_p8.hip applied to the token-major operands:
//
//   * one workgroup = one (256x256 output tile, slice of the token rows); a K-TILE here is 64 token
//     rows, staged as four 16 KiB half-tiles {A0, A1, B0, B1} of [64 tokens][128 columns] (256-byte
//     rows, the 32-byte segment index XOR-ed with (row & 7) on the LDS-DMA source address);
//   * fragments are read down the columns with ds_read_b64_tr_b16 (two reads = one 8-element MFMA
//     operand, k slots permuted identically for A and B, see gemm_tn.hip);
//   * four phases per K-tile = four 128x128 quadrants, quadrant order (A0,B0) (A0,B1) (A1,B1)
//     (A1,B0), phase 4 pre-reads B0 of the next K-tile; waves 4-7 run half a phase behind waves
//     0-3, so that on every SIMD one wave's 16 MFMAs cover the other's LDS reads and LDS-DMA issue;
//   * one half-tile of prefetch per phase (A1(c+1), B0(c+2), A0(c+2), B1(c+2)), five in flight,
//     s_waitcnt vmcnt(10) before the first barrier of every phase; hazards as in gemm_p8.hip.
// The fp32 partial tile is either added to the gradient with atomics shaped as two 128-byte runs
// per wave-instruction, or (with a caller workspace) stored to a slab and summed by a reduction pass;
// both are staged through the idle ring.
#include <atomic>
#include "common.h"

#define P8_PRIO_MODE 0   // 0: s_setprio 1 around every MFMA block; 1: none; 2: none + waves 4-7 at priority 1 for the whole kernel
                         // dgrad kernels of the other stream, no difference: 38.28 vs 38.35 ms.  gemm_p8.hip: no difference either way)
#define P8_PRIO(x) __builtin_amdgcn_s_setprio(x)

namespace {

using namespace memhip;

constexpr int BM = 256, BN = 256, BR = 64;
constexpr int kThreads = 512;
constexpr int kHalf = BR * 128 * 2;     // 16 KiB: 64 tokens x 128 columns
constexpr int kBuf = 4 * kHalf;         // A0 A1 B0 B1
constexpr int kRing = 2 * kBuf;         // 128 KiB
enum { HA0 = 0, HA1 = 1, HB0 = 2, HB1 = 3 };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __attribute__((aligned(256))) unsigned char g_tnp8_zero[256];   // zero-initialised
#ifdef P8_STAMP
__device__ unsigned long long g_tnp8_stamps[256 * 4];
#endif

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
// Transposing LDS read as inline asm.  With the builtin, hipcc (ROCm 7.2) cannot tell the read from
// the in-flight LDS-DMA writes and puts s_waitcnt vmcnt(0) in front of every group of reads -- the
// whole prefetch stream drained four times per K-tile (measured: 0.86 -> 1.1 PFLOP/s without it).
// The asm form is invisible to the waitcnt pass, so the consumer waits explicitly (TP_MFMA).
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16_b64(unsigned lds_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
  return r;
}
// rows r and r+16 of one 16-column block
template <int OFF>
__device__ __forceinline__ bf16x8 tr_pair(unsigned lds_addr) {
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lds_tr16_b64<OFF>(lds_addr);
  u.s.h = lds_tr16_b64<OFF + 16 * 256>(lds_addr);
  return u.v;
}

#define TP_WAIT_VM() asm volatile("s_waitcnt vmcnt(10)" ::: "memory")
#define TP_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// WS = false: the partial tile is added to `out` with fp32 atomics.  WS = true: it is written with
// plain 16-byte stores to slab `sp` of a workspace [splits][N][K] (ldo = K), summed afterwards by
// tn_reduce_kernel: 64 MB of atomics per GEMM run at ~1.3 TB/s, plain stores at 5-6 TB/s.
// Workgroups b, b+8, ... share an XCD (and its L2).  All tiles of one token slice read the same
// rows of A and B, so consecutive ids of the (slice, tile) space go to ONE XCD: bijective remap of
// blockIdx (measured before: 1.13 GB fetched per launch for 0.39 GB of operands, HBM-bound).
__device__ __forceinline__ int tn_xcd_wid() {
  const int gq = (int)gridDim.x / 8, gr = (int)gridDim.x % 8, xcd = (int)blockIdx.x % 8;
  return (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (int)blockIdx.x / 8;
}

// one workgroup: id `wid` of the (slice, tile) space of ONE product
template <bool WS>
__device__ __forceinline__ void tn_p8_body(const __bf16* __restrict__ A, long long lda, const __bf16* __restrict__ B,
                                           long long ldb, int R, int N, int K, float* __restrict__ out, long long ldo,
                                           int rows_per_split, int wid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntk = K / BN;
  const int tiles = (N / BM) * ntk;
  const int tile = wid % tiles, sp = wid / tiles;
  const int n0 = (tile / ntk) * BM, k0 = (tile % ntk) * BN;
  const int rbeg = sp * rows_per_split;
  int rend = rbeg + rows_per_split;
  rend = rend < R ? rend : R;
  if (rbeg >= rend) return;
  const int total = ((rend - rbeg + 2 * BR - 1) / (2 * BR)) * 2;     // K-tiles, rounded up to a pair (zero rows)

  // ---- LDS-DMA issue constants: a half-tile is 16 pieces of 4 token rows x 256 B; this wave moves
  // pieces 2*wave and 2*wave+1
  int prow[2];
  unsigned offA[2], offB[2];
  const int cpos = lane & 15;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    prow[j] = (wave * 2 + j) * 4 + (lane >> 4);
    const unsigned chunk = (unsigned)((((cpos >> 1) ^ (prow[j] & 7)) << 1) | (cpos & 1));
    offA[j] = (unsigned)((long long)prow[j] * lda * 2) + chunk * 16;
    offB[j] = (unsigned)((long long)prow[j] * ldb * 2) + chunk * 16;
  }
  const char* zsrc = reinterpret_cast<const char*>(g_tnp8_zero) + cpos * 16;
  auto stage = [&](int H, int buf, int g) {                 // K-tile g of this workgroup's slice
    char* slot = smem + buf * kBuf + H * kHalf + wave * 2048;
    const int r0 = rbeg + g * BR;
    const bool isA = H == HA0 || H == HA1;
    const int c0 = (isA ? n0 : k0) + ((H == HA1 || H == HB1) ? 128 : 0);
    const char* base = reinterpret_cast<const char*>(isA ? A : B) + ((long long)r0 * (isA ? lda : ldb) + c0) * 2;
    if (r0 + BR <= rend) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (P8_SADDR) glds16s(base, isA ? offA[j] : offB[j], slot + j * 1024);
        else glds16(base + (isA ? offA[j] : offB[j]), slot + j * 1024);
      }
    } else {                                                // tail of the slice: rows >= rend add zeros
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* src = (r0 + prow[j] < rend) ? base + (isA ? offA[j] : offB[j]) : zsrc;
        glds16(src, slot + j * 1024);
      }
    }
  };
  // prologue: K-tile 0 entirely, B0 A0 B1 of K-tile 1 (K-tiles >= total are all-zero rows)
  stage(HB0, 0, 0); stage(HA0, 0, 0); stage(HB1, 0, 0); stage(HA1, 0, 0);
  stage(HB0, 1, 1); stage(HA0, 1, 1); stage(HB1, 1, 1);
  TP_WAIT_VM();
  TP_BARRIER();
  if (wr == 1) TP_BARRIER();                                // waves 4-7 run half a phase behind

  // ---- transposing fragment reads: this lane addresses 4 columns (p) of token row 4g+q (+16)
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int row0 = 4 * g4 + q4, r7 = row0 & 7;
  const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem);
  unsigned aoff[4], boff[2];                     // LDS byte addresses
#pragma unroll
  for (int f = 0; f < 4; ++f) aoff[f] = lds0 + row0 * 256 + (((wr * 4 + f) ^ r7) << 5) + pp * 8;
#pragma unroll
  for (int f = 0; f < 2; ++f) boff[f] = lds0 + 2 * kHalf + row0 * 256 + (((wc * 2 + f) ^ r7) << 5) + pp * 8;

  f32x4 acc[4][4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define TP_ACC_EL(q, nf, kf, r) acc[q][nf][kf][r]
  bf16x8 a[4][2], bx[2][2], by[2][2];

#define TP_READ_A(half)                                                                                   \
  _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) {                                                      \
    a[nf][0] = tr_pair<(half) * kHalf>(aoff[nf] + bo);                                                    \
    a[nf][1] = tr_pair<(half) * kHalf + 8192>(aoff[nf] + bo);                                             \
  }
#define TP_READ_B(dst, boff_, half)                                                                       \
  _Pragma("unroll") for (int kf = 0; kf < 2; ++kf) {                                                      \
    dst[kf][0] = tr_pair<(half) * kHalf>(boff[kf] + (boff_));                                             \
    dst[kf][1] = tr_pair<(half) * kHalf + 8192>(boff[kf] + (boff_));                                      \
  }
#define TP_MFMA_BODY(q, bsrc)                                                                             \
    _Pragma("unroll") for (int rh = 0; rh < 2; ++rh) _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)     \
        _Pragma("unroll") for (int kf = 0; kf < 2; ++kf) acc[q][nf][kf] =                                 \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nf][rh], bsrc[kf][rh], acc[q][nf][kf], 0, 0, 0)
#define TP_MFMA(q, bsrc)                                                                                  \
  do {                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the asm reads are not tracked by the compiler */ \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    P8_PRIO(1);                                                                        \
    TP_MFMA_BODY(q, bsrc);                                                                                \
    P8_PRIO(0);                                                                        \
  } while (0)
#define TP_KTILE(bq0, bq1)                                                                                \
  do {                                                                                                    \
    const int bo = bc * kBuf;                                                                             \
    /* phase 1: quadrant (A0, B0) */                                                                      \
    TP_READ_A(0);                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    stage(HA1, bc ^ 1, c + bc + 1);                                                                       \
    TP_WAIT_VM();                                                                                         \
    TP_BARRIER();                                                                                         \
    TP_MFMA(0, bq0);                                                                                      \
    TP_BARRIER();                                                                                         \
    /* phase 2: quadrant (A0, B1) */                                                                      \
    TP_READ_B(bq1, bo, 1);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    stage(HB0, bc, c + bc + 2);                                                                           \
    TP_WAIT_VM();                                                                                         \
    TP_BARRIER();                                                                                         \
    TP_MFMA(1, bq1);                                                                                      \
    TP_BARRIER();                                                                                         \
    /* phase 3: quadrant (A1, B1) */                                                                      \
    TP_READ_A(1);                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    stage(HA0, bc, c + bc + 2);                                                                           \
    TP_WAIT_VM();                                                                                         \
    TP_BARRIER();                                                                                         \
    TP_MFMA(3, bq1);                                                                                      \
    TP_BARRIER();                                                                                         \
    /* phase 4: quadrant (A1, B0); B0 of the next K-tile comes from the other buffer */                   \
    TP_READ_B(bq1, (bc ^ 1) * kBuf, 0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    stage(HB1, bc, c + bc + 2);                                                                           \
    TP_WAIT_VM();                                                                                         \
    TP_BARRIER();                                                                                         \
    TP_MFMA(2, bq0);                                                                                      \
    TP_BARRIER();                                                                                         \
  } while (0)

  TP_READ_B(bx, 0, 0);
#ifdef P8_STAMP
  // diagnostic build (tools/build_variant.sh stamp -DP8_STAMP): wave 0 stamps s_memtime / s_memrealtime around its main
  // loop; the quotient is the shader clock the chip holds inside this kernel (tools/clock_probe.py)
  if (wave == 0) {
    unsigned long long t_, r_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
    if (lane == 0) { g_tnp8_stamps[(blockIdx.x & 255) * 4 + 0] = t_; g_tnp8_stamps[(blockIdx.x & 255) * 4 + 1] = r_; }
  }
#endif
  for (int c = 0; c < total; c += 2) {
    {
      constexpr int bc = 0;
      TP_KTILE(bx, by);
    }
    {
      constexpr int bc = 1;
      TP_KTILE(by, bx);
    }
  }
#ifdef P8_STAMP
  if (wave == 0) {
    unsigned long long t_, r_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
    if (lane == 0) { g_tnp8_stamps[(blockIdx.x & 255) * 4 + 2] = t_; g_tnp8_stamps[(blockIdx.x & 255) * 4 + 3] = r_; }
  }
#endif
  if (wr == 0) TP_BARRIER();                               // balances the stagger barrier of waves 4-7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the idle tail of the prefetch stream has landed
  TP_BARRIER();                                            // every wave is done with the ring

  // ---- epilogue: fp32 atomics.  C layout of a 16x16 tile: col (k) = lane & 15, row (n) =
  // 4 * (lane >> 4) + reg.  A pass moves 16 n-rows x 64 k (the wave's 32 columns in each B half)
  // through this wave's 4 KiB, then adds row by row: one wave-instruction = two 128-byte runs.
  float* wreg = reinterpret_cast<float*>(smem + wave * 4096);
  const int kcol = k0 + (lane >> 5) * 128 + wc * 32 + (lane & 31);
  float* dst = WS ? out + (long long)sp * N * ldo : out;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) * 4 + r;
            wreg[row * 64 + ((j * 32 + kf * 16 + (lane & 15)) ^ (((row >> 2) & 1) << 4))] = TP_ACC_EL(i * 2 + j, nf, kf, r);
          }
      const int nrow = n0 + i * 128 + wr * 64 + nf * 16;
      if constexpr (WS) {
        // 4 rows per instruction, 16 bytes per lane: lanes 0-7 the 32 columns of B half 0, 8-15 of half 1
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 4 + (lane >> 4), c4 = (lane & 15) * 4;
          // plain (cached) store: the reduction pass reads the slab right away, from L2 / Infinity Cache
          *reinterpret_cast<float4*>(dst + (long long)(nrow + row) * ldo + k0 + (c4 >> 5) * 128 + wc * 32 + (c4 & 31)) =
              *reinterpret_cast<const float4*>(wreg + row * 64 + (c4 ^ (((row >> 2) & 1) << 4)));
        }
      } else {
#pragma unroll
        for (int row = 0; row < 16; ++row) {
          const float v = wreg[row * 64 + (lane ^ (((row >> 2) & 1) << 4))];
          atomicAdd(dst + (long long)(nrow + row) * ldo + kcol, v);
        }
      }
    }
}

template <bool WS>
__global__ __launch_bounds__(kThreads) void gemm_tn_p8_kernel(const __bf16* __restrict__ A, long long lda,
                                                              const __bf16* __restrict__ B, long long ldb,
                                                              int R, int N, int K, float* __restrict__ out,
                                                              long long ldo, int rows_per_split) {
  tn_p8_body<WS>(A, lda, B, ldb, R, N, K, out, ldo, rows_per_split, tn_xcd_wid());
}

// ---- GROUPED launch (round 5): the weight gradients of up to four Linear layers whose operands are ready at the same time
// (fc2 + fc1, proj + qkv of a block) as ONE grid with ONE common split count.  Alone, a 768 x 768 gradient has 9 tiles and
// needs 28 row slices to fill 256 CUs (28 slabs of 2.4 MB for a 2.4 MB result, 1 800 rows of main loop per workgroup against
// the same 256 KB write-out); beside the 27 tiles of the qkv gradient both run with 7 slices -- the shape of the fc1 launch.
// Workgroup ids are laid out product after product, (slice, tile) inside a product, so an XCD still works on consecutive
// (slice, tile) ids of one product.
constexpr int kTnGroupMax = 4;
struct TnGroupProblem {
  const __bf16* A;
  const __bf16* B;
  float* ws;                 // this product's slabs [splits][N][K]
  float* out;                // the gradient (reduction pass)
  long long lda, ldb, ldo;
  int R, N, K;
  int rows_per_split, splits;
  int wg_begin;              // first workgroup id of the product
  int quad_begin;            // first float4 of the product in the reduction pass
};
struct TnGroup {
  TnGroupProblem p[kTnGroupMax];
  int count;
};
__global__ __launch_bounds__(kThreads) void gemm_tn_p8_group_kernel(const TnGroup g) {
  const int wid = tn_xcd_wid();
  int i = 0;
#pragma unroll
  for (int j = 1; j < kTnGroupMax; ++j)
    if (j < g.count && wid >= g.p[j].wg_begin) i = j;
  const TnGroupProblem& q = g.p[i];
  tn_p8_body<true>(q.A, q.lda, q.B, q.ldb, q.R, q.N, q.K, q.ws, (long long)q.K, q.rows_per_split, wid - q.wg_begin);
}
__global__ __launch_bounds__(256) void tn_reduce_group_kernel(const TnGroup g, int accumulate) {
  const int quad = blockIdx.x * 256 + threadIdx.x;
  int i = 0;
#pragma unroll
  for (int j = 1; j < kTnGroupMax; ++j)
    if (j < g.count && quad >= g.p[j].quad_begin) i = j;
  const TnGroupProblem& q = g.p[i];
  const long long i4 = (long long)(quad - q.quad_begin) * 4;
  const long long slab = (long long)q.N * q.K;
  if (i4 >= slab) return;
  float4 a = *reinterpret_cast<const float4*>(q.ws + i4);
  for (int s = 1; s < q.splits; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(q.ws + s * slab + i4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  const long long n = i4 / q.K, k = i4 - n * q.K;
  float4* o = reinterpret_cast<float4*>(q.out + n * q.ldo + k);
  if (accumulate) {
    const float4 c = *o;
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
  }
  *o = a;
}

// out[n][k] (+)= sum over the S slabs of the workspace
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int S, long long slab, int N, int K,
                                                        float* __restrict__ out, long long ldo, int accumulate) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= (long long)N * K) return;
  float4 a = *reinterpret_cast<const float4*>(ws + i4);
  for (int s = 1; s < S; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(ws + s * slab + i4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  const long long n = i4 / K, k = i4 - n * K;
  float4* o = reinterpret_cast<float4*>(out + n * ldo + k);
  if (accumulate) {
    const float4 c = *o;
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
  }
  *o = a;
}

}  // namespace

#ifdef P8_STAMP
extern "C" int memhip_debug_tnp8_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tnp8_stamps), sizeof(unsigned long long) * 256 * 4) == hipSuccess ? 0 : -1;
}
#endif

namespace memhip {

static void tn_p8_plan(int R, int N, int K, int num_cu, int& tiles, int& splits, int& rows_per_split) {
  tiles = (N / BM) * (K / BN);
  const int pairs = cdiv(R, 2 * BR);                       // the token rows advance in pairs of K-tiles
  splits = num_cu / tiles;
  if (splits < 1) splits = 1;
  if (splits > pairs / 2) splits = pairs / 2 > 0 ? pairs / 2 : 1;        // >= 256 rows per split
  rows_per_split = cdiv(pairs, splits) * 2 * BR;
  splits = cdiv(R, rows_per_split);
}
static int tn_p8_num_cu(hipStream_t s = nullptr) { return usable_cus(s); }

// bytes of workspace with which the partial tiles go through plain stores + a reduction pass
// (sized for EVERY CU: a launch stream without a reservation plans the most splits, whatever another stream reserved)
size_t gemm_tn_p8_workspace(int R, int N, int K) {
  if (N % BM != 0 || K % BN != 0 || R < 2048) return 0;
  const int num_cu = max_cus();
  if (!num_cu) return 0;
  size_t need = 0;
  for (int cu = num_cu; cu >= 8; cu -= 8) {                   // the split count is not monotone in the CU count: take the maximum
    int tiles, splits, rps;
    tn_p8_plan(R, N, K, cu, tiles, splits, rps);
    const size_t b = splits > 1 ? (size_t)splits * N * K * sizeof(float) : 0;
    need = b > need ? b : need;
  }
  return need;
}

// MEMHIP_EUNSUPPORTED when the shape does not fit (caller falls back to the other TN kernels).
int gemm_tn_p8_dispatch(const void* A, long long lda, const void* B, long long ldb, int R, int N, int K, float* out,
                        long long ldo, int accumulate, float* ws, size_t ws_bytes, hipStream_t s) {
  if (N % BM != 0 || K % BN != 0 || R < 2048) return MEMHIP_EUNSUPPORTED;
  const int num_cu = tn_p8_num_cu(s);
  if (!num_cu) return MEMHIP_EUNSUPPORTED;
  int tiles, splits, rows_per_split;
  tn_p8_plan(R, N, K, num_cu, tiles, splits, rows_per_split);
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_p8_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_p8_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_tn_p8: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const size_t need = (size_t)splits * N * K * sizeof(float);
  // (the reduction pass reads and writes `out` as float4: the pointer itself must be 16-byte aligned, not only ldo -- a caller's
  // 4-byte-aligned gradient view takes the atomic path below)
  if (ws && splits > 1 && ws_bytes >= need && ((uintptr_t)ws & 15) == 0 && ldo % 4 == 0 && ((uintptr_t)out & 15) == 0) {
    hipLaunchKernelGGL(gemm_tn_p8_kernel<true>, dim3(tiles * splits), dim3(kThreads), kRing, s, (const __bf16*)A, lda,
                       (const __bf16*)B, ldb, R, N, K, ws, (long long)K, rows_per_split);
    const long long quads = (long long)N * K / 4;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, ws, splits,
                       (long long)N * K, N, K, out, ldo, accumulate);
    return check_launch("gemm_bf16_tn(p8, workspace)");
  }
  if (!accumulate) {
    hipError_t e = hipMemset2DAsync(out, (size_t)ldo * sizeof(float), 0, (size_t)K * sizeof(float), (size_t)N, s);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_tn_p8: memset: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(gemm_tn_p8_kernel<false>, dim3(tiles * splits), dim3(kThreads), kRing, s, (const __bf16*)A, lda,
                     (const __bf16*)B, ldb, R, N, K, out, ldo, rows_per_split);
  return check_launch("gemm_bf16_tn(p8)");
}

// ---- grouped launch: plan.  One split count for the whole group: the smallest number of rounds (grids of num_cu workgroups)
// that keeps >= 80 % of the CUs busy, else the best of four.  (fc2 + fc1 of ViT-B, 72 tiles: ONE round of 216 workgroups with 3 row
// slices each -- 55 MB of slabs -- instead of two rounds with 7: in the step 34.35-34.39 ms against 34.59-34.83, and against
// 34.41-34.53 with fc2 / fc1 as single launches; the CUs such a round leaves idle take workgroups of the other stream.)
static bool tn_group_plan(const memhip_tn_problem_t* pr, int count, int num_cu, TnGroup& g, int& total_wgs, int& total_quads,
                          size_t& ws_floats) {
  if (count < 2 || count > kTnGroupMax || !num_cu) return false;
  int tiles_total = 0;
  for (int i = 0; i < count; ++i) {
    if (pr[i].N % BM != 0 || pr[i].K % BN != 0 || pr[i].R < 2048 || pr[i].ldo % 4 != 0 || ((uintptr_t)pr[i].out & 15) != 0) return false;
    tiles_total += (pr[i].N / BM) * (pr[i].K / BN);
  }
  int best_s = 0;
  double best_eff = 0.0;
  for (int r = 1; r <= 4; ++r) {
    const int sp = (r * num_cu) / tiles_total;
    if (sp < 2) continue;
    const double eff = (double)tiles_total * sp / ((double)r * num_cu);
    if (eff > best_eff + 1e-9) { best_eff = eff; best_s = sp; }
    if (eff >= 0.80) break;
  }
  if (best_s < 2) return false;
  total_wgs = 0; total_quads = 0; ws_floats = 0;
  g.count = count;
  for (int i = 0; i < count; ++i) {
    TnGroupProblem& q = g.p[i];
    const int tiles = (pr[i].N / BM) * (pr[i].K / BN);
    const int pairs = cdiv(pr[i].R, 2 * BR);
    int sp = best_s;
    if (sp > pairs / 2) sp = pairs / 2 > 0 ? pairs / 2 : 1;
    q.rows_per_split = cdiv(pairs, sp) * 2 * BR;
    q.splits = cdiv(pr[i].R, q.rows_per_split);
    q.A = (const __bf16*)pr[i].A; q.B = (const __bf16*)pr[i].B;
    q.lda = pr[i].lda; q.ldb = pr[i].ldb; q.ldo = pr[i].ldo;
    q.R = pr[i].R; q.N = pr[i].N; q.K = pr[i].K;
    q.out = pr[i].out;
    q.ws = nullptr;
    q.wg_begin = total_wgs;
    q.quad_begin = total_quads;
    total_wgs += tiles * q.splits;
    total_quads += (int)(((long long)pr[i].N * pr[i].K / 4 + 255) / 256 * 256);      // whole blocks per product
    ws_floats += (size_t)q.splits * pr[i].N * pr[i].K;
  }
  return true;
}

size_t gemm_tn_p8_group_workspace(const memhip_tn_problem_t* pr, int count) {
  const int num_cu = max_cus();
  size_t need = 0;
  for (int cu = num_cu; cu >= 8; cu -= 8) {
    TnGroup g;
    int wgs, quads;
    size_t fl;
    if (tn_group_plan(pr, count, cu, g, wgs, quads, fl) && fl * sizeof(float) > need) need = fl * sizeof(float);
  }
  return need;
}

// MEMHIP_EUNSUPPORTED: the caller runs the products one by one.
int gemm_tn_p8_group_dispatch(const memhip_tn_problem_t* pr, int count, int accumulate, float* ws, size_t ws_bytes, hipStream_t s) {
  TnGroup g;
  int wgs, quads;
  size_t fl;
  if (!ws || ((uintptr_t)ws & 15) != 0 || !tn_group_plan(pr, count, tn_p8_num_cu(s), g, wgs, quads, fl) ||
      fl * sizeof(float) > ws_bytes)
    return MEMHIP_EUNSUPPORTED;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_p8_group_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_tn_p8(group): set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  float* w = ws;
  for (int i = 0; i < count; ++i) {
    g.p[i].ws = w;
    w += (size_t)g.p[i].splits * g.p[i].N * g.p[i].K;
  }
  hipLaunchKernelGGL(gemm_tn_p8_group_kernel, dim3(wgs), dim3(kThreads), kRing, s, g);
  hipLaunchKernelGGL(tn_reduce_group_kernel, dim3((unsigned)(quads / 256)), dim3(256), 0, s, g, accumulate);
  return check_launch("gemm_bf16_tn_group(p8)");
}

}  // namespace memhip
