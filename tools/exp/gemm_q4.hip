// EXPERIMENT (not part of libmemhip.so): the four-wave form of the persistent 256x256x64 bf16 NT GEMM.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC tools/exp/gemm_q4.hip -o mem_amd/exp/gemm_q4.so
//   python tools/exp/q4_check.py
// Why: DESIGN.md section 9 -- gemm_p8 (eight waves, 128x64 wave tiles) moves 192 KB of fragments out of the LDS per K-tile and sits
// at the chip's power limit; four waves with 128x128 wave tiles (one per SIMD, 256 accumulator registers each, operands in the other
// half of the 512-register file) read 128 KB, need two barriers per K-tile instead of eight, and have no partner wave to hand the
// matrix core over to.  Same LDS images as gemm_p8 (16 KiB half-tiles of 128 rows x 64 k, 16-byte chunk c of row r at c ^ ((r >> 1) & 7)),
// two 64 KiB buffers; a K-tile's four half-tiles are restaged as soon as every wave has read its kh = 1 fragments (middle of the
// K-tile), so a piece is in flight for 1.5 K-tiles.  Per K-tile and wave: 128 MFMAs (16x16x32), 32 ds_read_b128, 16 LDS-DMA pieces.
// Epilogue of this experiment: bias + bf16, 8-byte stores straight from the accumulators (M % 256 == 0, N % 256 == 0, K % 128 == 0).
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kHalf = 128 * 128;          // bytes of a half-tile image
constexpr int kBuf = 4 * kHalf;           // A0 A1 B0 B1
constexpr int kGroupM = 8;

__device__ __forceinline__ int key_a(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
#define Q4_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define Q4_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define Q4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define Q4_BARRIER()                   \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)

__global__ __launch_bounds__(256) void q4_kernel(const __bf16* __restrict__ A, long long lda, const __bf16* __restrict__ B,
                                                 long long ldb, __bf16* __restrict__ out, long long ldo,
                                                 const float* __restrict__ bias, int M, int N, int K, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = K / BK;
  const int ntiles = ntm * ntn;
  const int per_xcd = (gridDim.x + 7) / 8;
  const int first = (gridDim.x % 8 == 0) ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  if (total <= 0) return;
  const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem);

  // staging: this wave moves pieces 4 wave .. 4 wave + 3 (8 rows x 128 B each) of every half-tile
  unsigned offA[4], offB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (4 * wave + j) * 8 + (lane >> 3);
    const unsigned ch = (unsigned)(((lane & 7) ^ key_a(row)) * 16);
    offA[j] = (unsigned)((long long)row * lda * 2) + ch;
    offB[j] = (unsigned)((long long)row * ldb * 2) + ch;
  }
  auto decode = [&](int id, int& tm, int& tn) {
    const int gsz = kGroupM * ntn;
    const int grp = id / gsz, rem = id - grp * gsz;
    const int rows = ntm - grp * kGroupM < kGroupM ? ntm - grp * kGroupM : kGroupM;
    tn = rem / rows;
    tm = grp * kGroupM + (rem - tn * rows);
  };
  // piece i = 0..15 of K-tile (tm, tn, kt) into buffer buf: half-tile i >> 2, piece j = i & 3 of this wave
  auto stage_piece = [&](int i, int buf, int tm, int tn, int kt) {
    const int H = i >> 2, j = i & 3;
    const unsigned slot = lds0 + buf * kBuf + H * kHalf + wave * 4096 + j * 1024;
    if (H < 2) {
      const char* base = reinterpret_cast<const char*>(A) + ((long long)(tm * BM + H * 128) * lda + kt * BK) * 2;
      glds16s(base, offA[j], slot);
    } else {
      const char* base = reinterpret_cast<const char*>(B) + ((long long)(tn * BN + (H - 2) * 128) * ldb + kt * BK) * 2;
      glds16s(base, offB[j], slot);
    }
  };
  int g2 = 0, id2 = first, k2 = 0, tm2, tn2;                 // cursor of the prefetch stream
  decode(id2, tm2, tn2);
  auto advance2 = [&]() {
    if (g2 + 1 < total) {
      ++g2;
      if (++k2 == nk) { k2 = 0; id2 += gridDim.x; decode(id2, tm2, tn2); }
    }
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) stage_piece(i, 0, tm2, tn2, k2);
  advance2();
#pragma unroll
  for (int i = 0; i < 16; ++i) stage_piece(i, 1, tm2, tn2, k2);
  advance2();

  // fragment addresses: row 16 x + (lane & 15), chunk 4 kh + (lane >> 4)
  const int sw = (lane >> 1) & 7;
  const unsigned roff = (unsigned)((lane & 15) * 128 + (((lane >> 4) ^ sw) << 4));
  const unsigned rdA = lds0 + wr * kHalf + roff;             // + buf * kBuf + mf * 2048, ^ 64 for kh = 1
  const unsigned rdB = lds0 + (2 + wc) * kHalf + roff;
  auto lds128 = [&](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(addr); };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a0[8], b0[8], a1[8], b1[8];

  Q4_WAIT_VM(16);
  Q4_BARRIER();
#pragma unroll
  for (int x = 0; x < 8; ++x) { a0[x] = lds128(rdA + x * 2048); b0[x] = lds128(rdB + x * 2048); }

  int c_tile = first;
// (inline asm with the accumulator TIED in an AGPR tuple: through the builtin hipcc does not update the 256 accumulator registers in
// place -- dst != srcC, copies to VGPRs behind s_nop 7, spills.  An accumulator tile is touched once per 64 MFMAs, so no dependent
// MFMA is ever close enough to need wait states the compiler no longer inserts.)
#define Q4_MFMA(mf, nf, as, bs) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mf][nf]) : "v"(bs[nf]), "v"(as[mf]))
  // one K-tile in buffer `bc` (compile-time): see the header for the order
#define Q4_KTILE(bc)                                                                                                  \
  do {                                                                                                                \
    /* phase 1: kh = 0 out of set 0; the kh = 1 fragments of this K-tile arrive in set 1 */                          \
    _Pragma("unroll") for (int mf = 0; mf < 8; ++mf) {                                                                \
      _Pragma("unroll") for (int nf = 0; nf < 8; ++nf) Q4_MFMA(mf, nf, a0, b0);                                       \
      a1[mf] = lds128((rdA ^ 64u) + (bc) * kBuf + mf * 2048);                                                         \
      b1[mf] = lds128((rdB ^ 64u) + (bc) * kBuf + mf * 2048);                                                         \
      Q4_FENCE();                                                                                                     \
    }                                                                                                                 \
    Q4_WAIT_LGKM0();                                                                                                  \
    Q4_BARRIER();                               /* every wave has read all it needs of buffer bc */                  \
    /* phase 2: kh = 1 out of set 1; K-tile g + 2 is staged into buffer bc ... */                                     \
    _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                                                                \
      _Pragma("unroll") for (int nf = 0; nf < 8; ++nf) {                                                              \
        Q4_MFMA(mf, nf, a1, b1);                                                                                      \
        if (nf & 1) { Q4_FENCE(); stage_piece(mf * 4 + (nf >> 1), bc, tm2, tn2, k2); Q4_FENCE(); }                    \
      }                                                                                                               \
    }                                                                                                                 \
    advance2();                                                                                                       \
    Q4_WAIT_VM(16);                             /* this wave's pieces of K-tile g + 1 have landed */                 \
    Q4_BARRIER();                                                                                                     \
    /* ... and the kh = 0 fragments of K-tile g + 1 arrive in set 0 from the other buffer */                          \
    _Pragma("unroll") for (int mf = 4; mf < 8; ++mf) {                                                                \
      _Pragma("unroll") for (int nf = 0; nf < 8; ++nf) {                                                              \
        Q4_MFMA(mf, nf, a1, b1);                                                                                      \
        if ((nf & 3) == 3) {                                                                                          \
          const int x = (mf - 4) * 2 + (nf >> 2);                                                                     \
          a0[x] = lds128(rdA + ((bc) ^ 1) * kBuf + x * 2048);                                                         \
          b0[x] = lds128(rdB + ((bc) ^ 1) * kBuf + x * 2048);                                                         \
          Q4_FENCE();                                                                                                 \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)

  auto epilogue = [&]() {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last MFMAs have left the pipe before the accumulators are read
    int tm, tn;
    decode(c_tile, tm, tn);
    const int r0 = tm * BM + wr * 128 + (lane & 15);
    const int n0 = tn * BN + wc * 128 + 4 * (lane >> 4);
#pragma unroll
    for (int mf = 0; mf < 8; ++mf)
#pragma unroll
      for (int nf = 0; nf < 8; ++nf) {
        const int n = n0 + nf * 16;
        f32x4 v = acc[mf][nf];
        if (bias) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
          v += bb;
        }
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4*>(out + (long long)(r0 + mf * 16) * ldo + n) = w;
        acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  for (int t = 0; t < my_tiles; ++t) {
    for (int c = 0; c < nk; c += 2) {
      Q4_KTILE(0);
      Q4_KTILE(1);
    }
    epilogue();
    c_tile += gridDim.x;
  }
}
}  // namespace

extern "C" int q4_gemm(const void* A, int64_t lda, const void* B, int64_t ldb, void* out, int64_t ldo, const float* bias, int M,
                       int N, int K, void* stream) {
  if (M % BM || N % BN || K % (2 * BK)) return -1;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(q4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBuf) != hipSuccess)
      return -2;
    done = true;
  }
  const int ntm = M / BM, ntn = N / BN;
  int grid = ntm * ntn < 256 ? ntm * ntn : 256;
  hipLaunchKernelGGL(q4_kernel, dim3(grid), dim3(256), 2 * kBuf, (hipStream_t)stream, (const __bf16*)A, (long long)lda,
                     (const __bf16*)B, (long long)ldb, (__bf16*)out, (long long)ldo, bias, M, N, K, ntm, ntn);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
