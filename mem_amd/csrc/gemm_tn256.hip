// 256x256-tile variant of the weight-gradient GEMM  out[N,K] += sum_r A[r,N] * B[r,K]  (see
// gemm_tn.hip for the contract).  Same reasons as gemm256.hip: half the operand bytes per flop
// through L2 and 12 (transposing) fragment reads per 32 MFMAs.
//
// One 512-thread workgroup = one (output tile, slice of the token rows): tiles * splits ~ #CUs.
// The token rows advance in stages of 32 rows x (256 + 256) columns = 32 KiB through a 4-slot LDS
// ring (LDS-DMA, stage s+3 issued while stage s is computed, counted vmcnt + raw barrier).  The LDS
// image is token-major (512-byte rows) with the 32-byte segment index XOR-ed with (row & 7); MFMA
// operands are read down the columns with ds_read_b64_tr_b16 using the k-slot permutation of
// gemm_tn.hip.  The fp32 result is added into the (pre-zeroed / accumulating) gradient with
// atomics shaped as 256 contiguous bytes per wave-instruction (staged through LDS).
#include "common.h"

namespace {

using namespace memhip;

constexpr int BM = 256, BN = 256, BR = 32;
constexpr int kThreads = 512;
constexpr int kTile = BR * 256 * 2;            // 16 KiB per operand
constexpr int kStage = 2 * kTile;              // 32 KiB
constexpr int kSlots = 4;
constexpr int kLoads = kStage / 16 / kThreads; // 4

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __attribute__((aligned(256))) unsigned char g_zero512[512];   // zero-initialised

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 tr_pair(const char* a0) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 512));
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lo;
  u.s.h = hi;
  return u.v;
}

__global__ __launch_bounds__(kThreads) void gemm_tn256_kernel(const __bf16* __restrict__ A, long long lda,
                                                              const __bf16* __restrict__ B, long long ldb,
                                                              int R, int N, int K, float* __restrict__ out,
                                                              long long ldo, int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;        // 2 (n) x 4 (k) waves, 128 x 64 outputs each
  const int ntk = K / BN;
  const int tiles = (N / BM) * ntk;
  const int tile = blockIdx.x % tiles, sp = blockIdx.x / tiles;
  const int n0 = (tile / ntk) * BM, k0 = (tile % ntk) * BN;
  const int rbeg = sp * rows_per_split;
  int rend = rbeg + rows_per_split;
  rend = rend < R ? rend : R;
  if (rbeg >= rend) return;
  const int total = (rend - rbeg + BR - 1) / BR;

  // ---- LDS-DMA issue constants: instruction j of this wave = 2 token rows (2 x 512 B)
  const int cpos = lane & 31;                                  // 16-byte chunk position inside the row
  auto issue = [&](int s) {
    char* slot = smem + (s % kSlots) * kStage;
    const int r0 = rbeg + s * BR;
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const int inst = wave * kLoads + j;                      // 0..31 ; 0..15 -> A, 16..31 -> B
      const bool isA = inst < 16;
      const int row = (inst & 15) * 2 + (lane >> 5);
      const int chunk = ((((cpos >> 1) ^ (row & 7)) << 1) | (cpos & 1));   // global 16-B chunk landing here
      const int gr = r0 + row;
      const __bf16* src = isA ? A + (long long)gr * lda + n0 + chunk * 8 : B + (long long)gr * ldb + k0 + chunk * 8;
      const void* g = gr < rend ? (const void*)src : (const void*)(g_zero512 + cpos * 16);
      glds16(g, slot + inst * 1024);
    }
  };
  issue(0);
  if (total > 1) issue(1);
  if (total > 2) issue(2);

  // ---- per-lane fragment addresses: block rows 4g+q (and +16), 8 bytes at column 4p of the 16-col block
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int row0 = 4 * g + q, r7 = row0 & 7;
  int a_off[8], b_off[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a_off[i] = row0 * 512 + ((((wr * 8 + i) ^ r7)) << 5) + pp * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) b_off[j] = kTile + row0 * 512 + ((((wc * 4 + j) ^ r7)) << 5) + pp * 8;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < total; ++s) {
    const int ahead = total - 1 - s;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kLoads) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 3 < total) issue(s + 3);
    const char* base = smem + (s % kSlots) * kStage;
    bf16x8 af[8], bfr[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = tr_pair(base + a_off[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = tr_pair(base + b_off[j]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  // ---- epilogue: fp32 atomics, 16 output rows x 64 columns per pass through a wave-private 4 KiB
  __builtin_amdgcn_s_barrier();                    // all waves are done reading the ring
  float* wreg = reinterpret_cast<float*>(smem + wave * 4096);
  const int nw = n0 + wr * 128, kw = k0 + wc * 64;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) wreg[((lane >> 4) * 4 + r) * 64 + ((j * 16 + (lane & 15)) ^ ((lane >> 4) << 4))] = acc[i][j][r];
#pragma unroll
    for (int row = 0; row < 16; ++row) {
      const float v = wreg[row * 64 + (lane ^ ((row >> 2) << 4))];
      atomicAdd(out + (long long)(nw + i * 16 + row) * ldo + kw + lane, v);
    }
  }
}

}  // namespace

namespace memhip {

// MEMHIP_EUNSUPPORTED when the shape does not fit (caller falls back to the 128x128 kernel).
int gemm_tn256_dispatch(const void* A, long long lda, const void* B, long long ldb, int R, int N, int K, float* out,
                        long long ldo, int accumulate, hipStream_t s) {
  if (N % BM != 0 || K % BN != 0 || R < 2048) return MEMHIP_EUNSUPPORTED;
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MEMHIP_EUNSUPPORTED;
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int tiles = (N / BM) * (K / BN);
  const int stages = cdiv(R, BR);
  int splits = num_cu / tiles;
  if (splits < 1) splits = 1;
  if (splits > stages / 8) splits = stages / 8 > 0 ? stages / 8 : 1;     // >= 256 rows per split
  const int rows_per_split = cdiv(stages, splits) * BR;
  splits = cdiv(R, rows_per_split);
  if (!accumulate) {
    hipError_t e = hipMemset2DAsync(out, (size_t)ldo * sizeof(float), 0, (size_t)K * sizeof(float), (size_t)N, s);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_tn256: memset: %s", hipGetErrorString(e));
  }
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * kStage);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_tn256: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn256_kernel, dim3(tiles * splits), dim3(kThreads), kSlots * kStage, s, (const __bf16*)A, lda,
                     (const __bf16*)B, ldb, R, N, K, out, ldo, rows_per_split);
  return check_launch("gemm_bf16_tn(256)");
}

}  // namespace memhip
