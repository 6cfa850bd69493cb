// 256x256-tile persistent variant of the bf16 NT GEMM (same contract and epilogues as gemm.hip).
//
// Measured on MI355X (tools/gemm_probe.py): with every operand cache-resident the 128x128 kernel
// runs 1.4 PFLOP/s, with real operands 0.6-0.9 -- the 128x128 tile moves 393 KB of operands per
// 25 MFLOP through the XCD L2 (~85 GB/s per CU at 1.4 PF, above what a CU can pull from L2).  A
// 256x256 tile halves the operand bytes per flop, and a 128x64 sub-tile per wave needs 12
// ds_read_b128 per 32 MFMAs instead of 16.
//
// Structure: one 512-thread workgroup per CU walks 256x256 output tiles in XCD-contiguous order
// (persistent).  K advances in 32-deep stages of 32 KiB (A 256x32 + B 256x32, 64-byte LDS rows)
// through a 4-slot LDS ring: the LDS-DMA loads of stage s+3 are issued while stage s is computed;
// counted s_waitcnt vmcnt(8) + raw s_barrier keep two stages in flight across the barrier, also
// across tile boundaries.  8 waves as 2(M) x 4(N); fragment addresses are per-lane constants plus
// immediates (no VALU address math in the loop).  The epilogue transposes through the ring slot
// that is free at that moment (refill deferred), 16 rows per pass, 16-byte global accesses.
#include "common.h"
#include "gemm_epilogue.hpp"

namespace {

using namespace memhip;

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int kThreads = 512;
constexpr int kATile = BM * BK * 2;            // 16 KiB
constexpr int kStage = 2 * kATile;             // 32 KiB
constexpr int kSlots = 4;
constexpr int kLoads = kStage / 16 / kThreads; // 4 LDS-DMA instructions per thread per stage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 64-byte rows (4 chunks of 16 B): chunk c of row r lives at position c ^ perm[(r >> 2) & 3];
// with perm = {0,3,2,1} every 16-lane ds_read_b128 group touches 16 distinct 16-byte bank slots.
__device__ __forceinline__ int rperm(int row) { return (4 - ((row >> 2) & 3)) & 3; }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(kThreads) void gemm256_kernel(GemmArgs p, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = p.K / BK;
  const int ntiles = ntm * ntn;
  // tile order: in every round of gridDim.x tiles each XCD (workgroups b, b+8, ...) takes a
  // contiguous run of tile ids (n fastest) => its L2 sees whole A row panels
  const int per_xcd = (gridDim.x + 7) / 8;
  const int first = (gridDim.x % 8 == 0) ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  if (total <= 0) return;

  // ---- per-lane constants of the LDS-DMA issue: instruction j of this wave covers 16 rows
  unsigned goff[kLoads];
  int grow_[kLoads];
#pragma unroll
  for (int j = 0; j < kLoads; ++j) {
    const int inst = wave * kLoads + j;                    // 0..31 ; 0..15 -> A, 16..31 -> B
    const int row = (inst & 15) * 16 + (lane >> 2);        // row inside its 256-row tile
    const int chunk = (lane & 3) ^ rperm(row);
    grow_[j] = row;
    goff[j] = (unsigned)((long long)row * (inst < 16 ? p.lda : p.ldb) * 2 + chunk * 16);
  }
  int h_tile = first, h_k = 0;
  auto issue = [&](int s) {
    const int tm = h_tile / ntn, tn = h_tile - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN, k0 = h_k * BK;
    char* slot = smem + (s % kSlots) * kStage;
    const bool edge = tm == ntm - 1;
    const char* abase = reinterpret_cast<const char*>(p.A) + ((long long)m0 * p.lda + k0) * 2;
    const char* bbase = reinterpret_cast<const char*>(p.B) + ((long long)n0 * p.ldb + k0) * 2;
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const int inst = wave * kLoads + j;
      const bool isA = inst < 16;
      unsigned off = goff[j];
      if (edge && isA) {                                   // last M tile: clamp rows to M-1
        int gr = m0 + grow_[j];
        gr = gr < p.M ? gr : p.M - 1;
        off = (unsigned)((long long)(gr - m0) * p.lda * 2) + (goff[j] - (unsigned)((long long)grow_[j] * p.lda * 2));
      }
      glds16((isA ? abase : bbase) + off, slot + inst * 1024);
    }
    if (++h_k == nk) { h_k = 0; h_tile += gridDim.x; }
  };
  issue(0);
  if (total > 1) issue(1);
  if (total > 2) issue(2);

  // ---- per-lane fragment read offset inside a tile (frag base rows are multiples of 16)
  const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ rperm(lane & 15)) << 4);
  const char* a_lane = smem + frag_off + wr * 128 * 64;
  const char* b_lane = smem + kATile + frag_off + wc * 64 * 64;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int c_tile = first, c_k = 0;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < total; ++s) {
    // stage s has landed (stages s+1, s+2 may be in flight); every wave is done with stage s-1
    const int ahead = total - 1 - s;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kLoads) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool last_k = (c_k == nk - 1);
    if (!last_k && s + 3 < total) issue(s + 3);
    const int so = (s % kSlots) * kStage;
    bf16x8 af[8], bfr[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_lane + so + i * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(b_lane + so + j * 1024);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (last_k) {
      // ---- epilogue of tile c_tile through the free ring slot (s+3) % 4: 4 KiB per wave = 16 rows
      // x 64 fp32 columns per pass; column block XOR-ed with bit 2 of the row so the writes are 2-way
      const int tm = c_tile / ntn, tn = c_tile - tm * ntn;
      const int mw = tm * BM + wr * 128, nw = tn * BN + wc * 64;
      float* wreg = reinterpret_cast<float*>(smem + ((s + 3) % kSlots) * kStage + wave * 4096);
      EpiCols cols;
      epi_cols_load<EPI>(p, nw + (lane & 7) * 8, cols);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) * 4 + r;
            wreg[row * 64 + ((j * 16 + (lane & 15)) ^ (((row >> 2) & 1) << 4))] = acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
          const int m = mw + i * 16 + row;
          float v[8];
          ld8(wreg + row * 64 + (c8 ^ (((row >> 2) & 1) << 4)), v);
          if (m < p.M) epilogue8<EPI>(p, m, nw + c8, v, cs, cols);
        }
      }
      colsum_flush(p, nw + (lane & 7) * 8, cs, lane);
      c_k = 0;
      c_tile += gridDim.x;
      if (s + 3 < total) {
        __builtin_amdgcn_s_barrier();            // every wave is out of its staging region
        issue(s + 3);                            // the deferred refill of that slot
      }
    } else {
      ++c_k;
    }
  }
}

template <int EPI>
int launch256(const GemmArgs& p, hipStream_t s, int num_cu) {
  const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * kStage);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm256: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int grid = ntm * ntn < num_cu ? ntm * ntn : num_cu;
  hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(grid), dim3(kThreads), kSlots * kStage, s, p, ntm, ntn);
  return check_launch("gemm_bf16_nt(256)");
}

}  // namespace

namespace memhip {

// Returns MEMHIP_EUNSUPPORTED when the shape does not fit this structure (caller falls back).
int gemm256_dispatch(const GemmArgs& p, hipStream_t s) {
  const bool vec = ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux | p.colscale_n) & 7) == 0;      // host twin of vec_ok()
  // N = 768 (3 tiles wide) leaves the third round of 591 tiles 31 % full on 256 CUs and measures
  // 5-10 % below the 128x128 kernel (tools/bench_gemm.py); wide N gains 12-25 %.  MEMHIP_GEMM256_MIN_N
  // overrides the threshold for experiments.
  const int min_n = opt(OPT_GEMM256_MIN_N);
  if (p.M < 4096 || p.N < min_n || p.N % BN != 0 || p.K % BK != 0 || !vec) return MEMHIP_EUNSUPPORTED;
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MEMHIP_EUNSUPPORTED;
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch256<MEMHIP_EPI_BIAS_BF16>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch256<MEMHIP_EPI_BIAS_GELU>(p, s, num_cu);
    case MEMHIP_EPI_RESIDUAL: return launch256<MEMHIP_EPI_RESIDUAL>(p, s, num_cu);
    case MEMHIP_EPI_DGELU: return launch256<MEMHIP_EPI_DGELU>(p, s, num_cu);
    case MEMHIP_EPI_F32: return launch256<MEMHIP_EPI_F32>(p, s, num_cu);
    default: return MEMHIP_EUNSUPPORTED;
  }
}

}  // namespace memhip
