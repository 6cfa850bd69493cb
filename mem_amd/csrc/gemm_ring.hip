// Persistent, ring-pipelined variant of the bf16 NT GEMM (same contract and epilogues as gemm.hip)
// for the large token-dimension products of the ViT:  C[M,N] = A[M,K] * B[N,K]^T.
//
// Why a second structure: the 128x128 / 2-workgroups-per-CU kernel drains its LDS-DMA (vmcnt(0))
// in front of every barrier and runs all co-resident workgroups in lock-step, so at K = 768 the
// prologue and the epilogue of every tile are exposed (~25 % of the time) and the main loop tops
// out near 900 TFLOP/s.  Here
//   * one 512-thread workgroup per CU walks a list of 256x128 output tiles (persistent);
//   * the (tile, k-step) sequence is ONE stream of 48-KiB stages through a 3-slot LDS ring: the
//     loads of stage s+2 are issued before stage s is computed and waited for with a COUNTED
//     s_waitcnt vmcnt(6) + raw s_barrier, so one stage is always in flight across the barrier and
//     the first stages of the next tile are already landing while this tile's epilogue runs;
//   * the epilogue transposes through the ring slot that is free at that moment (its refill is
//     deferred until after the epilogue), giving 16-byte global accesses as in gemm.hip.
// 8 waves as 4(M) x 2(N), 64x64 per wave, v_mfma_f32_16x16x32_bf16, XOR-swizzled LDS image via the
// LDS-DMA source address (identical fragment code to gemm.hip).
#include "common.h"
#include "gemm_epilogue.hpp"

namespace {

using namespace memhip;

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kThreads = 512;
constexpr int kATile = BM * BK * 2;            // 32 KiB
constexpr int kBTile = BN * BK * 2;            // 16 KiB
constexpr int kStage = kATile + kBTile;        // 48 KiB
constexpr int kSlots = 3;
constexpr int kLoadsPerStage = kStage / 16 / kThreads;   // 6 LDS-DMA instructions per thread per stage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ int swz_slot(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// one stage = A rows [m0, m0+256) and B rows [n0, n0+128), k in [k0, k0+64): 48 wave-instructions.
// Per-lane byte offsets (row * ld + swizzled chunk) are fixed for the whole kernel; a stage adds one
// wave-uniform (scalar) base per operand, so an issue is 6 x {s_add, global_load_lds saddr+voffset}.
struct StageOffs { unsigned off[kLoadsPerStage]; int row[kLoadsPerStage]; };

__device__ __forceinline__ StageOffs stage_offs(const GemmArgs& p, int wave, int lane) {
  StageOffs o;
#pragma unroll
  for (int j = 0; j < kLoadsPerStage; ++j) {
    const int inst = wave * kLoadsPerStage + j;           // 0..47 ; 0..31 -> A, 32..47 -> B
    const int trow = inst * 8 + (lane >> 3);
    const bool isA = inst < 32;
    const int row = isA ? trow : trow - 256;              // row inside its tile (swizzle uses this)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    o.row[j] = row;
    o.off[j] = (unsigned)((long long)row * (isA ? p.lda : p.ldb) * 2 + chunk * 16);
  }
  return o;
}

template <int EPI>
__global__ __launch_bounds__(kThreads) void gemm_ring_kernel(GemmArgs p, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = p.K / BK;
  const int ntiles = ntm * ntn;
  // tiles of this workgroup: blockIdx.x, + gridDim.x, ...  (tile id -> (m tile, n tile), n fastest
  // so that the workgroups running at the same time share A row panels in L2)

  // tile order: in every round of gridDim.x tiles each XCD (workgroups b, b+8, ...) takes a
  // contiguous run of tile ids => its L2 sees whole A row panels (n is the fast tile index)
  const int nxcd = 8;
  const int per_xcd = (gridDim.x + nxcd - 1) / nxcd;
  const int slot_in_round = (gridDim.x % nxcd == 0) ? ((int)blockIdx.x % nxcd) * per_xcd + (int)blockIdx.x / nxcd
                                                     : (int)blockIdx.x;
  const StageOffs so = stage_offs(p, wave, lane);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int my_tiles = (ntiles - slot_in_round + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;                          // stages of this workgroup
  if (total <= 0) return;
  // half-stage issue (3 of the 6 LDS-DMA instructions per thread)
  int h_tile = slot_in_round, h_k = 0;                      // issue cursor (half-stage granularity)
  auto issue_half = [&](int s, int half) {
    const int tm = h_tile / ntn, tn = h_tile - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN, k0 = h_k * BK;
    char* slot = smem + (s % kSlots) * kStage;
    const bool edge = tm == ntm - 1;
    const char* abase = reinterpret_cast<const char*>(p.A) + ((long long)m0 * p.lda + k0) * 2;
    const char* bbase = reinterpret_cast<const char*>(p.B) + ((long long)n0 * p.ldb + k0) * 2;
#pragma unroll
    for (int jj = 0; jj < kLoadsPerStage / 2; ++jj) {
      const int j = half * (kLoadsPerStage / 2) + jj;
      const int inst = wave * kLoadsPerStage + j;
      const bool isA = inst < 32;
      unsigned off = so.off[j];
      if (edge && isA) {
        int grow = m0 + so.row[j];
        grow = grow < p.M ? grow : p.M - 1;
        off = (unsigned)((long long)(grow - m0) * p.lda * 2) + (so.off[j] - (unsigned)((long long)so.row[j] * p.lda * 2));
      }
      glds16((isA ? abase : bbase) + off, slot + inst * 1024);
    }
    if (half == 1 && ++h_k == nk) { h_k = 0; h_tile += gridDim.x; }
  };
  issue_half(0, 0); issue_half(0, 1);
  if (total > 1) { issue_half(1, 0); issue_half(1, 1); }

  int c_tile = slot_in_round, c_k = 0;                      // compute cursor
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < total; ++s) {
    // stage s has landed (only stage s+1 may still be in flight) and every wave is done with s-1
    if (s + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerStage) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool last_k = (c_k == nk - 1);
    if (!last_k && s + 2 < total) { issue_half(s + 2, 0); issue_half(s + 2, 1); }
    const char* At = smem + (s % kSlots) * kStage;
    const char* Bt = At + kATile;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
      const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(At + swz_slot(wr * 64 + i * 16 + (lane & 15), chunk) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bt + swz_slot(wc * 64 + j * 16 + (lane & 15), chunk) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (last_k) {
      // ---- epilogue of tile c_tile through the free ring slot (s+2) % 3, wave-private 6 KiB
      const int tm = c_tile / ntn, tn = c_tile - tm * ntn;
      const int mw = tm * BM + wr * 64, nw = tn * BN + wc * 64;
      constexpr int LS = 72;
      float* wreg = reinterpret_cast<float*>(smem + ((s + 2) % kSlots) * kStage + wave * 6144);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            wreg[((lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] = acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
          const int m = mw + i * 16 + row;
          float v[8];
          ld8(wreg + row * LS + c8, v);
          if (m < p.M) epilogue8<EPI>(p, m, nw + c8, v, cs);
        }
      }
      colsum_flush(p, nw + (lane & 7) * 8, cs, lane);
      c_k = 0;
      c_tile += gridDim.x;
      if (s + 2 < total) {
        __builtin_amdgcn_s_barrier();            // every wave is out of its staging region
        issue_half(s + 2, 0);                    // the deferred refill of that slot
        issue_half(s + 2, 1);
      }
    } else {
      ++c_k;
    }
  }
}

template <int EPI>
int launch_ring(const GemmArgs& p, hipStream_t s, int num_cu) {
  const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * kStage);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm_ring: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  int grid = ntm * ntn < num_cu ? ntm * ntn : num_cu;
  hipLaunchKernelGGL(gemm_ring_kernel<EPI>, dim3(grid), dim3(kThreads), kSlots * kStage, s, p, ntm, ntn);
  return check_launch("gemm_bf16_nt(ring)");
}

}  // namespace

namespace memhip {

// Returns MEMHIP_EUNSUPPORTED when the shape does not fit this structure (caller falls back).
int gemm_ring_dispatch(const GemmArgs& p, hipStream_t s) {
  const bool vec = ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux | p.colscale_n) & 7) == 0;      // host twin of vec_ok()
  // measured (tools/bench_gemm.py): wins for N >= 1024 (+8..12 %), loses 4..10 % at N = 768 where
  // 1182 tiles over 256 CUs leave a 40 % empty last round -> those stay on the 128x128 kernel
  if (p.M < 2048 || p.N < 1024 || p.N % BN != 0 || p.K % BK != 0 || !vec) return MEMHIP_EUNSUPPORTED;
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MEMHIP_EUNSUPPORTED;
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: return launch_ring<MEMHIP_EPI_BIAS_BF16>(p, s, num_cu);
    case MEMHIP_EPI_BIAS_GELU: return launch_ring<MEMHIP_EPI_BIAS_GELU>(p, s, num_cu);
    case MEMHIP_EPI_RESIDUAL: return launch_ring<MEMHIP_EPI_RESIDUAL>(p, s, num_cu);
    case MEMHIP_EPI_DGELU: return launch_ring<MEMHIP_EPI_DGELU>(p, s, num_cu);
    case MEMHIP_EPI_F32: return launch_ring<MEMHIP_EPI_F32>(p, s, num_cu);
    default: return MEMHIP_EUNSUPPORTED;
  }
}

}  // namespace memhip
