// bf16 MFMA GEMM for the ViT contractions:  C[M,N] = A[M,K] * B[N,K]^T  (both operands
// K-contiguous, fp32 accumulate), with the reference's elementwise neighbours fused into the
// epilogue (bias, exact-erf GELU, layer-scale + stochastic-depth + residual, GELU', mask-token
// blend, fp32 gradient accumulation).  Reference ops: F.linear / nn.Linear / nn.Conv2d(k=s=16)
// in mem/modeling_finetune.py:61-70,132-155,203-209 and mem/modeling_pretrain.py:59,101-108.
//
// CDNA4 mapping: 128x128x64 macro-tile, 256 threads = 4 waves (2x2), each wave a 64x64
// sub-tile of 4x4 v_mfma_f32_16x16x32_bf16 fragments.  Operand tiles go HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), double-buffered; the LDS image is XOR-swizzled
// through the *source* address (LDS-DMA writes lane-linear) so that every ds_read_b128 fragment
// read is bank-conflict free.  Workgroup ids are remapped so that the 8 XCDs each walk a
// contiguous range of tiles (A row-panel reuse in the XCD-private L2).
#include <cstddef>
#include "common.h"
#include "gemm_epilogue.hpp"

namespace {

using namespace memhip;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kTileBytes = BM * BK * 2;       // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;   // A + B

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 16-byte slot index of (row, 16B-chunk) inside a [128][64] bf16 tile (128-B rows).
__device__ __forceinline__ int swz_slot(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Stage one [128][64] bf16 operand tile: 16 wave-instructions of 1 KiB (8 rows) each, 4 per wave.
// `src` points at (row0, k0) of a row-major [rows, ld] matrix; rows are clamped to `rows_valid-1`
// (out-of-range rows only feed outputs that the epilogue masks).
__device__ __forceinline__ void stage_tile(const __bf16* __restrict__ src, long long ld, int row0,
                                           int rows_valid, int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int inst = wave * 4 + j;
    const int row = inst * 8 + (lane >> 3);
    const int cpos = lane & 7;                       // chunk position in LDS
    const int chunk = cpos ^ ((row >> 1) & 7);       // global chunk that must land there
    int grow = row0 + row;
    grow = grow < rows_valid ? grow : rows_valid - 1;
    const __bf16* g = src + (long long)grow * ld + k0 + chunk * 8;
    glds16(g, lds_tile + inst * 1024);               // LDS dest = wave-uniform base + lane*16
  }
}

template <int EPI>
__global__ __launch_bounds__(kThreads, 2) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware, bijective remap of the linear block id (cdna guide T1)
  const int nwg = gridDim.x;
  int pid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntn = (p.N + BN - 1) / BN;
  const int tile_m = pid / ntn, tile_n = pid % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_tile(p.A, p.lda, m0, p.M, 0, smem, wave, lane);
  stage_tile(p.B, p.ldb, n0, p.N, 0, smem + kTileBytes, wave, lane);
  __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier

  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) {
      char* nxt = smem + (cur ^ 1) * kStageBytes;
      stage_tile(p.A, p.lda, m0, p.M, (t + 1) * BK, nxt, wave, lane);
      stage_tile(p.B, p.ldb, n0, p.N, (t + 1) * BK, nxt + kTileBytes, wave, lane);
    }
    const char* At = smem + cur * kStageBytes;
    const char* Bt = At + kTileBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
      const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wr * 64 + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(At + swz_slot(row, chunk) * 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wc * 64 + j * 16 + (lane & 15);
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bt + swz_slot(row, chunk) * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue.  C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
  // Fast path: each wave transposes its 64x64 fp32 sub-tile through its own 16 KiB slice of the
  // (now idle) staging LDS, 32 rows at a time, so that a lane owns 8 consecutive columns of one
  // row and every global access of the fused epilogue is a full 16-byte vector (a row of the
  // sub-tile = 256 contiguous bytes of fp32 / 128 of bf16).
  const int mw = m0 + wr * 64, nw = n0 + wc * 64;
  if (vec_ok(p) && nw + 64 <= p.N) {
    constexpr int LS = 72;                                  // padded row stride (floats)
    float* wreg = reinterpret_cast<float*>(smem + wave * 16384);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    EpiCols cols;
    epi_cols_load<EPI>(p, nw + (lane & 7) * 8, cols);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            wreg[(ii * 16 + (lane >> 4) * 4 + r) * LS + j * 16 + (lane & 15)] = acc[half * 2 + ii][j][r];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
        const int m = mw + half * 32 + row;
        float v[8];
        ld8(wreg + row * LS + c8, v);
        if (m < p.M) epilogue8<EPI>(p, m, nw + c8, v, cs, cols);
      }
    }
    colsum_flush(p, nw + (lane & 7) * 8, cs, lane);
    return;
  }
  float bias_n[4], vec_n[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + (lane & 15);
    const int nc = n < p.N ? n : p.N - 1;
    bias_n[j] = p.bias ? p.bias[nc] : 0.f;
    vec_n[j] = p.vec1 ? p.vec1[nc] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + (lane & 15);
        if (n < p.N) epilogue<EPI>(p, m, n, acc[i][j][r], bias_n[j], vec_n[j]);
      }
    }
  }
}

template <int EPI>
int launch(const GemmArgs& p, hipStream_t s) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes);
    if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "gemm: set smem attr: %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(ntm * ntn), dim3(kThreads), 2 * kStageBytes, s, p);
  return check_launch("gemm_bf16_nt");
}

}  // namespace

namespace memhip {
int gemm256_dispatch(const GemmArgs& p, hipStream_t s);
int gemm_p8_dispatch(const GemmArgs& p, hipStream_t s);
int gemm_p8_split_rows(const GemmArgs& p, hipStream_t s);
int gemm_p8_half_dispatch(const GemmArgs& p, hipStream_t s);
int gemm_p8_pair_dispatch(const GemmArgs& head, const GemmArgs& tail, hipStream_t s);
}

extern "C" int memhip_gemm_bf16_nt(const memhip_gemm_args_t* a, memhip_stream_t stream) {
  MEMHIP_REQUIRE(a, "gemm: null args");
  GemmArgs p;
  static_assert(sizeof(GemmArgs) >= sizeof(memhip_gemm_args_t) && offsetof(GemmArgs, m_base) >= sizeof(memhip_gemm_args_t) - 8,
                "GemmArgs = ABI struct + internal tail");
  __builtin_memset(&p, 0, sizeof(p));
  __builtin_memcpy(&p, a, sizeof(memhip_gemm_args_t));
  MEMHIP_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
  if (p.M == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
  MEMHIP_REQUIRE(p.A && p.B, "gemm: null operand");
  MEMHIP_REQUIRE(p.lda % 8 == 0 && p.ldb % 8 == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0,
                 "gemm: operands must be 16-byte aligned with ld %% 8 == 0");
  hipStream_t s = as_stream(stream);
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: MEMHIP_REQUIRE(p.out0, "gemm: out0"); break;
    case MEMHIP_EPI_BIAS_GELU: MEMHIP_REQUIRE(p.out0 && p.out1, "gemm: out0/out1"); break;
    case MEMHIP_EPI_RESIDUAL:
      MEMHIP_REQUIRE(p.resid, "gemm: residual args");
      MEMHIP_REQUIRE(!p.sample_map || (!p.rowmask && p.rows_per_sample > 0), "gemm: sample_map excludes rowmask");
      break;
    case MEMHIP_EPI_DGELU: MEMHIP_REQUIRE(p.out0 && p.aux, "gemm: dgelu args"); break;
    case MEMHIP_EPI_BIAS_GELU_DG: MEMHIP_REQUIRE(p.out0 && p.out1, "gemm: out0/out1"); break;
    case MEMHIP_EPI_MUL_AUX: MEMHIP_REQUIRE(p.out0 && p.aux, "gemm: mul_aux args"); break;
    case MEMHIP_EPI_F32: MEMHIP_REQUIRE(p.out0, "gemm: out0"); break;
    case MEMHIP_EPI_PATCH_EMBED: MEMHIP_REQUIRE(p.resid && p.vec1 && p.aux, "gemm: patch args"); break;
    default: return fail(MEMHIP_EINVAL, "gemm: unknown epilogue %d", p.epilogue);
  }
  // large token-dimension products: the phase-interleaved persistent kernel (gemm_p8.hip); the lockstep 256x256
  // kernel (gemm256.hip) takes shapes it does not (K a multiple of 64 but not of 128, or MEMHIP_GEMM_P8=0)
  const bool k256_on = opt(OPT_GEMM256) != 0;
  const bool p8_on = opt(OPT_GEMM_P8) != 0;
  if (p8_on) {
    // A persistent 256x256-tile launch whose last round would be poorly filled (N = 768: 591 tiles on
    // 256 CUs) only takes the rows of the full rounds; the remaining rows go to the 128x128 kernel
    // below (finer tiles, 2-3 workgroups per CU).  Rows are independent, so this is two launches of
    // the same contract on two row ranges.
    const bool split_on = opt(OPT_GEMM_SPLIT) != 0;
    const int split = split_on ? gemm_p8_split_rows(p, s) : 0;
    if (split > 0 && split < p.M && p.epilogue != MEMHIP_EPI_PATCH_EMBED) {
      const GemmArgs whole = p;
      GemmArgs head = p;
      head.M = split;
      {
        const long long r = split;
        p.A += r * p.lda;
        if (p.out0) p.out0 = (char*)p.out0 + r * p.ldo0 * (p.epilogue == MEMHIP_EPI_F32 ? 4 : 2);
        if (p.out1) p.out1 = (char*)p.out1 + r * p.ldo1 * 2;
        // (with a sample map the residual rows are addressed through the map: resid / aux stay where they are)
        if (p.resid && !p.sample_map) p.resid += r * p.ldr;
        if (p.aux && !(p.sample_map && p.epilogue == MEMHIP_EPI_RESIDUAL))
          p.aux = (const char*)p.aux + r * p.ldaux * (p.epilogue == MEMHIP_EPI_RESIDUAL ? 4 : 2);
        p.M -= split;
        p.m_base = split;
      }
      // one launch for both row ranges (gemm_p8.hip: gemm_p8_pair_kernel) when the epilogue has a paired form
      if (opt(OPT_GEMM_P8_HALF) != 0 && opt(OPT_GEMM_P8_PAIR) != 0) {
        const int rcp = gemm_p8_pair_dispatch(head, p, s);
        if (rcp != MEMHIP_EUNSUPPORTED) return rcp;
      }
      const int rc = gemm_p8_dispatch(head, s);
      if (rc == MEMHIP_OK) {
        // the left-over rows: the same phase structure on 128-row tiles (MEMHIP_GEMM_P8_HALF=0: 128x128 kernel)
        const bool half_on = opt(OPT_GEMM_P8_HALF) != 0;
        if (half_on) {
          const int rch = gemm_p8_half_dispatch(p, s);
          if (rch != MEMHIP_EUNSUPPORTED) return rch;
        }
        switch (p.epilogue) {
          case MEMHIP_EPI_BIAS_BF16: return launch<MEMHIP_EPI_BIAS_BF16>(p, s);
          case MEMHIP_EPI_BIAS_GELU: return launch<MEMHIP_EPI_BIAS_GELU>(p, s);
          case MEMHIP_EPI_RESIDUAL: return launch<MEMHIP_EPI_RESIDUAL>(p, s);
          case MEMHIP_EPI_DGELU: return launch<MEMHIP_EPI_DGELU>(p, s);
          case MEMHIP_EPI_BIAS_GELU_DG: return launch<MEMHIP_EPI_BIAS_GELU_DG>(p, s);
          case MEMHIP_EPI_MUL_AUX: return launch<MEMHIP_EPI_MUL_AUX>(p, s);
          default: return launch<MEMHIP_EPI_F32>(p, s);
        }
      }
      if (rc != MEMHIP_EUNSUPPORTED) return rc;
      p = whole;                                   // no persistent form for this call: the paths below take all rows
    } else {
      const int rc = gemm_p8_dispatch(p, s);
      if (rc != MEMHIP_EUNSUPPORTED) return rc;
    }
  }
  if (k256_on) {
    const int rc = gemm256_dispatch(p, s);
    if (rc != MEMHIP_EUNSUPPORTED) return rc;
  }
  switch (p.epilogue) {
    case MEMHIP_EPI_BIAS_BF16: MEMHIP_REQUIRE(p.out0, "gemm: out0"); return launch<MEMHIP_EPI_BIAS_BF16>(p, s);
    case MEMHIP_EPI_BIAS_GELU: MEMHIP_REQUIRE(p.out0 && p.out1, "gemm: out0/out1"); return launch<MEMHIP_EPI_BIAS_GELU>(p, s);
    case MEMHIP_EPI_RESIDUAL: return launch<MEMHIP_EPI_RESIDUAL>(p, s);
    case MEMHIP_EPI_DGELU: MEMHIP_REQUIRE(p.out0 && p.aux, "gemm: dgelu args"); return launch<MEMHIP_EPI_DGELU>(p, s);
    case MEMHIP_EPI_BIAS_GELU_DG: return launch<MEMHIP_EPI_BIAS_GELU_DG>(p, s);
    case MEMHIP_EPI_MUL_AUX: return launch<MEMHIP_EPI_MUL_AUX>(p, s);
    case MEMHIP_EPI_F32: MEMHIP_REQUIRE(p.out0, "gemm: out0"); return launch<MEMHIP_EPI_F32>(p, s);
    case MEMHIP_EPI_PATCH_EMBED: MEMHIP_REQUIRE(p.resid && p.vec1 && p.aux, "gemm: patch args"); return launch<MEMHIP_EPI_PATCH_EMBED>(p, s);
    default: return fail(MEMHIP_EINVAL, "gemm: unknown epilogue %d", p.epilogue);
  }
}
