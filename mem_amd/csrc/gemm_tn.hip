// Weight-gradient GEMM:  G[N,K] (+)= sum_m  dY[m,N] * X[m,K]   (reduction over the TOKEN rows).
// Reference op: the weight gradient of every F.linear / Conv2d on the path
// (mem/modeling_finetune.py:61-70,132-155,203-209; mem/modeling_pretrain.py:59), which autograd
// computes as dY^T @ X.
//
// Both operands are token-major (the reduction index is the ROW), so neither is K-contiguous the
// way an MFMA lane wants it.  Instead of materialising transposed copies in HBM, the tiles are
// staged row-major ([64 tokens][128 cols], LDS-DMA, 16 KiB each) and the fragments are read
// "down the columns" with ds_read_b64_tr_b16 (gfx950 transposing LDS read): two reads give a
// lane its 8 reduction elements.  The MFMA's k slots are permuted (slot 8g+j <-> token
// 4g+j / 16+4g+(j-4)); A and B use the same permutation, so the sum is unchanged, and with it the
// eight rows a 32-lane half touches are consecutive => an XOR of the 32-byte segment index with
// (row & 7) (applied through the LDS-DMA source address) makes the reads bank-conflict free.
//
// The output is small (<= 3072 x 768) and the reduction long (50 432 tokens), so the grid is
// split-K: S slices of the token range per output tile, combined with fp32 atomics into the
// (pre-zeroed) gradient buffer -- S is chosen so that tiles * S fills the 256 CUs.
#include "common.h"

namespace {

using namespace memhip;

constexpr int BM = 128, BN = 128, BR = 64;       // output tile (BM x BN), reduction rows per stage
constexpr int kThreads = 256;
constexpr int kTileBytes = BR * 128 * 2;        // 16 KiB
constexpr int kStageBytes = 2 * kTileBytes;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];   // zero-initialised

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Stage a [64 rows][128 cols] bf16 tile (256-B rows): 16 wave-instructions of 1 KiB (4 rows).
// LDS chunk position cpos (16 B) of row r holds global chunk ((cpos>>1) ^ (r&7))<<1 | (cpos&1).
// Rows >= rows_valid and chunks beyond cols_valid come from a zero page (they must add nothing).
__device__ __forceinline__ void stage_rows(const __bf16* __restrict__ src, long long ld, int row0, int rows_valid,
                                           int col0, int cols_valid, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int inst = wave * 4 + j;
    const int r = inst * 4 + (lane >> 4);
    const int cpos = lane & 15;
    const int chunk = (((cpos >> 1) ^ (r & 7)) << 1) | (cpos & 1);
    const int grow = row0 + r, gcol = col0 + chunk * 8;
    const void* g = (grow < rows_valid && gcol < cols_valid)
                        ? (const void*)(src + (long long)grow * ld + gcol)
                        : (const void*)(g_zero_page + cpos * 16);
    glds16(g, lds_tile + inst * 1024);
  }
}

// 8 bf16 of column `col` (multiple of 16 + lane&15 handled by the hardware transpose) for the
// k-step `ks` (32 rows): rows 4g+q and 16+4g+q, g = lane>>4.
__device__ __forceinline__ bf16x8 read_frag_tr(const char* tile, int ks, int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int colb = col0 + 4 * p;                      // this lane addresses 4 columns of row q
  const int seg = colb >> 4, inb = (colb & 15) * 2;
  const int r0 = ks * 32 + 4 * g + q, r1 = r0 + 16;
  const char* a0 = tile + r0 * 256 + ((seg ^ (r0 & 7)) << 5) + inb;
  const char* a1 = tile + r1 * 256 + ((seg ^ (r1 & 7)) << 5) + inb;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lo;
  u.s.h = hi;
  return u.v;
}

__global__ __launch_bounds__(kThreads, 2) void gemm_tn_kernel(const __bf16* __restrict__ A, long long lda,
                                                              const __bf16* __restrict__ B, long long ldb,
                                                              int R, int N, int K, float* __restrict__ out,
                                                              long long ldo, int splits, int rows_per_split,
                                                              int use_atomics) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntk = (K + BN - 1) / BN;
  const int tiles = ((N + BM - 1) / BM) * ntk;
  // consecutive block ids share one output tile's neighbours; the split index is the slow one
  const int tile = blockIdx.x % tiles, sp = blockIdx.x / tiles;
  const int n0 = (tile / ntk) * BM, k0 = (tile % ntk) * BN;
  const int rbeg = sp * rows_per_split;
  int rend = rbeg + rows_per_split;
  rend = rend < R ? rend : R;
  if (rbeg >= rend) return;
  const int nst = (rend - rbeg + BR - 1) / BR;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_rows(A, lda, rbeg, rend, n0, N, smem, wave, lane);
  stage_rows(B, ldb, rbeg, rend, k0, K, smem + kTileBytes, wave, lane);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nst; ++t) {
    if (t + 1 < nst) {
      char* nxt = smem + (cur ^ 1) * kStageBytes;
      stage_rows(A, lda, rbeg + (t + 1) * BR, rend, n0, N, nxt, wave, lane);
      stage_rows(B, ldb, rbeg + (t + 1) * BR, rend, k0, K, nxt + kTileBytes, wave, lane);
    }
    const char* At = smem + cur * kStageBytes;
    const char* Bt = At + kTileBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag_tr(At, ks, wr * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag_tr(Bt, ks, wc * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
      if (n >= N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wc * 64 + j * 16 + (lane & 15);
        if (k >= K) continue;
        float* o = out + (long long)n * ldo + k;
        if (use_atomics) atomicAdd(o, acc[i][j][r]);
        else *o = acc[i][j][r];
      }
    }
}

// column sums of a bf16 [R, C] matrix into fp32 (atomics): Linear bias gradients.  A workgroup covers tpr 16-byte
// column chunks x (256 / tpr) rows at a time, so that narrow matrices (C = 512: 64 chunks) still use all 256 threads
// (one thread column per chunk left three waves of every workgroup idle: 0.28 TB/s); eight rows per thread are in flight;
// the row groups are folded through LDS, one atomic per column and workgroup.
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16* __restrict__ in, long long ld, int R, int Cc,
                                                     int rows_per_block, int tpr, float* __restrict__ out) {
  __shared__ float red[256 * 8];
  const int rp = 256 / tpr;                               // row groups
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
  const bool active = ty < rp;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(R, r0 + rows_per_block);
  for (int cb = blockIdx.x * tpr; cb * 8 < Cc; cb += gridDim.x * tpr) {
    const int c = (cb + tx) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active && c < Cc) {
      int r = r0 + ty;
      for (; r + 7 * rp < r1; r += 8 * rp) {               // eight rows in flight per thread
        bf16x8 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const bf16x8*>(in + (long long)(r + u * rp) * ld + c);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < 8; ++k) s[k] += (float)v[u][k];
      }
      for (; r < r1; r += rp) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(in + (long long)r * ld + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = s[k];
    __syncthreads();
    if (ty == 0 && c < Cc) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = 0.f;
        for (int g = 0; g < rp; ++g) t += red[(g * tpr + tx) * 8 + k];
        atomicAdd(out + c + k, t);
      }
    }
  }
}

}  // namespace

namespace memhip {
int gemm_tn_p8_dispatch(const void* A, long long lda, const void* B, long long ldb, int R, int N, int K, float* out,
                        long long ldo, int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
size_t gemm_tn_p8_workspace(int R, int N, int K);
size_t gemm_tn_p8_group_workspace(const memhip_tn_problem_t* pr, int count);
int gemm_tn_p8_group_dispatch(const memhip_tn_problem_t* pr, int count, int accumulate, float* ws, size_t ws_bytes,
                              hipStream_t s);
}

extern "C" size_t memhip_gemm_bf16_tn_workspace(int R, int N, int K) { return memhip::gemm_tn_p8_workspace(R, N, K); }

extern "C" int memhip_gemm_bf16_tn_ws(const void* A, int64_t lda, const void* B, int64_t ldb, int R, int N, int K,
                                      float* out, int64_t ldo, int accumulate, void* workspace,
                                      size_t workspace_bytes, memhip_stream_t stream);

extern "C" int memhip_gemm_bf16_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int R, int N, int K,
                                   float* out, int64_t ldo, int accumulate, memhip_stream_t stream) {
  return memhip_gemm_bf16_tn_ws(A, lda, B, ldb, R, N, K, out, ldo, accumulate, nullptr, 0, stream);
}

extern "C" int memhip_gemm_bf16_tn_ws(const void* A, int64_t lda, const void* B, int64_t ldb, int R, int N, int K,
                                      float* out, int64_t ldo, int accumulate, void* workspace,
                                      size_t workspace_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && N > 0 && K > 0, "gemm_tn: bad shape R=%d N=%d K=%d", R, N, K);
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(A && B && out, "gemm_tn: null pointer");
  MEMHIP_REQUIRE(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 &&
                     ((uintptr_t)B & 15) == 0, "gemm_tn: operands must be 16-byte aligned, N/K/ld %% 8 == 0");
  hipStream_t s = as_stream(stream);
  const bool p8_on = opt(OPT_TN_P8) != 0;
  if (p8_on) {
    const int rc = gemm_tn_p8_dispatch(A, lda, B, ldb, R, N, K, out, ldo, accumulate, (float*)workspace,
                                       workspace_bytes, s);
    if (rc != MEMHIP_EUNSUPPORTED) return rc;
  }
  const int tiles = cdiv(N, BM) * cdiv(K, BN);
  const int stages = cdiv(R, BR);
  int splits = cdiv(768, tiles);                       // ~3 workgroups per CU
  if (splits > stages / 4) splits = stages / 4;        // keep >= 4 stages (256 rows) per split
  if (splits < 1) splits = 1;
  const int rows_per_split = cdiv(stages, splits) * BR;
  splits = cdiv(R, rows_per_split);
  const int use_atomics = (splits > 1 || accumulate) ? 1 : 0;
  if (splits > 1 && !accumulate)
    MEMHIP_HIP(hipMemset2DAsync(out, (size_t)ldo * sizeof(float), 0, (size_t)K * sizeof(float), (size_t)N, s));
  static bool attr_done = false;
  if (!attr_done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(kThreads), 2 * kStageBytes, s, (const __bf16*)A,
                     (long long)lda, (const __bf16*)B, (long long)ldb, R, N, K, out, (long long)ldo, splits,
                     rows_per_split, use_atomics);
  return check_launch("gemm_bf16_tn");
}

extern "C" size_t memhip_gemm_bf16_tn_group_workspace(const memhip_tn_problem_t* problems, int count) {
  if (!problems || count <= 0) return 0;
  size_t need = memhip::gemm_tn_p8_group_workspace(problems, count);
  for (int i = 0; i < count; ++i) {                        // ... and enough for the one-by-one path
    const size_t one = memhip::gemm_tn_p8_workspace(problems[i].R, problems[i].N, problems[i].K);
    need = one > need ? one : need;
  }
  return need;
}

extern "C" int memhip_gemm_bf16_tn_group(const memhip_tn_problem_t* problems, int count, int accumulate, void* workspace,
                                         size_t workspace_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(problems && count >= 1 && count <= 4, "gemm_tn_group: 1..4 problems, got %d", count);
  bool all = true;
  for (int i = 0; i < count; ++i) {
    const memhip_tn_problem_t& q = problems[i];
    MEMHIP_REQUIRE(q.R >= 0 && q.N > 0 && q.K > 0, "gemm_tn_group[%d]: bad shape R=%d N=%d K=%d", i, q.R, q.N, q.K);
    MEMHIP_REQUIRE(q.R == 0 || (q.A && q.B && q.out), "gemm_tn_group[%d]: null pointer", i);
    MEMHIP_REQUIRE(q.N % 8 == 0 && q.K % 8 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 && ((uintptr_t)q.A & 15) == 0 &&
                       ((uintptr_t)q.B & 15) == 0, "gemm_tn_group[%d]: operands must be 16-byte aligned, N/K/ld %% 8 == 0", i);
    all = all && q.R > 0;
  }
  hipStream_t s = as_stream(stream);
  if (all && count > 1 && opt(OPT_TN_P8) != 0 && opt(OPT_TN_GROUP) != 0) {
    const int rc = gemm_tn_p8_group_dispatch(problems, count, accumulate, (float*)workspace, workspace_bytes, s);
    if (rc != MEMHIP_EUNSUPPORTED) return rc;
  }
  for (int i = 0; i < count; ++i) {
    const memhip_tn_problem_t& q = problems[i];
    const int rc = memhip_gemm_bf16_tn_ws(q.A, q.lda, q.B, q.ldb, q.R, q.N, q.K, q.out, q.ldo, accumulate, workspace,
                                          workspace_bytes, stream);
    if (rc != MEMHIP_OK) return rc;
  }
  return MEMHIP_OK;
}

namespace {
__global__ __launch_bounds__(256) void colsum_fold_kernel(float* __restrict__ ws, int copies, int N, float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < copies; ++k) {
    s += ws[(long long)k * N + n];
    ws[(long long)k * N + n] = 0.f;
  }
  out[n] += s;
}
}  // namespace

extern "C" int memhip_colsum_fold(float* ws, int copies, int N, float* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(copies >= 1 && N > 0, "colsum_fold: bad shape");
  MEMHIP_REQUIRE(ws && out, "colsum_fold: null pointer");
  hipLaunchKernelGGL(colsum_fold_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), ws, copies, N, out);
  return check_launch("colsum_fold");
}

extern "C" int memhip_colsum_bf16(const void* in, int64_t ld, int R, int Cc, float* out, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && Cc > 0 && Cc % 8 == 0 && ld % 8 == 0, "colsum: bad shape");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out, "colsum: null pointer");
  const int chunks = Cc / 8;
  const int tpr = chunks >= 256 ? 256 : chunks;       // threads per row; the other 256 / tpr thread rows take more rows
  const int gx = cdiv(chunks, tpr);
  // about one workgroup per CU, at most 128 row blocks: every row block ends in one atomic per column, and atomics on
  // the same address serialise at ~0.17 us each (measured: 784 row blocks of a [50176, 768] matrix took 161 us, 135 of
  // them in the atomics; 128 row blocks keep 3 MB of loads in flight, enough for the read itself)
  int gy = 256 / gx;
  if (gy > 128) gy = 128;
  if (gy < 1) gy = 1;
  int rows_per_block = cdiv(R, gy);
  if (rows_per_block < 64) rows_per_block = 64;
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, cdiv(R, rows_per_block)), dim3(256), 0, as_stream(stream),
                     (const __bf16*)in, (long long)ld, R, Cc, rows_per_block, tpr, out);
  return check_launch("colsum_bf16");
}
