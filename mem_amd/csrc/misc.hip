// Layout / dtype movers around the GEMMs: fp32 master -> bf16 shadow weights (plain and
// transposed), bf16 transposes that give the weight-gradient contraction its K-contiguous
// operands (with the Linear bias gradient = column sums fused in), im2col for the k=s=16
// patch-embed conv (mem/modeling_finetune.py:203-209), cls-token rows and the shared relative
// position bias gather (mem/modeling_finetune.py:242-247).  All HBM-bound, 16-B lane accesses.
#include "common.h"

namespace {

using namespace memhip;

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, __bf16* __restrict__ out,
                                                   long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    reinterpret_cast<bf16x4*>(out)[i] = o;
  }
}

// 64x64 tile transpose through LDS.  IN = float (cast to bf16) or bf16.
// out[c][r] = in[r][c]; rows r in [R, R_pad) of the transposed output are zero-filled.
template <typename IN>
__global__ __launch_bounds__(256) void transpose_kernel(const IN* __restrict__ in, long long ldin, int R,
                                                        int Cc, __bf16* __restrict__ out, long long ldout,
                                                        int R_pad, float* __restrict__ cs0, int c0b, int c0e,
                                                        float* __restrict__ cs1, int c1b, int c1e) {
  __shared__ __bf16 tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int t = threadIdx.x;
  // load: 8 threads per row (8 elements each), 32 rows per pass
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rl = (t >> 3) + pass * 32, cl = (t & 7) * 8;
    const int r = r0 + rl;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + cl + k;
      __bf16 v = (__bf16)0.f;
      if (r < R && c < Cc) v = (__bf16)(float)in[(long long)r * ldin + c];
      tile[rl][cl + k] = v;
    }
  }
  __syncthreads();
  // store: output row = input column; 8 threads per output row, 8 consecutive r each
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = (t >> 3) + pass * 32, rl = (t & 7) * 8;
    const int c = c0 + cl;
    bf16x8 o;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { o[k] = tile[rl + k][cl]; s += (float)o[k]; }
    if (c < Cc && r0 + rl < R_pad)
      *reinterpret_cast<bf16x8*>(out + (long long)c * ldout + r0 + rl) = o;
    if (cs0 || cs1) {
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      if ((t & 7) == 0 && c < Cc) {
        if (cs0 && c >= c0b && c < c0e) atomicAdd(cs0 + (c - c0b), s);
        if (cs1 && c >= c1b && c < c1e) atomicAdd(cs1 + (c - c1b), s);
      }
    }
  }
}

// All weight transposes of a step in ONE launch (49 launches of ~8 us each otherwise): desc[i] = {in, ldin, R, Cc,
// out, ldout}, tile_prefix[i] = first 64x64 tile of matrix i (tile_prefix[n] = total).  Vectorised: a thread reads
// 8 consecutive fp32 of a row (two 16-byte loads) when the tile is interior.
struct TransposeDesc { const float* in; long long ldin; long long R; long long Cc; __bf16* out; long long ldout; };
__global__ __launch_bounds__(256) void transpose_cast_batched_kernel(const TransposeDesc* __restrict__ desc,
                                                                     const int* __restrict__ tile_prefix, int n) {
  __shared__ __bf16 tile[64][66];
  int mi = 0;
  while (mi + 1 < n && (int)blockIdx.x >= tile_prefix[mi + 1]) ++mi;          // n <= a few dozen
  const TransposeDesc d = desc[mi];
  const int local = blockIdx.x - tile_prefix[mi];
  const int R = (int)d.R, Cc = (int)d.Cc;
  const int tr = (R + 63) / 64;
  const int r0 = (local % tr) * 64, c0 = (local / tr) * 64;
  const int t = threadIdx.x;
  const int R_pad = (int)(d.ldout < (long long)tr * 64 ? d.ldout : (long long)tr * 64);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rl = (t >> 3) + pass * 32, cl = (t & 7) * 8;
    const int r = r0 + rl, c = c0 + cl;
    const float* src = d.in + (long long)r * d.ldin + c;
    if (r < R && c + 8 <= Cc && ((d.ldin & 3) == 0) && (((unsigned long long)d.in & 15) == 0)) {
      const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[rl][cl + k] = (__bf16)v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[rl][cl + k] = (r < R && c + k < Cc) ? (__bf16)src[k] : (__bf16)0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = (t >> 3) + pass * 32, rl = (t & 7) * 8;
    const int c = c0 + cl;
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = tile[rl + k][cl];
    if (c < Cc && r0 + rl < R_pad) *reinterpret_cast<bf16x8*>(d.out + (long long)c * d.ldout + r0 + rl) = o;
  }
}

// x f32 [B,C,H,W] -> patches bf16 [B*L, C*ph*pw], k = c*ph*pw + py*pw + px (Conv2d weight order)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                                     int ph, int pw, __bf16* __restrict__ out) {
  const int gw = W / pw, gh = H / ph;
  const int K = C * ph * pw, K8 = K >> 3;
  const long long total = (long long)B * gh * gw * K8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k8 = (int)(i % K8);
    const long long row = i / K8;
    const int k = k8 * 8;
    const int c = k / (ph * pw), rem = k - c * ph * pw, py = rem / pw, px = rem - py * pw;
    const int gx = (int)(row % gw);
    const long long t = row / gw;
    const int gy = (int)(t % gh), b = (int)(t / gh);
    const float* src = x + (((long long)b * C + c) * H + gy * ph + py) * W + gx * pw + px;
    const float4 a = reinterpret_cast<const float4*>(src)[0], d = reinterpret_cast<const float4*>(src)[1];
    bf16x8 o;
    o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
    o[4] = (__bf16)d.x; o[5] = (__bf16)d.y; o[6] = (__bf16)d.z; o[7] = (__bf16)d.w;
    reinterpret_cast<bf16x8*>(out)[i] = o;
  }
}

__global__ __launch_bounds__(256) void fill_cls_kernel(float* __restrict__ x, long long ldx, int B, int T, int D,
                                                       const float* __restrict__ cls) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) x[(long long)b * T * ldx + c] = cls[c];
}

// bias[h][q][k] = table[idx[q*T+k]][h] for q,k < T, 0 in the padding (TP >= T)
__global__ __launch_bounds__(256) void relpos_gather_kernel(const float* __restrict__ table,
                                                            const int* __restrict__ idx, int T, int TP, int Hh,
                                                            float* __restrict__ bias, float* __restrict__ biasT) {
  const long long total = (long long)Hh * TP * TP;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % TP);
    const long long t = i / TP;
    const int q = (int)(t % TP), h = (int)(t / TP);
    float v = 0.f;
    if (q < T && k < T) v = table[(long long)idx[q * T + k] * Hh + h];
    bias[i] = v;
    if (biasT) biasT[((long long)h * TP + k) * TP + q] = v;      // [h][key][query] copy for attn_bwd_kv
  }
}


// y[n] += sum_k W[n,k] * x[k]  (bf16 weights, fp32 vectors), one wave per output; optionally
// x_acc[k] += x[k] and zero[k] = 0 (see memhip_gemv_bf16_acc in memhip.h).
__global__ __launch_bounds__(256) void gemv_acc_kernel(const __bf16* __restrict__ W, long long ldw, int N, int K,
                                                       const float* __restrict__ x, float* __restrict__ y,
                                                       float* __restrict__ x_acc, float* __restrict__ zero) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n < N) {
    float s = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(W + (long long)n * ldw + k);
      const float4 a = *reinterpret_cast<const float4*>(x + k), b = *reinterpret_cast<const float4*>(x + k + 4);
      s += (float)w[0] * a.x + (float)w[1] * a.y + (float)w[2] * a.z + (float)w[3] * a.w + (float)w[4] * b.x +
           (float)w[5] * b.y + (float)w[6] * b.z + (float)w[7] * b.w;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[n] += s;
  }
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
      if (x_acc) x_acc[k] += x[k];
      if (zero) zero[k] = 0.f;
    }
  }
}

}  // namespace

extern "C" int memhip_cast_f32_bf16(const float* in, void* out, int64_t n, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && n % 4 == 0, "cast: n must be a multiple of 4");
  if (n == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out, "cast: null pointer");
  const long long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, (__bf16*)out, n4);
  return check_launch("cast_f32_bf16");
}

extern "C" int memhip_transpose_cast_f32_bf16(const float* in, int64_t ldin, int R, int Cc, void* out,
                                              int64_t ldout, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R > 0 && Cc > 0 && in && out, "transpose_cast: bad arguments");
  MEMHIP_REQUIRE(ldout % 8 == 0 && ldout >= R, "transpose_cast: ldout");
  const int R_pad = (int)(ldout < (int64_t)cdiv(R, 64) * 64 ? ldout : (int64_t)cdiv(R, 64) * 64);
  hipLaunchKernelGGL(transpose_kernel<float>, dim3(cdiv(R, 64), cdiv(Cc, 64)), dim3(256), 0, as_stream(stream), in,
                     (long long)ldin, R, Cc, (__bf16*)out, (long long)ldout, R_pad, (float*)nullptr, 0, 0,
                     (float*)nullptr, 0, 0);
  return check_launch("transpose_cast");
}

extern "C" int memhip_transpose_cast_batched(const void* desc, const int32_t* tile_prefix, int n, int total_tiles,
                                             memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && total_tiles >= 0, "transpose_cast_batched: bad arguments");
  if (n == 0 || total_tiles == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(desc && tile_prefix, "transpose_cast_batched: null pointer");
  static_assert(sizeof(TransposeDesc) == 48, "descriptor = 6 x 8 bytes");
  hipLaunchKernelGGL(transpose_cast_batched_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream),
                     (const TransposeDesc*)desc, tile_prefix, n);
  return check_launch("transpose_cast_batched");
}

extern "C" int memhip_transpose_bf16(const void* in, int64_t ldin, int R, int Cc, void* out, int64_t ldout,
                                     int R_pad, float* colsum0, int c0_begin, int c0_end, float* colsum1,
                                     int c1_begin, int c1_end, memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && Cc > 0 && in && out, "transpose: bad arguments");
  MEMHIP_REQUIRE(R_pad >= R && R_pad % 64 == 0 && ldout >= R_pad && ldout % 8 == 0,
                 "transpose: R_pad=%d must be a multiple of 64 and <= ldout", R_pad);
  if (R_pad == 0) return MEMHIP_OK;
  hipLaunchKernelGGL(transpose_kernel<__bf16>, dim3(R_pad / 64, cdiv(Cc, 64)), dim3(256), 0, as_stream(stream),
                     (const __bf16*)in, (long long)ldin, R, Cc, (__bf16*)out, (long long)ldout, R_pad, colsum0,
                     c0_begin, c0_end, colsum1, c1_begin, c1_end);
  return check_launch("transpose_bf16");
}

extern "C" int memhip_im2col_bf16(const float* x, int B, int C, int H, int W, int ph, int pw, void* out,
                                  memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && C > 0 && ph > 0 && pw > 0 && H % ph == 0 && W % pw == 0, "im2col: bad shape");
  MEMHIP_REQUIRE(pw % 8 == 0 && W % 4 == 0, "im2col: patch width must be a multiple of 8");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && out, "im2col: null pointer");
  const long long total = (long long)B * (H / ph) * (W / pw) * (C * ph * pw / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, B, C, H, W, ph, pw,
                     (__bf16*)out);
  return check_launch("im2col");
}

// dst[sample] = src[sample] for the listed samples (n_per_sample fp32 values each, a multiple of 4): the rows of the
// samples a stochastic-depth branch dropped pass through the residual stream unchanged (work-skipping mode)
__global__ __launch_bounds__(256) void copy_samples_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           const int* __restrict__ ids, long long n_per_sample) {
  const long long base = (long long)ids[blockIdx.y] * n_per_sample;
  const float4* s4 = reinterpret_cast<const float4*>(src + base);
  float4* d4 = reinterpret_cast<float4*>(dst + base);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_per_sample / 4; i += (long long)gridDim.x * 256) d4[i] = s4[i];
}

extern "C" int memhip_copy_samples_f32(const float* src, float* dst, const int32_t* ids, int n, int64_t n_per_sample,
                                       memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && n_per_sample > 0 && n_per_sample % 4 == 0, "copy_samples: bad shape");
  if (n == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(src && dst && ids, "copy_samples: null pointer");
  int gx = (int)((n_per_sample / 4 + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(copy_samples_kernel, dim3(gx, n), dim3(256), 0, as_stream(stream), src, dst, (const int*)ids,
                     (long long)n_per_sample);
  return check_launch("copy_samples");
}

// ---- zero fills of the step (gradient accumulators, padding rows): streaming 16-byte stores
__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, long long n16) {
  const long long step = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += step) p[i] = uint4{0u, 0u, 0u, 0u};
}
// ranges[2 r] = byte offset, ranges[2 r + 1] = byte count of range r (multiples of 16): every workgroup walks every range
__global__ __launch_bounds__(256) void zero_ranges_kernel(char* __restrict__ base, const long long* __restrict__ ranges, int n) {
  const long long step = (long long)gridDim.x * 256;
  for (int r = 0; r < n; ++r) {
    uint4* p = reinterpret_cast<uint4*>(base + ranges[2 * r]);
    const long long n16 = ranges[2 * r + 1] >> 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += step) p[i] = uint4{0u, 0u, 0u, 0u};
  }
}

extern "C" int memhip_zero(void* p, int64_t bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(bytes >= 0 && bytes % 16 == 0 && ((uintptr_t)p & 15) == 0, "zero: pointer and size must be multiples of 16 bytes");
  if (bytes == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(p, "zero: null pointer");
  long long g = (bytes / 16 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (uint4*)p, (long long)(bytes / 16));
  return check_launch("zero");
}

extern "C" int memhip_zero_ranges(void* base, const int64_t* ranges, int n, int64_t total_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && total_bytes >= 0, "zero_ranges: bad count");
  if (n == 0 || total_bytes == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(base && ranges && ((uintptr_t)base & 15) == 0, "zero_ranges: null / unaligned pointer");
  long long g = (total_bytes / 16 + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (char*)base, (const long long*)ranges, n);
  return check_launch("zero_ranges");
}

// ---- dead-row elimination in the last block (vit_engine.py: tail rows).  Only the rows that reach the head (the masked
// tokens) need the last block's MLP branch: it runs on those rows in compact form, and these two kernels connect the compact
// rows with the token-major residual stream.
//   residual_rows: out[i, :] = x[rows[i], :] + drop_path(gamma * y[i, :])   -- the arithmetic of the GEMM's RESIDUAL epilogue
//   (gemm_epilogue.hpp) on the already rounded bf16 branch output y: gamma * y rounded, the drop-path quotient correctly
//   rounded (reciprocal + one Newton step), times the row's keep flag, one rounding for the add.
//   scatter_rows:  dst[rows[i], :] = src[i, :]
__global__ __launch_bounds__(256) void residual_rows_kernel(const float* __restrict__ x, long long ldx, const int* __restrict__ rows,
                                                            const __bf16* __restrict__ y, long long ldy,
                                                            const float* __restrict__ gamma, const float* __restrict__ rowkeep,
                                                            float keep_prob, int R, int D, float* __restrict__ out, long long ldo) {
  const int per_row = D / 4;
  const float kp = rowkeep ? keep_prob : 1.0f;
  const float rk = __frcp_rn(kp);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)R * per_row; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / per_row), c = (int)(i % per_row) * 4;
    const float rm = rowkeep ? rowkeep[r] : 1.0f;
    const float4 xv = *reinterpret_cast<const float4*>(x + (long long)rows[r] * ldx + c);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float br = __fmul_rn(gamma ? gamma[c + k] : 1.0f, (float)y[(long long)r * ldy + c + k]);
      const float q0 = br * rk;
      const float q = fmaf(fmaf(-q0, kp, br), rk, q0);
      br = __fmul_rn(q, rm);
      o[k] = __fadd_rn(xs[k], br);
    }
    *reinterpret_cast<float4*>(out + (long long)r * ldo + c) = float4{o[0], o[1], o[2], o[3]};
  }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, long long lds, const int* __restrict__ rows,
                                                           int R, int D, float* __restrict__ dst, long long ldd) {
  const int per_row = D / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)R * per_row; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / per_row), c = (int)(i % per_row) * 4;
    *reinterpret_cast<float4*>(dst + (long long)rows[r] * ldd + c) = *reinterpret_cast<const float4*>(src + (long long)r * lds + c);
  }
}

extern "C" int memhip_residual_rows(const float* x, int64_t ldx, const int32_t* rows, const void* y, int64_t ldy, const float* gamma,
                                    const float* rowkeep, float keep_prob, int R, int D, float* out, int64_t ldo,
                                    memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "residual_rows: bad shape");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && rows && y && out, "residual_rows: null pointer");
  long long n = (long long)R * (D / 4);
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(residual_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, (long long)ldx, (const int*)rows,
                     (const __bf16*)y, (long long)ldy, gamma, rowkeep, keep_prob, R, D, out, (long long)ldo);
  return check_launch("residual_rows");
}

extern "C" int memhip_scatter_rows_f32(const float* src, int64_t lds, const int32_t* rows, int R, int D, float* dst, int64_t ldd,
                                       memhip_stream_t stream) {
  MEMHIP_REQUIRE(R >= 0 && D > 0 && D % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "scatter_rows: bad shape");
  if (R == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(src && rows && dst, "scatter_rows: null pointer");
  long long n = (long long)R * (D / 4);
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), src, (long long)lds, (const int*)rows, R, D,
                     dst, (long long)ldd);
  return check_launch("scatter_rows");
}

extern "C" int memhip_fill_cls(float* x, int64_t ldx, int B, int T, int D, const float* cls,
                               memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && T > 0 && D > 0, "fill_cls: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && cls, "fill_cls: null pointer");
  hipLaunchKernelGGL(fill_cls_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, (long long)ldx, B, T, D, cls);
  return check_launch("fill_cls");
}

extern "C" int memhip_relpos_gather(const float* table, const int32_t* index, int T, int TP, int heads,
                                    float* bias, float* biasT, memhip_stream_t stream) {
  MEMHIP_REQUIRE(T > 0 && TP >= T && heads > 0 && table && index && bias, "relpos_gather: bad arguments");
  const long long total = (long long)heads * TP * TP;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(relpos_gather_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), table, index, T, TP,
                     heads, bias, biasT);
  return check_launch("relpos_gather");
}

extern "C" int memhip_gemv_bf16_acc(const void* W, int64_t ldw, int N, int K, const float* x, float* y, float* x_acc,
                                    float* zero, memhip_stream_t stream) {
  MEMHIP_REQUIRE(N >= 0 && K >= 0 && K % 8 == 0 && ldw % 8 == 0, "gemv: K and ldw must be multiples of 8");
  if (N == 0 || K == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(W && x && y, "gemv: null pointer");
  hipLaunchKernelGGL(gemv_acc_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), (const __bf16*)W,
                     (long long)ldw, N, K, x, y, x_acc, zero);
  return check_launch("gemv_bf16_acc");
}
