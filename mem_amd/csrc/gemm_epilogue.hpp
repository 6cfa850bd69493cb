// Epilogues of the bf16 GEMM (see gemm.hip).  The rounding points mirror the reference under
// autocast: every Linear/Conv output is rounded to bf16 first (fp32 accumulate + fp32 bias), the
// residual stream / layer-scale / mask-token arithmetic that follows runs in fp32.
#pragma once
#include "common.h"

namespace memhip {

struct GemmArgs {   // memhip_gemm_args_t, followed by launcher-internal fields
  const __bf16* A; const __bf16* B;
  long long lda, ldb;
  int M, N, K, epilogue;
  void* out0; long long ldo0;
  void* out1; long long ldo1;
  const float* bias;
  const float* vec1;
  float* resid; long long ldr;
  const void* aux; long long ldaux;
  const float* rowmask;
  float keep_prob;
  float colscale; int colscale_n;
  int rows_per_sample;
  int accumulate;
  float* colsum;   // optional: += column sums of the (rounded) primary output
  const int* sample_map;   // RESIDUAL: compact sample -> sample whose residual rows this output row updates (NULL: identity)
  int colsum_copies;       // > 1: colsum holds that many accumulator copies of N floats; a workgroup uses copy blockIdx % copies
  int reserved0;
  // ---- internal (not part of the C ABI; zero when the struct is copied from memhip_gemm_args_t)
  int m_base;      // row offset of this launch inside the caller's problem (a GEMM may be launched in two
                   // row ranges): only the per-sample row mask index needs the absolute row
};

__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }

// the column-sum accumulator of this workgroup (memhip.h: colsum_copies)
__device__ __forceinline__ float* colsum_base(const GemmArgs& p) {
  return p.colsum_copies > 1 ? p.colsum + (long long)((int)blockIdx.x % p.colsum_copies) * p.N : p.colsum;
}

// erf(x) by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-erff class accuracy) -- one
// rcp, one exp and five FMAs instead of libm's branchy erff: the GELU epilogues are VALU-bound
// otherwise.  The result feeds a bf16 rounding, 3 orders of magnitude coarser than the error.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float y = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(y, x);
}

// exact-erf GELU (nn.GELU default, modeling_finetune.py:57,62) and its derivative
__device__ __forceinline__ float gelu_f(float x) {
  return x * 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = __expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& p, int m, int n, float acc, float bias_n,
                                         float vec_n) {
  if constexpr (EPI == MEMHIP_EPI_BIAS_BF16) {
    float v = acc + bias_n;
    __bf16 y = (__bf16)v;
    if (n < p.colscale_n) y = (__bf16)((float)y * p.colscale);
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = y;
    if (p.colsum) atomicAdd(colsum_base(p) + n, (float)y);
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU) {
    const __bf16 h = (__bf16)(acc + bias_n);
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = h;
    reinterpret_cast<__bf16*>(p.out1)[(long long)m * p.ldo1 + n] = (__bf16)gelu_f((float)h);
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU_DG) {
    const __bf16 h = (__bf16)(acc + bias_n);
    reinterpret_cast<_Float16*>(p.out0)[(long long)m * p.ldo0 + n] = (_Float16)gelu_grad_f((float)h);
    reinterpret_cast<__bf16*>(p.out1)[(long long)m * p.ldo1 + n] = (__bf16)gelu_f((float)h);
  } else if constexpr (EPI == MEMHIP_EPI_MUL_AUX) {
    const float da = bf16_round(acc);
    const float g = (float)reinterpret_cast<const _Float16*>(p.aux)[(long long)m * p.ldaux + n];
    const __bf16 o = (__bf16)(da * g);
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = o;
    if (p.colsum) atomicAdd(colsum_base(p) + n, (float)o);
  } else if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    const __bf16 y = (__bf16)(acc + bias_n);
    if (p.out0) reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = y;
    float t = p.vec1 ? __fmul_rn(vec_n, (float)y) : (float)y;            // gamma * branch
    if (p.rowmask) t = __fmul_rn(__fdiv_rn(t, p.keep_prob), p.rowmask[(m + p.m_base) / p.rows_per_sample]);
    long long rr = m;                                                    // residual row (relative to resid / aux)
    if (p.sample_map) {                                                  // work-skipping stochastic depth: kept samples only
      const int mm = m + p.m_base, c = mm / p.rows_per_sample;
      rr = (long long)p.sample_map[c] * p.rows_per_sample + (mm - c * p.rows_per_sample);
      t = __fdiv_rn(t, p.keep_prob);
    }
    // residual input: aux (fp32, ldaux) when given, else in place
    const float xin = p.aux ? reinterpret_cast<const float*>(p.aux)[rr * p.ldaux + n] : p.resid[rr * p.ldr + n];
    p.resid[rr * p.ldr + n] = __fadd_rn(xin, t);
  } else if constexpr (EPI == MEMHIP_EPI_DGELU) {
    const float da = bf16_round(acc);
    const float h = (float)reinterpret_cast<const __bf16*>(p.aux)[(long long)m * p.ldaux + n];
    const __bf16 o = (__bf16)(da * gelu_grad_f(h));
    reinterpret_cast<__bf16*>(p.out0)[(long long)m * p.ldo0 + n] = o;
    if (p.colsum) atomicAdd(colsum_base(p) + n, (float)o);
  } else if constexpr (EPI == MEMHIP_EPI_F32) {
    float* o = reinterpret_cast<float*>(p.out0) + (long long)m * p.ldo0 + n;
    *o = p.accumulate ? (*o + acc) : acc;
  } else if constexpr (EPI == MEMHIP_EPI_PATCH_EMBED) {
    // modeling_pretrain.py:101-108: x*(1-w) + mask_token*w, rows shifted by the cls token
    const float y = bf16_round(acc + bias_n);
    const int L = p.rows_per_sample;
    const int b = m / L, pi = m - b * L;
    const float w = (float)reinterpret_cast<const unsigned char*>(p.aux)[m];
    const float v = __fadd_rn(__fmul_rn(y, 1.0f - w), __fmul_rn(vec_n, w));
    p.resid[((long long)b * (L + 1) + 1 + pi) * p.ldr + n] = v;
  }
}


typedef __attribute__((ext_vector_type(8))) __bf16 ebf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 ebf16x2;
typedef __attribute__((ext_vector_type(2))) float ef32x2;

__device__ __forceinline__ void ld8(const float* p, float* o) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float* o) {
  reinterpret_cast<float4*>(p)[0] = float4{o[0], o[1], o[2], o[3]};
  reinterpret_cast<float4*>(p)[1] = float4{o[4], o[5], o[6], o[7]};
}

// The vector epilogues are VALU work that no MFMA overlaps (all waves of a workgroup reach them
// together), so they are written for instruction count: two values per v_cvt_pk_bf16_f32, the
// rounded value recovered with one shift / and, packed fp32 math (v_pk_fma_f32 ...) for the erf
// polynomial, ONE exponential shared by erf and the normal pdf in the GELU derivative, and every
// optional feature (bias, column scale, column sums, drop-path) behind a wave-uniform branch.
__device__ __forceinline__ unsigned pack_bf16x2(ef32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ebf16x2));
}
__device__ __forceinline__ ef32x2 unpack_bf16x2(unsigned u) {
  return ef32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
// fp16 pairs (the stored GELU derivative: 11 significant bits instead of bf16's 8, same 16 bits per value)
typedef __attribute__((ext_vector_type(2))) _Float16 ef16x2;
__device__ __forceinline__ unsigned pack_f16x2(ef32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ef16x2));
}
__device__ __forceinline__ ef32x2 unpack_f16x2(unsigned u) {
  return __builtin_convertvector(__builtin_bit_cast(ef16x2, u), ef32x2);
}
__device__ __forceinline__ ef32x2 fma2(ef32x2 a, ef32x2 b, ef32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ ef32x2 splat2(float v) { return ef32x2{v, v}; }

// 16-byte streaming store of a finished bf16 output row piece: nontemporal (measured +6 % on the
// bias+GELU product, which writes 620 MB per launch: the lines do not linger in L2 as dirty data)
typedef __attribute__((ext_vector_type(4))) unsigned eu32x4;
__device__ __forceinline__ void st_stream16(void* base, long long elem_off, unsigned a, unsigned b, unsigned c, unsigned d) {
  __builtin_nontemporal_store(eu32x4{a, b, c, d}, reinterpret_cast<eu32x4*>(reinterpret_cast<__bf16*>(base) + elem_off));
}

// erf(x / sqrt 2) and exp(-x^2 / 2) for two values (A&S 7.1.26, see erf_fast)
__device__ __forceinline__ void erf_exp2(ef32x2 x, ef32x2& erf, ef32x2& e) {
  const ef32x2 z = x * splat2(0.70710678118654752440f);
  const ef32x2 az = __builtin_elementwise_abs(z);
  const ef32x2 d = fma2(splat2(0.3275911f), az, splat2(1.0f));
  const ef32x2 t = ef32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  ef32x2 poly = fma2(splat2(1.061405429f), t, splat2(-1.453152027f));
  poly = fma2(poly, t, splat2(1.421413741f));
  poly = fma2(poly, t, splat2(-0.284496736f));
  poly = fma2(poly, t, splat2(0.254829592f));
  const ef32x2 w = az * az * splat2(-1.4426950408889634f);
  e = ef32x2{__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const ef32x2 y = fma2(-(poly * t), e, splat2(1.0f));
  erf = ef32x2{copysignf(y.x, z.x), copysignf(y.y, z.y)};
}
__device__ __forceinline__ ef32x2 gelu2(ef32x2 x) {
  ef32x2 erf, e;
  erf_exp2(x, erf, e);
  const ef32x2 hx = x * splat2(0.5f);
  return fma2(hx, erf, hx);
}
__device__ __forceinline__ void gelu_and_grad2(ef32x2 x, ef32x2& g, ef32x2& dg) {
  ef32x2 erf, e;
  erf_exp2(x, erf, e);
  const ef32x2 hx = x * splat2(0.5f);
  g = fma2(hx, erf, hx);
  const ef32x2 cdf = fma2(splat2(0.5f), erf, splat2(0.5f));
  dg = fma2(x * splat2(0.39894228040143267794f), e, cdf);
}
__device__ __forceinline__ ef32x2 gelu_grad2(ef32x2 x) {
  ef32x2 erf, e;
  erf_exp2(x, erf, e);
  const ef32x2 cdf = fma2(splat2(0.5f), erf, splat2(0.5f));
  return fma2(x * splat2(0.39894228040143267794f), e, cdf);
}

// Row-vector form: 8 consecutive output columns n..n+7 of row m (n % 8 == 0, n + 8 <= N, every
// leading dimension a multiple of 8 elements, colscale_n % 8 == 0): 16-byte global accesses only.
// Per-column operands of a lane's 8 columns, loaded once per tile (not once per row)
struct EpiCols {
  ef32x2 bias[4];
  float g[8];
};
// Absent per-column operands are read from constant lines (bias 0, layer scale 1) through a POINTER selection, and the
// arithmetic that uses them is unconditional: a branch around the load -- even a wave-uniform one -- ends the basic block,
// and at the join hipcc's waitcnt pass waits with s_waitcnt vmcnt(0) for the operand, i.e. also for every store issued so
// far and for the LDS-DMA stream that is prefetching the next tile (two to three such drains per output tile).
// (The selected pointer must stay a GLOBAL pointer: a select between a kernel-argument pointer and the address of a
// __device__ constant is a generic pointer to hipcc, the load becomes flat_load, and a flat load is preceded by
// s_waitcnt vmcnt(0) while LDS-DMA is in flight -- it might read LDS -- and followed by vmcnt(0) lgkmcnt(0).)
static __device__ __attribute__((aligned(32))) const float g_epi_zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
static __device__ __attribute__((aligned(32))) const float g_epi_one8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
static __device__ __attribute__((aligned(128))) const float g_epi_zero256[256] = {};   // a whole tile's worth (gemm_p8.hip)
struct EpiOnes256 { float v[256]; constexpr EpiOnes256() : v() { for (int i = 0; i < 256; ++i) v[i] = 1.0f; } };
static __device__ __attribute__((aligned(128))) const EpiOnes256 g_epi_one256_s{};
#define g_epi_one256 (g_epi_one256_s.v)
typedef const float __attribute__((address_space(1)))* gcf32_ptr;
typedef float __attribute__((ext_vector_type(4))) ef32x4;
typedef const ef32x4 __attribute__((address_space(1)))* gcf32x4_ptr;
__device__ __forceinline__ gcf32_ptr as_global(const float* q) { return (gcf32_ptr)q; }
__device__ __forceinline__ void ld8g(gcf32_ptr q, float* o) {
  const ef32x4 a = reinterpret_cast<gcf32x4_ptr>(q)[0], b = reinterpret_cast<gcf32x4_ptr>(q)[1];
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <int EPI>
__device__ __forceinline__ void epi_cols_load(const GemmArgs& p, int n, EpiCols& c) {
  if constexpr (EPI != MEMHIP_EPI_DGELU && EPI != MEMHIP_EPI_F32 && EPI != MEMHIP_EPI_MUL_AUX) {
    float b[8];
    ld8g(p.bias ? as_global(p.bias + n) : as_global(g_epi_zero8), b);
#pragma unroll
    for (int k = 0; k < 4; ++k) c.bias[k] = ef32x2{b[2 * k], b[2 * k + 1]};
  }
  if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    ld8g(p.vec1 ? as_global(p.vec1 + n) : as_global(g_epi_one8), c.g);
  }
}

// Per-row operands (the GELU input for GELU', the fp32 residual for the residual epilogue): loaded
// separately so that a caller can issue the loads of several rows before it consumes the first
// (a load issued per row right before its use exposes one HBM latency per row).
template <int EPI>
struct EpiRow {};
template <>
struct EpiRow<MEMHIP_EPI_DGELU> { uint4 h; };
template <>
struct EpiRow<MEMHIP_EPI_MUL_AUX> { uint4 h; };
template <>
struct EpiRow<MEMHIP_EPI_RESIDUAL> { float x[8]; float rm; long long row; };   // row: the residual row (sample_map resolved)
// No vector-memory instruction of a row epilogue sits behind a branch, not even a wave-uniform one: at the join hipcc's
// waitcnt pass gives up counting and puts s_waitcnt vmcnt(0) in front of the next use of a loaded row, which then also waits
// for every store issued so far (one store round trip per row).  Optional operands are therefore handled by POINTER selection:
// a missing drop-path mask reads the constant 1, a missing bf16 copy of the branch output goes to a scratch line.
__device__ const float g_epi_one = 1.0f;
__device__ const int g_epi_izero = 0;
typedef const int __attribute__((address_space(1)))* gci32_ptr;
__device__ __attribute__((aligned(256))) unsigned char g_epi_trash[1024];
// BIGROWS = false: the caller guarantees m + m_base < 2^21 whenever a drop-path mask is given (no integer division, no
// branch: the row epilogue stays one basic block)
template <int EPI, bool BIGROWS = true>
__device__ __forceinline__ void epi_row_load(const GemmArgs& p, int m, int n, EpiRow<EPI>& r) {
  if constexpr (EPI == MEMHIP_EPI_DGELU || EPI == MEMHIP_EPI_MUL_AUX) {
    r.h = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(p.aux) + (long long)m * p.ldaux + n);
  } else if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    // (base pointer and leading dimension are selected as scalars: one address computation per lane)
    const float* base = p.aux ? reinterpret_cast<const float*>(p.aux) : p.resid;
    const long long ld = p.aux ? p.ldaux : p.ldr;
    // sample of the row for the drop-path mask: (m + 0.5) / rows_per_sample is at least 0.5 / rows_per_sample away from an
    // integer; the fp32 product is off by at most (m + 0.5) * 2^-23 / rows_per_sample, so it truncates to the right sample
    // for m < 2^22 (taken up to 2^21; larger row indices use the integer division)
    const int mm = m + p.m_base;
    const bool per_sample = p.rowmask || p.sample_map;
    const float inv = per_sample ? __frcp_rn((float)p.rows_per_sample) : 0.f;     // neither: sample 0 of the constants
    int smp = (int)(((float)mm + 0.5f) * inv);
    if constexpr (BIGROWS) {
      if (per_sample && mm >= (1 << 21)) smp = mm / p.rows_per_sample;
    }
    const gcf32_ptr rmb = p.rowmask ? as_global(p.rowmask) : as_global(&g_epi_one);
    r.rm = rmb[p.rowmask ? smp : 0];
    // work-skipping stochastic depth: the row of the KEPT sample in the residual stream (identity without a map)
    const gci32_ptr smb = p.sample_map ? (gci32_ptr)p.sample_map : (gci32_ptr)&g_epi_izero;
    const int kid = smb[p.sample_map ? smp : 0];
    r.row = p.sample_map ? (long long)kid * p.rows_per_sample + (mm - smp * p.rows_per_sample) : (long long)m;
    ld8(base + r.row * ld + n, r.x);
  }
}

// the optional bf16 copy of the branch output (out0 of the residual epilogue): a global store either way (see as_global)
typedef eu32x4 __attribute__((address_space(1)))* gu32x4_ptr;
__device__ __forceinline__ void st_branch_copy(const GemmArgs& p, int m, int n, const unsigned* y) {
  const gu32x4_ptr dst = p.out0 ? (gu32x4_ptr)(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n)
                                : (gu32x4_ptr)(g_epi_trash + 2 * (n & 255));
  *dst = eu32x4{y[0], y[1], y[2], y[3]};
}

// Packed width (dwords per lane) of a finished 8-column row piece: what epi8_math leaves and epi8_store writes.
template <int EPI> struct EpiPk { static constexpr int W = 4; };                       // 8 bf16
template <> struct EpiPk<MEMHIP_EPI_BIAS_GELU> { static constexpr int W = 8; };        // h | gelu(h)
template <> struct EpiPk<MEMHIP_EPI_BIAS_GELU_DG> { static constexpr int W = 8; };     // gelu'(h) | gelu(h)
template <> struct EpiPk<MEMHIP_EPI_RESIDUAL> { static constexpr int W = 8; };         // 8 fp32 of the residual stream
template <> struct EpiPk<MEMHIP_EPI_F32> { static constexpr int W = 8; };
template <> struct EpiPk<MEMHIP_EPI_PATCH_EMBED> { static constexpr int W = 8; };

// COPY (residual epilogue): 1 = the bf16 copy of the branch output goes to out0, or to a scratch line when out0 is NULL
// (a select, which hipcc may turn into a branch); 0 = no copy is written (p.out0 is not looked at); 2 = out0 is known
// to be non-NULL (no select: the row epilogue stays one basic block)
//
// The row epilogue in two halves: epi8_math is the ARITHMETIC of 8 consecutive output columns of row m (register only, plus
// the optional bf16 branch copy of the residual epilogue, which leaves at once), epi8_store writes its result.  gemm_p8.hip
// computes both column halves of a row fragment with epi8_math and stores them in full 128-byte lines (pq_pack); everything
// else calls both back to back (epilogue8).
template <int EPI, int COPY = 1>
__device__ __forceinline__ void epi8_math(const GemmArgs& p, int m, int n, const float* acc, float* cs, const EpiCols& c,
                                          const EpiRow<EPI>& row, unsigned* out) {
  ef32x2 t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = ef32x2{acc[2 * k], acc[2 * k + 1]};
  if constexpr (EPI != MEMHIP_EPI_DGELU && EPI != MEMHIP_EPI_F32 && EPI != MEMHIP_EPI_MUL_AUX) {
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] += c.bias[k];                 // zeros when there is no bias (epi_cols_load)
  }
  if constexpr (EPI == MEMHIP_EPI_BIAS_BF16) {
    unsigned y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = pack_bf16x2(t[k]);
    if (p.colscale_n > 0 && n < p.colscale_n) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = pack_bf16x2(unpack_bf16x2(y[k]) * splat2(p.colscale));
    }
    if (p.colsum) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const ef32x2 f = unpack_bf16x2(y[k]);
        cs[2 * k] += f.x;
        cs[2 * k + 1] += f.y;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = y[k];
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      out[k] = pack_bf16x2(t[k]);
      out[4 + k] = pack_bf16x2(gelu2(unpack_bf16x2(out[k])));
    }
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU_DG) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ef32x2 g, dg;
      gelu_and_grad2(unpack_bf16x2(pack_bf16x2(t[k])), g, dg);
      out[k] = pack_f16x2(dg);
      out[4 + k] = pack_bf16x2(g);
    }
  } else if constexpr (EPI == MEMHIP_EPI_MUL_AUX) {
    const unsigned h[4] = {row.h.x, row.h.y, row.h.z, row.h.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = pack_bf16x2(unpack_bf16x2(pack_bf16x2(t[k])) * unpack_f16x2(h[k]));
    if (p.colsum) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const ef32x2 f = unpack_bf16x2(out[k]);
        cs[2 * k] += f.x;
        cs[2 * k + 1] += f.y;
      }
    }
  } else if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
    unsigned y[4];
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = row.x[k];
    float br[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      y[k] = pack_bf16x2(t[k]);
      const ef32x2 f = unpack_bf16x2(y[k]);
      br[2 * k] = f.x;
      br[2 * k + 1] = f.y;
    }
    // layer scale: gamma * branch (own rounding, as the reference); gamma = 1 (exact) when there is none
#pragma unroll
    for (int k = 0; k < 8; ++k) br[k] = __fmul_rn(c.g[k], br[k]);
    {                                          // drop path: branch / keep_prob * mask[sample]; without a mask both are the
      const float rm = row.rm;                 // constant 1 and every step below is exact (q0 = br, residual 0, q = br)
      const float kp = (p.rowmask || p.sample_map) ? p.keep_prob : 1.0f;
      const float rk = __frcp_rn(kp);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // correctly rounded quotient from one Newton step on the reciprocal (no denormal inputs here)
        const float q0 = br[k] * rk;
        const float q = fmaf(fmaf(-q0, kp, br[k]), rk, q0);
        br[k] = __fmul_rn(q, rm);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = __float_as_uint(__fadd_rn(x[k], br[k]));
    if constexpr (COPY == 1) st_branch_copy(p, m, n, y);
    if constexpr (COPY == 2)
      *(gu32x4_ptr)(reinterpret_cast<__bf16*>(p.out0) + (long long)m * p.ldo0 + n) = eu32x4{y[0], y[1], y[2], y[3]};
  } else if constexpr (EPI == MEMHIP_EPI_DGELU) {
    const unsigned h[4] = {row.h.x, row.h.y, row.h.z, row.h.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const ef32x2 da = unpack_bf16x2(pack_bf16x2(t[k]));            // the matmul output is bf16
      out[k] = pack_bf16x2(da * gelu_grad2(unpack_bf16x2(h[k])));
    }
    if (p.colsum) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const ef32x2 f = unpack_bf16x2(out[k]);
        cs[2 * k] += f.x;
        cs[2 * k + 1] += f.y;
      }
    }
  } else if constexpr (EPI == MEMHIP_EPI_F32) {
    float* o = reinterpret_cast<float*>(p.out0) + (long long)m * p.ldo0 + n;
    if (p.accumulate) {
      float x[8];
      ld8(o, x);
#pragma unroll
      for (int k = 0; k < 8; ++k) out[k] = __float_as_uint(x[k] + acc[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) out[k] = __float_as_uint(acc[k]);
    }
  } else if constexpr (EPI == MEMHIP_EPI_PATCH_EMBED) {
    const float w = (float)reinterpret_cast<const unsigned char*>(p.aux)[m];
    float mt[8];
    ld8(p.vec1 + n, mt);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const ef32x2 f = unpack_bf16x2(pack_bf16x2(t[k]));
      out[2 * k] = __float_as_uint(__fadd_rn(__fmul_rn(f.x, 1.0f - w), __fmul_rn(mt[2 * k], w)));
      out[2 * k + 1] = __float_as_uint(__fadd_rn(__fmul_rn(f.y, 1.0f - w), __fmul_rn(mt[2 * k + 1], w)));
    }
  }
}

// second half: `out` of epi8_math to memory.  rrow: the residual-stream row of the RESIDUAL epilogue (EpiRow::row).
template <int EPI>
__device__ __forceinline__ void epi8_store(const GemmArgs& p, int m, int n, long long rrow, const unsigned* out) {
  if constexpr (EPI == MEMHIP_EPI_BIAS_BF16 || EPI == MEMHIP_EPI_MUL_AUX || EPI == MEMHIP_EPI_DGELU) {
    st_stream16(p.out0, (long long)m * p.ldo0 + n, out[0], out[1], out[2], out[3]);
  } else if constexpr (EPI == MEMHIP_EPI_BIAS_GELU || EPI == MEMHIP_EPI_BIAS_GELU_DG) {
    st_stream16(p.out0, (long long)m * p.ldo0 + n, out[0], out[1], out[2], out[3]);
    st_stream16(p.out1, (long long)m * p.ldo1 + n, out[4], out[5], out[6], out[7]);
  } else {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = __uint_as_float(out[k]);
    if constexpr (EPI == MEMHIP_EPI_RESIDUAL) {
      st8(p.resid + rrow * p.ldr + n, x);
    } else if constexpr (EPI == MEMHIP_EPI_F32) {
      st8(reinterpret_cast<float*>(p.out0) + (long long)m * p.ldo0 + n, x);
    } else if constexpr (EPI == MEMHIP_EPI_PATCH_EMBED) {
      const int L = p.rows_per_sample;
      const int b = m / L, pi = m - b * L;
      st8(p.resid + ((long long)b * (L + 1) + 1 + pi) * p.ldr + n, x);
    }
  }
}

template <int EPI, int COPY = 1>
__device__ __forceinline__ void epilogue8(const GemmArgs& p, int m, int n, const float* acc, float* cs, const EpiCols& c,
                                          const EpiRow<EPI>& row) {
  unsigned out[EpiPk<EPI>::W];
  epi8_math<EPI, COPY>(p, m, n, acc, cs, c, row, out);
  long long rrow = m;
  if constexpr (EPI == MEMHIP_EPI_RESIDUAL) rrow = row.row;
  epi8_store<EPI>(p, m, n, rrow, out);
}

// The residual epilogue from the already rounded and packed branch output y (8 columns = 4 bf16 pairs): see gemm_p8.hip
__device__ __forceinline__ void epilogue8_residual_packed(const GemmArgs& p, int m, int n, const unsigned* y, const EpiCols& c,
                                                          const EpiRow<MEMHIP_EPI_RESIDUAL>& row) {
  float br[8], x[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const ef32x2 f = unpack_bf16x2(y[k]);
    br[2 * k] = f.x;
    br[2 * k + 1] = f.y;
  }
  if (p.vec1) {                                // layer scale: gamma * branch (own rounding, as the reference)
#pragma unroll
    for (int k = 0; k < 8; ++k) br[k] = __fmul_rn(c.g[k], br[k]);
  }
  if (p.rowmask || p.sample_map) {             // drop path: branch / keep_prob * mask[sample]
    const float rm = row.rm;
    const float rk = __frcp_rn(p.keep_prob);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float q0 = br[k] * rk;
      const float q = fmaf(fmaf(-q0, p.keep_prob, br[k]), rk, q0);
      br[k] = __fmul_rn(q, rm);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = __fadd_rn(row.x[k], br[k]);
  st_branch_copy(p, m, n, y);
  st8(p.resid + row.row * p.ldr + n, x);
}

template <int EPI>
__device__ __forceinline__ void epilogue8(const GemmArgs& p, int m, int n, const float* acc, float* cs, const EpiCols& c) {
  EpiRow<EPI> row;
  epi_row_load<EPI>(p, m, n, row);
  epilogue8<EPI>(p, m, n, acc, cs, c, row);
}
template <int EPI>
__device__ __forceinline__ void epilogue8(const GemmArgs& p, int m, int n, const float* acc, float* cs) {
  EpiCols c;
  epi_cols_load<EPI>(p, n, c);
  epilogue8<EPI>(p, m, n, acc, cs, c);
}

// FULL-LINE memory accesses out of the 16x16 accumulator layout (gemm_p8.hip).  A lane of that layout holds 8 consecutive
// columns of ONE row (r = lane & 15) per column half, so a 16-byte-per-lane access covers 16 rows x 64 bytes: sixteen
// half-used 128-byte lines.  The CU's vector-memory pipeline costs ~4.2 cycles per line TOUCHED per instruction (round 4:
// 128 such stores per 256x256 bf16 tile = 8.5 K cycles, the same pipeline the operand stream needs), so the two 16-byte
// pieces a and b of a lane that are 64 bytes apart in memory (the two column halves of a bf16 row; the two halves of a
// lane's 32 bytes of an fp32 row) are exchanged between lanes r and r ^ 8 (DPP row_ror:8) into
//   P = rows 0-7 of the fragment, 128 contiguous bytes each, and Q = rows 8-15:
// lane (r, c) then accesses row (r & 7) [P] / 8 + (r & 7) [Q] at 16-byte chunk 4 (r >> 3) + c (bf16: a | b = 64 B | 64 B)
// or 2 c + (r >> 3) (fp32: a, b = the lane's two 16-byte halves).  Two instructions, 8 full lines each.
__device__ __forceinline__ unsigned swap8(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
}
// hi = (lane & 8) != 0
__device__ __forceinline__ void pq_pack(const unsigned* a, const unsigned* b, bool hi, unsigned* P, unsigned* Q) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned sb = swap8(b[k]), sa = swap8(a[k]);
    P[k] = hi ? sb : a[k];
    Q[k] = hi ? b[k] : sa;
  }
}
__device__ __forceinline__ void pq_unpack(const unsigned* P, const unsigned* Q, bool hi, unsigned* a, unsigned* b) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned sq = swap8(Q[k]), sp = swap8(P[k]);
    a[k] = hi ? sq : P[k];
    b[k] = hi ? Q[k] : sp;
  }
}

// Column sums when the 16 lanes with equal (lane >> 4) hold the same 8 columns n..n+7 for 16
// different rows (accumulator-layout epilogue of gemm_p8.hip).
// value of lane - N inside the 16-lane row, 0 for the first N lanes of a row (v_mov_b32_dpp row_shr:N, bound_ctrl)
template <int N>
__device__ __forceinline__ float dpp_row_shr(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ void colsum_flush16(const GemmArgs& p, int n, float* cs, int lane) {
  if (!p.colsum) return;
  float* const dst = colsum_base(p);
  // sum over the 16 lanes of a row by DPP shifts (1, 2, 4, 8: the last lane of the row ends up with the whole sum) --
  // four VALU instructions per column; the ds_bpermute form of __shfl_xor was 32 dependent LDS round trips per flush
  // (~6 K cycles per output tile of the GELU' GEMM)
  float t[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = cs[k];
    v += dpp_row_shr<1>(v);
    v += dpp_row_shr<2>(v);
    v += dpp_row_shr<4>(v);
    v += dpp_row_shr<8>(v);
    t[k] = v;
    cs[k] = 0.f;
  }
  if ((lane & 15) == 15) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(dst + n + k, t[k]);
  }
}

// Column sums of the rounded output (bias gradient of the producing Linear): cs[8] holds this lane's
// partial sums for columns n..n+7 over the rows it handled; lanes with equal (lane & 7) own the same
// columns.  One atomic per column per wave.
__device__ __forceinline__ void colsum_flush(const GemmArgs& p, int n, float* cs, int lane) {
  if (!p.colsum) return;
  float* const dst = colsum_base(p);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = cs[k];
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if ((lane >> 3) == 0) atomicAdd(dst + n + k, v);
    cs[k] = 0.f;
  }
}

// all leading dimensions / pointers the vector epilogue touches are 16-byte friendly
__device__ __forceinline__ bool vec_ok(const GemmArgs& p) {
  return ((p.ldo0 | p.ldo1 | p.ldr | p.ldaux | p.colscale_n) & 7) == 0 && (p.N & 7) == 0;
}

}  // namespace memhip
