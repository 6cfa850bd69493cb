// Image-space augmentation chain between the rasterizer and the ViT, batched on the
// GPU with PER-SAMPLE parameters.  The random draws stay on the host (same generators, same order as the reference);
// these kernels are the arithmetic:
//   ToTensor + Resize(bilinear, antialias) / RandomCrop(pad_if_needed)      mem/datasets.py:637-642
//   ToUnit8 -> EventRandAugment (14 ops, 2 per sample) -> ToFloat32         mem/datasets.py:655-658, mem/transforms.py:292-484
//   ColorJitter(brightness, 0, saturation)                                  mem/datasets.py:34-38
// The reference calls torchvision's tensor ops for all of it (un-vendored; restated in oracle/aug_t.py, "parity
// unpinned (third party)"): uint8 images, float32 arithmetic, truncating casts, torch.round for resampled images.
// HBM-bound byte work (150 KB per 3x224x224 sample): one workgroup per sample, 16-byte accesses where the op is
// elementwise; statistics ops (Contrast mean, AutoContrast min/max, Equalize histograms) reduce in LDS.
#include "common.h"

// torch evaluates these chains as separate float32 tensor ops (mul, mul, add ...): no fused multiply-add anywhere.
// hipcc contracts a * b + c by default (and __fmul_rn / __fadd_rn are plain operators in the HIP headers), so this
// translation unit is built with -ffp-contract=off (csrc/Makefile; the pragma alone leaves inlined lambdas contracted).
#pragma clang fp contract(off)

namespace {

using namespace memhip;

constexpr int kT = 1024;

// ---------------------------------------------------------------- ToTensor + Resize(antialias) / crop
// torch's separable anti-aliased bilinear resize (aten UpSampleKernel.cpp, _compute_indices_min_size_weights_aa):
// per output index i: center = scale (i + 0.5), support = max(scale, 1), taps j in [xmin, xmin + xsize) with triangle
// weights normalised to 1; horizontal pass first (intermediate rounded to f32), then vertical.  fp32 throughout.
// A tap's weight is recomputed where it is used (tri(j) * norm: the same two float32 operations torch performs when it
// normalises its weight table), so the support is unbounded: a 440 x 640 canvas resized to 128 x 128 takes 11 taps
// (2 * ceil(scale) + 1), and there is no table to overflow.
struct Taps { int lo, n; float center, invscale, norm; };

__device__ __forceinline__ float aa_weight(const Taps& t, int j) {
  float x = ((float)(j + t.lo) - t.center + 0.5f) * t.invscale;
  x = fabsf(x);
  const float w = x < 1.0f ? 1.0f - x : 0.0f;
  return w * t.norm;
}

__device__ __forceinline__ Taps aa_taps(int i, int in_size, float scale) {
  Taps t;
  const float support = scale >= 1.0f ? scale : 1.0f;
  t.invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  t.center = scale * ((float)i + 0.5f);
  int lo = (int)(t.center - support + 0.5f);
  lo = lo > 0 ? lo : 0;
  int hi = (int)(t.center + support + 0.5f);
  hi = hi < in_size ? hi : in_size;
  const int n = hi - lo > 0 ? hi - lo : 0;
  t.lo = lo; t.n = n; t.norm = 1.0f;
  float total = 0.f;
  for (int j = 0; j < n; ++j) total += aa_weight(t, j);          // weights in tap order, as torch sums them
  t.norm = total != 0.f ? 1.0f / total : 0.f;
  return t;
}

// mode 0: resize every sample from its own (h, w) = dims[b] to (OH, OW);
// mode 1: crop window (OH, OW) at offs[b] = (top, left) of the image padded by (pad_t, pad_l) zeros (RandomCrop with
//         pad_if_needed pads BOTH sides by the deficit, torchvision RandomCrop.forward), no resampling.
__global__ __launch_bounds__(256) void resample_kernel(const uint8_t* __restrict__ in, const int32_t* __restrict__ dims,
                                                       long long slot_bytes, int fixed_h, int fixed_w, int mode,
                                                       const int32_t* __restrict__ offs, int OH, int OW,
                                                       float* __restrict__ out) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int h = dims ? dims[2 * b] : fixed_h, w = dims ? dims[2 * b + 1] : fixed_w;
  const uint8_t* src = in + (long long)b * slot_bytes + (long long)c * h * w;
  float* dst = out + ((long long)b * 3 + c) * OH * OW;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= OH * OW) return;
  const int oy = p / OW, ox = p - oy * OW;
  if (h <= 0 || w <= 0) { dst[p] = 0.f; return; }
  if (mode == 1) {
    const int pad_t = h < OH ? OH - h : 0, pad_l = w < OW ? OW - w : 0;
    const int y = oy + (offs ? offs[2 * b] : 0) - pad_t, x = ox + (offs ? offs[2 * b + 1] : 0) - pad_l;
    dst[p] = (y >= 0 && y < h && x >= 0 && x < w) ? (float)src[(long long)y * w + x] / 255.0f : 0.f;
    return;
  }
  if (h == OH && w == OW) { dst[p] = (float)src[p] / 255.0f; return; }   // F.resize returns the input when sizes match
  const float sx = (float)w / (float)OW, sy = (float)h / (float)OH;
  const Taps tx = aa_taps(ox, w, sx), ty = aa_taps(oy, h, sy);
  float acc = 0.f;
  for (int jy = 0; jy < ty.n; ++jy) {
    const uint8_t* row = src + (long long)(ty.lo + jy) * w + tx.lo;
    float hsum = (float)row[0] / 255.0f * aa_weight(tx, 0);       // horizontal pass of this input row
    for (int jx = 1; jx < tx.n; ++jx) hsum += (float)row[jx] / 255.0f * aa_weight(tx, jx);
    if (jy == 0) acc = hsum * aa_weight(ty, 0);
    else acc += hsum * aa_weight(ty, jy);
  }
  dst[p] = acc;
}

// ---------------------------------------------------------------- elementwise converters / ColorJitter
__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ x, long long n, uint8_t* __restrict__ y) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      // (255 * x).to(torch.uint8): truncation toward zero, then the low 8 bits (values here are in [0, 1])
      const unsigned a = (unsigned)(int)(255.0f * v.x) & 255u, b = (unsigned)(int)(255.0f * v.y) & 255u;
      const unsigned c = (unsigned)(int)(255.0f * v.z) & 255u, d = (unsigned)(int)(255.0f * v.w) & 255u;
      *reinterpret_cast<unsigned*>(y + i) = a | (b << 8) | (c << 16) | (d << 24);
    } else {
      for (long long k = i; k < n; ++k) y[k] = (uint8_t)((unsigned)(int)(255.0f * x[k]) & 255u);
    }
  }
}

struct JitterParams { int order; float bf, bf1, sf, sf1; };   // order: 0 none, 1 brightness only, 2 saturation only,
                                                              // 3 brightness then saturation, 4 saturation then brightness
// in: u8 (value / 255 = ToFloat32 first) or f32 [B,3,HW]; out f32 [B,out_chans,HW] (out_chans 2 = [pos, neg])
__global__ __launch_bounds__(256) void color_jitter_kernel(const void* __restrict__ in, int in_is_u8, int HW,
                                                           const JitterParams* __restrict__ params, float* __restrict__ out,
                                                           int out_chans) {
  const int b = blockIdx.y;
  JitterParams q{0, 1.f, 0.f, 1.f, 0.f};
  if (params) q = params[b];
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long i = ((long long)b * 3 + c) * HW + p;
      v[c] = in_is_u8 ? (float)reinterpret_cast<const uint8_t*>(in)[i] / 255.0f : reinterpret_cast<const float*>(in)[i];
    }
    auto bright = [&]() {                          // _blend(img, zeros, f) = (f*img + (1-f)*0).clamp(0, 1)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float t = __fadd_rn(__fmul_rn(q.bf, v[c]), __fmul_rn(q.bf1, 0.f));
        v[c] = fminf(fmaxf(t, 0.f), 1.f);
      }
    };
    auto satur = [&]() {                           // _blend(img, gray, f), gray = 0.2989 r + 0.587 g + 0.114 b
      const float g = __fadd_rn(__fadd_rn(__fmul_rn(0.2989f, v[0]), __fmul_rn(0.587f, v[1])), __fmul_rn(0.114f, v[2]));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float t = __fadd_rn(__fmul_rn(q.sf, v[c]), __fmul_rn(q.sf1, g));
        v[c] = fminf(fmaxf(t, 0.f), 1.f);
      }
    };
    if (q.order == 1 || q.order == 3) bright();
    if (q.order >= 2) satur();
    if (q.order == 4) bright();
    if (out_chans == 3) {
#pragma unroll
      for (int c = 0; c < 3; ++c) out[((long long)b * 3 + c) * HW + p] = v[c];
    } else {
      out[((long long)b * 2) * HW + p] = v[0];
      out[((long long)b * 2 + 1) * HW + p] = v[2];
    }
  }
}

// ---------------------------------------------------------------- EventRandAugment ops on uint8 [3, H, W]
struct RandAugOp { int op; float mag; float theta[6]; };   // theta: inverse affine matrix (torchvision
                                                           // _get_inverse_affine_matrix), float32
enum { OP_IDENTITY, OP_SHEARX, OP_SHEARY, OP_TRANSX, OP_TRANSY, OP_ROTATE, OP_BRIGHT, OP_COLOR, OP_CONTRAST, OP_SHARP,
       OP_POSTERIZE, OP_SOLARIZE, OP_AUTOCONTRAST, OP_EQUALIZE };

__device__ __forceinline__ uint8_t trunc_u8(float v) {          // clamp(0, 255).to(uint8)
  v = fminf(fmaxf(v, 0.f), 255.f);
  return (uint8_t)(int)v;
}
__device__ __forceinline__ uint8_t gray_u8(float r, float g, float b) {
  return (uint8_t)(int)__fadd_rn(__fadd_rn(__fmul_rn(0.2989f, r), __fmul_rn(0.587f, g)), __fmul_rn(0.114f, b));
}
__device__ __forceinline__ uint8_t blend_u8(float ratio, float ratio1, float a, float b) {
  return trunc_u8(__fadd_rn(__fmul_rn(ratio, a), __fmul_rn(ratio1, b)));
}

__device__ int block_sum_i(int v, int* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int s = 0;
  for (int i = 0; i < kT / 64; ++i) s += sh[i];
  return s;
}

__global__ __launch_bounds__(kT) void rand_aug_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                      const RandAugOp* __restrict__ ops, int H, int W) {
  __shared__ int sh[kT / 64];
  __shared__ int hist[3][256];
  __shared__ int lut[3][256];
  __shared__ int mm[3][2];
  const int b = blockIdx.x;
  const int HW = H * W;
  const uint8_t* src = in + (long long)b * 3 * HW;
  uint8_t* dst = out + (long long)b * 3 * HW;
  const RandAugOp q = ops[b];
  const int op = q.op;                                   // workgroup-uniform
  if (op == OP_IDENTITY) {
    for (int i = threadIdx.x * 16; i < 3 * HW; i += kT * 16) {
      if (i + 16 <= 3 * HW && (((uintptr_t)(src + i) | (uintptr_t)(dst + i)) & 15) == 0)
        *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
      else
        for (int k = i; k < i + 16 && k < 3 * HW; ++k) dst[k] = src[k];
    }
  } else if (op >= OP_SHEARX && op <= OP_ROTATE) {
    // F.affine / F.rotate on a tensor: grid = base_grid @ (theta^T / [0.5 w, 0.5 h]), grid_sample(bilinear, zeros,
    // align_corners=False) on the float image, torch.round, uint8
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const float t00 = q.theta[0] / hw, t01 = q.theta[1] / hw, t02 = q.theta[2] / hw;
    const float t10 = q.theta[3] / hh, t11 = q.theta[4] / hh, t12 = q.theta[5] / hh;
    for (int p = threadIdx.x; p < HW; p += kT) {
      const int oy = p / W, ox = p - oy * W;
      const float bx = (float)ox + (-(float)W * 0.5f + 0.5f), by = (float)oy + (-(float)H * 0.5f + 0.5f);
      const float gx = __fadd_rn(__fadd_rn(__fmul_rn(bx, t00), __fmul_rn(by, t01)), t02);
      const float gy = __fadd_rn(__fadd_rn(__fmul_rn(bx, t10), __fmul_rn(by, t11)), t12);
      const float ix = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), -1.f), 2.f);
      const float iy = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), -1.f), 2.f);
      const float xw = floorf(ix), yn = floorf(iy);
      const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
      const float nw = __fmul_rn(s, e), ne = __fmul_rn(s, w), sw = __fmul_rn(n, e), se = __fmul_rn(n, w);
      const int x0 = (int)xw, y0 = (int)yn;
      const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W, ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const uint8_t* ch = src + (long long)c * HW;
        const float vnw = (xa && ya) ? (float)ch[y0 * W + x0] : 0.f, vne = (xb && ya) ? (float)ch[y0 * W + x0 + 1] : 0.f;
        const float vsw = (xa && yb) ? (float)ch[(y0 + 1) * W + x0] : 0.f, vse = (xb && yb) ? (float)ch[(y0 + 1) * W + x0 + 1] : 0.f;
        const float r = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(vnw, nw), __fmul_rn(vne, ne)), __fmul_rn(vsw, sw)), __fmul_rn(vse, se));
        dst[(long long)c * HW + p] = (uint8_t)(int)rintf(r);
      }
    }
  } else if (op == OP_BRIGHT || op == OP_COLOR || op == OP_CONTRAST) {
    // _blend(img, other, ratio) = (ratio * img + (1 - ratio) * other).clamp(0, 255).to(uint8); the host passes
    // theta[1] = float32(ratio), theta[0] = float32(1.0 - ratio) with ratio = 1.0 + magnitude evaluated in double
    const float f1 = q.theta[0];
    float mean = 0.f;
    if (op == OP_CONTRAST) {
      int s = 0;
      for (int p = threadIdx.x; p < HW; p += kT) s += gray_u8((float)src[p], (float)src[HW + p], (float)src[2 * HW + p]);
      s = block_sum_i(s, sh);
      mean = (float)s / (float)HW;                          // integer-valued sum < 2^24: exact in float32 in any order
    }
    for (int p = threadIdx.x; p < HW; p += kT) {
      const float r = (float)src[p], g = (float)src[HW + p], bl = (float)src[2 * HW + p];
      float o = 0.f;
      if (op == OP_COLOR) o = (float)gray_u8(r, g, bl);
      else if (op == OP_CONTRAST) o = mean;
      dst[p] = blend_u8(q.theta[1], f1, r, o);
      dst[HW + p] = blend_u8(q.theta[1], f1, g, o);
      dst[2 * HW + p] = blend_u8(q.theta[1], f1, bl, o);
    }
  } else if (op == OP_SHARP) {
    // degenerate = 3x3 blur [[1,1,1],[1,5,1],[1,1,1]] / 13 (float32, rounded to uint8) inside, the image itself on the
    // one-pixel border; result = blend(img, degenerate, factor)
    const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
    for (int p = threadIdx.x; p < HW; p += kT) {
      const int y = p / W, x = p - y * W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const uint8_t* ch = src + (long long)c * HW;
        const float v = (float)ch[p];
        float d = v;
        if (H > 2 && W > 2) {
          if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
            float a = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
              for (int dx = -1; dx <= 1; ++dx)
                a = __fadd_rn(a, __fmul_rn((float)ch[(y + dy) * W + x + dx], (dy == 0 && dx == 0) ? k5 : k1));
            d = (float)(uint8_t)(int)rintf(a);
          }
          dst[(long long)c * HW + p] = blend_u8(q.theta[1], q.theta[0], v, d);
        } else {
          dst[(long long)c * HW + p] = ch[p];
        }
      }
    }
  } else if (op == OP_POSTERIZE) {
    const int bits = (int)q.mag;
    const unsigned m = (256u - (1u << (8 - bits))) & 255u;
    for (int i = threadIdx.x; i < 3 * HW; i += kT) dst[i] = (uint8_t)(src[i] & m);
  } else if (op == OP_SOLARIZE) {
    for (int i = threadIdx.x; i < 3 * HW; i += kT) {
      const uint8_t v = src[i];
      dst[i] = ((float)v >= q.mag) ? (uint8_t)(255 - v) : v;
    }
  } else if (op == OP_AUTOCONTRAST) {
    if (threadIdx.x < 6) mm[threadIdx.x >> 1][threadIdx.x & 1] = (threadIdx.x & 1) ? 0 : 255;
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
      int lo = 255, hi = 0;
      for (int p = threadIdx.x; p < HW; p += kT) { const int v = src[c * HW + p]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
      for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_xor(lo, o), d = __shfl_xor(hi, o); lo = a < lo ? a : lo; hi = d > hi ? d : hi; }
      if ((threadIdx.x & 63) == 0) { atomicMin(&mm[c][0], lo); atomicMax(&mm[c][1], hi); }
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
      float mn = (float)mm[c][0];
      float scale = 255.0f / ((float)mm[c][1] - mn);
      if (mm[c][1] == mm[c][0]) { mn = 0.f; scale = 1.f; }
      for (int p = threadIdx.x; p < HW; p += kT)
        dst[c * HW + p] = trunc_u8(__fmul_rn(__fadd_rn((float)src[c * HW + p], -mn), scale));
    }
  } else if (op == OP_EQUALIZE) {
    for (int i = threadIdx.x; i < 3 * 256; i += kT) (&hist[0][0])[i] = 0;
    __syncthreads();
    for (int c = 0; c < 3; ++c)
      for (int p = threadIdx.x; p < HW; p += kT) atomicAdd(&hist[c][src[c * HW + p]], 1);
    __syncthreads();
    if (threadIdx.x < 3) {
      const int c = threadIdx.x;
      int last = -1;
      for (int v = 0; v < 256; ++v) if (hist[c][v] != 0) last = v;
      const int step = (HW - (last >= 0 ? hist[c][last] : 0)) / 255;       // floor(sum(nonzero_hist[:-1]) / 255)
      if (step == 0) {
        for (int v = 0; v < 256; ++v) lut[c][v] = v;                        // channel returned unchanged
      } else {
        int cum = 0;
        for (int v = 0; v < 256; ++v) {
          const int l = v == 0 ? 0 : (cum + step / 2) / step;               // lut shifted right by one (pad [1, 0])[:-1]
          lut[c][v] = l > 255 ? 255 : l;
          cum += hist[c][v];
        }
      }
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c)
      for (int p = threadIdx.x; p < HW; p += kT) dst[c * HW + p] = (uint8_t)lut[c][src[c * HW + p]];
  }
}

}  // namespace

extern "C" int memhip_resample_to_f32(const uint8_t* in, const int32_t* dims, int64_t slot_bytes, int fixed_h, int fixed_w,
                                      int mode, const int32_t* offs, int B, int OH, int OW, float* out,
                                      memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && OH > 0 && OW > 0 && (mode == 0 || mode == 1), "resample: bad arguments");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out && (dims || (fixed_h > 0 && fixed_w > 0)) && slot_bytes > 0, "resample: null pointer / no size");
  hipLaunchKernelGGL(resample_kernel, dim3((OH * OW + 255) / 256, 3, B), dim3(256), 0, as_stream(stream), in, dims,
                     (long long)slot_bytes, fixed_h, fixed_w, mode, offs, OH, OW, out);
  return check_launch("resample_to_f32");
}

extern "C" int memhip_to_uint8(const float* x, int64_t n, uint8_t* y, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0, "to_uint8: negative size");
  if (n == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && y && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 3) == 0, "to_uint8: null / unaligned pointer");
  long long nb = (n / 4 + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
  hipLaunchKernelGGL(to_uint8_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), x, (long long)n, y);
  return check_launch("to_uint8");
}

extern "C" int memhip_color_jitter(const void* in, int in_is_u8, int B, int H, int W, const void* params, float* out,
                                   int out_chans, memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && (out_chans == 2 || out_chans == 3), "color_jitter: bad arguments");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out, "color_jitter: null pointer");
  static_assert(sizeof(JitterParams) == 20, "memhip_jitter_t ABI layout");
  int gx = (H * W + 255) / 256;
  gx = gx > 64 ? 64 : gx;
  hipLaunchKernelGGL(color_jitter_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream), in, in_is_u8, H * W,
                     reinterpret_cast<const JitterParams*>(params), out, out_chans);
  return check_launch("color_jitter");
}

extern "C" int memhip_rand_augment_u8(const uint8_t* in, uint8_t* out, const void* ops, int B, int H, int W,
                                      memhip_stream_t stream) {
  MEMHIP_REQUIRE(B >= 0 && H > 0 && W > 0, "rand_augment: bad shape");
  if (B == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(in && out && ops && in != out, "rand_augment: null pointer / in-place");
  static_assert(sizeof(RandAugOp) == 32, "memhip_randaug_op_t ABI layout");
  hipLaunchKernelGGL(rand_aug_kernel, dim3(B), dim3(kT), 0, as_stream(stream), in, out, reinterpret_cast<const RandAugOp*>(ops), H, W);
  return check_launch("rand_augment_u8");
}
