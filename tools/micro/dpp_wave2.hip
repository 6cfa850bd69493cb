// wave_shr / wave_shl chains as used by attn16's bucket gradient: does shl1(c1) give lane + 1's c1?
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }
__global__ void k(const float* x, const float* link, const float* f, float* out) {
  const int l = threadIdx.x;
  float xb[4];
  for (int e = 0; e < 4; ++e) xb[e] = x[l * 4 + e];
  const float lk = link[l], fEnd = f[l], fD1 = f[64 + l], fD2 = f[128 + l];
  const float c1 = fmaf(shr1(xb[0]), lk, xb[1]);
  const float c2 = fmaf(shr1(c1), lk, xb[2]);
  const float c3 = fmaf(shr1(c2), lk, xb[3]);
  const float w = fmaf(fD2, shl1(shl1(c2)), fmaf(fD1, shl1(c1), fEnd * xb[0]));
  out[l] = c1; out[64 + l] = c2; out[128 + l] = c3; out[192 + l] = w; out[256 + l] = shl1(c1); out[320 + l] = shl1(shl1(c2));
}
int main() {
  float hx[256], hl[64], hf[192], ho[384];
  for (int i = 0; i < 256; ++i) hx[i] = (float)((i * 37) % 101) * 0.01f;
  for (int i = 0; i < 64; ++i) { hl[i] = (i % 14 == 0 || i == 32) ? 0.f : 1.f; hf[i] = (i % 5 == 0); hf[64 + i] = (i % 3 == 0); hf[128 + i] = (i % 7 == 0); }
  float *dx, *dl, *df, *dout;
  (void)hipMalloc(&dx, sizeof(hx)); (void)hipMalloc(&dl, sizeof(hl)); (void)hipMalloc(&df, sizeof(hf)); (void)hipMalloc(&dout, sizeof(ho));
  (void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice); (void)hipMemcpy(dl, hl, sizeof(hl), hipMemcpyHostToDevice); (void)hipMemcpy(df, hf, sizeof(hf), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dl, df, dout);
  (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  float c1[64], c2[64], c3[64]; int bad = 0;
  for (int l = 0; l < 64; ++l) c1[l] = fmaf(l ? hx[(l - 1) * 4] : 0.f, hl[l], hx[l * 4 + 1]);
  for (int l = 0; l < 64; ++l) c2[l] = fmaf(l ? c1[l - 1] : 0.f, hl[l], hx[l * 4 + 2]);
  for (int l = 0; l < 64; ++l) c3[l] = fmaf(l ? c2[l - 1] : 0.f, hl[l], hx[l * 4 + 3]);
  for (int l = 0; l < 64; ++l) {
    const float s1 = l + 1 < 64 ? c1[l + 1] : 0.f, s2 = l + 2 < 64 ? c2[l + 2] : 0.f;
    const float w = fmaf(hf[128 + l], s2, fmaf(hf[64 + l], s1, hf[l] * hx[l * 4]));
    if (ho[l] != c1[l] || ho[64 + l] != c2[l] || ho[128 + l] != c3[l] || ho[192 + l] != w || ho[256 + l] != s1 || ho[320 + l] != s2) {
      if (bad++ < 8) printf("lane %d: c1 %g/%g c2 %g/%g c3 %g/%g w %g/%g shl(c1) %g/%g shl2(c2) %g/%g\n", l, ho[l], c1[l], ho[64 + l], c2[l], ho[128 + l], c3[l], ho[192 + l], w, ho[256 + l], s1, ho[320 + l], s2);
    }
  }
  printf("bad lanes: %d\n", bad);
  return 0;
}
