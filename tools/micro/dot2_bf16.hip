// v_dot2c_f32_bf16 as "unpack one half of a packed bf16 pair and add an fp32": which half do the selectors pick, and is the
// sum correctly rounded?   hipcc --offload-arch=gfx950 -O3 -o /tmp/dot2 tools/micro/dot2_bf16.hip && /tmp/dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__global__ void k(const float* a, const float* c, float* o, unsigned sel_lo, unsigned sel_hi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const f32x2 v = {a[2 * i], a[2 * i + 1]};
  const bf16x2 pk = __builtin_convertvector(v, bf16x2);
  // constant selectors (the compiler may encode them as inline constants)
  o[4 * i + 0] = __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(bf16x2, 0x00003F80u), c[2 * i], false);
  o[4 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(bf16x2, 0x3F800000u), c[2 * i + 1], false);
  // run-time selectors (registers)
  o[4 * i + 2] = __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(bf16x2, sel_lo), c[2 * i], false);
  o[4 * i + 3] = __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(bf16x2, sel_hi), c[2 * i + 1], false);
}
static float bf16r(float x) {
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7FFF + ((u >> 16) & 1); u &= 0xFFFF0000u;
  float r; memcpy(&r, &u, 4); return r;
}
int main() {
  const int n = 1 << 16;
  float *a, *c, *o;
  hipMallocManaged(&a, 2 * n * 4); hipMallocManaged(&c, 2 * n * 4); hipMallocManaged(&o, 4 * n * 4);
  srand(1);
  for (int i = 0; i < 2 * n; ++i) {
    a[i] = ((rand() / (float)RAND_MAX) - 0.5f) * ((i & 7) == 0 ? 1e-3f : 40.f);
    c[i] = ((rand() / (float)RAND_MAX) - 0.5f) * ((i & 3) == 0 ? 1e-4f : 8.f);
  }
  c[5] = -INFINITY; c[8] = -INFINITY;
  k<<<n / 256, 256>>>(a, c, o, 0x00003F80u, 0x3F800000u);
  hipDeviceSynchronize();
  int bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float w0 = bf16r(a[2 * i]) + c[2 * i], w1 = bf16r(a[2 * i + 1]) + c[2 * i + 1];
    const float want[4] = {w0, w1, w0, w1};
    for (int j = 0; j < 4; ++j)
      if (!(o[4 * i + j] == want[j])) { if (bad[j]++ < 3) printf("j=%d i=%d got %.9g want %.9g (a %.9g %.9g c %.9g %.9g)\n", j, i, o[4*i+j], want[j], a[2*i], a[2*i+1], c[2*i], c[2*i+1]); }
  }
  printf("mismatches: const-lo %d const-hi %d reg-lo %d reg-hi %d of %d\n", bad[0], bad[1], bad[2], bad[3], n);
  return 0;
}
