// Do the GFX9 whole-wave DPP shifts (wave_shr:1 = 0x138, wave_shl:1 = 0x130) exist on gfx950, and what do they do across the 16-lane rows?
// hipcc -O2 --offload-arch=gfx950 tools/micro/dpp_wave.hip -o /tmp/dpp_wave && /tmp/dpp_wave
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x138, 0xf, 0xf, false);        // wave_shr:1, old = -1
  out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[128 + l] = __builtin_amdgcn_update_dpp(0, l, 0x138, 0xf, 0xf, true);    // bound_ctrl: 0 for lane 0
}
int main() {
  int* d; (void)hipMalloc(&d, 192 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 3; ++r) { printf("%s:", r == 0 ? "wave_shr1" : r == 1 ? "wave_shl1" : "wave_shr1 bc"); for (int l = 0; l < 64; ++l) printf(" %d", h[r * 64 + l]); printf("\n"); }
  return 0;
}
