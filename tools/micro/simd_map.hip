// Which SIMD does wave w of a workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
// hipcc -O2 --offload-arch=gfx950 tools/micro/simd_map.hip -o /tmp/simd_map && /tmp/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  extern __shared__ char smem[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 0) smem[0] = 1;
}
int main() {
  unsigned* d; hipMalloc(&d, 512 * 16 * 4);
  for (int threads : {448, 512}) {
    hipMemset(d, 0xff, 512 * 16 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 150 * 1024, 0, d);
    hipDeviceSynchronize();
    static unsigned h[512 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) hist[w][(h[b * 16 + w] >> 4) & 3]++;
    printf("%d threads: wave -> SIMD histogram over 256 workgroups\n", threads);
    for (int w = 0; w < threads / 64; ++w) printf("  wave %d: simd0 %3d simd1 %3d simd2 %3d simd3 %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("  first workgroups:"); for (int b = 0; b < 3; ++b) { printf(" ["); for (int w = 0; w < threads / 64; ++w) printf("%u", (h[b * 16 + w] >> 4) & 3); printf("]"); } printf("\n");
  }
  return 0;
}
