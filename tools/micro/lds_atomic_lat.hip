// microbenchmark: what a run of back-to-back LDS atomics costs the wave that issues them (cycles per instruction, s_memtime
// around N atomics + s_waitcnt lgkmcnt(0)), by address pattern, by type, alone on the CU or with the other waves doing the same.
//   pattern 0: 64 distinct consecutive words          pattern 1: lane r + 32 hits the word of lane r + 4 (the attn16 bucket pattern)
//   pattern 2: all lanes one word                     pattern 3: 64 distinct words, stride 2 (32 banks used twice)
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_lat.out lds_atomic_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TYPE>   // 0: ds_add_u32, 1: ds_add_f32, 2: ds_read_b32 (for scale)
__global__ __launch_bounds__(512) void k(unsigned long long* out, int pattern, int nwaves_active) {
  __shared__ int bins[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) bins[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int w;
  if (pattern == 0) w = lane;
  else if (pattern == 1) w = (lane & 31) + 4 * (lane >> 5);
  else if (pattern == 2) w = 0;
  else w = 2 * lane;
  w += wave * 256;
  const unsigned addr = (unsigned)(unsigned long long)((__attribute__((address_space(3))) int*)&bins[w]);
  unsigned long long t0 = 0, t1 = 0;
  if (wave < nwaves_active) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (TYPE == 0) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(addr), "v"(3), "n"(512 * (i & 1)) : "memory");
      else if (TYPE == 1) asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(addr), "v"(1.0f), "n"(512 * (i & 1)) : "memory");
      else { int r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(512 * (i & 1)) : "memory"); acc += r; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (acc == 12345) bins[0] = acc;
  }
  __syncthreads();
  if (lane == 0 && wave < nwaves_active) out[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8 * 8 * 8);
  unsigned long long h[8];
  const char* tn[3] = {"ds_add_u32", "ds_add_f32", "ds_read_b32"};
  for (int type = 0; type < 3; ++type)
    for (int pattern = 0; pattern < 4; ++pattern)
      for (int nw = 1; nw <= 8; nw += 7) {
        for (int rep = 0; rep < 2; ++rep) {
          if (type == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, d, pattern, nw);
          if (type == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, d, pattern, nw);
          if (type == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(512), 0, 0, d, pattern, nw);
          hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-12s pattern %d, %d wave(s): wave 0: %llu shader cycles (s_memtime) per 32 instructions = %.1f each\n", tn[type],
               pattern, nw, h[0], h[0] / 32.0);
      }
  return 0;
}
