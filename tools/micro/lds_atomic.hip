// microbenchmark: LDS float vs integer atomic add rate (distinct addresses per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ float bins[1024];
  for (int i = threadIdx.x; i < 1024; i += 512) bins[i] = 0.f;
  __syncthreads();
  int idx = (threadIdx.x * 7) & 1023;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) atomicAdd(&bins[idx], 1.0f);
    else if (MODE == 1) atomicAdd((int*)&bins[idx], 3);
    else if (MODE == 2) bins[idx] += 1.0f;   // plain RMW (race; timing only)
    idx = (idx + 65) & 1023;
  }
  __syncthreads();
  if (threadIdx.x < 1024) out[blockIdx.x * 1024 + threadIdx.x % 1024] = bins[threadIdx.x % 1024];
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("mode %d: %.1f us total, %.1f ns per wave-instruction per CU (8 waves)\n", mode, ms * 1e3, ms * 1e6 / (iters * 8.0));
    }
  }
  return 0;
}
