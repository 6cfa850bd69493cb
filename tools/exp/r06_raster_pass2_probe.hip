// Probe for pass 2 of the two-pass rasterizer: how fast do 640 workgroups (one per (sample, band), 16 waves each, one per CU as with
// 128 KB of LDS) read their key segments -- 832 bytes out of every 9216-byte chunk slot (the shipped chunk-major layout) against the
// same bytes laid out band-major (one contiguous stream per workgroup) -- with 1, 4 or 8 segment loads in flight per wave?
// No counting, no stores.  Build: hipcc -O3 --offload-arch=gfx950 -o variants/p2p tools/exp/r06_raster_pass2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NCH = 245, SLOT = 9216, SEG = 832, NB = 10;

template <bool BANDMAJOR, int DEPTH>
__global__ __launch_bounds__(1024) void probe(const char* __restrict__ keys, unsigned* sink) {
  extern __shared__ unsigned lds[];
  const int band = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  for (int c0 = wave; c0 < NCH; c0 += 16 * DEPTH) {
    uint4 q[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      int c = c0 + 16 * u; if (c >= NCH) c = NCH - 1;
      const size_t off = BANDMAJOR ? ((size_t)(b * NB + band) * NCH + c) * SEG : ((size_t)b * NCH + c) * SLOT + (size_t)band * SEG;
      q[u] = lane < SEG / 16 ? *reinterpret_cast<const uint4*>(keys + off + lane * 16) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) acc += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
  }
  if (acc == 0x12345u) sink[0] = lds[0];
}

template <bool BM, int D>
static void run(const char* name, const char* keys, unsigned* sink, int B, size_t lds_bytes) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<BM, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float sum = 0.f; const int reps = 10;
  for (int i = 0; i < reps + 2; ++i) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<BM, D>), dim3(NB, B), dim3(1024), lds_bytes, 0, keys, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (i >= 2) sum += ms;
  }
  const double mb = (double)B * NB * NCH * SEG / 1e6;
  printf("%-14s depth %d  B %3d  LDS %3zu KB: %6.1f us for %.0f MB = %.2f TB/s\n", name, D, B, lds_bytes >> 10, sum / reps * 1e3, mb, mb / (sum / reps) / 1e6);
}

int main() {
  const int Bmax = 64;
  char* keys; unsigned* sink;
  const size_t bytes = (size_t)Bmax * NCH * SLOT;
  CK(hipMalloc(&keys, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(keys, 1, bytes));
  char* evict; CK(hipMalloc(&evict, 1ull << 30));
  for (int B : {32, 64}) for (size_t lds : {(size_t)131072, (size_t)65536}) {
    CK(hipMemset(evict, 0, 1ull << 30));
    run<false, 1>("chunk-major", keys, sink, B, lds); run<false, 4>("chunk-major", keys, sink, B, lds); run<false, 8>("chunk-major", keys, sink, B, lds);
    run<true, 1>("band-major", keys, sink, B, lds); run<true, 4>("band-major", keys, sink, B, lds); run<true, 8>("band-major", keys, sink, B, lds);
  }
  return 0;
}
