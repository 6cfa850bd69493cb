// What does ONE CU's vector-memory pipeline charge per instruction?  (round 4: the GEMM epilogue's 128 stores per tile cost
// ~66 cycles each, its LDS-DMA loads ~36.)  Every workgroup (512 threads, one per CU, 256 of them) issues `iters` rounds of
// one 16-byte-per-lane instruction per wave in a given access SHAPE and reports its own cycles; all CUs run at once.
//   mode 0: LDS-DMA load  (global_load_lds_dwordx4)     mode 1: plain store     mode 2: nontemporal store
//   mode 3: register load (global_load_dwordx4)
//   shape 0: 8 rows x 128 B  (rows `stride` bytes apart: the GEMM operand piece)
//   shape 1: 16 rows x 64 B                              (the old epilogue store)
//   shape 2: 1 KiB contiguous
//   shape 3: 4 rows x 256 B
// The footprint per workgroup is `span` bytes, walked cyclically (small span: L2 resident; large: streams from HBM).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/vmem_pipe.hip -o tools/micro/vmem_pipe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void pipe_kernel(char* buf, long long span, int stride, int shape, int iters, unsigned long long* cyc,
                                                   unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* base = buf + (long long)blockIdx.x * span;
  // per-lane offset inside one instruction's footprint (all 32-bit: the loop must not be bound by its own address arithmetic)
  unsigned loff, ext;
  if (shape == 0) { loff = (unsigned)(lane >> 3) * stride + (lane & 7) * 16; ext = 8u * stride; }
  else if (shape == 1) { loff = (unsigned)(lane >> 2) * stride + (lane & 3) * 16; ext = 16u * stride; }
  else if (shape == 2) { loff = lane * 16; ext = 1024; }
  else { loff = (unsigned)(lane >> 4) * stride + (lane & 15) * 16; ext = 4u * stride; }
  // consecutive instructions of the workgroup walk disjoint footprints: the neighbouring column block (128 / 64 / 256 bytes
  // further) until the row (stride = 2048 bytes) is used up, then the next row group
  const unsigned colw = shape == 0 ? 128 : shape == 1 ? 64 : shape == 2 ? 1024 : 256;
  const unsigned cshift = shape == 0 ? 4 : shape == 1 ? 5 : shape == 2 ? 0 : 3;      // log2(2048 / colw); shape 2: one "column"
  const unsigned cmask = (1u << cshift) - 1;
  const unsigned uspan = (unsigned)span;
  u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned idx = wave;
#pragma unroll 8
  for (int it = 0; it < iters; ++it) {
    unsigned off = (idx >> cshift) * ext + (idx & cmask) * colw;
    off = (off & (uspan - 1)) + loff;                // span: a power of two, >= every footprint
    if (MODE == 0) glds16(base + off, smem + wave * 1024);
    else if (MODE == 1) *reinterpret_cast<u32x4*>(base + off) = v;
    else if (MODE == 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + off));
    else { const u32x4 x = *reinterpret_cast<const u32x4*>(base + off); acc += x; }
    idx += 8;
    if (MODE == 0 && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (MODE == 3 && acc[0] == 0x12345u) sink[0] = acc[1];
}

int main(int argc, char** argv) {
  const int grid = 256;
  const long long maxspan = 8LL << 20;
  char* buf; unsigned long long* cyc; unsigned* sink;
  hipMalloc(&buf, grid * maxspan + (4 << 20)); hipMalloc(&cyc, grid * 8); hipMalloc(&sink, 64);
  hipMemset(buf, 1, grid * maxspan);
  hipFuncSetAttribute(reinterpret_cast<const void*>(pipe_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192);
  const char* mname[] = {"lds-dma load", "plain store", "nt store", "reg load"};
  const char* sname[] = {"8 rows x 128 B", "16 rows x 64 B", "1 KiB contiguous", "4 rows x 256 B"};
  const int iters = 2048;                                   // per wave: 2048 x 1 KiB = 2 MiB per wave, 16 MiB per workgroup
  // grid 8 / 32: one / four workgroups per XCD -- the memory system is idle, what remains is the CU's own pipeline;
  // grid 256: every CU at once.  span 24 KiB per workgroup: L2 hits; 120 KiB: 30 MiB in all (L2 + Infinity Cache);
  // 8 MiB: 2 GiB in all (HBM)
  for (int mode = 0; mode < 4; ++mode)
    for (int g : {8, 32, 256})
      for (long long span : {32LL << 10, 128LL << 10, 8LL << 20})
        for (int shape = 0; shape < 4; ++shape) {
          if (g != 256 && span > (128LL << 10)) continue;
          const int stride = 2048;
          long long sp = span;
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          float best = 1e30f; std::vector<unsigned long long> h(g);
          for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(pipe_kernel<0>, dim3(g), dim3(512), 8192, 0, buf, sp, stride, shape, iters, cyc, sink);
            if (mode == 1) hipLaunchKernelGGL(pipe_kernel<1>, dim3(g), dim3(512), 8192, 0, buf, sp, stride, shape, iters, cyc, sink);
            if (mode == 2) hipLaunchKernelGGL(pipe_kernel<2>, dim3(g), dim3(512), 8192, 0, buf, sp, stride, shape, iters, cyc, sink);
            if (mode == 3) hipLaunchKernelGGL(pipe_kernel<3>, dim3(g), dim3(512), 8192, 0, buf, sp, stride, shape, iters, cyc, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
          }
          hipMemcpy(h.data(), cyc, g * 8, hipMemcpyDeviceToHost);
          std::sort(h.begin(), h.end());
          const double instr = 8.0 * iters, bytes = instr * 1024.0;
          printf("%-13s %-17s grid %3d span %5lld KiB/WG: %6.1f cycles per instruction per CU, %5.1f B/clk/CU, %6.1f GB/s/CU, chip %6.2f TB/s\n",
                 mname[mode], sname[shape], g, sp >> 10, h[g / 2] / instr, bytes / h[g / 2], bytes / (best * 1e-3) / 1e9, g * bytes / (best * 1e-3) / 1e12);
          fflush(stdout);
        }
  return 0;
}
