// How fast does the chip take the store / load pattern of a GEMM epilogue?  256 workgroups x 512 threads each walk 256x256
// bf16 tiles of a [50432 x 3072] matrix (persistent, same tile order as gemm_p8); a wave covers 128 rows x 64 columns of the
// tile and writes it with 16-byte lane accesses in one of these shapes per wave-instruction:
//   mode 0: 16 rows x 64 B  (4 lanes per row)   -- the accumulator layout of gemm_p8's epilogue
//   mode 1:  8 rows x 128 B (8 lanes per row)   -- full 128-byte lines
//   mode 2:  4 rows x 256 B
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/store_pattern.out tools/micro/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MODE, bool NT, bool LOAD>
__global__ __launch_bounds__(512) void k(unsigned short* out, const unsigned short* in, int M, int N, int ntm, int ntn) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = ntm * ntn;
  for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
    const int tm = id / ntn, tn = id - tm * ntn;
    // wave region: rows tm*256 + wr*128 .. +128; columns: two 32-column groups (mode 0) or one 64-column group (modes 1, 2)
    u32x4 acc = {(unsigned)id, (unsigned)lane, 3u, 4u};
    if (MODE == 0) {
      for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
          const int row = tm * 256 + wr * 128 + i * 16 + (lane & 15);
          const int col = tn * 256 + j * 128 + wc * 32 + (lane >> 4) * 8;
          if (row < M) {
            u32x4* p = reinterpret_cast<u32x4*>(out + (long long)row * N + col);
            if (LOAD) acc += *reinterpret_cast<const u32x4*>(in + (long long)row * N + col);
            if (NT) __builtin_nontemporal_store(acc, p); else *p = acc;
          }
        }
    } else if (MODE == 1) {
      for (int i = 0; i < 16; ++i) {
        const int row = tm * 256 + wr * 128 + i * 8 + (lane & 7);
        const int col = tn * 256 + wc * 64 + (lane >> 3) * 8;
        if (row < M) {
          u32x4* p = reinterpret_cast<u32x4*>(out + (long long)row * N + col);
          if (LOAD) acc += *reinterpret_cast<const u32x4*>(in + (long long)row * N + col);
          if (NT) __builtin_nontemporal_store(acc, p); else *p = acc;
        }
      }
    } else {
      for (int i = 0; i < 16; ++i) {   // 4 rows x 256 B: the wave covers 64 rows x 128 columns instead
        const int row = tm * 256 + (wave >> 1) * 64 + i * 4 + (lane & 3);
        const int col = tn * 256 + (wave & 1) * 128 + (lane >> 2) * 8;
        if (row < M) {
          u32x4* p = reinterpret_cast<u32x4*>(out + (long long)row * N + col);
          if (LOAD) acc += *reinterpret_cast<const u32x4*>(in + (long long)row * N + col);
          if (NT) __builtin_nontemporal_store(acc, p); else *p = acc;
        }
      }
    }
  }
}
int main() {
  const int M = 50432, N = 3072;
  unsigned short *o, *in;
  hipMalloc(&o, (size_t)M * N * 2); hipMalloc(&in, (size_t)M * N * 2);
  hipMemset(in, 1, (size_t)M * N * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int ntm = (M + 255) / 256, ntn = N / 256;
  auto run = [&](auto kern, const char* name) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, o, in, M, N, ntm, ntn);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    printf("%-34s %7.1f us  %6.2f TB/s (bytes of the matrix / time)\n", name, best * 1e3, (double)M * N * 2 / best / 1e9);
  };
  run(k<0, true, false>, "store 16 rows x 64 B, nontemporal");
  run(k<0, false, false>, "store 16 rows x 64 B, plain");
  run(k<1, true, false>, "store 8 rows x 128 B, nontemporal");
  run(k<1, false, false>, "store 8 rows x 128 B, plain");
  run(k<2, true, false>, "store 4 rows x 256 B, nontemporal");
  run(k<2, false, false>, "store 4 rows x 256 B, plain");
  run(k<0, true, true>, "load+store 16 x 64 B, nt");
  run(k<1, true, true>, "load+store 8 x 128 B, nt");
  run(k<2, true, true>, "load+store 4 x 256 B, nt");
  return 0;
}
