// Probe: how fast can the 2 GB event stream of the config #4 rasterizer (64 x 1 M rows of 32 B) be READ, by access shape?
//   shape 0: the kernel's shape -- a lane owns one 32-byte row, two 16-byte loads at a 32-byte lane stride
//   shape 1: contiguous -- a wave-instruction covers 1 KiB, lane l loads bytes [16 l, 16 l + 16) (a row is split over a lane pair)
//   shape 2: shape 1 with nontemporal loads
//   shape 3: the same KiB per wave-instruction, nontemporal, but lanes 0-31 take the (x, y) halves of 32 rows and lanes 32-63 their (t, p) halves
//            (the lane order a v_permlane32_swap exchange wants)
// Build: hipcc -O3 --offload-arch=gfx950 -o variants/rrp tools/exp/r06_raster_read_probe.hip ; run: variants/rrp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE, int THREADS, int EPT>
__global__ __launch_bounds__(THREADS) void read_probe(const d2v* __restrict__ ev, long long nrows, unsigned int* sink) {
  constexpr int CHUNK = THREADS * EPT;
  const long long nchunks = nrows / CHUNK;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double acc = 0.0;
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    d2v r[2 * EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (SHAPE == 0) {
        const long long row = c * CHUNK + (long long)k * THREADS + tid;
        r[2 * k] = ev[2 * row];
        r[2 * k + 1] = ev[2 * row + 1];
      } else {
        // the wave's 64 rows of step k = 2 KiB = 128 double2: lane l takes element l and 64 + l
        const long long base = 2 * (c * CHUNK + (long long)k * THREADS + wave * 64);
        if (SHAPE == 3) {
          const int e = lane & 31, hf = lane >> 5;
          r[2 * k] = __builtin_nontemporal_load(ev + base + 2 * e + hf);
          r[2 * k + 1] = __builtin_nontemporal_load(ev + base + 64 + 2 * e + hf);
        } else if (SHAPE == 2) {
          r[2 * k] = __builtin_nontemporal_load(ev + base + lane);
          r[2 * k + 1] = __builtin_nontemporal_load(ev + base + 64 + lane);
        } else {
          r[2 * k] = ev[base + lane];
          r[2 * k + 1] = ev[base + 64 + lane];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * EPT; ++k) acc += r[k].x + r[k].y;
  }
  if (acc == 123456.789) sink[0] = 1;
}

template <int SHAPE, int THREADS, int EPT>
static void run(const char* name, const d2v* ev, long long nrows, unsigned int* sink, int grid) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f, sum = 0.f;
  const int reps = 12;
  for (int i = 0; i < reps + 2; ++i) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((read_probe<SHAPE, THREADS, EPT>), dim3(grid), dim3(THREADS), 0, 0, ev, nrows, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (i >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  const double gb = (double)nrows * 32 / 1e9;
  printf("%-44s grid %5d  avg %.1f us = %.2f TB/s   best %.1f us = %.2f TB/s\n", name, grid, sum / reps * 1e3, gb / (sum / reps), best * 1e3,
         gb / best);
}

int main() {
  const long long nrows = 64ll << 20;
  d2v* ev; unsigned int* sink;
  CK(hipMalloc(&ev, nrows * 32)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(ev, 0, nrows * 32));
  for (int grid : {1024, 4096}) {
    run<0, 512, 8>("rows per lane, 512 thr x 8 rows", ev, nrows, sink, grid);
    run<1, 512, 8>("contiguous, 512 thr x 8 rows", ev, nrows, sink, grid);
    run<2, 512, 8>("contiguous nt, 512 thr x 8 rows", ev, nrows, sink, grid);
    run<3, 512, 8>("contiguous nt, halves by lane half", ev, nrows, sink, grid);
    run<0, 512, 4>("rows per lane, 512 thr x 4 rows", ev, nrows, sink, grid);
    run<1, 512, 4>("contiguous, 512 thr x 4 rows", ev, nrows, sink, grid);
    run<0, 256, 8>("rows per lane, 256 thr x 8 rows", ev, nrows, sink, grid);
    run<1, 256, 8>("contiguous, 256 thr x 8 rows", ev, nrows, sink, grid);
    run<1, 1024, 4>("contiguous, 1024 thr x 4 rows", ev, nrows, sink, grid);
    run<1, 256, 16>("contiguous, 256 thr x 16 rows", ev, nrows, sink, grid);
  }
  return 0;
}
