// ABI bookkeeping: version, arch, thread-local error string, kernel-selection options.
#include "common.h"
#include <atomic>
#include <mutex>
#include <cstring>

namespace memhip {
thread_local char g_err[512] = "";

// Kernel-selection switches for A/B measurements (tools/): set ONLY through memhip_set_option() -- the library
// reads no environment variable.  Defaults = the shipped dispatch.
static std::atomic<int> g_opt[OPT_COUNT_] = {
    /*gemm_p8*/ {1}, /*gemm256*/ {1}, /*gemm_split*/ {1}, /*gemm_p8_half*/ {1}, /*gemm_p8_min_n*/ {512},
    /*gemm256_min_n*/ {1024}, /*tn_p8*/ {1}, /*raster_lds*/ {1}, /*attn16*/ {1}, /*attn16_stagger (cycles)*/ {40000}, /*attn16_stagger_fwd*/ {0}, /*gemm_stagger (cycles per K-tile)*/ {0}, /*gemm_prefetch (epilogue-operand L2 prefetch; measured slower: the touches share the in-order vmcnt queue with the operand stream)*/ {0},
    /*reserve_cus*/ {0}, /*ln_bwd_grid (workgroups of the LayerNorm-backward kernels; re-swept in round 5 inside the two-stream step: 768 -> 2048 = -0.25 ms)*/ {2048},
    /*attn_win (slot-layout streaming attention for windows 40 / 20 wide, attn_win.hip; 0 = attn_stream.hip)*/ {1},
    /*gemm_p8_pair (full rounds + ragged round of an NT product in ONE launch; 0 = two launches)*/ {1},
    /*tn_group (memhip_gemm_bf16_tn_group runs its products as ONE grid with one split count; 0 = one by one)*/ {1},
    /*conv_waves (fp16x2 tokenizer convolutions: 16 = eight waves per workgroup and the phase-interleaved 256 x 128 tile where its grid fills the chip twice, 8 = eight waves, 128 x 128 tiles only, 4 = four waves)*/ {16},
    /*raster_bands (bands per sample of the two-pass rasterizer; 0 = chosen from the batch size, see memhip_rasterize_binned_f64)*/ {0}};
static const char* const g_opt_name[OPT_COUNT_] = {"gemm_p8", "gemm256", "gemm_split", "gemm_p8_half", "gemm_p8_min_n",
                                                   "gemm256_min_n", "tn_p8", "raster_lds", "attn16", "attn16_stagger", "attn16_stagger_fwd", "gemm_stagger", "gemm_prefetch", "reserve_cus", "ln_bwd_grid", "attn_win", "gemm_p8_pair", "tn_group", "conv_waves", "raster_bands"};
int opt(int id) { return g_opt[id].load(std::memory_order_relaxed); }

// CU reservations per stream (memhip_stream_reserve_cus): a small fixed table, keyed by the caller's stream handle
struct StreamReserve { std::atomic<bool> used{false}; std::atomic<void*> stream{nullptr}; std::atomic<int> cus{0}; };
static StreamReserve g_reserve[32];
static int stream_reserved(hipStream_t s) {
  for (auto& r : g_reserve)
    if (r.used.load(std::memory_order_acquire) && r.stream.load(std::memory_order_relaxed) == (void*)s)
      return r.cus.load(std::memory_order_relaxed);
  return 0;
}

static int device_cu_count() {
  static std::atomic<int> device_cus{0};
  int n = device_cus.load(std::memory_order_relaxed);
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    device_cus.store(n, std::memory_order_relaxed);
  }
  return n;
}

// every CU of the device, no reservation applied: workspace queries size for this (the largest grid any stream can plan)
int max_cus() { return device_cu_count(); }

int usable_cus(hipStream_t s) {
  const int n = device_cu_count();
  if (!n) return 0;
  int r = opt(OPT_RESERVE_CUS);                       // A/B tools: every stream
  const int rs = stream_reserved(s);                   // the caller's reservation for launches on THIS stream
  r = r > rs ? r : rs;
  r = r < 0 ? 0 : (r + 7) / 8 * 8;
  return n - r >= 8 ? n - r : (n >= 8 ? 8 : n);
}
}  // namespace memhip

extern "C" {
int memhip_abi_version(void) { return MEMHIP_ABI_VERSION; }
const char* memhip_last_error(void) { return memhip::g_err; }
const char* memhip_arch(void) { return "gfx950"; }
#ifndef MEMHIP_BUILD_FLAGS
#define MEMHIP_BUILD_FLAGS ""
#endif
const char* memhip_build_flags(void) { return MEMHIP_BUILD_FLAGS; }

int memhip_stream_reserve_cus(memhip_stream_t stream, int cus) {
  MEMHIP_REQUIRE(cus >= 0, "stream_reserve_cus: negative count");
  using memhip::g_reserve;
  static std::mutex mu;                                 // writers only (a few calls per step); readers scan lock-free
  std::lock_guard<std::mutex> lock(mu);
  for (auto& r : g_reserve)
    if (r.used.load(std::memory_order_relaxed) && r.stream.load(std::memory_order_relaxed) == stream) {
      r.cus.store(cus, std::memory_order_relaxed);
      if (cus == 0) r.used.store(false, std::memory_order_release);
      return MEMHIP_OK;
    }
  if (cus == 0) return MEMHIP_OK;
  for (auto& r : g_reserve)
    if (!r.used.load(std::memory_order_relaxed)) {
      r.stream.store(stream, std::memory_order_relaxed);
      r.cus.store(cus, std::memory_order_relaxed);
      r.used.store(true, std::memory_order_release);
      return MEMHIP_OK;
    }
  return memhip::fail(MEMHIP_EINVAL, "stream_reserve_cus: more than 32 streams carry a reservation");
}

int memhip_set_option(const char* name, int value) {
  MEMHIP_REQUIRE(name, "set_option: null name");
  for (int i = 0; i < memhip::OPT_COUNT_; ++i)
    if (!strcmp(name, memhip::g_opt_name[i])) {
      memhip::g_opt[i].store(value, std::memory_order_relaxed);
      return MEMHIP_OK;
    }
  return memhip::fail(MEMHIP_EINVAL, "set_option: unknown option '%s'", name);
}

int memhip_get_option(const char* name, int* value) {
  MEMHIP_REQUIRE(name && value, "get_option: null argument");
  for (int i = 0; i < memhip::OPT_COUNT_; ++i)
    if (!strcmp(name, memhip::g_opt_name[i])) {
      *value = memhip::opt(i);
      return MEMHIP_OK;
    }
  return memhip::fail(MEMHIP_EINVAL, "get_option: unknown option '%s'", name);
}
}
