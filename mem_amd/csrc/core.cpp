// ABI bookkeeping: version, arch, thread-local error string, kernel-selection options.
#include "common.h"
#include <atomic>
#include <cstring>

namespace memhip {
thread_local char g_err[512] = "";

// Kernel-selection switches for A/B measurements (tools/): set ONLY through memhip_set_option() -- the library
// reads no environment variable.  Defaults = the shipped dispatch.
static std::atomic<int> g_opt[OPT_COUNT_] = {
    /*gemm_p8*/ {1}, /*gemm256*/ {1}, /*gemm_split*/ {1}, /*gemm_p8_half*/ {1}, /*gemm_p8_min_n*/ {512},
    /*gemm256_min_n*/ {1024}, /*tn_p8*/ {1}, /*tn256*/ {1}, /*raster_lds*/ {1}, /*attn16*/ {1}, /*attn16_stagger (cycles)*/ {40000}, /*attn16_stagger_fwd*/ {0}, /*gemm_stagger (cycles per K-tile)*/ {0}, /*gemm_prefetch (epilogue-operand L2 prefetch; measured slower: the touches share the in-order vmcnt queue with the operand stream)*/ {0},
    /*reserve_cus*/ {0}, /*ln_bwd_grid (workgroups of the LayerNorm-backward kernels)*/ {768},
    /*gemm_p8s (256x128 tiles with a streamed epilogue: gemm_p8s.hip)*/ {0},
    /*gemm_p8d (256x256 tiles, stores deferred into the next tile: gemm_p8d.hip; measured slower: off)*/ {0}};
static const char* const g_opt_name[OPT_COUNT_] = {"gemm_p8", "gemm256", "gemm_split", "gemm_p8_half", "gemm_p8_min_n",
                                                   "gemm256_min_n", "tn_p8", "tn256", "raster_lds", "attn16", "attn16_stagger", "attn16_stagger_fwd", "gemm_stagger", "gemm_prefetch", "reserve_cus", "ln_bwd_grid", "gemm_p8s", "gemm_p8d"};
int opt(int id) { return g_opt[id].load(std::memory_order_relaxed); }

int usable_cus() {
  static std::atomic<int> device_cus{0};
  int n = device_cus.load(std::memory_order_relaxed);
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    device_cus.store(n, std::memory_order_relaxed);
  }
  int r = opt(OPT_RESERVE_CUS);
  r = r < 0 ? 0 : (r + 7) / 8 * 8;
  return n - r >= 8 ? n - r : (n >= 8 ? 8 : n);
}
}  // namespace memhip

extern "C" {
int memhip_abi_version(void) { return MEMHIP_ABI_VERSION; }
const char* memhip_last_error(void) { return memhip::g_err; }
const char* memhip_arch(void) { return "gfx950"; }

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
