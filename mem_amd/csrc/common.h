// Shared host-side helpers for libmemhip.so (gfx950 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/memhip.h"

namespace memhip {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MEMHIP_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MEMHIP_OK;
}

// kernel-selection options (core.cpp; memhip_set_option)
enum { OPT_GEMM_P8, OPT_GEMM256, OPT_GEMM_SPLIT, OPT_GEMM_P8_HALF, OPT_GEMM_P8_MIN_N, OPT_GEMM256_MIN_N, OPT_TN_P8,
       OPT_RASTER_LDS, OPT_ATTN16, OPT_ATTN16_STAGGER, OPT_ATTN16_STAGGER_FWD, OPT_GEMM_STAGGER, OPT_GEMM_PREFETCH, OPT_RESERVE_CUS, OPT_LN_BWD_GRID, OPT_ATTN_WIN, OPT_GEMM_P8_PAIR, OPT_TN_GROUP, OPT_CONV_WAVES, OPT_RASTER_BANDS, OPT_COUNT_ };
int opt(int id);
// CUs the persistent one-workgroup-per-CU launches (GEMMs, attention) size their grids for: the device's CU count minus
// the `reserve_cus` option (a multiple of 8 is kept, so that every XCD gives up the same number).  The data-parallel
// reducer sets the option while gradient buckets are in flight: RCCL's channel kernels need CUs of their own -- a
// persistent grid that covers every CU would otherwise finish its last workgroups one full workgroup-duration late.
int usable_cus(hipStream_t s = nullptr);
int max_cus();     // all CUs, reservations ignored (workspace sizing)

inline hipStream_t as_stream(memhip_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace memhip

#define MEMHIP_REQUIRE(cond, ...) \
  do { if (!(cond)) return memhip::fail(MEMHIP_EINVAL, __VA_ARGS__); } while (0)
#define MEMHIP_HIP(call) \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return memhip::fail(MEMHIP_ELAUNCH, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
