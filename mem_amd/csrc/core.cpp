// ABI bookkeeping: version, arch, thread-local error string.
#include "common.h"

namespace memhip { thread_local char g_err[512] = ""; }

extern "C" {
int memhip_abi_version(void) { return MEMHIP_ABI_VERSION; }
const char* memhip_last_error(void) { return memhip::g_err; }
const char* memhip_arch(void) { return "gfx950"; }
}
