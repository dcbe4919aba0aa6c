// Library identity and error plumbing for libsynchformer_hip.so.
#include "sf_common.h"
#include "../../include/synchformer_hip.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void sf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sf_abi_version(void) { return SF_ABI_VERSION; }
extern "C" const char* sf_last_error(void) { return g_err; }
extern "C" const char* sf_build_info(void) { return "libsynchformer_hip gfx950 (hipcc -O3, wave64, mfma_f32_16x16x32_bf16)"; }
