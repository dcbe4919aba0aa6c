// Library identity and error plumbing for libsynchformer_hip.so.
#include "sf_common.h"
#include "../../include/synchformer_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <set>
#include <utility>

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

static std::mutex g_prep_mutex;
static std::set<std::pair<const void*, int>> g_prepared;          // (kernel, device) whose dynamic-LDS limit has been raised
static int g_cus[64] = {0};

int sf_prepare_kernel(const void* kernel, int lds_bytes, const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { sf_set_error("%s: hipGetDevice failed", who); return -1; }
  std::lock_guard<std::mutex> lock(g_prep_mutex);
  if (g_prepared.count({kernel, dev})) return 0;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) { sf_set_error("%s: hipFuncSetAttribute(%d bytes of LDS): %s", who, lds_bytes, hipGetErrorString(e)); return (int)e; }
  g_prepared.insert({kernel, dev});
  return 0;
}

int sf_cu_count(const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { sf_set_error("%s: device query failed", who); return 0; }
  std::lock_guard<std::mutex> lock(g_prep_mutex);
  if (!g_cus[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { sf_set_error("%s: device query failed", who); return 0; }
    g_cus[dev] = prop.multiProcessorCount;
  }
  return g_cus[dev];
}
