// Library-level entry points: version, error strings, load-time options.
#include <cstdlib>
#include <cstring>
#include "common.h"

#include <atomic>

namespace fdm {
thread_local int g_last_hip_error = 0;

int device_cu_count() {
  constexpr int MAX_DEV = 64;
  static std::atomic<int> cache[MAX_DEV];   // zero-initialised; 0 = not queried yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < MAX_DEV) {
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int cu = 0;
  if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
  if (dev >= 0 && dev < MAX_DEV) cache[dev].store(cu, std::memory_order_relaxed);
  return cu;
}
}  // namespace fdm

#ifdef FDM_DEBUG_BUILD
namespace {
struct Init {
  Init() {
    const char* e = std::getenv("FDM_FP8_SW_CVT");
    fdm::g_fp8_sw_cvt = (e != nullptr && std::strcmp(e, "0") != 0);
  }
} g_init;
}  // namespace
#endif

extern "C" {

int fdm_version(void) { return 100; }

const char* fdm_error_string(int code) {
  switch (code) {
    case FDM_OK: return "ok";
    case FDM_ERR_INVALID: return "invalid argument (null pointer, negative size, or misaligned pointer/stride)";
    case FDM_ERR_UNSUPPORTED: return "unsupported shape or mode";
    case FDM_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
  }
}

int fdm_abi_version(void) { return FDM_ABI_VERSION; }

const char* fdm_last_hip_error_string(void) { return hipGetErrorString((hipError_t)fdm::g_last_hip_error); }

}  // extern "C"
