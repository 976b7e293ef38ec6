// Library-level entry points: version, error strings, load-time options.
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace fdm {
int g_last_hip_error = 0;
}

namespace {
struct Init {
  Init() {
    const char* e = std::getenv("FDM_FP8_SW_CVT");
    fdm::g_fp8_sw_cvt = (e != nullptr && std::strcmp(e, "0") != 0);
  }
} g_init;
}  // namespace

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
