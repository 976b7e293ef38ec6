// symbols the library's other translation units provide (attention.hip alone is built into each A/B variant .so)
#include "../../fastdm_amd/csrc/common.h"
namespace fdm { thread_local int g_last_hip_error = 0; int device_cu_count() { return 256; } }
