// Library-level entry points of include/mmr.h (version, error strings).
#include "common.hpp"

#include <string.h>

namespace mmr {
static thread_local char g_hip_err[256] = "";
void set_hip_error(hipError_t e)
{
    const char* s = hipGetErrorString(e);
    strncpy(g_hip_err, s ? s : "unknown HIP error", sizeof(g_hip_err) - 1);
    g_hip_err[sizeof(g_hip_err) - 1] = 0;
}
}  // namespace mmr

extern "C" int mmr_version(void) { return 100; }

extern "C" const char* mmr_error_string(int code)
{
    switch (code) {
        case MMR_OK: return "ok";
        case MMR_EINVAL: return "invalid argument (shape, pointer or unsupported combination)";
        case MMR_EHIP: return "HIP runtime error";
        case MMR_EUNSUPPORTED: return "not supported by this build";
        default: return "unknown error code";
    }
}

extern "C" const char* mmr_last_hip_error(void) { return mmr::g_hip_err; }
