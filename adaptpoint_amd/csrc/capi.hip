// capi.hip -- library-level entry points of the C ABI (include/adaptpoint_amd.h).
#include "apn_common.h"

extern "C" int apn_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *apn_error_string(int code) {
    if (code == APN_OK) return "success";
    if (code == APN_EINVAL) return "invalid argument (negative size, null pointer or size beyond the launch limits)";
    return hipGetErrorString((hipError_t)code);
}
