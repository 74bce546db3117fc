// capi.hip -- library-level entry points of the C ABI (include/adaptpoint_amd.h).
#include "apn_common.h"

extern "C" int apn_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *apn_error_string(int code) {
    if (code == APN_OK) return "success";
    if (code == APN_EINVAL) return "invalid argument (negative size, null pointer or size beyond the launch limits)";
    return hipGetErrorString((hipError_t)code);
}

// Diagnostic: one thread writes the device's constant-rate wall clock (100 MHz) to stamps[slot].  A launch like any
// other, so it can be captured into a hipGraph: phase boundaries of a replayed step, on every branch of the graph,
// without a profiler in the way (rocprofv3's kernel trace runs the branches of a graph one after the other).
namespace apn {
__global__ void stamp_kernel(unsigned long long *stamps, int slot) { stamps[slot] = wall_clock64(); }
}

extern "C" int apn_debug_stamp(void *stamps, int slot, void *stream) {
    if (stamps == nullptr || slot < 0) return APN_EINVAL;
    hipLaunchKernelGGL(apn::stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long *)stamps, slot);
    return (int)hipGetLastError();
}
