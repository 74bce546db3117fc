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

// Diagnostic (scripts/debug_vgpr_hold.py): a long-running kernel that only HOLDS values -- 24 registers per lane, kept
// live by opaque asm statements, a workgroup barrier and one LDS atomic per turn (the rhythm of the FPS step) -- and checks
// them at the end.  bad[i] counts lanes whose i-th value changed; bad[24 + k] (k < 8) samples of (index << 32 | value).
namespace apn {
__global__ __launch_bounds__(256) void vgpr_hold_kernel(int turns, unsigned long long *bad) {
    __shared__ unsigned long long slot[4];
    const unsigned lane = threadIdx.x;
    unsigned r[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) r[i] = 0x10000u * (unsigned)(i + 1) + lane * 131u + blockIdx.x;
    if (lane < 4) slot[lane] = 0ull;
    __syncthreads();
    for (int t = 0; t < turns; ++t) {
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("" : "+v"(r[i]));
        if ((lane & 63) == (unsigned)(t & 63))
            atomicMax(&slot[t & 3], ((unsigned long long)r[0] << 32) | (unsigned)t);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned k = (unsigned)slot[t & 3];
        if (k == 0xFFFFFFFFu) r[23] ^= 1u;                     // (never true: keeps the read alive)
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const unsigned want = 0x10000u * (unsigned)(i + 1) + lane * 131u + blockIdx.x;
        if (r[i] != want) {
            const unsigned long long n = atomicAdd(&bad[i], 1ull);
            if (n < 8) bad[24 + (i & 7)] = ((unsigned long long)i << 32) | r[i];
        }
    }
}
}  // namespace apn

extern "C" int apn_debug_vgpr_hold(int blocks, int turns, unsigned long long *bad, void *stream) {
    if (blocks <= 0 || turns < 0 || !bad) return APN_EINVAL;
    hipLaunchKernelGGL(apn::vgpr_hold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, turns, bad);
    return (int)hipGetLastError();
}
