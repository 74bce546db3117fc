// apn_common.h -- shared device helpers for the gfx950 kernels.
//
// Build contract: every translation unit is compiled with -ffp-contract=off, so
// the only fused multiply-adds in the numerics are the explicit __builtin_fmaf
// calls below.  That pins the squared-distance rounding to ONE documented form
// (the reference leaves it to nvcc's contraction: sampling_gpu.cu:140,
// ball_query_gpu.cu:39, interpolate_gpu.cu:42):
//
//      d2 = fma(dz, dz, fma(dx, dx, dy * dy))
//
// which is what LLVM's fadd-of-fmul contraction produces for
// (dx*dx + dy*dy) + dz*dz (oracle variant APO_DIST_FMA_XY).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/adaptpoint_amd.h"

#define APN_WAVE 64

namespace apn {

__device__ __forceinline__ float dist2(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}

// DPP controls (gfx9 encoding).
enum : int {
    DPP_QUAD_XOR1 = 0xB1,        // quad_perm:[1,0,3,2]
    DPP_QUAD_XOR2 = 0x4E,        // quad_perm:[2,3,0,1]
    DPP_ROW_HALF_MIRROR = 0x141,
    DPP_ROW_MIRROR = 0x140,
    DPP_ROW_BCAST15 = 0x142,
    DPP_ROW_BCAST31 = 0x143,
};

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    // old = 0 is the identity of umax: lanes that receive no data (masked rows)
    // keep their own value, and the DPP combiner can fold this into v_max_u32_dpp.
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
    return o > v ? o : v;
}

// Max over the 64 lanes of a wave; the result is valid in lane 63 and is
// returned wave-uniform.  Order-free (integer max), so no tie rule lives here.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_max_u32<DPP_QUAD_XOR1>(v);
    v = dpp_max_u32<DPP_QUAD_XOR2>(v);
    v = dpp_max_u32<DPP_ROW_HALF_MIRROR>(v);
    v = dpp_max_u32<DPP_ROW_MIRROR>(v);
    v = dpp_max_u32<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_max_u32<DPP_ROW_BCAST31, 0xC>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// copies of the register-resident backward pass's dL/dW2 accumulators (sa_fused.hip: sa_bwd_kernel's tail;
// cleared by bwd_consts2, added up by bwd_consts1 in sa_glue.hip)


// ---------------------------------------------------------------------------------------------
// Order-independent accumulators ("acc sets") for the per-channel sums that cross workgroups
// inside the fused set-abstraction chain (BatchNorm statistics and their backward terms).
// A producer workgroup adds its float partial sum v with TWO 64-bit integer atomics,
//     hi += floor(v)            lo += (v - floor(v)) * 2^52            (value = hi + lo * 2^-52),
// so the total is exact for every addend (|v| < 2^62, resolution 2^-52 absolute) and does not
// depend on the order the atomics retire in: the consumer kernels read the SAME bits run after run
// without a fold launch between producer and consumer.  ACC_COPIES copies per set spread the
// same-address traffic (workgroup w adds into copy w % ACC_COPIES; 500-800 workgroups on one set of
// addresses serialise for microseconds); the consumer adds the copies.  Cell ncol of a copy is a
// flag: a non-finite or out-of-range addend bumps it and the consumer returns NaN (loud, not silent).
// Layout: set[copy][ncol + 1][2] unsigned 64-bit words, caller- (i.e. earlier-kernel-) zeroed.
constexpr int ACC_COPIES = 8;
__host__ __device__ constexpr int acc_words(int ncol) { return ACC_COPIES * (ncol + 1) * 2; }

__device__ __forceinline__ void acc_add(unsigned long long *set, int ncol, int copy, int col, float v) {
    unsigned long long *cell = set + ((size_t)copy * (ncol + 1) + col) * 2;
    const double dv = (double)v;
    if (!(__builtin_fabs(dv) < 4611686018427387904.0)) {      // NaN, inf or >= 2^62
        atomicAdd(set + ((size_t)copy * (ncol + 1) + ncol) * 2, 1ull);
        return;
    }
    const double dh = __builtin_floor(dv);
    atomicAdd(cell, (unsigned long long)(long long)dh);
    atomicAdd(cell + 1, (unsigned long long)((dv - dh) * 4503599627370496.0));
}

// the total of column `col` over the copies (NaN if any copy's flag is set)
__device__ __forceinline__ double acc_read(const unsigned long long *set, int ncol, int col) {
    long long hi = 0;
    unsigned long long lo = 0, bad = 0;
#pragma unroll
    for (int k = 0; k < ACC_COPIES; ++k) {
        const unsigned long long *cp = set + (size_t)k * (ncol + 1) * 2;
        hi += (long long)cp[2 * col];
        lo += cp[2 * col + 1];
        bad |= cp[2 * ncol];
    }
    if (bad) return __builtin_nan("");
    return (double)hi + (double)lo * (1.0 / 4503599627370496.0);
}

// Instruction-issue priority of the CRITICAL stream's waves (s_setprio 0..3, default 0): the seven launches of the fused
// set-abstraction step raise it at entry, so that on a SIMD they share with the index stream's waves (FPS: five always-ready
// waves per SIMD in a tight VALU + LDS loop; the ball query: the chip's wave slots full of distance tests) the arbiter issues
// THEIR ready instructions first.  Inside one kernel every wave has the same priority: nothing changes there.  Measured
// (profiles/r05_index_interference_*.txt): the MLP-stream kernels were 2-3x slower beside the sampler (issue slots, not
// registers: the sampler holds 19 VGPRs per wave).
#ifndef APN_MLP_PRIO
#define APN_MLP_PRIO 3
#endif
__device__ __forceinline__ void critical_stream_priority() {
    if (APN_MLP_PRIO > 0) __builtin_amdgcn_s_setprio(APN_MLP_PRIO);
}

__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}


// Raising a kernel's dynamic-LDS limit is a property of the CURRENT DEVICE's copy of the function: the "done" flag is
// kept per device (bit d of the mask), so a process that launches on a second GPU sets the attribute there too; the
// atomic mask makes the first launches of two host threads safe (setting the attribute twice is harmless).
struct DynLdsOnce {
    unsigned long long mask = 0ull;
};
inline hipError_t set_dyn_lds(DynLdsOnce &once, const void *fn, int bytes) {
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && ((__atomic_load_n(&once.mask, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) return e;
    if (tracked) __atomic_fetch_or(&once.mask, 1ull << dev, __ATOMIC_RELEASE);
    return hipSuccess;
}

}  // namespace apn

#define APN_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)
