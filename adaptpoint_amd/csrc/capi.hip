// capi.hip -- library-level entry points of the C ABI (include/adaptpoint_amd.h).
#include "apn_common.h"

extern "C" int apn_version(void) { return 100; /* 0.1.0 */ }

// The compiler flags this library was built with (adaptpoint_amd/build.py stamps them in): the loader refuses a build
// without -fno-slp-vectorize / -ffp-contract=off, the two flags the results depend on.
#ifndef APN_BUILD_FLAGS
#define APN_BUILD_FLAGS "unknown"
#endif
extern "C" const char *apn_build_flags(void) { return APN_BUILD_FLAGS; }

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

// Diagnostic (scripts/debug_vgpr_hold.py vpk): the instruction form the SLP vectoriser had put into the FPS step,
//     v_pk_add_f32 d, a, pair  op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]        (d.lo = a.lo - pair.lo, d.hi = a.hi - pair.lo)
// with pair.lo rewritten every turn (a v_mov just ahead of it, as there) and pair.hi a LIVE unrelated value, in a loop that
// checks every result.  bad[0] = results whose high half is wrong, bad[1] = of those, the ones that equal a.hi - pair.hi
// (the other half used), bad[2] = low half wrong, bad[3] = results checked.
namespace apn {
typedef float vpk_f2 __attribute__((ext_vector_type(2)));
// FORM 0: `op_sel_hi:[1,0]` -- the pair's LOW register feeds both lanes (round 3's suspect; 1.6e11 results, no error).
// FORM 1: `op_sel:[0,1]` -- the pair's HIGH register feeds both lanes: the form round 4's instruction-by-instruction edits of
// the vectorised builds blame (profiles/r04_packed_fp32_op_sel.md); the operand is moved into the HIGH half, the low half live.
template <int FORM>
__global__ __launch_bounds__(256) void vpk_probe_kernel(int turns, unsigned long long *bad) {
    __shared__ unsigned long long slot[4];
    const unsigned lane = threadIdx.x;
    const float a0 = 0.25f + 0.001f * (float)lane, a1 = -0.5f + 0.002f * (float)lane;
    const float other = 1000.0f + (float)lane;                 // what the pair's other half holds (a live value)
    unsigned long long wrong_hi = 0, other_half = 0, wrong_lo = 0, seen = 0;
    if (lane < 4) slot[lane] = 0ull;
    __syncthreads();
    for (int t = 0; t < turns; ++t) {
        // the operand arrives as in the FPS step: one lane leaves it in an LDS table entry, every lane reads the entry back
        // with a 96-bit broadcast read behind a barrier, and the third word is moved into the pair's low half
        __shared__ float4 entry[2];
        const float cw = 0.001f * (float)(t & 1023) + 0.01f * (float)(blockIdx.x & 7);
        if (lane == (unsigned)(t & 255)) entry[t & 1] = make_float4(cw + 1.0f, cw + 2.0f, cw, 0.0f);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float4 e = entry[t & 1];
        const float c = e.z;
        vpk_f2 a = {a0, a1}, pair, d;
        float lo = 0.0f, hi = other;
        asm volatile("" : "+v"(hi));
        asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(c));
        if (FORM == 0) {
            pair = (vpk_f2){lo, hi};
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(pair));
        } else if (FORM == 1) {
            pair = (vpk_f2){hi, lo};
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(pair));
        } else {                                                  // FORM 2: form 1 behind eight idle cycles
            pair = (vpk_f2){hi, lo};
            asm volatile("s_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(pair));
        }
        const float want_lo = a0 - cw, want_hi = a1 - cw;
        if (e.x != cw + 1.0f || e.y != cw + 2.0f) ++wrong_lo;     // (the read itself)
        if (FORM == 0) {
            if (d.y != want_hi) { ++wrong_hi; if (d.y == a1 - hi) ++other_half; }
            if (d.x != want_lo) ++wrong_lo;
        } else {                                                  // (here the LOW lane is the one that crosses halves)
            if (d.x != want_lo) {
                if (wrong_hi == 0) {                              // a few samples: {turn, block, lane}, {found, wanted}, {operand, other}
                    const unsigned long long k = atomicAdd(&bad[7], 1ull);
                    if (k < 8) {
                        bad[8 + 3 * k] = ((unsigned long long)t << 32) | (blockIdx.x << 8) | lane;
                        bad[9 + 3 * k] = ((unsigned long long)__float_as_uint(d.x) << 32) | __float_as_uint(want_lo);
                        bad[10 + 3 * k] = ((unsigned long long)__float_as_uint(c) << 32) | __float_as_uint(d.y);
                    }
                }
                ++wrong_hi;
                if (d.x == a0 - hi) ++other_half;
            }
            if (d.y != want_hi) ++wrong_lo;
        }
        ++seen;
        if ((t & 7) == 0) {                                    // the FPS step's rhythm: an LDS atomic and a barrier now and then
            if ((lane & 63) == (unsigned)(t & 63)) atomicMax(&slot[t & 3], (unsigned long long)t);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    if (wrong_hi) atomicAdd(&bad[0], wrong_hi);
    if (other_half) atomicAdd(&bad[1], other_half);
    if (wrong_lo) atomicAdd(&bad[2], wrong_lo);
    atomicAdd(&bad[3], seen);
}
}  // namespace apn

extern "C" int apn_debug_vpk_probe(int blocks, int turns, int form, unsigned long long *bad, void *stream) {
    if (blocks <= 0 || turns < 0 || !bad || form < 0 || form > 2) return APN_EINVAL;
    if (form == 0)
        hipLaunchKernelGGL(apn::vpk_probe_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, turns, bad);
    else if (form == 1)
        hipLaunchKernelGGL(apn::vpk_probe_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, turns, bad);
    else
        hipLaunchKernelGGL(apn::vpk_probe_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, turns, bad);
    return (int)hipGetLastError();
}
