// apn_mfma.h -- the bf16 MFMA idioms shared by the fused kernels (sa_fused.hip, attention.hip):
// v_mfma_f32_32x32x16_bf16 with plain or split (hi + lo, "bf16x3") operands.
//
// Operand / accumulator lane maps of v_mfma_f32_32x32x16_bf16 (wave64, lane l: r = l & 31,
// h = l >> 5):  A: lane holds row r, k = 8h .. 8h+7;  B: lane holds column r, k = 8h .. 8h+7;
// C/D: lane holds column r, register i <-> row acc_row(i, h).
#pragma once
#include "apn_common.h"

namespace apn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Operand precision.  NS = 1: operands rounded to bf16 (8 significant bits).  NS = 2: every
// f32 operand is split into hi + lo bf16 parts (16 significant bits) and a product is three
// MFMAs, hi*hi + hi*lo + lo*hi ("bf16x3"; the dropped lo*lo term is 2^-18 relative): the
// contraction then agrees with an fp32 one to ~1e-5 at 3x the (small) MFMA cost.
template <int NS>
struct Frag {
    bf16x8 p[NS];
};

template <int NS>
__device__ __forceinline__ Frag<NS> make_frag(const float (&v)[8]) {
    Frag<NS> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hi = (__bf16)v[j];
        o.p[0][j] = hi;
        if (NS == 2) o.p[NS - 1][j] = (__bf16)(v[j] - (float)hi);
    }
    return o;
}

template <int NS>
__device__ __forceinline__ Frag<NS> pack8(const f32x16 &v, int base) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = v[base + j];
    return make_frag<NS>(t);
}

template <int NS>
__device__ __forceinline__ f32x16 mfma(const Frag<NS> &a, const Frag<NS> &b, f32x16 c) {
    if (NS == 2) {                       // small terms first, the hi*hi term last
        c = mfma(a.p[NS - 1], b.p[0], c);
        c = mfma(a.p[0], b.p[NS - 1], c);
    }
    return mfma(a.p[0], b.p[0], c);
}

}  // namespace apn
