// augment.hip -- the per-anchor transforms of the AdaptPoint augmentor: from the imitator's nine numbers per anchor
// to a 3x3 matrix A = R diag(s) and an offset t (`AdaptPoint_Augmentor.local_transformaton`,
// openpoints/models_adaptpoint/generator_component4_15.py:236-297), forward and backward.  The reference (and this
// package's PyTorch form, adaptpoint_amd.augmentor.anchor_transforms_composed) spends ~40 elementwise launches on the
// B x 4 anchors forward and as many again backward; here one launch each way, one thread per anchor.
//   angles a = pi (tanh(p[0:3]) r_range) / 180 * keep[0]
//   scales s = (sigmoid(p[3:6]) (s_range - 1) + 1) * keep[1] * axes, a scale of 0 ("axis not scaled") becomes 1
//   offset t = tanh(p[6:9]) t_range * keep[2] * axes
//   R as the reference composes it (:288-290; its centre entry is sz sy sx + cz cy),  A[r][c] = R[r][c] s[c].
#include <hip/hip_runtime.h>

#include "../../include/adaptpoint_amd.h"
#include "apn_common.h"

namespace apn {

struct AnchorTerms {
    float th[3], sg[3], tt[3];       // tanh(p0..2), sigmoid(p3..5), tanh(p6..8)
    float sn[3], cs[3];              // sin / cos of the angles (x, y, z)
    float s[3];
    bool unit[3];                    // the scale was 0 and became 1
};

__device__ __forceinline__ AnchorTerms anchor_terms(const float *p, const float *keep, const float *axes, float r_range,
                                                    float s_range) {
    AnchorTerms a;
    const float kpi = 3.14159265358979323846f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a.th[i] = tanhf(p[i]);
        a.sg[i] = 1.0f / (1.0f + expf(-p[3 + i]));
        a.tt[i] = tanhf(p[6 + i]);
        const float ang = kpi * (a.th[i] * r_range) / 180.0f * keep[0];
        a.sn[i] = sinf(ang);
        a.cs[i] = cosf(ang);
        const float s = (a.sg[i] * (s_range - 1.0f) + 1.0f) * keep[1] * axes[i];
        a.unit[i] = s == 0.0f;
        a.s[i] = a.unit[i] ? 1.0f : s;
    }
    return a;
}

__device__ __forceinline__ void anchor_rotation(const AnchorTerms &a, float (&R)[9]) {
    const float sx = a.sn[0], sy = a.sn[1], sz = a.sn[2], cx = a.cs[0], cy = a.cs[1], cz = a.cs[2];
    R[0] = cz * cy; R[1] = cz * sy * sx - sz * cx; R[2] = cz * sy * cx + sz * sx;
    R[3] = sz * cy; R[4] = sz * sy * sx + cz * cy; R[5] = sz * sy * cx - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
}

__global__ __launch_bounds__(64) void anchor_transforms_kernel(int n, const float *__restrict__ prob,
                                                               const float *__restrict__ keep,
                                                               const float *__restrict__ axes, float r_range,
                                                               float s_range, float t_range, float *__restrict__ lin,
                                                               float *__restrict__ off) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float *p = prob + (size_t)i * 9, *k = keep + (size_t)i * 3, *ax = axes + (size_t)i * 3;
    const AnchorTerms a = anchor_terms(p, k, ax, r_range, s_range);
    float R[9];
    anchor_rotation(a, R);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) lin[(size_t)i * 9 + 3 * r + c] = R[3 * r + c] * a.s[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) off[(size_t)i * 3 + c] = a.tt[c] * t_range * k[2] * ax[c];
}

__global__ __launch_bounds__(64) void anchor_transforms_grad_kernel(int n, const float *__restrict__ prob,
                                                                    const float *__restrict__ keep,
                                                                    const float *__restrict__ axes, float r_range,
                                                                    float s_range, float t_range,
                                                                    const float *__restrict__ g_lin,
                                                                    const float *__restrict__ g_off,
                                                                    float *__restrict__ g_prob) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float *p = prob + (size_t)i * 9, *k = keep + (size_t)i * 3, *ax = axes + (size_t)i * 3;
    const AnchorTerms a = anchor_terms(p, k, ax, r_range, s_range);
    float R[9], gR[9], gs[3] = {0.0f, 0.0f, 0.0f};
    anchor_rotation(a, R);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = g_lin ? g_lin[(size_t)i * 9 + 3 * r + c] : 0.0f;
            gs[c] = __builtin_fmaf(g, R[3 * r + c], gs[c]);
            gR[3 * r + c] = g * a.s[c];
        }
    const float sx = a.sn[0], sy = a.sn[1], sz = a.sn[2], cx = a.cs[0], cy = a.cs[1], cz = a.cs[2];
    // gradients w.r.t. the six sines / cosines, entry by entry of anchor_rotation
    const float g_sx = gR[1] * cz * sy + gR[2] * sz + gR[4] * sz * sy - gR[5] * cz + gR[7] * cy;
    const float g_cx = -gR[1] * sz + gR[2] * cz * sy + gR[5] * sz * sy + gR[8] * cy;
    const float g_sy = gR[1] * cz * sx + gR[2] * cz * cx + gR[4] * sz * sx + gR[5] * sz * cx - gR[6];
    const float g_cy = gR[0] * cz + gR[3] * sz + gR[4] * cz + gR[7] * sx + gR[8] * cx;
    const float g_sz = -gR[1] * cx + gR[2] * sx + gR[3] * cy + gR[4] * sy * sx + gR[5] * sy * cx;
    const float g_cz = gR[0] * cy + gR[1] * sy * sx + gR[2] * sy * cx + gR[4] * cy - gR[5] * sx;
    const float g_ang[3] = {g_sx * cx - g_cx * sx, g_sy * cy - g_cy * sy, g_sz * cz - g_cz * sz};
    const float kpi = 3.14159265358979323846f;
    float *gp = g_prob + (size_t)i * 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gp[c] = g_ang[c] * (kpi * r_range / 180.0f * k[0]) * (1.0f - a.th[c] * a.th[c]);
        gp[3 + c] = a.unit[c] ? 0.0f : gs[c] * (s_range - 1.0f) * k[1] * ax[c] * a.sg[c] * (1.0f - a.sg[c]);
        const float go = g_off ? g_off[(size_t)i * 3 + c] : 0.0f;
        gp[6 + c] = go * t_range * k[2] * ax[c] * (1.0f - a.tt[c] * a.tt[c]);
    }
}

}  // namespace apn

extern "C" int apn_anchor_transforms(int n, const float *prob, const float *keep, const float *axes, float r_range,
                                     float s_range, float t_range, float *lin, float *off, void *stream) {
    using namespace apn;
    if (n < 0) return APN_EINVAL;
    if (n == 0) return APN_OK;
    if (!prob || !keep || !axes || !lin || !off) return APN_EINVAL;
    hipLaunchKernelGGL(anchor_transforms_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, prob, keep, axes,
                       r_range, s_range, t_range, lin, off);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_anchor_transforms_grad(int n, const float *prob, const float *keep, const float *axes, float r_range,
                                          float s_range, float t_range, const float *g_lin, const float *g_off,
                                          float *g_prob, void *stream) {
    using namespace apn;
    if (n < 0) return APN_EINVAL;
    if (n == 0) return APN_OK;
    if (!prob || !keep || !axes || !g_prob) return APN_EINVAL;
    hipLaunchKernelGGL(anchor_transforms_grad_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, prob, keep,
                       axes, r_range, s_range, t_range, g_lin, g_off, g_prob);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
