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

// ------------------------------------------------------------------------------------------------------------------
// The deformation itself (generator_component4_15.py:156-167, 204-232, 313-327, and the mask at :180): per cloud,
//   w_mn = exp(-|(a_m - x_n) o axes|^2 / (2 sigma^2)),   z_n = sum_m w_mn ((x_n - a_m) A_m + t_m + a_m) / sum_m w_mn,
//   mu = mean_n z_n,  r = max_n |z_n - mu|,  out_n = (z_n - mu) * (0.999999 / r) * mask_n
// -- kernel regression, blend, unit sphere and mask, ~30 PyTorch launches forward and ~50 backward -- as one workgroup
// per cloud, forward and backward one launch each.  Gradients go to A (B,M,3,3), t (B,M,3) and the mask (B,N); the
// input cloud and the anchors (its own points) carry none in the generator step, the weights w depend on them alone.
constexpr int DF_MAXM = 8;        // anchors per cloud
constexpr int DF_MAXP = 16;       // points per thread: n <= 4096

__device__ __forceinline__ float df_block_sum(float v, float *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

struct DeformShared {
    float a[DF_MAXM][3], lin[DF_MAXM][9], off[DF_MAXM][3], ax[3];
};

__device__ __forceinline__ void df_load_shared(DeformShared &sh, int b, int M, const float *anchors, const float *lin,
                                               const float *off, const float *axes) {
    const int t = threadIdx.x;
    if (t < M * 3) { sh.a[t / 3][t % 3] = anchors[((size_t)b * M) * 3 + t]; sh.off[t / 3][t % 3] = off ? off[((size_t)b * M) * 3 + t] : 0.0f; }
    if (t < M * 9 && lin) sh.lin[t / 9][t % 9] = lin[((size_t)b * M) * 9 + t];
    if (t < 3) sh.ax[t] = axes[(size_t)b * 3 + t];
    __syncthreads();
}

// weights of point x towards the anchors, normalised: wn[m] = w_m / sum w
__device__ __forceinline__ void df_weights(const DeformShared &sh, int M, const float (&x)[3], float inv2s2, float (&wn)[DF_MAXM]) {
    float ws = 0.0f;
#pragma unroll
    for (int m = 0; m < DF_MAXM; ++m) {
        wn[m] = 0.0f;
        if (m < M) {
            float d2 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float d = (sh.a[m][c] - x[c]) * sh.ax[c]; d2 = __builtin_fmaf(d, d, d2); }
            wn[m] = expf(-d2 * inv2s2);
            ws += wn[m];
        }
    }
    const float inv = 1.0f / ws;
#pragma unroll
    for (int m = 0; m < DF_MAXM; ++m) wn[m] *= inv;
}

__global__ __launch_bounds__(256) void deform_forward_kernel(int N, int M, const float *__restrict__ xyz,
                                                             const float *__restrict__ anchors,
                                                             const float *__restrict__ lin, const float *__restrict__ off,
                                                             const float *__restrict__ axes, const float *__restrict__ mask,
                                                             float inv2s2, float *__restrict__ z_out,
                                                             float *__restrict__ stat, float *__restrict__ out) {
    __shared__ DeformShared sh;
    __shared__ float scratch[4];
    __shared__ unsigned long long best[4];
    const int b = blockIdx.x, t = threadIdx.x;
    df_load_shared(sh, b, M, anchors, lin, off, axes);
    float z[DF_MAXP][3];
    float sum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < DF_MAXP; ++i) {
        const int n = t + 256 * i;
        z[i][0] = z[i][1] = z[i][2] = 0.0f;
        if (n < N) {
            const float *px = xyz + ((size_t)b * N + n) * 3;
            const float x[3] = {px[0], px[1], px[2]};
            float wn[DF_MAXM];
            df_weights(sh, M, x, inv2s2, wn);
#pragma unroll
            for (int m = 0; m < DF_MAXM; ++m) {
                if (m < M) {
                    const float d[3] = {x[0] - sh.a[m][0], x[1] - sh.a[m][1], x[2] - sh.a[m][2]};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = d[0] * sh.lin[m][c] + d[1] * sh.lin[m][3 + c] + d[2] * sh.lin[m][6 + c] + sh.off[m][c] + sh.a[m][c];
                        z[i][c] = __builtin_fmaf(wn[m], v, z[i][c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sum[c] += z[i][c];
                z_out[((size_t)b * N + n) * 3 + c] = z[i][c];
            }
        }
    }
    float mu[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) mu[c] = df_block_sum(sum[c], scratch) / (float)N;
    // the farthest point from the centre: (radius bits, ~index) as one 64-bit key, the lowest index among equals
    unsigned long long key = 0ull;
#pragma unroll
    for (int i = 0; i < DF_MAXP; ++i) {
        const int n = t + 256 * i;
        if (n < N) {
            const float dx = z[i][0] - mu[0], dy = z[i][1] - mu[1], dz = z[i][2] - mu[2];
            const float r = sqrtf(dx * dx + dy * dy + dz * dz);
            const unsigned long long k = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned)(~n);
            key = k > key ? k : key;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if ((t & 63) == 0) best[t >> 6] = key;
    __syncthreads();
    unsigned long long kb = best[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) kb = best[w] > kb ? best[w] : kb;
    const float r = __uint_as_float((unsigned)(kb >> 32));
    const float s = (1.0f / r) * 0.999999f;
    if (t == 0) {
        float *st = stat + (size_t)b * 8;
        st[0] = mu[0]; st[1] = mu[1]; st[2] = mu[2]; st[3] = r;
        reinterpret_cast<int *>(st)[4] = (int)(~(unsigned)kb);
    }
#pragma unroll
    for (int i = 0; i < DF_MAXP; ++i) {
        const int n = t + 256 * i;
        if (n < N) {
            const float mk = mask ? mask[(size_t)b * N + n] : 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) out[((size_t)b * N + n) * 3 + c] = (z[i][c] - mu[c]) * s * mk;
        }
    }
}

__global__ __launch_bounds__(256) void deform_backward_kernel(int N, int M, const float *__restrict__ xyz,
                                                              const float *__restrict__ anchors,
                                                              const float *__restrict__ axes, const float *__restrict__ mask,
                                                              float inv2s2, const float *__restrict__ z_in,
                                                              const float *__restrict__ stat, const float *__restrict__ g_out,
                                                              float *__restrict__ g_lin, float *__restrict__ g_off,
                                                              float *__restrict__ g_mask) {
    __shared__ DeformShared sh;
    __shared__ float scratch[4];
    __shared__ float wsum[4][DF_MAXM * 12];
    const int b = blockIdx.x, t = threadIdx.x;
    df_load_shared(sh, b, M, anchors, nullptr, nullptr, axes);
    const float *st = stat + (size_t)b * 8;
    const float mu[3] = {st[0], st[1], st[2]}, r = st[3];
    const int kfar = reinterpret_cast<const int *>(st)[4];
    const float s = (1.0f / r) * 0.999999f;
    float gu[DF_MAXP][3];
    float s3[3] = {0.0f, 0.0f, 0.0f}, qs = 0.0f;
#pragma unroll
    for (int i = 0; i < DF_MAXP; ++i) {
        const int n = t + 256 * i;
        gu[i][0] = gu[i][1] = gu[i][2] = 0.0f;
        if (n < N) {
            const float mk = mask ? mask[(size_t)b * N + n] : 1.0f;
            float gm = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float g = g_out[((size_t)b * N + n) * 3 + c], zc = z_in[((size_t)b * N + n) * 3 + c] - mu[c];
                gm = __builtin_fmaf(g, zc * s, gm);
                gu[i][c] = g * mk;
                s3[c] += gu[i][c];
                qs = __builtin_fmaf(gu[i][c], zc, qs);
            }
            if (g_mask) g_mask[(size_t)b * N + n] = gm;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) s3[c] = df_block_sum(s3[c], scratch);
    qs = df_block_sum(qs, scratch);
    const float g_r = -(0.999999f / (r * r)) * qs;
    float e[3], spread[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        e[c] = (z_in[((size_t)b * N + kfar) * 3 + c] - mu[c]) / r;
        spread[c] = -(s * s3[c] + g_r * e[c]) / (float)N;
    }
    float acc[DF_MAXM * 12];
#pragma unroll
    for (int j = 0; j < DF_MAXM * 12; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < DF_MAXP; ++i) {
        const int n = t + 256 * i;
        if (n < N) {
            const float *px = xyz + ((size_t)b * N + n) * 3;
            const float x[3] = {px[0], px[1], px[2]};
            float wn[DF_MAXM], gz[3];
            df_weights(sh, M, x, inv2s2, wn);
#pragma unroll
            for (int c = 0; c < 3; ++c) gz[c] = s * gu[i][c] + spread[c] + (n == kfar ? g_r * e[c] : 0.0f);
#pragma unroll
            for (int m = 0; m < DF_MAXM; ++m) {
                if (m < M) {
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[m * 12 + 3 * rr + c] = __builtin_fmaf(wn[m] * (x[rr] - sh.a[m][rr]), gz[c], acc[m * 12 + 3 * rr + c]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[m * 12 + 9 + c] = __builtin_fmaf(wn[m], gz[c], acc[m * 12 + 9 + c]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DF_MAXM * 12; ++j) {
        if (j < M * 12) {
            float v = acc[j];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((t & 63) == 0) wsum[t >> 6][j] = v;
        }
    }
    __syncthreads();
    if (t < M * 12) {
        const float v = (wsum[0][t] + wsum[1][t]) + (wsum[2][t] + wsum[3][t]);
        const int m = t / 12, j = t % 12;
        if (j < 9) g_lin[((size_t)b * M + m) * 9 + j] = v;
        else g_off[((size_t)b * M + m) * 3 + j - 9] = v;
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

extern "C" int apn_deform_forward(int b, int n, int m, const float *xyz, const float *anchors, const float *lin,
                                  const float *off, const float *axes, const float *mask, float sigma, float *z,
                                  float *stat, float *out, void *stream) {
    using namespace apn;
    if (b < 0 || n <= 0 || m <= 0 || m > DF_MAXM || n > 256 * DF_MAXP || sigma <= 0.0f) return APN_EINVAL;
    if (b == 0) return APN_OK;
    if (!xyz || !anchors || !lin || !off || !axes || !z || !stat || !out) return APN_EINVAL;
    hipLaunchKernelGGL(deform_forward_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, n, m, xyz, anchors, lin, off, axes,
                       mask, 0.5f / (sigma * sigma), z, stat, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_deform_backward(int b, int n, int m, const float *xyz, const float *anchors, const float *axes,
                                   const float *mask, float sigma, const float *z, const float *stat, const float *g_out,
                                   float *g_lin, float *g_off, float *g_mask, void *stream) {
    using namespace apn;
    if (b < 0 || n <= 0 || m <= 0 || m > DF_MAXM || n > 256 * DF_MAXP || sigma <= 0.0f) return APN_EINVAL;
    if (b == 0) return APN_OK;
    if (!xyz || !anchors || !axes || !z || !stat || !g_out || !g_lin || !g_off) return APN_EINVAL;
    hipLaunchKernelGGL(deform_backward_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, n, m, xyz, anchors, axes, mask,
                       0.5f / (sigma * sigma), z, stat, g_out, g_lin, g_off, g_mask);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
