// spectral.hip -- spectral normalisation of a weight matrix, the parametrisation every layer of the AdaptPoint
// discriminator carries (`PointDiscriminator1` / `PointNetSetAbstraction_SpectralNorm`,
// openpoints/models_adaptpoint/point_discriminator.py:17-73, 149-191: torch.nn.utils.spectral_norm on each Conv2d /
// Linear).  PyTorch evaluates one training-mode forward of it as ~13 small launches (mv, norm, clamp, div, mv, norm,
// clamp, div, two clones, mv, vdot, div) and its backward as ~7; the joint GAN step runs 21 such forwards and 14
// backwards -- ~370 launches of 4-5 us.  Here: three launches forward, two backward.
//
//   power iteration (torch/nn/utils/parametrizations.py, _SpectralNorm._power_method / forward), W (R x C):
//     u' = normalize(W v),  v' = normalize(W^T u'),  sigma = u'^T W v',  Wn = W / sigma.
//   With t = W v and s = W^T t:  u' = t / |t|,  W^T u' = s / |t|,  v' = s / |s|,  sigma = |s| / |t|
//   (normalize(x) = x / max(|x|, eps), followed literally below so that degenerate inputs behave the same).
//   Backward (u', v' constants, as in PyTorch):  dL/dW = gWn / sigma - (sum(gWn o Wn) / sigma) u' v'^T.
#include <hip/hip_runtime.h>

#include "../../include/adaptpoint_amd.h"
#include "apn_common.h"

namespace apn {

__device__ __forceinline__ double sn_block_sum(double v, double *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// t[r] = sum_c W[r][c] x[c]: one wave per row
__global__ __launch_bounds__(256) void sn_mv_kernel(int R, int C, const float *__restrict__ W,
                                                    const float *__restrict__ x, float *__restrict__ t) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float *w = W + (size_t)row * C;
    float acc = 0.0f;
    for (int c = lane; c < C; c += 64) acc = __builtin_fmaf(w[c], x[c], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) t[row] = acc;
}

// s[c] = sum_r W[r][c] t[r]: 64 columns per workgroup, sixteen interleaved row groups added in a fixed order
// (four groups of 256 threads left the 1024 x 512 layer's sixteen workgroups at 19 us)
__global__ __launch_bounds__(1024) void sn_mtv_kernel(int R, int C, const float *__restrict__ W,
                                                      const float *__restrict__ t, float *__restrict__ s) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, col = blockIdx.x * 64 + lane, g = threadIdx.x >> 6;
    float acc = 0.0f;
    if (col < C)
        for (int r = g; r < R; r += 16) acc = __builtin_fmaf(W[(size_t)r * C + col], t[r], acc);
    red[g][lane] = acc;
    __syncthreads();
    if (g == 0 && col < C) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += red[k][lane];
        s[col] = sum;
    }
}

// training: (t, s) -> u', v' (into the buffers and into the copies kept for the backward), sigma, Wn = W / sigma;
// otherwise t = W v with the stored v: sigma = u . t.  Every workgroup derives the scalars itself (R + C <= a few
// thousand values), workgroup 0 writes the vectors.
__global__ __launch_bounds__(256) void sn_apply_kernel(int R, int C, const float *__restrict__ W,
                                                       const float *__restrict__ t, const float *__restrict__ s,
                                                       int training, float eps, float *__restrict__ u,
                                                       float *__restrict__ v, float *__restrict__ uc,
                                                       float *__restrict__ vc, float *__restrict__ sigma_out,
                                                       float *__restrict__ Wn) {
    __shared__ double scratch[4];
    const int tid = threadIdx.x;
    float sigma;
    if (training) {
        double a = 0.0, b = 0.0;
        for (int i = tid; i < R; i += 256) a += (double)t[i] * (double)t[i];
        for (int i = tid; i < C; i += 256) b += (double)s[i] * (double)s[i];
        a = sn_block_sum(a, scratch);
        b = sn_block_sum(b, scratch);
        const float nt = fmaxf((float)sqrt(a), eps);              // u' = t / nt
        const float nq = (float)(sqrt(b) / (double)nt);           // |W^T u'|
        const float nqc = fmaxf(nq, eps);                         // v' = (s / nt) / nqc
        sigma = nq * nq / nqc;                                    // v' . (W^T u')
        if (blockIdx.x == 0) {
            for (int i = tid; i < R; i += 256) {
                const float x = t[i] / nt;
                u[i] = x; uc[i] = x;
            }
            for (int i = tid; i < C; i += 256) {
                const float x = (s[i] / nt) / nqc;
                v[i] = x; vc[i] = x;
            }
        }
    } else {
        double a = 0.0;
        for (int i = tid; i < R; i += 256) a += (double)u[i] * (double)t[i];
        sigma = (float)sn_block_sum(a, scratch);
        if (blockIdx.x == 0) {
            for (int i = tid; i < R; i += 256) uc[i] = u[i];
            for (int i = tid; i < C; i += 256) vc[i] = v[i];
        }
    }
    if (blockIdx.x == 0 && tid == 0) sigma_out[0] = sigma;
    const size_t n = (size_t)R * C;
    for (size_t e = (size_t)blockIdx.x * 256 + tid; e < n; e += (size_t)gridDim.x * 256) Wn[e] = W[e] / sigma;
}

// part[b] = sum over workgroup b's elements of g[e] * Wn[e]
__global__ __launch_bounds__(256) void sn_grad_dot_kernel(size_t n, const float *__restrict__ g,
                                                          const float *__restrict__ Wn, double *__restrict__ part) {
    __shared__ double scratch[4];
    double a = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
        a += (double)g[e] * (double)Wn[e];
    a = sn_block_sum(a, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// gW = g / sigma - (sum(g o Wn) / sigma) u v^T
__global__ __launch_bounds__(256) void sn_grad_apply_kernel(int R, int C, const float *__restrict__ g,
                                                            const double *__restrict__ part, int nparts,
                                                            const float *__restrict__ sigma_p,
                                                            const float *__restrict__ uc, const float *__restrict__ vc,
                                                            float *__restrict__ gW) {
    __shared__ double scratch[4];
    double d = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) d += part[i];
    d = sn_block_sum(d, scratch);                             // the same fixed tree in every workgroup
    const float sigma = sigma_p[0], k = (float)(d / (double)sigma);
    const size_t n = (size_t)R * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
        gW[e] = g[e] / sigma - k * uc[r] * vc[c];
    }
}

// ---- all layers of a network in one set of launches (the discriminator: seven layers, three forwards per step) -------
// The same five kernels with the layer on blockIdx.y and the layers' pointers in a by-value table: 3 launches forward and
// 2 backward for ALL layers instead of 3 and 2 per layer (63 + 28 launches of 4-7 us per joint step on the second lane).
constexpr int SN_MAX_LAYERS = 8;
struct SnJob {
    const float *w;
    float *u, *v, *t, *s, *uc, *vc, *sigma, *wn;
    int R, C, nb;                                 // nb: workgroups of the element-wise kernels for this layer
};
struct SnBatch { SnJob j[SN_MAX_LAYERS]; };
struct SnGradJob {
    const float *g, *wn, *sigma, *uc, *vc;
    double *part;
    float *gw;
    int R, C, nb;
};
struct SnGradBatch { SnGradJob j[SN_MAX_LAYERS]; };

__global__ __launch_bounds__(256) void sn_mv_many_kernel(SnBatch b) {
    const SnJob &j = b.j[blockIdx.y];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= j.R) return;
    const float *w = j.w + (size_t)row * j.C;
    float acc = 0.0f;
    for (int c = lane; c < j.C; c += 64) acc = __builtin_fmaf(w[c], j.v[c], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) j.t[row] = acc;
}

__global__ __launch_bounds__(1024) void sn_mtv_many_kernel(SnBatch b) {
    __shared__ float red[16][64];
    const SnJob &j = b.j[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.C) return;                              // (whole workgroup: no barrier is skipped by part of it)
    const int lane = threadIdx.x & 63, col = blockIdx.x * 64 + lane, g = threadIdx.x >> 6;
    float acc = 0.0f;
    if (col < j.C)
        for (int r = g; r < j.R; r += 16) acc = __builtin_fmaf(j.w[(size_t)r * j.C + col], j.t[r], acc);
    red[g][lane] = acc;
    __syncthreads();
    if (g == 0 && col < j.C) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += red[k][lane];
        j.s[col] = sum;
    }
}

__global__ __launch_bounds__(256) void sn_apply_many_kernel(SnBatch b, int training, float eps) {
    __shared__ double scratch[4];
    const SnJob &j = b.j[blockIdx.y];
    if ((int)blockIdx.x >= j.nb) return;
    const int tid = threadIdx.x, R = j.R, C = j.C;
    float sigma;
    if (training) {
        double a = 0.0, bb = 0.0;
        for (int i = tid; i < R; i += 256) a += (double)j.t[i] * (double)j.t[i];
        for (int i = tid; i < C; i += 256) bb += (double)j.s[i] * (double)j.s[i];
        a = sn_block_sum(a, scratch);
        bb = sn_block_sum(bb, scratch);
        const float nt = fmaxf((float)sqrt(a), eps);
        const float nq = (float)(sqrt(bb) / (double)nt);
        const float nqc = fmaxf(nq, eps);
        sigma = nq * nq / nqc;
        if (blockIdx.x == 0) {
            for (int i = tid; i < R; i += 256) {
                const float x = j.t[i] / nt;
                j.u[i] = x; j.uc[i] = x;
            }
            for (int i = tid; i < C; i += 256) {
                const float x = (j.s[i] / nt) / nqc;
                j.v[i] = x; j.vc[i] = x;
            }
        }
    } else {
        double a = 0.0;
        for (int i = tid; i < R; i += 256) a += (double)j.u[i] * (double)j.t[i];
        sigma = (float)sn_block_sum(a, scratch);
        if (blockIdx.x == 0) {
            for (int i = tid; i < R; i += 256) j.uc[i] = j.u[i];
            for (int i = tid; i < C; i += 256) j.vc[i] = j.v[i];
        }
    }
    if (blockIdx.x == 0 && tid == 0) j.sigma[0] = sigma;
    const size_t n = (size_t)R * C;
    for (size_t e = (size_t)blockIdx.x * 256 + tid; e < n; e += (size_t)j.nb * 256) j.wn[e] = j.w[e] / sigma;
}

__global__ __launch_bounds__(256) void sn_grad_dot_many_kernel(SnGradBatch b) {
    __shared__ double scratch[4];
    const SnGradJob &j = b.j[blockIdx.y];
    if ((int)blockIdx.x >= j.nb) return;
    const size_t n = (size_t)j.R * j.C;
    double a = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)j.nb * 256)
        a += (double)j.g[e] * (double)j.wn[e];
    a = sn_block_sum(a, scratch);
    if (threadIdx.x == 0) j.part[blockIdx.x] = a;
}

__global__ __launch_bounds__(256) void sn_grad_apply_many_kernel(SnGradBatch b) {
    __shared__ double scratch[4];
    const SnGradJob &j = b.j[blockIdx.y];
    if ((int)blockIdx.x >= j.nb) return;
    double d = 0.0;
    for (int i = threadIdx.x; i < j.nb; i += 256) d += j.part[i];
    d = sn_block_sum(d, scratch);
    const float sigma = j.sigma[0], k = (float)(d / (double)sigma);
    const size_t n = (size_t)j.R * j.C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)j.nb * 256) {
        const int r = (int)(e / j.C), c = (int)(e - (size_t)r * j.C);
        j.gw[e] = j.g[e] / sigma - k * j.uc[r] * j.vc[c];
    }
}

static int sn_blocks(size_t n) {
    size_t b = (n + 1023) / 1024;             // four elements per thread
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

}  // namespace apn

extern "C" int apn_spectral_norm_blocks(int rows, int cols) { return apn::sn_blocks((size_t)rows * cols); }

extern "C" int apn_spectral_norm(int rows, int cols, const float *w, int training, float eps, float *u, float *v,
                                 float *scratch, float *u_used, float *v_used, float *sigma, float *w_normalized,
                                 void *stream) {
    using namespace apn;
    if (rows <= 0 || cols <= 0 || !w || !u || !v || !scratch || !u_used || !v_used || !sigma || !w_normalized)
        return APN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float *t = scratch, *s = scratch + rows;
    if (training) {
        hipLaunchKernelGGL(sn_mv_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, cols, w, v, t);
        APN_LAUNCH_CHECK();
        hipLaunchKernelGGL(sn_mtv_kernel, dim3((cols + 63) / 64), dim3(1024), 0, st, rows, cols, w, t, s);
        APN_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(sn_mv_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, cols, w, v, t);
        APN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sn_apply_kernel, dim3(sn_blocks((size_t)rows * cols)), dim3(256), 0, st, rows, cols, w, t, s,
                       training, eps, u, v, u_used, v_used, sigma, w_normalized);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_spectral_norm_grad(int rows, int cols, const float *g, const float *w_normalized, const float *sigma,
                                      const float *u_used, const float *v_used, double *part, float *g_w, void *stream) {
    using namespace apn;
    if (rows <= 0 || cols <= 0 || !g || !w_normalized || !sigma || !u_used || !v_used || !part || !g_w) return APN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)rows * cols;
    const int nb = sn_blocks(n);
    hipLaunchKernelGGL(sn_grad_dot_kernel, dim3(nb), dim3(256), 0, st, n, g, w_normalized, part);
    APN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_grad_apply_kernel, dim3(nb), dim3(256), 0, st, rows, cols, g, part, nb, sigma, u_used, v_used,
                       g_w);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// All layers at once: arrays of `n_layers` (<= 8) entries, one per layer, with the meaning of apn_spectral_norm's
// arguments; scratch[i] holds rows[i] + cols[i] floats.
extern "C" int apn_spectral_norm_many(int n_layers, const int *rows, const int *cols, const float *const *w, int training,
                                      float eps, float *const *u, float *const *v, float *const *scratch,
                                      float *const *u_used, float *const *v_used, float *const *sigma,
                                      float *const *w_normalized, void *stream) {
    using namespace apn;
    if (n_layers <= 0 || n_layers > SN_MAX_LAYERS || !rows || !cols || !w || !u || !v || !scratch || !u_used || !v_used ||
        !sigma || !w_normalized)
        return APN_EINVAL;
    SnBatch b;
    int max_mv = 0, max_mtv = 0, max_nb = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (rows[i] <= 0 || cols[i] <= 0 || !w[i] || !u[i] || !v[i] || !scratch[i] || !u_used[i] || !v_used[i] || !sigma[i] ||
            !w_normalized[i])
            return APN_EINVAL;
        SnJob &j = b.j[i];
        j.w = w[i]; j.u = u[i]; j.v = v[i]; j.t = scratch[i]; j.s = scratch[i] + rows[i];
        j.uc = u_used[i]; j.vc = v_used[i]; j.sigma = sigma[i]; j.wn = w_normalized[i];
        j.R = rows[i]; j.C = cols[i]; j.nb = sn_blocks((size_t)rows[i] * cols[i]);
        max_mv = max_mv > (rows[i] + 3) / 4 ? max_mv : (rows[i] + 3) / 4;
        max_mtv = max_mtv > (cols[i] + 63) / 64 ? max_mtv : (cols[i] + 63) / 64;
        max_nb = max_nb > j.nb ? max_nb : j.nb;
    }
    for (int i = n_layers; i < SN_MAX_LAYERS; ++i) b.j[i] = b.j[0];
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sn_mv_many_kernel, dim3(max_mv, n_layers), dim3(256), 0, st, b);
    APN_LAUNCH_CHECK();
    if (training) {
        hipLaunchKernelGGL(sn_mtv_many_kernel, dim3(max_mtv, n_layers), dim3(1024), 0, st, b);
        APN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sn_apply_many_kernel, dim3(max_nb, n_layers), dim3(256), 0, st, b, training, eps);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// part[i]: apn_spectral_norm_blocks(rows[i], cols[i]) doubles
extern "C" int apn_spectral_norm_grad_many(int n_layers, const int *rows, const int *cols, const float *const *g,
                                           const float *const *w_normalized, const float *const *sigma,
                                           const float *const *u_used, const float *const *v_used, double *const *part,
                                           float *const *g_w, void *stream) {
    using namespace apn;
    if (n_layers <= 0 || n_layers > SN_MAX_LAYERS || !rows || !cols || !g || !w_normalized || !sigma || !u_used || !v_used ||
        !part || !g_w)
        return APN_EINVAL;
    SnGradBatch b;
    int max_nb = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (rows[i] <= 0 || cols[i] <= 0 || !g[i] || !w_normalized[i] || !sigma[i] || !u_used[i] || !v_used[i] || !part[i] ||
            !g_w[i])
            return APN_EINVAL;
        SnGradJob &j = b.j[i];
        j.g = g[i]; j.wn = w_normalized[i]; j.sigma = sigma[i]; j.uc = u_used[i]; j.vc = v_used[i]; j.part = part[i];
        j.gw = g_w[i]; j.R = rows[i]; j.C = cols[i]; j.nb = sn_blocks((size_t)rows[i] * cols[i]);
        max_nb = max_nb > j.nb ? max_nb : j.nb;
    }
    for (int i = n_layers; i < SN_MAX_LAYERS; ++i) b.j[i] = b.j[0];
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sn_grad_dot_many_kernel, dim3(max_nb, n_layers), dim3(256), 0, st, b);
    APN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_grad_apply_many_kernel, dim3(max_nb, n_layers), dim3(256), 0, st, b);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
