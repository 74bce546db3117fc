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
