// sa_glue.hip -- the small kernels between the fused set-abstraction passes
// (csrc/sa_fused.hip): partial-row reductions in float64, BatchNorm folding and
// running-statistics update, the (B,M,64) <-> (B,64,M) layout changes, the
// per-channel constants of the backward, and the three "everything downstream of
// dL/dy1 is linear" products.  They replace ~150 tiny PyTorch launches per step
// (and two pathological long-K rocBLAS GEMMs) with a fixed sequence of ~10, all
// graph-capturable: no host reads, no allocation.
#include "apn_common.h"

namespace apn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// out[c] += sum over this workgroup's slice of rows of part[row][c], in float64.
// 256 threads = 64 columns x 4 row groups; grid = (ceil(ncol/64), row slices); the
// slices meet through float64 atomics on the zeroed output (a handful per column).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float *__restrict__ part, int rows,
                                                          int ncol, double *__restrict__ out) {
    __shared__ double red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(r0 + per, rows);
    double s = 0.0;
    if (c < ncol)
        for (int r = r0 + ry; r < r1; r += 4) s += (double)part[(size_t)r * ncol + c];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < ncol)
        atomicAdd(out + c, red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

// BatchNorm fold.  sums = {sum[C], sumsq[C]} over `count` positions (already summed
// over ranks when SyncBatchNorm is on).  pack = {scale, shift, mean, invstd}[C].
// training: batch statistics (biased variance), running buffers updated with the
// unbiased variance (torch.nn.BatchNorm semantics); otherwise the running buffers.
__global__ void bn_fold_kernel(const double *__restrict__ sums, int c, double count,
                               const float *__restrict__ gamma, const float *__restrict__ beta,
                               float eps, float momentum, float *__restrict__ running_mean,
                               float *__restrict__ running_var, long long *__restrict__ nbt,
                               int training, float *__restrict__ pack) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && training && nbt) *nbt += 1;
    if (i >= c) return;
    double mean, var;
    if (training) {
        mean = sums[i] / count;
        var = sums[c + i] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (running_mean) {
            const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
            running_mean[i] = (float)((1.0 - momentum) * running_mean[i] + momentum * mean);
            running_var[i] = (float)((1.0 - momentum) * running_var[i] + momentum * unbiased);
        }
    } else {
        mean = running_mean[i];
        var = running_var[i];
    }
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[i] : 1.0, b = beta ? (double)beta[i] : 0.0;
    pack[i] = (float)(g * inv);
    pack[c + i] = (float)(b - mean * g * inv);
    pack[2 * c + i] = (float)mean;
    pack[3 * c + i] = (float)inv;
}

// sign of gamma2 per channel (+1 / -1): which extreme of y2 the K-pool keeps.
__global__ void sign_kernel(const float *__restrict__ gamma, int c, float *__restrict__ sgn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) sgn[i] = (!gamma || gamma[i] >= 0.0f) ? 1.0f : -1.0f;
}

// out[b][c][m] = ysel[b][m][c] * scale2[c] + shift2[c]      (C = 64)
__global__ __launch_bounds__(256) void fwd_out_kernel(int m, const float *__restrict__ ysel,
                                                      const float *__restrict__ pack2,
                                                      float *__restrict__ out) {
    __shared__ float tile[64][65];
    const int cloud = blockIdx.y, m0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {   // j = query within tile, tx = channel
        const int q = m0 + j;
        tile[j][tx] = q < m ? ysel[((size_t)cloud * m + q) * 64 + tx] * pack2[tx] + pack2[64 + tx] : 0.f;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4)     // c = channel, tx = query
        if (m0 + tx < m) out[((size_t)cloud * 64 + c) * m + m0 + tx] = tile[tx][c];
}

// goa[b][m][c] = g_out[b][c][m] * scale2[c]; partial sums of S1 = sum g, S2 = sum g*yhat_sel,
// yhat_sel = (ysel - mean2) * invstd2.   part: [gridDim.x*gridDim.y][128].
__global__ __launch_bounds__(256) void bwd_prep_kernel(int m, const float *__restrict__ g_out,
                                                       const float *__restrict__ ysel,
                                                       const float *__restrict__ pack2,
                                                       float *__restrict__ goa,
                                                       float *__restrict__ part) {
    __shared__ float tile[64][65];
    __shared__ float red[4][2][64];
    const int cloud = blockIdx.y, m0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4)     // read (c, query tx): coalesced over queries
        tile[tx][c] = m0 + tx < m ? g_out[((size_t)cloud * 64 + c) * m + m0 + tx] : 0.0f;
    __syncthreads();
    const float sc = pack2[tx], mu = pack2[128 + tx], iv = pack2[192 + tx];
    float s1 = 0.0f, s2 = 0.0f;
    for (int j = ty; j < 64; j += 4) {   // (query j, channel tx): coalesced over channels
        const int q = m0 + j;
        if (q < m) {
            const float g = tile[j][tx];
            const size_t o = ((size_t)cloud * m + q) * 64 + tx;
            goa[o] = g * sc;
            s1 += g;
            s2 += g * ((ysel[o] - mu) * iv);
        }
    }
    red[ty][0][tx] = s1;
    red[ty][1][tx] = s2;
    __syncthreads();
    if (ty < 2) {
        float *row = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 128;
        row[ty * 64 + tx] = red[0][ty][tx] + red[1][ty][tx] + red[2][ty][tx] + red[3][ty][tx];
    }
}

// Per-channel constants of dL/dy2 = goa*[pos==ksel] + y2*D2 + E2 and their images
// through W2:  qm = W2^T diag(D2) W2 (32x32), evec = E2 W2.  Also dL/dgamma2, dL/dbeta2.
// One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void bwd_consts2_kernel(const double *__restrict__ S,
                                                           const float *__restrict__ pack2,
                                                           const float *__restrict__ w2, double count,
                                                           int training, float *__restrict__ d2e2,
                                                           float *__restrict__ qm,
                                                           float *__restrict__ evec,
                                                           float *__restrict__ g_gamma2,
                                                           float *__restrict__ g_beta2) {
    __shared__ double D[64], E[64];
    const int t = threadIdx.x;
    if (t < 64) {
        const double sc = pack2[t], mu = pack2[128 + t], iv = pack2[192 + t];
        const double s1 = S[t], s2 = S[64 + t];
        double d = 0.0, e = 0.0;
        if (training) {
            d = -sc * iv * s2 / count;
            e = -sc * s1 / count + sc * mu * iv * s2 / count;
        }
        D[t] = d; E[t] = e;
        d2e2[t] = (float)d;
        d2e2[64 + t] = (float)e;
        if (g_gamma2) g_gamma2[t] = (float)s2;
        if (g_beta2) g_beta2[t] = (float)s1;
    }
    __syncthreads();
    const int k = t >> 5, mid = t & 31;   // 32 x 32
    double q = 0.0;
    for (int c = 0; c < 64; ++c) q += (double)w2[c * 32 + k] * D[c] * (double)w2[c * 32 + mid];
    qm[k * 32 + mid] = (float)q;
    if (t < 32) {
        double e = 0.0;
        for (int c = 0; c < 64; ++c) e += E[c] * (double)w2[c * 32 + t];
        evec[t] = (float)e;
    }
}

// dL/dy1 = g_u*ca + yhat1*cb + cc ; dL/dgamma1 = T2, dL/dbeta1 = T1.
__global__ void bwd_consts1_kernel(const double *__restrict__ T, const float *__restrict__ pack1,
                                   double count, int training, float *__restrict__ cabc,
                                   float *__restrict__ g_gamma1, float *__restrict__ g_beta1) {
    const int i = threadIdx.x;
    if (i >= 32) return;
    const double sc = pack1[i];
    cabc[i] = (float)sc;
    cabc[32 + i] = training ? (float)(-sc * T[32 + i] / count) : 0.0f;
    cabc[64 + i] = training ? (float)(-sc * T[i] / count) : 0.0f;
    if (g_gamma1) g_gamma1[i] = (float)T[32 + i];
    if (g_beta1) g_beta1[i] = (float)T[i];
}

// dL/df[b][i][n] = sum_mid G[b][n][mid] * W1[mid][3+i];  optionally
// dL/dp[b][n][d] = sum_mid G[b][n][mid] * W1[mid][d] / r  (accumulated: +=).
__global__ __launch_bounds__(256) void bwd_input_grad_kernel(int n, const float *__restrict__ G,
                                                             const float *__restrict__ w1,
                                                             float inv_r, float *__restrict__ g_f,
                                                             float *__restrict__ g_p) {
    __shared__ float sw[32][36];     // W1[mid][35]
    __shared__ float sg[64][33];     // G tile [point][mid]
    const int cloud = blockIdx.y, n0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 32 * 35; e += 256) sw[e / 35][e % 35] = w1[e];
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int pt = e >> 5, mid = e & 31;
        sg[pt][mid] = n0 + pt < n ? G[((size_t)cloud * n + n0 + pt) * 32 + mid] : 0.0f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // tx = point, ty = channel group
    for (int i = ty; i < 32; i += 4) {
        float s = 0.0f;
#pragma unroll
        for (int mid = 0; mid < 32; ++mid) s += sg[tx][mid] * sw[mid][3 + i];
        if (n0 + tx < n) g_f[((size_t)cloud * 32 + i) * n + n0 + tx] = s;
    }
    if (g_p && ty < 3 && n0 + tx < n) {
        float s = 0.0f;
#pragma unroll
        for (int mid = 0; mid < 32; ++mid) s += sg[tx][mid] * sw[mid][ty];
        g_p[((size_t)cloud * n + n0 + tx) * 3 + ty] += s * inv_r;
    }
}

// dL/dnew_p[q][d] = -sum_mid H[q][mid] * W1[mid][d] / r
__global__ __launch_bounds__(256) void bwd_query_grad_kernel(int total_q, const float *__restrict__ H,
                                                             const float *__restrict__ w1,
                                                             float inv_r, float *__restrict__ g_q) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= total_q) return;
    float s[3] = {0.f, 0.f, 0.f};
    for (int mid = 0; mid < 32; ++mid) {
        const float hv = H[(size_t)q * 32 + mid];
#pragma unroll
        for (int d = 0; d < 3; ++d) s[d] += hv * w1[mid * 35 + d];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) g_q[(size_t)q * 3 + d] = -s[d] * inv_r;
}

// Partial products over points for dL/dW1:  part[block][mid][38] with columns
//   0..2   sum_n G[n][mid] * xyz[n][d]          (block's points)
//   3..5   sum_q H[q][mid] * new_xyz[q][d]      (block's queries)
//   6..37  sum_n G[n][mid] * ft[n][i]
// The caller sums blocks in float64 and forms (col0-2 - col3-5)/r.  A workgroup stages
// WG_PTS points (and its share of queries) in LDS; thread (mid, group) owns 5 columns.
constexpr int WG_PTS = 64;
__global__ __launch_bounds__(256) void bwd_weight_grad_kernel(int total_n, int total_q,
                                                              const float *__restrict__ G,
                                                              const float *__restrict__ H,
                                                              const __bf16 *__restrict__ ft,
                                                              const float *__restrict__ xyz,
                                                              const float *__restrict__ new_xyz,
                                                              int q_per_block,
                                                              float *__restrict__ part) {
    __shared__ float sG[WG_PTS][33], sX[WG_PTS][36], sH[WG_PTS][33], sQ[WG_PTS][4];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * WG_PTS, q0 = blockIdx.x * q_per_block;
    for (int e = tid; e < WG_PTS * 32; e += 256) {
        const int pt = e >> 5, c = e & 31;
        const bool ok = n0 + pt < total_n;
        sG[pt][c] = ok ? G[(size_t)(n0 + pt) * 32 + c] : 0.0f;
        sX[pt][3 + c] = ok ? (float)ft[(size_t)(n0 + pt) * 32 + c] : 0.0f;
        const bool okq = pt < q_per_block && q0 + pt < total_q;
        sH[pt][c] = okq ? H[(size_t)(q0 + pt) * 32 + c] : 0.0f;
    }
    for (int e = tid; e < WG_PTS * 3; e += 256) {
        const int pt = e / 3, d = e % 3;
        sX[pt][d] = n0 + pt < total_n ? xyz[(size_t)(n0 + pt) * 3 + d] : 0.0f;
        sQ[pt][d] = (pt < q_per_block && q0 + pt < total_q) ? new_xyz[(size_t)(q0 + pt) * 3 + d] : 0.0f;
    }
    __syncthreads();
    const int mid = tid & 31, grp = tid >> 5;   // 8 groups x 5 columns = 40 >= 38
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int pt = 0; pt < WG_PTS; ++pt) {
        const float g = sG[pt][mid], hq = sH[pt][mid];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int col = grp * 5 + j;            // compile-time pattern per group after unroll
            float x;
            if (col < 3) x = g * sX[pt][col];
            else if (col < 6) x = hq * sQ[pt][col - 3];
            else if (col < 38) x = g * sX[pt][col - 3];
            else x = 0.0f;
            acc[j] += x;
        }
    }
    float *row = part + (size_t)blockIdx.x * 32 * 38;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int col = grp * 5 + j;
        if (col < 38) row[mid * 38 + col] = acc[j];
    }
}

// g_w1[mid][0..2] = (s[mid][0..2] - s[mid][3..5]) / r ; g_w1[mid][3+i] = s[mid][6+i]
__global__ void bwd_w1_final_kernel(const double *__restrict__ s, double inv_r, float *__restrict__ g_w1) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 32 * 35) return;
    const int mid = e / 35, col = e % 35;
    g_w1[e] = col < 3 ? (float)((s[mid * 38 + col] - s[mid * 38 + 3 + col]) * inv_r)
                      : (float)s[mid * 38 + 3 + col];
}

__global__ void cast_d2f_kernel(const double *__restrict__ s, int nelem, float *__restrict__ o) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nelem) o[e] = (float)s[e];
}

}  // namespace apn

#define APN_ST ((hipStream_t)stream)

extern "C" int apn_sa_reduce_rows(const float *part, int rows, int ncol, double *out, void *stream) {
    if (rows < 0 || ncol <= 0 || !part || !out) return APN_EINVAL;
    hipError_t me = hipMemsetAsync(out, 0, sizeof(double) * (size_t)ncol, APN_ST);
    if (me != hipSuccess) return (int)me;
    int slices = (rows + 31) / 32;           // ~32 rows per workgroup slice
    if (slices < 1) slices = 1;
    if (slices > 64) slices = 64;
    hipLaunchKernelGGL(apn::reduce_rows_kernel, dim3((ncol + 63) / 64, slices), dim3(256), 0, APN_ST,
                       part, rows, ncol, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bn_fold(const double *sums, int c, double count, const float *gamma,
                              const float *beta, float eps, float momentum, float *running_mean,
                              float *running_var, void *num_batches_tracked, int training,
                              float *pack, void *stream) {
    if (c <= 0 || !pack || (training && !sums) || (!training && (!running_mean || !running_var)))
        return APN_EINVAL;
    hipLaunchKernelGGL(apn::bn_fold_kernel, dim3((c + 63) / 64), dim3(64), 0, APN_ST, sums, c, count,
                       gamma, beta, eps, momentum, running_mean, running_var,
                       (long long *)num_batches_tracked, training, pack);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_sign(const float *gamma, int c, float *sgn, void *stream) {
    if (c <= 0 || !sgn) return APN_EINVAL;
    hipLaunchKernelGGL(apn::sign_kernel, dim3((c + 63) / 64), dim3(64), 0, APN_ST, gamma, c, sgn);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_fwd_out(int b, int m, const float *ysel, const float *pack2, float *out,
                              void *stream) {
    if (b <= 0 || m <= 0 || b > 65535 || !ysel || !pack2 || !out) return APN_EINVAL;
    hipLaunchKernelGGL(apn::fwd_out_kernel, dim3((m + 63) / 64, b), dim3(256), 0, APN_ST, m, ysel,
                       pack2, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_prep_rows(int b, int m) { return b * ((m + 63) / 64); }

extern "C" int apn_sa_bwd_prep(int b, int m, const float *g_out, const float *ysel,
                               const float *pack2, float *goa, float *part, void *stream) {
    if (b <= 0 || m <= 0 || b > 65535 || !g_out || !ysel || !pack2 || !goa || !part) return APN_EINVAL;
    hipLaunchKernelGGL(apn::bwd_prep_kernel, dim3((m + 63) / 64, b), dim3(256), 0, APN_ST, m, g_out,
                       ysel, pack2, goa, part);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_consts2(const double *S, const float *pack2, const float *w2, double count,
                                  int training, float *d2e2, float *qm, float *evec,
                                  float *g_gamma2, float *g_beta2, void *stream) {
    if (!S || !pack2 || !w2 || !d2e2 || !qm || !evec) return APN_EINVAL;
    hipLaunchKernelGGL(apn::bwd_consts2_kernel, dim3(1), dim3(1024), 0, APN_ST, S, pack2, w2, count,
                       training, d2e2, qm, evec, g_gamma2, g_beta2);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_consts1(const double *T, const float *pack1, double count, int training,
                                  float *cabc, float *g_gamma1, float *g_beta1, void *stream) {
    if (!T || !pack1 || !cabc) return APN_EINVAL;
    hipLaunchKernelGGL(apn::bwd_consts1_kernel, dim3(1), dim3(64), 0, APN_ST, T, pack1, count,
                       training, cabc, g_gamma1, g_beta1);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_input_grad(int b, int n, int m, const float *G, const float *H,
                                     const float *w1, float radius, float *g_f, float *g_p,
                                     float *g_newp, void *stream) {
    if (b <= 0 || n <= 0 || b > 65535 || !G || !w1 || !g_f) return APN_EINVAL;
    hipLaunchKernelGGL(apn::bwd_input_grad_kernel, dim3((n + 63) / 64, b), dim3(256), 0, APN_ST, n, G,
                       w1, 1.0f / radius, g_f, g_p);
    APN_LAUNCH_CHECK();
    if (g_newp) {
        if (!H || m <= 0) return APN_EINVAL;
        hipLaunchKernelGGL(apn::bwd_query_grad_kernel, dim3((b * m + 255) / 256), dim3(256), 0, APN_ST,
                           b * m, H, w1, 1.0f / radius, g_newp);
        APN_LAUNCH_CHECK();
    }
    return APN_OK;
}

extern "C" int apn_sa_bwd_weight_rows(int b, int n) { return (b * n + apn::WG_PTS - 1) / apn::WG_PTS; }

extern "C" int apn_sa_bwd_weight_grad(int b, int n, int m, const float *G, const float *H,
                                      const void *ft, const float *xyz, const float *new_xyz,
                                      float *part, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0 || !G || !H || !ft || !xyz || !new_xyz || !part) return APN_EINVAL;
    const int blocks = apn_sa_bwd_weight_rows(b, n);
    const int qpb = (b * m + blocks - 1) / blocks;   // queries are spread evenly over the blocks
    if (qpb > apn::WG_PTS) return APN_EINVAL;        // needs m <= n (always true after sampling)
    hipLaunchKernelGGL(apn::bwd_weight_grad_kernel, dim3(blocks), dim3(256), 0, APN_ST, b * n, b * m, G,
                       H, (const __bf16 *)ft, xyz, new_xyz, qpb, part);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_w1_final(const double *sums, float radius, float *g_w1, void *stream) {
    if (!sums || !g_w1) return APN_EINVAL;
    hipLaunchKernelGGL(apn::bwd_w1_final_kernel, dim3((32 * 35 + 255) / 256), dim3(256), 0, APN_ST,
                       sums, 1.0 / (double)radius, g_w1);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_cast_d2f(const double *src, int nelem, float *dst, void *stream) {
    if (nelem <= 0 || !src || !dst) return APN_EINVAL;
    hipLaunchKernelGGL(apn::cast_d2f_kernel, dim3((nelem + 255) / 256), dim3(256), 0, APN_ST, src,
                       nelem, dst);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
