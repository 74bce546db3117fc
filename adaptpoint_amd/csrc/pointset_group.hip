// pointset_group.hip -- the grouping stage of the AdaptPoint imitator's PointsetGrouper
// (openpoints/models_adaptpoint/generator_component4_15.py:394-431, normalize = "anchor")
// without its four materialised (B, np, K, C) tensors:
//
//     grouped = points[b, idx[b,q,k], :]                       (index_points, :413)
//     grouped = alpha * (grouped - points[b, fidx[b,q], :]) + beta      (:422-427)
//     out[b, :, q] = max_k grouped                             (:429)
//
// points is POINT-major (B,N,C) f32 in the reference already, so a neighbour is one contiguous
// row: a thread owns (query, 4 channels), walks the K neighbour rows with 16-byte loads
// (consecutive threads = consecutive 16 bytes of the same row), keeps the running maximum of
// the TRANSFORMED value -- computed exactly as PyTorch does, sub, mul, add, no contraction --
// and the position of its first occurrence (torch.max(dim) semantics), and the (C, np) output
// tile goes through LDS so that both the row reads and the channel-major writes are coalesced.
// At stage 1 of the generator (B=32, N=1024, np=512, K=24, C=128) the reference moves
// 4 x 201 MB through HBM for this; here the 16 MB table is read through L2 and 8 MB written.
//
// Backward: the gradient of a max goes to the selected neighbour and, negated, to the anchor
// (float atomics into a caller-zeroed (B,N,C) buffer, 2 per output element); dL/dalpha and
// dL/dbeta leave as one partial row per workgroup, summed by the caller.
#include "apn_common.h"

namespace apn {

// Queries per workgroup: 16 for narrow rows, fewer for wide ones, so that the later stages of
// the generator (few queries, C = 512 / 1024) still spread over the chip (B * np / QT
// workgroups) and the (C, QT) output tile stays small.
__host__ __device__ inline int pg_qt(int c) {
    const int q = 2048 / c;
    return q < 2 ? 2 : q > 16 ? 16 : q;
}

__global__ __launch_bounds__(256) void pointset_group_max_kernel(
    int n, int m, int c, int k, const float *__restrict__ points, const int *__restrict__ idx,
    const int *__restrict__ fidx, const float *__restrict__ alpha, const float *__restrict__ beta,
    float *__restrict__ out, unsigned char *__restrict__ ksel) {
    extern __shared__ int s_dyn[];
    const int PG_QT = pg_qt(c);
    int *s_idx = s_dyn;                                   // [QT][k]
    int *s_anchor = s_dyn + PG_QT * k;                    // [QT]
    float *s_out = reinterpret_cast<float *>(s_anchor + PG_QT);   // [c][QT + 1]
    const int cloud = blockIdx.y, q0 = blockIdx.x * PG_QT, tid = threadIdx.x;
    const int q_here = m - q0 < PG_QT ? m - q0 : PG_QT;
    for (int e = tid; e < q_here * k; e += 256) s_idx[e] = idx[((size_t)cloud * m + q0) * k + e];
    if (tid < q_here) s_anchor[tid] = fidx[(size_t)cloud * m + q0 + tid];
    __syncthreads();
    const int c4n = c >> 2;
    const float4 *P4 = reinterpret_cast<const float4 *>(points) + (size_t)cloud * n * c4n;
    for (int e = tid; e < q_here * c4n; e += 256) {
        const int q = e / c4n, c4 = e - q * c4n;
        const float4 a = P4[(size_t)s_anchor[q] * c4n + c4];
        const float4 al = reinterpret_cast<const float4 *>(alpha)[c4];
        const float4 be = reinterpret_cast<const float4 *>(beta)[c4];
        float best[4];
        int bk[4] = {0, 0, 0, 0};
        const int *row = s_idx + q * k;
        for (int kk = 0; kk < k; ++kk) {
            const float4 x = P4[(size_t)row[kk] * c4n + c4];
            // alpha * (x - anchor) + beta, three separately rounded operations (:426-427)
            const float v[4] = {(x.x - a.x) * al.x + be.x, (x.y - a.y) * al.y + be.y,
                                (x.z - a.z) * al.z + be.z, (x.w - a.w) * al.w + be.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (kk == 0) {
                    best[j] = v[j];
                } else if (v[j] > best[j]) {          // strict: the first maximal position stays
                    best[j] = v[j];
                    bk[j] = kk;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_out[(4 * c4 + j) * (PG_QT + 1) + q] = best[j];
        uchar4 ks;
        ks.x = (unsigned char)bk[0]; ks.y = (unsigned char)bk[1];
        ks.z = (unsigned char)bk[2]; ks.w = (unsigned char)bk[3];
        reinterpret_cast<uchar4 *>(ksel)[((size_t)cloud * m + q0 + q) * c4n + c4] = ks;
    }
    __syncthreads();
    for (int e = tid; e < c * PG_QT; e += 256) {
        const int ch = e / PG_QT, q = e - ch * PG_QT;
        if (q < q_here) out[((size_t)cloud * c + ch) * m + q0 + q] = s_out[ch * (PG_QT + 1) + q];
    }
}

// part[block][2c] = {sum g * (x_sel - anchor), sum g} over the block's queries; c/4 divides 256.
__global__ __launch_bounds__(256) void pointset_group_max_grad_kernel(
    int n, int m, int c, int k, const float *__restrict__ points, const int *__restrict__ idx,
    const int *__restrict__ fidx, const float *__restrict__ alpha,
    const unsigned char *__restrict__ ksel, const float *__restrict__ g_out,
    float *__restrict__ g_points, float *__restrict__ part) {
    extern __shared__ int s_dyn[];
    const int PG_QT = pg_qt(c);
    int *s_idx = s_dyn;                                   // [QT][k]
    int *s_anchor = s_dyn + PG_QT * k;                    // [QT]
    float *s_g = reinterpret_cast<float *>(s_anchor + PG_QT);     // [c][QT + 1]
    float *s_red = s_g + c * (PG_QT + 1);                 // [256][8]
    const int cloud = blockIdx.y, q0 = blockIdx.x * PG_QT, tid = threadIdx.x;
    const int q_here = m - q0 < PG_QT ? m - q0 : PG_QT;
    for (int e = tid; e < q_here * k; e += 256) s_idx[e] = idx[((size_t)cloud * m + q0) * k + e];
    if (tid < q_here) s_anchor[tid] = fidx[(size_t)cloud * m + q0 + tid];
    for (int e = tid; e < c * PG_QT; e += 256) {
        const int ch = e / PG_QT, q = e - ch * PG_QT;
        s_g[ch * (PG_QT + 1) + q] = q < q_here ? g_out[((size_t)cloud * c + ch) * m + q0 + q] : 0.0f;
    }
    __syncthreads();
    const int c4n = c >> 2;
    const float *P = points + (size_t)cloud * n * c;
    float *GP = g_points + (size_t)cloud * n * c;
    float da[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    const int c4 = tid % c4n;                             // fixed per thread: 256 % c4n == 0
    const float4 al = reinterpret_cast<const float4 *>(alpha)[c4];
    const float alv[4] = {al.x, al.y, al.z, al.w};
    for (int q = tid / c4n; q < q_here; q += 256 / c4n) {
        const uchar4 ks = reinterpret_cast<const uchar4 *>(ksel)[((size_t)cloud * m + q0 + q) * c4n + c4];
        const int kj[4] = {ks.x, ks.y, ks.z, ks.w};
        const size_t arow = (size_t)s_anchor[q] * c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = 4 * c4 + j;
            const float g = s_g[ch * (PG_QT + 1) + q];
            const size_t srow = (size_t)s_idx[q * k + kj[j]] * c;
            const float ag = alv[j] * g;
            atomicAdd(GP + srow + ch, ag);
            atomicAdd(GP + arow + ch, -ag);
            da[j] += g * (P[srow + ch] - P[arow + ch]);
            db[j] += g;
        }
    }
    // fold the 256 / c4n threads that share a channel quad
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_red[tid * 8 + j] = da[j]; s_red[tid * 8 + 4 + j] = db[j]; }
    __syncthreads();
    if (tid < c4n) {
        float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = tid; t < 256; t += c4n)
#pragma unroll
            for (int j = 0; j < 4; ++j) { sa[j] += s_red[t * 8 + j]; sb[j] += s_red[t * 8 + 4 + j]; }
        float *row = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * c;
#pragma unroll
        for (int j = 0; j < 4; ++j) { row[4 * tid + j] = sa[j]; row[c + 4 * tid + j] = sb[j]; }
    }
}

// The same gradient with the cloud's slice of g_points held in LDS: workgroup (channel block of CB, cloud) owns
// g_points[cloud][:, block] as a tile [n][CB], every (query, channel) adds its two terms there with LDS atomics
// (ds_add_f32) and the tile leaves with plain coalesced stores -- no global float atomics (the scatter above issues
// 2 b m c of them: 116 us per stage at b = 32, the atomic units' rate).  Rows of `part`: the cloud's first row holds
// the block's {dalpha, dbeta} shares, its other rows are zeroed for these channels.
// NOT run-to-run bit-reproducible: several (query, channel-group) threads of the workgroup can add into the same
// cell and float addition order follows wave scheduling (last-bit differences in g_points; dalpha / dbeta are
// fixed-order).  No replay-identity test depends on this gradient (tests compare it to a reference at 1e-5).
__global__ __launch_bounds__(256) void pointset_group_max_grad_tile_kernel(
    int n, int m, int c, int k, int cb_size, int rows_per_cloud, const float *__restrict__ points,
    const int *__restrict__ idx, const int *__restrict__ fidx, const float *__restrict__ alpha,
    const unsigned char *__restrict__ ksel, const float *__restrict__ g_out, float *__restrict__ g_points,
    float *__restrict__ part) {
    extern __shared__ float s_tile[];                      // [n][CB] | g tile [CB][65] | reduction [256][2]
    const int CB = cb_size, G = 256 / CB;
    float *s_g = s_tile + (size_t)n * CB;
    float *s_red = s_g + CB * 65;
    const int cloud = blockIdx.y, ch0 = blockIdx.x * CB, tid = threadIdx.x;
    const int cc = tid % CB, grp = tid / CB, ch = ch0 + cc;
    for (int e = tid; e < n * CB; e += 256) s_tile[e] = 0.0f;
    const float *P = points + (size_t)cloud * n * c;
    const float al = alpha[ch];
    float da = 0.0f, db = 0.0f;
    for (int q0 = 0; q0 < m; q0 += 64) {
        __syncthreads();
        // g_out (B, C, M) is query-contiguous: stage 64 queries of the block's channels (coalesced along the queries)
        for (int e = tid; e < CB * 64; e += 256) {
            const int j = e >> 6, q = e & 63;
            s_g[j * 65 + q] = q0 + q < m ? g_out[((size_t)cloud * c + ch0 + j) * m + q0 + q] : 0.0f;
        }
        __syncthreads();
        const int qe = m - q0 < 64 ? m - q0 : 64;
        // this thread's queries of the chunk, q = grp + G i (at most eight: G >= 8), in PHASES: the pooled slots, then the
        // neighbours they name, then the two feature values -- three rounds of independent loads per chunk.  (Query by
        // query the three loads formed a dependent chain per iteration: 64 chains of ~1 us per workgroup at stage 1.)
        int sl[8], an[8];
        float gq[8];
        bool in[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = grp + G * i;
            in[i] = q < qe;
            const size_t qa = (size_t)cloud * m + q0 + (in[i] ? q : 0);         // (clamped: every load unconditional)
            sl[i] = (int)ksel[qa * c + ch];
            an[i] = fidx[qa];
            gq[i] = in[i] ? s_g[cc * 65 + q] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t qa = (size_t)cloud * m + q0 + (in[i] ? grp + G * i : 0);
            sl[i] = idx[qa * k + sl[i]];
        }
        float ps[8], pa[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ps[i] = P[(size_t)sl[i] * c + ch];
            pa[i] = P[(size_t)an[i] * c + ch];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                      // (ascending queries, as before: the same sums)
            if (in[i]) {
                const float ag = al * gq[i];
                atomicAdd(&s_tile[sl[i] * CB + cc], ag);
                atomicAdd(&s_tile[an[i] * CB + cc], -ag);
                da = __builtin_fmaf(gq[i], ps[i] - pa[i], da);
                db += gq[i];
            }
        }
    }
    __syncthreads();
    float *GP = g_points + (size_t)cloud * n * c + ch0;
    for (int e = tid; e < n * CB; e += 256) GP[(size_t)(e / CB) * c + (e % CB)] = s_tile[e];
    s_red[tid * 2] = da;
    s_red[tid * 2 + 1] = db;
    __syncthreads();
    if (tid < CB) {
        float sa = 0.0f, sb = 0.0f;
        for (int t = tid; t < 256; t += CB) { sa += s_red[t * 2]; sb += s_red[t * 2 + 1]; }
        float *row = part + (size_t)cloud * rows_per_cloud * 2 * c;
        row[ch0 + tid] = sa;
        row[c + ch0 + tid] = sb;
        for (int r = 1; r < rows_per_cloud; ++r) {
            row[(size_t)r * 2 * c + ch0 + tid] = 0.0f;
            row[(size_t)r * 2 * c + c + ch0 + tid] = 0.0f;
        }
    }
}

static int pg_check(int b, int n, int m, int c, int k) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || k <= 0 || k > 255 || b > 65535) return APN_EINVAL;
    const int c4n = c >> 2;
    if ((c & 3) || c4n > 256 || (256 % c4n)) return APN_EINVAL;      // c in {4, 8, ..., 1024}, powers of two
    return APN_OK;
}

}  // namespace apn

extern "C" int apn_pointset_group_rows(int b, int m, int c) {
    if (c <= 0) return 0;
    const int qt = apn::pg_qt(c);
    return b * ((m + qt - 1) / qt);
}

extern "C" int apn_pointset_group_max(int b, int n, int m, int c, int k, const float *points,
                                      const int *idx, const int *fidx, const float *alpha,
                                      const float *beta, float *out, void *ksel, void *stream) {
    using namespace apn;
    if (int e = pg_check(b, n, m, c, k)) return e;
    if (!points || !idx || !fidx || !alpha || !beta || !out || !ksel) return APN_EINVAL;
    const int PG_QT = pg_qt(c);
    const size_t dyn = sizeof(int) * (PG_QT * k + PG_QT) + sizeof(float) * (size_t)c * (PG_QT + 1);
    if (dyn > 48 * 1024) {
        hipError_t ae = hipFuncSetAttribute((const void *)pointset_group_max_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (ae != hipSuccess) return (int)ae;
    }
    hipLaunchKernelGGL(pointset_group_max_kernel, dim3((m + PG_QT - 1) / PG_QT, b), dim3(256), dyn,
                       (hipStream_t)stream, n, m, c, k, points, idx, fidx, alpha, beta, out,
                       (unsigned char *)ksel);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_pointset_group_max_grad(int b, int n, int m, int c, int k, const float *points,
                                           const int *idx, const int *fidx, const float *alpha,
                                           const void *ksel, const float *g_out, float *g_points,
                                           float *part, void *stream) {
    using namespace apn;
    if (int e = pg_check(b, n, m, c, k)) return e;
    if (!points || !idx || !fidx || !alpha || !ksel || !g_out || !g_points || !part) return APN_EINVAL;
    const int PG_QT = pg_qt(c);
    // the LDS-tile form where a cloud's slice [n][CB] fits (CB channels per workgroup, a divisor of 256 and of c)
    // (the widest block that still gives every CU a workgroup, else the narrowest that fits)
    int cbs = 0;
    for (int t = 32; t >= 8; t >>= 1)
        if (c % t == 0 && (size_t)n * t * sizeof(float) <= 128 * 1024) {
            cbs = t;
            if ((long long)(c / t) * b >= 256) break;
        }
    if (cbs) {
        const size_t dyn = sizeof(float) * ((size_t)n * cbs + (size_t)cbs * 65 + 512);
        hipError_t ae = hipFuncSetAttribute((const void *)pointset_group_max_grad_tile_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(pointset_group_max_grad_tile_kernel, dim3(c / cbs, b), dim3(256), dyn, (hipStream_t)stream, n, m,
                           c, k, cbs, (m + PG_QT - 1) / PG_QT, points, idx, fidx, alpha, (const unsigned char *)ksel, g_out,
                           g_points, part);
        APN_LAUNCH_CHECK();
        return APN_OK;
    }
    const size_t dyn = sizeof(int) * (PG_QT * k + PG_QT) + sizeof(float) * ((size_t)c * (PG_QT + 1) + 256 * 8);
    if (dyn > 48 * 1024) {
        hipError_t ae = hipFuncSetAttribute((const void *)pointset_group_max_grad_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (ae != hipSuccess) return (int)ae;
    }
    hipLaunchKernelGGL(pointset_group_max_grad_kernel, dim3((m + PG_QT - 1) / PG_QT, b), dim3(256), dyn,
                       (hipStream_t)stream, n, m, c, k, points, idx, fidx, alpha,
                       (const unsigned char *)ksel, g_out, g_points, part);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
