// group_points.hip -- neighbourhood gather / scatter-add and the per-point
// gather pair, for gfx950.
//
// Replaces (openpoints/cpp/pointnet2_batch/src/):
//   group_points_kernel_fast       group_points_gpu.cu:53-92
//   group_points_grad_kernel_fast  group_points_gpu.cu:14-50
//   gather_points_kernel_fast      sampling_gpu.cu:15-51
//   gather_points_grad_kernel_fast sampling_gpu.cu:53-90
//
// Forward ops are HBM-write bound (the (B,C,M,K) output is 16x..32x the input):
// every lane owns four consecutive outputs, reads its four indices once as an
// int4 and reuses them for a tile of channels, gathers 4-byte elements out of a
// 4 KiB channel row that stays in L1/L2, and stores float4 -- 1 KiB per wave
// instruction.  The reference instead launches one thread per output element
// with the channel in blockIdx.y, re-reading idx per channel.
//
// The backward ops replace the reference's one-global-atomic-per-element
// (16.8 M float atomics at B=32,C=32,M=512,K=32) with a workgroup-private
// accumulator: a workgroup owns whole (b, c) rows of grad_points, accumulates
// them in LDS with ds_add_f32 and adds each row to memory once with plain
// read-modify-write (the caller passes zeros, group.py:111 / subsample.py:136;
// "+=" keeps the reference's accumulate semantics).  Global float atomics remain
// only as the any-size fallback.  Summation order differs from the reference's
// (which is itself unordered), so results agree to float rounding, not bits.
#include "apn_common.h"

namespace apn {

constexpr int GP_THREADS = 256;
constexpr int GP_CT = 8;  // channels per workgroup in the forward gathers

// out[b,c,j] = points[b,c,idx[b,j]],  j over a flattened row of length `len`
// (len = npoints*nsample for grouping, npoints for gather).  VEC = 4 when
// len % 4 == 0 (16-byte aligned rows), else 1.
template <int VEC>
__global__ __launch_bounds__(GP_THREADS) void gather_rows_kernel(
    int c, int n, int len, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out) {
    const int cloud = blockIdx.z;
    const int c0 = blockIdx.y * GP_CT;
    const int c1 = min(c0 + GP_CT, c);
    const int j = (blockIdx.x * GP_THREADS + threadIdx.x) * VEC;
    if (j >= len) return;
    const int *ix = idx + (size_t)cloud * len + j;
    const float *src = points + ((size_t)cloud * c + c0) * n;
    float *dst = out + ((size_t)cloud * c + c0) * len + j;
    if (VEC == 4) {
        const int4 i4 = *reinterpret_cast<const int4 *>(ix);
        for (int ch = c0; ch < c1; ++ch, src += n, dst += len) {
            float4 v;
            v.x = src[i4.x]; v.y = src[i4.y]; v.z = src[i4.z]; v.w = src[i4.w];
            *reinterpret_cast<float4 *>(dst) = v;
        }
    } else {
        const int i1 = ix[0];
        for (int ch = c0; ch < c1; ++ch, src += n, dst += len) dst[0] = src[i1];
    }
}

static int launch_gather_rows(int b, int c, int n, int len, const float *points, const int *idx,
                              float *out, hipStream_t st) {
    if (b == 0 || c == 0 || len == 0) return APN_OK;
    if (b > 65535) return APN_EINVAL;
    const bool vec = (len % 4) == 0 &&
                     ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int per_block = GP_THREADS * (vec ? 4 : 1);
    dim3 grid((len + per_block - 1) / per_block, (c + GP_CT - 1) / GP_CT, b);
    if (grid.y > 65535) return APN_EINVAL;
    if (vec)
        hipLaunchKernelGGL(gather_rows_kernel<4>, grid, dim3(GP_THREADS), 0, st, c, n, len, points,
                           idx, out);
    else
        hipLaunchKernelGGL(gather_rows_kernel<1>, grid, dim3(GP_THREADS), 0, st, c, n, len, points,
                           idx, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// grad_points[b,c,idx[b,j]] += grad_out[b,c,j].  One workgroup owns `ct`
// channels of one cloud: acc[ct][n] lives in LDS.
constexpr int SC_THREADS = 512;

__global__ __launch_bounds__(SC_THREADS) void scatter_rows_lds_kernel(
    int c, int n, int len, int ct, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
    extern __shared__ float acc[];  // [ct][n]
    const int cloud = blockIdx.y;
    const int c0 = blockIdx.x * ct;
    const int nc = min(ct, c - c0);
    const int tid = threadIdx.x;
    for (int i = tid; i < nc * n; i += SC_THREADS) acc[i] = 0.0f;
    __syncthreads();
    const int *ix = idx + (size_t)cloud * len;
    const float *g = grad_out + ((size_t)cloud * c + c0) * len;
    // Eight positions per thread and step, every load of the step issued before its first use (clamped positions, the
    // surplus dropped): one position and one channel at a time each load was waited for before the next was issued -- a
    // chain of dependent round trips per thread (106 us for 78 MB at B = 32)
    constexpr int U = 8;
    for (int j0 = tid; j0 < len; j0 += U * SC_THREADS) {
        int t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * SC_THREADS;
            t[u] = ix[j < len ? j : len - 1];
        }
        for (int ch = 0; ch < nc; ++ch) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * SC_THREADS;
                v[u] = g[(size_t)ch * len + (j < len ? j : len - 1)];
            }
            // Same-address LDS atomics serialise, and a ball query's row ends in a run of copies of its first hit
            // (ball_query_gpu.cu:41-45) -- with 32 slots and ~8 hits, 24 lanes of a half-wave add into ONE cell.  The lanes
            // of each group of 32 that hold their group's first target are summed in registers (five exchanges) and added
            // once by the group's first lane; the others add for themselves.
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = j0 + u * SC_THREADS < len;
                const int t0 = __shfl(t[u], (int)(threadIdx.x & 32u));
                const bool lead = (threadIdx.x & 31u) == 0u;
                const bool same = live && t[u] == t0;
                float s = same ? v[u] : 0.0f;
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 8);
                s += __shfl_xor(s, 4);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 1);
                if (live && (lead || !same)) atomicAdd(&acc[ch * n + t[u]], lead ? s : v[u]);
            }
        }
    }
    __syncthreads();
    float *dst = grad_points + ((size_t)cloud * c + c0) * n;
    const int total = nc * n;
    for (int i0 = tid; i0 < total; i0 += 4 * SC_THREADS) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SC_THREADS;
            d[u] = dst[i < total ? i : total - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SC_THREADS;
            if (i < total) dst[i] = d[u] + acc[i];
        }
    }
}

__global__ __launch_bounds__(GP_THREADS) void scatter_rows_atomic_kernel(
    int c, int n, int len, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
    const int cloud = blockIdx.z;
    const int ch = blockIdx.y;
    const int j = blockIdx.x * GP_THREADS + threadIdx.x;
    if (j >= len) return;
    const int t = idx[(size_t)cloud * len + j];
    atomicAdd(grad_points + ((size_t)cloud * c + ch) * n + t,
              grad_out[((size_t)cloud * c + ch) * len + j]);
}

static int launch_scatter_rows(int b, int c, int n, int len, const float *grad_out, const int *idx,
                               float *grad_points, hipStream_t st) {
    if (b == 0 || c == 0 || len == 0 || n == 0) return APN_OK;
    if (b > 65535) return APN_EINVAL;
    const size_t lds_budget = 64 * 1024;  // keeps two workgroups per CU resident
    if ((size_t)n * sizeof(float) <= lds_budget) {
        int ct = (int)(lds_budget / ((size_t)n * sizeof(float)));
        if (ct > c) ct = c;
        // enough workgroups to cover the chip: shrink the channel tile if needed
        while (ct > 1 && (long long)b * ((c + ct - 1) / ct) < 512) ct = (ct + 1) / 2;
        dim3 grid((c + ct - 1) / ct, b);
        hipLaunchKernelGGL(scatter_rows_lds_kernel, grid, dim3(SC_THREADS),
                           (size_t)ct * n * sizeof(float), st, c, n, len, ct, grad_out, idx,
                           grad_points);
    } else {
        if (c > 65535) return APN_EINVAL;
        dim3 grid((len + GP_THREADS - 1) / GP_THREADS, c, b);
        hipLaunchKernelGGL(scatter_rows_atomic_kernel, grid, dim3(GP_THREADS), 0, st, c, n, len,
                           grad_out, idx, grad_points);
    }
    APN_LAUNCH_CHECK();
    return APN_OK;
}

}  // namespace apn

extern "C" int apn_group_points(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out, void *stream) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return APN_EINVAL;
    const long long len = (long long)npoints * nsample;
    if (len > 0x7fffffffLL) return APN_EINVAL;
    if (b && c && len && (!points || !idx || !out)) return APN_EINVAL;
    return apn::launch_gather_rows(b, c, n, (int)len, points, idx, out, (hipStream_t)stream);
}

extern "C" int apn_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int *idx, float *grad_points,
                                     void *stream) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return APN_EINVAL;
    const long long len = (long long)npoints * nsample;
    if (len > 0x7fffffffLL) return APN_EINVAL;
    if (b && c && len && n && (!grad_out || !idx || !grad_points)) return APN_EINVAL;
    return apn::launch_scatter_rows(b, c, n, (int)len, grad_out, idx, grad_points,
                                    (hipStream_t)stream);
}

namespace apn {

// The training loop's resampler (examples/classification/train_autoaug.py:493-498, train.py:265):
// from the rows picked by FPS, keep the random subset `choice` (one draw for the whole batch) and
// emit the two layouts the classifier consumes -- pos (B,S,3) row-major and x (B,CX,S)
// channel-major -- in ONE pass.  Thread = one kept point: its source row (c <= 8 floats) is read
// once; the x stores are coalesced along s for every channel.
__global__ __launch_bounds__(256) void resample_rows_kernel(int n, int c, int p_all, int s_cnt, int cx,
                                                            const float *__restrict__ points,
                                                            const int *__restrict__ fidx,
                                                            const int *__restrict__ choice,
                                                            float *__restrict__ pos,
                                                            float *__restrict__ x) {
    const int s = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (s >= s_cnt) return;
    const int src = fidx[(size_t)b * p_all + choice[s]];
    const float *__restrict__ row = points + ((size_t)b * n + src) * c;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < c ? row[k] : 0.0f;
    float *__restrict__ po = pos + ((size_t)b * s_cnt + s) * 3;
    po[0] = v[0]; po[1] = v[1]; po[2] = v[2];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < cx) x[((size_t)b * cx + k) * s_cnt + s] = v[k];
}

}  // namespace apn

extern "C" int apn_resample_points(int b, int n, int c, int p_all, int s_cnt, int cx,
                                   const float *points, const int *fidx, const int *choice,
                                   float *pos, float *x, void *stream) {
    if (b < 0 || n <= 0 || c < 3 || c > 8 || cx < 0 || cx > c || p_all <= 0 || s_cnt < 0)
        return APN_EINVAL;
    if (b == 0 || s_cnt == 0) return APN_OK;
    if (b > 65535 || !points || !fidx || !choice || !pos || (cx && !x)) return APN_EINVAL;
    hipLaunchKernelGGL(apn::resample_rows_kernel, dim3((s_cnt + 255) / 256, b), dim3(256), 0,
                       (hipStream_t)stream, n, c, p_all, s_cnt, cx, points, fidx, choice, pos, x);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_gather_points(int b, int c, int n, int npoints, const float *points,
                                 const int *idx, float *out, void *stream) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0) return APN_EINVAL;
    if (b && c && npoints && (!points || !idx || !out)) return APN_EINVAL;
    return apn::launch_gather_rows(b, c, n, npoints, points, idx, out, (hipStream_t)stream);
}

extern "C" int apn_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                      const int *idx, float *grad_points, void *stream) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0) return APN_EINVAL;
    if (b && c && npoints && n && (!grad_out || !idx || !grad_points)) return APN_EINVAL;
    return apn::launch_scatter_rows(b, c, n, npoints, grad_out, idx, grad_points,
                                    (hipStream_t)stream);
}
