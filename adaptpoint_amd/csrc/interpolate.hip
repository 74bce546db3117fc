// interpolate.hip -- three-nearest-neighbour search and inverse-distance
// interpolation (forward + backward) for gfx950.
//
// Replaces (openpoints/cpp/pointnet2_batch/src/interpolate_gpu.cu):
//   three_nn_kernel_fast                :16-81
//   three_interpolate_kernel_fast       :84-124
//   three_interpolate_grad_kernel_fast  :127-168
//
// three_nn: a DPP quad of four lanes per unknown point; the known cloud is staged per
// workgroup into LDS as float4 and each lane of the quad walks one contiguous quarter
// in index order (one ds_read_b128 per candidate), then the quad merges.  The reference
// keeps its three bests as doubles initialised to 1e40 and compares the float
// distance against them (:37,44-56), then narrows to float on store (:57).
// Every value ever stored in a best is a float, and 1e40 both compares above
// every float and narrows to +inf, so float bests initialised to +inf give
// bit-identical outputs; that is what the kernel keeps.
//
// three_interpolate: out = fma(w2,p2, fma(w0,p0, w1*p1)) -- the contraction of
// the reference's single expression (:103), pinned explicitly.  A lane owns one
// target point and loops over a channel tile, so idx/weight are read once.
//
// three_interpolate_grad: a workgroup owns whole (b, c) rows of grad_points and
// accumulates them in LDS (ds_add_f32), then adds each row to memory once;
// global float atomics remain as the any-size fallback (see group_points.hip).
#include "apn_common.h"

namespace apn {

constexpr int NN_THREADS = 256;
constexpr int NN_SPLIT = 4;     // lanes that share one unknown point (a DPP quad)
constexpr int NN_CHUNK = 2048;  // known points staged per pass (float4 each: 32 KiB + padding)

struct Top3 {
    float d1, d2, d3;
    int i1, i2, i3;
};

// The reference's strict-< insertion cascade (interpolate_gpu.cu:44-56): an element equal to
// an existing best is placed AFTER it, so earlier candidates win ties.
__device__ __forceinline__ void top3_insert(Top3 &t, float d, int k) {
    if (d < t.d1) {
        t.d3 = t.d2; t.i3 = t.i2; t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = k;
    } else if (d < t.d2) {
        t.d3 = t.d2; t.i3 = t.i2; t.d2 = d; t.i2 = k;
    } else if (d < t.d3) {
        t.d3 = d; t.i3 = k;
    }
}

// `early` holds candidates with lower indices than `late`; both sorted by (distance, index).
// Feeding late's triple through the cascade after early's is exactly what the sequential scan
// would have done with those six survivors.
__device__ __forceinline__ Top3 top3_merge(Top3 early, const Top3 &late) {
    top3_insert(early, late.d1, late.i1);
    top3_insert(early, late.d2, late.i2);
    top3_insert(early, late.d3, late.i3);
    return early;
}

template <int CTRL>
__device__ __forceinline__ Top3 top3_from_partner(const Top3 &t) {
    Top3 o;
    o.d1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t.d1), CTRL, 0xF, 0xF, true));
    o.d2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t.d2), CTRL, 0xF, 0xF, true));
    o.d3 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t.d3), CTRL, 0xF, 0xF, true));
    o.i1 = __builtin_amdgcn_mov_dpp(t.i1, CTRL, 0xF, 0xF, true);
    o.i2 = __builtin_amdgcn_mov_dpp(t.i2, CTRL, 0xF, 0xF, true);
    o.i3 = __builtin_amdgcn_mov_dpp(t.i3, CTRL, 0xF, 0xF, true);
    return o;
}

// Four adjacent lanes (a DPP quad) share one unknown point: each scans one contiguous quarter
// of the staged known points in index order, then the quad merges its four sorted triples
// pairwise (earlier quarter first), which reproduces the sequential scan's result exactly.
// The known cloud is staged per workgroup as float4 {x,y,z,-}; quarter q starts at element
// q*(quarter+1), the +1 keeping the four streams on different LDS banks.
__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(
    int n, int m, const float *__restrict__ unknown, const float *__restrict__ known,
    float *__restrict__ out_dist2, int *__restrict__ idx) {
    extern __shared__ float4 s_kn[];
    const int cloud = blockIdx.y;
    const int tid = threadIdx.x;
    const int sub = tid & (NN_SPLIT - 1);
    const int pt = blockIdx.x * (NN_THREADS / NN_SPLIT) + (tid >> 2);
    known += (size_t)cloud * m * 3;
    const bool live = pt < n;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (live) {
        const float *u = unknown + ((size_t)cloud * n + pt) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    const float inf = __builtin_huge_valf();
    Top3 best{inf, inf, inf, 0, 0, 0};
    for (int base = 0; base < m; base += NN_CHUNK) {
        const int len = min(NN_CHUNK, m - base);
        const int quarter = (len + NN_SPLIT - 1) / NN_SPLIT;
        __syncthreads();
        for (int i = tid; i < len; i += NN_THREADS) {
            const float *kp = known + (size_t)(base + i) * 3;
            const int q = i / quarter;
            s_kn[i + q] = make_float4(kp[0], kp[1], kp[2], 0.0f);
        }
        __syncthreads();
        Top3 mine{inf, inf, inf, 0, 0, 0};
        const int k0 = sub * quarter, k1 = min(k0 + quarter, len);
        const float4 *src = s_kn + sub;            // + sub: the per-quarter bank shift
        for (int k = k0; k < k1; ++k) {
            const float4 p = src[k];
            top3_insert(mine, dist2(ux - p.x, uy - p.y, uz - p.z), base + k);
        }
        // quad merge: lanes 2q,2q+1 first (xor 1), then the two pairs (xor 2)
        Top3 other = top3_from_partner<DPP_QUAD_XOR1>(mine);
        mine = (sub & 1) ? top3_merge(other, mine) : top3_merge(mine, other);
        other = top3_from_partner<DPP_QUAD_XOR2>(mine);
        mine = (sub & 2) ? top3_merge(other, mine) : top3_merge(mine, other);
        best = top3_merge(best, mine);             // chunks arrive in index order
    }
    if (live && sub == 0) {
        float *o = out_dist2 + ((size_t)cloud * n + pt) * 3;
        int *oi = idx + ((size_t)cloud * n + pt) * 3;
        o[0] = best.d1; o[1] = best.d2; o[2] = best.d3;
        oi[0] = best.i1; oi[1] = best.i2; oi[2] = best.i3;
    }
}

constexpr int TI_THREADS = 256;
constexpr int TI_CT = 8;

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
    const int cloud = blockIdx.z;
    const int c0 = blockIdx.y * TI_CT;
    const int c1 = min(c0 + TI_CT, c);
    const int pt = blockIdx.x * TI_THREADS + threadIdx.x;
    if (pt >= n) return;
    const int *ix = idx + ((size_t)cloud * n + pt) * 3;
    const float *w = weight + ((size_t)cloud * n + pt) * 3;
    const int ia = ix[0], ib = ix[1], ic = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *p = points + ((size_t)cloud * c + c0) * m;
    float *o = out + ((size_t)cloud * c + c0) * n + pt;
    for (int ch = c0; ch < c1; ++ch, p += m, o += n)
        o[0] = __builtin_fmaf(w2, p[ic], __builtin_fmaf(w0, p[ia], w1 * p[ib]));
}

constexpr int TG_THREADS = 512;

__global__ __launch_bounds__(TG_THREADS) void three_interpolate_grad_lds_kernel(
    int c, int n, int m, int ct, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    extern __shared__ float acc[];  // [ct][m]
    const int cloud = blockIdx.y;
    const int c0 = blockIdx.x * ct;
    const int nc = min(ct, c - c0);
    const int tid = threadIdx.x;
    for (int i = tid; i < nc * m; i += TG_THREADS) acc[i] = 0.0f;
    __syncthreads();
    const float *g = grad_out + ((size_t)cloud * c + c0) * n;
    for (int pt = tid; pt < n; pt += TG_THREADS) {
        const int *ix = idx + ((size_t)cloud * n + pt) * 3;
        const float *w = weight + ((size_t)cloud * n + pt) * 3;
        const int ia = ix[0], ib = ix[1], ic = ix[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        // eight channels per step, their gradients requested together (one channel at a time each load was waited for
        // before the next: a chain of dependent round trips per thread)
        for (int ch0 = 0; ch0 < nc; ch0 += 8) {
            float go[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) go[u] = g[(size_t)(ch0 + u < nc ? ch0 + u : nc - 1) * n + pt];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ch0 + u < nc) {
                    atomicAdd(&acc[(ch0 + u) * m + ia], go[u] * w0);
                    atomicAdd(&acc[(ch0 + u) * m + ib], go[u] * w1);
                    atomicAdd(&acc[(ch0 + u) * m + ic], go[u] * w2);
                }
            }
        }
    }
    __syncthreads();
    float *dst = grad_points + ((size_t)cloud * c + c0) * m;
    const int total = nc * m;
    for (int i0 = tid; i0 < total; i0 += 4 * TG_THREADS) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * TG_THREADS;
            d[u] = dst[i < total ? i : total - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * TG_THREADS;
            if (i < total) dst[i] = d[u] + acc[i];
        }
    }
}

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_grad_atomic_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    const int cloud = blockIdx.z;
    const int ch = blockIdx.y;
    const int pt = blockIdx.x * TI_THREADS + threadIdx.x;
    if (pt >= n) return;
    const int *ix = idx + ((size_t)cloud * n + pt) * 3;
    const float *w = weight + ((size_t)cloud * n + pt) * 3;
    const float go = grad_out[((size_t)cloud * c + ch) * n + pt];
    float *dst = grad_points + ((size_t)cloud * c + ch) * m;
    atomicAdd(dst + ix[0], go * w[0]);
    atomicAdd(dst + ix[1], go * w[1]);
    atomicAdd(dst + ix[2], go * w[2]);
}

}  // namespace apn

extern "C" int apn_three_nn(int b, int n, int m, const float *unknown, const float *known,
                            float *dist2, int *idx, void *stream) {
    using namespace apn;
    if (b < 0 || n < 0 || m < 0 || b > 65535) return APN_EINVAL;
    if (b == 0 || n == 0) return APN_OK;
    if (!unknown || !dist2 || !idx || (m > 0 && !known)) return APN_EINVAL;
    const int per_block = NN_THREADS / NN_SPLIT;
    dim3 grid((n + per_block - 1) / per_block, b);
    const int chunk = m < NN_CHUNK ? m : NN_CHUNK;
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(NN_THREADS), sizeof(float4) * (chunk + NN_SPLIT),
                       (hipStream_t)stream, n, m, unknown, known, dist2, idx);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_three_interpolate(int b, int c, int m, int n, const float *points,
                                     const int *idx, const float *weight, float *out,
                                     void *stream) {
    using namespace apn;
    if (b < 0 || c < 0 || m < 0 || n < 0 || b > 65535) return APN_EINVAL;
    if (b == 0 || c == 0 || n == 0) return APN_OK;
    if (!points || !idx || !weight || !out) return APN_EINVAL;
    dim3 grid((n + TI_THREADS - 1) / TI_THREADS, (c + TI_CT - 1) / TI_CT, b);
    if (grid.y > 65535) return APN_EINVAL;
    hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(TI_THREADS), 0, (hipStream_t)stream, c,
                       m, n, points, idx, weight, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                          const int *idx, const float *weight,
                                          float *grad_points, void *stream) {
    using namespace apn;
    if (b < 0 || c < 0 || m < 0 || n < 0 || b > 65535) return APN_EINVAL;
    if (b == 0 || c == 0 || n == 0 || m == 0) return APN_OK;
    if (!grad_out || !idx || !weight || !grad_points) return APN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_budget = 64 * 1024;
    if ((size_t)m * sizeof(float) <= lds_budget) {
        int ct = (int)(lds_budget / ((size_t)m * sizeof(float)));
        if (ct > c) ct = c;
        while (ct > 1 && (long long)b * ((c + ct - 1) / ct) < 512) ct = (ct + 1) / 2;
        dim3 grid((c + ct - 1) / ct, b);
        hipLaunchKernelGGL(three_interpolate_grad_lds_kernel, grid, dim3(TG_THREADS),
                           (size_t)ct * m * sizeof(float), st, c, n, m, ct, grad_out, idx, weight,
                           grad_points);
    } else {
        if (c > 65535) return APN_EINVAL;
        dim3 grid((n + TI_THREADS - 1) / TI_THREADS, c, b);
        hipLaunchKernelGGL(three_interpolate_grad_atomic_kernel, grid, dim3(TI_THREADS), 0, st, c,
                           n, m, grad_out, idx, weight, grad_points);
    }
    APN_LAUNCH_CHECK();
    return APN_OK;
}
