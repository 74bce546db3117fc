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

// Gradient as a GATHER (round 3): the three (target, weight) pairs of every fine point are turned, per cloud and in LDS,
// into the list of pairs each coarse point receives -- counts by integer LDS atomics (order-free), offsets by a scan,
// the lists filled and then SORTED by target, so the sums below run in a fixed order: bit-reproducible, and no float
// atomic anywhere (the scatter form queues the runs of equal neighbours that nearby fine points share on one LDS
// address: 66 us for 16 MB at c = 1024 .. 128).  Then the workgroup's [ct][n] tile of the upstream gradient is staged
// in LDS with whole-line loads and every (coarse point, channel) sums its entries from there; the result is ADDED to
// the caller's buffer with plain, coalesced read-modify-writes (the contract: the buffer arrives zeroed, group.py-style).
// Threads: J along the coarse points x CG channel groups (small clouds: m = 64 would leave 7/8 of a flat mapping idle).
// LDS: off[m + 1] | cur[m] | ent_i[3n] | ent_w[3n] | part[NT] | tile[ct][n]
constexpr int TGG_THREADS = 512;
constexpr int TGG_CPT = 16;                       // channels per thread (accumulators)

__global__ __launch_bounds__(TGG_THREADS) void three_interpolate_grad_gather_kernel(
    int c, int n, int m, int ct, int J, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    extern __shared__ int tg_lds[];
    int *off = tg_lds;                               // [m + 1]
    int *cur = off + (m + 1);                        // [m]
    int *ent_i = cur + m;                            // [3n]
    float *ent_w = reinterpret_cast<float *>(ent_i + 3 * n);     // [3n]
    int *part = reinterpret_cast<int *>(ent_w + 3 * n);          // [NT]
    float *tile = reinterpret_cast<float *>(part + TGG_THREADS); // [ct][n]
    const int cloud = blockIdx.y, c0 = blockIdx.x * ct, nc = min(ct, c - c0), tid = threadIdx.x;
    const int *ix = idx + (size_t)cloud * n * 3;
    const float *wv = weight + (size_t)cloud * n * 3;
    const int ne = 3 * n;

    // the tile's loads first: they are in flight while the map is built
    const float *g = grad_out + ((size_t)cloud * c + c0) * n;
    const int total = nc * n;
    for (int i0 = tid; i0 < total; i0 += 8 * TGG_THREADS) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * TGG_THREADS;
            v[u] = g[i < total ? i : total - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * TGG_THREADS;
            if (i < total) tile[i] = v[u];
        }
    }
    for (int j = tid; j <= m; j += TGG_THREADS) off[j] = 0;
    __syncthreads();
    for (int e = tid; e < ne; e += TGG_THREADS) {
        const int j = ix[e];
        if ((unsigned)j < (unsigned)m) atomicAdd(&off[j + 1], 1);        // (an index outside the cloud is dropped, not followed)
    }
    __syncthreads();
    // exclusive scan of off[1 .. m] in place: K consecutive entries per thread, the threads' totals by one wave
    const int K = (m + TGG_THREADS - 1) / TGG_THREADS;
    int local = 0;
    for (int k = 0; k < K; ++k) {
        const int j = tid * K + k;
        if (j < m) local += off[j + 1];
    }
    part[tid] = local;
    __syncthreads();
    if (tid < 64) {
        int s8[TGG_THREADS / 64], sum = 0;
#pragma unroll
        for (int u = 0; u < TGG_THREADS / 64; ++u) { s8[u] = part[tid * (TGG_THREADS / 64) + u]; sum += s8[u]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (tid >= d) incl += o;
        }
        int run = incl - sum;
#pragma unroll
        for (int u = 0; u < TGG_THREADS / 64; ++u) { part[tid * (TGG_THREADS / 64) + u] = run; run += s8[u]; }
    }
    __syncthreads();
    {
        int run = part[tid];
        for (int k = 0; k < K; ++k) {
            const int j = tid * K + k;
            if (j < m) {
                const int cnt = off[j + 1];
                cur[j] = run;                        // where coarse point j's list starts (also its fill cursor)
                run += cnt;
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < m; j += TGG_THREADS) off[j] = cur[j];
    __syncthreads();
    for (int e = tid; e < ne; e += TGG_THREADS) {
        const int j = ix[e];
        if ((unsigned)j < (unsigned)m) {
            const int pos = atomicAdd(&cur[j], 1);
            ent_i[pos] = e / 3;
            ent_w[pos] = wv[e];
        }
    }
    __syncthreads();
    // cur[j] is now the END of list j; sort every list by target (insertion sort: lists hold ~3n/m entries), equal
    // targets (a fine point naming one coarse point twice) by their weights' bits: the order of the addends is a function
    // of the inputs alone.
    for (int j = tid; j < m; j += TGG_THREADS) {
        const int lo = off[j], hi = cur[j];
        for (int a = lo + 1; a < hi; ++a) {
            const int ki = ent_i[a];
            const float kw = ent_w[a];
            int b = a - 1;
            while (b >= lo && (ent_i[b] > ki || (ent_i[b] == ki && __float_as_uint(ent_w[b]) > __float_as_uint(kw)))) {
                ent_i[b + 1] = ent_i[b];
                ent_w[b + 1] = ent_w[b];
                --b;
            }
            ent_i[b + 1] = ki;
            ent_w[b + 1] = kw;
        }
    }
    __syncthreads();
    // gather: thread (jj, cg) owns coarse points jj, jj + J, ... and channels cg, cg + CG, ... (at most TGG_CPT of them)
    const int CG = TGG_THREADS / J, jj = tid % J, cg = tid / J;
    float *dst = grad_points + ((size_t)cloud * c + c0) * m;
    for (int j = jj; j < m; j += J) {
        const int lo = off[j], hi = cur[j];
        float acc[TGG_CPT];
#pragma unroll
        for (int u = 0; u < TGG_CPT; ++u) acc[u] = 0.0f;
        for (int a = lo; a < hi; ++a) {
            const int i = ent_i[a];
            const float w = ent_w[a];
#pragma unroll
            for (int u = 0; u < TGG_CPT; ++u) {
                const int ch = cg + u * CG;
                if (ch < nc) acc[u] = __builtin_fmaf(w, tile[ch * n + i], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < TGG_CPT; ++u) {
            const int ch = cg + u * CG;
            if (ch < nc) dst[(size_t)ch * m + j] += acc[u];
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
    {
        // the gather form, while the cloud's map and at least one channel row fit the LDS budget
        const size_t meta = sizeof(int) * ((size_t)2 * m + 1 + TGG_THREADS) + (size_t)24 * n;
        const size_t budget = 96 * 1024;
        if (meta + (size_t)4 * n <= budget && m <= 8192) {
            int J = 64;
            while (J < TGG_THREADS && J < m) J *= 2;
            const int CG = TGG_THREADS / J;
            int ct = (int)((budget - meta) / ((size_t)4 * n));
            if (ct > TGG_CPT * CG) ct = TGG_CPT * CG;
            if (ct > c) ct = c;
            while (ct > CG && (long long)b * ((c + ct - 1) / ct) < 256) ct = (ct + 1) / 2;
            dim3 grid((c + ct - 1) / ct, b);
            if (grid.x <= 65535) {
                const size_t dyn = meta + (size_t)4 * n * ct;
                static const bool raised = [] {
                    return hipFuncSetAttribute((const void *)three_interpolate_grad_gather_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
                }();
                if (raised || dyn <= 64 * 1024) {
                    hipLaunchKernelGGL(three_interpolate_grad_gather_kernel, grid, dim3(TGG_THREADS), dyn, st, c, n, m, ct, J,
                                       grad_out, idx, weight, grad_points);
                    APN_LAUNCH_CHECK();
                    return APN_OK;
                }
            }
        }
    }
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
