// sa_geo.hip -- occurrence statistics of the neighbourhoods: the last piece of a block's INDEX stage.
//
// For every support point n of a cloud, over the positions (query q, slot k) whose neighbour it is
// (idx[q][k] == n):
//     occ[n] = number of such positions,        D[n] = sum over them of d = (p_n - new_p[q]) / radius
// and, over ALL positions of a batch, the second moments  DD = sum d d^T (left as per-cloud shares).
// They depend on coordinates and neighbour indices only -- never on features or weights -- so they are
// computed where FPS and the ball query run (beside the feature path: another stream, the next batch).
// With them the grouped convolution's first BatchNorm needs NO pass over the positions
// (sa_fused.hip: sa_prep_stats_kernel): y1[q,k] = W1f f_n + W1p d is affine in per-point and per-position
// terms, so  sum y1 = sum_n occ F_n + W1p D_n  and  sum y1^2 = sum_n (occ F_n^2 + 2 F_n W1p D_n) + W1p DD W1p^T,
// F_n = W1f f_n; the backward's per-point sums of yhat1 use the same occ, D (sa_glue.hip: bwd_point_grads).
// The reference evaluates BatchNorm over the materialised (B,C,M,K) tensor instead
// (openpoints/models/backbone/pointnext.py:166 over group.py:248-254).
//
// Reproducibility: the sums over a point's positions arrive in an arbitrary order, so they are accumulated as
// INTEGERS: occ exactly, D in units of 2^-36 (|d| < 1 inside a ball; anything below 2^11 per position still
// fits).  geo[n] = {occ, Dx, Dy, Dz} as four int64.  One workgroup owns a slab of GEO_SLAB points of one cloud
// and accumulates it in LDS (64-bit LDS atomics; plain stores at the end): a first version with 64-bit global
// atomics took 549 us for 640 clouds and slowed every atomic of the feature stream beside it; no accumulator
// needs clearing beforehand.  dd[cloud][slab][6] (float64) = the workgroup's share of the second moments
// {xx, xy, xz, yy, yz, zz}; the consumer adds the shares of its batch in a fixed order.
#include "apn_common.h"

namespace apn {

constexpr double GEO_UNIT = 68719476736.0;        // 2^36
constexpr int GEO_SLAB = 2048;                    // points per workgroup: 64 KB of LDS accumulators

// Grid (slabs, B), 1024 threads: a wave walks the cloud's queries two at a time, lane = (query half, slot k).
// The run of slots that repeat slot 0 (the ball query's fill, ball_query_gpu.cu:41-45; for any other index row:
// slots equal to slot 0, the same point seen from the same query) is folded into slot 0 with its multiplicity.
__global__ __launch_bounds__(1024) void sa_geo_kernel(int n, int m, float radius, const float *__restrict__ xyz,
                                                      const float *__restrict__ new_xyz,
                                                      const int *__restrict__ idx, long long *__restrict__ geo,
                                                      double *__restrict__ dd) {
    extern __shared__ unsigned long long acc[];                 // [slab points][4]
    __shared__ float red[16][6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane & 31, half = lane >> 5;
    const int cloud = blockIdx.y, n0 = blockIdx.x * GEO_SLAB, nslab = (n - n0 < GEO_SLAB ? n - n0 : GEO_SLAB);
    for (int e = tid; e < nslab * 4; e += 1024) acc[e] = 0ull;
    __syncthreads();
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int qb = wave * 2; qb < m; qb += 32) {
        const int q = qb + half;
        const bool live_q = q < m;
        const size_t qg = (size_t)cloud * m + (live_q ? q : 0);
        const int nb = idx[qg * 32 + k];
        const int nb0 = __shfl(nb, half * 32);
        const bool fill = k > 0 && nb == nb0;
        const unsigned long long bal = __ballot(fill);
        const int nfill = __popc((unsigned)(bal >> (32 * half)));
        const int mult = !live_q ? 0 : (k == 0 ? 1 + nfill : (fill ? 0 : 1));
        const int loc = nb - n0;
        if (mult && loc >= 0 && loc < nslab) {
            const float *p = xyz + ((size_t)cloud * n + nb) * 3;
            const float *qq = new_xyz + qg * 3;
            // group.py:250-253: (grouped_xyz - query) then /= radius
            const float dx = (p[0] - qq[0]) / radius, dy = (p[1] - qq[1]) / radius, dz = (p[2] - qq[2]) / radius;
            unsigned long long *g = acc + (size_t)loc * 4;
            const long long mm = mult;
            atomicAdd(g, (unsigned long long)mm);
            atomicAdd(g + 1, (unsigned long long)(mm * __double2ll_rn((double)dx * GEO_UNIT)));
            atomicAdd(g + 2, (unsigned long long)(mm * __double2ll_rn((double)dy * GEO_UNIT)));
            atomicAdd(g + 3, (unsigned long long)(mm * __double2ll_rn((double)dz * GEO_UNIT)));
            const float fm = (float)mult;
            v[0] += fm * dx * dx; v[1] += fm * dx * dy; v[2] += fm * dx * dz;
            v[3] += fm * dy * dy; v[4] += fm * dy * dz; v[5] += fm * dz * dz;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) v[j] += __shfl_xor(v[j], s);
        if (lane == 0) red[wave][j] = v[j];
    }
    __syncthreads();
    {
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(geo + ((size_t)cloud * n + n0) * 4);
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(acc);
        for (int e = tid; e < nslab * 2; e += 1024) dst[e] = src[e];
    }
    if (tid < 6) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += (double)red[w][tid];
        dd[((size_t)cloud * gridDim.x + blockIdx.x) * 6 + tid] = s;
    }
}

}  // namespace apn

// float64 values of dd per cloud: 6 per slab of the cloud's n points
extern "C" int apn_sa_geo_dd_doubles(int n) { return n > 0 ? 6 * ((n + apn::GEO_SLAB - 1) / apn::GEO_SLAB) : 0; }

// geo: int64 [b][n][4]; dd: float64 [b][apn_sa_geo_dd_doubles(n)].  ONE launch; nothing needs clearing.
// nsample must be 32.
extern "C" int apn_sa_point_geo(int b, int n, int m, int nsample, float radius, const float *xyz,
                                const float *new_xyz, const int *idx, void *geo, void *dd, void *stream) {
    using namespace apn;
    if (b <= 0 || n <= 0 || m <= 0 || nsample != 32 || b > 65535 || !(radius > 0.0f)) return APN_EINVAL;
    if (!xyz || !new_xyz || !idx || !geo || !dd || ((uintptr_t)geo & 15)) return APN_EINVAL;
    const int slabs = (n + GEO_SLAB - 1) / GEO_SLAB;
    const size_t lds = (size_t)(n < GEO_SLAB ? n : GEO_SLAB) * 32;
    if (lds > 48 * 1024) {
        if (hipError_t e = hipFuncSetAttribute((const void *)sa_geo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds))
            return (int)e;
    }
    hipLaunchKernelGGL(sa_geo_kernel, dim3(slabs, b), dim3(1024), lds, (hipStream_t)stream, n, m, radius, xyz, new_xyz,
                       idx, (long long *)geo, (double *)dd);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
