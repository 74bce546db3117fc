// ball_query.hip -- radius neighbour search for gfx950.
//
// Replaces ball_query_kernel_fast and its launcher
// (openpoints/cpp/pointnet2_batch/src/ball_query_gpu.cu:15-73).
//
// The reference gives each query ONE thread that walks all N support points in
// order (B*M threads, N dependent iterations each, divergent early exit).  Here
// a query belongs to a whole WAVE: the 64 lanes test 64 consecutive support
// points at once, a ballot turns the hits into a 64-bit mask, and
// mbcnt (popcount of the lower lanes) gives every hit its slot in index order
// -- the order the sequential scan would have produced.  The cloud's xyz is
// staged once per workgroup into LDS as three planes (x[], y[], z[]), read back
// with conflict-free ds_read_b32, and reused by every query of the tile; the
// query's own coordinates are wave-uniform (SGPRs).
//
// Semantics kept bit-exact (ball_query_gpu.cu:29-48): radius2 = radius*radius in
// float32; strict d2 < radius2 with d2 = fma(dz,dz, fma(dx,dx, dy*dy)) on
// (query - point) differences; the first hit pre-fills all nsample slots; the
// scan stops once nsample hits are stored; a query with no hit writes nothing
// (its row keeps the caller's zeros, group.py:194).
#include "apn_common.h"

namespace apn {

constexpr int BQ_WAVES = 4;                // waves per workgroup
constexpr int BQ_CHUNK = 4096;             // support points staged per LDS pass (48 KiB)

// Grid: (ceil(M / queries_per_block), B).  Each wave owns queries
// q0 + wave, q0 + wave + BQ_WAVES, ... of its block's tile.
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    int n, int m, float radius2, int nsample, int q_per_block, int zero_empty,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int *__restrict__ idx) {
    // Dynamic LDS: three coordinate planes of `chunk` floats, then per query of
    // the tile its hit count so far and its first hit.
    extern __shared__ float s_dyn[];
    const int chunk = min(n, BQ_CHUNK);
    float *sx = s_dyn, *sy = s_dyn + chunk, *sz = s_dyn + 2 * chunk;
    int *s_cnt = reinterpret_cast<int *>(s_dyn + 3 * chunk);

    const int cloud = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q_begin = blockIdx.x * q_per_block;
    const int q_end = min(q_begin + q_per_block, m);

    xyz += (size_t)cloud * n * 3;
    new_xyz += (size_t)cloud * m * 3;
    idx += (size_t)cloud * m * nsample;

    int *cnt_of = s_cnt;
    int *first_of = s_cnt + q_per_block;
    for (int i = tid; i < q_per_block; i += BQ_WAVES * 64) { cnt_of[i] = 0; first_of[i] = 0; }

    for (int base = 0; base < n; base += BQ_CHUNK) {
        const int len = min(BQ_CHUNK, n - base);
        __syncthreads();  // previous pass done with the planes (and cnt init visible)
        // Coalesced staging: the chunk is 3*len consecutive floats.
        for (int i = tid; i < 3 * len; i += BQ_WAVES * 64) {
            const float v = xyz[(size_t)base * 3 + i];
            const int p = i / 3, c = i - p * 3;
            (c == 0 ? sx : c == 1 ? sy : sz)[p] = v;
        }
        __syncthreads();

        for (int q = q_begin + wave; q < q_end; q += BQ_WAVES) {
            const int ql = q - q_begin;
            int cnt = __builtin_amdgcn_readfirstlane(cnt_of[ql]);  // wave-uniform
            if (cnt >= nsample) continue;
            int first = __builtin_amdgcn_readfirstlane(first_of[ql]);
            const float qx = new_xyz[q * 3 + 0];
            const float qy = new_xyz[q * 3 + 1];
            const float qz = new_xyz[q * 3 + 2];
            int *row = idx + (size_t)q * nsample;
            for (int k0 = 0; k0 < len && cnt < nsample; k0 += 64) {
                const int k = k0 + lane;
                bool hit = false;
                if (k < len) {
                    const float d2 = dist2(qx - sx[k], qy - sy[k], qz - sz[k]);
                    hit = d2 < radius2;
                }
                const unsigned long long mask = __ballot(hit);
                if (mask == 0ull) continue;
                if (cnt == 0) first = base + k0 + (int)__builtin_ctzll(mask);
                const int slot = cnt + (int)__builtin_amdgcn_mbcnt_hi(
                                           (unsigned)(mask >> 32),
                                           __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (hit && slot < nsample) row[slot] = base + k;
                cnt += (int)__builtin_popcountll(mask);
            }
            if (lane == 0) { cnt_of[ql] = cnt; first_of[ql] = first; }
        }
    }

    // Tail of each row: slots the scan never reached repeat the first hit
    // (ball_query_gpu.cu:41-45).  Rows of empty balls stay untouched.
    for (int q = q_begin + wave; q < q_end; q += BQ_WAVES) {
        const int ql = q - q_begin;
        const int cnt = __builtin_amdgcn_readfirstlane(cnt_of[ql]);
        if ((cnt == 0 && !zero_empty) || cnt >= nsample) continue;
        const int first = __builtin_amdgcn_readfirstlane(first_of[ql]);   // 0 for an empty ball
        int *row = idx + (size_t)q * nsample;
        for (int l = cnt + lane; l < nsample; l += 64) row[l] = first;
    }
}

}  // namespace apn

static int ball_query_impl(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                           const float *xyz, int *idx, int zero_empty, void *stream) {
    using namespace apn;
    if (b < 0 || n < 0 || m < 0 || nsample < 0) return APN_EINVAL;
    if (b == 0 || m == 0 || nsample == 0) return APN_OK;
    if (n == 0 && !zero_empty) return APN_OK;
    if (!new_xyz || !xyz || !idx) return APN_EINVAL;
    const float radius2 = radius * radius;  // ball_query_gpu.cu:29 (float32 product)
    // Tile so that B * blocks_x comfortably exceeds the 256 CUs while each
    // workgroup still amortises its LDS staging over several queries per wave.
    int q_per_block = 32;
    while (q_per_block > 4 && (long long)b * ((m + q_per_block - 1) / q_per_block) < 1024)
        q_per_block >>= 1;
    dim3 grid((m + q_per_block - 1) / q_per_block, b);
    const int chunk = n < BQ_CHUNK ? n : BQ_CHUNK;
    const size_t dyn = sizeof(float) * 3 * chunk + sizeof(int) * 2 * q_per_block;
    hipLaunchKernelGGL(ball_query_kernel, grid, dim3(BQ_WAVES * 64), dyn, (hipStream_t)stream, n,
                       m, radius2, nsample, q_per_block, zero_empty, new_xyz, xyz, idx);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_ball_query(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx, void *stream) {
    return ball_query_impl(b, n, m, radius, nsample, new_xyz, xyz, idx, 0, stream);
}

// Same search, but rows of empty balls are WRITTEN as zeros, so the caller need not
// pre-zero idx (the state group.py:194 + the reference kernel leave behind).
extern "C" int apn_ball_query_zero(int b, int n, int m, float radius, int nsample,
                                   const float *new_xyz, const float *xyz, int *idx, void *stream) {
    return ball_query_impl(b, n, m, radius, nsample, new_xyz, xyz, idx, 1, stream);
}
