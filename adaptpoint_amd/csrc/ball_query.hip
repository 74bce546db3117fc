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
#include "ball_query_body.h"

namespace apn {

constexpr int BQ_WAVES = 4;                // waves per workgroup
#ifndef APN_BQ_FAT
#define APN_BQ_FAT 16                      // waves per workgroup of a LARGE launch (0: the 4-wave form always); see ball_query_impl
#endif
constexpr int BQ_FAT_WAVES = APN_BQ_FAT > 0 ? APN_BQ_FAT : 16;

// Grid: 8 * ceil(B / 8) * blocks_x workgroups, numbered so that ALL the tiles of a cloud run on one XCD (workgroups w
// and w + 8 share an XCD and its L2 under round-robin placement -- a speed assumption only): workgroup w = 8 s + x is
// tile s mod blocks_x of cloud 8 (s / blocks_x) + x.  With a (tiles, clouds) grid a cloud's sixteen tiles were dealt to
// all eight XCDs and each pulled the cloud through its own L2 (PMC: 111 MB per stacked launch for 54 MB of payload).
template <int QPW, int WAVES = BQ_WAVES>
__global__ __launch_bounds__(WAVES * 64) void ball_query_kernel(
    int b, int blocks_x, int n, int m, float radius2, int nsample, int q_per_block, int zero_empty,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int *__restrict__ idx) {
    extern __shared__ float s_dyn[];
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
    const int cloud = 8 * (s / blocks_x) + x, bx = s % blocks_x;
    if (cloud >= b) return;
    ball_query_body<WAVES, QPW>(n, m, radius2, nsample, q_per_block, zero_empty, new_xyz, xyz, idx, cloud, bx, s_dyn);
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
    // Large launches (the stacked index stage of a pipelined step: 640 clouds): workgroups of SIXTEEN waves over tiles of 128
    // queries.  The search fills every wave slot of the chip either way; with 4-wave workgroups a slot group of four falls
    // free at a time and the next 4-wave workgroup of this launch takes it at once, so an 8- or 16-wave workgroup of ANOTHER
    // stream's kernel waits until two or four retirements happen to coincide on one CU (round 5's kernel trace: the feature
    // stream's 8-wave kernels at 90-140 us beside this one against 7-13 alone); a retiring 16-wave workgroup frees room for
    // them at once.  Same waves, same queries per wave, one staging of the cloud per 128 queries instead of per 32.
    const int fat_q = 8 * BQ_FAT_WAVES;
    const bool fat = APN_BQ_FAT > 0 && q_per_block == 32 && (long long)b * ((m + fat_q - 1) / fat_q) >= 1024 && m >= fat_q;
    if (fat) q_per_block = fat_q;
    const int blocks_x = (m + q_per_block - 1) / q_per_block;
    const long long wgs = 8LL * ((b + 7) / 8) * blocks_x;
    if (wgs > 0x7fffffffLL) return APN_EINVAL;
    const int chunk = n < BQ_CHUNK ? n : BQ_CHUNK;
    const size_t dyn = sizeof(float) * 3 * chunk + sizeof(int) * 2 * q_per_block;
    // full tiles (32 queries: the stacked index stages) give every wave EIGHT queries per pass over the staged cloud: more
    // independent ballot / count chains per LDS read (184 -> 173 us for 640 clouds); smaller tiles keep four, so that all
    // four waves have queries
    if (fat)
        hipLaunchKernelGGL((ball_query_kernel<8, BQ_FAT_WAVES>), dim3((unsigned)wgs), dim3(BQ_FAT_WAVES * 64), dyn, (hipStream_t)stream, b, blocks_x,
                           n, m, radius2, nsample, q_per_block, zero_empty, new_xyz, xyz, idx);
    else if (q_per_block >= BQ_WAVES * 8)
        hipLaunchKernelGGL(ball_query_kernel<8>, dim3((unsigned)wgs), dim3(BQ_WAVES * 64), dyn, (hipStream_t)stream, b, blocks_x,
                           n, m, radius2, nsample, q_per_block, zero_empty, new_xyz, xyz, idx);
    else
        hipLaunchKernelGGL(ball_query_kernel<BQ_QPW>, dim3((unsigned)wgs), dim3(BQ_WAVES * 64), dyn, (hipStream_t)stream, b,
                           blocks_x, n, m, radius2, nsample, q_per_block, zero_empty, new_xyz, xyz, idx);
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
