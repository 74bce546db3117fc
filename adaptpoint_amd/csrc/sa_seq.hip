// sa_seq.hip -- whole-direction launch sequences of the fused set-abstraction block.
//
// Each kernel of csrc/sa_fused.hip / sa_glue.hip is 5-50 us; issued one by one from
// Python (argument marshalling + allocation between launches) the host needs ~10 us per
// launch and the GPU idles in between.  These two entry points issue a whole forward /
// backward back-to-back from C (one foreign call each), which keeps an EAGER step
// GPU-bound -- the mode used at world_size > 1, where collectives sit between phases.
//
// `phases` selects which part of the direction to enqueue, so that the caller can place
// an all-reduce of the BatchNorm sums between them (SyncBatchNorm):
//   forward : 1 = prep + BatchNorm-1 sums (per point) | 2 = main pass | 4 = output
//   backward: 1 = (zero +) entry                      | 2 = main pass (over the tile map, g_u rows stored through the row
//             map: no float atomics)                   | 4 = point gradients + finalize
// 3 + 4 launches per step (round 2: 6 + 6): every BatchNorm fold / constants kernel became a prologue of its
// consumer.  Single rank: phases = 7, one call; the consumers read the partial rows / accumulator sets
// themselves.  With `sums*` pointers (float64, already reduced over ranks) they use those instead.
#include "apn_common.h"

#define APN_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

namespace apn {
__global__ __launch_bounds__(256) void zero16_kernel(uint4 *__restrict__ p, long long n16) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n16; e += stride) p[e] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void fill32_kernel(unsigned *__restrict__ p, unsigned word, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) p[e] = word;
}
}  // namespace apn

// bytes a multiple of 16, base 16-byte aligned.  A kernel, not hipMemsetAsync: a captured memset node aborted
// hipGraph replays on this stack (measured-and-rejected 14).
extern "C" int apn_zero_fill(void *base, long long bytes, void *stream) {
    if (bytes < 0 || (bytes & 15) || (bytes && (!base || ((uintptr_t)base & 15)))) return APN_EINVAL;
    if (!bytes) return APN_OK;
    long long blocks = (bytes / 16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(apn::zero16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4 *)base,
                       bytes / 16);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_forward_seq(
    int phases, int precision, int b, int n, int m, float radius, const float *xyz, const float *new_xyz,
    const float *f, const int *idx, const int *tmap, const int *fidx, const void *geo, const void *dd,
    const float *w1, const float *w2, const float *ws, const float *bs,
    const float *g1, const float *b1, float *rm1, float *rv1, void *nbt1, float eps1, float mom1,
    int train1,
    const float *g2, const float *b2, float *rm2, float *rv2, void *nbt2, float eps2, float mom2,
    int train2,
    double count, int relu, void *ft, float *part1, const double *sums1, const double *sums2,
    float *pack1, float *pack2, void *acc2, float *ysel, void *ksel,
    float *out, float *zero_base, long long zero_floats, void *stream) {
    if (phases & 1)
        APN_TRY(apn_sa_prep_stats(b, n, f, geo, dd, w1, precision, train1, ft, part1, acc2, apn_sa_acc_words(128),
                                  stream));
    if (phases & 2)
        APN_TRY(apn_sa_fwd_main(b, n, m, precision, radius, xyz, new_xyz, ft, idx, tmap, w1, w2, g1, b1, rm1, rv1,
                                nbt1, eps1, mom1, train1, count, part1, apn_sa_prep_rows(b, n), sums1, pack1, g2,
                                ysel, ksel, acc2, stream));
    if (phases & 4)
        APN_TRY(apn_sa_fwd_out(b, n, m, ysel, acc2, sums2, g2, b2, rm2, rv2, nbt2, eps2, mom2, train2, count, pack2,
                               ws ? ft : nullptr, precision, ws ? fidx : nullptr, ws, bs, relu, out, zero_base,
                               zero_floats, stream));
    return APN_OK;
}

extern "C" int apn_sa_backward_seq(
    int phases, int precision, int b, int n, int m, float radius, const float *xyz, const float *new_xyz,
    const int *idx, const int *tmap, const int *fidx, const void *geo, const float *w1, const float *w2,
    const float *ws, const void *ft, const float *pack1, const float *pack2, const float *ysel,
    const void *ksel, const float *out, int relu, int train1, int train2, double count,
    const float *g_out, long long gs_b, long long gs_c, long long gs_m,
    // cleared here unless zero_bytes == 0 (then apn_sa_fwd_out cleared them): gip (B*N*32, only with ws) | accS | accT,
    // contiguous from zero_base
    void *zero_base, long long zero_bytes, float *gip, void *accS, void *accT,
    // the row map of the tile map (apn_sa_rowmap_many: index-stage data) and GU (32*B*M rows of 32 floats, scratch): the
    // backward pass stores a row's g_u at its place in the point-sorted order, the per-point kernel sums a point's rows
    const int *pcnt_poff, const int *rowdst, float *GU,
    // scratch
    float *goa, float *partWs, float *partW2, float *partW, const double *sumsS, const double *sumsT,
    float *HA, float *HB,
    // gradients out
    float *g_f, float *g_p, float *g_newp, float *g_w1, float *g_w2, float *g_g1, float *g_b1, float *g_g2,
    float *g_b2, float *g_ws, float *g_bs, void *stream) {
    if (phases & 1) {
        if (zero_bytes) APN_TRY(apn_zero_fill(zero_base, zero_bytes, stream));
        APN_TRY(apn_sa_bwd_prep(b, n, m, g_out, gs_b, gs_c, gs_m, out, relu, ysel, pack2, ws ? ft : nullptr, precision,
                                ws ? fidx : nullptr, ws, goa, accS, partWs, gip,
                                pcnt_poff ? pcnt_poff + 2ll * b * n : nullptr, stream));      // (the row map's per-cloud verdict on the picks)
    }
    if (phases & 2)
        APN_TRY(apn_sa_bwd_main(b, n, m, precision, radius, xyz, new_xyz, ft, idx, tmap, w1, w2, pack1, pack2, accS,
                                sumsS, count, train2, goa, ksel, accT, partW2, rowdst, GU, HA, HB, stream));
    if (phases & 4) {
        APN_TRY(apn_sa_bwd_point_grads(b, n, m, GU, pcnt_poff, geo, HA, HB, accT, sumsT, count, train1, pack1, ft, precision, xyz,
                                       new_xyz, w1, gip, radius, partW, g_f, g_p, g_newp, stream));
        APN_TRY(apn_sa_bwd_finalize(partW, apn_sa_bwd_weight_rows(b, n), radius, g_w1, partWs,
                                    apn_sa_bwd_prep_rows(b, m), g_ws, partW2, apn_sa_bwd_main_rows(b, m), g_w2, accS,
                                    sumsS, accT, sumsT, g_bs, g_g2, g_b2, g_g1, g_b1, stream));
    }
    return APN_OK;
}

// FPS (+ sampled coordinates), ball query and the neighbourhoods' occurrence statistics back-to-back: the index
// stage of a block.  temp (B,N) is filled with 1e10 here (subsample.py:94); with temp == null (n <= 16384) the
// sampler starts from 1e10 in registers and leaves no min-distances behind: one launch less.
// geo / dd (optional, nsample == 32): apn_sa_point_geo's outputs.
// The same for a level of an index PYRAMID (block k + 1 samples from block k's samples, in pick order): the sampler is
// apn_furthest_point_sampling_nested -- a copy of the previous level's first m picks wherever that level's record
// (tie_prev) says every arg-max up to there was unique, the full sampler elsewhere; tie_out: this level's record.
extern "C" int apn_sa_sample_seq_nested(int b, int n, int m, float radius, int nsample, const float *xyz,
                                        const int *tie_prev, int *tie_out, int *fidx, float *new_xyz, int *idx, void *geo,
                                        void *dd, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0 || !tie_out) return APN_EINVAL;
    APN_TRY(apn_furthest_point_sampling_nested(b, n, m, xyz, tie_prev, fidx, new_xyz, tie_out, stream));
    APN_TRY(apn_ball_query_zero(b, n, m, radius, nsample, new_xyz, xyz, idx, stream));
    if (geo) APN_TRY(apn_sa_point_geo(b, n, m, nsample, radius, xyz, new_xyz, idx, geo, dd, stream));
    return APN_OK;
}

extern "C" int apn_sa_sample_seq(int b, int n, int m, float radius, int nsample, const float *xyz,
                                 float *temp, int *fidx, float *new_xyz, int *idx, void *geo, void *dd,
                                 void *stream) {
    if (b <= 0 || n <= 0 || m <= 0) return APN_EINVAL;
    if (temp) {   // 1e10f = 0x501502F9 as 32-bit words (a kernel, not a memset: the index stage replays from hipGraphs)
        long long blocks = ((long long)b * n + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(apn::fill32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned *)temp,
                           0x501502F9u, (long long)b * n);
        APN_LAUNCH_CHECK();
    }
    APN_TRY(apn_furthest_point_sampling_xyz(b, n, m, xyz, temp, fidx, new_xyz, stream));
    APN_TRY(apn_ball_query_zero(b, n, m, radius, nsample, new_xyz, xyz, idx, stream));
    if (geo) APN_TRY(apn_sa_point_geo(b, n, m, nsample, radius, xyz, new_xyz, idx, geo, dd, stream));
    return APN_OK;
}
