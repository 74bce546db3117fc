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
//   forward : 1 = prep + stats1          | 2 = fold1 + main     | 4 = fold2 + out
//   backward: 1 = zero + prep            | 2 = consts2 + main   | 4 = consts1 + point grads + finalize
// With part pointers the consumer kernels sum the partial rows themselves (single rank:
// phases = 7, one call); with `sums*` pointers (float64, already reduced over ranks) they
// use those instead.
#include "apn_common.h"

#define APN_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

extern "C" int apn_sa_forward_seq(
    int phases, int precision, int b, int n, int m, float radius, const float *xyz, const float *new_xyz,
    const float *f, const int *idx, const int *tmap, const int *fidx, const float *w1, const float *w2,
    const float *ws, const float *bs,
    const float *g1, const float *b1, float *rm1, float *rv1, void *nbt1, float eps1, float mom1,
    int train1,
    const float *g2, const float *b2, float *rm2, float *rv2, void *nbt2, float eps2, float mom2,
    int train2,
    double count, int relu, void *ft, float *part1, float *part2, const double *sums1,
    const double *sums2, float *pack1, float *pack2, float *sgn2, float *ysel, void *ksel,
    float *out, float *zero_base, long long zero_floats, void *stream) {
    const int rows = apn_sa_grid_rows(b, m, tmap != nullptr);
    if (phases & 1) {
        APN_TRY(apn_sa_prep_features(b, 32, n, f, ft, precision, stream));
        if (train1)
            APN_TRY(apn_sa_fwd_stats1(b, n, m, 32, 32, 64, 32, precision, radius, xyz, new_xyz, ft, idx,
                                      tmap, w1, part1, stream));
    }
    if (phases & 2) {
        APN_TRY(apn_sa_bn_fold(sums1 ? nullptr : part1, rows, sums1, 32, count, g1, b1, eps1, mom1,
                               rm1, rv1, nbt1, train1, pack1, g2, 64, sgn2, stream));
        APN_TRY(apn_sa_fwd_main(b, n, m, 32, 32, 64, 32, precision, radius, xyz, new_xyz, ft, idx, tmap, w1,
                                w2, pack1, pack1 + 32, sgn2, ysel, ksel, part2, stream));
    }
    if (phases & 4) {
        APN_TRY(apn_sa_bn_fold(sums2 ? nullptr : part2, rows, sums2, 64, count, g2, b2, eps2, mom2,
                               rm2, rv2, nbt2, train2, pack2, nullptr, 0, nullptr, stream));
        APN_TRY(apn_sa_fwd_out(b, n, m, ysel, pack2, ws ? ft : nullptr, precision,
                               ws ? fidx : nullptr, ws, bs, relu, out, zero_base, zero_floats, stream));
    }
    return APN_OK;
}

extern "C" int apn_sa_backward_seq(
    int phases, int precision, int b, int n, int m, float radius, const float *xyz, const float *new_xyz,
    const float *f, const int *idx, const int *tmap, const int *fidx, const float *w1, const float *w2,
    const float *ws, const void *ft, const float *pack1, const float *pack2, const float *ysel,
    const void *ksel, const float *out, int relu, int train1, int train2, double count,
    const float *g_out, long long gs_b, long long gs_c, long long gs_m,
    // zero-filled here unless zero_bytes == 0 (then apn_sa_fwd_out cleared them): A (B*N*32) | geo (B*N*4) |
    // gip (B*N*32, only with ws) by one memset (phase 1); the accumulators gw2_acc (copies x 64*32) and gram (copies x (32*32 + 32)), copies =
    // apn_sa_bwd_acc_copies(), by the consts2 launch (phase 2); g_w2 (64*32) is written by consts1
    float *zero_base, size_t zero_bytes, float *g_w2, float *gw2_acc, float *gram, float *A, float *geo, float *gip,
    // scratch
    float *goa, float *partS, float *partWs, float *partT, float *partW, const double *sumsS,
    const double *sumsT, float *d2e2, float *qm, float *evec, float *cabc, float *HA, float *HB,
    // gradients out
    float *g_f, float *g_p, float *g_newp, float *g_w1, float *g_g1, float *g_b1, float *g_g2,
    float *g_b2, float *g_ws, float *g_bs, void *stream) {
    const int rows_t = apn_sa_bwd_main_rows(b, m);
    const int prow = apn_sa_bwd_prep_rows(b, m);
    if (phases & 1) {
        if (zero_bytes) {                 // 0: the forward's last launch already cleared the region
            hipError_t me = hipMemsetAsync(zero_base, 0, zero_bytes, (hipStream_t)stream);
            if (me != hipSuccess) return (int)me;
        }
        APN_TRY(apn_sa_bwd_prep(b, n, m, g_out, gs_b, gs_c, gs_m, out, relu, ysel, pack2, ws ? ft : nullptr, precision,
                                ws ? fidx : nullptr, ws, goa, partS, partWs, gip, stream));
    }
    if (phases & 2) {
        APN_TRY(apn_sa_bwd_consts2(sumsS ? nullptr : partS, prow, sumsS, pack2, w2, count, train2,
                                   d2e2, qm, evec, g_g2, g_b2, gw2_acc, gram, stream));
        APN_TRY(apn_sa_bwd_main(b, n, m, 32, 32, 64, 32, precision, radius, xyz, new_xyz, ft, idx, tmap, w1,
                                w2, pack1, qm, evec, goa, ksel, partT, gw2_acc, gram, A, geo, HA, HB,
                                stream));
    }
    if (phases & 4) {
        APN_TRY(apn_sa_bwd_consts1(sumsT ? nullptr : partT, rows_t, sumsT, pack1, count, train1, cabc,
                                   g_g1, g_b1, w2, d2e2, gram, gw2_acc, g_w2, stream));
        APN_TRY(apn_sa_bwd_point_grads(b, n, m, A, geo, HA, HB, cabc, pack1, ft, precision, xyz,
                                       new_xyz, w1, gip, radius, partW, g_f, g_p, g_newp, stream));
        APN_TRY(apn_sa_bwd_finalize(partW, apn_sa_bwd_weight_rows(b, n), radius, g_w1, partWs, prow,
                                    g_ws, partS, g_bs, stream));
    }
    return APN_OK;
}

// FPS (+ sampled coordinates) and ball query back-to-back: the index stage of a block.
// temp (B,N) is filled with 1e10 here (subsample.py:94); with temp == null (n <= 16384) the
// sampler starts from 1e10 in registers and leaves no min-distances behind: one launch less.
extern "C" int apn_sa_sample_seq(int b, int n, int m, float radius, int nsample, const float *xyz,
                                 float *temp, int *fidx, float *new_xyz, int *idx, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0) return APN_EINVAL;
    if (temp) {   // 1e10f = 0x501502F9 as 32-bit words
        hipError_t me = hipMemsetD32Async((hipDeviceptr_t)temp, 0x501502F9, (size_t)b * n,
                                          (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    }
    APN_TRY(apn_furthest_point_sampling_xyz(b, n, m, xyz, temp, fidx, new_xyz, stream));
    APN_TRY(apn_ball_query_zero(b, n, m, radius, nsample, new_xyz, xyz, idx, stream));
    return APN_OK;
}
