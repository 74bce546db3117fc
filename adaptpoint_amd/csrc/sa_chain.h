// sa_chain.h -- pieces shared by the kernels of the fused set-abstraction chain (sa_fused.hip, sa_glue.hip):
// BatchNorm folded inside its consumer kernel instead of in a launch of its own.
#pragma once
#include "apn_common.h"

namespace apn {

// BatchNorm parameters of one layer as the kernels take them (torch.nn.BatchNorm semantics).
struct BnArgs {
    const float *gamma, *beta;       // may be null (affine = False)
    float *rmean, *rvar;             // running buffers (null: no tracking)
    long long *nbt;                  // num_batches_tracked (null: not incremented)
    float eps, momentum;
    int training;                    // batch statistics (and running-buffer update) or the running buffers
    double count;                    // positions behind the sums on THIS rank (reduced sums carry their own)
};

// {scale, shift, mean, invstd} of channel i from its float64 sum / sum of squares over `count` positions;
// `writer` (one thread per channel in the whole grid) also updates the running buffers and leaves the four
// values in pack[4][c] for the backward.
// (g, b: the channel's gamma and beta, loaded by the caller TOGETHER with its first loads: read here, behind the fold's
// barriers, they were one more dependent memory round trip in every consumer's prologue)
__device__ __forceinline__ void bn_channel(const BnArgs &bn, int c, int i, double sum, double sumsq, double count,
                                           bool writer, float *__restrict__ pack, float &scale, float &shift,
                                           double g, double b) {
    double mean, var;
    if (bn.training) {
        mean = sum / count;
        var = sumsq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (writer && bn.rmean) {
            const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
            bn.rmean[i] = (float)((1.0 - bn.momentum) * (double)bn.rmean[i] + bn.momentum * mean);
            bn.rvar[i] = (float)((1.0 - bn.momentum) * (double)bn.rvar[i] + bn.momentum * unbiased);
        }
    } else {
        mean = bn.rmean[i];
        var = bn.rvar[i];
    }
    const double inv = 1.0 / sqrt(var + (double)bn.eps);
    scale = (float)(g * inv);
    shift = (float)(b - mean * g * inv);
    if (writer) {
        pack[i] = scale;
        pack[c + i] = shift;
        pack[2 * c + i] = (float)mean;
        pack[3 * c + i] = (float)inv;
    }
}

__device__ __forceinline__ void bn_channel(const BnArgs &bn, int c, int i, double sum, double sumsq, double count,
                                           bool writer, float *__restrict__ pack, float &scale, float &shift) {
    bn_channel(bn, c, i, sum, sumsq, count, writer, pack, scale, shift, bn.gamma ? (double)bn.gamma[i] : 1.0,
               bn.beta ? (double)bn.beta[i] : 0.0);
}

}  // namespace apn
