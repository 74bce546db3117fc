// sa_glue.hip -- the small kernels between the fused set-abstraction passes
// (csrc/sa_fused.hip): BatchNorm folding and running-statistics update, the
// (B,M,64) <-> (B,64,M) layout changes with the block's skip branch (Conv1d on the
// sampled points, add, ReLU) and its backward folded in, the per-channel constants of
// the backward, and the "everything downstream of dL/dy1 is linear" products.  They
// replace ~150 tiny PyTorch launches per step (and two pathological long-K rocBLAS
// GEMMs), all graph-capturable: no host reads, no allocation.  Round 3: the BatchNorm folds and
// per-channel constants that used to be launches of their own (bn_fold x2, bwd_consts2, bwd_consts1) are
// PROLOGUES of their consumer kernels: per-channel sums cross workgroups through the integer accumulator sets
// of apn_common.h (order-independent: the same bits every run) or through a few dozen partial rows, and every
// consumer workgroup folds them itself.  Weight gradients leave their producers as one partial row per
// workgroup and are summed in float64, in a fixed order, by the last launch (float atomics are used only
// for the scatters into per-point buffers).  bn_fold / reduce_rows remain for the width-generic family.
#include "apn_common.h"
#include "sa_chain.h"
#include "apn_mfma.h"

namespace apn {

// Diagnostics (builds with -DAPN_WG_STAMPS only: scripts/stamp_glue.py): wall-clock stamps of a workgroup's phases,
// 16 per workgroup, written by thread 0 -- where the few microseconds of these latency-bound kernels go.
#ifdef APN_WG_STAMPS
__device__ unsigned long long *d_wg_stamps = nullptr;
__device__ __forceinline__ void wg_stamp(int k) {
    unsigned long long *st = d_wg_stamps;
    if (st && threadIdx.x == 0)
        st[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + k] = wall_clock64();
}
#else
__device__ __forceinline__ void wg_stamp(int) {}
#endif


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// sums[c] = sum_rows part[row][c] in float64, for a workgroup of 1024 threads and
// ncol a power of two in 4..128 (rows 16-byte aligned): threads = ncol/4 column quads x row groups, 16-byte loads
// 8 deep (a single workgroup pulls ~1 us per 16 KB when its loads are narrow and serial:
// 128 KB of rows was 8 us), then a tree over the row groups in LDS.  Fixed summation
// order: deterministic.  With part == null the caller's `sums` (already reduced, e.g.
// all-reduced over ranks) are copied instead.  Result valid in `out` after the trailing
// barrier.
__device__ __forceinline__ void block_sum_rows(const float *__restrict__ part, int rows, int ncol,
                                               const double *__restrict__ sums, double *out) {
    __shared__ double red[1024 * 4];
    const int t = threadIdx.x;
    if (!part) {
        if (t < ncol) out[t] = sums[t];
        __syncthreads();
        return;
    }
    const int quads = ncol >> 2, q = t % quads, g = t / quads, groups = 1024 / quads;
    const float4 *__restrict__ p4 = (const float4 *)part;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = g;
    for (; r + 7 * groups < rows; r += 8 * groups) {   // 8 independent 16-byte loads in flight
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p4[(size_t)(r + u * groups) * quads + q];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w;
        }
    }
    for (; r < rows; r += groups) {
        const float4 v = p4[(size_t)r * quads + q];
        s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
    double *mine = red + (size_t)g * ncol + 4 * q;
    mine[0] = s0; mine[1] = s1; mine[2] = s2; mine[3] = s3;
    __syncthreads();
    for (int st = groups >> 1; st > 0; st >>= 1) {
        if (g < st) {
            const double *other = mine + (size_t)st * ncol;
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[j] += other[j];
        }
        __syncthreads();
    }
    if (t < ncol) out[t] = red[t];
    __syncthreads();
}

// Row sums only (the SyncBatchNorm path: reduce -> all_reduce -> consumer with part == null).
// out[ncol] = this rank's position count, out[ncol + 1] = 1: summed over ranks with the rest they
// become the GLOBAL count and the world size, which the consumers below read from the reduced
// vector (ranks may hold different batch sizes; no host-side count * world).
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float *__restrict__ part, int rows,
                                                           int ncol, double count,
                                                           double *__restrict__ out) {
    __shared__ double s[128];
    block_sum_rows(part, rows, ncol, nullptr, s);
    if (threadIdx.x < ncol) out[threadIdx.x] = s[threadIdx.x];
    if (threadIdx.x == 0) { out[ncol] = count; out[ncol + 1] = 1.0; }
}

// Float64 sums over the partial rows of TWO column quads (16-byte column groups qa and qb of a
// row of `quads` quads), for a workgroup of 256 threads: thread = (row group t >> 1, quad
// t & 1), four 16-byte loads in flight, then a fixed tree over the 128 row groups
// (deterministic).  Result: red[0][0..3] = quad qa, red[1][0..3] = quad qb, valid after return.
// The per-channel folds below are sliced this way -- one workgroup per 4 channels instead of
// one workgroup pulling every column -- because a lone workgroup reads ~16 KB per microsecond:
// the whole 128-256 KB row set took 6-8 us, a 16 KB slice takes ~1.
__device__ __forceinline__ void slice_sum_rows(const float *__restrict__ part, int rows, int quads,
                                               int qa, int qb, double (*red)[4]) {
    const int t = threadIdx.x, grp = t >> 1;
    const float4 *__restrict__ p4 = reinterpret_cast<const float4 *>(part) + ((t & 1) ? qb : qa);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = grp;
    for (; r + 3 * 128 < rows; r += 4 * 128) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p4[(size_t)(r + 128 * u) * quads];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w;
        }
    }
    for (; r < rows; r += 128) {
        const float4 v = p4[(size_t)r * quads];
        s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
    red[t][0] = s0; red[t][1] = s1; red[t][2] = s2; red[t][3] = s3;
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) {
        if (grp < st) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[t][j] += red[t + 2 * st][j];
        }
        __syncthreads();
    }
}

// BatchNorm fold.  {sum[C], sumsq[C]} over `count` positions come as partial rows
// (part, rows) or as reduced sums.  pack = {scale, shift, mean, invstd}[C].
// training: batch statistics (biased variance), running buffers updated with the
// unbiased variance (torch.nn.BatchNorm semantics); otherwise the running buffers.
// Rider: sgn_out[i] = sign (+1/-1) of ANOTHER BatchNorm's gamma -- which extreme of y2
// the K-pool keeps.  Grid: C/4 workgroups of 256 threads, workgroup b owns channels 4b..4b+3.
__global__ __launch_bounds__(256) void bn_fold_kernel(
    const float *__restrict__ part, int rows, const double *__restrict__ sums_in, int c, double count,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
    float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt,
    int training, float *__restrict__ pack, const float *__restrict__ sgn_gamma, int sgn_c,
    float *__restrict__ sgn_out) {
    __shared__ double red[256][4];
    const int t = threadIdx.x, i = blockIdx.x * 4 + t;          // i: this thread's channel (t < 4)
    // everything this thread will need from memory is requested BEFORE the row sums, so that
    // its latency hides behind them
    float g32 = 1.0f, b32 = 0.0f, rm = 0.0f, rv = 1.0f;
    if (t < 4) {
        if (gamma) g32 = gamma[i];
        if (beta) b32 = beta[i];
        if (running_mean) { rm = running_mean[i]; rv = running_var[i]; }
    }
    if (blockIdx.x == 0) {
        if (t == 0 && training && nbt) *nbt += 1;
        if (sgn_out)
            for (int k = t; k < sgn_c; k += 256) sgn_out[k] = (!sgn_gamma || sgn_gamma[k] >= 0.0f) ? 1.0f : -1.0f;
    }
    double sum = 0.0, sumsq = 0.0;
    if (training) {
        if (part) {
            slice_sum_rows(part, rows, c / 2, blockIdx.x, c / 4 + blockIdx.x, red);
            if (t < 4) { sum = red[0][t]; sumsq = red[1][t]; }
        } else if (t < 4) {
            sum = sums_in[i];
            sumsq = sums_in[c + i];
            count = sums_in[2 * c];          // global count (reduce_rows_kernel)
        }
    }
    if (t >= 4) return;
    double mean, var;
    if (training) {
        mean = sum / count;
        var = sumsq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (running_mean) {
            const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
            running_mean[i] = (float)((1.0 - momentum) * rm + momentum * mean);
            running_var[i] = (float)((1.0 - momentum) * rv + momentum * unbiased);
        }
    } else {
        mean = rm;
        var = rv;
    }
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double g = g32, b = b32;
    pack[i] = (float)(g * inv);
    pack[c + i] = (float)(b - mean * g * inv);
    pack[2 * c + i] = (float)mean;
    pack[3 * c + i] = (float)inv;
}

// Stage the skip branch's operands of one 64-query tile with NT threads: sws = Ws (64x32,
// row stride WST) and sfi[q][i] = f[b][i][fidx[b][m0+q]] read from the point-major table(s).
// Loads are issued in independent batches (indices, then rows) so that the dependent
// index -> row chain is paid once, not once per loop iteration.  Optionally records the source
// point of each query.
template <int NT, int QT, int WST, int FST>
__device__ __forceinline__ void stage_skip_operands(int cloud, int n, int m, int m0,
                                                    const __bf16 *__restrict__ ft,
                                                    const __bf16 *__restrict__ ft_lo,
                                                    const int *__restrict__ fidx,
                                                    const float *__restrict__ ws,
                                                    float (*sws)[WST], float (*sfi)[FST], int *ssrc) {
    constexpr int G = NT / 32, QPT = QT / G, WPT = 64 / G;     // row groups; queries and weight rows per thread
    const int tid = threadIdx.x, i = tid & 31, q0 = tid >> 5;     // queries q0 + G t
    int src[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
        const int q = m0 + q0 + G * t;
        const int got = fidx[(size_t)cloud * m + (q < m ? q : m - 1)];
        src[t] = q < m ? got : -1;
    }
    float wv[WPT];
#pragma unroll
    for (int t = 0; t < WPT; ++t) wv[t] = ws[(q0 + G * t) * 32 + i];
    // (every load unconditional, on a clamped index, and selected afterwards: a load under a condition sits in a
    // basic block of its own and is waited for before the next one is issued -- four serial round trips here)
    const __bf16 *lo_tab = ft_lo ? ft_lo : ft;
    __bf16 hv[QPT], lv[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
        const size_t o = ((size_t)cloud * n + (src[t] < 0 ? 0 : src[t])) * 32 + i;
        hv[t] = ft[o];
        lv[t] = lo_tab[o];
    }
    float v[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
        const float f = (float)hv[t] + (ft_lo ? (float)lv[t] : 0.0f);
        v[t] = src[t] >= 0 ? f : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
        sfi[q0 + G * t][i] = v[t];
        if (ssrc && i == 0) ssrc[q0 + G * t] = src[t] < 0 ? 0 : src[t];
    }
#pragma unroll
    for (int t = 0; t < WPT; ++t) sws[q0 + G * t][i] = wv[t];
}

// out[b][c][m] = act( ysel[b][m][c] * scale2[c] + shift2[c] + identity[b][c][m] )   (C = 64)
// identity = Ws * f[b][:, fidx[b][m]] + bs  (the block's skip Conv1d on the sampled points,
// pointnext.py:157-161) when ws != null; act = ReLU when relu != 0 (pointnext.py:167-168).
// The sampled points' features come from the point-major bf16 table(s) of the block
// (hi [+ lo]): one contiguous 64-byte row per query instead of 32 strided 4-byte loads.
// One workgroup of 16 QT threads per tile of QT queries (QT = 64: sixteen waves; QT = 32: eight waves, twice the
// workgroups -- the headline shape has only B*M/64 = 256 tiles of 64, ONE workgroup per CU, and the kernel is a chain of
// dependent phases (index -> row gather, fold, normalise, product, store): with two workgroups per CU one's loads run
// under the other's arithmetic).  thread = (query, 4 channels): the query's 32 inputs sit in registers, the weight rows
// are 16-byte LDS broadcasts.
template <int QT>
__global__ __launch_bounds__(QT * 16) void fwd_out_kernel(int n, int m, const float *__restrict__ ysel,
                                                          const unsigned long long *__restrict__ acc2,
                                                          const double *__restrict__ sums2, BnArgs bn2,
                                                          float *__restrict__ pack2,
                                                          const __bf16 *__restrict__ ft,
                                                          const __bf16 *__restrict__ ft_lo,
                                                          const int *__restrict__ fidx,
                                                          const float *__restrict__ ws,
                                                          const float *__restrict__ bs, int relu,
                                                          float *__restrict__ out, float4 *__restrict__ zero,
                                                          long long zero_n4) {
    constexpr int NT = QT * 16, NW = NT / 64, HALVES = 64 / QT;   // waves; query groups per wave
    __shared__ float tile[QT][65];
    __shared__ __attribute__((aligned(16))) float sfi[QT][36];
    __shared__ double ftot[128];
    __shared__ float s_sc[64], s_sh[64];
    // BatchNorm-2 folded here (formerly a launch of its own): {sum, sumsq}[64] of y2 from the forward pass's
    // accumulator set (integer atomics: the same bits in every workgroup and every run) or from reduced sums
    const bool first = blockIdx.x == 0 && blockIdx.y == 0;
    critical_stream_priority();
    wg_stamp(0);
    double count2 = bn2.count;
    double g2pre = 1.0, b2pre = 0.0;               // requested with the first loads, used behind the barrier
    if (threadIdx.x < 64) {
        if (bn2.gamma) g2pre = (double)bn2.gamma[threadIdx.x];
        if (bn2.beta) b2pre = (double)bn2.beta[threadIdx.x];
    }
    if (bn2.training) {
        if (threadIdx.x < 128) ftot[threadIdx.x] = sums2 ? sums2[threadIdx.x] : acc_read(acc2, 128, threadIdx.x);
        if (sums2) count2 = sums2[128];
    }
    __shared__ __attribute__((aligned(16))) float sws[64][36];
    __shared__ float s_bs[64];
    const int cloud = blockIdx.y, m0 = blockIdx.x * QT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // ty = 0..NW-1
    if (threadIdx.x >= 128 && threadIdx.x < 192) s_bs[tx] = (ws && bs) ? bs[tx] : 0.0f;   // (one load, not one per output)
    float ys4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // j = query within tile, tx = channel (requested before the fold's barrier)
        const int q = m0 + ty + NW * k;
        ys4[k] = ysel[((size_t)cloud * m + (q < m ? q : m - 1)) * 64 + tx];     // (rows past m: dropped below)
    }
    if (ws) stage_skip_operands<NT, QT, 36, 36>(cloud, n, m, m0, ft, ft_lo, fidx, ws, sws, sfi, nullptr);
    wg_stamp(1);
    __syncthreads();
    wg_stamp(2);
    if (threadIdx.x < 64) {
        if (first && threadIdx.x == 0 && bn2.training && bn2.nbt) *bn2.nbt += 1;
        float sc, sh;
        bn_channel(bn2, 64, threadIdx.x, bn2.training ? ftot[threadIdx.x] : 0.0, bn2.training ? ftot[64 + threadIdx.x] : 0.0,
                   count2, first, pack2, sc, sh, g2pre, b2pre);
        s_sc[threadIdx.x] = sc;
        s_sh[threadIdx.x] = sh;
    }
    __syncthreads();
    {
        const float sc = s_sc[tx], sh = s_sh[tx];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = ty + NW * k;
            tile[j][tx] = m0 + j < m ? ys4[k] * sc + sh : 0.f;
        }
    }
    wg_stamp(3);
    __syncthreads();
    wg_stamp(4);
    if (ws) {
        // identity branch: Ws (64 x 32) times the tile's sampled features (32 x QT) as MFMA products on split (hi + lo)
        // operands -- six MFMAs per 32 x 32 block of outputs, one wave per block (the multiply-add loop over LDS that
        // stood here took 2.3 us of the kernel's 7: two LDS reads per multiply-add)
        constexpr int QB = QT / 32;
        if (ty < 2 * QB) {                               // wave-uniform
            const int cb = ty & 1, qb = ty >> 1, r = tx & 31, h = tx >> 5;
            f32x16 acc = {0};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float av[8], bv[8];
                const float4 a0 = *reinterpret_cast<const float4 *>(&sws[32 * cb + r][16 * st + 8 * h]);
                const float4 a1 = *reinterpret_cast<const float4 *>(&sws[32 * cb + r][16 * st + 8 * h + 4]);
                const float4 b0 = *reinterpret_cast<const float4 *>(&sfi[32 * qb + r][16 * st + 8 * h]);
                const float4 b1 = *reinterpret_cast<const float4 *>(&sfi[32 * qb + r][16 * st + 8 * h + 4]);
                av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
                acc = mfma<2>(make_frag<2>(av), make_frag<2>(bv), acc);
            }
            const int ql = 32 * qb + r;                  // lane = query (column), register e <-> channel (row)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = 32 * cb + acc_row(e, h);
                float v = tile[ql][c] + (acc[e] + s_bs[c]);
                if (relu) v = fmaxf(v, 0.0f);
                if (m0 + ql < m) out[((size_t)cloud * 64 + c) * m + m0 + ql] = v;
            }
        }
    } else {
        const int ql = tx % QT, hh = tx / QT;     // query of the tile; which four-channel group of the wave
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * (ty * HALVES + hh) + k;
            float v = tile[ql][c];
            if (relu) v = fmaxf(v, 0.0f);
            if (m0 + ql < m) out[((size_t)cloud * 64 + c) * m + m0 + ql] = v;
        }
    }
    wg_stamp(5);
    // The backward pass accumulates into A / gip / its accumulator sets with atomics; they are zeroed HERE, by the last forward
    // launch (race-free: their writers run after it), so that the backward needs no fill launch of its own.  At the END of
    // the kernel: the memory counter is in order, and a variable number of stores ahead of the loads made every wait for
    // a load wait for the whole fill.
    if (zero) {
        const long long nb = (long long)gridDim.x * gridDim.y, bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        for (long long i = bid * NT + threadIdx.x; i < zero_n4; i += nb * NT) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    wg_stamp(6);
}

// Backward entry.  g = g_out * [out > 0] (when relu) is dL/d(pre-activation).
//   goa[b][m][c] = g * scale2[c]                      -> the fused passes
//   partS[blk][128] = {S1 = sum g, S2 = sum g * yhat_sel} of the block's 64 queries
//                     (S1 is also dL/dbs), blk = blockIdx.y * gridDim.x + blockIdx.x
// and, with the skip branch (ws != null):
//   partWs[blk][64*32] = sum_q g[q][c] * fi[q][i]        (dL/dWs of the block's queries)
//   gip[b][n][i]   += sum_c ws[c][i] * g[q][c]  at n = fidx[b][q]   (dL/df through the skip,
//                     point-major rows: 128-byte atomic segments)
template <int QT>
__global__ __launch_bounds__(QT * 16) void bwd_prep_kernel(int n, int m, const float *__restrict__ g_out,
                                                           long long gs_b, long long gs_c, long long gs_m,
                                                           const float *__restrict__ out, int relu,
                                                           const float *__restrict__ ysel,
                                                           const float *__restrict__ pack2,
                                                           const __bf16 *__restrict__ ft,
                                                           const __bf16 *__restrict__ ft_lo,
                                                           const int *__restrict__ fidx,
                                                           const float *__restrict__ ws,
                                                           float *__restrict__ goa,
                                                           unsigned long long *__restrict__ accS,
                                                           float *__restrict__ partWs,
                                                           float *__restrict__ gip,
                                                           const int *__restrict__ dup) {
    // 16 QT threads per tile of QT queries, as in fwd_out_kernel
    constexpr int NT = QT * 16, NW = NT / 64, HALVES = 64 / QT;
    __shared__ float tile[QT][65];       // g[q][c]
    __shared__ float red[NW][2][64];
    __shared__ __attribute__((aligned(16))) float sfi[QT][36];
    __shared__ float sws[64][33];
    __shared__ int ssrc[QT];
    constexpr int DW = QT == 32 ? 4 : 2; // waves per 32-query block of dL/dfi (64-query tiles: 64 KB of static LDS allow two)
    __shared__ float sdfi[DW][QT][33];   // dL/dfi, the waves' partial blocks
    const int cloud = blockIdx.y, m0 = blockIdx.x * QT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // ty = 0..NW-1
    const int ql = tx % QT, hh = tx / QT;
    critical_stream_priority();
    wg_stamp(0);
    // the cloud's picks are m different points (the row map's verdict, index-stage data): every gip row receives ONE share,
    // so the shares are plain stores into the zeroed rows instead of float atomics (2 MB of them per launch: ~1 us of
    // the kernel's tail at the chip's atomic rate); a collapsed cloud (a point picked twice) keeps the atomics
    const bool distinct = dup && dup[cloud] == 0;
    // every load of the phase is issued before the first one is used (clamped indices, selected afterwards)
    const int qc = m0 + ql < m ? m0 + ql : m - 1;
    const float *outp = relu ? out : g_out;               // (no activation: any readable address, value unused)
    float gv[4], ov[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {        // read (c, query ql): coalesced over queries
        const int c = (ty + NW * k) * HALVES + hh;
        gv[k] = g_out[cloud * gs_b + c * gs_c + qc * gs_m];
        ov[k] = relu ? outp[((size_t)cloud * 64 + c) * m + qc] : 1.0f;
    }
    float ys[4];                         // ysel of (query ty + NW k, channel tx), used after the barrier
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = m0 + ty + NW * k;
        ys[k] = ysel[((size_t)cloud * m + (q < m ? q : m - 1)) * 64 + tx];
    }
    if (ws) stage_skip_operands<NT, QT, 33, 36>(cloud, n, m, m0, ft, ft_lo, fidx, ws, sws, sfi, ssrc);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = (ty + NW * k) * HALVES + hh;
        tile[ql][c] = (m0 + ql < m && ov[k] > 0.0f) ? gv[k] : 0.0f;
    }
    const float sc = pack2[tx], mu = pack2[128 + tx], iv = pack2[192 + tx];     // (requested in front of the barrier, not behind it)
    wg_stamp(1);
    __syncthreads();
    wg_stamp(2);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {        // (query j, channel tx): coalesced over channels
        const int j = ty + NW * k, q = m0 + j;
        if (q < m) {
            const float g = tile[j][tx];
            const float gs = g * sc;
            goa[((size_t)cloud * m + q) * 64 + tx] = gs;
            s1 += g;
            s2 += g * ((ys[k] - mu) * iv);
        }
    }
    red[ty][0][tx] = s1;
    red[ty][1][tx] = s2;
    wg_stamp(3);
    __syncthreads();
    wg_stamp(4);
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    float sacc = 0.0f;
    if (ty < 2) {
#pragma unroll
        for (int k = 0; k < NW; ++k) sacc += red[k][ty][tx];
    }
    wg_stamp(5);
    if (ws) {
        // The block's two products as MFMAs on split (hi + lo) operands, one wave per 32 x 32 block of results (the
        // multiply-add loops over LDS that stood here took 1.5 + 4 us of the kernel's 12):
        //   dL/dfi[q][i] = sum_c g[q][c] ws[c][i]   (QT x 32, K = 64)   -> point-major gip rows, atomics
        //   dL/dWs[c][i] = sum_q g[q][c] fi[q][i]   (64 x 32, K = QT)   -> this block's partial row
        // dL/dfi's K = 64 is dealt to FOUR waves (32-query tiles), one 16-deep step each, their partial blocks summed through LDS and added
        // to gip by all threads, two elements each: as one wave's four steps and sixteen atomics per lane this was the
        // longest stretch of the kernel (3.8 of 12 us, five of eight waves idle beside it).
        constexpr int QB = QT / 32;
        const int r = tx & 31, h = tx >> 5;
        if (ty < DW * QB) {                              // wave-uniform: query block ty / DW, steps 4/DW (ty % DW) ...
            const int qb = ty / DW, part = ty % DW;
            f32x16 acc = {0};
#pragma unroll
            for (int s4 = 0; s4 < 4 / DW; ++s4) {
                const int st = (4 / DW) * part + s4;
                float av[8], bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    av[j] = tile[32 * qb + r][16 * st + 8 * h + j];
                    bv[j] = sws[16 * st + 8 * h + j][r];
                }
                acc = mfma<2>(make_frag<2>(av), make_frag<2>(bv), acc);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)                  // lane = input i (column), register e <-> query (row)
                sdfi[part][32 * qb + acc_row(e, h)][r] = acc[e];
        } else if (ty < DW * QB + 2) {                   // channel block ty - DW QB
            const int cb = ty - DW * QB;
            f32x16 acc = {0};
#pragma unroll
            for (int st = 0; st < QT / 16; ++st) {
                float av[8], bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    av[j] = tile[16 * st + 8 * h + j][32 * cb + r];
                    bv[j] = sfi[16 * st + 8 * h + j][r];
                }
                acc = mfma<2>(make_frag<2>(av), make_frag<2>(bv), acc);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)                  // lane = input i, register e <-> channel
                partWs[blk * 2048 + (32 * cb + acc_row(e, h)) * 32 + r] = acc[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < QT * 32 / NT; ++k) {          // (query q, input i): a 128-byte atomic segment per 32 lanes
            const int o = threadIdx.x + NT * k, q = o >> 5, i = o & 31;
            float v = sdfi[0][q][i] + sdfi[1][q][i];
            if (DW == 4) v += sdfi[DW - 2][q][i] + sdfi[DW - 1][q][i];
            float *dst = gip + ((size_t)cloud * n + ssrc[q]) * 32 + i;
            if (m0 + q < m) {
                if (distinct) *dst = v;
                else atomicAdd(dst, v);
            }
        }
        wg_stamp(6);
    }
    // the accumulator set's adds LAST: the compiler placed a wait for every outstanding memory operation inside the loop
    // that followed them (2.5 us of every workgroup, waiting for atomics nobody reads here)
    if (ty < 2) acc_add(accS, 128, (int)(blk % ACC_COPIES), ty * 64 + tx, sacc);
    wg_stamp(7);
}

// Everything downstream of dL/dy1 = g_u*ca + yhat1*cb + cc, which is only ever needed summed
// per source point (G) and per query (H).  The backward pass (sa_fused.hip) left the sums of
// g_u (A per point, HA per query), of yhat1 per query (HB) and the occurrence statistics of
// every point (geo = {count, sum of relative positions}); with the batch constants ca, cb,
// cc (cabc) now known,
//   G[n][mid] = ca*A[n][mid] + cb*inv1*(W1[mid] . [geo_xyz[n]; count f_n] - count*mean1) + cc*count
//   H[q][mid] = ca*HA[q][mid] + cb*HB[q][mid] + cc*K
// (W1, f as the MFMA saw them: rounded to the operand precision).  One workgroup per tile of
// WG_PTS points of one cloud forms them in LDS and emits
//   partW[block][mid][38]  products over the tile for dL/dW1, columns
//       0..2   sum_n G[n][mid] * xyz[n][d]
//       3..5   sum_q H[q][mid] * new_xyz[q][d]     (query tile of the same flat index)
//       6..37  sum_n G[n][mid] * ft[n][i]
//     bwd_finalize sums the blocks in float64 and forms (col0-2 - col3-5)/r;
//   dL/df[b][i][n]  = sum_mid G[b][n][mid] * W1[mid][3+i]  (+ gip[b][n][i], the skip branch);
//   dL/dp[b][n][d] += sum_mid G[b][n][mid] * W1[mid][d] / r          (optional)
//   dL/dnew_p[q][d] = -sum_mid H[q][mid] * W1[mid][d] / r            (optional)
// One workgroup of 1024 threads per tile (there are B*N/64 tiles, two per CU).
constexpr int WG_PTS = 64;
// threads per workgroup of bwd_point_grads_kernel: 512 -- two workgroups fit a CU (1024: one, and the grid of B*N/64
// workgroups ran as two rounds of latency chains), rows of the tile per thread, mid channels per thread
constexpr int PG_NT = 512, PG_K = WG_PTS * 32 / PG_NT, PG_M = 32 * 64 / PG_NT;
constexpr int PG_RB = 8;        // rows of a point requested per round trip by the gather of bwd_point_grads_kernel

__global__ __launch_bounds__(PG_NT) void bwd_point_grads_kernel(
    int n, int total_q, int split, const float *__restrict__ GU, const int *__restrict__ pcnt,
    const int *__restrict__ poff, const long long *__restrict__ geo,
    const float *__restrict__ HA, const float *__restrict__ HB, const unsigned long long *__restrict__ accT,
    const double *__restrict__ sumsT, double count, int train1,
    const float *__restrict__ pack1, const __bf16 *__restrict__ ft,
    const __bf16 *__restrict__ ft_lo, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ w1, const float *__restrict__ gip,
    float inv_r, float *__restrict__ partW, float *__restrict__ g_f, float *__restrict__ g_p,
    float *__restrict__ g_q) {
    __shared__ __attribute__((aligned(16))) float swt[35][36];   // W1 transposed: [input col][mid]
    __shared__ __attribute__((aligned(16))) float swr[32][36];   // W1 rounded to the operand precision: [mid][f 0..31 | xyz]
    __shared__ float sG[WG_PTS][33];    // A, then G [point][mid]
    __shared__ float sH[WG_PTS][33];    // H [query][mid]
    __shared__ float sI[WG_PTS][33];    // gip [point][i]
    __shared__ float sB[WG_PTS][41];    // columns 0..2 xyz, 3..5 new_xyz (query rows), 6..37 ft
    __shared__ float sGeo[WG_PTS][4];   // count, sum of relative positions
    __shared__ float sc[5][32];         // ca, cb, cc, mean1, inv1
    const int tid = threadIdx.x;
    critical_stream_priority();
    wg_stamp(0);
    const int cloud = blockIdx.y, n0 = blockIdx.x * WG_PTS;
    const int block = cloud * gridDim.x + blockIdx.x;
    const size_t p0 = (size_t)cloud * n + n0;            // first point row of the tile
    const int q0 = block * WG_PTS;                       // first query row (flat)
    const int n_here = n - n0 < WG_PTS ? n - n0 : WG_PTS;
    // every load of the tile is issued FIRST, before the constants' own chain (accumulator set -> barrier): they depend on
    // nothing but the block's position (round 3 issued them behind that barrier: one more memory round trip, ~1.5 us of
    // the kernel's 10); clamped rows, selected afterwards: under conditions each load sat in a block of its own
    const int c_ld = tid & 31;
    float f_hi[PG_K], f_lo[PG_K], g_i[PG_K], h_a[PG_K], h_b[PG_K];
    // the point's rows of g_u (round 5: stored by the backward pass at consecutive places of the point-sorted order --
    // apn_sa_rowmap_many -- instead of added into A with float atomics): how many and where, requested FIRST: the rows
    // themselves depend on them (one more memory round trip, issued while the tile's other loads are in flight)
    // (mapping of this gather: thread = (point tid >> 3, four mid channels 4 (tid & 7)): a row is eight 16-byte loads, a
    // wave instruction covers eight rows)
    const int g_pt = tid >> 3, g_c4 = (tid & 7) * 4;
    int r_cnt, r_off;
    {
        const size_t gp = p0 + (g_pt < n_here ? g_pt : 0);
        r_cnt = g_pt < n_here ? pcnt[gp] : 0;
        r_off = poff[gp];
    }
    float4 a_g = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const __bf16 *lo_tab = ft_lo ? ft_lo : ft;
        const float *gip_tab = gip ? gip : HA;               // (no skip branch: any readable table, value unused)
#pragma unroll
        for (int k = 0; k < PG_K; ++k) {
            const int pt = (tid >> 5) + (PG_NT / 32) * k;
            const size_t pr = (p0 + (pt < n_here ? pt : 0)) * 32 + c_ld;
            const size_t qr = (size_t)(q0 + pt < total_q ? q0 + pt : 0) * 32 + c_ld;
            f_hi[k] = (float)ft[pr];
            f_lo[k] = (float)lo_tab[pr];
            g_i[k] = gip_tab[gip ? pr : qr];
            h_a[k] = HA[qr];
            h_b[k] = HB[qr];
        }
    }
    float xq_v[2] = {0.0f, 0.0f};
    long long geo_v = 0ll;
    if (tid < WG_PTS * 3) {
        const int pt = tid / 3, d = tid % 3;
        xq_v[0] = xyz[(p0 + (pt < n_here ? pt : 0)) * 3 + d];
        xq_v[1] = new_xyz[(size_t)(q0 + pt < total_q ? q0 + pt : 0) * 3 + d];
    } else if (tid >= PG_NT - WG_PTS * 4) {
        const int e = tid - (PG_NT - WG_PTS * 4), pt = e >> 2, j = e & 3;
        geo_v = geo[(p0 + (pt < n_here ? pt : 0)) * 4 + j];
    }
    for (int e = tid; e < 32 * 35; e += PG_NT) {
        const int mid = e / 35, col = e % 35;
        const float w = w1[e];
        const __bf16 hi = (__bf16)w;
        float wr = (float)hi;
        if (split) wr += (float)(__bf16)(w - wr);
        swt[col][mid] = w;
        swr[mid][col < 3 ? 32 + col : col - 3] = wr;
    }
    // The batch constants of dL/dy1 = g_u*ca + yhat1*cb + cc (formerly a launch of its own): BatchNorm-1's reduction
    // terms T1 = sum g_u, T2 = sum g_u*yhat1 from the backward pass's accumulator set (or reduced sums), every
    // thread for its own mid channel c = tid & 31 (same bits in every workgroup).
    // (threads 0..63 read one column of the set each; the 96 constants reach the other threads through LDS)
    if (tid < 64) {
        const int c = tid & 31;
        const float sca = pack1[c];
        float val = 0.0f;
        if (train1) {
            const double t = sumsT ? sumsT[tid] : acc_read(accT, 64, tid);      // tid < 32: T1, else T2
            const double cnt = sumsT ? sumsT[64] : count;
            val = (float)(-(double)sca * t / cnt);
        }
        sc[tid < 32 ? 2 : 1][c] = val;                    // cc from T1, cb from T2
        if (tid < 32) sc[0][c] = sca;
    } else if (tid >= 96 && tid < 160) sc[3 + ((tid - 96) >> 5)][tid & 31] = pack1[64 + tid - 96];   // mean1, inv1
    {
        // A[point][4 channels] = sum of the point's rows, ascending (a fixed order: bit-reproducible), EIGHT rows per round
        // trip: a point of the headline clouds is in 3.8 neighbourhoods on average and in more than eight 2 % of the time, so
        // all but a few threads are done after one batch (with four rows per batch a third of the points needed a second
        // dependent round trip: stamps, +1.5 us on the kernel's first phase); a hot point (collapsed clouds) loops on
        const float *rows = GU + (size_t)(r_cnt > 0 ? r_off : 0) * 32 + g_c4;
        for (int j0 = 0; j0 < r_cnt; j0 += PG_RB) {
            float4 v[PG_RB];
#pragma unroll
            for (int u = 0; u < PG_RB; ++u) {
                const int j = j0 + u < r_cnt ? j0 + u : r_cnt - 1;
                v[u] = *reinterpret_cast<const float4 *>(rows + (size_t)j * 32);
            }
#pragma unroll
            for (int u = 0; u < PG_RB; ++u)
                if (j0 + u < r_cnt) { a_g.x += v[u].x; a_g.y += v[u].y; a_g.z += v[u].z; a_g.w += v[u].w; }
        }
    }
    wg_stamp(1);
    __syncthreads();
    wg_stamp(2);
    {
        const int c = tid & 31;                           // fixed per thread: e = tid + 1024 k
        const float ca = sc[0][c], cb = sc[1][c], cc = sc[2][c];
#pragma unroll
        for (int k = 0; k < PG_K; ++k) {
            const int pt = (tid >> 5) + (PG_NT / 32) * k;
            const bool ok = pt < n_here;
            // (sG: filled below, by the gather's own mapping)
            sB[pt][6 + c] = ok ? f_hi[k] + (ft_lo ? f_lo[k] : 0.0f) : 0.0f;
            sI[pt][c] = (gip && ok) ? g_i[k] : 0.0f;
            sH[pt][c] = q0 + pt < total_q ? __builtin_fmaf(ca, h_a[k], __builtin_fmaf(cb, h_b[k], cc * 32.0f)) : 0.0f;
        }
    }
    sG[g_pt][g_c4] = a_g.x; sG[g_pt][g_c4 + 1] = a_g.y;          // (points past the cloud's end: r_cnt = 0, zeros)
    sG[g_pt][g_c4 + 2] = a_g.z; sG[g_pt][g_c4 + 3] = a_g.w;      // (rows of 33 floats: not 16-byte aligned)
    if (tid < WG_PTS * 3) {
        const int pt = tid / 3, d = tid % 3;
        const float xv = xq_v[0], qv = xq_v[1];
        sB[pt][d] = pt < n_here ? xv : 0.0f;
        sB[pt][3 + d] = q0 + pt < total_q ? qv : 0.0f;
        sB[pt][38 + d] = 0.0f;                            // pad columns
    } else if (tid >= PG_NT - WG_PTS * 4) {               // (the last four waves: the first three take the rows above)
        const int e = tid - (PG_NT - WG_PTS * 4), pt = e >> 2, j = e & 3;
        // the index stage's occurrence statistics (sa_geo.hip): {count, sum of relative positions in units of 2^-36}
        const long long gv = pt < n_here ? geo_v : 0ll;
        sGeo[pt][j] = j == 0 ? (float)gv : (float)((double)gv * (1.0 / 68719476736.0));
    }
    wg_stamp(3);
    __syncthreads();
    wg_stamp(4);

    const int tx = tid & 63, ty = tid >> 6;               // ty = 0..PG_NT/64-1, wave-uniform
    {   // G in place of A: thread (point tx, mid channels PG_M ty .. PG_M ty + PG_M - 1)
        const float cnt = sGeo[tx][0], gx = sGeo[tx][1], gy = sGeo[tx][2], gz = sGeo[tx][3];
        float fx[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) fx[i] = sB[tx][6 + i];
#pragma unroll
        for (int mm = 0; mm < PG_M; ++mm) {
            const int mid = PG_M * ty + mm;
            const float4 *wrow = reinterpret_cast<const float4 *>(swr[mid]);
            float accf = 0.0f;
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                const float4 w = wrow[i4];
                accf = __builtin_fmaf(w.x, fx[4 * i4], accf);
                accf = __builtin_fmaf(w.y, fx[4 * i4 + 1], accf);
                accf = __builtin_fmaf(w.z, fx[4 * i4 + 2], accf);
                accf = __builtin_fmaf(w.w, fx[4 * i4 + 3], accf);
            }
            const float4 wx = wrow[8];
            const float ys = wx.x * gx + wx.y * gy + wx.z * gz + cnt * accf;
            const float yhs = sc[4][mid] * (ys - cnt * sc[3][mid]);
            sG[tx][mid] = sc[0][mid] * sG[tx][mid] + sc[1][mid] * yhs + sc[2][mid] * cnt;
        }
    }
    wg_stamp(5);
    __syncthreads();
    wg_stamp(6);

    // The tile's two dense products as MFMAs on split (hi + lo) operands, one wave per 32 x 32 block of results (the
    // multiply-add loops over LDS that stood here took 6 of the workgroup's 12 us):
    //   partW[mid][col] = sum_pt X[pt][mid] sB[pt][col]   (32 x 38, K = 64; X = H for the query columns 3..5, else G)
    //   dL/df[i][pt]    = sum_mid W1[mid][3 + i] G[pt][mid] (+ gip)   (32 x 64, K = 32)
    {
        const int r = tx & 31, h = tx >> 5;
        if (ty < 3) {
            // wave 0: columns 0..31 but 3..5 (from G); wave 1: columns 32..37 (from G); wave 2: columns 3..5 (from H)
            const float (*X)[33] = ty == 2 ? sH : sG;
            const int col = ty == 1 ? 32 + r : r;
            const bool mine = ty == 0 ? !(r >= 3 && r <= 5) : (ty == 1 ? r < 6 : (r >= 3 && r <= 5));
            f32x16 acc = {0};
#pragma unroll
            for (int st = 0; st < WG_PTS / 16; ++st) {
                float av[8], bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    av[j] = X[16 * st + 8 * h + j][r];                       // rows = mid, k = point
                    bv[j] = mine ? sB[16 * st + 8 * h + j][col] : 0.0f;      // columns, k = point
                }
                acc = mfma<2>(make_frag<2>(av), make_frag<2>(bv), acc);
            }
            float *row = partW + (size_t)block * 32 * 38;
            if (mine) {
#pragma unroll
                for (int e = 0; e < 16; ++e) row[acc_row(e, h) * 38 + col] = acc[e];     // lane = column, register <-> mid
            }
        } else if (ty < 5) {
            const int pb = ty - 3;                               // point block
            f32x16 acc = {0};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float av[8], bv[8];
                const float4 a0 = *reinterpret_cast<const float4 *>(&swt[3 + r][16 * st + 8 * h]);
                const float4 a1 = *reinterpret_cast<const float4 *>(&swt[3 + r][16 * st + 8 * h + 4]);
                av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = sG[32 * pb + r][16 * st + 8 * h + j];
                acc = mfma<2>(make_frag<2>(av), make_frag<2>(bv), acc);
            }
            const int pt = 32 * pb + r;                          // lane = point (column), register <-> input channel
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = acc_row(e, h);
                if (pt < n_here) g_f[((size_t)cloud * 32 + i) * n + n0 + pt] = acc[e] + sI[pt][i];
            }
        }
    }
    if (!g_p && !g_q) {                                          // (wave-uniform; the common case)
        wg_stamp(7);
        return;
    }

    // dL/dp, dL/dnew_p: the point's / query's 32 mid values in registers, W1 columns as wave-uniform 16-byte broadcasts
    float gr[32];
#pragma unroll
    for (int mid = 0; mid < 32; ++mid) gr[mid] = sG[tx][mid];
    auto dot_col = [&](const float (&v)[32], int col) {
        const float4 *wcol = reinterpret_cast<const float4 *>(swt[col]);
        float s = 0.0f;
#pragma unroll
        for (int i4 = 0; i4 < 8; ++i4) {
            const float4 w = wcol[i4];
            s = __builtin_fmaf(v[4 * i4], w.x, s);
            s = __builtin_fmaf(v[4 * i4 + 1], w.y, s);
            s = __builtin_fmaf(v[4 * i4 + 2], w.z, s);
            s = __builtin_fmaf(v[4 * i4 + 3], w.w, s);
        }
        return s;
    };
    if (ty < 3 && g_p && tx < n_here) g_p[(p0 + tx) * 3 + ty] += dot_col(gr, ty) * inv_r;
    if (ty >= 4 && ty < 7 && g_q && q0 + tx < total_q) {
        float hr[32];
#pragma unroll
        for (int mid = 0; mid < 32; ++mid) hr[mid] = sH[tx][mid];
        g_q[(size_t)(q0 + tx) * 3 + ty - 4] = -dot_col(hr, ty - 4) * inv_r;
    }
    wg_stamp(7);
}

// Backward launch 4 of 4: every parameter gradient out of the partial rows (column sums in float64, fixed order)
// and the accumulator sets:
//   g_w1[mid][0..2] = (W[mid][0..2] - W[mid][3..5]) / r ; g_w1[mid][3+i] = W[mid][6+i],  W = sum_rows partW[row][32*38]
//   g_ws[c][i] = sum_rows partWs[row][64*32] ;  g_w2[c][mid] = sum_rows partW2[row][64*32]
//   g_bs[c] = S1[c], g_beta2 = S1 * gscale, g_gamma2 = S2 * gscale ; g_beta1 = T1 * gscale, g_gamma1 = T2 * gscale
// gscale = 1, or 1 / world with reduced sums (SyncBatchNorm: global sum / world = the mean over ranks of the
// rank-local sums -- what torch.nn.SyncBatchNorm + DistributedDataParallel leave in .grad; g_bs is a conv bias
// gradient and stays this rank's own sum).
// A workgroup owns 32 output elements x 16 row groups (round 3: 16 x 16 -- 64-byte segments of every 128-byte line, PMC
// 23 MB for 10 MB of rows); loads are issued 16 deep: the ~30 rows of a thread are two round trips instead of four.
constexpr int FIN_COLS = 32, FIN_GROUPS = 16;
__device__ __forceinline__ double col_sum(const float *__restrict__ base, int rows, int stride,
                                          int col, int g) {
    double s = 0.0;
    int r = g;
    for (; r + 15 * FIN_GROUPS < rows; r += 16 * FIN_GROUPS) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = base[(size_t)(r + FIN_GROUPS * u) * stride + col];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; r + 3 * FIN_GROUPS < rows; r += 4 * FIN_GROUPS) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = base[(size_t)(r + FIN_GROUPS * u) * stride + col];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (double)v[u];
    }
    for (; r < rows; r += FIN_GROUPS) s += (double)base[(size_t)r * stride + col];
    return s;
}

constexpr int FIN_W1 = 32 * 35, FIN_WS = 2048, FIN_W2 = 2048, FIN_SMALL = 64 * 3 + 32 * 2;
constexpr int FIN_TOTAL = FIN_W1 + FIN_WS + FIN_W2 + FIN_SMALL;

__global__ __launch_bounds__(FIN_COLS * FIN_GROUPS) void bwd_finalize_kernel(
    const float *__restrict__ partW, int rowsW, double inv_r, float *__restrict__ g_w1,
    const float *__restrict__ partWs, int rowsS, float *__restrict__ g_ws,
    const float *__restrict__ partW2, int rows2, float *__restrict__ g_w2,
    const unsigned long long *__restrict__ accS, const double *__restrict__ sumsS,
    const unsigned long long *__restrict__ accT, const double *__restrict__ sumsT,
    float *__restrict__ g_bs, float *__restrict__ g_g2, float *__restrict__ g_b2, float *__restrict__ g_g1,
    float *__restrict__ g_b1) {
    __shared__ double red[FIN_GROUPS][FIN_COLS];
    critical_stream_priority();
    const int o = threadIdx.x % FIN_COLS, g = threadIdx.x / FIN_COLS;
    const int e = blockIdx.x * FIN_COLS + o;
    double v = 0.0;
    float *dst = nullptr;
    if (e < FIN_W1) {
        const int mid = e / 35, col = e % 35;
        dst = g_w1 + e;
        if (col < 3)
            v = (col_sum(partW, rowsW, 1216, mid * 38 + col, g) -
                 col_sum(partW, rowsW, 1216, mid * 38 + 3 + col, g)) * inv_r;
        else
            v = col_sum(partW, rowsW, 1216, mid * 38 + 3 + col, g);
    } else if (e < FIN_W1 + FIN_WS) {
        if (g_ws) { dst = g_ws + (e - FIN_W1); v = col_sum(partWs, rowsS, 2048, e - FIN_W1, g); }
    } else if (e < FIN_W1 + FIN_WS + FIN_W2) {
        if (g_w2) { dst = g_w2 + (e - FIN_W1 - FIN_WS); v = col_sum(partW2, rows2, 2048, e - FIN_W1 - FIN_WS, g); }
    } else if (e < FIN_TOTAL) {
        if (g == 0) {
            const int k = e - FIN_W1 - FIN_WS - FIN_W2;
            if (k < 192) {                 // 0..63 g_bs, 64..127 g_beta2 (S1), 128..191 g_gamma2 (S2)
                const int c = k & 63, which = k >> 6;
                const double scale = (sumsS && which) ? 1.0 / sumsS[129] : 1.0;
                if (which == 0) {
                    // the skip bias: this rank's own sum of g (accS is this rank's set also when reduced sums are given)
                    if (g_bs) { dst = g_bs + c; v = acc_read(accS, 128, c); }
                } else {
                    const int col = which == 1 ? c : 64 + c;
                    dst = which == 1 ? (g_b2 ? g_b2 + c : nullptr) : (g_g2 ? g_g2 + c : nullptr);
                    if (dst) v = (sumsS ? sumsS[col] : acc_read(accS, 128, col)) * scale;
                }
            } else {                       // 192..223 g_beta1 (T1), 224..255 g_gamma1 (T2)
                const int c = (k - 192) & 31, which = (k - 192) >> 5;
                const double scale = sumsT ? 1.0 / sumsT[65] : 1.0;
                dst = which == 0 ? (g_b1 ? g_b1 + c : nullptr) : (g_g1 ? g_g1 + c : nullptr);
                if (dst) v = (sumsT ? sumsT[32 * which + c] : acc_read(accT, 64, 32 * which + c)) * scale;
            }
        }
    }
    red[g][o] = v;
    __syncthreads();
    if (g == 0 && dst) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < FIN_GROUPS; ++k) acc += red[k][o];
        *dst = (float)acc;
    }
}

// Accumulator set -> float64 sums (the SyncBatchNorm path: reduce -> all_reduce -> consumer with sums):
// out[0..ncol) = the totals, out[ncol] = this rank's position count, out[ncol + 1] = 1 (summed over ranks:
// the GLOBAL count and the world size, which the consumers read from the reduced vector).
__global__ __launch_bounds__(128) void reduce_acc_kernel(const unsigned long long *__restrict__ acc, int ncol,
                                                         double count, double *__restrict__ out) {
    const int t = threadIdx.x;
    if (t < ncol) out[t] = acc_read(acc, ncol, t);
    if (t == 0) { out[ncol] = count; out[ncol + 1] = 1.0; }
}

}  // namespace apn

#define APN_ST ((hipStream_t)stream)

extern "C" int apn_sa_reduce_rows(const float *part, int rows, int ncol, double count, double *out,
                                  void *stream) {
    if (rows < 0 || ncol < 4 || ncol > 128 || (1024 % ncol) || !part || !out) return APN_EINVAL;
    if ((uintptr_t)part & 15) return APN_EINVAL;
    hipLaunchKernelGGL(apn::reduce_rows_kernel, dim3(1), dim3(1024), 0, APN_ST, part, rows, ncol, count,
                       out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_reduce_acc(const void *acc, int ncol, double count, double *out, void *stream) {
    if (ncol < 1 || ncol > 128 || !acc || !out) return APN_EINVAL;
    hipLaunchKernelGGL(apn::reduce_acc_kernel, dim3(1), dim3(128), 0, APN_ST, (const unsigned long long *)acc, ncol,
                       count, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bn_fold(const float *part, int rows, const double *sums, int c, double count,
                              const float *gamma, const float *beta, float eps, float momentum,
                              float *running_mean, float *running_var, void *num_batches_tracked,
                              int training, float *pack, const float *sgn_gamma, int sgn_c,
                              float *sgn_out, void *stream) {
    if (c < 4 || c > 1024 || (c & 3) || !pack || sgn_c > 1024) return APN_EINVAL;
    if (training && !part && !sums) return APN_EINVAL;
    if (!training && (!running_mean || !running_var)) return APN_EINVAL;
    if (part && ((uintptr_t)part & 15)) return APN_EINVAL;
    hipLaunchKernelGGL(apn::bn_fold_kernel, dim3(c / 4), dim3(256), 0, APN_ST, part, rows, sums, c, count,
                       gamma, beta, eps, momentum, running_mean, running_var,
                       (long long *)num_batches_tracked, training, pack, sgn_gamma, sgn_c, sgn_out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// queries per workgroup of the output / backward-entry kernels: 32 while tiles of 64 would leave CUs with one workgroup
static int apn_sa_glue_tile(int b, int m) { return (long long)b * ((m + 63) / 64) <= 512 ? 32 : 64; }

extern "C" int apn_sa_fwd_out(int b, int n, int m, const float *ysel, const void *acc2, const double *sums2,
                              const float *g2, const float *b2, float *rm2, float *rv2, void *nbt2, float eps2,
                              float mom2, int train2, double count, float *pack2, const void *ft, int precision,
                              const int *fidx, const float *ws, const float *bs, int relu, float *out,
                              float *zero_base, long long zero_floats, void *stream) {
    if (b <= 0 || m <= 0 || b > 65535 || !ysel || !pack2 || !out) return APN_EINVAL;
    if (train2 && !acc2 && !sums2) return APN_EINVAL;
    if (!train2 && (!rm2 || !rv2)) return APN_EINVAL;
    if (ws && (!ft || !fidx || n <= 0 || (precision != 1 && precision != 2))) return APN_EINVAL;
    const __bf16 *hi = (const __bf16 *)ft;
    const __bf16 *lo = (ws && precision == 2) ? hi + (size_t)b * n * 32 : nullptr;
    if (zero_base && (zero_floats < 0 || (zero_floats & 3) || ((uintptr_t)zero_base & 15))) return APN_EINVAL;
    apn::BnArgs bn{g2, b2, rm2, rv2, (long long *)nbt2, eps2, mom2, train2, count};
    if (apn_sa_glue_tile(b, m) == 32)
        hipLaunchKernelGGL(apn::fwd_out_kernel<32>, dim3((m + 31) / 32, b), dim3(512), 0, APN_ST, n, m, ysel,
                           (const unsigned long long *)(sums2 ? nullptr : acc2), sums2, bn, pack2, hi, lo, fidx, ws, bs,
                           relu, out, (float4 *)zero_base, zero_base ? zero_floats / 4 : 0);
    else
        hipLaunchKernelGGL(apn::fwd_out_kernel<64>, dim3((m + 63) / 64, b), dim3(1024), 0, APN_ST, n, m, ysel,
                           (const unsigned long long *)(sums2 ? nullptr : acc2), sums2, bn, pack2, hi, lo, fidx, ws, bs,
                           relu, out, (float4 *)zero_base, zero_base ? zero_floats / 4 : 0);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_prep_rows(int b, int m) {
    const int qt = apn_sa_glue_tile(b, m);
    return b * ((m + qt - 1) / qt);
}

extern "C" int apn_sa_bwd_prep(int b, int n, int m, const float *g_out, long long gs_b,
                               long long gs_c, long long gs_m, const float *out, int relu,
                               const float *ysel, const float *pack2, const void *ft, int precision,
                               const int *fidx, const float *ws, float *goa, void *accS,
                               float *partWs, float *gip, const int *dup, void *stream) {
    if (b <= 0 || m <= 0 || b > 65535 || !g_out || !ysel || !pack2 || !goa || !accS) return APN_EINVAL;
    if (relu && !out) return APN_EINVAL;
    if (ws && (!ft || !fidx || !partWs || !gip || n <= 0 || (precision != 1 && precision != 2)))
        return APN_EINVAL;
    const __bf16 *hi = (const __bf16 *)ft;
    const __bf16 *lo = (ws && precision == 2) ? hi + (size_t)b * n * 32 : nullptr;
    if (apn_sa_glue_tile(b, m) == 32)
        hipLaunchKernelGGL(apn::bwd_prep_kernel<32>, dim3((m + 31) / 32, b), dim3(512), 0, APN_ST, n, m, g_out,
                           gs_b, gs_c, gs_m, out, relu, ysel, pack2, hi, lo, fidx, ws, goa, (unsigned long long *)accS,
                           partWs, gip, dup);
    else
        hipLaunchKernelGGL(apn::bwd_prep_kernel<64>, dim3((m + 63) / 64, b), dim3(1024), 0, APN_ST, n, m, g_out,
                           gs_b, gs_c, gs_m, out, relu, ysel, pack2, hi, lo, fidx, ws, goa, (unsigned long long *)accS,
                           partWs, gip, dup);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

#ifdef APN_WG_STAMPS
// (diagnostic builds only) attach a buffer of 16 stamps per workgroup for the next launches of this file's kernels
extern "C" __attribute__((visibility("default"))) int apn_sa_debug_wg_stamps(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(apn::d_wg_stamps), &buf, sizeof(buf));
}
#endif

extern "C" int apn_sa_bwd_weight_rows(int b, int n) { return b * ((n + apn::WG_PTS - 1) / apn::WG_PTS); }

extern "C" int apn_sa_bwd_point_grads(int b, int n, int m, const float *GU, const int *pcnt_poff, const void *geo,
                                      const float *HA, const float *HB, const void *accT, const double *sumsT,
                                      double count, int train1, const float *pack1, const void *ft, int precision,
                                      const float *xyz, const float *new_xyz, const float *w1,
                                      const float *gip, float radius, float *partW, float *g_f,
                                      float *g_p, float *g_newp, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0 || b > 65535) return APN_EINVAL;
    if (!GU || !pcnt_poff || !geo || !HA || !HB || (!accT && !sumsT) || !pack1 || !ft || !xyz || !new_xyz || !w1 || !partW ||
        !g_f)
        return APN_EINVAL;
    if (precision != 1 && precision != 2) return APN_EINVAL;
    if (m > n) return APN_EINVAL;                    // query tiles are walked with the point tiles
    const __bf16 *hi = (const __bf16 *)ft;
    const __bf16 *lo = precision == 2 ? hi + (size_t)b * n * 32 : nullptr;
    hipLaunchKernelGGL(apn::bwd_point_grads_kernel, dim3((n + apn::WG_PTS - 1) / apn::WG_PTS, b),
                       dim3(apn::PG_NT), 0, APN_ST, n, b * m, precision == 2 ? 1 : 0, GU, pcnt_poff,
                       pcnt_poff + (size_t)b * n, (const long long *)geo, HA, HB,
                       (const unsigned long long *)accT, sumsT, count, train1,
                       pack1, hi, lo, xyz, new_xyz, w1, gip, 1.0f / radius, partW, g_f, g_p, g_newp);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_finalize(const float *partW, int rows_w, float radius, float *g_w1,
                                   const float *partWs, int rows_s, float *g_ws, const float *partW2, int rows_2,
                                   float *g_w2, const void *accS, const double *sumsS,
                                   const void *accT, const double *sumsT, float *g_bs, float *g_g2, float *g_b2,
                                   float *g_g1, float *g_b1, void *stream) {
    if (!partW || !g_w1 || (g_ws && !partWs) || (g_w2 && !partW2) || !accS || (!accT && !sumsT))
        return APN_EINVAL;
    hipLaunchKernelGGL(apn::bwd_finalize_kernel, dim3((apn::FIN_TOTAL + apn::FIN_COLS - 1) / apn::FIN_COLS),
                       dim3(apn::FIN_COLS * apn::FIN_GROUPS), 0, APN_ST, partW,
                       rows_w, 1.0 / (double)radius, g_w1, partWs, rows_s, g_ws, partW2, rows_2, g_w2,
                       (const unsigned long long *)accS, sumsS, (const unsigned long long *)accT, sumsT, g_bs,
                       g_g2, g_b2, g_g1, g_b1);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
