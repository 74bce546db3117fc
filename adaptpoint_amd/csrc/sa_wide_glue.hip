// sa_wide_glue.hip -- the small kernels around the width-generic fused passes (csrc/sa_wide.hip):
// operand images, the upstream gradient's layout change with BatchNorm-2's row sums, the
// per-channel constants of the two BatchNorm backwards, the occurrence statistics of the points
// and the per-point / per-query terms of dL/dy1.  Each replaces a chain of 5-20 tiny tensor ops
// (an eager step of one block was ~200 launches, most of them 3-5 us of launch latency).
// Dense products whose contraction runs over points or channels (U = W1f f, dL/df = G W1f,
// dL/dW1 = G^T f, Qm = W2^T D2 W2, W2 Gram) stay library GEMMs in the caller.
#include "apn_common.h"
#include "apn_mfma.h"

namespace apn {

// ------------------------------------------------------------------------------------------
// B image of Bm (Kd x Nc): rows k < K0 from src0, the rest from src1 (both row-major with Nc
// columns; trans0: src0 is stored (Nc x K0) and read transposed).  One thread per 16-byte word
// [cb][kc][j][s][part][lane] (layout: adaptpoint_amd/fused_wide.py::mfma_b_image).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_image_kernel(const float *__restrict__ src0, int K0, int trans0,
                                                         const float *__restrict__ src1, int Kd, int Nc, int ct,
                                                         uint4 *__restrict__ img) {
    const int nkc = Kd / 32, ncb = Nc / (32 * ct);
    const int total = ncb * nkc * ct * 2 * 2 * 64;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= total) return;
    const int lane = w & 63, part = (w >> 6) & 1, s = (w >> 7) & 1;
    int rest = w >> 8;
    const int j = rest % ct; rest /= ct;
    const int kc = rest % nkc;
    const int cb = rest / nkc;
    const int col = (cb * ct + j) * 32 + (lane & 31), k0 = kc * 32 + s * 16 + (lane >> 5) * 8;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        float v;
        if (k < K0) v = trans0 ? src0[(size_t)col * K0 + k] : src0[(size_t)k * Nc + col];
        else v = src1[(size_t)(k - K0) * Nc + col];
        const __bf16 hi = (__bf16)v;
        o[e] = part == 0 ? hi : (__bf16)(v - (float)hi);
    }
    img[w] = __builtin_bit_cast(uint4, o);
}

// ------------------------------------------------------------------------------------------
// Upstream gradient g (B,O,M; any strides, a broadcast is never materialised), masked by the block's final ReLU
// when out_act (the block's output) is given ->
//   goa (B,M,O) = g * scale2 (the gradient that reaches y2 at the pooled slot),
//   partS[b * mt + tile][2*O] = {sum_m g, sum_m g * yhat_sel}, yhat_sel = (ysel - mean2) * invstd2.
// Block = 64 queries x 64 channels through an LDS tile (reads coalesced along m, writes along c).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_bwd_prep_kernel(int m, int O, const float *__restrict__ g,
                                                            long long gs_b, long long gs_c, long long gs_m,
                                                            const float *__restrict__ ysel,
                                                            const float *__restrict__ pack2,
                                                            const float *__restrict__ out_act,
                                                            float *__restrict__ gpre,
                                                            float *__restrict__ goa, float *__restrict__ partS) {
    __shared__ float tile[64][65];
    __shared__ float red[2][4][64];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, m0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 64; cc += 4) {          // tile[c][m]: coalesced along m
        const int q = m0 + tx;
        float gv = q < m ? g[b * gs_b + (long long)(c0 + cc) * gs_c + q * gs_m] : 0.0f;
        // the block ended in a ReLU over (pooled + skip): its output (B,O,M) tells where the gradient passes
        if (out_act && q < m && !(out_act[((size_t)b * O + c0 + cc) * m + q] > 0.0f)) gv = 0.0f;
        tile[cc][tx] = gv;
    }
    __syncthreads();
    const int c = c0 + tx;                          // this thread's channel; queries ty, ty+4, ...
    const float sc = pack2[c], mu = pack2[2 * O + c], iv = pack2[3 * O + c];
    float s1 = 0.0f, s2 = 0.0f;
    for (int qq = ty; qq < 64; qq += 4) {
        const int q = m0 + qq;
        if (q < m) {
            const float gv = tile[tx][qq];
            const size_t o = ((size_t)b * m + q) * O + c;
            goa[o] = gv * sc;
            if (gpre) gpre[o] = gv;                    // the gradient before BatchNorm-2's scale: the skip branch's
            s1 += gv;
            s2 = __builtin_fmaf(gv, (ysel[o] - mu) * iv, s2);
        }
    }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0) {
        const size_t row = (size_t)b * gridDim.x + blockIdx.x;
        partS[row * 2 * O + c] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
        partS[row * 2 * O + O + c] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
    }
}

// Fixed-order float64 sum over the rows of column c of part[rows][stride] for one 256-thread block
// (thread = (column tx < 64, row group ty < 4)); result valid for ty == 0 after the call.
__device__ __forceinline__ double block_col_sum(const float *__restrict__ part, int rows, int stride, int c,
                                                bool ok, double (*red)[64]) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double s = 0.0;
    if (ok) {
        int r = ty;
        for (; r + 28 < rows; r += 32) {           // 8 independent loads in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(r + 4 * u) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; r < rows; r += 4) s += (double)part[(size_t)r * stride + c];
    }
    __syncthreads();
    red[ty][tx] = s;
    __syncthreads();
    return (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// BatchNorm-2 backward constants from partS: dL/dy2 = goa [slot == ksel] + y2 * D2 + E2.
//   d2e2 = {D2[O], E2[O]} (float32), dgamma2 = S2, dbeta2 = S1.  Grid: O/64 blocks.
// With `sums` (float64 {S1[O], S2[O], global count, world}: the rows summed and all-reduced over ranks,
// SyncBatchNorm) the statistics use the global values and dgamma / dbeta are reported as global / world.
__global__ __launch_bounds__(256) void wide_consts2_kernel(const float *__restrict__ partS, int rows, int O,
                                                           const double *__restrict__ sums,
                                                           const float *__restrict__ pack2, double count,
                                                           int training, float *__restrict__ d2e2,
                                                           float *__restrict__ g_gamma2, float *__restrict__ g_beta2) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, c = blockIdx.x * 64 + tx;
    double s1, s2, gscale = 1.0;
    if (sums) {
        s1 = c < O ? sums[c] : 0.0;
        s2 = c < O ? sums[O + c] : 0.0;
        count = sums[2 * O];
        gscale = 1.0 / sums[2 * O + 1];
    } else {
        s1 = block_col_sum(partS, rows, 2 * O, c, c < O, red);
        s2 = block_col_sum(partS, rows, 2 * O, O + c, c < O, red);
    }
    if (threadIdx.x >= 64 || c >= O) return;
    const double sc = pack2[c], mu = pack2[2 * O + c], iv = pack2[3 * O + c];
    double d = 0.0, e = 0.0;
    if (training) {
        d = -sc * iv * s2 / count;
        e = -sc * s1 / count + sc * mu * iv * s2 / count;
    }
    d2e2[c] = (float)d;
    d2e2[O + c] = (float)e;
    if (g_gamma2) g_gamma2[c] = (float)(s2 * gscale);
    if (g_beta2) g_beta2[c] = (float)(s1 * gscale);
}

// BatchNorm-1 backward constants from partT = {T1 = sum g_u, T2 = sum g_u yhat1}[H]:
//   dL/dy1 = ca g_u + cb yhat1 + cc;  cabc = {ca, cb, cc}[H]; dgamma1 = T2, dbeta1 = T1.
__global__ __launch_bounds__(256) void wide_consts1_kernel(const float *__restrict__ partT, int rows, int H,
                                                           const double *__restrict__ sums,
                                                           const float *__restrict__ pack1, double count,
                                                           int training, float *__restrict__ cabc,
                                                           float *__restrict__ g_gamma1, float *__restrict__ g_beta1) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, c = blockIdx.x * 64 + tx;
    double t1, t2, gscale = 1.0;
    if (sums) {
        t1 = c < H ? sums[c] : 0.0;
        t2 = c < H ? sums[H + c] : 0.0;
        count = sums[2 * H];
        gscale = 1.0 / sums[2 * H + 1];
    } else {
        t1 = block_col_sum(partT, rows, 2 * H, c, c < H, red);
        t2 = block_col_sum(partT, rows, 2 * H, H + c, c < H, red);
    }
    if (threadIdx.x >= 64 || c >= H) return;
    const double sc = pack1[c];
    cabc[c] = (float)sc;
    cabc[H + c] = training ? (float)(-sc * t2 / count) : 0.0f;
    cabc[2 * H + c] = training ? (float)(-sc * t1 / count) : 0.0f;
    if (g_gamma1) g_gamma1[c] = (float)(t2 * gscale);
    if (g_beta1) g_beta1[c] = (float)(t1 * gscale);
}

// dL/dU per point and (minus) dL/dV per query from dL/dy1 = ca g_u + cb yhat1 + cc (the standalone form, for
// the widths whose dense products are library GEMMs; csrc/sa_wide_dense.hip: wide_point_grads fuses it):
//   G[b,n,h]  = ca sum_{rows of n} GU[row][h] + cb inv1 (occ (U - mean1) - (SP . W1p[h]) / r) + cc occ
//               (rows through the inverse map pcnt / poff / plist, ascending: a fixed order, no atomics)
//   Hq[b,q,h] = ca HA + cb HB + 32 cc                                                  (in place over HA)
// W1 (H x ldw): its first three columns are W1p.  Blocks [0, pblocks) do points, the rest queries.
__global__ __launch_bounds__(256) void wide_point_terms_kernel(long long npts, long long nqry, int H, int pblocks,
                                                               const float *__restrict__ cabc,
                                                               const float *__restrict__ pack1,
                                                               const float *__restrict__ U,
                                                               const float *__restrict__ geo,
                                                               const float *__restrict__ w1, int ldw, float inv_r,
                                                               const float *__restrict__ GU,
                                                               const int *__restrict__ pcnt,
                                                               const int *__restrict__ poff,
                                                               const int *__restrict__ plist,
                                                               float *__restrict__ G, float *__restrict__ HA,
                                                               const float *__restrict__ HB) {
    if ((int)blockIdx.x < pblocks) {
        const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
        if (e >= npts * H) return;
        const int h = (int)(e % H);
        const long long pt = e / H;
        const int cnt = pcnt[pt];
        const int *__restrict__ l = plist + poff[pt];
        float acc = 0.0f;
        for (int i = 0; i < cnt; ++i) acc += GU[(size_t)l[i] * H + h];
        const float4 ge = *reinterpret_cast<const float4 *>(geo + pt * 4);
        const float occ = ge.x;
        const float spw = __builtin_fmaf(ge.w, w1[(size_t)h * ldw + 2],
                                         __builtin_fmaf(ge.z, w1[(size_t)h * ldw + 1], ge.y * w1[(size_t)h * ldw]));
        const float yh = pack1[3 * H + h] * (occ * (U[e] - pack1[2 * H + h]) - spw * inv_r);
        G[e] = __builtin_fmaf(cabc[h], acc, __builtin_fmaf(cabc[H + h], yh, cabc[2 * H + h] * occ));
    } else {
        const long long e = (long long)(blockIdx.x - pblocks) * 256 + threadIdx.x;
        if (e >= nqry * H) return;
        const int h = (int)(e % H);
        HA[e] = __builtin_fmaf(cabc[h], HA[e], __builtin_fmaf(cabc[H + h], HB[e], 32.0f * cabc[2 * H + h]));
    }
}

}  // namespace apn

using namespace apn;

extern "C" int apn_sa_wide_image(const float *src0, int k0, int trans0, const float *src1, int kd, int nc, int ct,
                                 void *image, void *stream) {
    if (!src0 || !image || kd <= 0 || nc <= 0 || ct < 1 || ct > 4 || (kd % 32) || (nc % (32 * ct)) || k0 < 0 ||
        k0 > kd || (k0 < kd && !src1))
        return APN_EINVAL;
    const int total = (nc / 32) * (kd / 32) * 256;
    hipLaunchKernelGGL(wide_image_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src0, k0,
                       trans0, src1, kd, nc, ct, (uint4 *)image);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_bwd_prep_rows(int b, int m) { return b * ((m + 63) / 64); }

extern "C" int apn_sa_wide_bwd_prep(int b, int m, int c_out, const float *g, long long gs_b, long long gs_c,
                                    long long gs_m, const float *ysel, const float *pack2, const float *out_act,
                                    float *gpre, float *goa, float *part_s, void *stream) {
    if (b <= 0 || m <= 0 || b > 65535 || c_out <= 0 || (c_out % 64) || !g || !ysel || !pack2 || !goa || !part_s)
        return APN_EINVAL;
    hipLaunchKernelGGL(wide_bwd_prep_kernel, dim3((m + 63) / 64, c_out / 64, b), dim3(256), 0, (hipStream_t)stream,
                       m, c_out, g, gs_b, gs_c, gs_m, ysel, pack2, out_act, gpre, goa, part_s);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_consts2(const float *part_s, int rows, const double *sums, int c_out,
                                   const float *pack2, double count, int training, float *d2e2,
                                   float *g_gamma2, float *g_beta2, void *stream) {
    if ((!part_s && !sums) || rows < 0 || c_out <= 0 || !pack2 || !d2e2) return APN_EINVAL;
    hipLaunchKernelGGL(wide_consts2_kernel, dim3((c_out + 63) / 64), dim3(256), 0, (hipStream_t)stream, part_s, rows,
                       c_out, sums, pack2, count, training, d2e2, g_gamma2, g_beta2);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_consts1(const float *part_t, int rows, const double *sums, int c_mid,
                                   const float *pack1, double count, int training, float *cabc,
                                   float *g_gamma1, float *g_beta1, void *stream) {
    if ((!part_t && !sums) || rows < 0 || c_mid <= 0 || !pack1 || !cabc) return APN_EINVAL;
    hipLaunchKernelGGL(wide_consts1_kernel, dim3((c_mid + 63) / 64), dim3(256), 0, (hipStream_t)stream, part_t, rows,
                       c_mid, sums, pack1, count, training, cabc, g_gamma1, g_beta1);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_point_terms(int b, int n, int m, int c_mid, const float *cabc, const float *pack1,
                                       const float *U, const float *geo, const float *w1, int ldw, float radius,
                                       const float *GU, const int *pcnt_poff, const int *plist, float *G, float *HA,
                                       const float *HB, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c_mid <= 0 || !cabc || !pack1 || !U || !geo || !w1 || ldw < 3 || !GU ||
        !pcnt_poff || !plist || !G || !HA || !HB || !(radius > 0.0f))
        return APN_EINVAL;
    const long long npts = (long long)b * n, nqry = (long long)b * m;
    const long long pb = (npts * c_mid + 255) / 256, qb = (nqry * c_mid + 255) / 256;
    if (pb + qb > 0x7fffffffLL) return APN_EINVAL;
    hipLaunchKernelGGL(wide_point_terms_kernel, dim3((unsigned)(pb + qb)), dim3(256), 0, (hipStream_t)stream, npts,
                       nqry, c_mid, (int)pb, cabc, pack1, U, geo, w1, ldw, 1.0f / radius, GU, pcnt_poff,
                       pcnt_poff + npts, plist, G, HA, HB);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// ------------------------------------------------------------------------------------------
// Tile map: distinct-hit packing of the neighbourhoods (part of the INDEX stage: coordinates only).
//
// A ball query lists a query's cnt distinct hits in slots 0..cnt-1 and fills slots cnt..31 with copies
// of slot 0 (ball_query_gpu.cu:41-48); at PointNeXt-S stage 1 cnt averages 7.6 of 32.  Every copy of a
// position computes the same y1, a1, y2, so the fused passes only need ONE row per distinct hit plus its
// multiplicity (slot 0: 33 - cnt, the others 1): statistics, Gram and gradient sums weight a row by it,
// the max over K runs over the distinct rows.  Whole queries are packed greedily, in order, into 32-row
// MFMA tiles (a cloud's tiles are contiguous, clouds in order): 557 tiles instead of 2048 per 4 clouds.
//   rowinfo[tile][row] = qlocal | slot << 8 | mult << 16 | (row 0 only: queries in the tile) << 24;
//   mult == 0 marks a padding row (qlocal 255: no query's);  tq0[tile] = the tile's first query (global id b * M + q).
// An index row that does not have the ball-query structure (any other grouper) is kept whole:
// cnt = 32, every slot its own row.  mode 0 skips the analysis: one tile per query.
// ------------------------------------------------------------------------------------------
namespace apn {

// Rows of query q of cloud c (mode 1: the ball-query structure folded; any other row, or mode 0, kept whole).
__device__ __forceinline__ int query_rows(const int *__restrict__ row, int mode) {
    if (!mode) return 32;
    const int4 *__restrict__ r4 = reinterpret_cast<const int4 *>(row);
    unsigned differ = 0;
    int first = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int4 v = r4[k];
        if (k == 0) first = v.x;
        differ |= (unsigned)(v.x != first) << (4 * k) | (unsigned)(v.y != first) << (4 * k + 1) |
                  (unsigned)(v.z != first) << (4 * k + 2) | (unsigned)(v.w != first) << (4 * k + 3);
    }
    const int c = __popc(differ) + 1;
    // ball-query structure: the differing slots are exactly 1..c-1 (and, being hits in index order, distinct)
    return differ == ((c >= 32 ? 0xffffffffu : ((1u << c) - 1u)) & ~1u) ? c : 32;
}

// One block per cloud: next-fit packing of its M queries into 32-row tiles WITHOUT a serial pass over the
// queries.  With P the prefix sums of the row counts, the tile that starts at query s ends before
// next(s) = the first j with P[j+1] - P[s] > 32 (a binary search per query, all in parallel); the tile starts
// are the orbit 0, next(0), next(next(0)), ...: the t-th start is reached through the binary digits of t over the
// doubled maps next^(2^k) (log M rounds), every tile independently.
//   qmeta[q] = tile_local | row0 << 16 | starts_tile << 24 | number in its tile << 25;  cnt8[q] = its row count
//   tcount[c] = tiles of the cloud;  tnq_local[c m + t] = queries in tile t.   M <= 2048.
// (blockIdx.y = which of several independent maps: the batches of a stacked index stage; their neighbour arrays lie
// idx_stride ints apart, their blobs blob_bytes apart)
__global__ __launch_bounds__(256) void tilemap_pack_kernel(int m, int mode, int levels, const int *__restrict__ idx,
                                                           unsigned char *__restrict__ cnt8,
                                                           unsigned *__restrict__ qmeta, int *__restrict__ tcount,
                                                           unsigned short *__restrict__ tnq_local, long long idx_stride,
                                                           long long blob_bytes) {
    {
        const long long z = blockIdx.y, off = z * blob_bytes;
        idx += z * idx_stride;
        cnt8 += off;
        qmeta = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(qmeta) + off);
        tcount = reinterpret_cast<int *>(reinterpret_cast<char *>(tcount) + off);
        tnq_local = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(tnq_local) + off);
    }
    extern __shared__ int sh[];
    int *P = sh;                              // [m + 1]  exclusive prefix of the counts
    int *start = P + (m + 1);                 // [m + 1]  tile starts
    int *jump = start + (m + 1);              // [levels][m + 1]
    __shared__ int wsum[256];
    __shared__ int ntile_s;
    const int t = threadIdx.x, c = blockIdx.x;
    const int per = (m + 255) / 256;          // consecutive queries per thread (<= 8)
    int cn[8], loc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = t * per + i;
        cn[i] = (i < per && q < m) ? query_rows(idx + ((size_t)c * m + q) * 32, mode) : 0;
        loc += cn[i];
    }
    wsum[t] = loc;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = t >= d ? wsum[t - d] : 0;
        __syncthreads();
        wsum[t] += v;
        __syncthreads();
    }
    int run = wsum[t] - loc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = t * per + i;
        if (i < per && q < m) {
            P[q] = run;
            cnt8[(size_t)c * m + q] = (unsigned char)cn[i];
            run += cn[i];
        }
    }
    if (t == 255) P[m] = wsum[255];
    if (t == 0) ntile_s = 0;
    __syncthreads();
    // next(q): first j in (q, m] with P[j + 1] > P[q] + 32, i.e. the first query that no longer fits; next(m) = m
    for (int q = t; q <= m; q += 256) {
        int nx = m;
        if (q < m) {
            const int lim = P[q] + 32;
            int lo = q + 1, hi = m;           // answer in [lo, hi]: smallest j >= q + 1 with P[j + 1] > lim, or m
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (P[mid + 1] > lim) hi = mid; else lo = mid + 1;
            }
            nx = lo;
        }
        jump[q] = nx;
    }
    __syncthreads();
    for (int k = 1; k < levels; ++k) {
        int *prev = jump + (k - 1) * (m + 1), *cur = jump + k * (m + 1);
        for (int q = t; q <= m; q += 256) cur[q] = prev[prev[q]];
        __syncthreads();
    }
    // tile t starts at next^t(0)
    for (int tl = t; tl <= m; tl += 256) {
        int s0 = 0;
        for (int k = 0; k < levels; ++k)
            if ((tl >> k) & 1) s0 = jump[k * (m + 1) + s0];
        start[tl] = tl < m ? s0 : m;
    }
    __syncthreads();
    for (int tl = t; tl < m; tl += 256) {
        if (start[tl] < m) {
            const int nxt = start[tl + 1];
            tnq_local[(size_t)c * m + tl] = (unsigned short)(nxt - start[tl]);
            if (nxt >= m) ntile_s = tl + 1;                       // the last tile: exactly one thread
        }
    }
    __syncthreads();
    const int T = ntile_s;
    if (t == 0) tcount[c] = T;
    for (int q = t; q < m; q += 256) {
        int lo = 0, hi = T - 1;               // largest tile with start <= q
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (start[mid] <= q) lo = mid; else hi = mid - 1;
        }
        const int s0 = start[lo];
        qmeta[(size_t)c * m + q] = (unsigned)lo | ((unsigned)(P[q] - P[s0]) << 16) | (q == s0 ? 1u << 24 : 0u) |
                                   ((unsigned)(q - s0) << 25);
    }
}

// The serial form of the same packing, for clouds of more than 2048 queries: one WAVE per cloud, 64 queries per
// round, the sequential part on the scalar unit over v_readlane.
__global__ __launch_bounds__(256) void tilemap_pack_serial_kernel(int b, int m, int mode, const int *__restrict__ idx,
                                                                  unsigned char *__restrict__ cnt8,
                                                                  unsigned *__restrict__ qmeta,
                                                                  int *__restrict__ tcount,
                                                                  unsigned short *__restrict__ tnq_local) {
    const int lane = lane_id();
    const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (c >= b) return;
    int tile = 0, fill = 0, nq = 0;                       // wave-uniform
    for (int base = 0; base < m; base += 64) {
        const int mine = base + lane < m ? query_rows(idx + ((size_t)c * m + base + lane) * 32, mode) : 0;
        if (base + lane < m) cnt8[(size_t)c * m + base + lane] = (unsigned char)mine;
        const int lim = m - base < 64 ? m - base : 64;
        unsigned meta = 0;
        for (int i = 0; i < lim; ++i) {
            const int cn = __builtin_amdgcn_readlane(mine, i);
            bool starts = base + i == 0;
            if (fill + cn > 32) {
                if (lane == 0) tnq_local[(size_t)c * m + tile] = (unsigned short)nq;
                ++tile; fill = 0; nq = 0; starts = true;
            }
            const unsigned v = (unsigned)tile | ((unsigned)fill << 16) | (starts ? 1u << 24 : 0u) | ((unsigned)nq << 25);
            meta = lane == i ? v : meta;
            fill += cn;
            ++nq;
        }
        if (lane < lim) qmeta[(size_t)c * m + base + lane] = meta;
    }
    if (lane == 0) {
        tnq_local[(size_t)c * m + tile] = (unsigned short)nq;
        tcount[c] = tile + 1;
    }
}

// Block = (cloud, 64 of its queries), a wave per 16 queries: write the rows (and the padding behind a tile's last
// query).  The cloud's first tile = the tile counts of the clouds before it, summed here (toff[c] kept for the
// inverse map); the last cloud's first block also leaves the number of tiles in use.
__global__ __launch_bounds__(256) void tilemap_fill_kernel(int b, int m, const unsigned char *__restrict__ cnt8,
                                                           const unsigned *__restrict__ qmeta,
                                                           const int *__restrict__ tcount, int *__restrict__ toff,
                                                           const unsigned short *__restrict__ tnq_local,
                                                           int *__restrict__ ntiles, int *__restrict__ tq0,
                                                           unsigned *__restrict__ rowinfo,
                                                           const int *__restrict__ idx, int *__restrict__ rownn,
                                                           long long idx_stride, long long blob_bytes) {
    {
        const long long z = blockIdx.z, off = z * blob_bytes;
        idx += z * idx_stride;
        cnt8 += off;
        qmeta = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(qmeta) + off);
        tcount = reinterpret_cast<const int *>(reinterpret_cast<const char *>(tcount) + off);
        toff = reinterpret_cast<int *>(reinterpret_cast<char *>(toff) + off);
        tnq_local = reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(tnq_local) + off);
        ntiles = reinterpret_cast<int *>(reinterpret_cast<char *>(ntiles) + off);
        tq0 = reinterpret_cast<int *>(reinterpret_cast<char *>(tq0) + off);
        rowinfo = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(rowinfo) + off);
        rownn = reinterpret_cast<int *>(reinterpret_cast<char *>(rownn) + off);
    }
    __shared__ int red[256];
    const int c = blockIdx.y, t = threadIdx.x;
    int s = 0;
    for (int j = t; j < c; j += 256) s += tcount[j];
    red[t] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (t < d) red[t] += red[t + d];
        __syncthreads();
    }
    const int base_tile = red[0];
    if (blockIdx.x == 0 && t == 0) {
        toff[c] = base_tile;
        if (c == b - 1) ntiles[0] = base_tile + tcount[c];
    }
    // A wave writes the rows of sixteen queries, each half of it (32 lanes = the 32 slots) eight of them.  Round 5: every load
    // the sixteen queries need -- count, packing record, the next query's record, the 32 neighbours -- is requested up front,
    // unconditionally on clamped indices (one round trip; the tile's query count of a starting query: a second one).  The
    // loop that stood here paid sixteen dependent round trips per wave with half the lanes returned: 48 of the stacked
    // launch's 55 us.
    const int lane = t & 63, w = t >> 6, l32 = lane & 31, half = lane >> 5;
    int cnv[8], nbv[8];
    unsigned metav[8], nextv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ql = blockIdx.x * 64 + w * 16 + 2 * j + half;
        const size_t q = (size_t)c * m + (ql < m ? ql : m - 1);
        cnv[j] = cnt8[q];
        metav[j] = qmeta[q];
        nextv[j] = qmeta[ql + 1 < m ? q + 1 : q];
        nbv[j] = idx[q * 32 + l32];
    }
    unsigned tnqv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tnqv[j] = tnq_local[(size_t)c * m + (metav[j] & 0xffff)];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ql = blockIdx.x * 64 + w * 16 + 2 * j + half;
        const int nb0 = __shfl(nbv[j], half * 32);                       // slot 0's neighbour (for the padding rows)
        if (ql >= m) continue;
        const size_t q = (size_t)c * m + ql;
        const int cn = cnv[j];
        const unsigned meta = metav[j];
        const int tl = meta & 0xffff, row0 = (meta >> 16) & 0xff;
        const bool starts = (meta >> 24) & 1;
        const bool ends = ql == m - 1 || ((nextv[j] >> 24) & 1);          // the next query opens a tile
        const size_t tile = (size_t)base_tile + tl;
        const unsigned qlocal = (meta >> 25) & 0x3fu;
        if (l32 < cn) {
            const unsigned mult = l32 == 0 ? (unsigned)(33 - cn) : 1u;
            unsigned v = qlocal | ((unsigned)l32 << 8) | (mult << 16);
            if (starts && l32 == 0) v |= tnqv[j] << 24;
            rowinfo[tile * 32 + row0 + l32] = v;
            rownn[tile * 32 + row0 + l32] = nbv[j];
        }
        if (ends && row0 + cn + l32 < 32) {                  // padding: multiplicity 0, a valid neighbour to load
            rowinfo[tile * 32 + row0 + cn + l32] = 0xffu;                // query 255: belongs to none
            rownn[tile * 32 + row0 + cn + l32] = nb0;
        }
        if (starts && l32 == 0) tq0[tile] = (int)q;
    }
}

// ------------------------------------------------------------------------------------------
// Inverse map (index stage too): for every support point the rows that gather it, so that the backward
// pass sums dL/dU per point in a FIXED order with plain loads instead of float atomics:
//   pcnt[b n]  rows that gather point n;  poff[b n]  where its list starts in plist;
//   plist      row ids (tile * 32 + row), ascending within a point's list;
//   geo[b n]   {sum of mult, sum of mult * new_xyz[query]} over the list (what BatchNorm-1's mean term needs).
// Counting sort: count (integer atomics: the counts do not depend on the order), scan per cloud, fill
// (order left to the atomics), then every list is sorted -- the result is a pure function of idx.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ const int *tm_toff(const int *tmap, int nq, int b) {
    return tmap + 4 + ((nq + 3) & ~3) + (size_t)64 * nq + b;
}

// pcnt = 0 and (optional) fq = -1, one launch (a kernel, not hipMemsetAsync: the whole index stage must
// replay from a captured hipGraph, and a captured memset node aborted the replay here)
__global__ __launch_bounds__(256) void csr_init_kernel(long long npts, int *__restrict__ pcnt, int *__restrict__ fq) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= npts) return;
    pcnt[e] = 0;
    if (fq) fq[e] = -1;
}

// one thread per row of the tile map.  fill = 0: count the rows of every point (and, with fidx, record which query
// each sampled point is: fq[b n] -- FPS picks are distinct); fill = 1: write the row ids behind the cursors.
__global__ __launch_bounds__(256) void csr_count_fill_kernel(int nq, int n, int m, int fill,
                                                             const int *__restrict__ idx,
                                                             const int *__restrict__ tmap, int *__restrict__ pcnt,
                                                             int *__restrict__ poff, int *__restrict__ plist,
                                                             const int *__restrict__ fidx, int *__restrict__ fq) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;       // row id
    const int tile = (int)(e >> 5);
    if (tile >= tmap[0]) return;
    const unsigned info = reinterpret_cast<const unsigned *>(tmap + 4 + ((nq + 3) & ~3))[e];
    if (((info >> 16) & 0xffu) == 0) return;
    const int q = tmap[4 + tile] + (int)(info & 0xffu);
    const size_t cbase = (size_t)(q / m) * n;
    const size_t gn = cbase + tmap[4 + ((nq + 3) & ~3) + (size_t)32 * nq + e];           // rownn[e]
    if (!fill) {
        atomicAdd(pcnt + gn, 1);
        if (fq && ((info >> 8) & 0xffu) == 0) fq[cbase + fidx[q]] = q % m;                 // one row per query has slot 0
    } else {
        plist[atomicAdd(poff + gn, 1)] = (int)e;
    }
}

// one block per cloud: poff = (first row id of the cloud) + exclusive scan of pcnt
__global__ __launch_bounds__(1024) void csr_scan_kernel(int nq, int b, int n, const int *__restrict__ tmap,
                                                        const int *__restrict__ pcnt, int *__restrict__ poff) {
    __shared__ int part[1024];
    const int t = threadIdx.x, cloud = blockIdx.x, per = (n + 1023) / 1024;
    const int *__restrict__ cnt = pcnt + (size_t)cloud * n;
    int s = 0;
    for (int i = 0; i < per; ++i) {
        const int k = t * per + i;
        if (k < n) s += cnt[k];
    }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = tm_toff(tmap, nq, b)[cloud] * 32 + part[t] - s;
    for (int i = 0; i < per; ++i) {
        const int k = t * per + i;
        if (k < n) { poff[(size_t)cloud * n + k] = run; run += cnt[k]; }
    }
}

// Sort every point's list (after the fill poff is the list's END; the entries arrive in the order the fill's atomics
// happened to run), restore poff to the list's start, sum geo over the sorted list.  Two kernels share the points:
//   csr_sort_short  one THREAD per point, lists of up to CSR_SHORT entries: insertion sort in an LDS slice;
//   csr_sort_long   one WAVE per point, longer lists (dense neighbourhoods: a point gathered by 30-1000 rows):
//                   bitonic network over an LDS buffer of up to CSR_LONG entries (beyond that -- a point in more than
//                   CSR_LONG neighbourhoods -- lane 0 falls back to Shell's sort in place).
// Both leave the same ascending list; geo's sum order is fixed per kernel (so per list length): reproducible.
constexpr int CSR_SHORT = 16, CSR_LONG = 2048;

__device__ __forceinline__ void geo_add(const int *__restrict__ tmap, const unsigned *__restrict__ rows,
                                        const float *__restrict__ new_xyz, int e, float &occ, float &sx, float &sy,
                                        float &sz) {
    const unsigned info = rows[e];
    const float mult = (float)((info >> 16) & 0xffu);
    const float *__restrict__ q = new_xyz + (size_t)(tmap[4 + (e >> 5)] + (int)(info & 0xffu)) * 3;
    occ += mult;
    sx = __builtin_fmaf(mult, q[0], sx);
    sy = __builtin_fmaf(mult, q[1], sy);
    sz = __builtin_fmaf(mult, q[2], sz);
}

__global__ __launch_bounds__(128) void csr_sort_short_kernel(int nq, long long npts, const int *__restrict__ tmap,
                                                             const int *__restrict__ pcnt, int *__restrict__ poff,
                                                             int *__restrict__ plist,
                                                             const float *__restrict__ new_xyz,
                                                             float *__restrict__ geo) {
    __shared__ int slice[CSR_SHORT][129];                   // [entry][thread]: conflict-free per-thread arrays
    const long long gn = (long long)blockIdx.x * 128 + threadIdx.x;
    if (gn >= npts) return;
    const int c = pcnt[gn];
    if (c > CSR_SHORT) return;                              // csr_sort_long's
    const int start = poff[gn] - c;
    poff[gn] = start;
    int *__restrict__ l = plist + start;
    const unsigned *__restrict__ rows = reinterpret_cast<const unsigned *>(tmap + 4 + ((nq + 3) & ~3));
    const int t = threadIdx.x;
    for (int i = 0; i < c; ++i) {
        const int v = l[i];
        int j = i;
        for (; j > 0 && slice[j - 1][t] > v; --j) slice[j][t] = slice[j - 1][t];
        slice[j][t] = v;
    }
    float occ = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (int i = 0; i < c; ++i) {
        const int e = slice[i][t];
        l[i] = e;
        if (geo) geo_add(tmap, rows, new_xyz, e, occ, sx, sy, sz);
    }
    if (geo) *reinterpret_cast<float4 *>(geo + gn * 4) = make_float4(occ, sx, sy, sz);
}

__global__ __launch_bounds__(256) void csr_sort_long_kernel(int nq, long long npts, const int *__restrict__ tmap,
                                                            const int *__restrict__ pcnt, int *__restrict__ poff,
                                                            int *__restrict__ plist,
                                                            const float *__restrict__ new_xyz,
                                                            float *__restrict__ geo) {
    __shared__ int buf[4][CSR_LONG];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const long long gn = (long long)blockIdx.x * 4 + w;
    if (gn >= npts) return;
    const int c = pcnt[gn];
    if (c <= CSR_SHORT) return;                             // csr_sort_short's
    const int start = poff[gn] - c;
    int *__restrict__ l = plist + start;
    const unsigned *__restrict__ rows = reinterpret_cast<const unsigned *>(tmap + 4 + ((nq + 3) & ~3));
    float occ = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
    if (c <= CSR_LONG) {
        int n = 32;
        while (n < c) n <<= 1;
        int *b = buf[w];
        // (one wave owns b: its LDS operations execute in program order, no barrier between the stages)
        for (int i = lane; i < n; i += 64) b[i] = i < c ? l[i] : 0x7fffffff;
        for (int k = 2; k <= n; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < n / 2; t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // the lower index of pair t at distance j
                    const int p = i | j;
                    const bool up = (i & k) == 0;
                    const int x = b[i], y = b[p];
                    if ((x > y) == up) { b[i] = y; b[p] = x; }
                }
            }
        }
        for (int i = lane; i < c; i += 64) {
            const int e = b[i];
            l[i] = e;
            if (geo) geo_add(tmap, rows, new_xyz, e, occ, sx, sy, sz);
        }
    } else if (lane == 0) {
        for (int gap = 1093; gap > 0; gap = gap == 1 ? 0 : (gap - 1) / 3) {
            for (int i = gap; i < c; ++i) {                     // Shell sort, gaps 1093, 364, 121, 40, 13, 4, 1
                const int v = l[i];
                int j = i;
                for (; j >= gap && l[j - gap] > v; j -= gap) l[j] = l[j - gap];
                l[j] = v;
            }
        }
        if (geo)
            for (int i = 0; i < c; ++i) geo_add(tmap, rows, new_xyz, l[i], occ, sx, sy, sz);
    }
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        occ += __shfl_xor(occ, k); sx += __shfl_xor(sx, k); sy += __shfl_xor(sy, k); sz += __shfl_xor(sz, k);
    }
    if (lane == 0) {
        poff[gn] = start;
        if (geo) *reinterpret_cast<float4 *>(geo + gn * 4) = make_float4(occ, sx, sy, sz);
    }
}


// The same inverse map by ONE launch, one workgroup per cloud, everything in LDS (round 4: the six launches above cost
// 46-72 us per stage on the classifier's critical path for a few thousand rows per cloud).  A list entry travels as
//   (row within the cloud) << 16 | mult << 10 | (query within the cloud)
// so that sorting the words sorts the rows and geo needs nothing but the cloud's query coordinates (LDS too).
// Counts and cursors by LDS integer atomics (order-free); lists of up to CSR_INS entries sorted by their point's thread
// (insertion, in place), lists of up to 64 by a wave with one entry per lane (bitonic network over lane exchanges: 21
// stages), longer ones by a wave through ranks (the entries of a list are distinct).  A thread's insertion sort is a chain
// of dependent LDS operations: at 64 entries per list it took ~100 us on collapsed clouds.  plist / pcnt / poff are the
// pure function of idx the kernels above compute; geo is summed in ascending row order (lists up to CSR_INS) or
// lane-strided + butterfly (longer): fixed orders.  For m <= 1024 (query bits), rows <= 65536 and what fits in LDS.
constexpr int CSR_INS = 16;      // lists up to here: insertion by the point's thread; up to 64: a wave's bitonic network in registers
// Round 5: optional `rowdst` = the INVERSE of plist (rowdst[row] = the row's place in the point-sorted order): the
// register-resident backward pass stores a row's g_u at that place, so that a point's rows are CONTIGUOUS and its consumer
// sums them in a fixed order without a list (csrc/sa_fused.hip, sa_glue.hip).  blockIdx.y = batch z of a stacked index
// stage: map z at tmap + z * tmap_stride ints, every other array at its own per-batch stride (plist may be null then).
__global__ __launch_bounds__(1024) void csr_cloud_kernel(int nq, int b, int n, int m, const int *__restrict__ tmap,
                                                         const float *__restrict__ new_xyz, int *__restrict__ pcnt,
                                                         int *__restrict__ poff, int *__restrict__ plist,
                                                         float *__restrict__ geo, const int *__restrict__ fidx,
                                                         int *__restrict__ fq, int rows_cap, int bitonic_words,
                                                         int *__restrict__ rowdst, long long tmap_stride,
                                                         int *__restrict__ dup) {
    extern __shared__ int csm[];
    {
        const long long z = blockIdx.y, npts = (long long)b * n;
        tmap += z * tmap_stride;
        pcnt += z * (2 * npts + (dup ? b : 0));
        poff += z * (2 * npts + (dup ? b : 0));
        if (plist) plist += z * 32 * nq;
        if (rowdst) rowdst += z * 32 * nq;
        if (new_xyz) new_xyz += z * 3 * nq;
        if (geo) geo += z * 4 * npts;
        if (fidx) fidx += z * nq;
        if (fq) fq += z * npts;
        if (dup) dup += z * (2 * npts + b);            // (the row map's blob of a batch: pcnt | poff | dup, apn_sa_rowmap_ints)
    }
    __shared__ int part[1024];
    __shared__ int nlong, sdup;
    int *scnt = csm, *soff = csm + n, *slist = csm + 2 * n;
    float *sq = reinterpret_cast<float *>(slist + rows_cap);            // [m][3]
    int *slong = reinterpret_cast<int *>(sq + 3 * m);                    // points with long lists (at most rows_cap / CSR_INS)
    int *sbit = slong + rows_cap / CSR_INS + 1;                          // [16 waves][bitonic_words]: a long list, padded to a power of two
    const int cloud = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int *__restrict__ tq0 = tmap + 4;
    const unsigned *__restrict__ rows = reinterpret_cast<const unsigned *>(tmap + 4 + ((nq + 3) & ~3));
    const int *__restrict__ rownn = reinterpret_cast<const int *>(rows + (size_t)32 * nq);
    const int *__restrict__ tcount = rownn + (size_t)32 * nq, *__restrict__ toff = tcount + b;
    const int row0 = toff[cloud] * 32, nrows = tcount[cloud] * 32;
    const size_t cbase = (size_t)cloud * n;
    const int qbase = cloud * m;
    for (int k = t; k < n; k += 1024) { scnt[k] = 0; if (fq) fq[cbase + k] = -1; }
    if (new_xyz)
        for (int e = t; e < 3 * m; e += 1024) sq[e] = new_xyz[(size_t)qbase * 3 + e];
    if (t == 0) { nlong = 0; sdup = 0; }
    // dup[cloud] = 0 iff the cloud's m picks (fidx) are m DIFFERENT points -- what lets the skip branch's gradient rows be
    // plain stores instead of float atomics (csrc/sa_glue.hip, bwd_prep_kernel); 1: a point was picked twice (a collapsed
    // cloud), or the picks are not known here.  A bit per point in `part` (free until the scan below).
    const bool marks = dup && fidx && n <= 32 * 1024;
    if (marks) part[t] = 0;
    __syncthreads();
    if (marks) {
        bool twice = false;
        for (int q = t; q < m; q += 1024) {
            const int f = fidx[qbase + q];
            const unsigned bit = 1u << (f & 31);
            twice |= (atomicOr(reinterpret_cast<unsigned *>(&part[f >> 5]), bit) & bit) != 0u;
        }
        if (twice) sdup = 1;
    }
    // count (and which query each sampled point is: one row per query has slot 0).  Round 5: a thread's rows are requested
    // in batches of eight -- record, neighbour and first query of the tile, clamped, unconditional -- and, when the cloud has
    // at most 8192 rows (always at stage 1), KEPT for the fill pass: a loop of one dependent round trip per row (4-5 per thread
    // and pass at stage 1) was most of this kernel's 16 us
    constexpr int RB = 8;
    unsigned k_info[RB];
    int k_nn[RB], k_q0[RB];
    const bool keep = nrows <= RB * 1024;                   // (workgroup-uniform)
    auto load_rows = [&](int base) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int e = base + t + 1024 * u;
            const int ec = row0 + (e < nrows ? e : (nrows > 0 ? nrows - 1 : 0));
            k_info[u] = rows[ec];
            k_nn[u] = rownn[ec];
            k_q0[u] = tq0[ec >> 5];
            if (e >= nrows) k_info[u] = 0u;                 // (multiplicity 0: not a row)
        }
    };
    for (int base = 0; base < nrows; base += RB * 1024) {
        load_rows(base);
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const unsigned info = k_info[u];
            if (((info >> 16) & 0xffu) == 0) continue;
            atomicAdd(&scnt[k_nn[u]], 1);
            if (fq && ((info >> 8) & 0xffu) == 0) {
                const int q = k_q0[u] + (int)(info & 0xffu);
                fq[cbase + fidx[q]] = q - qbase;
            }
        }
    }
    __syncthreads();
    if (dup && t == 0) dup[2 * (long long)b * n + cloud] = marks ? sdup : 1;
    // exclusive scan of the counts: soff = where a point's list starts (within the cloud)
    const int per = (n + 1023) / 1024;
    int sum = 0;
    for (int i = 0; i < per; ++i) {
        const int k = t * per + i;
        if (k < n) sum += scnt[k];
    }
    {
        // inclusive scan over the 1024 threads: inside a wave by lane exchanges, the sixteen wave totals by wave 0 -- two
        // barriers (round 5; the Hillis-Steele form over LDS took twenty, ~5 of the kernel's 16 us)
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        __shared__ int wtot[16];
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            int w = lane < 16 ? wtot[lane] : 0;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int v = __shfl_up(w, d);
                if (lane >= d) w += v;
            }
            if (lane < 16) wtot[lane] = w;
        }
        __syncthreads();
        part[t] = incl + (wave > 0 ? wtot[wave - 1] : 0);
    }
    __syncthreads();                                    // (part[1023], the cloud's total, is read by every thread at the end)
    {
        int run = part[t] - sum;
        for (int i = 0; i < per; ++i) {
            const int k = t * per + i;
            if (k < n) {
                const int c = scnt[k];
                soff[k] = run;
                pcnt[cbase + k] = c;
                poff[cbase + k] = row0 + run;
                if (c > CSR_INS) slong[atomicAdd(&nlong, 1)] = k;
                scnt[k] = run;                                  // from here on: the list's cursor
                run += c;
            }
        }
    }
    __syncthreads();
    // fill (the order inside a list is the atomics': sorted next)
    for (int base = 0; base < nrows; base += RB * 1024) {
        if (!keep) load_rows(base);
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const unsigned info = k_info[u];
            const unsigned mult = (info >> 16) & 0xffu;
            if (mult == 0) continue;
            const int e = base + t + 1024 * u;
            const int q = k_q0[u] + (int)(info & 0xffu) - qbase;
            slist[atomicAdd(&scnt[k_nn[u]], 1)] = (int)(((unsigned)e << 16) | (mult << 10) | (unsigned)q);
        }
    }
    __syncthreads();
    auto geo_term = [&](int w, float &occ, float &sx, float &sy, float &sz) {
        const float mult = (float)((w >> 10) & 63);
        const float *__restrict__ q = sq + 3 * (w & 1023);
        occ += mult;
        sx = __builtin_fmaf(mult, q[0], sx);
        sy = __builtin_fmaf(mult, q[1], sy);
        sz = __builtin_fmaf(mult, q[2], sz);
    };
    // short lists: the point's thread
    for (int k = t; k < n; k += 1024) {
        const int start = soff[k], c = scnt[k] - start;
        if (c > CSR_INS) continue;
        int *l = slist + start;
        for (int i = 1; i < c; ++i) {
            const int v = l[i];
            int j = i;
            for (; j > 0 && l[j - 1] > v; --j) l[j] = l[j - 1];
            l[j] = v;
        }
        float occ = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (int i = 0; i < c; ++i) geo_term(l[i], occ, sx, sy, sz);
        if (geo) *reinterpret_cast<float4 *>(geo + (cbase + k) * 4) = make_float4(occ, sx, sy, sz);
    }
    // longer lists: a wave each
    for (int li = wave; li < nlong; li += 16) {
        const int k = slong[li], start = soff[k], c = scnt[k] - start;
        int *l = slist + start;
        if (c <= 64) {                                          // (wave-uniform) one entry per lane, bitonic over the lanes
            int v = lane < c ? l[lane] : 0x7fffffff;
#pragma unroll
            for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    const int o = __shfl_xor(v, j);
                    const bool up = (lane & kk) == 0, low = (lane & j) == 0;
                    v = (low == up) ? (v < o ? v : o) : (v > o ? v : o);
                }
            }
            if (lane < c) l[lane] = v;
            float occ = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
            if (lane < c) geo_term(v, occ, sx, sy, sz);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                occ += __shfl_xor(occ, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
            }
            if (lane == 0 && geo) *reinterpret_cast<float4 *>(geo + (cbase + k) * 4) = make_float4(occ, sx, sy, sz);
            continue;
        }
        if (c <= bitonic_words) {
            // a bitonic network over this wave's LDS buffer (a thread's insertion sort is a chain of dependent LDS
            // operations, the rank form below is quadratic: 17 us for a list of 256 -- collapsed clouds have 32 such
            // lists per cloud)  (one wave owns the buffer: its LDS operations execute in program order)
            int nn2 = 128;
            while (nn2 < c) nn2 <<= 1;
            int *bb = sbit + wave * bitonic_words;
            for (int i = lane; i < nn2; i += 64) bb[i] = i < c ? l[i] : 0x7fffffff;
            for (int kk = 2; kk <= nn2; kk <<= 1) {
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int t2 = lane; t2 < nn2 / 2; t2 += 64) {
                        const int i = ((t2 & ~(j - 1)) << 1) | (t2 & (j - 1));       // the lower index of pair t2 at distance j
                        const int pp = i | j;
                        const bool up = (i & kk) == 0;
                        const int x = bb[i], y = bb[pp];
                        if ((x > y) == up) { bb[i] = y; bb[pp] = x; }
                    }
                }
            }
            float occ = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
            for (int i = lane; i < c; i += 64) {
                const int e2 = bb[i];
                l[i] = e2;
                geo_term(e2, occ, sx, sy, sz);
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                occ += __shfl_xor(occ, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
            }
            if (lane == 0 && geo) *reinterpret_cast<float4 *>(geo + (cbase + k) * 4) = make_float4(occ, sx, sy, sz);
            continue;
        }
        // ranks first, all of them, then the moves (one wave's LDS operations execute in order)
        int mine[16], rank[16];
        const int cc = c < 1024 ? c : 1024;                     // (c <= m <= 1024: a point is in a query's rows once)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = lane + 64 * u;
            mine[u] = i < cc ? l[i] : 0x7fffffff;
            rank[u] = 0;
        }
        for (int j = 0; j < cc; ++j) {
            const int v = l[j];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (64 * u < cc) rank[u] += v < mine[u] ? 1 : 0;         // (wave-uniform)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lane + 64 * u < cc) l[rank[u]] = mine[u];
        float occ = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (int i = lane; i < cc; i += 64) geo_term(l[i], occ, sx, sy, sz);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            occ += __shfl_xor(occ, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
        }
        if (lane == 0 && geo) *reinterpret_cast<float4 *>(geo + (cbase + k) * 4) = make_float4(occ, sx, sy, sz);
    }
    __syncthreads();
    const int total = part[1023];
    for (int i = t; i < total; i += 1024) {
        const int row = row0 + (int)((unsigned)slist[i] >> 16);
        if (plist) plist[row0 + i] = row;
        if (rowdst) rowdst[row] = row0 + i;
    }
}

// rowdst from plist (the multi-launch builder's path): one thread per point walks its sorted list
__global__ __launch_bounds__(256) void rowdst_kernel(long long npts, const int *__restrict__ pcnt, const int *__restrict__ poff,
                                                     const int *__restrict__ plist, int *__restrict__ rowdst) {
    const long long gn = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gn >= npts) return;
    const int c = pcnt[gn], s0 = poff[gn];
    for (int i = 0; i < c; ++i) rowdst[plist[s0 + i]] = s0 + i;
}

}  // namespace apn

// The map is ONE int32 blob (BM = b * m, BM4 = BM rounded up to 4):
//   [0] tiles in use   [4, 4 + BM) tq0   [4 + BM4, + 32 BM) rowinfo   [.., + 32 BM) rownn = each row's neighbour
//   (idx[query][slot], so that the passes need no second dependent load)   then scratch of this builder
// (per-cloud tile counts and offsets, per-query packing records); apn_sa_wide_tilemap_ints(b, m) = its size.
static size_t tilemap_rows_off(int bm) { return 4 + (size_t)((bm + 3) & ~3); }

extern "C" int apn_sa_wide_tilemap_ints(int b, int m) {
    if (b <= 0 || m <= 0 || (long long)b * m > 0x7fffffffLL / 64) return 0;
    const size_t bm = (size_t)b * m;
    // (a multiple of four ints: maps stacked back to back by apn_sa_wide_tilemap_many stay 16-byte aligned)
    return (int)((tilemap_rows_off((int)bm) + 64 * bm + 2 * (size_t)b + bm + (bm + 1) / 2 + (bm + 3) / 4 + 8 + 3) & ~(size_t)3);
}

// count maps in one pair of launches: map z reads idx + z * b * m * 32 and fills tmap + z * tilemap_ints(b, m)
// (the index stage of a whole hipGraph replay: twenty batches' maps were forty launches of 12-15 us on the index stream)
extern "C" int apn_sa_wide_tilemap_many(int count, int b, int m, int mode, const int *idx, int *tmap, void *stream) {
    if (count <= 0 || count > 65535 || b <= 0 || m <= 0 || m > 65535 || b > 65535 ||
        (long long)b * m > 0x7fffffffLL / 64 || !idx || !tmap)
        return APN_EINVAL;
    const int nq = b * m;
    const long long idx_stride = (long long)nq * 32, blob_bytes = (long long)apn_sa_wide_tilemap_ints(b, m) * 4;
    int *tq0 = tmap + 4;
    unsigned *rowinfo = (unsigned *)(tmap + tilemap_rows_off(nq));
    int *rownn = (int *)(rowinfo + (size_t)32 * nq);
    int *tcount = rownn + (size_t)32 * nq, *toff = tcount + b;
    unsigned *qmeta = (unsigned *)(toff + b);
    unsigned short *tnq_local = (unsigned short *)(qmeta + nq);
    unsigned char *cnt8 = (unsigned char *)(tnq_local + 2 * (((size_t)nq + 1) / 2));
    hipStream_t st = (hipStream_t)stream;
    if (m <= 2048) {
        int levels = 1;
        while ((1 << levels) <= m) ++levels;                     // tile numbers < m need `levels` binary digits
        const size_t lds = (size_t)(levels + 2) * (m + 1) * sizeof(int);
        if (lds > 48 * 1024) {
            if (hipError_t e = hipFuncSetAttribute((const void *)apn::tilemap_pack_kernel,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
                return (int)e;
        }
        hipLaunchKernelGGL(apn::tilemap_pack_kernel, dim3(b, count), dim3(256), lds, st, m, mode, levels, idx, cnt8, qmeta,
                           tcount, tnq_local, idx_stride, blob_bytes);
    } else {
        for (int z = 0; z < count; ++z)
            hipLaunchKernelGGL(apn::tilemap_pack_serial_kernel, dim3((b + 3) / 4), dim3(256), 0, st, b, m, mode,
                               idx + z * idx_stride, cnt8 + z * blob_bytes,
                               (unsigned *)((char *)qmeta + z * blob_bytes), (int *)((char *)tcount + z * blob_bytes),
                               (unsigned short *)((char *)tnq_local + z * blob_bytes));
    }
    hipLaunchKernelGGL(apn::tilemap_fill_kernel, dim3((m + 63) / 64, b, count), dim3(256), 0, st, b, m, cnt8, qmeta, tcount,
                       toff, tnq_local, tmap, tq0, rowinfo, idx, rownn, idx_stride, blob_bytes);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_tilemap(int b, int m, int mode, const int *idx, int *tmap, void *stream) {
    return apn_sa_wide_tilemap_many(1, b, m, mode, idx, tmap, stream);
}

// pcnt_poff: int32[2 b n] (counts, then list starts); plist: int32[32 b m]; geo: float[4 b n] (optional);
// optional: fidx (b,m) = the point every query is -> fq int32[b n] = the query a point is, or -1.
extern "C" int apn_sa_wide_csr(int b, int n, int m, const int *idx, const float *new_xyz, const int *tmap,
                               int *pcnt_poff, int *plist, float *geo, const int *fidx, int *fq, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0 || b > 65535 || (long long)b * m > 0x7fffffffLL / 64 ||
        (long long)b * n > 0x7fffffffLL / 8 || !idx || !new_xyz || !tmap || !pcnt_poff || !plist ||
        ((fidx != nullptr) != (fq != nullptr)))
        return APN_EINVAL;
    const int nq = b * m;
    const long long npts = (long long)b * n;
    int *pcnt = pcnt_poff, *poff = pcnt_poff + npts;
    hipStream_t st = (hipStream_t)stream;
    {
        // one launch, one workgroup per cloud, when a cloud's map fits in LDS
        const long long rows_cap = (long long)m * 32;
        long long lds = 4 * (2ll * n + rows_cap + 3ll * m + rows_cap / apn::CSR_INS + 1);
        int bitonic_words = 0;                                   // per wave: a buffer for lists of up to this many rows
        for (int wds = 1024; wds >= 128; wds >>= 1)
            if (lds + 16ll * wds * 4 <= 150 * 1024) { bitonic_words = wds; break; }
        if (m <= 1024 && rows_cap <= 65536 && lds <= 150 * 1024) {
            lds += 16ll * bitonic_words * 4;
            static apn::DynLdsOnce configured;       // (per device: apn_common.h)
            if (hipError_t e = apn::set_dyn_lds(configured, (const void *)apn::csr_cloud_kernel, 150 * 1024)) return (int)e;
            hipLaunchKernelGGL(apn::csr_cloud_kernel, dim3(b), dim3(1024), (size_t)lds, st, nq, b, n, m, tmap, new_xyz, pcnt,
                               poff, plist, geo, fidx, fq, (int)rows_cap, bitonic_words, (int *)nullptr, 0ll, (int *)nullptr);
            APN_LAUNCH_CHECK();
            return APN_OK;
        }
    }
    hipLaunchKernelGGL(apn::csr_init_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, npts, pcnt, fq);
    const unsigned rb = (unsigned)(((long long)nq * 32 + 255) / 256);
    hipLaunchKernelGGL(apn::csr_count_fill_kernel, dim3(rb), dim3(256), 0, st, nq, n, m, 0, idx, tmap, pcnt, poff, plist,
                       fidx, fq);
    hipLaunchKernelGGL(apn::csr_scan_kernel, dim3(b), dim3(1024), 0, st, nq, b, n, tmap, pcnt, poff);
    hipLaunchKernelGGL(apn::csr_count_fill_kernel, dim3(rb), dim3(256), 0, st, nq, n, m, 1, idx, tmap, pcnt, poff, plist,
                       fidx, fq);
    hipLaunchKernelGGL(apn::csr_sort_short_kernel, dim3((unsigned)((npts + 127) / 128)), dim3(128), 0, st, nq, npts, tmap,
                       pcnt, poff, plist, new_xyz, geo);
    hipLaunchKernelGGL(apn::csr_sort_long_kernel, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, st, nq, npts, tmap,
                       pcnt, poff, plist, new_xyz, geo);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// The ROW MAP of the register-resident backward pass (round 5), `count` stacked batches in one launch: per batch z
//   pcnt_poff + z * 2 b n   int32[2 b n]: per support point the number of tile-map rows that gather it, and the first place
//                           of its rows in the point-sorted order (places share the row ids' index space: a cloud's places
//                           lie inside the cloud's own range of row ids);
//   rowdst + z * 32 b m     int32[32 b m]: the place of every live row of map z (tmap + z * apn_sa_wide_tilemap_ints(b, m)): the
//                           row of GU (apn_sa_rowmap_places(b, n, m) rows) its g_u is stored in.  A point's rows occupy
//                           consecutive places in ascending row order.  (Measured and removed in round 5: the first ELL = 8
//                           rows of a point at a MAP-FREE address point * ELL + j, requested together with the count -- one
//                           round trip instead of two dependent ones, but 64 KB instead of 31 KB per 64-point tile and fully
//                           scattered stores: the per-point kernel 15.3 -> 17.1 us.)
// The map is a pure function of the neighbour indices (index-stage data, like the tile map).  scratch: int32[32 b m], used by the multi-launch path only (clouds whose map does not
// fit one workgroup's LDS).
// rows of GU a backward needs: one per tile-map row
extern "C" int apn_sa_rowmap_places(int b, int n, int m) {
    const long long rows = 32ll * b * m;
    return (b <= 0 || n <= 0 || m <= 0 || rows > 0x7fffffffLL / 128) ? 0 : (int)rows;
}

extern "C" int apn_sa_rowmap_ints(int b, int n) {
    const long long v = 2ll * b * n + b;
    return (b <= 0 || n <= 0 || v > 0x7fffffffLL) ? 0 : (int)v;
}

namespace apn {
__global__ __launch_bounds__(256) void rowmap_dup_unknown_kernel(int b, int *__restrict__ dup) {
    for (int k = threadIdx.x; k < b; k += 256) dup[k] = 1;
}
}  // namespace apn

extern "C" int apn_sa_rowmap_many(int count, int b, int n, int m, const int *tmap, const int *fidx, int *pcnt_poff, int *rowdst,
                                  int *scratch, void *stream) {
    if (32ll * b * m > 0x7fffffffLL / 128) return APN_EINVAL;     // places as 32-bit byte offsets / 128
    if (count <= 0 || count > 65535 || b <= 0 || n <= 0 || m <= 0 || b > 65535 || (long long)b * m > 0x7fffffffLL / 64 ||
        (long long)b * n > 0x7fffffffLL / 8 || !tmap || !pcnt_poff || !rowdst || !scratch)
        return APN_EINVAL;
    const int nq = b * m;
    const long long npts = (long long)b * n, tstride = apn_sa_wide_tilemap_ints(b, m);
    hipStream_t st = (hipStream_t)stream;
    const long long rows_cap = (long long)m * 32;
    long long lds = 4 * (2ll * n + rows_cap + 3ll * m + rows_cap / apn::CSR_INS + 1);
    int bitonic_words = 0;
    for (int wds = 1024; wds >= 128; wds >>= 1)
        if (lds + 16ll * wds * 4 <= 150 * 1024) { bitonic_words = wds; break; }
    if (m <= 1024 && rows_cap <= 65536 && lds <= 150 * 1024) {
        lds += 16ll * bitonic_words * 4;
        static apn::DynLdsOnce configured;
        if (hipError_t e = apn::set_dyn_lds(configured, (const void *)apn::csr_cloud_kernel, 150 * 1024)) return (int)e;
        hipLaunchKernelGGL(apn::csr_cloud_kernel, dim3(b, count), dim3(1024), (size_t)lds, st, nq, b, n, m, tmap,
                           (const float *)nullptr, pcnt_poff, pcnt_poff + npts, (int *)nullptr, (float *)nullptr,
                           fidx, (int *)nullptr, (int)rows_cap, bitonic_words, rowdst, tstride, pcnt_poff);
        APN_LAUNCH_CHECK();
        return APN_OK;
    }
    const unsigned rb = (unsigned)(((long long)nq * 32 + 255) / 256);
    for (int z = 0; z < count; ++z) {
        const int *tm = tmap + z * tstride;
        int *pcnt = pcnt_poff + z * (2 * npts + b), *poff = pcnt + npts, *rd = rowdst + (long long)z * 32 * nq;
        hipLaunchKernelGGL(apn::rowmap_dup_unknown_kernel, dim3(1), dim3(256), 0, st, b, pcnt + 2 * npts);   // (picks not examined on this path)
        hipLaunchKernelGGL(apn::csr_init_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, npts, pcnt, (int *)nullptr);
        hipLaunchKernelGGL(apn::csr_count_fill_kernel, dim3(rb), dim3(256), 0, st, nq, n, m, 0, (const int *)nullptr, tm, pcnt, poff,
                           scratch, (const int *)nullptr, (int *)nullptr);
        hipLaunchKernelGGL(apn::csr_scan_kernel, dim3(b), dim3(1024), 0, st, nq, b, n, tm, pcnt, poff);
        hipLaunchKernelGGL(apn::csr_count_fill_kernel, dim3(rb), dim3(256), 0, st, nq, n, m, 1, (const int *)nullptr, tm, pcnt, poff,
                           scratch, (const int *)nullptr, (int *)nullptr);
        hipLaunchKernelGGL(apn::csr_sort_short_kernel, dim3((unsigned)((npts + 127) / 128)), dim3(128), 0, st, nq, npts, tm, pcnt,
                           poff, scratch, (const float *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL(apn::csr_sort_long_kernel, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, st, nq, npts, tm, pcnt,
                           poff, scratch, (const float *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL(apn::rowdst_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, npts, pcnt, poff, scratch, rd);
    }
    APN_LAUNCH_CHECK();
    return APN_OK;
}
