// sa_wide_dense.hip -- the kernels of the width-generic fused path (csrc/sa_wide.hip) whose work runs over
// POINTS, QUERIES or CHANNELS rather than over the B*M*K positions:
//
//   wide_fwd_prep     conv1 hoisted to the points: U = W1f f + W1p p / r (B,N,H), V = W1p new_p / r (B,M,H),
//                     and the MFMA image of W2^T -- one launch
//   wide_out          out (B,O,M) = ysel (B,M,O) * scale2 + shift2 (BatchNorm-2 applied to the pooled extreme)
//   wide_bwd_mid      BatchNorm-2 backward constants D2, E2 (+ dgamma2, dbeta2), Qm = W2^T diag(D2) W2,
//                     evec = E2 W2 and the MFMA image of [W2 ; Qm] -- one launch
//   wide_bwd_fin      BatchNorm-1 backward constants {ca, cb, cc} (+ dgamma1, dbeta1) and
//                     dL/dW2 = R_S + D2 (W2 Gram) + E2 (x) suma
//   wide_point_grads  per point: dL/dU = G (summed over the rows that gather the point, in the fixed order of the
//                     index stage's inverse map -- no float atomics), dL/df = G W1f, dL/dp = G W1p / r; per query:
//                     dL/dnew_p; and every workgroup's share of dL/dW1 = G^T [p / r, f] - Hq^T [new_p / r, 0]
//
// Reference semantics: openpoints/models/backbone/pointnext.py:157-166 (convs + BatchNorm + max) over
// group.py:235-255 (grouping with relative positions / radius); the backward is the chain rule through them.
// All are float32 FMA kernels (the MFMA work is in sa_wide.hip); sums that leave a workgroup leave as one
// partial row per workgroup, added in a fixed order by wide_colsum: every result is run-to-run reproducible.
#include "apn_common.h"
#include "apn_mfma.h"

namespace apn {

// ------------------------------------------------------------------------------------------
// forward prep.  Blocks [0, pblocks): 64 points each; [pblocks, pblocks + qblocks): V; the rest: image words.
// Point blocks: thread (point tx, wave ty) accumulates HPW = H/4 channels h0 = ty * HPW ... of its point (the
// weights are wave-uniform: scalar loads), the tile leaves through LDS as whole rows of U.
// ------------------------------------------------------------------------------------------
template <int HPW>
__global__ __launch_bounds__(256, 2) void wide_fwd_prep_kernel(int B, int C, int N, int M, int pblocks, int qblocks,
                                                            const float *__restrict__ f, const float *__restrict__ p,
                                                            const float *__restrict__ new_p,
                                                            const float *__restrict__ w1, float inv_r,
                                                            float *__restrict__ U, float *__restrict__ V,
                                                            const float *__restrict__ w2, int O, int ct,
                                                            uint4 *__restrict__ img, const int *__restrict__ fq,
                                                            float *__restrict__ fs, const float *__restrict__ geo,
                                                            float *__restrict__ part1) {
    constexpr int H = 4 * HPW;
    extern __shared__ float sm[];                       // [64][H + 1]
    const int ldw = C + 3;
    const int blk = blockIdx.x;
    if (blk < pblocks) {
        const int tx = threadIdx.x & 63;
        const int h0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) * HPW);
        const long long npts = (long long)B * N;
        const long long pt = (long long)blk * 64 + tx;
        const long long ptc = pt < npts ? pt : npts - 1;
        const int b = (int)(ptc / N), n = (int)(ptc % N);
        const float *__restrict__ fb = f + (size_t)b * C * N + n;
        float acc[HPW];
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) acc[hh] = 0.0f;
        const float px = p[ptc * 3], py = p[ptc * 3 + 1], pz = p[ptc * 3 + 2];
        // the weights, transposed, once per block: wt[c'][h] -- a point's FMA chain then reads its wave's HPW
        // output channels of input channel c' as broadcast 16-byte LDS words
        float *wt = sm + 64 * (H + 1);
        for (int e = threadIdx.x; e < H * ldw; e += 256) {
            const int hh = e / ldw, cc = e - hh * ldw;
            wt[cc * H + hh] = w1[e];
        }
        for (int c0 = 0; c0 < C; c0 += 32) {
            float fv[32];                                        // one round trip for 32 channels of this point
#pragma unroll
            for (int j = 0; j < 32; ++j) fv[j] = c0 + j < C ? fb[(size_t)(c0 + j) * N] : 0.0f;
            if (c0 == 0) __syncthreads();
            if (fs && h0 == 0 && pt < npts) {
                // the sampled points' own features as query-major rows fs[b][q][:] (what the residual branch reads,
                // forward and backward): this thread holds its point's channels anyway
                const int q = fq[pt];
                if (q >= 0) {
                    float4 *__restrict__ dst = reinterpret_cast<float4 *>(fs + ((size_t)b * M + q) * C + c0);
#pragma unroll
                    for (int j4 = 0; j4 < 8; ++j4)
                        if (c0 + 4 * j4 < C) dst[j4] = make_float4(fv[4 * j4], fv[4 * j4 + 1], fv[4 * j4 + 2], fv[4 * j4 + 3]);
                }
            }
#pragma unroll 8
            for (int j = 0; j < 32; ++j) {
                if (c0 + j < C) {
                    const float4 *wr = reinterpret_cast<const float4 *>(wt + (3 + c0 + j) * H + h0);
#pragma unroll
                    for (int v = 0; v < HPW / 4; ++v) {
                        const float4 wv = wr[v];
                        acc[4 * v] = __builtin_fmaf(wv.x, fv[j], acc[4 * v]);
                        acc[4 * v + 1] = __builtin_fmaf(wv.y, fv[j], acc[4 * v + 1]);
                        acc[4 * v + 2] = __builtin_fmaf(wv.z, fv[j], acc[4 * v + 2]);
                        acc[4 * v + 3] = __builtin_fmaf(wv.w, fv[j], acc[4 * v + 3]);
                    }
                }
            }
        }
        // BatchNorm-1's statistics without a pass over the positions: y1 = U[n] - V[q] over the (query, slot)
        // pairs, and everything a point contributes is known from the index stage's geo = {occ, SP = sum of the
        // gathering queries' coordinates}:
        //   sum y1   = sum_n occ U            - 32 sum_q V
        //   sum y1^2 = sum_n (occ U^2 - 2 U (W1p . SP) / r)  + 32 sum_q V^2     (V[q] = W1p . new_p[q] / r)
        // one partial row per block {sum, sumsq}[H] (the query terms come from the V blocks below)
        float4 ge = make_float4(0.f, 0.f, 0.f, 0.f);
        if (part1 && pt < npts) ge = *reinterpret_cast<const float4 *>(geo + pt * 4);
        float t1[HPW], tq[HPW];
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const float wx = wt[h0 + hh], wy = wt[H + h0 + hh], wz = wt[2 * H + h0 + hh];
            const float pw = __builtin_fmaf(wz, pz, __builtin_fmaf(wy, py, wx * px));
            const float u = __builtin_fmaf(pw, inv_r, acc[hh]);
            sm[tx * (H + 1) + h0 + hh] = u;
            const float spw = __builtin_fmaf(wz, ge.w, __builtin_fmaf(wy, ge.z, wx * ge.y)) * inv_r;
            t1[hh] = ge.x * u;
            tq[hh] = u * __builtin_fmaf(ge.x, u, -2.0f * spw);
        }
        if (part1) {
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) {
#pragma unroll
                for (int hh = 0; hh < HPW; ++hh) {
                    t1[hh] += __shfl_xor(t1[hh], k);
                    tq[hh] += __shfl_xor(tq[hh], k);
                }
            }
            if (tx == 0) {
#pragma unroll
                for (int hh = 0; hh < HPW; ++hh) {
                    part1[(size_t)blk * 2 * H + h0 + hh] = t1[hh];
                    part1[(size_t)blk * 2 * H + H + h0 + hh] = tq[hh];
                }
            }
        }
        __syncthreads();
        const long long base = (long long)blk * 64;
        for (int e = threadIdx.x; e < 64 * H; e += 256) {
            const int pl = e / H, h = e - pl * H;
            if (base + pl < npts) U[(base + pl) * H + h] = sm[pl * (H + 1) + h];
        }
    } else if (blk < pblocks + qblocks) {
        // 64 queries: thread (channel h, query subgroup); the block's row of part1 = {-32 sum V, 32 sum V^2}
        const int h = threadIdx.x % H, qs = threadIdx.x / H;
        const long long nqry = (long long)B * M, qbase = (long long)(blk - pblocks) * 64;
        const float *__restrict__ wr = w1 + (size_t)h * ldw;
        const float wx = wr[0], wy = wr[1], wz = wr[2];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 4
        for (int ql = qs; ql < 64; ql += 256 / H) {
            const long long q = qbase + ql;
            if (q < nqry) {
                const float *__restrict__ qp = new_p + q * 3;
                const float v = __builtin_fmaf(wz, qp[2], __builtin_fmaf(wy, qp[1], wx * qp[0])) * inv_r;
                V[q * H + h] = v;
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);
            }
        }
        if (part1) {
            sm[qs * H + h] = s1;
            sm[256 + qs * H + h] = s2;
            __syncthreads();
            if (threadIdx.x < H) {
                float a1 = 0.0f, a2 = 0.0f;
                for (int g = 0; g < 256 / H; ++g) { a1 += sm[g * H + h]; a2 += sm[256 + g * H + h]; }
                part1[(size_t)blk * 2 * H + h] = -32.0f * a1;
                part1[(size_t)blk * 2 * H + H + h] = 32.0f * a2;
            }
        }
    } else {
        // image of W2^T (H x O): word [cb][kc][j][s][part][lane], see wide_image_kernel (sa_wide_glue.hip)
        const int nkc = H / 32, ncb = O / (32 * ct);
        const int total = ncb * nkc * ct * 2 * 2 * 64;
        const int w = (blk - pblocks - qblocks) * 256 + threadIdx.x;
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
            const float v = w2[(size_t)col * H + k0 + e];
            const __bf16 hi = (__bf16)v;
            o[e] = part == 0 ? hi : (__bf16)(v - (float)hi);
        }
        img[w] = __builtin_bit_cast(uint4, o);
    }
}

// ------------------------------------------------------------------------------------------
// out[b][o][q] = act(ysel[b][q][o] * scale2[o] + shift2[o] + skip[b][o][q]): 32 x 64 tiles through LDS,
// grid (M/32, O/64, B).  skip (optional) = Ws fs[b, q, :] + bs: the block's residual branch, a 1x1
// convolution of the SAMPLED points' own features (pointnext.py:150-153, 167-168; fs = their rows, gathered
// by wide_fwd_prep), 64 input channels at a time through LDS; act = ReLU when `relu`.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_out_kernel(int M, int O, int C, const float *__restrict__ ysel,
                                                       const float *__restrict__ pack2, const float *__restrict__ fs,
                                                       const float *__restrict__ ws,
                                                       const float *__restrict__ bs, int relu,
                                                       float *__restrict__ out) {
    // tile = 32 queries x 64 channels (small tiles: the kernel is a load -> LDS -> store chain, it needs many
    // blocks in flight, not big ones); thread (query tx, group ty) owns channels ty, ty + 8, ...
    __shared__ float tile[32][65];    // [q][o]  BatchNorm-2 applied to the pooled extreme
    __shared__ float fg[32][65];      // [q][c]
    __shared__ float wt[64][65];      // [o][c]
    const int b = blockIdx.z, c0 = blockIdx.y * 64, m0 = blockIdx.x * 32;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    // (every load of a phase unconditional on a clamped index, the surplus dropped: a load under a condition is
    // waited for before the next one is issued)
    {
        float yv[8];
        const float sc2 = pack2[c0 + (t & 63)], sh2 = pack2[O + c0 + (t & 63)];       // c = e & 63 = t & 63 for every k
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = (t + 256 * k) >> 6;
            yv[k] = ysel[((size_t)b * M + (m0 + q < M ? m0 + q : M - 1)) * O + c0 + (t & 63)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = (t + 256 * k) >> 6;
            tile[q][t & 63] = m0 + q < M ? __builtin_fmaf(yv[k], sc2, sh2) : 0.0f;
        }
    }
    float bsv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsv[i] = bs ? bs[c0 + (t >> 5) + 8 * i] : 0.0f;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    if (ws) {
        for (int k0 = 0; k0 < C; k0 += 64) {
            __syncthreads();
            float fv[8], wv[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = t + 256 * k, q = e >> 6, c = e & 63;
                fv[k] = fs[((size_t)b * M + (m0 + q < M ? m0 + q : M - 1)) * C + (k0 + c < C ? k0 + c : C - 1)];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int e = t + 256 * k, o = e >> 6, c = e & 63;
                wv[k] = ws[(size_t)(c0 + o) * C + (k0 + c < C ? k0 + c : C - 1)];
            }
            const bool cin = k0 + (t & 63) < C;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                fg[(t + 256 * k) >> 6][t & 63] = (cin && m0 + ((t + 256 * k) >> 6) < M) ? fv[k] : 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) wt[(t + 256 * k) >> 6][t & 63] = cin ? wv[k] : 0.0f;
            __syncthreads();
            const int kn = C - k0 < 64 ? C - k0 : 64;
#pragma unroll 4
            for (int c = 0; c < kn; ++c) {
                const float x = fg[tx][c];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(wt[ty + 8 * i][c], x, acc[i]);
            }
        }
    }
    __syncthreads();
    const int q = m0 + tx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cc = ty + 8 * i;
        float v = tile[tx][cc] + acc[i] + bsv[i];
        if (relu) v = v > 0.0f ? v : 0.0f;
        if (q < M) out[((size_t)b * O + c0 + cc) * M + q] = v;
    }
}

// ------------------------------------------------------------------------------------------
// backward, between the prep of the upstream gradient and the pass over the positions.
// Block (cb, kc) of the image of Z = [W2 ; Qm] ((O + H) x H); 1024 threads.
//   every Qm block (and block 0) first forms D2, E2 [O] from partS (fixed-order float64 column sums) or from
//   `sums` (float64 {S1[O], S2[O], global count, world}: SyncBatchNorm, already all-reduced);
//   block 0 also writes d2e2, dgamma2, dbeta2 and evec = E2 W2.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wide_bwd_mid_kernel(const float *__restrict__ partS, int rows,
                                                            const double *__restrict__ sums, int H, int O, int ct,
                                                            const float *__restrict__ pack2, double count,
                                                            int training, const float *__restrict__ w2,
                                                            float *__restrict__ d2e2, float *__restrict__ g_gamma2,
                                                            float *__restrict__ g_beta2, float *__restrict__ evec,
                                                            uint4 *__restrict__ zimg) {
    // Everything a block needs arrives in ONE round of loads (the kernel is a chain of dependent steps on a few
    // KB: its time is the number of round trips, not the bytes): W2 (O x H <= 32 KB) and the rows of partS.
    extern __shared__ double dsm[];
    double *red = dsm;                                            // [1024]
    double *S = red + 1024;                                       // S1 | S2 [2 O]
    float *de = reinterpret_cast<float *>(S + 2 * O);             // D2[O], E2[O]
    float *w2s = de + 2 * O;                                      // [O][H + 1]
    float *qt = w2s + O * (H + 1);                                // [32][32 ct + 1]
    const int nkc = (O + H) / 32;
    const int cb = blockIdx.x / nkc, kc = blockIdx.x % nkc;
    const bool qm_rows = kc >= O / 32;
    const int t = threadIdx.x;
    const int ncol = 2 * O, groups = 1024 / ncol;                 // 2 O <= 256: >= 4 row groups
    const int col = t % ncol, grp = t / ncol;
    double gscale = 1.0;
    if (sums) { count = sums[2 * O]; gscale = 1.0 / sums[2 * O + 1]; }
    double acc = 0.0;
    if (!sums && grp < groups) {
        for (int r0 = grp; r0 < rows; r0 += 16 * groups) {        // 16 independent loads in flight
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {      // (clamped row, dropped below: a load under a condition is waited for on the spot)
                const int rr = r0 + u * groups;
                v[u] = partS[(size_t)(rr < rows ? rr : rows - 1) * ncol + col];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += r0 + u * groups < rows ? (double)v[u] : 0.0;
        }
    }
    for (int e = t; e < O * H; e += 1024) w2s[(e / H) * (H + 1) + e % H] = w2[e];
    red[t] = acc;
    __syncthreads();
    if (t < ncol) {
        double tot = 0.0;
        if (sums) tot = sums[t];
        else
            for (int g = 0; g < groups; ++g) tot += red[g * ncol + t];
        S[t] = tot;
    }
    __syncthreads();
    if (t < O) {
        const double s1 = S[t], s2 = S[O + t];
        const double sc = pack2[t], mu = pack2[2 * O + t], iv = pack2[3 * O + t];
        double dv = 0.0, ev = 0.0;
        if (training) {
            dv = -sc * iv * s2 / count;
            ev = -sc * s1 / count + sc * mu * iv * s2 / count;
        }
        de[t] = (float)dv;
        de[O + t] = (float)ev;
        if (blockIdx.x == 0) {
            d2e2[t] = (float)dv;
            d2e2[O + t] = (float)ev;
            if (g_gamma2) g_gamma2[t] = (float)(s2 * gscale);
            if (g_beta2) g_beta2[t] = (float)(s1 * gscale);
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && t < H) {
        float e0 = 0.0f, e1 = 0.0f;
#pragma unroll 8
        for (int c = 0; c < O; c += 2) {
            e0 = __builtin_fmaf(de[O + c], w2s[c * (H + 1) + t], e0);
            e1 = __builtin_fmaf(de[O + c + 1], w2s[(c + 1) * (H + 1) + t], e1);
        }
        evec[t] = e0 + e1;
    }
    const int words = ct * 256;                                   // [j][s][part][lane] of this chunk
    uint4 *__restrict__ dst = zimg + (size_t)blockIdx.x * words;
    if (qm_rows) {
        // Qm tile: rows k' = kq0 + (t >> 5), columns mid0 + 32 j + (t & 31);  Qm = W2^T diag(D2) W2
        const int kq = (kc - O / 32) * 32 + (t >> 5), r = t & 31, mid0 = cb * ct * 32;
        float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
        for (int c = 0; c < O; ++c) {
            const float a = w2s[c * (H + 1) + kq] * de[c];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < ct) q[j] = __builtin_fmaf(a, w2s[c * (H + 1) + mid0 + 32 * j + r], q[j]);
        }
        const int ldq = 32 * ct + 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < ct) qt[(t >> 5) * ldq + 32 * j + r] = q[j];
        __syncthreads();
    }
    if (t < words) {
        const int lane = t & 63, part = (t >> 6) & 1, s = (t >> 7) & 1, j = t >> 8;
        const int k0 = s * 16 + (lane >> 5) * 8;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = qm_rows ? qt[(k0 + e) * (32 * ct + 1) + j * 32 + (lane & 31)]
                                    : w2s[(kc * 32 + k0 + e) * (H + 1) + (cb * ct + j) * 32 + (lane & 31)];
            const __bf16 hi = (__bf16)v;
            o[e] = part == 0 ? hi : (__bf16)(v - (float)hi);
        }
        dst[t] = __builtin_bit_cast(uint4, o);
    }
}

// ------------------------------------------------------------------------------------------
// backward, after the passes over the positions.  R (float64): [(O + H) H] = [R_S ; Gram], then suma[H].
//   block 0 (first): BatchNorm-1 backward constants from partT (or `sums`): cabc = {ca, cb, cc}[H], dgamma1, dbeta1
//   every block: 256 elements of dL/dW2[c][mid] = R_S + D2[c] (W2[c] . Gram[:, mid]) + E2[c] suma[mid]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wide_bwd_fin_kernel(const float *__restrict__ partT, int rows,
                                                            const double *__restrict__ sums, int H, int O,
                                                            const float *__restrict__ pack1, double count, int training,
                                                            float *__restrict__ cabc, float *__restrict__ g_gamma1,
                                                            float *__restrict__ g_beta1, const double *__restrict__ R,
                                                            const float *__restrict__ d2e2,
                                                            const float *__restrict__ w2, float *__restrict__ g_w2) {
    // the LAST block does the BatchNorm-1 constants (a column sum over the rows of partT: 1024 threads = 2H columns
    // x row groups, 16 loads in flight each); the others 1024 elements of dL/dW2 each, Gram staged in LDS
    extern __shared__ double dsm[];
    const int t = threadIdx.x;
    if (blockIdx.x == gridDim.x - 1) {
        double *red = dsm;                                         // [1024]
        const int ncol = 2 * H, groups = 1024 / ncol;
        const int col = t % ncol, grp = t / ncol;
        double gscale = 1.0;
        if (sums) { count = sums[2 * H]; gscale = 1.0 / sums[2 * H + 1]; }
        double acc = 0.0;
        if (!sums && grp < groups) {
            for (int r0 = grp; r0 < rows; r0 += 16 * groups) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int rr = r0 + u * groups;
                    v[u] = partT[(size_t)(rr < rows ? rr : rows - 1) * ncol + col];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += r0 + u * groups < rows ? (double)v[u] : 0.0;
            }
        }
        red[t] = acc;
        __syncthreads();
        if (t < ncol) {
            double tot = 0.0;
            if (sums) tot = sums[t];
            else
                for (int g = 0; g < groups; ++g) tot += red[g * ncol + t];
            red[t] = tot;                                           // (row 0 of red: own column only)
        }
        __syncthreads();
        if (t < H) {
            const double t1 = red[t], t2 = red[H + t], sc = pack1[t];
            cabc[t] = (float)sc;
            cabc[H + t] = training ? (float)(-sc * t2 / count) : 0.0f;
            cabc[2 * H + t] = training ? (float)(-sc * t1 / count) : 0.0f;
            if (g_gamma1) g_gamma1[t] = (float)(t2 * gscale);
            if (g_beta1) g_beta1[t] = (float)(t1 * gscale);
        }
        return;
    }
    double *gram = dsm;                                            // [H][H]
    float *w2s = reinterpret_cast<float *>(gram + (H * H > 1024 ? H * H : 1024));      // this block's 1024 elements of W2
    const double *__restrict__ gsrc = R + (size_t)O * H;
    for (int e = t; e < H * H; e += 1024) gram[e] = gsrc[e];
    const int e = blockIdx.x * 1024 + t;
    const bool ok = e < O * H;
    w2s[t] = ok ? w2[e] : 0.0f;
    const int c = ok ? e / H : 0, mid = ok ? e - c * H : 0;
    const double r0 = ok ? R[e] : 0.0, sm = R[(size_t)(O + H) * H + mid];
    const float dc = d2e2[c], ec = d2e2[O + c];
    __syncthreads();
    const float *wr = w2s + (t / H) * H;                           // row c of W2
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < H; ++k) acc += (double)wr[k] * gram[k * H + mid];
    if (ok) g_w2[e] = (float)(r0 + (double)dc * acc + (double)ec * sm);
}

// ------------------------------------------------------------------------------------------
// per-point gradients (C, H <= 64: every weight sits in LDS).  Block = 64 consecutive points + its share of
// the queries (qpb consecutive ones), 256 threads.
//   1. G[pt][h] = ca sum_{rows of pt} GU[row][h] + cb inv1 (occ (U - mean1) - SP . W1p[h] / r) + cc occ   -> LDS
//      (rows: the index stage's inverse map pcnt / poff / plist, ascending: a fixed order)
//   2. dL/df[b][c][n] = sum_h G W1f[h][c]  (+ sum_o gpre[q][o] Ws[o][c] when point n is query q: the residual
//      branch's input gradient);  dL/dp[pt][d] = sum_h G W1p[h][d] / r
//   3. this block's share of dL/dW1[h][c'] = sum_pt G[pt][h] X[c'][pt], X = [p / r ; f] (LDS), minus its
//      queries' Hq[q][h] new_p[q][d] / r in the coordinate columns;  Hq = ca HA + cb HB + 32 cc;
//      of dL/dWs[o][c] = sum_q gpre[q][o] fs[q][c] and of dL/dbs[o] = sum_q gpre[q][o]
//   4. dL/dnew_p[q][d] = -sum_h Hq W1p[h][d] / r
//   Wpart[block][H (C + 3) + O C + O]: summed over blocks by wide_colsum_f32 (fixed order).
// ------------------------------------------------------------------------------------------
struct PointGradArgs {
    int B, C, N, M, O, qpb;              // qpb: queries per block;  O: skip-branch rows (0: no skip branch)
    const float *GU;                     // (rows, H) dL/da1-side sums per row (sa_wide.hip: wide_bwd_main)
    const int *pcnt, *poff, *plist;
    const float *geo;                    // (B N, 4)
    const float *U, *f, *p, *new_p;
    const float *HA, *HB;                // (B M, H)
    const float *cabc, *pack1, *w1;
    const float *gpre;                   // (B M, O) gradient at the block's pre-activation output
    const int *fq;                       // (B N): the query a point is (-1: none)
    const float *fs;                     // (B M, C): the sampled points' own features (wide_fwd_prep)
    const float *ws;                     // (O, C)
    float inv_r;
    float *g_f, *g_p, *g_q, *Wpart;      // g_p / g_q may be null
};

__host__ __device__ inline int up4(int x) { return (x + 3) & ~3; }

// LDS layout of wide_point_grads (floats; every sub-array starts 16-byte aligned)
struct PgLds {
    int gs, xs, w1f, w1p, hqs, nps, fqs, wss, gps, fgs, gpt, total;
    __host__ __device__ PgLds(int H, int C, int O, int qpb) {
        int o = 0;
        gs = o; o += up4(64 * (H + 1));
        xs = o; o += up4((C + 3) * 65);
        w1f = o; o += H * C;
        w1p = o; o += H * 4;
        hqs = o; o += up4(qpb * (H + 1));
        nps = o; o += up4(qpb * 3);
        fqs = o; o += 64;
        wss = o; o += O * up4(C);
        gps = o; o += up4(qpb * (O + 1));
        fgs = o; o += O ? up4(C * (qpb + 1)) : 0;
        gpt = o; o += O ? up4(64 * (O + 1)) : 0;
        total = o;
    }
};

// n values through registers, NB loads in flight per thread: the fills of the LDS tiles below are dependent chains
// (index -> row) only across phases, never inside a loop.  `load` must itself be free of conditions around its loads
// (clamp the address, select the value): a load under a condition sits in a basic block of its own, next to its
// first use, and is waited for before the next one is issued
template <int NB, typename Load, typename Store>
__device__ __forceinline__ void fill_batched(int n, Load load, Store store, int tid = threadIdx.x, int nthreads = 256) {
    for (int e0 = tid; e0 < n; e0 += nthreads * NB) {
        float v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = load(e0 + nthreads * u < n ? e0 + nthreads * u : n - 1);   // (the surplus: dropped)
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (e0 + nthreads * u < n) store(e0 + nthreads * u, v[u]);
    }
}

template <int HPW>
__global__ __launch_bounds__(256, 2) void wide_point_grads_kernel(PointGradArgs a) {
    constexpr int H = 4 * HPW;
    extern __shared__ float sm[];
    const int C = a.C, ldw = C + 3, O = a.O, qpb = a.qpb, C4 = up4(C);
    const PgLds L(H, C, O, qpb);
    float *Gs = sm + L.gs;                           // [64][H + 1]
    float *Xs = sm + L.xs;                           // [C + 3][65]
    float *W1f = sm + L.w1f;                         // [H][C]      feature columns of W1
    float *W1p = sm + L.w1p;                         // [H][4]      coordinate columns
    float *Hqs = sm + L.hqs;                         // [qpb][H + 1]
    float *nps = sm + L.nps;                         // [qpb][3]    the block's queries' coordinates / r
    int *fqs = reinterpret_cast<int *>(sm + L.fqs);  // [64]        the query each of the block's points is, or -1
    float *Wss = sm + L.wss;                         // [O][C4]     (skip branch)
    float *gps = sm + L.gps;                         // [qpb][O + 1]  gpre rows of the block's queries
    float *fgs = sm + L.fgs;                         // [C][qpb + 1]  their source points' features
    float *gpt = sm + L.gpt;                         // [64][O + 1]   gpre rows of the block's POINTS that are queries
    const int tx = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long npts = (long long)a.B * a.N, nqry = (long long)a.B * a.M;
    const long long pbase = (long long)blockIdx.x * 64;
    const long long pt = pbase + tx;
    const bool ok = pt < npts;
    const long long ptc = ok ? pt : npts - 1;
    const int b = (int)(ptc / a.N), n = (int)(ptc % a.N);
    const long long q0 = (long long)blockIdx.x * qpb;
    const int nq = (int)(q0 >= nqry ? 0 : (nqry - q0 < qpb ? nqry - q0 : qpb));
    const int h0 = ty * HPW;
    // ---- phase 0: everything that does not depend on another load
    const int cnt = ok ? a.pcnt[ptc] : 0;
    const int poff = a.poff[ptc];
    const int *__restrict__ l = a.plist + poff;
    const float4 ge = *reinterpret_cast<const float4 *>(a.geo + ptc * 4);
    float ur[HPW];
#pragma unroll
    for (int v = 0; v < HPW / 4; ++v) {
        const float4 x = *reinterpret_cast<const float4 *>(a.U + ptc * H + h0 + 4 * v);
        ur[4 * v] = x.x; ur[4 * v + 1] = x.y; ur[4 * v + 2] = x.z; ur[4 * v + 3] = x.w;
    }
    if (O) {
        if (ty == 0) fqs[tx] = ok ? a.fq[ptc] : -1;
    }
    // the tiles of this phase, one wave each: four independent load chains in flight instead of one after the other
    if (ty == 0) {
        // X tile: coordinates / r, then the features
        fill_batched<4>(3 * 64, [&](int e) {
            const long long g = pbase + (e & 63);
            return a.p[(g < npts ? g : npts - 1) * 3 + (e >> 6)] * a.inv_r;
        }, [&](int e, float v) { Xs[(e >> 6) * 65 + (e & 63)] = pbase + (e & 63) < npts ? v : 0.0f; }, tx, 64);
        fill_batched<16>(C * 64, [&](int e) {
            const long long g0 = pbase + (e & 63), g = g0 < npts ? g0 : npts - 1;
            return a.f[((size_t)(g / a.N) * C + (e >> 6)) * a.N + (int)(g % a.N)];
        }, [&](int e, float v) { Xs[(3 + (e >> 6)) * 65 + (e & 63)] = pbase + (e & 63) < npts ? v : 0.0f; }, tx, 64);
    } else if (ty == 1) {
        fill_batched<16>(H * ldw, [&](int e) { return a.w1[e]; }, [&](int e, float v) {
            const int hh = e / ldw, cc = e - hh * ldw;
            if (cc < 3) W1p[hh * 4 + cc] = v; else W1f[hh * C + cc - 3] = v;
        }, tx, 64);
        fill_batched<4>(nq * 3, [&](int e) { return a.new_p[q0 * 3 + e] * a.inv_r; }, [&](int e, float v) { nps[e] = v; }, tx, 64);
        fill_batched<16>(nq * H, [&](int e) {
            const int hh = e % H;
            const size_t g = (size_t)q0 * H + e;
            return __builtin_fmaf(a.cabc[hh], a.HA[g], __builtin_fmaf(a.cabc[H + hh], a.HB[g], 32.0f * a.cabc[2 * H + hh]));
        }, [&](int e, float v) { Hqs[(e / H) * (H + 1) + e % H] = v; }, tx, 64);
    } else if (O) {
        if (ty == 2)
            fill_batched<16>(O * C, [&](int e) { return a.ws[e]; }, [&](int e, float v) { Wss[(e / C) * C4 + e % C] = v; }, tx, 64);
        else if (a.Wpart)
            fill_batched<16>(nq * O, [&](int e) { return a.gpre[(size_t)q0 * O + e]; },
                             [&](int e, float v) { gps[(e / O) * (O + 1) + e % O] = v; }, tx, 64);
    }
    __syncthreads();
    // ---- phase 1: what needed an index first (the skip branch's rows; the rows that gather each point)
    float acc[HPW];
#pragma unroll
    for (int v = 0; v < HPW; ++v) acc[v] = 0.0f;
    // A point's first PG_CAP rows are summed by its own lane; what a HOT point has beyond them (a point of a cluster of
    // duplicates is a neighbour of every query around it: AdaptPoint's masked points collapse onto each other, and a list
    // reaches M rows) is summed by the whole wave, lane = row -- with one lane per point the wave walked the longest of
    // its 64 lists four rows per dependent round trip: 298 us for the classifier's second stage on generated clouds
    // against 64 us on the real ones.  Fixed orders both (ascending rows, then a butterfly): bit-reproducible.
    constexpr int PG_CAP = 32;
    const int cnt_own = cnt < PG_CAP ? cnt : PG_CAP;
    for (int i0 = 0; i0 < cnt_own; i0 += 4) {                  // four rows in flight
        int rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rr[u] = l[i0 + u < cnt_own ? i0 + u : cnt_own - 1];   // (clamped: the surplus is dropped)
        float4 x[4][HPW / 4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 *__restrict__ g = reinterpret_cast<const float4 *>(a.GU + (size_t)rr[u] * H + h0);
#pragma unroll
            for (int v = 0; v < HPW / 4; ++v) x[u][v] = g[v];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u < cnt_own) {                   // (registers only: no load inside)
#pragma unroll
                for (int v = 0; v < HPW / 4; ++v) {
                    acc[4 * v] += x[u][v].x; acc[4 * v + 1] += x[u][v].y; acc[4 * v + 2] += x[u][v].z; acc[4 * v + 3] += x[u][v].w;
                }
            }
        }
    }
    for (unsigned long long hot = __builtin_amdgcn_ballot_w64(cnt > PG_CAP); hot; hot &= hot - 1) {     // (wave-uniform)
        const int src = __builtin_ctzll(hot);
        const int pc = __builtin_amdgcn_readlane(cnt, src);
        const int *__restrict__ pl = a.plist + __builtin_amdgcn_readlane(poff, src);
        float part[HPW];
#pragma unroll
        for (int v = 0; v < HPW; ++v) part[v] = 0.0f;
        // 256 rows per step, four per lane: one round of index loads, one round of row loads (the hot points of a collapsed
        // cloud are its FIRST 32 points -- one block holds them all, and with 64 rows per step it walked 32 lists of M rows
        // at two round trips per 64 rows: the whole kernel waited for that block)
        for (int i0 = PG_CAP; i0 < pc; i0 += 256) {
            int row[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + tx + 64 * u;
                row[u] = pl[i < pc ? i : pc - 1];
            }
            float4 x[4][HPW / 4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 *__restrict__ g = reinterpret_cast<const float4 *>(a.GU + (size_t)row[u] * H + h0);
#pragma unroll
                for (int v = 0; v < HPW / 4; ++v) x[u][v] = g[v];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + tx + 64 * u < pc) {
#pragma unroll
                    for (int v = 0; v < HPW / 4; ++v) {
                        part[4 * v] += x[u][v].x; part[4 * v + 1] += x[u][v].y; part[4 * v + 2] += x[u][v].z; part[4 * v + 3] += x[u][v].w;
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < HPW; ++v) {
            float t = part[v];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (tx == src) acc[v] += t;
        }
    }
    if (O) {
        if (ty < 3)
            fill_batched<16>(64 * O, [&](int e) {
                const int pl = e / O, o = e - pl * O;
                const int q = fqs[pl];
                const long long g0 = pbase + pl, g = g0 < npts ? g0 : npts - 1;
                return a.gpre[((size_t)(g / a.N) * a.M + (q >= 0 ? q : 0)) * O + o];
            }, [&](int e, float v) { gpt[(e / O) * (O + 1) + e % O] = fqs[e / O] >= 0 ? v : 0.0f; }, (int)threadIdx.x, 192);
        else if (a.Wpart)
            fill_batched<16>(nq * C, [&](int e) { return a.fs[(size_t)q0 * C + e]; },
                             [&](int e, float v) { fgs[(e % C) * (qpb + 1) + e / C] = v; }, tx, 64);
    }
    // 1. G = ca sum GU + cb inv1 (occ (U - mean1) - SP . W1p / r) + cc occ
#pragma unroll
    for (int v = 0; v < HPW; ++v) {
        const int h = h0 + v;
        const float spw = __builtin_fmaf(ge.w, W1p[h * 4 + 2], __builtin_fmaf(ge.z, W1p[h * 4 + 1], ge.y * W1p[h * 4]));
        const float yh = a.pack1[3 * H + h] * (ge.x * (ur[v] - a.pack1[2 * H + h]) - spw * a.inv_r);
        const float G = __builtin_fmaf(a.cabc[h], acc[v], __builtin_fmaf(a.cabc[H + h], yh, a.cabc[2 * H + h] * ge.x));
        Gs[tx * (H + 1) + h] = ok ? G : 0.0f;
    }
    __syncthreads();
    // 2. wave ty: channels [cbeg, cend) (at most 16, a multiple of 4): dL/df = G W1f (+ Ws^T gpre at the sampled points)
    {
        const int cpw = up4((C + 3) / 4);
        const int cbeg = ty * cpw, cend = min(C, cbeg + cpw);
        float o16[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) o16[j] = 0.0f;
        if (cbeg < cend) {
#pragma unroll 4
            for (int h = 0; h < H; ++h) {
                const float g = Gs[tx * (H + 1) + h];
                const float4 *wr = reinterpret_cast<const float4 *>(W1f + h * C + cbeg);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (cbeg + 4 * k < cend) {
                        const float4 wv = wr[k];
                        o16[4 * k] = __builtin_fmaf(g, wv.x, o16[4 * k]);
                        o16[4 * k + 1] = __builtin_fmaf(g, wv.y, o16[4 * k + 1]);
                        o16[4 * k + 2] = __builtin_fmaf(g, wv.z, o16[4 * k + 2]);
                        o16[4 * k + 3] = __builtin_fmaf(g, wv.w, o16[4 * k + 3]);
                    }
                }
            }
            if (O) {                               // rows of points that are no query are zero
#pragma unroll 4
                for (int o = 0; o < O; ++o) {
                    const float g = gpt[tx * (O + 1) + o];
                    const float4 *wr = reinterpret_cast<const float4 *>(Wss + o * C4 + cbeg);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (cbeg + 4 * k < cend) {
                            const float4 wv = wr[k];
                            o16[4 * k] = __builtin_fmaf(g, wv.x, o16[4 * k]);
                            o16[4 * k + 1] = __builtin_fmaf(g, wv.y, o16[4 * k + 1]);
                            o16[4 * k + 2] = __builtin_fmaf(g, wv.z, o16[4 * k + 2]);
                            o16[4 * k + 3] = __builtin_fmaf(g, wv.w, o16[4 * k + 3]);
                        }
                    }
                }
            }
            if (ok) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (cbeg + j < cend) a.g_f[((size_t)b * C + cbeg + j) * a.N + n] = o16[j];
            }
        }
        if (a.g_p && ty < 3) {
            float s = 0.0f;
#pragma unroll 4
            for (int h = 0; h < H; ++h) s = __builtin_fmaf(Gs[tx * (H + 1) + h], W1p[h * 4 + ty], s);
            if (ok) a.g_p[ptc * 3 + ty] = s * a.inv_r;
        }
    }
    // 3. 4 x 4 register tiles of dW1[h][c'] over the 64 points (columns cq, cq + tcols, ...: lanes consecutive),
    //    then of dWs[o][c] over the block's queries -- unless no weight takes a gradient (Wpart null: the GAN's feedback
    //    pass runs the classifier with frozen weights; its second stage spent most of this kernel's 300 us here)
    const bool wshare = a.Wpart != nullptr;
    float *__restrict__ wrow = a.Wpart + (size_t)blockIdx.x * ((size_t)H * ldw + (size_t)O * C + O);
    const int tcols = (ldw + 3) / 4, ntile = wshare ? (H / 4) * tcols : 0;
    const int scols = (C + 3) / 4, stiles = O && wshare ? (O / 4) * scols : 0;
    for (int tile = threadIdx.x; tile < ntile + stiles; tile += 256) {
        float t4[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t4[i][j] = 0.0f;
        if (tile < ntile) {
            const int hq = tile / tcols, cq = tile - hq * tcols;
            const int hb = hq * 4;
#pragma unroll 4
            for (int q = 0; q < 64; ++q) {
                float g[4], x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = Gs[q * (H + 1) + hb + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = cq + j * tcols < ldw ? Xs[(cq + j * tcols) * 65 + q] : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t4[i][j] = __builtin_fmaf(g[i], x[j], t4[i][j]);
            }
            if (cq < 3) {        // coordinate column d = cq: minus this block's queries' Hq[q][h] new_p[q][d] / r
#pragma unroll 4
                for (int q = 0; q < nq; ++q) {
                    const float xd = nps[q * 3 + cq];
#pragma unroll
                    for (int i = 0; i < 4; ++i) t4[i][0] = __builtin_fmaf(-Hqs[q * (H + 1) + hb + i], xd, t4[i][0]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cq + j * tcols < ldw) wrow[(size_t)(hb + i) * ldw + cq + j * tcols] = t4[i][j];
        } else {
            const int st = tile - ntile;
            const int oq = st / scols, cq = st - oq * scols;
            const int ob = oq * 4;
#pragma unroll 4
            for (int q = 0; q < nq; ++q) {
                float g[4], x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = gps[q * (O + 1) + ob + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = cq + j * scols < C ? fgs[(cq + j * scols) * (qpb + 1) + q] : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t4[i][j] = __builtin_fmaf(g[i], x[j], t4[i][j]);
            }
            float *__restrict__ srow = wrow + (size_t)H * ldw;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cq + j * scols < C) srow[(size_t)(ob + i) * C + cq + j * scols] = t4[i][j];
        }
    }
    if (O && wshare) {
        for (int o = threadIdx.x; o < O; o += 256) {
            float s = 0.0f;
            for (int q = 0; q < nq; ++q) s += gps[q * (O + 1) + o];
            wrow[(size_t)H * ldw + (size_t)O * C + o] = s;
        }
    }
    // 4. dL/dnew_p = -Hq W1p / r
    if (a.g_q) {
        for (int e = threadIdx.x; e < nq * 3; e += 256) {
            const int q = e / 3, d = e - q * 3;
            float s = 0.0f;
#pragma unroll 4
            for (int h = 0; h < H; ++h) s = __builtin_fmaf(Hqs[q * (H + 1) + h], W1p[h * 4 + d], s);
            a.g_q[(q0 + q) * 3 + d] = -s * a.inv_r;
        }
    }
}

// column sums like wide_colsum (sa_wide.hip), one level, float32 result: out[c] = sum_r part[r][c] in float64
__global__ __launch_bounds__(1024) void wide_colsum_f32_kernel(const float *__restrict__ part, int rows, int ncol,
                                                               float *__restrict__ out) {
    __shared__ double red[32][33];
    const int tx = threadIdx.x & 31, g = threadIdx.x >> 5;                         // 32 columns x 32 row groups
    const int c = blockIdx.x * 32 + tx;
    double s = 0.0;
    if (c < ncol) {
        for (int r0 = g; r0 < rows; r0 += 512) {                                   // 16 independent loads in flight
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int rr = r0 + 32 * u;
                v[u] = part[(size_t)(rr < rows ? rr : rows - 1) * ncol + c];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += r0 + 32 * u < rows ? (double)v[u] : 0.0;
        }
    }
    red[g][tx] = s;
    __syncthreads();
    if (g == 0 && c < ncol) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += red[k][tx];
        out[c] = (float)tot;
    }
}

static bool dense_shape_ok(int H, int O) {
    return (H == 32 || H == 64 || H == 128 || H == 256) && O == 2 * H;
}

}  // namespace apn

using namespace apn;

#define APN_DENSE_DISPATCH(H_, ...)                              \
    switch (H_) {                                                \
    case 32: { constexpr int HPW = 8; __VA_ARGS__; } break;      \
    case 64: { constexpr int HPW = 16; __VA_ARGS__; } break;     \
    case 128: { constexpr int HPW = 32; __VA_ARGS__; } break;    \
    case 256: { constexpr int HPW = 64; __VA_ARGS__; } break;    \
    default: return APN_EINVAL;                                  \
    }

extern "C" int apn_sa_wide_fwd_prep(int b, int c_in, int n, int m, int c_mid, int c_out, float radius, const float *f,
                                    const float *p, const float *new_p, const float *w1, const float *w2, float *U,
                                    float *V, void *w2_image, const int *fq, float *fs, const float *geo, float *part1,
                                    void *stream) {
    if (b <= 0 || c_in <= 0 || n <= 0 || m <= 0 || !dense_shape_ok(c_mid, c_out) || !(radius > 0.0f) || !f || !p ||
        !new_p || !w1 || !w2 || !U || !V || !w2_image || (fs && (!fq || (c_in % 4))) || (part1 && !geo))
        return APN_EINVAL;
    const long long pb = ((long long)b * n + 63) / 64, qb = ((long long)b * m + 63) / 64;
    const int ct = c_out / 32 >= 4 ? 4 : c_out / 32;
    const long long ib = (c_out / 32) * (c_mid / 32);          // 256 words per (column tile, k chunk)
    if (pb + qb + ib > 0x7fffffffLL) return APN_EINVAL;
    APN_DENSE_DISPATCH(c_mid, {
        const size_t lds = ((size_t)64 * (4 * HPW + 1) + (size_t)4 * HPW * (c_in + 3)) * sizeof(float);
        if (lds > 160 * 1024) return APN_EINVAL;
        if (lds > 48 * 1024) {
            if (hipError_t e = hipFuncSetAttribute((const void *)wide_fwd_prep_kernel<HPW>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
                return (int)e;
        }
        hipLaunchKernelGGL((wide_fwd_prep_kernel<HPW>), dim3((unsigned)(pb + qb + ib)), dim3(256), lds,
                           (hipStream_t)stream, b, c_in, n, m, (int)pb, (int)qb, f, p, new_p, w1, 1.0f / radius, U, V,
                           w2, c_out, ct, (uint4 *)w2_image, fq, fs, geo, part1);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// rows of part1 written by apn_sa_wide_fwd_prep: one per 64 points, one per 64 queries
extern "C" int apn_sa_wide_fwd_prep_rows(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return (int)(((long long)b * n + 63) / 64 + ((long long)b * m + 63) / 64);
}

extern "C" int apn_sa_wide_out(int b, int m, int c_out, const float *ysel, const float *pack2, int c_in,
                               const float *fs, const float *ws, const float *bs, int relu, float *out,
                               void *stream) {
    if (b <= 0 || m <= 0 || b > 65535 || c_out <= 0 || (c_out % 64) || !ysel || !pack2 || !out) return APN_EINVAL;
    if (ws && (!fs || c_in <= 0)) return APN_EINVAL;
    hipLaunchKernelGGL(wide_out_kernel, dim3((m + 31) / 32, c_out / 64, b), dim3(256), 0, (hipStream_t)stream, m, c_out,
                       c_in, ysel, pack2, fs, ws, bs, relu, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_bwd_mid(const float *part_s, int rows, const double *sums, int c_mid, int c_out,
                                   const float *pack2, double count, int training, const float *w2, float *d2e2,
                                   float *g_gamma2, float *g_beta2, float *evec, void *z_image, void *stream) {
    if ((!part_s && !sums) || rows < 0 || !dense_shape_ok(c_mid, c_out) || !pack2 || !w2 || !d2e2 || !evec || !z_image)
        return APN_EINVAL;
    if (c_mid > 64) return APN_EINVAL;                              // W2 is held in LDS
    const int ct = c_mid / 32 >= 4 ? 4 : c_mid / 32;
    const int blocks = ((c_out + c_mid) / 32) * (c_mid / (32 * ct));
    const size_t lds = (size_t)(1024 + 2 * c_out) * sizeof(double) +
                       ((size_t)2 * c_out + (size_t)c_out * (c_mid + 1) + (size_t)32 * (32 * ct + 1)) * sizeof(float);
    if (lds > 48 * 1024) {
        if (hipError_t e = hipFuncSetAttribute((const void *)wide_bwd_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds))
            return (int)e;
    }
    hipLaunchKernelGGL(wide_bwd_mid_kernel, dim3(blocks), dim3(1024), lds, (hipStream_t)stream, part_s, rows, sums, c_mid,
                       c_out, ct, pack2, count, training, w2, d2e2, g_gamma2, g_beta2, evec, (uint4 *)z_image);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_bwd_fin(const float *part_t, int rows, const double *sums, int c_mid, int c_out,
                                   const float *pack1, double count, int training, float *cabc, float *g_gamma1,
                                   float *g_beta1, const double *R, const float *d2e2, const float *w2, float *g_w2,
                                   void *stream) {
    if ((!part_t && !sums) || rows < 0 || !dense_shape_ok(c_mid, c_out) || !pack1 || !cabc || !R || !d2e2 || !w2 || !g_w2)
        return APN_EINVAL;
    if (c_mid > 64) return APN_EINVAL;
    const size_t lds = (size_t)(c_mid * c_mid > 1024 ? c_mid * c_mid : 1024) * sizeof(double) + 1024 * sizeof(float);
    hipLaunchKernelGGL(wide_bwd_fin_kernel, dim3((c_out * c_mid + 1023) / 1024 + 1), dim3(1024), lds, (hipStream_t)stream, part_t,
                       rows, sums, c_mid, c_out, pack1, count, training, cabc, g_gamma1, g_beta1, R, d2e2, w2, g_w2);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_point_grads_rows(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return (int)(((long long)b * n + 63) / 64);
}

extern "C" int apn_sa_wide_point_grads_cols(int c_in, int c_mid, int c_skip) {
    return c_mid * (c_in + 3) + c_skip * c_in + c_skip;
}

extern "C" int apn_sa_wide_point_grads(int b, int c_in, int n, int m, int c_mid, float radius, const float *GU,
                                       const int *pcnt_poff, const int *plist, const float *geo, const float *U,
                                       const float *f, const float *p, const float *new_p, const float *HA,
                                       const float *HB, const float *cabc, const float *pack1, const float *w1,
                                       int c_skip, const float *gpre, const int *fq, const float *fs, const float *ws,
                                       float *g_f, float *g_p, float *g_q, float *w_part, void *stream) {
    if (b <= 0 || c_in <= 0 || c_in > 64 || (c_in % 4) || n <= 0 || m <= 0 || (c_mid != 32 && c_mid != 64) || !(radius > 0.0f) ||
        !GU || !pcnt_poff || !plist || !geo || !U || !f || !p || !new_p || !HA || !HB || !cabc || !pack1 || !w1 || !g_f ||
        c_skip < 0 || (c_skip % 4) || (c_skip && (!gpre || !fq || !fs || !ws)))
        return APN_EINVAL;
    const long long npts = (long long)b * n, nqry = (long long)b * m;
    const long long blocks = (npts + 63) / 64;
    if (blocks > 0x7fffffffLL) return APN_EINVAL;
    PointGradArgs a;
    a.B = b; a.C = c_in; a.N = n; a.M = m; a.O = c_skip;
    a.qpb = (int)((nqry + blocks - 1) / blocks);
    a.GU = GU; a.pcnt = pcnt_poff; a.poff = pcnt_poff + npts; a.plist = plist; a.geo = geo;
    a.U = U; a.f = f; a.p = p; a.new_p = new_p; a.HA = HA; a.HB = HB;
    a.cabc = cabc; a.pack1 = pack1; a.w1 = w1; a.inv_r = 1.0f / radius;
    a.gpre = gpre; a.fq = fq; a.fs = fs; a.ws = ws;
    a.g_f = g_f; a.g_p = g_p; a.g_q = g_q; a.Wpart = w_part;
    APN_DENSE_DISPATCH(c_mid, {
        const size_t lds = (size_t)PgLds(4 * HPW, c_in, c_skip, a.qpb).total * sizeof(float);
        if (lds > 160 * 1024) return APN_EINVAL;
        if (lds > 48 * 1024) {
            if (hipError_t e = hipFuncSetAttribute((const void *)wide_point_grads_kernel<HPW>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
                return (int)e;
        }
        hipLaunchKernelGGL((wide_point_grads_kernel<HPW>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_colsum_f32(const float *part, int rows, int ncol, float *out, void *stream) {
    if (rows < 0 || ncol <= 0 || !part || !out) return APN_EINVAL;
    hipLaunchKernelGGL(wide_colsum_f32_kernel, dim3((ncol + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, rows, ncol,
                       out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
