// attention.hip -- multi-head self-attention over all points of a cloud, head_dim 16, without the
// (B, H, M, M) score tensor: SURVEY section 8(f) row 2, the imitator's Anchor_selfattention
// (openpoints/models_adaptpoint/generator_component4_15.py:467-474):
//
//     attn = softmax(q @ k^T / sqrt(16));  out = attn @ v          q, k, v: (B, H, M, 16)
//
// The reference materialises attn twice (scores, softmax) -- 2 x 537 MB at B=32, H=4, M=1024 --
// plus their gradients.  Here one wave owns 32 queries of one (cloud, head) and streams over the
// keys in tiles of 32 with the usual running max / running sum ("flash") recurrence, all on
// v_mfma_f32_32x32x16_bf16 with split (hi + lo) operands, i.e. fp32-grade products:
//
//     S^T tile (32 keys x 32 queries) = K_tile (32 x 16) . Q^T (16 x 32)        one k-step
//     O^T (16 x 32 queries)          += V_tile^T (16 x 32 keys) . P^T (32 x 32)  two k-steps
//
// S is produced TRANSPOSED so that the query sits on the lane (accumulator column) and the keys
// run over the lane's registers: the soft-max statistics of a query are then in-register
// reductions (+ one exchange between the two half-waves), and the accumulator of S^T is already
// in the operand layout of the second product (cf. apn_mfma.h) -- no data crosses lanes.
// head_dim = 16 is exactly one k-step of the first product; the second wastes half a tile
// (rows 16..31 of O^T are never read).
//
// q, k, v arrive as (B, M, H*16) f32, the layout the reference holds them in before its
// reshape/permute (:460-466).  attn_prep writes bf16 hi|lo images once per call: Qs, Ks
// (B,H,M,[hi 16 | lo 16]) with log2(e)/4 folded into Qs, and Vt (B,H,[hi|lo],16,M), V transposed
// with the keys of every 32-tile permuted into accumulator-row order.  Workgroup = 8 waves =
// 256 queries of one (cloud, head); K / V tiles are staged through LDS in chunks of 256 keys
// and shared by the eight waves.
#include "apn_common.h"
#include "apn_mfma.h"

namespace apn {

constexpr int AT_D = 16;           // head dim
constexpr int AT_WAVES = 8;        // waves per workgroup = query tiles per workgroup
constexpr int AT_CHUNK = 256;      // keys staged per LDS pass
constexpr float AT_LOG2E = 1.4426950408889634f;

// position p of a 32-tile <-> key acc_row(8 s + j, h) with p = 16 s + 8 h + j
__device__ __forceinline__ int at_perm_key(int p) {
    const int s = p >> 4, h = (p >> 3) & 1, j = p & 7;
    return acc_row(8 * s + j, h);
}

// bf16 hi|lo operand images of one (B, M, H*16) f32 tensor `src`, one workgroup (a wave) per
// (32-point tile, head, cloud); M % 32 == 0:
//   rows  (B,H,M,[hi 16 | lo 16])  of src * row_scale      -- a point is an operand ROW / COLUMN
//   trans (B,H,[hi|lo],16,M)       of src * trans_scale    -- src^T with the points of every
//          32-tile permuted into accumulator-row order: a point is a CONTRACTION index
//   delta (B,H,M) = sum_d src[d] * other[d]                 -- backward: rowsum(dO * O)
// Any output may be null.
__global__ __launch_bounds__(64) void attn_image_kernel(int m, int heads, const float *__restrict__ src,
                                                        float row_scale, float trans_scale,
                                                        __bf16 *__restrict__ rows,
                                                        __bf16 *__restrict__ trans,
                                                        const float *__restrict__ other,
                                                        float *__restrict__ delta) {
    __shared__ float sv[32][AT_D + 1];
    const int tile = blockIdx.x, head = blockIdx.y, cloud = blockIdx.z, c = heads * AT_D;
    const int lane = threadIdx.x;
    const size_t bh = (size_t)cloud * heads + head;
    // lane -> (point = lane >> 1, half = lane & 1): 8 dims each
    const int row = lane >> 1, d0 = (lane & 1) * 8;
    const size_t at = ((size_t)cloud * m + tile * 32 + row) * c + head * AT_D + d0;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = src[at + j];
    if (rows) {
        __bf16 *dst = rows + (bh * m + tile * 32 + row) * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = x[j] * row_scale;
            const __bf16 hi = (__bf16)y;
            dst[d0 + j] = hi;
            dst[16 + d0 + j] = (__bf16)(y - (float)hi);
        }
    }
    if (delta) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j] * other[at + j];
        s += __shfl_xor(s, 1);
        if ((lane & 1) == 0) delta[bh * m + tile * 32 + row] = s;
    }
    if (trans) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[row][d0 + j] = x[j] * trans_scale;
        __syncthreads();
        // lane -> (d = lane >> 2, 8 positions p0 = (lane & 3) * 8 of the permuted tile)
        const int d = lane >> 2, p0 = (lane & 3) * 8;
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = sv[at_perm_key(p0 + j)][d];
            hi[j] = (__bf16)y;
            lo[j] = (__bf16)(y - (float)hi[j]);
        }
        __bf16 *tb = trans + bh * 2 * AT_D * m;
        *reinterpret_cast<bf16x8 *>(tb + (size_t)d * m + tile * 32 + p0) = hi;
        *reinterpret_cast<bf16x8 *>(tb + (size_t)(AT_D + d) * m + tile * 32 + p0) = lo;
    }
}

// LDS images of a chunk of keys: Ks rows padded to 80 bytes, Vt rows [part][d] of CHUNK bf16
// padded by 8.
constexpr int AT_KROW = 40;                 // bf16 per staged K row (32 + 8 pad)
constexpr int AT_VROW = AT_CHUNK + 8;       // bf16 per staged Vt row

// rows image chunk -> LDS (`n` rows of 64 bytes, padded to AT_KROW)
__device__ __forceinline__ void attn_stage_rows(int n, const __bf16 *__restrict__ src, __bf16 *dst) {
    for (int e = threadIdx.x; e < n * 4; e += AT_WAVES * 64) {
        const int row = e >> 2, piece = e & 3;
        *reinterpret_cast<uint4 *>(dst + row * AT_KROW + piece * 8) =
            *reinterpret_cast<const uint4 *>(src + (size_t)row * 32 + piece * 8);
    }
}

// trans image chunk -> LDS (32 rows (part, d) of `n` bf16, row stride m in memory, AT_VROW in LDS)
__device__ __forceinline__ void attn_stage_trans(int n, int m, const __bf16 *__restrict__ src, __bf16 *dst) {
    const int ppr = n >> 3;
    for (int e = threadIdx.x; e < 32 * ppr; e += AT_WAVES * 64) {
        const int row = e / ppr, piece = e - row * ppr;
        *reinterpret_cast<uint4 *>(dst + row * AT_VROW + piece * 8) =
            *reinterpret_cast<const uint4 *>(src + (size_t)row * m + piece * 8);
    }
}

__device__ __forceinline__ Frag<2> attn_row_frag(const __bf16 *row, int h) {
    Frag<2> f;
    f.p[0] = *reinterpret_cast<const bf16x8 *>(row + 8 * h);
    f.p[1] = *reinterpret_cast<const bf16x8 *>(row + 16 + 8 * h);
    return f;
}

// A operand whose contraction index runs over a staged, permuted 32-tile: rows d = r & 15
__device__ __forceinline__ Frag<2> attn_trans_frag(const __bf16 *img, int r, int h, int t, int st) {
    Frag<2> f;
    const __bf16 *row = img + (r & 15) * AT_VROW + t * 32 + 16 * st + 8 * h;
    f.p[0] = *reinterpret_cast<const bf16x8 *>(row);
    f.p[1] = *reinterpret_cast<const bf16x8 *>(row + AT_D * AT_VROW);
    return f;
}

// out (B,M,H*16) f32; lse (B,H,M) f32 = running max + log2(running sum), in log2 units.
__global__ __launch_bounds__(AT_WAVES * 64) void attn_fwd_kernel(int m, int heads,
                                                                 const __bf16 *__restrict__ Qs,
                                                                 const __bf16 *__restrict__ Ks,
                                                                 const __bf16 *__restrict__ Vt,
                                                                 float *__restrict__ out,
                                                                 float *__restrict__ lse) {
    __shared__ __attribute__((aligned(16))) __bf16 sK[AT_CHUNK * AT_KROW];
    __shared__ __attribute__((aligned(16))) __bf16 sV[32 * AT_VROW];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int head = blockIdx.y, cloud = blockIdx.z;
    const size_t bh = (size_t)cloud * heads + head;
    const int q0 = (blockIdx.x * AT_WAVES + wave) * 32;      // this wave's query tile
    const bool live = q0 < m;                                // whole tiles only (m % 32 == 0)
    // B operand of S^T = K . Q^T: lane (query r, h) holds d = 8h .. 8h+7
    // (idle waves of a ragged last workgroup run the loop on query 0's row and store nothing)
    const Frag<2> qf = attn_row_frag(Qs + (bh * m + (live ? q0 + r : 0)) * 32, h);
    f32x16 o = {0};
    float mx = -INFINITY, lsum = 0.0f;
    const __bf16 *Ks_bh = Ks + bh * m * 32, *Vt_bh = Vt + bh * 2 * AT_D * m;
    for (int k0 = 0; k0 < m; k0 += AT_CHUNK) {
        __syncthreads();
        const int keys = m - k0 < AT_CHUNK ? m - k0 : AT_CHUNK, tiles = keys / 32;
        attn_stage_rows(keys, Ks_bh + (size_t)k0 * 32, sK);
        attn_stage_trans(keys, m, Vt_bh + k0, sV);
        __syncthreads();
        for (int t = 0; t < tiles; ++t) {
            // A operand: lane (key r, h) holds d = 8h .. 8h+7 of key t*32 + r
            const Frag<2> kf = attn_row_frag(sK + (t * 32 + r) * AT_KROW, h);
            f32x16 s = {0};
            s = mfma<2>(kf, qf, s);                      // s[i]: key acc_row(i, h), query r (log2 units)
            float tmax = s[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) tmax = __builtin_fmaxf(tmax, s[i]);
            tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, 32));
            const float mnew = __builtin_fmaxf(mx, tmax);
            const float alpha = __builtin_amdgcn_exp2f(mx - mnew);     // exp2(-inf) = 0 on the first tile
            float psum = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[i] = __builtin_amdgcn_exp2f(s[i] - mnew);
                psum += s[i];
            }
            lsum = lsum * alpha + psum;
            mx = mnew;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= alpha;
            // O^T += V^T . P^T, k = keys in accumulator-row order: B operand element j of step st
            // is register 8 st + j; A operand from the permuted Vt image
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                o = mfma<2>(attn_trans_frag(sV, r, h, t, st), pack8<2>(s, 8 * st), o);
            }
        }
    }
    if (!live) return;
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.0f / lsum;
    // o[i], i < 8: d = acc_row(i, h) in {0..3, 8..11} + 4h, query r
    float *orow = out + ((size_t)cloud * m + q0 + r) * (heads * AT_D) + head * AT_D;
    *reinterpret_cast<float4 *>(orow + 4 * h) = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    *reinterpret_cast<float4 *>(orow + 8 + 4 * h) = make_float4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
    if (h == 0) lse[bh * m + q0 + r] = mx + __builtin_amdgcn_logf(lsum);   // v_log_f32 = log2
}

// ---------------------------------------------------------------------------
// Backward.  With P = softmax probabilities (recomputed from the saved log-sum-exp),
// delta_i = sum_d dO_i[d] O_i[d]:
//     dV_j = sum_i P_ij dO_i            dP_ij = dO_i . V_j
//     dS_ij = P_ij (dP_ij - delta_i)    dQ_i = sum_j dS_ij K_j / 4     dK_j = sum_i dS_ij Q_i / 4
// Two kernels, each the forward's shape with the roles of queries and keys chosen so that the
// contraction index of every second product is the accumulator-row index of the first:
//   attn_bwd_kv: a wave owns 32 KEYS, streams over the queries;  S = Q K^T (queries on rows)
//   attn_bwd_q : a wave owns 32 QUERIES, streams over the keys;  S^T = K Q^T (as the forward)
// ---------------------------------------------------------------------------

// dk, dv (B,M,H*16) f32.
__global__ __launch_bounds__(AT_WAVES * 64) void attn_bwd_kv_kernel(
    int m, int heads, const __bf16 *__restrict__ Qs, const __bf16 *__restrict__ Qt4,
    const __bf16 *__restrict__ Ks, const __bf16 *__restrict__ Vs, const __bf16 *__restrict__ dOs,
    const __bf16 *__restrict__ dOt, const float *__restrict__ lse, const float *__restrict__ delta,
    float *__restrict__ dk, float *__restrict__ dv) {
    __shared__ __attribute__((aligned(16))) __bf16 sQ[AT_CHUNK * AT_KROW];    // Qs rows of the chunk
    __shared__ __attribute__((aligned(16))) __bf16 sG[AT_CHUNK * AT_KROW];    // dOs rows
    __shared__ __attribute__((aligned(16))) __bf16 sQt[32 * AT_VROW];         // Qt4
    __shared__ __attribute__((aligned(16))) __bf16 sGt[32 * AT_VROW];         // dOt
    __shared__ __attribute__((aligned(16))) float sL[AT_CHUNK], sD[AT_CHUNK]; // lse, delta of the chunk
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int head = blockIdx.y, cloud = blockIdx.z;
    const size_t bh = (size_t)cloud * heads + head;
    const int j0 = (blockIdx.x * AT_WAVES + wave) * 32;      // this wave's key tile
    const bool live = j0 < m;
    const int jr = live ? j0 + r : 0;
    // B operands (lane = key column): K^T for S, V^T for dP
    const Frag<2> kf = attn_row_frag(Ks + (bh * m + jr) * 32, h);
    const Frag<2> vf = attn_row_frag(Vs + (bh * m + jr) * 32, h);
    f32x16 dvt = {0}, dkt = {0};                             // dV^T, dK^T: rows d, columns keys
    const size_t img = bh * (size_t)m * 32;
    for (int i0 = 0; i0 < m; i0 += AT_CHUNK) {
        __syncthreads();
        const int qs = m - i0 < AT_CHUNK ? m - i0 : AT_CHUNK, tiles = qs / 32;
        attn_stage_rows(qs, Qs + img + (size_t)i0 * 32, sQ);
        attn_stage_rows(qs, dOs + img + (size_t)i0 * 32, sG);
        attn_stage_trans(qs, m, Qt4 + img + i0, sQt);
        attn_stage_trans(qs, m, dOt + img + i0, sGt);
        for (int e = threadIdx.x; e < qs; e += AT_WAVES * 64) {
            sL[e] = lse[bh * m + i0 + e];
            sD[e] = delta[bh * m + i0 + e];
        }
        __syncthreads();
        for (int t = 0; t < tiles; ++t) {
            // S = Q K^T, dP = dO V^T: rows = queries acc_row(i, h), columns = keys (lane)
            f32x16 s = {0}, dp = {0};
            s = mfma<2>(attn_row_frag(sQ + (t * 32 + r) * AT_KROW, h), kf, s);
            dp = mfma<2>(attn_row_frag(sG + (t * 32 + r) * AT_KROW, h), vf, dp);
            // per-query statistics: query acc_row(i, h) = 4h + (i & 3) + 8 (i >> 2)
            float lq[16], dq_[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 l4 = *reinterpret_cast<const float4 *>(sL + t * 32 + 8 * g4 + 4 * h);
                const float4 d4 = *reinterpret_cast<const float4 *>(sD + t * 32 + 8 * g4 + 4 * h);
                lq[4 * g4] = l4.x; lq[4 * g4 + 1] = l4.y; lq[4 * g4 + 2] = l4.z; lq[4 * g4 + 3] = l4.w;
                dq_[4 * g4] = d4.x; dq_[4 * g4 + 1] = d4.y; dq_[4 * g4 + 2] = d4.z; dq_[4 * g4 + 3] = d4.w;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pij = __builtin_amdgcn_exp2f(s[i] - lq[i]);
                s[i] = pij;                                   // P
                dp[i] = pij * (dp[i] - dq_[i]);               // dS
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                dvt = mfma<2>(attn_trans_frag(sGt, r, h, t, st), pack8<2>(s, 8 * st), dvt);
                dkt = mfma<2>(attn_trans_frag(sQt, r, h, t, st), pack8<2>(dp, 8 * st), dkt);
            }
        }
    }
    if (!live) return;
    const size_t o = ((size_t)cloud * m + j0 + r) * (heads * AT_D) + head * AT_D;
    *reinterpret_cast<float4 *>(dv + o + 4 * h) = make_float4(dvt[0], dvt[1], dvt[2], dvt[3]);
    *reinterpret_cast<float4 *>(dv + o + 8 + 4 * h) = make_float4(dvt[4], dvt[5], dvt[6], dvt[7]);
    *reinterpret_cast<float4 *>(dk + o + 4 * h) = make_float4(dkt[0], dkt[1], dkt[2], dkt[3]);
    *reinterpret_cast<float4 *>(dk + o + 8 + 4 * h) = make_float4(dkt[4], dkt[5], dkt[6], dkt[7]);
}

// dq (B,M,H*16) f32.
__global__ __launch_bounds__(AT_WAVES * 64) void attn_bwd_q_kernel(
    int m, int heads, const __bf16 *__restrict__ Qs, const __bf16 *__restrict__ Ks,
    const __bf16 *__restrict__ Kt4, const __bf16 *__restrict__ Vs, const __bf16 *__restrict__ dOs,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dq) {
    __shared__ __attribute__((aligned(16))) __bf16 sK[AT_CHUNK * AT_KROW];    // Ks rows of the chunk
    __shared__ __attribute__((aligned(16))) __bf16 sVr[AT_CHUNK * AT_KROW];   // Vs rows
    __shared__ __attribute__((aligned(16))) __bf16 sKt[32 * AT_VROW];         // Kt4
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int head = blockIdx.y, cloud = blockIdx.z;
    const size_t bh = (size_t)cloud * heads + head;
    const int q0 = (blockIdx.x * AT_WAVES + wave) * 32;
    const bool live = q0 < m;
    const int qr = live ? q0 + r : 0;
    // B operands (lane = query column): Q^T for S^T, dO^T for dP^T
    const Frag<2> qf = attn_row_frag(Qs + (bh * m + qr) * 32, h);
    const Frag<2> gf = attn_row_frag(dOs + (bh * m + qr) * 32, h);
    const float lq = lse[bh * m + qr], dl = delta[bh * m + qr];
    f32x16 dqt = {0};
    const size_t img = bh * (size_t)m * 32;
    for (int k0 = 0; k0 < m; k0 += AT_CHUNK) {
        __syncthreads();
        const int keys = m - k0 < AT_CHUNK ? m - k0 : AT_CHUNK, tiles = keys / 32;
        attn_stage_rows(keys, Ks + img + (size_t)k0 * 32, sK);
        attn_stage_rows(keys, Vs + img + (size_t)k0 * 32, sVr);
        attn_stage_trans(keys, m, Kt4 + img + k0, sKt);
        __syncthreads();
        for (int t = 0; t < tiles; ++t) {
            // S^T = K Q^T, dP^T = V dO^T: rows = keys acc_row(i, h), columns = queries (lane)
            f32x16 s = {0}, dp = {0};
            s = mfma<2>(attn_row_frag(sK + (t * 32 + r) * AT_KROW, h), qf, s);
            dp = mfma<2>(attn_row_frag(sVr + (t * 32 + r) * AT_KROW, h), gf, dp);
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = __builtin_amdgcn_exp2f(s[i] - lq) * (dp[i] - dl);   // dS^T
#pragma unroll
            for (int st = 0; st < 2; ++st)
                dqt = mfma<2>(attn_trans_frag(sKt, r, h, t, st), pack8<2>(dp, 8 * st), dqt);
        }
    }
    if (!live) return;
    const size_t o = ((size_t)cloud * m + q0 + r) * (heads * AT_D) + head * AT_D;
    *reinterpret_cast<float4 *>(dq + o + 4 * h) = make_float4(dqt[0], dqt[1], dqt[2], dqt[3]);
    *reinterpret_cast<float4 *>(dq + o + 8 + 4 * h) = make_float4(dqt[4], dqt[5], dqt[6], dqt[7]);
}

static int attn_check(int b, int m, int heads) {
    if (b <= 0 || m <= 0 || heads <= 0 || (m & 31) || b > 65535 || heads > 65535) return APN_EINVAL;
    return APN_OK;
}

}  // namespace apn

// Operand images: six of b*heads*m*32 bf16 (64 bytes per point and head) each, in `images`:
//   [0] Qs rows (q * log2(e)/4)   [1] Ks rows   [2] Vt trans   [3] Vs rows   [4] Qt4 trans (q/4)
//   [5] Kt4 trans (k/4);  the forward uses 0..2, the backward all six.
namespace apn {

// ------------------------------------------------------------------------------------------------------------------
// The same attention for FEW points (m <= 32: the imitator's 4-anchor head, generator_component4_15.py:572): one wave
// per (cloud, head), lane i = query i, everything in float32 registers -- the keys and values of the head are 2 x m x 16
// numbers.  Forward saves nothing but `out`; the backward recomputes the probabilities.  Sums over the queries (dK, dV)
// are wave reductions in a fixed order: bit-reproducible.
// ------------------------------------------------------------------------------------------------------------------
constexpr int AT_SMALL_MAX = 32;

__device__ __forceinline__ float at_wave_sum32(float v) {         // over lanes 0..31 (both halves hold their own sum)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <bool BWD>
__global__ __launch_bounds__(64) void attn_small_kernel(int m, int heads, const float *__restrict__ q,
                                                        const float *__restrict__ k, const float *__restrict__ v,
                                                        const float *__restrict__ g_out, float *__restrict__ out,
                                                        float *__restrict__ dq, float *__restrict__ dk,
                                                        float *__restrict__ dv) {
    __shared__ float sk[AT_SMALL_MAX][AT_D + 1], sv[AT_SMALL_MAX][AT_D + 1];
    const int head = blockIdx.x, cloud = blockIdx.y, lane = threadIdx.x, c = heads * AT_D;
    const size_t base = (size_t)cloud * m * c + head * AT_D;
    for (int e = lane; e < m * AT_D; e += 64) {
        const int j = e / AT_D, d = e % AT_D;
        sk[j][d] = k[base + (size_t)j * c + d];
        sv[j][d] = v[base + (size_t)j * c + d];
    }
    const int i = lane < m ? lane : m - 1;                         // (lanes past m compute a copy of the last query, dropped)
    float qi[AT_D], go[AT_D];
#pragma unroll
    for (int d = 0; d < AT_D; ++d) {
        qi[d] = q[base + (size_t)i * c + d];
        go[d] = BWD ? g_out[base + (size_t)i * c + d] : 0.0f;
    }
    __syncthreads();
    // scores / sqrt(16), soft-max over the keys (generator_component4_15.py:468-470)
    float p[AT_SMALL_MAX];
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < AT_SMALL_MAX; ++j) {
        float sc = 0.0f;
        if (j < m) {
#pragma unroll
            for (int d = 0; d < AT_D; ++d) sc = __builtin_fmaf(qi[d], sk[j][d], sc);
            sc *= 0.25f;
            mx = __builtin_fmaxf(mx, sc);
        }
        p[j] = sc;
    }
    float z = 0.0f;
#pragma unroll
    for (int j = 0; j < AT_SMALL_MAX; ++j) {
        p[j] = j < m ? __expf(p[j] - mx) : 0.0f;
        z += p[j];
    }
    const float rz = 1.0f / z;
#pragma unroll
    for (int j = 0; j < AT_SMALL_MAX; ++j) p[j] *= rz;
    if (!BWD) {
        float o[AT_D];
#pragma unroll
        for (int d = 0; d < AT_D; ++d) o[d] = 0.0f;
#pragma unroll
        for (int j = 0; j < AT_SMALL_MAX; ++j)
            if (j < m) {
#pragma unroll
                for (int d = 0; d < AT_D; ++d) o[d] = __builtin_fmaf(p[j], sv[j][d], o[d]);
            }
        if (lane < m) {
#pragma unroll
            for (int d = 0; d < AT_D; ++d) out[base + (size_t)lane * c + d] = o[d];
        }
        return;
    }
    // backward: dP = dO V^T, dS = P (dP - rowsum(P dP)), dQ = dS K / 4, dK = dS^T Q / 4, dV = P^T dO
    const bool live = lane < m;
    float ds[AT_SMALL_MAX], dot = 0.0f;
#pragma unroll
    for (int j = 0; j < AT_SMALL_MAX; ++j) {
        float dp = 0.0f;
        if (j < m) {
#pragma unroll
            for (int d = 0; d < AT_D; ++d) dp = __builtin_fmaf(go[d], sv[j][d], dp);
        }
        ds[j] = dp;
        dot = __builtin_fmaf(p[j], dp, dot);
    }
    float gq[AT_D];
#pragma unroll
    for (int d = 0; d < AT_D; ++d) gq[d] = 0.0f;
#pragma unroll
    for (int j = 0; j < AT_SMALL_MAX; ++j) {
        ds[j] = j < m ? p[j] * (ds[j] - dot) * 0.25f : 0.0f;
        if (j < m) {
#pragma unroll
            for (int d = 0; d < AT_D; ++d) gq[d] = __builtin_fmaf(ds[j], sk[j][d], gq[d]);
        }
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < AT_D; ++d) dq[base + (size_t)lane * c + d] = gq[d];
    }
    // column sums over the queries: lanes 0..31 carry the queries (lanes >= m and the upper half add zero)
    const bool mine = lane < m && lane < 32;
#pragma unroll
    for (int j = 0; j < AT_SMALL_MAX; ++j) {
        if (j < m) {                                             // wave-uniform
#pragma unroll
            for (int d = 0; d < AT_D; ++d) {
                const float a = at_wave_sum32(mine ? ds[j] * qi[d] : 0.0f);
                const float b = at_wave_sum32(mine ? p[j] * go[d] : 0.0f);
                if (lane == 0) {
                    dk[base + (size_t)j * c + d] = a;
                    dv[base + (size_t)j * c + d] = b;
                }
            }
        }
    }
}

}  // namespace apn

// Few points (0 < m <= 32): out, or (g_out given) dq, dk, dv; q, k, v, g_out, out, d* are (B, m, heads*16) f32.
extern "C" int apn_attention_small_max(void) { return apn::AT_SMALL_MAX; }

extern "C" int apn_attention_small_fwd(int b, int m, int heads, const float *q, const float *k, const float *v,
                                       float *out, void *stream) {
    using namespace apn;
    if (b <= 0 || m <= 0 || m > AT_SMALL_MAX || heads <= 0 || heads > 65535 || b > 65535 || !q || !k || !v || !out)
        return APN_EINVAL;
    hipLaunchKernelGGL(attn_small_kernel<false>, dim3(heads, b), dim3(64), 0, (hipStream_t)stream, m, heads, q, k, v,
                       (const float *)nullptr, out, (float *)nullptr, (float *)nullptr, (float *)nullptr);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_attention_small_bwd(int b, int m, int heads, const float *q, const float *k, const float *v,
                                       const float *g_out, float *dq, float *dk, float *dv, void *stream) {
    using namespace apn;
    if (b <= 0 || m <= 0 || m > AT_SMALL_MAX || heads <= 0 || heads > 65535 || b > 65535 || !q || !k || !v || !g_out ||
        !dq || !dk || !dv)
        return APN_EINVAL;
    hipLaunchKernelGGL(attn_small_kernel<true>, dim3(heads, b), dim3(64), 0, (hipStream_t)stream, m, heads, q, k, v, g_out,
                       (float *)nullptr, dq, dk, dv);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_attention_prep(int b, int m, int heads, const float *q, const float *k,
                                  const float *v, void *images, int for_backward, void *stream) {
    using namespace apn;
    if (int e = attn_check(b, m, heads)) return e;
    if (!q || !k || !v || !images) return APN_EINVAL;
    const size_t n = (size_t)b * heads * m * 32;
    __bf16 *im = (__bf16 *)images;
    const dim3 grid(m / 32, heads, b);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_image_kernel, grid, dim3(64), 0, st, m, heads, q, 0.25f * AT_LOG2E, 0.25f,
                       im, for_backward ? im + 4 * n : nullptr, (const float *)nullptr, (float *)nullptr);
    hipLaunchKernelGGL(attn_image_kernel, grid, dim3(64), 0, st, m, heads, k, 1.0f, 0.25f, im + n,
                       for_backward ? im + 5 * n : nullptr, (const float *)nullptr, (float *)nullptr);
    hipLaunchKernelGGL(attn_image_kernel, grid, dim3(64), 0, st, m, heads, v, 1.0f, 1.0f,
                       for_backward ? im + 3 * n : nullptr, im + 2 * n, (const float *)nullptr,
                       (float *)nullptr);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_attention_fwd(int b, int m, int heads, const void *images, float *out, float *lse,
                                 void *stream) {
    using namespace apn;
    if (int e = attn_check(b, m, heads)) return e;
    if (!images || !out || !lse) return APN_EINVAL;
    const size_t n = (size_t)b * heads * m * 32;
    const __bf16 *im = (const __bf16 *)images;
    const int per_wg = AT_WAVES * 32;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((m + per_wg - 1) / per_wg, heads, b), dim3(AT_WAVES * 64), 0,
                       (hipStream_t)stream, m, heads, im, im + n, im + 2 * n, out, lse);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// g_out (B,M,H*16) = dL/d out; out, lse from the forward; images from apn_attention_prep(...,
// for_backward = 1); scratch: 2 * b*heads*m*32 bf16 (images of g_out) + b*heads*m floats (delta).
extern "C" int apn_attention_bwd(int b, int m, int heads, const void *images, const float *out,
                                 const float *lse, const float *g_out, void *scratch, float *dq,
                                 float *dk, float *dv, void *stream) {
    using namespace apn;
    if (int e = attn_check(b, m, heads)) return e;
    if (!images || !out || !lse || !g_out || !scratch || !dq || !dk || !dv) return APN_EINVAL;
    const size_t n = (size_t)b * heads * m * 32;
    const __bf16 *im = (const __bf16 *)images;
    __bf16 *gs = (__bf16 *)scratch, *gt = gs + n;
    float *delta = (float *)(gt + n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_image_kernel, dim3(m / 32, heads, b), dim3(64), 0, st, m, heads, g_out, 1.0f,
                       1.0f, gs, gt, out, delta);
    const int per_wg = AT_WAVES * 32;
    const dim3 grid((m + per_wg - 1) / per_wg, heads, b);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(AT_WAVES * 64), 0, st, m, heads, im, im + 4 * n,
                       im + n, im + 3 * n, gs, gt, lse, delta, dk, dv);
    hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(AT_WAVES * 64), 0, st, m, heads, im, im + n,
                       im + 5 * n, im + 3 * n, gs, lse, delta, dq);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
