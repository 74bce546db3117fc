// sa_wide.hip -- the grouped shared-MLP of a PointNeXt set-abstraction block for ANY of the
// model's widths (C_mid = H in {32, 64, 128, 256}, C_out = O = 2H, K = 32 neighbours), gfx950.
//
// Reference block (openpoints/models/backbone/pointnext.py:157-168 over QueryAndGroup,
// openpoints/models/layers/group.py:235-255): y1 = Conv2d(cat[dp, f[idx]]), a1 = ReLU(BN(y1)),
// y2 = Conv2d(a1), out = max_K BN(y2).  Two observations shape these kernels:
//
//  1. conv1 is linear and its input is a gather, so it commutes with the gather:
//         y1[q,k] = W1f f[idx[q,k]] + W1p (p[idx[q,k]] - new_p[q]) / r = U[idx[q,k]] - V[q]
//     with ONE row per point  U[n] = W1f f[n] + W1p p[n] / r   (B,N,H)   and one per query
//     V[q] = W1p new_p[q] / r (B,M,H).  U and V are plain dense products over points (the caller
//     forms them; 16x fewer flops than the per-position convolution at M = N/2, K = 32) and y1 is
//     an fp32 difference of two gathered rows -- no MFMA, no operand rounding before BatchNorm-1.
//  2. after the ReLU the chain is per position: y2 = a1 W2^T is the one contraction that must run
//     over all B*M*K positions.  It runs on v_mfma_f32_32x32x16_bf16 with split (hi + lo)
//     operands, one WAVE per query (32 positions = one 32-row tile): the wave builds its A
//     fragments in registers straight from the gathered rows (lane = position, 8 consecutive
//     channels), the B fragments (weights) are shared by the workgroup's waves through LDS in
//     fragment order (conflict-free ds_read_b128), and the accumulator has lane = channel,
//     register = position, so BatchNorm statistics and the max over K are in-register
//     reductions plus one exchange between the lane halves.  No (.,M,K) tensor exists.
//
// Kernels (all: grid-stride over query tiles, 4 waves per workgroup, statistics leave as one
// partial row per workgroup, summed in float64 by the caller: deterministic):
//   wide_stats1     sum / sum of squares of y1                              (forward pass 1)
//   wide_fwd_main   a1 -> y2 -> {sum, sumsq} of y2, ext_K y2 + its slot     (forward pass 2)
//   wide_bwd_main   dL/da1 = S W2 + a1 Qm + evec -> g_u -> T1, T2, per-point sums A (atomics),
//                   per-query sums HA, HB                                   (backward)
//   wide_wgrad      R = [S^T ; a1^T] a1, sum a1  (the sparse part of dL/dW2 and the Gram matrix)
// The weight operands arrive as "B images": bf16 hi/lo parts laid out in MFMA fragment order
// by the caller (adaptpoint_amd/fused_wide.py::mfma_b_image), one contiguous block per
// (column block, 32-deep k chunk).
#include "apn_common.h"
#include "apn_mfma.h"

namespace apn {

struct WideArgs {
    int ntiles;          // B * M query tiles (32 positions each)
    int n, m;            // support points / queries per cloud
    const float *U;      // (B,N,H)
    const float *V;      // (B,M,H)
    const int *idx;      // (B,M,32)
};

constexpr int WIDE_WAVES = 4;

__device__ __forceinline__ float shfl_xor32(float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane_id() ^ 32) << 2, __float_as_int(v)));
}
__device__ __forceinline__ int shfl_xor32i(int v) {
    return __builtin_amdgcn_ds_bpermute((lane_id() ^ 32) << 2, v);
}

// ------------------------------------------------------------------------------------------
// pass 1: part[block][2H] = {sum[H], sumsq[H]} of y1 = U[idx] - V over the block's tiles
// ------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void wide_stats1_kernel(WideArgs a, float *__restrict__ part) {
    constexpr int CPL = H >= 64 ? H / 64 : 1;      // channels per lane
    constexpr int PP = H >= 64 ? 1 : 2;            // positions per wave pass
    __shared__ float red[WIDE_WAVES][2 * H];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    float s[CPL], ss[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) s[j] = ss[j] = 0.0f;
    for (int tile = blockIdx.x * WIDE_WAVES + w; tile < a.ntiles; tile += gridDim.x * WIDE_WAVES) {
        const int cloud = tile / a.m;
        const int *__restrict__ ip = a.idx + (size_t)tile * 32;
        const float *__restrict__ ub = a.U + (size_t)cloud * a.n * H;
        float vq[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) vq[j] = a.V[(size_t)tile * H + (PP == 2 ? (lane & 31) : lane + 64 * j)];
#pragma unroll 8
        for (int p0 = 0; p0 < 32; p0 += PP) {
            const int nn = ip[p0 + (PP == 2 ? (lane >> 5) : 0)];
            const float *__restrict__ row = ub + (size_t)nn * H;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const float y = row[PP == 2 ? (lane & 31) : lane + 64 * j] - vq[j];
                s[j] += y;
                ss[j] = __builtin_fmaf(y, y, ss[j]);
            }
        }
    }
    if (PP == 2) {
        s[0] += shfl_xor32(s[0]);
        ss[0] += shfl_xor32(ss[0]);
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int ch = PP == 2 ? (lane & 31) : lane + 64 * j;
        if (PP == 1 || lane < 32) {
            red[w][ch] = s[j];
            red[w][H + ch] = ss[j];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * H; e += 256)
        part[(size_t)blockIdx.x * 2 * H + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// ------------------------------------------------------------------------------------------
// Shared pieces of the MFMA kernels
// ------------------------------------------------------------------------------------------
// A fragment of a1 = relu(scale1 * (U[n] - V[q]) + shift1): lane (pos r, h), 8 channels from ch0.
__device__ __forceinline__ Frag<2> a1_frag(const float *__restrict__ row, const float *__restrict__ vq,
                                           const float *__restrict__ pack1, int H, int ch0) {
    const float4 u0 = *reinterpret_cast<const float4 *>(row + ch0);
    const float4 u1 = *reinterpret_cast<const float4 *>(row + ch0 + 4);
    const float4 v0 = *reinterpret_cast<const float4 *>(vq + ch0);
    const float4 v1 = *reinterpret_cast<const float4 *>(vq + ch0 + 4);
    const float4 c0 = *reinterpret_cast<const float4 *>(pack1 + ch0);
    const float4 c1 = *reinterpret_cast<const float4 *>(pack1 + ch0 + 4);
    const float4 d0 = *reinterpret_cast<const float4 *>(pack1 + H + ch0);
    const float4 d1 = *reinterpret_cast<const float4 *>(pack1 + H + ch0 + 4);
    float t[8];
    t[0] = __builtin_fmaf(u0.x - v0.x, c0.x, d0.x); t[1] = __builtin_fmaf(u0.y - v0.y, c0.y, d0.y);
    t[2] = __builtin_fmaf(u0.z - v0.z, c0.z, d0.z); t[3] = __builtin_fmaf(u0.w - v0.w, c0.w, d0.w);
    t[4] = __builtin_fmaf(u1.x - v1.x, c1.x, d1.x); t[5] = __builtin_fmaf(u1.y - v1.y, c1.y, d1.y);
    t[6] = __builtin_fmaf(u1.z - v1.z, c1.z, d1.z); t[7] = __builtin_fmaf(u1.w - v1.w, c1.w, d1.w);
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = t[e] > 0.0f ? t[e] : 0.0f;
    return make_frag<2>(t);
}

// One chunk of a B image = CT column tiles x 2 k-steps x {hi, lo} x 64 lanes x 16 bytes.
template <int CT>
struct Chunk {
    static constexpr int WORDS = CT * 2 * 2 * 64;       // uint4 words
};

// Copy chunk `ci` of the image into an LDS slot (all 256 threads; contiguous, coalesced).
template <int CT>
__device__ __forceinline__ void stage_chunk(const uint4 *__restrict__ img, int ci, uint4 *slot) {
    const uint4 *__restrict__ src = img + (size_t)ci * Chunk<CT>::WORDS;
#pragma unroll
    for (int i = 0; i < Chunk<CT>::WORDS / 256; ++i) slot[threadIdx.x + 256 * i] = src[threadIdx.x + 256 * i];
}

// Streaming form of the same copy, split around the MFMAs of the current chunk: the global loads of
// chunk ci+1 are issued before them (registers), the LDS stores after them.
template <int CT>
struct ChunkRegs {
    uint4 v[Chunk<CT>::WORDS / 256];
};
template <int CT>
__device__ __forceinline__ void fetch_chunk(const uint4 *__restrict__ img, int ci, ChunkRegs<CT> &r) {
    const uint4 *__restrict__ src = img + (size_t)ci * Chunk<CT>::WORDS;
#pragma unroll
    for (int i = 0; i < Chunk<CT>::WORDS / 256; ++i) r.v[i] = src[threadIdx.x + 256 * i];
}
template <int CT>
__device__ __forceinline__ void store_chunk(const ChunkRegs<CT> &r, uint4 *slot) {
#pragma unroll
    for (int i = 0; i < Chunk<CT>::WORDS / 256; ++i) slot[threadIdx.x + 256 * i] = r.v[i];
}

template <int CT>
__device__ __forceinline__ Frag<2> chunk_frag(const uint4 *slot, int j, int s, int lane) {
    Frag<2> f;
    f.p[0] = __builtin_bit_cast(bf16x8, slot[((j * 2 + s) * 2 + 0) * 64 + lane]);
    f.p[1] = __builtin_bit_cast(bf16x8, slot[((j * 2 + s) * 2 + 1) * 64 + lane]);
    return f;
}

// ------------------------------------------------------------------------------------------
// pass 2 (forward): a1 -> y2 = a1 W2^T -> statistics + ext over the K neighbours
//   img    B image of W2^T (H x O), CT column tiles per block
//   pack1  {scale1[H], shift1[H], ...}; sgn2[O] = +1/-1: which extreme of y2 the pool keeps
//   ysel/ksel (B*M, O): the extreme and its slot;  part[block][2*O] = {sum, sumsq} of y2
// LDS: the whole image when it fits (RES), else two streaming slots.
// ------------------------------------------------------------------------------------------
template <int H, int O, int CT, bool RES>
__global__ __launch_bounds__(256) void wide_fwd_main_kernel(WideArgs a, const uint4 *__restrict__ img,
                                                            const float *__restrict__ pack1,
                                                            const float *__restrict__ sgn2,
                                                            float *__restrict__ ysel,
                                                            unsigned char *__restrict__ ksel,
                                                            float *__restrict__ part) {
    constexpr int NKC = H / 32, NCB = O / (32 * CT), NCH = NKC * NCB;
    constexpr int SLOTS = RES ? NCH : 2;
    extern __shared__ uint4 dyn[];
    uint4 *wl = dyn;                                                 // SLOTS chunks
    float *st = reinterpret_cast<float *>(dyn + SLOTS * Chunk<CT>::WORDS);   // [4 waves][2*O]
    const int lane = lane_id(), w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    for (int e = threadIdx.x; e < WIDE_WAVES * 2 * O; e += 256) st[e] = 0.0f;
    if (RES) {
        for (int ci = 0; ci < NCH; ++ci) stage_chunk<CT>(img, ci, wl + ci * Chunk<CT>::WORDS);
    }
    __syncthreads();
    float *mine = st + w * 2 * O;
    const int step = gridDim.x * WIDE_WAVES;
    const int rounds = (a.ntiles + step - 1) / step;
    // streaming: chunk ci of the (cyclic) sequence lives in slot ci & 1; the first one is staged here and
    // every iteration prefetches its successor (the sequence wraps from the last chunk of a tile to chunk 0)
    ChunkRegs<CT> pre;
    if (!RES) {
        stage_chunk<CT>(img, 0, wl);
        __syncthreads();
    }
    int seq = 0;                                   // running chunk counter (parity = slot)
    for (int it = 0; it < rounds; ++it) {
        const int tile_raw = blockIdx.x * WIDE_WAVES + w + it * step;
        const bool valid = tile_raw < a.ntiles;
        const int tile = valid ? tile_raw : a.ntiles - 1;
        const int cloud = tile / a.m;
        const int nn = a.idx[(size_t)tile * 32 + r];
        const float *__restrict__ row = a.U + ((size_t)cloud * a.n + nn) * H;
        const float *__restrict__ vq = a.V + (size_t)tile * H;
#pragma unroll 1
        for (int cb = 0; cb < NCB; ++cb) {
            f32x16 acc[CT];
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
#pragma unroll 1
            for (int kc = 0; kc < NKC; ++kc) {
                const int ci = cb * NKC + kc;
                const uint4 *slot;
                if (RES) {
                    slot = wl + ci * Chunk<CT>::WORDS;
                } else {
                    slot = wl + (seq & 1) * Chunk<CT>::WORDS;
                    fetch_chunk<CT>(img, ci + 1 == NCH ? 0 : ci + 1, pre);      // in flight behind the MFMAs
                }
                Frag<2> af[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) af[s] = a1_frag(row, vq, pack1, H, kc * 32 + s * 16 + h * 8);
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc[j] = mfma<2>(af[s], chunk_frag<CT>(slot, j, s, lane), acc[j]);
                if (!RES) {
                    // the other slot held chunk seq-1: every wave finished reading it before the barrier
                    // that ended the previous iteration, so it can be overwritten now
                    store_chunk<CT>(pre, wl + ((seq + 1) & 1) * Chunk<CT>::WORDS);
                    __syncthreads();
                    ++seq;
                }
            }
            // epilogue of this column block: lane = channel, register = position acc_row(i, h)
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int col = (cb * CT + j) * 32 + r;
                const float sg = sgn2[col];
                float s1 = 0.0f, s2 = 0.0f, best = -__builtin_inff();
                int bpos = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float y = acc[j][i];
                    s1 += y;
                    s2 = __builtin_fmaf(y, y, s2);
                    const float v = y * sg;
                    if (v > best) { best = v; bpos = acc_row(i, h); }     // ascending positions: first maximum
                }
                s1 += shfl_xor32(s1);
                s2 += shfl_xor32(s2);
                const float ob = shfl_xor32(best);
                const int op = shfl_xor32i(bpos);
                if (ob > best || (ob == best && op < bpos)) { best = ob; bpos = op; }
                if (h == 0 && valid) {
                    mine[col] += s1;                     // this wave owns its row of `st`: plain update
                    mine[O + col] += s2;
                    ysel[(size_t)tile * O + col] = best * sg;
                    ksel[(size_t)tile * O + col] = (unsigned char)bpos;
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * O; e += 256)
        part[(size_t)blockIdx.x * 2 * O + e] = (st[e] + st[2 * O + e]) + (st[4 * O + e] + st[6 * O + e]);
}

// ------------------------------------------------------------------------------------------
// backward over the positions.
//   dL/da1[pos, :] = S[pos, :] W2 + a1[pos, :] Qm + evec,  S[pos, c] = goa[q, c] [ksel[q, c] == pos]
//   (Qm = W2^T diag(D2) W2, evec = E2 W2: BatchNorm-2's feedback without recomputing y2),
//   g_u = dL/da1 [a1 > 0];  the block's row of part: {T1 = sum g_u, T2 = sum g_u yhat1}[H];
//   A[b, n, :] += g_u of every position that gathers point n (float atomics, the ball-query
//   fill run folded first); HA[q, :] = sum_k g_u, HB[q, :] = sum_k yhat1.
//   img: B image of Z = [W2 ; Qm] ((O + H) x H), CT = min(4, H/32) column tiles per block.
//   pack1 = {scale1, shift1, mean1, invstd1}[H].
// ------------------------------------------------------------------------------------------
template <int H, int O, int CT, bool RES>
__global__ __launch_bounds__(256) void wide_bwd_main_kernel(WideArgs a, const uint4 *__restrict__ img,
                                                            const float *__restrict__ pack1,
                                                            const float *__restrict__ evec,
                                                            const float *__restrict__ goa,
                                                            const unsigned char *__restrict__ ksel,
                                                            float *__restrict__ A, float *__restrict__ HA,
                                                            float *__restrict__ HB, float *__restrict__ part) {
    constexpr int NKS = O / 32, NKC = (O + H) / 32, NCB = H / (32 * CT), NCH = NKC * NCB;
    constexpr int SLOTS = RES ? NCH : 2;
    extern __shared__ uint4 dyn[];
    uint4 *wl = dyn;
    float *st = reinterpret_cast<float *>(dyn + SLOTS * Chunk<CT>::WORDS);   // [4 waves][2*H]
    const int lane = lane_id(), w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    for (int e = threadIdx.x; e < WIDE_WAVES * 2 * H; e += 256) st[e] = 0.0f;
    if (RES) {
        for (int ci = 0; ci < NCH; ++ci) stage_chunk<CT>(img, ci, wl + ci * Chunk<CT>::WORDS);
    }
    __syncthreads();
    float *mine = st + w * 2 * H;
    const int step = gridDim.x * WIDE_WAVES;
    const int rounds = (a.ntiles + step - 1) / step;
    ChunkRegs<CT> pre;
    if (!RES) {
        stage_chunk<CT>(img, 0, wl);
        __syncthreads();
    }
    int seq = 0;
    for (int it = 0; it < rounds; ++it) {
        const int tile_raw = blockIdx.x * WIDE_WAVES + w + it * step;
        const bool valid = tile_raw < a.ntiles;
        const int tile = valid ? tile_raw : a.ntiles - 1;
        const int cloud = tile / a.m;
        const int *__restrict__ ip = a.idx + (size_t)tile * 32;
        const int nn = ip[r];
        const float *__restrict__ ub = a.U + (size_t)cloud * a.n * H;
        const float *__restrict__ row = ub + (size_t)nn * H;
        const float *__restrict__ vq = a.V + (size_t)tile * H;
        const float *__restrict__ gq = goa + (size_t)tile * O;
        const unsigned char *__restrict__ kq = ksel + (size_t)tile * O;
        // neighbour of every accumulator row of this lane: positions acc_row(i, h) = 4 runs of 4
        int nrow[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int4 v = *reinterpret_cast<const int4 *>(ip + 8 * g + 4 * h);
            nrow[4 * g] = v.x; nrow[4 * g + 1] = v.y; nrow[4 * g + 2] = v.z; nrow[4 * g + 3] = v.w;
        }
        const int first = ip[0];
#pragma unroll 1
        for (int cb = 0; cb < NCB; ++cb) {
            f32x16 acc[CT];
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
#pragma unroll 1
            for (int kc = 0; kc < NKC; ++kc) {
                const int ci = cb * NKC + kc;
                const uint4 *slot;
                if (RES) {
                    slot = wl + ci * Chunk<CT>::WORDS;
                } else {
                    slot = wl + (seq & 1) * Chunk<CT>::WORDS;
                    fetch_chunk<CT>(img, ci + 1 == NCH ? 0 : ci + 1, pre);
                }
                Frag<2> af[2];
                if (kc < NKS) {            // rows of S: the upstream gradient at the pooled slot
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int c0 = kc * 32 + s * 16 + h * 8;
                        const float4 g0 = *reinterpret_cast<const float4 *>(gq + c0);
                        const float4 g1 = *reinterpret_cast<const float4 *>(gq + c0 + 4);
                        const uint2 kk = *reinterpret_cast<const uint2 *>(kq + c0);
                        float t[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned sel = ((e < 4 ? kk.x : kk.y) >> (8 * (e & 3))) & 0xffu;
                            t[e] = sel == (unsigned)r ? t[e] : 0.0f;
                        }
                        af[s] = make_frag<2>(t);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        af[s] = a1_frag(row, vq, pack1, H, (kc - NKS) * 32 + s * 16 + h * 8);
                }
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc[j] = mfma<2>(af[s], chunk_frag<CT>(slot, j, s, lane), acc[j]);
                if (!RES) {
                    store_chunk<CT>(pre, wl + ((seq + 1) & 1) * Chunk<CT>::WORDS);
                    __syncthreads();
                    ++seq;
                }
            }
            // epilogue: lane = mid channel, register = position
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int mid = (cb * CT + j) * 32 + r;
                const float sc = pack1[mid], sh = pack1[H + mid], mu = pack1[2 * H + mid], iv = pack1[3 * H + mid];
                const float ev = evec[mid], vv = vq[mid];
                float t1 = 0.0f, t2 = 0.0f, hb = 0.0f, gfirst = 0.0f;
                float u[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) u[i] = ub[(size_t)nrow[i] * H + mid];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float y1 = u[i] - vv;
                    const float yh = (y1 - mu) * iv;
                    const float gu = __builtin_fmaf(y1, sc, sh) > 0.0f ? acc[j][i] + ev : 0.0f;
                    t1 += gu;
                    t2 = __builtin_fmaf(gu, yh, t2);
                    hb += yh;
                    if (nrow[i] == first) gfirst += gu;          // slot 0 and the fill run behind the hits
                    else if (valid) atomicAdd(A + ((size_t)cloud * a.n + nrow[i]) * H + mid, gu);
                }
                t1 += shfl_xor32(t1);
                t2 += shfl_xor32(t2);
                hb += shfl_xor32(hb);
                gfirst += shfl_xor32(gfirst);
                if (h == 0 && valid) {
                    atomicAdd(A + ((size_t)cloud * a.n + first) * H + mid, gfirst);
                    mine[mid] += t1;
                    mine[H + mid] += t2;
                    HA[(size_t)tile * H + mid] = t1;
                    HB[(size_t)tile * H + mid] = hb;
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * H; e += 256)
        part[(size_t)blockIdx.x * 2 * H + e] = (st[e] + st[2 * H + e]) + (st[4 * H + e] + st[6 * H + e]);
}

// ------------------------------------------------------------------------------------------
// weight-gradient products over the positions:
//   R[(O + H), H] = [S^T ; a1^T] a1   (rows 0..O-1: sum_q goa[q,c] a1[q, ksel[q,c], :], the sparse
//   part of dL/dW2; rows O..: the Gram matrix sum a1^T a1),  suma[H] = sum a1.
// Workgroup = NW waves; wave w owns the 32-row block rb = blockIdx.y * NW + w and all H columns
// (H/32 accumulator tiles); the workgroup's range of query tiles is blockIdx.x of gridDim.x
// (split-K).  Per tile the a1 operand (k = position, column = mid channel) is built ONCE by the
// workgroup into LDS in fragment order; its a1^T rows are the same fragments.
// Outputs: Rpart[split][(O+H)][H], sumapart[split][H] (summed by the caller in float64).
// ------------------------------------------------------------------------------------------
template <int H, int O, int NW>
__global__ __launch_bounds__(NW * 64) void wide_wgrad_kernel(WideArgs a, const float *__restrict__ pack1,
                                                             const float *__restrict__ goa,
                                                             const unsigned char *__restrict__ ksel,
                                                             float *__restrict__ Rpart,
                                                             float *__restrict__ sumapart) {
    constexpr int NJ = H / 32, NT = NW * 64, NFRAG = NJ * 2 * 64;      // fragment-lanes per tile
    __shared__ uint4 bl[NJ * 2 * 2 * 64];                              // [j][s][part][lane]
    __shared__ float sred[NFRAG];
    const int lane = lane_id(), w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int rb = blockIdx.y * NW + w;                                // this wave's row block
    const bool rb_ok = rb < (O + H) / 32;
    const bool s_rows = rb < O / 32;
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    constexpr int FPT = (NFRAG + NT - 1) / NT;                         // fragments per thread
    float suma[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) suma[f] = 0.0f;
    const int per = (a.ntiles + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per, t1 = min(a.ntiles, t0 + per);
    for (int tile = t0; tile < t1; ++tile) {
        const int cloud = tile / a.m;
        const int *__restrict__ ip = a.idx + (size_t)tile * 32;
        const float *__restrict__ ub = a.U + (size_t)cloud * a.n * H;
        const float *__restrict__ vq = a.V + (size_t)tile * H;
        __syncthreads();                                               // previous tile's reads are done
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
            const int fi = threadIdx.x + NT * f;                       // (j, s, lane') = fragment-lane
            if (fi < NFRAG) {
                const int fl = fi & 63, s = (fi >> 6) & 1, j = fi >> 7;
                const int mid = j * 32 + (fl & 31), p0 = s * 16 + (fl >> 5) * 8;
                const float sc = pack1[mid], sh = pack1[H + mid], vv = vq[mid];
                float t[8], sum = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y1 = ub[(size_t)ip[p0 + e] * H + mid] - vv;
                    const float v = __builtin_fmaf(y1, sc, sh);
                    t[e] = v > 0.0f ? v : 0.0f;
                    sum += t[e];
                }
                suma[f] += sum;
                const Frag<2> fr = make_frag<2>(t);
                bl[((j * 2 + s) * 2 + 0) * 64 + fl] = __builtin_bit_cast(uint4, fr.p[0]);
                bl[((j * 2 + s) * 2 + 1) * 64 + fl] = __builtin_bit_cast(uint4, fr.p[1]);
            }
        }
        __syncthreads();
        if (!rb_ok) continue;
        Frag<2> af[2];
        if (s_rows) {                   // A = S^T: row = channel c of this lane, k = position
            const int c = rb * 32 + r;
            const float gv = goa[(size_t)tile * O + c];
            const int kp = ksel[(size_t)tile * O + c];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (s * 16 + h * 8 + e) == kp ? gv : 0.0f;
                af[s] = make_frag<2>(t);
            }
        } else {                        // A = a1^T rows of mid block rb - O/32: the same fragments
            const int j0 = rb - O / 32;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                af[s].p[0] = __builtin_bit_cast(bf16x8, bl[((j0 * 2 + s) * 2 + 0) * 64 + lane]);
                af[s].p[1] = __builtin_bit_cast(bf16x8, bl[((j0 * 2 + s) * 2 + 1) * 64 + lane]);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                Frag<2> bf;
                bf.p[0] = __builtin_bit_cast(bf16x8, bl[((j * 2 + s) * 2 + 0) * 64 + lane]);
                bf.p[1] = __builtin_bit_cast(bf16x8, bl[((j * 2 + s) * 2 + 1) * 64 + lane]);
                acc[j] = mfma<2>(af[s], bf, acc[j]);
            }
    }
    if (rb_ok) {
        float *__restrict__ out = Rpart + ((size_t)blockIdx.x * (O + H) + rb * 32) * H;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) out[(size_t)acc_row(i, h) * H + j * 32 + r] = acc[j][i];
    }
    // sum a1: the 4 fragment-lanes (s, h) of one mid channel, fixed order
    if (blockIdx.y == 0) {
        __syncthreads();
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
            const int fi = threadIdx.x + NT * f;
            if (fi < NFRAG) sred[fi] = suma[f];
        }
        __syncthreads();
        for (int mid = threadIdx.x; mid < H; mid += NT) {
            const int j = mid >> 5, rr = mid & 31;
            const float v = (sred[(j * 2 + 0) * 64 + rr] + sred[(j * 2 + 0) * 64 + 32 + rr]) +
                            (sred[(j * 2 + 1) * 64 + rr] + sred[(j * 2 + 1) * 64 + 32 + rr]);
            sumapart[(size_t)blockIdx.x * H + mid] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// out[ncol] (float64) = column sums of part[rows][ncol] (float32), fixed order: deterministic.
// Block = 64 columns x 4 row groups; used for every "sum the partial rows" step of this path
// (a library reduction with cross-block semaphores is avoided on purpose: the step must replay
// identically from a hipGraph).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void wide_colsum_kernel(const T *__restrict__ part, int rows, int ncol,
                                                          double *__restrict__ out) {
    // grid = (column blocks of 64, row chunks): out[blockIdx.y][ncol] = sums over this chunk's rows
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < ncol) {
        int r = r0 + g;
        for (; r + 12 < r1; r += 16) {
            const T a = part[(size_t)r * ncol + c], b = part[(size_t)(r + 4) * ncol + c];
            const T d = part[(size_t)(r + 8) * ncol + c], e = part[(size_t)(r + 12) * ncol + c];
            s0 += (double)a; s1 += (double)b; s2 += (double)d; s3 += (double)e;
        }
        for (; r < r1; r += 4) s0 += (double)part[(size_t)r * ncol + c];
    }
    red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && c < ncol)
        out[(size_t)blockIdx.y * ncol + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static bool wide_shape_ok(int H, int O) {
    return (H == 32 || H == 64 || H == 128 || H == 256) && O == 2 * H;
}

static bool wide_args_ok(int b, int n, int m, const void *U, const void *V, const void *idx) {
    return b > 0 && n > 0 && m > 0 && (long long)b * m <= 0x7fffffffLL / 64 && U && V && idx;
}

template <int H>
static constexpr int fwd_ct() { return 2 * H / 32 >= 4 ? 4 : 2 * H / 32; }     // column tiles per block, O = 2H
template <int H>
static constexpr int bwd_ct() { return H / 32 >= 4 ? 4 : H / 32; }

// whole image resident in LDS when it and the statistics rows fit comfortably (two workgroups per CU)
template <int H>
static constexpr bool fwd_res() { return (size_t)(H / 32) * (2 * H / (32 * fwd_ct<H>())) * Chunk<fwd_ct<H>()>::WORDS * 16 <= 48 * 1024; }
template <int H>
static constexpr bool bwd_res() { return (size_t)(3 * H / 32) * (H / (32 * bwd_ct<H>())) * Chunk<bwd_ct<H>()>::WORDS * 16 <= 48 * 1024; }

static int wide_grid(int ntiles) {
    const int want = (ntiles + WIDE_WAVES - 1) / WIDE_WAVES;
    return want < 512 ? want : 512;            // two workgroups per CU; every workgroup leaves one partial row
}

}  // namespace apn

using namespace apn;

extern "C" int apn_sa_wide_grid(int b, int m) {
    if (b <= 0 || m <= 0) return 0;
    return wide_grid(b * m);
}

#define APN_WIDE_DISPATCH(H_, ...)   \
    switch (H_) {                    \
    case 32: { constexpr int H = 32; __VA_ARGS__; } break;    \
    case 64: { constexpr int H = 64; __VA_ARGS__; } break;    \
    case 128: { constexpr int H = 128; __VA_ARGS__; } break;  \
    case 256: { constexpr int H = 256; __VA_ARGS__; } break;  \
    default: return APN_EINVAL;      \
    }

extern "C" int apn_sa_wide_colsum_chunks(int rows, int ncol) {
    // row chunks of the first pass: enough workgroups to fill the chip, at least 64 rows each
    if (rows <= 0 || ncol <= 0) return 1;
    const int colblocks = (ncol + 63) / 64;
    int chunks = (512 + colblocks - 1) / colblocks;
    if (chunks > (rows + 63) / 64) chunks = (rows + 63) / 64;
    return chunks < 1 ? 1 : chunks;
}

extern "C" int apn_sa_wide_colsum(const float *part, int rows, int ncol, double *scratch, double *out,
                                  void *stream) {
    if (rows < 0 || ncol <= 0 || !part || !out) return APN_EINVAL;
    const int chunks = apn_sa_wide_colsum_chunks(rows, ncol);
    if (chunks > 1 && !scratch) return APN_EINVAL;
    const dim3 g1((ncol + 63) / 64, chunks);
    hipLaunchKernelGGL(wide_colsum_kernel<float>, g1, dim3(256), 0, (hipStream_t)stream, part, rows, ncol,
                       chunks > 1 ? scratch : out);
    APN_LAUNCH_CHECK();
    if (chunks > 1) {
        hipLaunchKernelGGL(wide_colsum_kernel<double>, dim3((ncol + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream,
                           (const double *)scratch, chunks, ncol, out);
        APN_LAUNCH_CHECK();
    }
    return APN_OK;
}

extern "C" int apn_sa_wide_stats1(int b, int n, int m, int c_mid, const float *U, const float *V,
                                  const int *idx, float *part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx) || !part) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx};
    const int grid = wide_grid(a.ntiles);
    APN_WIDE_DISPATCH(c_mid, hipLaunchKernelGGL((wide_stats1_kernel<H>), dim3(grid), dim3(256), 0,
                                                (hipStream_t)stream, a, part));
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_fwd_main(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                    const int *idx, const void *w2_image, const float *pack1,
                                    const float *sgn2, float *ysel, void *ksel, float *part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx) || !wide_shape_ok(c_mid, c_out)) return APN_EINVAL;
    if (!w2_image || !pack1 || !sgn2 || !ysel || !ksel || !part) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx};
    const int grid = wide_grid(a.ntiles);
    APN_WIDE_DISPATCH(c_mid, {
        constexpr int O = 2 * H, CT = fwd_ct<H>();
        constexpr bool RES = fwd_res<H>();
        constexpr int NCH = (H / 32) * (O / (32 * CT));
        const size_t lds = (size_t)(RES ? NCH : 2) * Chunk<CT>::WORDS * 16 + (size_t)WIDE_WAVES * 2 * O * 4;
        hipLaunchKernelGGL((wide_fwd_main_kernel<H, O, CT, RES>), dim3(grid), dim3(256), lds,
                           (hipStream_t)stream, a, (const uint4 *)w2_image, pack1, sgn2, ysel,
                           (unsigned char *)ksel, part);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_bwd_main(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                    const int *idx, const void *z_image, const float *pack1,
                                    const float *evec, const float *goa, const void *ksel, float *A,
                                    float *HA, float *HB, float *part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx) || !wide_shape_ok(c_mid, c_out)) return APN_EINVAL;
    if (!z_image || !pack1 || !evec || !goa || !ksel || !A || !HA || !HB || !part) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx};
    const int grid = wide_grid(a.ntiles);
    APN_WIDE_DISPATCH(c_mid, {
        constexpr int O = 2 * H, CT = bwd_ct<H>();
        constexpr bool RES = bwd_res<H>();
        constexpr int NCH = ((O + H) / 32) * (H / (32 * CT));
        const size_t lds = (size_t)(RES ? NCH : 2) * Chunk<CT>::WORDS * 16 + (size_t)WIDE_WAVES * 2 * H * 4;
        hipLaunchKernelGGL((wide_bwd_main_kernel<H, O, CT, RES>), dim3(grid), dim3(256), lds,
                           (hipStream_t)stream, a, (const uint4 *)z_image, pack1, evec, goa,
                           (const unsigned char *)ksel, A, HA, HB, part);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_wgrad_splits(int b, int m, int c_mid) {
    if (b <= 0 || m <= 0 || !wide_shape_ok(c_mid, 2 * c_mid)) return 0;
    const int ntiles = b * m;
    const int groups = (3 * c_mid / 32 + 7) / 8;           // workgroups per split (8 row blocks each)
    int splits = 512 / groups;
    if (splits > ntiles / 4) splits = ntiles / 4;          // at least 4 query tiles per workgroup
    return splits < 1 ? 1 : splits;
}

extern "C" int apn_sa_wide_wgrad(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                 const int *idx, const float *pack1, const float *goa, const void *ksel,
                                 int splits, float *r_part, float *suma_part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx) || !wide_shape_ok(c_mid, c_out)) return APN_EINVAL;
    if (!pack1 || !goa || !ksel || !r_part || !suma_part || splits < 1) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx};
    APN_WIDE_DISPATCH(c_mid, {
        constexpr int O = 2 * H, NRB = (O + H) / 32, NW = NRB < 8 ? NRB : 8;
        const dim3 grid(splits, (NRB + NW - 1) / NW);
        hipLaunchKernelGGL((wide_wgrad_kernel<H, O, NW>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a,
                           pack1, goa, (const unsigned char *)ksel, r_part, suma_part);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}
