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
#include <type_traits>

namespace apn {

struct WideArgs {
    int ntiles;          // B * M queries = upper bound of the tiles in use
    int n, m;            // support points / queries per cloud
    const float *U;      // (B,N,H)
    const float *V;      // (B,M,H)
    const int *idx;      // (B,M,32)
    const int *tmap;     // tile map (csrc/sa_wide_glue.hip: apn_sa_wide_tilemap): 32 ROWS per tile, a row = one
                         // distinct neighbour of one query with its multiplicity
};

// Tile-map accessors.  rowinfo = qlocal | slot << 8 | mult << 16 (| queries of the tile << 24 in row 0).
__device__ __forceinline__ int tm_tiles(const WideArgs &a) { return a.tmap[0]; }
__device__ __forceinline__ const int *tm_tq0(const WideArgs &a) { return a.tmap + 4; }
__device__ __forceinline__ const unsigned *tm_rows(const WideArgs &a) {
    return reinterpret_cast<const unsigned *>(a.tmap + 4 + ((a.ntiles + 3) & ~3));
}
__device__ __forceinline__ const int *tm_nn(const WideArgs &a) {
    return a.tmap + 4 + ((a.ntiles + 3) & ~3) + (size_t)32 * a.ntiles;
}
__device__ __forceinline__ int ri_q(unsigned info) { return (int)(info & 0xffu); }
__device__ __forceinline__ int ri_slot(unsigned info) { return (int)((info >> 8) & 0xffu); }
__device__ __forceinline__ int ri_mult(unsigned info) { return (int)((info >> 16) & 0xffu); }

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
    const int nt = tm_tiles(a);
    const int *__restrict__ tq0 = tm_tq0(a);
    const unsigned *__restrict__ rows = tm_rows(a);
    for (int tile = blockIdx.x * WIDE_WAVES + w; tile < nt; tile += gridDim.x * WIDE_WAVES) {
        const int q0 = tq0[tile];
        const unsigned *__restrict__ ri = rows + (size_t)tile * 32;
        const int *__restrict__ rnn = tm_nn(a) + (size_t)tile * 32;
        const float *__restrict__ ub = a.U + (size_t)(q0 / a.m) * a.n * H;
#pragma unroll 8
        for (int p0 = 0; p0 < 32; p0 += PP) {
            const unsigned info = ri[p0 + (PP == 2 ? (lane >> 5) : 0)];
            const int q = q0 + (ri_mult(info) ? ri_q(info) : 0);
            const float wgt = (float)ri_mult(info);              // 0: padding row
            const int nn = rnn[p0 + (PP == 2 ? (lane >> 5) : 0)];
            const float *__restrict__ row = ub + (size_t)nn * H;
            const float *__restrict__ vr = a.V + (size_t)q * H;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int ch = PP == 2 ? (lane & 31) : lane + 64 * j;
                const float y = row[ch] - vr[ch];
                const float wy = wgt * y;
                s[j] += wy;
                ss[j] = __builtin_fmaf(wy, y, ss[j]);
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
// The raw words of one chunk's A operand (two k-steps), requested a chunk ahead of their use by the backward pass:
// rows of S (f[s][0..1] = the gradient's 8 channels, kk = their pooled slots) or of a1 (f[s][0..7] = U, V and
// BatchNorm-1's scale / shift for 8 channels).
struct RawA {
    float4 f[2][8];
    uint2 kk[2];
};

__device__ __forceinline__ void a1_raw(const float *__restrict__ ub, unsigned uo, const float *__restrict__ vb,
                                       unsigned vo, const float *__restrict__ pack1, int H, int ch0, float4 (&f)[8]) {
    f[0] = *reinterpret_cast<const float4 *>(ub + (uo + ch0));
    f[1] = *reinterpret_cast<const float4 *>(ub + (uo + ch0 + 4));
    f[2] = *reinterpret_cast<const float4 *>(vb + (vo + ch0));
    f[3] = *reinterpret_cast<const float4 *>(vb + (vo + ch0 + 4));
    f[4] = *reinterpret_cast<const float4 *>(pack1 + ch0);
    f[5] = *reinterpret_cast<const float4 *>(pack1 + ch0 + 4);
    f[6] = *reinterpret_cast<const float4 *>(pack1 + H + ch0);
    f[7] = *reinterpret_cast<const float4 *>(pack1 + H + ch0 + 4);
}

// the multiplicity-weighted a1 fragment from those words (the arithmetic of a1_frag below)
__device__ __forceinline__ Frag<2> a1_from_raw(const float4 (&f)[8], float wgt) {
    float t[8];
    t[0] = __builtin_fmaf(f[0].x - f[2].x, f[4].x, f[6].x); t[1] = __builtin_fmaf(f[0].y - f[2].y, f[4].y, f[6].y);
    t[2] = __builtin_fmaf(f[0].z - f[2].z, f[4].z, f[6].z); t[3] = __builtin_fmaf(f[0].w - f[2].w, f[4].w, f[6].w);
    t[4] = __builtin_fmaf(f[1].x - f[3].x, f[5].x, f[7].x); t[5] = __builtin_fmaf(f[1].y - f[3].y, f[5].y, f[7].y);
    t[6] = __builtin_fmaf(f[1].z - f[3].z, f[5].z, f[7].z); t[7] = __builtin_fmaf(f[1].w - f[3].w, f[5].w, f[7].w);
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (t[e] > 0.0f ? t[e] : 0.0f) * wgt;
    return make_frag<2>(t);
}

// A fragment of a1 = relu(scale1 * (U[n] - V[q]) + shift1): lane (pos r, h), 8 channels from ch0.
// ub / vb: wave-uniform bases (the cloud's rows of U, the tile's rows of V); uo / vo: this lane's row offsets in
// elements -- 32-bit, so the loads take the scalar-base + vector-offset form (no 64-bit address arithmetic).
__device__ __forceinline__ Frag<2> a1_frag(const float *__restrict__ ub, unsigned uo, const float *__restrict__ vb,
                                           unsigned vo, const float *__restrict__ pack1, int H, int ch0,
                                           float wgt = 1.0f, bool weighted = false) {
    const float4 u0 = *reinterpret_cast<const float4 *>(ub + (uo + ch0));
    const float4 u1 = *reinterpret_cast<const float4 *>(ub + (uo + ch0 + 4));
    const float4 v0 = *reinterpret_cast<const float4 *>(vb + (vo + ch0));
    const float4 v1 = *reinterpret_cast<const float4 *>(vb + (vo + ch0 + 4));
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
    if (weighted) {
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] *= wgt;
    }
    return make_frag<2>(t);
}

// What a wave needs to start on a tile: lane (r, h) holds row r's record and neighbour, and the tile's first
// query.  Loaded one tile AHEAD (the three loads are independent of each other; the rows of U they lead to are
// the only dependent round trip left in the loop) and SPECULATIVELY: the first head is requested before the
// number of tiles in use is known (the arrays hold B*M tiles, so any tile < B*M is readable); a head beyond
// the tiles in use is replaced by `no_tile` (no query, multiplicity 0, addresses of row 0).
struct TileHead {
    unsigned info;
    int nn, q0;
};
__device__ __forceinline__ TileHead load_head(const WideArgs &a, int tile, int r) {
    TileHead t;
    t.info = tm_rows(a)[(size_t)tile * 32 + r];
    t.nn = tm_nn(a)[(size_t)tile * 32 + r];
    t.q0 = tm_tq0(a)[tile];
    return t;
}
__device__ __forceinline__ TileHead no_tile() {
    TileHead t;
    t.info = 0xffu; t.nn = 0; t.q0 = 0;
    return t;
}
// The records of this lane's 16 accumulator rows acc_row(i, h) (row p's record sits in lane p).
__device__ __forceinline__ void row_meta(unsigned info, int h, unsigned (&meta)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) meta[i] = (unsigned)__builtin_amdgcn_ds_bpermute(acc_row(i, h) << 2, (int)info);
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

// The chunk loop's rendezvous: this wave's LDS operations done, then the workgroup's barrier.  (__syncthreads() also waits
// for every global load in flight -- vmcnt(0) -- i.e. for the prefetches of the NEXT step issued a moment ago: one L2 round
// trip per chunk step on the critical path.  The slots are handed over through LDS only; the loads the stores depend on are
// waited for by the stores themselves.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
__global__ __launch_bounds__(256, (H == 32 ? 3 : 2)) void wide_fwd_main_kernel(WideArgs a, const uint4 *__restrict__ img,
                                                            const float *__restrict__ pack1,
                                                            const float *__restrict__ sgn2,
                                                            float *__restrict__ ysel,
                                                            unsigned char *__restrict__ ksel,
                                                            float *__restrict__ part) {
    constexpr int NKC = H / 32, NCB = O / (32 * CT), NCH = NKC * NCB;
    constexpr int SLOTS = RES ? NCH : 2;
    constexpr bool AHEADA = H >= 128;        // a1's rows requested one chunk step ahead, BatchNorm-1's constants in LDS
    extern __shared__ uint4 dyn[];
    uint4 *wl = dyn;                                                 // SLOTS chunks
    float *st = reinterpret_cast<float *>(dyn + SLOTS * Chunk<CT>::WORDS);   // [4 waves][2*O]
    float *cst = st + WIDE_WAVES * 2 * O;                            // wide blocks (H >= 128): BatchNorm-1's {scale, shift}[H]
    const int lane = lane_id(), r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int step = gridDim.x * WIDE_WAVES;
    const int first = blockIdx.x * WIDE_WAVES + w;                   // wave-uniform tile numbers
    TileHead head = load_head(a, first < a.ntiles ? first : a.ntiles - 1, r);
    for (int e = threadIdx.x; e < WIDE_WAVES * 2 * O; e += 256) st[e] = 0.0f;
    if (AHEADA) {
        for (int e = threadIdx.x; e < 2 * H; e += 256) cst[e] = pack1[e];
    }
    if (RES) {
        for (int ci = 0; ci < NCH; ++ci) stage_chunk<CT>(img, ci, wl + ci * Chunk<CT>::WORDS);
    }
    __syncthreads();
    float *mine = st + w * 2 * O;
    const int nt = tm_tiles(a);
    const int rounds = (nt + step - 1) / step;
    // streaming: chunk ci of the (cyclic) sequence lives in slot ci & 1; the first one is staged here and
    // every iteration prefetches its successor (the sequence wraps from the last chunk of a tile to chunk 0)
    ChunkRegs<CT> pre;
    if (!RES) {
        stage_chunk<CT>(img, 0, wl);
        __syncthreads();
    }
    int seq = 0;                                   // running chunk counter (parity = slot)
    for (int it = 0; it < rounds; ++it) {
        const int tile_raw = first + it * step;
        const bool valid = tile_raw < nt;
        const TileHead cur = valid ? head : no_tile();
        {
            const int nx = tile_raw + step;
            head = load_head(a, nx < a.ntiles ? nx : a.ntiles - 1, r);          // consumed by the next iteration
        }
        const int q0 = __builtin_amdgcn_readfirstlane(cur.q0);
        const int nq = __builtin_amdgcn_readfirstlane((int)(cur.info >> 24));      // lane 0 holds row 0
        const float *__restrict__ ub = a.U + (size_t)(q0 / a.m) * a.n * H;         // wave-uniform bases
        const float *__restrict__ vb = a.V + (size_t)q0 * H;
        const unsigned uo = (unsigned)cur.nn * H, vo = (ri_mult(cur.info) ? ri_q(cur.info) : 0u) * H;
        float *__restrict__ ysel0 = ysel + (size_t)q0 * O;
        unsigned char *__restrict__ ksel0 = ksel + (size_t)q0 * O;
        unsigned meta[16];
        row_meta(cur.info, h, meta);
        // wide blocks (H >= 128): the rows of U and V a chunk step turns into its a1 fragments are requested ONE STEP
        // AHEAD and BatchNorm-1's constants come from LDS -- requested and consumed in the same step, sixteen 16-byte loads
        // stood between the barrier and the step's MFMAs: ~3.5-4 us per step against 0.8 us of MFMAs
        float4 raw[2][4], rnx[2][4];
        auto load_uv = [&](int kc, float4 (&f)[2][4]) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const unsigned ch0 = kc * 32 + s2 * 16 + h * 8;
                f[s2][0] = *reinterpret_cast<const float4 *>(ub + (uo + ch0));
                f[s2][1] = *reinterpret_cast<const float4 *>(ub + (uo + ch0 + 4));
                f[s2][2] = *reinterpret_cast<const float4 *>(vb + (vo + ch0));
                f[s2][3] = *reinterpret_cast<const float4 *>(vb + (vo + ch0 + 4));
            }
        };
        auto a1_lds = [&](const float4 (&f)[4], int ch0) {
            const float4 c0 = *reinterpret_cast<const float4 *>(cst + ch0), c1 = *reinterpret_cast<const float4 *>(cst + ch0 + 4);
            const float4 d0 = *reinterpret_cast<const float4 *>(cst + H + ch0), d1 = *reinterpret_cast<const float4 *>(cst + H + ch0 + 4);
            float t[8];
            t[0] = __builtin_fmaf(f[0].x - f[2].x, c0.x, d0.x); t[1] = __builtin_fmaf(f[0].y - f[2].y, c0.y, d0.y);
            t[2] = __builtin_fmaf(f[0].z - f[2].z, c0.z, d0.z); t[3] = __builtin_fmaf(f[0].w - f[2].w, c0.w, d0.w);
            t[4] = __builtin_fmaf(f[1].x - f[3].x, c1.x, d1.x); t[5] = __builtin_fmaf(f[1].y - f[3].y, c1.y, d1.y);
            t[6] = __builtin_fmaf(f[1].z - f[3].z, c1.z, d1.z); t[7] = __builtin_fmaf(f[1].w - f[3].w, c1.w, d1.w);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = t[e] > 0.0f ? t[e] : 0.0f;
            return make_frag<2>(t);
        };
        if (AHEADA) load_uv(0, raw);
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
                if (AHEADA) load_uv(kc + 1 == NKC ? 0 : kc + 1, rnx);           // (a1 does not depend on the column block)
                Frag<2> af[2];
                if (!AHEADA) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s] = a1_frag(ub, uo, vb, vo, pack1, H, kc * 32 + s * 16 + h * 8);
                } else {
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s] = a1_lds(raw[s], kc * 32 + s * 16 + h * 8);
                }
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc[j] = mfma<2>(af[s], chunk_frag<CT>(slot, j, s, lane), acc[j]);
                if (!RES) {
                    // the other slot held chunk seq-1: every wave finished reading it before the barrier
                    // that ended the previous iteration, so it can be overwritten now
                    store_chunk<CT>(pre, wl + ((seq + 1) & 1) * Chunk<CT>::WORDS);
                    lds_barrier();
                    ++seq;
                }
                if (AHEADA) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int e = 0; e < 4; ++e) raw[s][e] = rnx[s][e];
                }
            }
            // epilogue of this column block: lane = channel, register = row acc_row(i, h) of the tile
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int col = (cb * CT + j) * 32 + r;
                const float sg = sgn2[col];
                float s1 = 0.0f, s2 = 0.0f;
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float y = acc[j][i];
                    const float wy = (float)ri_mult(meta[i]) * y;            // the row stands for `mult` positions
                    s1 += wy;
                    s2 = __builtin_fmaf(wy, y, s2);
                    v[i] = y * sg;
                }
                s1 += shfl_xor32(s1);
                s2 += shfl_xor32(s2);
                if (h == 0) {                            // (a tile beyond the map has multiplicity 0 everywhere)
                    mine[col] += s1;                     // this wave owns its row of `st`: plain update
                    mine[O + col] += s2;
                }
                // the pool, query by query (padding rows carry query 255: never selected); rows ascend with
                // the slot, so the first maximum is the lowest slot
#pragma unroll 1
                for (int jq = 0; jq < nq; ++jq) {
                    float best = -__builtin_inff();
                    int bi = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bool up = ri_q(meta[i]) == jq && v[i] > best;
                        best = up ? v[i] : best;
                        bi = up ? (int)meta[i] : bi;
                    }
                    int bslot = ri_slot((unsigned)bi);
                    const float ob = shfl_xor32(best);
                    const int op = shfl_xor32i(bslot);
                    if (ob > best || (ob == best && op < bslot)) { best = ob; bslot = op; }
                    if (h == 0) {
                        ysel0[(unsigned)(jq * O + col)] = best * sg;
                        ksel0[(unsigned)(jq * O + col)] = (unsigned char)bslot;
                    }
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
//   GU[tile * 32 + row, :] = g_u summed over the row's positions (dense rows; the per-point sums are taken
//   later through the index stage's inverse map, in a fixed order: no float atomics);
//   HA[q, :] = sum_k g_u, HB[q, :] = sum_k yhat1.
//   img: B image of Z = [W2 ; Qm] ((O + H) x H), CT = min(4, H/32) column tiles per block.
//   pack1 = {scale1, shift1, mean1, invstd1}[H].
// ------------------------------------------------------------------------------------------
template <int H, int O, int CT, bool RES, bool WG, int OCC>
__global__ __launch_bounds__(256, OCC) void wide_bwd_main_kernel(WideArgs a, const uint4 *__restrict__ img,
                                                                 const float *__restrict__ pack1,
                                                                 const float *__restrict__ evec,
                                                                 const float *__restrict__ goa,
                                                                 const unsigned char *__restrict__ ksel,
                                                                 float *__restrict__ GU, float *__restrict__ HA,
                                                                 float *__restrict__ HB, float *__restrict__ part,
                                                                 float *__restrict__ Rpart) {
    constexpr int NKS = O / 32, NKC = (O + H) / 32, NCB = H / (32 * CT), NCH = NKC * NCB;
    constexpr int SLOTS = RES ? NCH : 2;
    static_assert(!WG || (H == 32 && O == 64 && CT == 1), "the fused weight-gradient products are for H = 32");
    constexpr int A1LD = 36;                                          // row stride of the a1 tile in LDS (16-byte rows, banks spread)
    constexpr int RW = (O + H) * H + H, RWP = (O + H) * (H + 1) + H;  // the workgroup's {R_S ; Gram ; suma}: dense / padded rows
    extern __shared__ uint4 dyn[];
    uint4 *wl = dyn;
    float *st = reinterpret_cast<float *>(dyn + SLOTS * Chunk<CT>::WORDS);   // [4 waves][2*H]
    float *wgl = st + WIDE_WAVES * 2 * H;                                    // WG: [4 waves] a1 tiles, later the fold
    const int lane = lane_id(), r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int step = gridDim.x * WIDE_WAVES;
    const int first = blockIdx.x * WIDE_WAVES + w;                   // wave-uniform tile numbers
    TileHead head = load_head(a, first < a.ntiles ? first : a.ntiles - 1, r);
    for (int e = threadIdx.x; e < WIDE_WAVES * 2 * H; e += 256) st[e] = 0.0f;
    if (RES) {
        for (int ci = 0; ci < NCH; ++ci) stage_chunk<CT>(img, ci, wl + ci * Chunk<CT>::WORDS);
    }
    __syncthreads();
    float *mine = st + w * 2 * H;
    float *a1s = wgl + w * 32 * A1LD;                                // this wave's a1 tile [row][mid]
    const int nt = tm_tiles(a);
    const int rounds = (nt + step - 1) / step;
    // streaming blocks: chunk c is read from slot c & 1; chunk c + 1 sits in one register set (requested during step c - 1,
    // stored into the other slot at the end of step c), chunk c + 2 is requested into the other set at the start of step c
    static_assert(RES || Chunk<CT>::WORDS == 1024, "the streaming blocks move four words per thread and chunk");
    // (eight named registers, not two arrays: one of two arrays -- or of two ChunkRegs -- stayed in scratch memory)
    uint4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;
    pa0 = pa1 = pa2 = pa3 = pb0 = pb1 = pb2 = pb3 = make_uint4(0u, 0u, 0u, 0u);
    if (!RES) {
        stage_chunk<CT>(img, 0, wl);
        const uint4 *__restrict__ src = img + (size_t)(NCH > 1 ? 1 : 0) * Chunk<CT>::WORDS + threadIdx.x;
        pb0 = src[0]; pb1 = src[256]; pb2 = src[512]; pb3 = src[768];
        __syncthreads();
    }
    // fused weight-gradient products (WG): Gram = a1^T diag(mult) a1 (MFMA, H x H), R_S = S^T a1 (O x H: lane =
    // output channel c, exact fp32 sums of the pooled rows of a1), suma
    f32x16 gram;
    float sacc[WG ? H : 1];
    float suma = 0.0f;
    if (WG) {
#pragma unroll
        for (int i = 0; i < 16; ++i) gram[i] = 0.0f;
#pragma unroll
        for (int m = 0; m < H; ++m) sacc[m] = 0.0f;
    }
    for (int it = 0; it < rounds; ++it) {
        const int tile_raw = first + it * step;
        const bool valid = tile_raw < nt;
        const int tile = valid ? tile_raw : 0;
        const TileHead cur = valid ? head : no_tile();
        {
            const int nx = tile_raw + step;
            head = load_head(a, nx < a.ntiles ? nx : a.ntiles - 1, r);
        }
        const int q0 = __builtin_amdgcn_readfirstlane(cur.q0);
        const int nq = __builtin_amdgcn_readfirstlane((int)(cur.info >> 24));
        const bool live = ri_mult(cur.info) != 0;
        const unsigned qloc = live ? ri_q(cur.info) : 0u;
        const int rslot = ri_slot(cur.info);
        const float wrow = (float)ri_mult(cur.info);
        // wave-uniform bases; per-lane offsets stay 32-bit
        const float *__restrict__ ub = a.U + (size_t)(q0 / a.m) * a.n * H;
        const float *__restrict__ vb = a.V + (size_t)q0 * H;
        const float *__restrict__ gb = goa + (size_t)q0 * O;
        const unsigned char *__restrict__ kb = ksel + (size_t)q0 * O;
        float *__restrict__ gub = GU + (size_t)tile * 32 * H;
        float *__restrict__ hab = HA + (size_t)q0 * H;
        float *__restrict__ hbb = HB + (size_t)q0 * H;
        const unsigned uo = (unsigned)cur.nn * H, vo = qloc * H, go = qloc * O;
        // record and neighbour of every accumulator row of this lane (row p's sit in lane p)
        unsigned meta[16];
        row_meta(cur.info, h, meta);
        // (row offsets into U and V of the accumulator rows: kept in registers across the MFMA chain only where the
        // epilogue's loads are issued ahead of it, CT <= 2; the wide blocks recompute them in the epilogue)
        unsigned urow[CT <= 2 ? 16 : 1], vrow[CT <= 2 ? 16 : 1];
        if (CT <= 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                urow[i] = (unsigned)__builtin_amdgcn_ds_bpermute(acc_row(i, h) << 2, (int)uo);
                vrow[i] = (ri_mult(meta[i]) ? ri_q(meta[i]) : 0u) * H;
            }
        }
#pragma unroll 1
        for (int cb = 0; cb < NCB; ++cb) {
            // this lane's y1 operands of the epilogue (lane = mid channel): issued before the MFMA chain
            // (a tile of ONE query -- dense neighbourhoods pack that way -- has one row of V: one load, not 16)
            float u[CT <= 2 ? CT : 1][16], vv[CT <= 2 ? CT : 1][16];
            if (CT <= 2) {
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const unsigned mid = (cb * CT + j) * 32 + r;
#pragma unroll
                    for (int i = 0; i < 16; ++i) u[j][i] = ub[urow[i] + mid];
                    if (nq == 1) {
                        const float v1 = vb[mid];
#pragma unroll
                        for (int i = 0; i < 16; ++i) vv[j][i] = v1;
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) vv[j][i] = vb[vrow[i] + mid];
                    }
                }
            }
            f32x16 acc[CT];
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
            // evec == NULL: BatchNorm-2 normalises with its running statistics (eval mode, e.g. the classifier in the
            // AdaptPoint feedback pass): D2 = E2 = 0, so Qm = 0 and the a1 Qm third of the chain is skipped
            const int kend = evec ? NKC : NKS;
            auto load_raw = [&](int kc, RawA &rw) {
                if (kc < NKS) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned c0 = go + kc * 32 + s * 16 + h * 8;
                        rw.f[s][0] = *reinterpret_cast<const float4 *>(gb + c0);
                        rw.f[s][1] = *reinterpret_cast<const float4 *>(gb + (c0 + 4));
                        rw.kk[s] = *reinterpret_cast<const uint2 *>(kb + c0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 2; ++s) a1_raw(ub, uo, vb, vo, pack1, H, (kc - NKS) * 32 + s * 16 + h * 8, rw.f[s]);
                }
            };
            // (one wave per SIMD has the registers for two chunks of words; the two-waves-per-SIMD variants request
            // and consume in the same step, as before: their tiles hold fewer chunks and a partner wave covers the wait)
            constexpr bool AHEAD = OCC == 1;
            static_assert(RES || AHEAD, "the streaming blocks run one wave per SIMD");
            auto frags_from = [&](int kc, const RawA &rw, Frag<2> (&af)[2]) {
                if (kc < NKS) {            // rows of S: the upstream gradient at the pooled slot
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float4 g0 = rw.f[s][0], g1 = rw.f[s][1];
                        const uint2 kk = rw.kk[s];
                        float t[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned sel = ((e < 4 ? kk.x : kk.y) >> (8 * (e & 3))) & 0xffu;
                            t[e] = (live && sel == (unsigned)rslot) ? t[e] : 0.0f;
                        }
                        af[s] = make_frag<2>(t);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s] = a1_from_raw(rw.f[s], wrow);
                }
            };
            if constexpr (!RES) {
                // Two steps per iteration with the roles of the register sets fixed per step (kend is even): nothing is
                // copied at the back edge -- a copy of a load's destination waits for the load, i.e. for the prefetch
                // issued a moment ago.  Per step: request chunk + 2 (registers) and the A words of step + 1, build this
                // step's fragments from the words requested one step ago, MFMAs, store chunk + 1 into the other slot.
                auto after = [&](int cbx, int kcx, int &cb2, int &kc2) {      // the chunk behind (cbx, kcx) in the cyclic sequence
                    if (kcx + 1 < kend) { cb2 = cbx; kc2 = kcx + 1; }
                    else { cb2 = cbx + 1 < NCB ? cbx + 1 : 0; kc2 = 0; }
                };
                // (the two steps are written out with the register sets NAMED: handed to a step function by reference, or
                // selected inside a lambda, the compiler kept the sets in scratch memory)
                RawA raw_a, raw_b;
// CT_ / NT_: what the step's A words are and what the next step's are -- 0 rows of S, 1 rows of a1, 2 none -- as
// compile-time constants: a load under a run-time condition makes the paths of a step differ in their loads in flight,
// and the compiler then waits for ALL of them (vmcnt(0)) where the paths meet, the prefetches included
#define APN_STREAM_STEP(KC, I0, I1, I2, I3, F0, F1, F2, F3, CURR, NXTR, PAR, CT_, NT_)                              \
    {                                                                                                               \
        int c1_, k1_, c2_, k2_;                                                                                     \
        after(cb, (KC), c1_, k1_);                                                                                  \
        after(c1_, k1_, c2_, k2_);                                                                                  \
        {                                                                                                           \
            const uint4 *__restrict__ src_ = img + (size_t)(c2_ * NKC + k2_) * Chunk<CT>::WORDS + threadIdx.x;       \
            I0 = src_[0]; I1 = src_[256]; I2 = src_[512]; I3 = src_[768];                                            \
        }                                                                                                           \
        if ((NT_) == 0) load_raw_s((KC) + 1, NXTR);                                                                 \
        if ((NT_) == 1) load_raw_a((KC) + 1, NXTR);                                                                 \
        const uint4 *slot_ = wl + (PAR) * Chunk<CT>::WORDS;                                                         \
        Frag<2> af_[2];                                                                                             \
        if ((CT_) == 0) frags_s(CURR, af_);                                                                         \
        else frags_a(CURR, af_);                                                                                    \
        _Pragma("unroll") for (int j = 0; j < CT; ++j)                                                              \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                           \
                acc[j] = mfma<2>(af_[s], chunk_frag<CT>(slot_, j, s, lane), acc[j]);                                 \
        {                                                                                                           \
            uint4 *dst_ = wl + (1 - (PAR)) * Chunk<CT>::WORDS + threadIdx.x;                                         \
            dst_[0] = F0; dst_[256] = F1; dst_[512] = F2; dst_[768] = F3;                                            \
        }                                                                                                           \
        lds_barrier();                                                                                              \
    }
#define APN_STEP_EVEN(KC, CT_, NT_) APN_STREAM_STEP(KC, pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, raw_a, raw_b, 0, CT_, NT_)
#define APN_STEP_ODD(KC, CT_, NT_) APN_STREAM_STEP(KC, pb0, pb1, pb2, pb3, pa0, pa1, pa2, pa3, raw_b, raw_a, 1, CT_, NT_)
                auto load_raw_s = [&](int kc, RawA &rw) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned c0 = go + kc * 32 + s * 16 + h * 8;
                        rw.f[s][0] = *reinterpret_cast<const float4 *>(gb + c0);
                        rw.f[s][1] = *reinterpret_cast<const float4 *>(gb + (c0 + 4));
                        rw.kk[s] = *reinterpret_cast<const uint2 *>(kb + c0);
                    }
                };
                auto load_raw_a = [&](int kc, RawA &rw) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) a1_raw(ub, uo, vb, vo, pack1, H, (kc - NKS) * 32 + s * 16 + h * 8, rw.f[s]);
                };
                auto frags_s = [&](const RawA &rw, Frag<2> (&af)[2]) { frags_from(0, rw, af); };
                auto frags_a = [&](const RawA &rw, Frag<2> (&af)[2]) { frags_from(NKS, rw, af); };
                static_assert(NKS % 2 == 0 && NKC % 2 == 0 && NKS >= 2 && NKC - NKS >= 2, "steps come in pairs");
                load_raw_s(0, raw_a);
                // the rows of S: chunks 0 .. NKS - 1
#pragma unroll 1
                for (int kc = 0; kc < NKS - 2; kc += 2) {
                    APN_STEP_EVEN(kc, 0, 0)
                    APN_STEP_ODD(kc + 1, 0, 0)
                }
                APN_STEP_EVEN(NKS - 2, 0, 0)
                if (evec) {
                    APN_STEP_ODD(NKS - 1, 0, 1)
                    // the rows of a1: chunks NKS .. NKC - 1
#pragma unroll 1
                    for (int kc = NKS; kc < NKC - 2; kc += 2) {
                        APN_STEP_EVEN(kc, 1, 1)
                        APN_STEP_ODD(kc + 1, 1, 1)
                    }
                    APN_STEP_EVEN(NKC - 2, 1, 1)
                    APN_STEP_ODD(NKC - 1, 1, 2)
                } else {
                    APN_STEP_ODD(NKS - 1, 0, 2)
                }
#undef APN_STEP_EVEN
#undef APN_STEP_ODD
#undef APN_STREAM_STEP
            } else {
            RawA cur_raw;
            if (AHEAD) load_raw(0, cur_raw);
#pragma unroll 1
            for (int kc = 0; kc < kend; ++kc) {
                const int ci = cb * NKC + kc;
                const uint4 *slot = wl + ci * Chunk<CT>::WORDS;
                // this chunk's A operand from the words requested one chunk ago; the next chunk's are requested now and
                // arrive behind the MFMAs
                RawA nxt;
                Frag<2> af[2];
                if (AHEAD) {
                    if (kc + 1 < kend) load_raw(kc + 1, nxt);
                    frags_from(kc, cur_raw, af);
                } else if (kc < NKS) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned c0 = go + kc * 32 + s * 16 + h * 8;
                        const float4 g0 = *reinterpret_cast<const float4 *>(gb + c0);
                        const float4 g1 = *reinterpret_cast<const float4 *>(gb + (c0 + 4));
                        const uint2 kk = *reinterpret_cast<const uint2 *>(kb + c0);
                        float t[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned sel = ((e < 4 ? kk.x : kk.y) >> (8 * (e & 3))) & 0xffu;
                            t[e] = (live && sel == (unsigned)rslot) ? t[e] : 0.0f;
                        }
                        af[s] = make_frag<2>(t);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        af[s] = a1_frag(ub, uo, vb, vo, pack1, H, (kc - NKS) * 32 + s * 16 + h * 8, wrow, true);
                }
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc[j] = mfma<2>(af[s], chunk_frag<CT>(slot, j, s, lane), acc[j]);
                if (AHEAD) cur_raw = nxt;
            }
            }
            // epilogue: lane = mid channel, register = row of the tile
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const unsigned mid = (cb * CT + j) * 32 + r;
                const float sc = pack1[mid], sh = pack1[H + mid], mu = pack1[2 * H + mid], iv = pack1[3 * H + mid];
                const float ev = evec ? evec[mid] : 0.0f;
                float t1 = 0.0f, t2 = 0.0f;
                float gu[16], yw[16];
                float a1r[WG ? 16 : 1];
                float uw[CT > 2 ? 16 : 1], vw[CT > 2 ? 16 : 1];
                if (CT > 2) {                     // wide blocks: this column tile's operands, all loads in flight
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        uw[i] = ub[(unsigned)__builtin_amdgcn_ds_bpermute(acc_row(i, h) << 2, (int)uo) + mid];
                    if (nq == 1) {
                        const float v1 = vb[mid];
#pragma unroll
                        for (int i = 0; i < 16; ++i) vw[i] = v1;
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) vw[i] = vb[(ri_mult(meta[i]) ? ri_q(meta[i]) : 0u) * H + mid];
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float ui, vi;
                    if (CT <= 2) { ui = u[j][i]; vi = vv[j][i]; }
                    else { ui = uw[i]; vi = vw[i]; }
                    const float wgt = (float)ri_mult(meta[i]);
                    const float y1 = ui - vi;
                    const float yh = (y1 - mu) * iv;
                    const float pre1 = __builtin_fmaf(y1, sc, sh);
                    // the row's `mult` positions share a1; only one of them can hold a pooled slot (S)
                    const float g = pre1 > 0.0f ? __builtin_fmaf(ev, wgt, acc[j][i]) : 0.0f;
                    t1 += g;
                    t2 = __builtin_fmaf(g, yh, t2);
                    if (wgt != 0.0f) gub[(unsigned)(acc_row(i, h) * H) + mid] = g;
                    gu[i] = g;
                    yw[i] = wgt * yh;
                    if (WG) {
                        a1r[i] = pre1 > 0.0f ? pre1 : 0.0f;
                        a1s[acc_row(i, h) * A1LD + r] = a1r[i];
                    }
                }
                t1 += shfl_xor32(t1);
                t2 += shfl_xor32(t2);
                if (h == 0) {
                    mine[mid] += t1;
                    mine[H + mid] += t2;
                }
#pragma unroll 1
                for (int jq = 0; jq < nq; ++jq) {
                    float ha = 0.0f, hb = 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bool in = ri_q(meta[i]) == jq;     // padding rows carry query 255
                        ha += in ? gu[i] : 0.0f;
                        hb += in ? yw[i] : 0.0f;
                    }
                    ha += shfl_xor32(ha);
                    hb += shfl_xor32(hb);
                    if (h == 0) {
                        hab[(unsigned)(jq * H) + mid] = ha;
                        hbb[(unsigned)(jq * H) + mid] = hb;
                    }
                }
                if (WG) {
                    // Gram += (mult a1)^T a1: a1 in the accumulator layout (lane = mid, register = row) IS an MFMA
                    // operand whose k index runs over the tile's rows in register order, the same on both sides
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float tb[8], ta[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            tb[e] = a1r[8 * s + e];
                            ta[e] = tb[e] * (float)ri_mult(meta[8 * s + e]);
                            suma += ta[e];
                        }
                        gram = mfma<2>(make_frag<2>(ta), make_frag<2>(tb), gram);
                    }
                    // R_S[c][:] += goa[q][c] a1[row of (q, ksel[q][c])][:], lane = c: the pooled row of a1 comes from
                    // this wave's LDS tile (written above; LDS executes a wave's instructions in order)
                    float gv_nx = gb[lane];                      // (one query ahead: no round trip per query)
                    int ks_nx = (int)kb[lane];
#pragma unroll 1
                    for (int jq = 0; jq < nq; ++jq) {
                        const unsigned long long mine_q = __ballot(lane < 32 && live && ri_q(cur.info) == (unsigned)jq);
                        const int row0 = __builtin_ctzll(mine_q);                    // first row of query jq in the tile
                        const float gv = gv_nx;
                        const int kr = row0 + ks_nx;
                        if (jq + 1 < nq) {
                            gv_nx = gb[(unsigned)((jq + 1) * O) + lane];
                            ks_nx = (int)kb[(unsigned)((jq + 1) * O) + lane];
                        }
                        const float4 *__restrict__ ar = reinterpret_cast<const float4 *>(a1s + kr * A1LD);
#pragma unroll
                        for (int m4 = 0; m4 < H / 4; ++m4) {
                            const float4 x = ar[m4];
                            sacc[4 * m4] = __builtin_fmaf(gv, x.x, sacc[4 * m4]);
                            sacc[4 * m4 + 1] = __builtin_fmaf(gv, x.y, sacc[4 * m4 + 1]);
                            sacc[4 * m4 + 2] = __builtin_fmaf(gv, x.z, sacc[4 * m4 + 2]);
                            sacc[4 * m4 + 3] = __builtin_fmaf(gv, x.w, sacc[4 * m4 + 3]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * H; e += 256)
        part[(size_t)blockIdx.x * 2 * H + e] = (st[e] + st[2 * H + e]) + (st[4 * H + e] + st[6 * H + e]);
    if (WG) {
        // the workgroup's {R_S ; Gram ; suma}: the four waves fold through LDS (rows padded to H + 1: lane c writes
        // row c), one partial row per workgroup
        float *mywr = wgl + w * RWP;
        suma += shfl_xor32(suma);
#pragma unroll
        for (int m = 0; m < H; ++m) mywr[lane * (H + 1) + m] = sacc[m];
#pragma unroll
        for (int i = 0; i < 16; ++i) mywr[(O + acc_row(i, h)) * (H + 1) + r] = gram[i];
        if (h == 0) mywr[(O + H) * (H + 1) + r] = suma;
        __syncthreads();
        for (int e = threadIdx.x; e < RW; e += 256) {
            const int row = e / H, col = e - row * H;
            const int src = row < O + H ? row * (H + 1) + col : (O + H) * (H + 1) + col;
            Rpart[(size_t)blockIdx.x * RW + e] = (wgl[src] + wgl[RWP + src]) + (wgl[2 * RWP + src] + wgl[3 * RWP + src]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight-gradient products over the positions:
//   R[(O + H), H] = [S^T ; a1^T] a1   (rows 0..O-1: sum_q goa[q,c] a1[q, ksel[q,c], :], the sparse
//   part of dL/dW2; rows O..: the Gram matrix sum a1^T a1),  suma[H] = sum a1.
// Workgroup = NW waves; wave w owns the 32-row block rb = blockIdx.y * NW + w and all H columns
// (H/32 accumulator tiles); the workgroup's range of query tiles is blockIdx.x of gridDim.x
// (split-K).  Per tile the a1 operand (k = position, column = mid channel) is built ONCE by the
// workgroup into LDS in fragment order; its a1^T rows are the same fragments.
// Output: Rpart[split][(O+H) H + H] = {R, suma} (one row per split, summed by the caller in float64).
// ------------------------------------------------------------------------------------------
// Round 4: the tile's a1 stays [position][mid channel] in LDS, as it is loaded: every thread reads 16 bytes of four
// channels of a row of U and of V (the fragment-order build read one float per load: 16 + 16 four-byte loads per
// fragment-lane, 256 vector-memory instructions per tile and workgroup at H = 256 -- their issue was the tile's time),
// and the MFMA fragments, whose k index runs over the positions, come out by TRANSPOSED reads (ds_read_b64_tr_b16: a
// 16-lane group reads 4 positions x 16 channels, every lane receives its channel's 4 positions).  Same bf16 hi / lo
// values, same products in the same order: R is bit-identical to the fragment-order kernel's.
template <int H>
struct WgTile {
    // bf16 per position row: + 64 bytes.  A transposed read is banked per 32-LANE half (four position rows x two 16-channel
    // groups of 32 bytes): a row must shift by 64 bytes modulo the 256-byte bank row for the eight pieces to cover it once
    // (with + 32 bytes, until round 5, the second group of row q fell on the first of row q + 1: every such read took 2x)
    static constexpr int TR = H + 32;
    static constexpr int PLANE = 32 * TR;
    static constexpr int BYTES = 2 * 2 * PLANE * 2;   // {a1, mult a1} x {hi, lo}
};

// fragment (rows / columns blk .. blk + 31 = mid channels, lane r its channel; k = positions 16 s + 8 h .. + 7)
template <int H>
__device__ __forceinline__ Frag<2> wg_fragment(const __bf16 *tile, int blk, int s, int r, int h) {
    typedef short tr_s4 __attribute__((ext_vector_type(4)));
    typedef short tr_s8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) tr_s4 tr_lds;
    const int li = r & 15;
    const __bf16 *src = tile + (s * 16 + h * 8 + (li >> 2)) * WgTile<H>::TR + blk + (r & 16) + 4 * (li & 3);
    Frag<2> f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const tr_s4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds *)(src + p * WgTile<H>::PLANE));
        const tr_s4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds *)(src + p * WgTile<H>::PLANE + 4 * WgTile<H>::TR));
        const tr_s8 v8 = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        f.p[p] = __builtin_bit_cast(bf16x8, v8);
    }
    return f;
}

template <int H, int O, int NW>
__global__ __launch_bounds__(NW * 64) void wide_wgrad_kernel(WideArgs a, const float *__restrict__ pack1,
                                                             const float *__restrict__ goa,
                                                             const unsigned char *__restrict__ ksel,
                                                             float *__restrict__ Rpart) {
    constexpr int NJ = H / 32, NT = NW * 64;
    constexpr int C4 = H / 4, PG = NT / C4, PASSES = (32 + PG - 1) / PG;     // float4 per row, position groups, positions per thread
    static_assert(NT % C4 == 0, "whole rows per pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_raw[];
    __bf16 *ta = reinterpret_cast<__bf16 *>(wg_raw);                           // a1       [hi | lo][position][TR]
    __bf16 *tw = ta + 2 * WgTile<H>::PLANE;                                    // mult a1  (the Gram product's other side)
    __shared__ unsigned rinfo[32];
    const int lane = lane_id(), w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int rb = blockIdx.y * NW + w;                                // this wave's row block
    const bool rb_ok = rb < (O + H) / 32;
    const bool s_rows = rb < O / 32;
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    const int c4 = threadIdx.x % C4, pg = threadIdx.x / C4;            // this thread's four channels, its position group
    const float4 sc = *reinterpret_cast<const float4 *>(pack1 + 4 * c4), sh = *reinterpret_cast<const float4 *>(pack1 + H + 4 * c4);
    float suma4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nt = tm_tiles(a);
    const int *__restrict__ tq0 = tm_tq0(a);
    const unsigned *__restrict__ rows = tm_rows(a);
    const int per = (nt + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per, t1 = min(nt, t0 + per);
    // A tile's row records and neighbours are requested ONE TILE AHEAD, by every thread for the positions it builds (and by
    // threads 0..31 for the LDS copy the MFMA phase selects with): the build's loads of U and V then start at the top of
    // the tile instead of behind {records -> LDS -> barrier}, and one barrier per tile goes.
    unsigned pinf[PASSES], linf = 0u;
    int pnbr[PASSES];
    auto request = [&](int tl) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int pos = pg + PG * ps < 32 ? pg + PG * ps : 31;
            pinf[ps] = rows[(size_t)tl * 32 + pos];
            pnbr[ps] = tm_nn(a)[(size_t)tl * 32 + pos];
        }
        linf = rows[(size_t)tl * 32 + (threadIdx.x & 31)];
    };
    if (t0 < t1) request(t0);
    for (int tile = t0; tile < t1; ++tile) {
        const int q0 = tq0[tile];
        const float *__restrict__ ub = a.U + (size_t)(q0 / a.m) * a.n * H;
        float4 uu[PASSES], vv[PASSES];
        float multv[PASSES];
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {                          // (all loads of the tile, then the arithmetic)
            const unsigned info = ri_mult(pinf[ps]) ? pinf[ps] : (pinf[ps] & ~0xffu);     // padding: query 0 of the tile
            uu[ps] = *reinterpret_cast<const float4 *>(ub + (size_t)pnbr[ps] * H + 4 * c4);
            vv[ps] = *reinterpret_cast<const float4 *>(a.V + (size_t)(q0 + ri_q(info)) * H + 4 * c4);
            multv[ps] = (float)ri_mult(info);
        }
        __syncthreads();                                               // previous tile's reads are done
        if (threadIdx.x < 32) rinfo[threadIdx.x] = ri_mult(linf) ? linf : (linf & ~0xffu);
        request(tile + 1 < t1 ? tile + 1 : tile);
        {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int pos = pg + PG * ps;
                if (pos < 32) {
                    const float mult = multv[ps];
                    const float y[4] = {uu[ps].x - vv[ps].x, uu[ps].y - vv[ps].y, uu[ps].z - vv[ps].z, uu[ps].w - vv[ps].w};
                    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    bf16x4 ah, al, wh, wl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = __builtin_fmaf(y[e], scv[e], shv[e]);
                        const float t = v > 0.0f ? v : 0.0f, tw_ = t * mult;
                        suma4[e] += tw_;
                        const __bf16 th = (__bf16)t, wh_ = (__bf16)tw_;
                        ah[e] = th; al[e] = (__bf16)(t - (float)th);
                        wh[e] = wh_; wl[e] = (__bf16)(tw_ - (float)wh_);
                    }
                    const int o = pos * WgTile<H>::TR + 4 * c4;
                    *reinterpret_cast<bf16x4 *>(ta + o) = ah;
                    *reinterpret_cast<bf16x4 *>(ta + WgTile<H>::PLANE + o) = al;
                    *reinterpret_cast<bf16x4 *>(tw + o) = wh;
                    *reinterpret_cast<bf16x4 *>(tw + WgTile<H>::PLANE + o) = wl;
                }
            }
        }
        __syncthreads();
        if (!rb_ok) continue;
        Frag<2> af[2];
        if (s_rows) {                   // A = S^T: row = channel c of this lane, k = row of the tile
            const int c = rb * 32 + r;
            // (all sixteen slots and gradients requested before the first is used: `mult && ksel == slot ? goa : 0` was a
            // chain of up to 32 dependent loads per lane and tile -- padding rows name query 0 of the tile: valid addresses)
            const int nq = __builtin_amdgcn_readfirstlane((int)(rinfo[0] >> 24));      // (row 0 holds the tile's query count)
            if (nq == 1) {
                // a tile of ONE query (dense neighbourhoods pack that way: stages 3-4): one slot and one gradient per
                // lane, not sixteen copies of each (32 vector-memory instructions per wave and tile)
                const size_t qc = (size_t)q0 * O + c;
                const int k1 = (int)ksel[qc];
                const float g1 = goa[qc];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned info = rinfo[s * 16 + h * 8 + e];
                        t[e] = (ri_mult(info) != 0 && k1 == ri_slot(info)) ? g1 : 0.0f;
                    }
                    af[s] = make_frag<2>(t);
                }
            } else {
                unsigned char ks[2][8];
                float gv[2][8];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const size_t qc = (size_t)(q0 + ri_q(rinfo[s * 16 + h * 8 + e])) * O + c;
                        ks[s][e] = ksel[qc];
                        gv[s][e] = goa[qc];
                    }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned info = rinfo[s * 16 + h * 8 + e];
                        t[e] = (ri_mult(info) != 0 && (int)ks[s][e] == ri_slot(info)) ? gv[s][e] : 0.0f;
                    }
                    af[s] = make_frag<2>(t);
                }
            }
        } else {                        // A = (mult a1)^T rows of mid block rb - O/32
            const int j0 = rb - O / 32;
#pragma unroll
            for (int s = 0; s < 2; ++s) af[s] = wg_fragment<H>(tw, j0 * 32, s, r, h);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[j] = mfma<2>(af[s], wg_fragment<H>(ta, j * 32, s, r, h), acc[j]);
    }
    if (rb_ok) {
        float *__restrict__ out = Rpart + (size_t)blockIdx.x * ((O + H) * H + H) + (size_t)rb * 32 * H;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) out[(size_t)acc_row(i, h) * H + j * 32 + r] = acc[j][i];
    }
    // sum a1: the position groups of one mid channel, fixed order
    if (blockIdx.y == 0) {
        float *sred = reinterpret_cast<float *>(wg_raw);               // [PG][H]
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) sred[pg * H + 4 * c4 + e] = suma4[e];
        __syncthreads();
        for (int mid = threadIdx.x; mid < H; mid += NT) {
            float v = 0.0f;
#pragma unroll 1
            for (int g = 0; g < PG; ++g) v += sred[g * H + mid];
            Rpart[(size_t)blockIdx.x * ((O + H) * H + H) + (size_t)(O + H) * H + mid] = v;
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

static bool wide_args_ok(int b, int n, int m, const void *U, const void *V, const void *idx, const void *tmap) {
    return b > 0 && n > 0 && m > 0 && (long long)b * m <= 0x7fffffffLL / 64 && U && V && idx && tmap;
}

template <int H>
static constexpr int fwd_ct() { return 2 * H / 32 >= 4 ? 4 : 2 * H / 32; }     // column tiles per block, O = 2H
template <int H>
static constexpr int bwd_ct() { return H / 32 >= 4 ? 4 : H / 32; }

// whole image resident in LDS when it and the statistics rows fit comfortably (two workgroups per CU)
template <int H>
#ifndef APN_FWD_RES_BYTES
#define APN_FWD_RES_BYTES (48 * 1024)
#endif
static constexpr bool fwd_res() { return (size_t)(H / 32) * (2 * H / (32 * fwd_ct<H>())) * Chunk<fwd_ct<H>()>::WORDS * 16 <= APN_FWD_RES_BYTES; }
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
                                  const int *idx, const int *tmap, float *part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx, tmap) || !part) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx, tmap};
    const int grid = wide_grid(a.ntiles);
    APN_WIDE_DISPATCH(c_mid, hipLaunchKernelGGL((wide_stats1_kernel<H>), dim3(grid), dim3(256), 0,
                                                (hipStream_t)stream, a, part));
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_fwd_main(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                    const int *idx, const int *tmap, const void *w2_image, const float *pack1,
                                    const float *sgn2, float *ysel, void *ksel, float *part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx, tmap) || !wide_shape_ok(c_mid, c_out)) return APN_EINVAL;
    if (!w2_image || !pack1 || !sgn2 || !ysel || !ksel || !part) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx, tmap};
    const int grid = wide_grid(a.ntiles);
    APN_WIDE_DISPATCH(c_mid, {
        constexpr int O = 2 * H, CT = fwd_ct<H>();
        constexpr bool RES = fwd_res<H>();
        constexpr int NCH = (H / 32) * (O / (32 * CT));
        const size_t lds = (size_t)(RES ? NCH : 2) * Chunk<CT>::WORDS * 16 + (size_t)WIDE_WAVES * 2 * O * 4 +
                           (H >= 128 ? (size_t)2 * H * 4 : 0);
        if (lds > 64 * 1024) {
            static DynLdsOnce configured;            // (per instantiation and device)
            if (hipError_t e = set_dyn_lds(configured, (const void *)wide_fwd_main_kernel<H, O, CT, RES>, (int)lds))
                return (int)e;
        }
        hipLaunchKernelGGL((wide_fwd_main_kernel<H, O, CT, RES>), dim3(grid), dim3(256), lds,
                           (hipStream_t)stream, a, (const uint4 *)w2_image, pack1, sgn2, ysel,
                           (unsigned char *)ksel, part);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_wide_bwd_main(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                    const int *idx, const int *tmap, const void *z_image, const float *pack1,
                                    const float *evec, const float *goa, const void *ksel, float *GU,
                                    float *HA, float *HB, float *part, float *r_part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx, tmap) || !wide_shape_ok(c_mid, c_out)) return APN_EINVAL;
    if (!z_image || !pack1 || !goa || !ksel || !GU || !HA || !HB || !part) return APN_EINVAL;
    if (c_mid == 32 && !r_part) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx, tmap};
    const int grid = wide_grid(a.ntiles);
    APN_WIDE_DISPATCH(c_mid, {
        constexpr int O = 2 * H, CT = bwd_ct<H>();
        constexpr bool RES = bwd_res<H>();
        constexpr bool WG = H == 32;
        constexpr int NCH = ((O + H) / 32) * (H / (32 * CT));
        const size_t lds = (size_t)(RES ? NCH : 2) * Chunk<CT>::WORDS * 16 + (size_t)WIDE_WAVES * 2 * H * 4 +
                           (WG ? (size_t)WIDE_WAVES * ((O + H) * (H + 1) + H) * 4 : 0);
        auto kern = wide_bwd_main_kernel<H, O, CT, RES, WG, (H <= 64 ? 2 : 1)>;       // two waves per SIMD where 256 registers do
        if (lds > 48 * 1024) {
            if (hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
                return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds,
                           (hipStream_t)stream, a, (const uint4 *)z_image, pack1, evec, goa,
                           (const unsigned char *)ksel, GU, HA, HB, part, r_part);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// 1: the weight-gradient products {[S^T ; a1^T] a1, sum a1} come out of apn_sa_wide_bwd_main itself, as
// r_part[apn_sa_wide_grid][(O+H) H + H]; 0: apn_sa_wide_wgrad computes them in its own pass.
extern "C" int apn_sa_wide_wgrad_fused(int c_mid) { return c_mid == 32 ? 1 : 0; }

extern "C" int apn_sa_wide_wgrad_splits(int b, int m, int c_mid) {
    if (b <= 0 || m <= 0 || !wide_shape_ok(c_mid, 2 * c_mid)) return 0;
    const int ntiles = b * m;
    const int groups = (3 * c_mid / 32 + 7) / 8;           // workgroups per split (8 row blocks each)
    int splits = 512 / groups;
    if (splits > ntiles / 4) splits = ntiles / 4;          // at least 4 query tiles per workgroup
    return splits < 1 ? 1 : splits;
}

extern "C" int apn_sa_wide_wgrad(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                 const int *idx, const int *tmap, const float *pack1, const float *goa,
                                 const void *ksel, int splits, float *r_part, void *stream) {
    if (!wide_args_ok(b, n, m, U, V, idx, tmap) || !wide_shape_ok(c_mid, c_out)) return APN_EINVAL;
    if (!pack1 || !goa || !ksel || !r_part || splits < 1) return APN_EINVAL;
    WideArgs a{b * m, n, m, U, V, idx, tmap};
    APN_WIDE_DISPATCH(c_mid, {
        constexpr int O = 2 * H, NRB = (O + H) / 32, NW = NRB < 8 ? NRB : 8;
        const dim3 grid(splits, (NRB + NW - 1) / NW);
        constexpr int lds = WgTile<H>::BYTES > NW * 64 / (H / 4) * H * 4 ? WgTile<H>::BYTES : NW * 64 / (H / 4) * H * 4;
        if (lds > 48 * 1024) {
            static DynLdsOnce configured;            // (per instantiation and device)
            if (hipError_t e = set_dyn_lds(configured, (const void *)wide_wgrad_kernel<H, O, NW>, lds)) return (int)e;
        }
        hipLaunchKernelGGL((wide_wgrad_kernel<H, O, NW>), grid, dim3(NW * 64), lds, (hipStream_t)stream, a,
                           pack1, goa, (const unsigned char *)ksel, r_part);
    });
    APN_LAUNCH_CHECK();
    return APN_OK;
}
