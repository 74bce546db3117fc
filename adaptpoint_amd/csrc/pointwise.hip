// pointwise.hip -- the imitator's per-point MLP layer, Conv1d(k = 1, no bias) + BatchNorm1d (+ ReLU)
// (`ConvBNReLU1D`, openpoints/models_adaptpoint/generator_component4_15.py:92-104; the embedding, the four
// `extract_feat_list` layers and the `fuse` layer of the four feature-propagation decoders, :330-366, 588-657),
// forward and backward, on channels-first (B, C, N) float32 tensors as torch.nn.Conv1d takes them.
//
// One MFMA contraction kernel serves the three products of a layer,
//     y_b  (O x N) = W   (O x C) x_b  (C x N)          forward
//     gx_b (C x N) = W^T (C x O) gy_b (O x N)          gradient w.r.t. the input
//     gW   (O x C) = sum_b gy_b (O x N) x_b^T (N x C)   gradient w.r.t. the weight (split over (b, position) ranges,
//                                                       the shares added in a fixed order: reproducible)
// by describing each operand as "k-contiguous" (element (i, k) at i * ld + k) or "row-contiguous" (k * ld + i):
// the float32 operands are read exactly as they lie in HBM (no transposed copies, no NHWC detour), split into two
// or three bf16 planes on the way into LDS (three or six MFMAs per product: ~4e-6 of an fp32 contraction, or
// fp32-class -- see pw_gemm_kernel), and every one of the 128 x 128 workgroup tile's fragment reads is a
// conflict-free 16-byte LDS read.
// BatchNorm's batch statistics are column sums of the forward product and leave the contraction's epilogue as one
// partial row per workgroup; the normalise + ReLU pass folds them itself.  The backward pass forms
// gy = dL/dy once (two passes over (g, y): the sums BatchNorm's gradient needs, then the apply) and both
// gradient contractions read it.  The same kernel, with other epilogues or as a raw-operand entry, also serves the
// discriminator's per-point layers (convolution + bias + ReLU; convolution + ReLU + max over the points) and the
// small dense products around the fused set-abstraction blocks (apn_pw_contract).
#include <hip/hip_runtime.h>

#include "../../include/adaptpoint_amd.h"
#include "apn_mfma.h"
#include <type_traits>

namespace apn {

constexpr int PW_T = 128;     // workgroup tile: 128 x 128 outputs, eight waves of 64 x 32
constexpr int PW_KC = 32;     // contraction indices per chunk (two MFMA k-steps)
constexpr int PW_ROW = 40;    // bf16 per LDS row: 32 + 8 pad (80-byte rows: 16 lanes' 16-byte reads cover all banks)
constexpr int PW_TROW = 160;  // bf16 per LDS row of a row-contiguous operand's tile, kept [k][i] as it lies: 128 + 32 pad.
                              // 320-byte rows: a transposed read is banked per 32-LANE half over 64 banks -- four k-rows x two
                              // 16-row column groups of 32 bytes each -- and a k-row must shift by 64 bytes for the eight
                              // pieces to cover the 256-byte bank row once (with 288-byte rows, until round 5, the second
                              // column group of k-row q fell on the first of k-row q + 1: every transposed read took 2x)
static_assert(32 * PW_TROW <= PW_T * PW_ROW, "a row-contiguous tile plane fits the plane stride");


struct PwOperand {
    const float *p;
    long long batch;          // elements between consecutive batch entries (0: shared)
    int ld;
};

struct PwGemm {
    PwOperand A, B;           // A (R x K), B (K x Q)
    float *D;                 // D (R x Q) per grid.z slice
    long long d_batch;
    int ldd;
    int R, Q, K;
    int cpb;                  // chunks per batch entry = ceil(K / 32)
    int cps;                  // chunks per grid.z slice
    int total;                // chunks in all
    int a_vec, b_vec;         // operand readable as aligned float4 at 32-bit byte offsets (k-contiguous: along k;
                              // row-contiguous: along the rows, which then come in whole quads: R resp. Q a multiple of 4)
    float *part;              // optional [(z * gridDim.x + x)][2][R]: this tile's row sums and M2 (squared deviations from the tile's row means)
    float *pool_val;          // optional [(z * gridDim.x + x)][R]: the tile's row maxima (D is then not written) ...
    int *pool_idx;            // ... and the column each was found at (the lowest one among equals)
};

// Loader roles of the 512 threads for one 128 x 32 operand chunk (8 values = two float4 per thread), chosen so that
// every wave-instruction reads whole lines:
//   k-contiguous operand (element (i, k) at i * ld + k): 8 lanes x float4 cover the 32 k of a row, a wave-instruction
//     covers 8 rows; thread (seg = t & 7, rg = t >> 3) holds k = 4 seg .. 4 seg + 3 of rows rg and 64 + rg
//     (a first version gave a thread 16 k of ONE row, 32 lines of 16 bytes per instruction: 3x slower end to end);
//   row-contiguous operand (k * ld + i): 32 lanes x float4 cover the 128 rows at one k; thread (i4 = 4 (t & 31),
//     kq = t >> 5) holds rows i4 .. i4 + 3 at k = kq and kq + 16, and the tile stays in LDS as it lies, [k][row]: the
//     MFMA fragments come out of it by TRANSPOSED reads (pw_fragment).  Until round 4 such an operand was read value by
//     value (lane along the rows, 8 loads of 4 bytes per thread) so that the tile could be written [row][k]: four times
//     the vector-memory instructions, and their issue -- not the bytes -- is what the loads cost this kernel.
//   The general form (a chunk that ends inside K, an operand that is not 16-byte regular) keeps value-by-value loads.
// Addresses are clamped into the operand (every load legal and unpredicated).
typedef float pw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pw_bf16x2 __attribute__((ext_vector_type(2)));
#ifdef PW_STAMPS              // diagnostic builds (scripts/pw_stamps.py): wall-clock stamps of a workgroup's phases, by thread 0
__device__ unsigned long long *d_pw_stamps = nullptr;
__device__ __forceinline__ void pw_stamp(int k) {
    unsigned long long *st = d_pw_stamps;
    if (st && threadIdx.x == 0)
        st[(blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) * 8 + k] = wall_clock64();
}
#else
__device__ __forceinline__ void pw_stamp(int) {}
#endif
constexpr int PW_NLT = 512;                       // loader threads: all of the workgroup's
constexpr int PW_LV = 128 * 32 / PW_NLT;          // values per thread, operand and chunk
constexpr int PW_LJ = PW_LV / 4;                  // ... as float4s
constexpr int PW_KCR = PW_NLT / 8;                // k-contiguous operand: rows between a thread's float4s
constexpr int PW_RCK = PW_NLT / 32;               // row-contiguous operand: k between a thread's float4s

// NS bf16 planes of the pair (a, b) as packed words (a in the low half): one v_cvt_pk_bf16_f32 per plane for BOTH
// values, the remainders by one two-wide subtraction -- ~4.5 vector instructions per value instead of ~9 for the
// value-by-value form
template <int NS>
__device__ __forceinline__ void pw_split_pair(float a, float b, unsigned (&pl)[NS]) {
    pw_f32x2 v = {a, b};
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        const pw_bf16x2 hv = __builtin_convertvector(v, pw_bf16x2);
        const unsigned w = __builtin_bit_cast(unsigned, hv);
        pl[p] = w;
        if (p + 1 < NS) {
            const pw_f32x2 u = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
            v = v - u;
        }
    }
}

template <bool KCONT>
struct PwLoader {
    int i0, kofs;            // general form.  KCONT: rg, 4 seg; else: i, PW_LV q
    unsigned rows;           // bit j: row j of this thread lies inside the operand
    int i4, kq;              // steady-state form of a row-contiguous operand: rows i4 .. i4 + 3 (one float4) at k = kq + PW_RCK j
    bool quad;               // ... and whether that quad lies inside the operand (whole quads: lim % 4 == 0)
    unsigned off[PW_LJ];     // steady-state form: byte offsets of this thread's float4s from the chunk's (uniform) origin

    __device__ __forceinline__ void init(int t, int origin, int lim) {
        if (KCONT) {
            i0 = t >> 3; kofs = 4 * (t & 7);
            rows = 0u;
#pragma unroll
            for (int j = 0; j < PW_LJ; ++j) rows |= origin + PW_KCR * j + i0 < lim ? 1u << j : 0u;
            i4 = kq = 0; quad = false;
        } else {
            i0 = t & 127; kofs = PW_LV * (t >> 7);
            rows = origin + i0 < lim ? 1u : 0u;
            i4 = 4 * (t & 31); kq = t >> 5;
            quad = origin + i4 < lim;
        }
    }
    // The steady-state loads (chunk wholly inside K, operand readable as float4): 16-byte loads whatever the layout -- a
    // row-contiguous operand was read value by value until round 4 (a vector-memory instruction issued beside the
    // matrix waves' MFMAs costs the loader ~60 cycles whatever its width) -- at 32-bit offsets that do not change
    // from chunk to chunk, from an origin that is the same for the whole wave (a scalar register pair: no vector
    // address arithmetic per chunk; the host checks that the operand's slice per batch entry is under 4 GB).
    __device__ __forceinline__ void init_fast(const PwOperand &op, int origin, int lim) {
#pragma unroll
        for (int j = 0; j < PW_LJ; ++j) {
            if (KCONT) {
                const int i = origin + PW_KCR * j + i0;
                off[j] = 4u * ((unsigned)(i < lim ? i : lim - 1) * (unsigned)op.ld + (unsigned)kofs);
            } else {
                off[j] = 4u * ((unsigned)(kq + PW_RCK * j) * (unsigned)op.ld + (unsigned)(quad ? origin + i4 : lim - 4));
            }
        }
    }
    // base: the batch entry's origin (uniform)
    __device__ __forceinline__ void load_fast(const PwOperand &op, const float *__restrict__ base, int k0,
                                              float (&v)[PW_LV]) const {
        const char *org = reinterpret_cast<const char *>(base + (KCONT ? (size_t)k0 : (size_t)k0 * op.ld));
#pragma unroll
        for (int j = 0; j < PW_LJ; ++j) {
            const float4 q = *reinterpret_cast<const float4 *>(org + off[j]);
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    }
    // The general loads -> the number of leading k of this thread's values that lie inside the operand.  `full`: the
    // chunk lies wholly inside K (wave-uniform): no clamping of the k indices
    __device__ __forceinline__ int load(const PwOperand &op, const float *__restrict__ base, int origin, int lim,
                                        int k0, int K, int vec, bool full, float (&v)[PW_LV]) const {
        const int kk = k0 + kofs, left = K - kk;
        if (KCONT) {
#pragma unroll
            for (int j = 0; j < PW_LJ; ++j) {
                const int i = origin + PW_KCR * j + i0;
                const float *src = base + (size_t)(i < lim ? i : lim - 1) * op.ld;
                if (vec) {
                    const float4 q = *reinterpret_cast<const float4 *>(src + (full || kk < K - 4 ? kk : K - 4));
                    v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * j + e] = src[full || kk + e < K ? kk + e : K - 1];
                }
            }
            return full ? 4 : (left < 0 ? 0 : (left > 4 ? 4 : left));
        } else {
            const int i = origin + i0;
            const float *src = base + (i < lim ? i : lim - 1);
#pragma unroll
            for (int j = 0; j < PW_LV; ++j) v[j] = src[(size_t)(full || kk + j < K ? kk + j : K - 1) * op.ld];
            return full ? PW_LV : (left < 0 ? 0 : (left > PW_LV ? PW_LV : left));
        }
    }
    // NS bf16 planes of every value (hi, the rounded remainder, (NS = 3) the remainder of that: 16 or 24 significant
    // bits) into the LDS tile: [plane][row][k] for a k-contiguous operand, [plane][k][row] for a row-contiguous one
    template <int NS>
    __device__ __forceinline__ void stage(__bf16 *tile, const float (&v)[PW_LV], int nk) const {
        if (KCONT) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
            for (int j = 0; j < PW_LJ; ++j) {
                bf16x4 pl[NS];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float f = (((rows >> j) & 1u) && e < nk) ? v[4 * j + e] : 0.0f;
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        const __bf16 x = (__bf16)f;
                        pl[p][e] = x;
                        f -= (float)x;
                    }
                }
                __bf16 *dst = tile + (PW_KCR * j + i0) * PW_ROW + kofs;
#pragma unroll
                for (int p = 0; p < NS; ++p) *reinterpret_cast<bf16x4 *>(dst + p * PW_T * PW_ROW) = pl[p];
            }
        } else {
            const int n = rows ? nk : 0;
#pragma unroll
            for (int j = 0; j < PW_LV; ++j) {
                float f = j < n ? v[j] : 0.0f;
                __bf16 *dst = tile + (kofs + j) * PW_TROW + i0;
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    const __bf16 x = (__bf16)f;
                    dst[p * PW_T * PW_ROW] = x;
                    f -= (float)x;
                }
            }
        }
    }
    // The same for the steady-state form, two values at a time.  Rows outside the operand hold copies of its last rows
    // (the clamped loads): rows of A outside R produce rows of D that are never stored, rows of B outside Q columns
    // that every epilogue masks (the stores, the statistics' `valid`, the pooling's -inf).
    template <int NS>
    __device__ __forceinline__ void stage_fast(__bf16 *tile, const float (&v)[PW_LV]) const {
#pragma unroll
        for (int j = 0; j < PW_LJ; ++j) {
            unsigned p01[NS], p23[NS];
            pw_split_pair<NS>(v[4 * j], v[4 * j + 1], p01);
            pw_split_pair<NS>(v[4 * j + 2], v[4 * j + 3], p23);
            __bf16 *dst = KCONT ? tile + (PW_KCR * j + i0) * PW_ROW + kofs : tile + (kq + PW_RCK * j) * PW_TROW + i4;
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2 *>(dst + p * PW_T * PW_ROW) = make_uint2(p01[p], p23[p]);
        }
    }
};

// One operand fragment of a matrix wave -- rows blk .. blk + 31 (lane r its row), the eight k of half h of k-step s -- per
// plane: one 16-byte read of a k-contiguous tile, or two transposed reads of a row-contiguous one (ds_read_b64_tr_b16:
// a 16-lane group reads a block of 4 k-rows x 16 rows and every lane receives its row's 4 k; lane 4 q + p of a group
// addresses k-row q, rows 4 p .. 4 p + 3).
template <bool KCONT, int NS>
__device__ __forceinline__ void pw_fragment(const __bf16 *tile, int blk, int s, int r, int h, bf16x8 (&f)[NS]) {
    if (KCONT) {
        const __bf16 *src = tile + (blk + r) * PW_ROW + s * 16 + h * 8;
#pragma unroll
        for (int p = 0; p < NS; ++p) f[p] = *reinterpret_cast<const bf16x8 *>(src + p * PW_T * PW_ROW);
    } else {
        typedef short tr_s4 __attribute__((ext_vector_type(4)));
        typedef short tr_s8 __attribute__((ext_vector_type(8)));
        typedef __attribute__((address_space(3))) tr_s4 tr_lds;
        const int li = r & 15;
        const __bf16 *src = tile + (s * 16 + h * 8 + (li >> 2)) * PW_TROW + blk + (r & 16) + 4 * (li & 3);
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            const tr_s4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds *)(src + p * PW_T * PW_ROW));
            const tr_s4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds *)(src + p * PW_T * PW_ROW + 4 * PW_TROW));
            const tr_s8 v8 = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            f[p] = __builtin_bit_cast(bf16x8, v8);
        }
    }
}

// NS = 2: products as hi*hi + hi*lo + lo*hi (three MFMAs, ~4e-6 of an fp32 contraction: the operands keep 16 bits);
// NS = 3: six MFMAs over three planes, every term down to 2^-24 of the product -- fp32-class results.
//
// Eight waves (2 x 4, each 64 rows x 32 columns of the 128 x 128 tile), LDS tiles double-buffered: while the MFMAs
// of chunk c read one buffer, chunk c + 1 (loaded one step earlier) is split into the other and the loads of chunk
// c + 2 are in flight -- ONE barrier per chunk.
template <int NS>
constexpr int pw_lds_bytes() { return 2 * 2 * NS * PW_T * PW_ROW * 2 + 4 * 2 * PW_T * 4; }

// LDS writes of this wave done, then the workgroup's rendezvous (global loads stay in flight across it)
__device__ __forceinline__ void pw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool A_KC, bool B_KC, int NS>
__global__ __launch_bounds__(512, 1) void pw_gemm_kernel(PwGemm g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_lds[];
    constexpr int TILE = NS * PW_T * PW_ROW;                       // bf16 per operand tile
    __bf16 *lds = reinterpret_cast<__bf16 *>(pw_lds);              // [buffer][A | B][plane][row][k]
    float (*red)[2][PW_T] = reinterpret_cast<float (*)[2][PW_T]>(pw_lds + 2 * 2 * TILE * 2);
    pw_stamp(0);
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), wr = wave >> 2, wq = wave & 3;
    // XCD-aware tile numbers (gridDim.z a multiple of 8: one batch entry per z): workgroups w and w + 8 share an XCD and
    // its L2 under round-robin placement (a speed assumption only); dealt out in launch order the tiles of one batch entry
    // -- which read the same operand slices -- went to all eight XCDs and each pulled the slice through its own L2.  Here
    // all gridDim.x * gridDim.y tiles of an entry run on one XCD.
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    if ((gridDim.z & 7u) == 0u) {
        const unsigned per = gridDim.x * gridDim.y;
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned slot = lin >> 3, w = slot % per;
        bzi = (int)((slot / per) * 8u + (lin & 7u));
        bxi = (int)(w % gridDim.x);
        byi = (int)(w / gridDim.x);
    }
    const int R0 = byi * PW_T, Q0 = bxi * PW_T, z = bzi;
    PwLoader<A_KC> la;
    PwLoader<B_KC> lb;
    la.init(t, R0, g.R);
    lb.init(t, Q0, g.Q);
    int c0 = z * g.cps, c1 = c0 + g.cps;
    if (c1 > g.total) c1 = g.total;

    f32x16 acc[2][2] = {{f32x16{0}, f32x16{0}}, {f32x16{0}, f32x16{0}}};
    float va0[PW_LV], vb0[PW_LV], va1[PW_LV], vb1[PW_LV];
    int ka0 = 0, kb0 = 0, ka1 = 0, kb1 = 0;
    // chunks are fetched in order: (batch entry, chunk inside it) advance by counting, not by a division per fetch
    int fz = c0 / g.cpb, fk = c0 - fz * g.cpb;
    auto fetch = [&](float (&xa)[PW_LV], float (&xb)[PW_LV], int &ka, int &kb) {
        const int k0 = fk * PW_KC;
        const bool full = k0 + PW_KC <= g.K;
        ka = la.load(g.A, g.A.p + g.A.batch * fz, R0, g.R, k0, g.K, A_KC && g.a_vec, full, xa);
        kb = lb.load(g.B, g.B.p + g.B.batch * fz, Q0, g.Q, k0, g.K, B_KC && g.b_vec, full, xb);
        if (++fk == g.cpb) { fk = 0; ++fz; }
    };
    // The steady-state form (every chunk whole, k-contiguous operands readable as float4 -- kernel-uniform): the SAME
    // loads in every step, issued unconditionally (past the slice's end the last chunk is read again and dropped).
    // With a conditional fetch the compiler cannot count the loads in flight: it waited for ALL of them (vmcnt(0))
    // before splitting the previous chunk, i.e. for the loads it had just issued -- a full memory round trip per chunk.
    const bool fast = g.K % PW_KC == 0 && g.a_vec && g.b_vec;
    if (fast) {
        la.init_fast(g.A, R0, g.R);
        lb.init_fast(g.B, Q0, g.Q);
    }
    int fc = c0;
    auto fetch_fast = [&](float (&xa)[PW_LV], float (&xb)[PW_LV]) {
        const int k0 = fk * PW_KC;
        la.load_fast(g.A, g.A.p + g.A.batch * fz, k0, xa);
        lb.load_fast(g.B, g.B.p + g.B.batch * fz, k0, xb);
        const int adv = fc + 1 < c1 ? 1 : 0;              // (selects, no branch: the step stays one basic block)
        fc += adv;
        fk += adv;
        const int wrap = fk == g.cpb ? 1 : 0;
        fk = wrap ? 0 : fk;
        fz += wrap;
    };
    // Four independent accumulator chains per wave -- (row block i, k-step s) -- issued in turn: a chain's next MFMA
    // needs the previous one's result (64 cycles away), and with fewer than four the matrix pipe waits on it whenever
    // this wave is the only one of its SIMD in the compute phase.  The two k-steps' sums are added in the epilogue.
    auto compute = [&](const __bf16 *As, const __bf16 *Bs) {
        bf16x8 a[2][2][NS], b[2][NS];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            pw_fragment<B_KC, NS>(Bs, wq * 32, s, r, h, b[s]);
#pragma unroll
            for (int j = 0; j < 2; ++j) pw_fragment<A_KC, NS>(As, wr * 64 + j * 32, s, r, h, a[s][j]);
        }
        // small terms first; (plane of A, plane of B) per term
        constexpr int TERMS = NS == 3 ? 6 : 3;
        constexpr int ta[6] = {NS == 3 ? 1 : 0, NS == 3 ? 0 : 1, NS == 3 ? 2 : 0, 0, 1, 0};
        constexpr int tb[6] = {1, NS == 3 ? 2 : 0, 0, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < TERMS; ++t) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][s] = mfma(a[s][i][ta[t]], b[s][tb[t]], acc[i][s]);
            }
        }
    };
    // buffer PAR holds chunk c, (xa, xb) chunk c + 1; (fa, fb) are free and take chunk c + 2
    // (the buffer a step reads is a compile-time constant -- chunk parity relative to the slice's first chunk, and the
    // loop is unrolled by two: the compiler then knows that the MFMA phase's LDS reads and the split phase's LDS writes
    // never overlap and may interleave the two phases' instructions)
    // FASTC: 0 = the general form; 1 = steady state, conditions on the slice's end kept; 2 = steady state with at
    // least two more chunks behind this one: no condition at all -- one basic block per step, in which the scheduler
    // is free to place the split phase's vector instructions between the MFMAs
    auto step = [&](auto fastc, auto parc, int c, float (&fa)[PW_LV], float (&fb)[PW_LV], int &fka, int &fkb,
                    float (&xa)[PW_LV], float (&xb)[PW_LV], int xka, int xkb) {
        constexpr int FASTC = decltype(fastc)::value;
        constexpr bool FAST = FASTC != 0, FREE = FASTC == 2;
        constexpr int PAR = decltype(parc)::value;
        __bf16 *cur = lds + PAR * 2 * TILE, *nxt = lds + (1 - PAR) * 2 * TILE;
        if (FAST) fetch_fast(fa, fb);
        else if (c + 2 < c1) fetch(fa, fb, fka, fkb);
        auto split = [&]() {
            if (FREE || c + 1 < c1) {
                if (FAST) {
                    la.template stage_fast<NS>(nxt, xa);
                    lb.template stage_fast<NS>(nxt + TILE, xb);
                } else {
                    la.template stage<NS>(nxt, xa, xka);
                    lb.template stage<NS>(nxt + TILE, xb, xkb);
                }
            }
        };
        // (measured and rejected: the two waves of a SIMD taking the two phases in opposite order, two barriers per
        // chunk -- 3-5 % slower with or without four accumulator chains: DESIGN.md section 7c)
        if (!FAST || FREE || c < c1) compute(cur, cur + TILE);
        split();
        pw_barrier();
    };
    if (fast && c0 < c1) {                     // (an empty slice reads nothing: its chunk indices lie outside the operands)
        fetch_fast(va0, vb0);
        fetch_fast(va1, vb1);
        {
            __bf16 *first = lds;
            la.template stage_fast<NS>(first, va0);
            lb.template stage_fast<NS>(first + TILE, vb0);
        }
        __syncthreads();
        pw_stamp(1);
        // (two steps per iteration, BOTH unconditional -- an odd last step only fetches and meets the barrier: with the
        // second step under a condition the two register sets changed roles across the loop's back edge by copies, and
        // a copy of a load's destination waits for the load)
        int c = c0;
        for (; c + 2 < c1; c += 2) {               // both chunks of the pair have a successor: nothing conditional
            step(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, c, va0, vb0, ka0, kb0, va1, vb1, 0, 0);
            step(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, c + 1, va1, vb1, ka1, kb1, va0, vb0, 0, 0);
        }
        for (; c < c1; c += 2) {
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, c, va0, vb0, ka0, kb0, va1, vb1, 0, 0);
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, c + 1, va1, vb1, ka1, kb1, va0, vb0, 0, 0);
        }
    } else if (!fast) {
        if (c0 < c1) {
            fetch(va0, vb0, ka0, kb0);
            if (c0 + 1 < c1) fetch(va1, vb1, ka1, kb1);
            __bf16 *first = lds;
            la.template stage<NS>(first, va0, ka0);
            lb.template stage<NS>(first + TILE, vb0, kb0);
        }
        __syncthreads();
        for (int c = c0; c < c1; c += 2) {
            step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, c, va0, vb0, ka0, kb0, va1, vb1, ka1, kb1);
            if (c + 1 < c1)
                step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, c + 1, va1, vb1, ka1, kb1, va0, vb0, ka0,
                     kb0);
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i][0] += acc[i][1];
    pw_stamp(2);
    const int q = Q0 + wq * 32 + r;
    if (g.pool_val) {
        // row maxima over the tile's columns instead of the tile: the wave's 64 x 32 block through LDS as in the
        // statistics epilogue below (written with lane = column, read back with lane = row), every lane scanning its
        // row's columns in ascending order -- a later column wins only if larger, so among equals the lowest stays
        constexpr int PROW = 36;
        float *ptile = reinterpret_cast<float *>(pw_lds) + wave * (64 * PROW);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) ptile[(i * 32 + acc_row(e, h)) * PROW + r] = acc[i][0][e];
        }
        int nwp = g.Q - (Q0 + 32 * wq);                         // this block's columns inside Q (wave-uniform)
        nwp = nwp < 0 ? 0 : (nwp > 32 ? 32 : nwp);
        const float *prow = ptile + lane * PROW;                // (one wave's LDS operations execute in order)
        float bestv = -__builtin_inff();
        int besti = Q0 + 32 * wq;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(prow + 4 * j);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool up = 4 * j + e < nwp && vv[e] > bestv;
                bestv = up ? vv[e] : bestv;
                besti = up ? Q0 + 32 * wq + 4 * j + e : besti;
            }
        }
        red[wq][0][wr * 64 + lane] = bestv;
        reinterpret_cast<int *>(&red[wq][1][0])[wr * 64 + lane] = besti;
        __syncthreads();
        if (t < PW_T && R0 + t < g.R) {
            float best = red[0][0][t];
            int bi = reinterpret_cast<int *>(&red[0][1][0])[t];
#pragma unroll
            for (int w = 1; w < 4; ++w) {                    // ascending columns: a later wave wins only if larger
                const float cand = red[w][0][t];
                if (cand > best) { best = cand; bi = reinterpret_cast<int *>(&red[w][1][0])[t]; }
            }
            const size_t o = ((size_t)z * gridDim.x + bxi) * g.R + R0 + t;
            g.pool_val[o] = best;
            g.pool_idx[o] = bi;
        }
        return;
    }
    // D: lane = column, register e <-> row acc_row(e, h): 32 consecutive columns per store
    float *D = g.D + g.d_batch * z;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = R0 + wr * 64 + i * 32 + acc_row(e, h);
            if (row < g.R && q < g.Q) D[(size_t)row * g.ldd + q] = acc[i][0][e];
        }
    }
    pw_stamp(3);
    if (g.part) {
        // BatchNorm's statistics of this tile, per row (= channel): {sum, M2 = sum of squared deviations from the
        // TILE's mean} over its columns < Q -- not {sum, sum of squares}: E[y^2] - mean^2 loses mean^2 / var digits, and
        // the imitator has channels with a few samples each (BatchNorm over 2 clouds x 4 anchors) whose mean is 100x
        // their spread.  Each wave's 32 columns are summed around a shift c = the row's value in the wave's first
        // column (any sample will do: the cancellation left is (mean - c)^2 / var = O(1)); the four waves' blocks and,
        // in pw_bn_act_kernel, the tiles are combined in float64 by the pairwise update of Chan et al.
        // The wave's 64 x 32 block goes through LDS (the operand tiles are free behind the main loop's last barrier):
        // written as it lies in the accumulators (lane = column), read back with lane = ROW -- every lane then sums
        // its row's 32 columns itself.  (Until round 4 the sums were formed across the lanes: 94 lane exchanges and
        // ~600 vector instructions per wave, 3.5-4 us of every forward contraction, a third of a small layer's.)
        constexpr int SROW = 36;                                                                    // floats per row: 32 + 4 pad
        float *ytile = reinterpret_cast<float *>(pw_lds) + wave * (64 * SROW);
        float (*redc)[PW_T] = reinterpret_cast<float (*)[PW_T]>(pw_lds + 8 * 64 * SROW * 4);       // [wq][row]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) ytile[(i * 32 + acc_row(e, h)) * SROW + r] = acc[i][0][e];
        }
        int nw = g.Q - (Q0 + 32 * wq);                          // this block's columns inside Q (wave-uniform)
        nw = nw < 0 ? 0 : (nw > 32 ? 32 : nw);
        const float *src = ytile + lane * SROW;                 // (one wave's LDS operations execute in order)
        const float c = src[0];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(src + 4 * j);
            const float d0 = 4 * j < nw ? v.x - c : 0.0f, d1 = 4 * j + 1 < nw ? v.y - c : 0.0f;
            const float d2 = 4 * j + 2 < nw ? v.z - c : 0.0f, d3 = 4 * j + 3 < nw ? v.w - c : 0.0f;
            s1 += d0; s2 = __builtin_fmaf(d0, d0, s2);
            s1 += d1; s2 = __builtin_fmaf(d1, d1, s2);
            s1 += d2; s2 = __builtin_fmaf(d2, d2, s2);
            s1 += d3; s2 = __builtin_fmaf(d3, d3, s2);
        }
        red[wq][0][wr * 64 + lane] = s1;
        red[wq][1][wr * 64 + lane] = s2;
        redc[wq][wr * 64 + lane] = c;
        __syncthreads();
        if (t < PW_T && R0 + t < g.R && Q0 + PW_T <= g.Q) {
            // a tile wholly inside Q (workgroup-uniform): four blocks of 32 columns, every count a power of two -- the
            // same combination without a division (the general form below makes sixteen float64 divisions per row:
            // 1.5 of this epilogue's 4 us)
            double mw[4], m2w[4], sum = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const double a = (double)red[w][0][t], b = (double)red[w][1][t];
                mw[w] = (double)redc[w][t] + a * 0.03125;
                m2w[w] = b - a * a * 0.03125;
                sum += mw[w];
            }
            const double mean = sum * 0.25;
            double m2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) m2 += m2w[w] + 32.0 * (mw[w] - mean) * (mw[w] - mean);
            float *dst = g.part + ((size_t)z * gridDim.x + bxi) * 2 * g.R + R0 + t;
            dst[0] = (float)(mean * 128.0);
            dst[g.R] = (float)(m2 < 0.0 ? 0.0 : m2);
        } else if (t < PW_T && R0 + t < g.R) {
            double n = 0.0, mean = 0.0, m2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                int nw = g.Q - (Q0 + 32 * w);
                nw = nw < 0 ? 0 : (nw > 32 ? 32 : nw);
                if (nw > 0) {
                    const double a = (double)red[w][0][t], b = (double)red[w][1][t];
                    const double mw = (double)redc[w][t] + a / nw, m2w = b - a * a / nw;
                    const double tot = n + nw, delta = mw - mean;
                    m2 += m2w + delta * delta * n * nw / tot;
                    mean += delta * nw / tot;
                    n = tot;
                }
            }
            float *dst = g.part + ((size_t)z * gridDim.x + bxi) * 2 * g.R + R0 + t;
            dst[0] = (float)(mean * n);
            dst[g.R] = (float)(m2 < 0.0 ? 0.0 : m2);
        }
    }
    pw_stamp(4);
}

// sum of one double per thread over the workgroup, returned to every thread
__device__ __forceinline__ double pw_block_sum(double v, double *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

template <typename F>
__device__ __forceinline__ void pw_for_rows(int B, int N, int C, int ch, F fn) {
    for (int b = blockIdx.y; b < B; b += gridDim.y) fn(((size_t)b * C + ch) * N);
}

// BatchNorm (batch statistics folded here from the contraction's partial rows, or the running ones) + ReLU.
// grid (C, S): workgroup (c, s) serves channel c of the clouds b = s mod S; stat [4][C] = {mean, invstd, scale, shift}.
__global__ __launch_bounds__(256) void pw_bn_act_kernel(int B, int C, int N, const float *__restrict__ y,
                                                        const float *__restrict__ part, int tiles,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float eps, float momentum, int training, int relu,
                                                        float *__restrict__ run_mean, float *__restrict__ run_var,
                                                        long long *__restrict__ batches, float *__restrict__ stat,
                                                        float *__restrict__ out) {
    __shared__ double scratch[4];
    const int c = blockIdx.x, t = threadIdx.x;
    float mean, inv;
    if (training) {
        // part[tile] = {sum, M2 around the tile's mean} (pw_gemm_kernel's epilogue); tile i covers the columns
        // [128 (i mod tx), ...) of a cloud: M2 = sum_t m2_t + sum_t n_t (mean_t - mean)^2, in float64
        const int tx = (N + PW_T - 1) / PW_T;
        double a = 0.0;
        for (int i = t; i < tiles; i += 256) a += (double)part[(size_t)i * 2 * C + c];
        a = pw_block_sum(a, scratch);
        const double cnt = (double)B * N, mu = a / cnt;
        double q = 0.0;
        for (int i = t; i < tiles; i += 256) {
            int nt = N - PW_T * (i % tx);
            nt = nt > PW_T ? PW_T : nt;
            const double dm = (double)part[(size_t)i * 2 * C + c] / nt - mu;
            q += (double)part[(size_t)i * 2 * C + C + c] + dm * dm * nt;
        }
        q = pw_block_sum(q, scratch);
        double var = q / cnt;
        if (var < 0.0) var = 0.0;
        mean = (float)mu;
        inv = (float)(1.0 / sqrt(var + (double)eps));
        if (blockIdx.y == 0 && t == 0) {
            if (run_mean) {
                run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
                run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)(var * (cnt > 1.0 ? cnt / (cnt - 1.0) : 1.0));
            }
            if (batches && c == 0) batches[0] += 1;
        }
    } else {
        mean = run_mean[c];
        inv = 1.0f / sqrtf(run_var[c] + eps);
    }
    const float sc = (gamma ? gamma[c] : 1.0f) * inv, sh = (beta ? beta[c] : 0.0f) - mean * sc;
    if (blockIdx.y == 0 && t == 0) {
        stat[c] = mean;
        stat[C + c] = inv;
        stat[2 * C + c] = sc;
        stat[3 * C + c] = sh;
    }
    const bool vec = (N & 3) == 0;
    pw_for_rows(B, N, C, c, [&](size_t off) {
        if (vec) {
            const float4 *src = reinterpret_cast<const float4 *>(y + off);
            float4 *dst = reinterpret_cast<float4 *>(out + off);
            for (int i = t; i < N / 4; i += 256) {
                float4 v = src[i];
                v.x = __builtin_fmaf(v.x, sc, sh); v.y = __builtin_fmaf(v.y, sc, sh);
                v.z = __builtin_fmaf(v.z, sc, sh); v.w = __builtin_fmaf(v.w, sc, sh);
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                dst[i] = v;
            }
        } else {
            for (int i = t; i < N; i += 256) {
                const float v = __builtin_fmaf(y[off + i], sc, sh);
                out[off + i] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    });
}

// pass 1 of the backward: partb[s][2][C] = {sum g m, sum g m yhat} over workgroup (c, s)'s clouds, m = the ReLU mask
__global__ __launch_bounds__(256) void pw_bwd_sums_kernel(int B, int C, int N, const float *__restrict__ g,
                                                          const float *__restrict__ y, const float *__restrict__ stat,
                                                          int relu, float *__restrict__ partb) {
    __shared__ double scratch[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const float mean = stat[c], inv = stat[C + c], sc = stat[2 * C + c], sh = stat[3 * C + c];
    float s1 = 0.0f, s2 = 0.0f;
    double d1 = 0.0, d2 = 0.0;
    pw_for_rows(B, N, C, c, [&](size_t off) {
        for (int i = t; i < N; i += 256) {
            const float yv = y[off + i];
            const float gm = (!relu || __builtin_fmaf(yv, sc, sh) > 0.0f) ? g[off + i] : 0.0f;
            s1 += gm;
            s2 = __builtin_fmaf(gm, (yv - mean) * inv, s2);
        }
        d1 += (double)s1; d2 += (double)s2;     // one cloud's share at a time into the wide accumulator
        s1 = 0.0f; s2 = 0.0f;
    });
    d1 = pw_block_sum(d1, scratch);
    d2 = pw_block_sum(d2, scratch);
    if (t == 0) {
        partb[(size_t)blockIdx.y * 2 * C + c] = (float)d1;
        partb[(size_t)blockIdx.y * 2 * C + C + c] = (float)d2;
    }
}

// pass 2: gy = scale (g m - mean(g m) - yhat mean(g m yhat)) (training) or scale g m (running statistics);
// dL/dgamma = sum g m yhat, dL/dbeta = sum g m
__global__ __launch_bounds__(256) void pw_bwd_apply_kernel(int B, int C, int N, const float *__restrict__ g,
                                                           const float *__restrict__ y, const float *__restrict__ stat,
                                                           int relu, int training, const float *__restrict__ partb,
                                                           int splits, float *__restrict__ gy,
                                                           float *__restrict__ g_gamma, float *__restrict__ g_beta) {
    const int c = blockIdx.x, t = threadIdx.x;
    const float mean = stat[c], inv = stat[C + c], sc = stat[2 * C + c], sh = stat[3 * C + c];
    double d1 = 0.0, d2 = 0.0;
    for (int s = 0; s < splits; ++s) {          // every thread the same fixed order
        d1 += (double)partb[(size_t)s * 2 * C + c];
        d2 += (double)partb[(size_t)s * 2 * C + C + c];
    }
    if (blockIdx.y == 0 && t == 0) {
        if (g_gamma) g_gamma[c] = (float)d2;
        if (g_beta) g_beta[c] = (float)d1;
    }
    // gy = sc gm + k0 + k1 y
    float k0 = 0.0f, k1 = 0.0f;
    if (training) {
        const double cnt = (double)B * N;
        const double m1 = d1 / cnt, m2 = d2 / cnt;
        k1 = (float)(-(double)sc * (double)inv * m2);
        k0 = (float)(-(double)sc * m1 + (double)sc * (double)inv * m2 * (double)mean);
    }
    const bool vec = (N & 3) == 0;
    pw_for_rows(B, N, C, c, [&](size_t off) {
        if (vec) {
            const float4 *gs = reinterpret_cast<const float4 *>(g + off), *ys = reinterpret_cast<const float4 *>(y + off);
            float4 *dst = reinterpret_cast<float4 *>(gy + off);
            for (int i = t; i < N / 4; i += 256) {
                const float4 gv = gs[i], yv = ys[i];
                float4 o;
                o.x = __builtin_fmaf((!relu || __builtin_fmaf(yv.x, sc, sh) > 0.0f) ? gv.x : 0.0f, sc, __builtin_fmaf(k1, yv.x, k0));
                o.y = __builtin_fmaf((!relu || __builtin_fmaf(yv.y, sc, sh) > 0.0f) ? gv.y : 0.0f, sc, __builtin_fmaf(k1, yv.y, k0));
                o.z = __builtin_fmaf((!relu || __builtin_fmaf(yv.z, sc, sh) > 0.0f) ? gv.z : 0.0f, sc, __builtin_fmaf(k1, yv.z, k0));
                o.w = __builtin_fmaf((!relu || __builtin_fmaf(yv.w, sc, sh) > 0.0f) ? gv.w : 0.0f, sc, __builtin_fmaf(k1, yv.w, k0));
                dst[i] = o;
            }
        } else {
            for (int i = t; i < N; i += 256) {
                const float yv = y[off + i];
                const float gm = (!relu || __builtin_fmaf(yv, sc, sh) > 0.0f) ? g[off + i] : 0.0f;
                gy[off + i] = __builtin_fmaf(gm, sc, __builtin_fmaf(k1, yv, k0));
            }
        }
    });
}

// Both passes as ONE launch for a channel whose B * N values fit the registers of one workgroup (1024 threads x PW_BV
// float4 of g and of y: B * N <= 32768, N a multiple of 4): workgroup c reads channel c's (g, y) once, forms the two sums
// (per-thread float partial sums over <= 32 values, the workgroup's total in float64, a fixed order), then applies them to
// the values it still holds.  As two launches every layer paid two dependent launches of 7-11 us, each a round trip over
// the same (g, y) (18-21 us per layer, 0.61 ms of a joint training step).
constexpr int PW_BV = 8;
__global__ __launch_bounds__(1024) void pw_bwd_fused_kernel(int B, int C, int N, const float *__restrict__ g,
                                                            const float *__restrict__ y, const float *__restrict__ stat,
                                                            int relu, int training, float *__restrict__ gy,
                                                            float *__restrict__ g_gamma, float *__restrict__ g_beta) {
    __shared__ double scratch[2][16];
    const int c = blockIdx.x, t = threadIdx.x;
    const int n4 = N >> 2, total4 = B * n4;
    float4 gv[PW_BV], yv[PW_BV];
    size_t off[PW_BV];
#pragma unroll
    for (int j = 0; j < PW_BV; ++j) {                 // every load issued before the first value is used (clamped, selected below)
        int i = t + 1024 * j;
        i = i < total4 ? i : total4 - 1;
        const int b = i / n4, o = i - b * n4;
        off[j] = ((size_t)b * C + c) * n4 + o;
        gv[j] = reinterpret_cast<const float4 *>(g)[off[j]];
        yv[j] = reinterpret_cast<const float4 *>(y)[off[j]];
    }
    const float mean = stat[c], inv = stat[C + c], sc = stat[2 * C + c], sh = stat[3 * C + c];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < PW_BV; ++j) {
        const bool in = t + 1024 * j < total4;
        float *gp = reinterpret_cast<float *>(&gv[j]);
        const float *yp = reinterpret_cast<const float *>(&yv[j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gm = (in && (!relu || __builtin_fmaf(yp[e], sc, sh) > 0.0f)) ? gp[e] : 0.0f;
            gp[e] = gm;                               // the masked gradient replaces g
            s1 += gm;
            s2 = __builtin_fmaf(gm, (yp[e] - mean) * inv, s2);
        }
    }
    double d1 = (double)s1, d2 = (double)s2;
    for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
    if ((t & 63) == 0) { scratch[0][t >> 6] = d1; scratch[1][t >> 6] = d2; }
    __syncthreads();
    d1 = 0.0; d2 = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { d1 += scratch[0][w]; d2 += scratch[1][w]; }     // every thread the same fixed order
    if (t == 0) {
        if (g_gamma) g_gamma[c] = (float)d2;
        if (g_beta) g_beta[c] = (float)d1;
    }
    float k0 = 0.0f, k1 = 0.0f;                       // gy = sc gm + k0 + k1 y  (pw_bwd_apply_kernel)
    if (training) {
        const double cnt = (double)B * N;
        const double m1 = d1 / cnt, m2 = d2 / cnt;
        k1 = (float)(-(double)sc * (double)inv * m2);
        k0 = (float)(-(double)sc * m1 + (double)sc * (double)inv * m2 * (double)mean);
    }
#pragma unroll
    for (int j = 0; j < PW_BV; ++j) {
        if (t + 1024 * j < total4) {
            float4 o;
            o.x = __builtin_fmaf(gv[j].x, sc, __builtin_fmaf(k1, yv[j].x, k0));
            o.y = __builtin_fmaf(gv[j].y, sc, __builtin_fmaf(k1, yv[j].y, k0));
            o.z = __builtin_fmaf(gv[j].z, sc, __builtin_fmaf(k1, yv[j].z, k0));
            o.w = __builtin_fmaf(gv[j].w, sc, __builtin_fmaf(k1, yv[j].w, k0));
            reinterpret_cast<float4 *>(gy)[off[j]] = o;
        }
    }
}

// out[e] = sum_s part[s][e] in a fixed order: 64 elements x 4 interleaved split lanes per workgroup
__global__ __launch_bounds__(256) void pw_fold_kernel(const float *__restrict__ part, int splits, size_t n,
                                                      float *__restrict__ out) {
    __shared__ double red[4][64];
    const size_t e = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = threadIdx.x >> 6;
    double s = 0.0;
    if (e < n)
        for (int k = q; k < splits; k += 4) s += (double)part[(size_t)k * n + e];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && e < n) out[e] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// ---- convolution + bias + ReLU + max over the points of a cloud (the last layer of the discriminator's group-all
// stage, point_discriminator.py:183-189: the (B, 1024, N) activation it pools is never written) ------------------
// out[b][o] = [relu](max_n y[b][o][n] + bias[o]) from the tiles' maxima (ascending tiles: the lowest position wins ties)
__global__ __launch_bounds__(256) void pw_pool_finish_kernel(int total, int O, int tiles, const float *__restrict__ pv,
                                                             const int *__restrict__ pi, const float *__restrict__ bias,
                                                             int relu, float *__restrict__ out, int *__restrict__ idx) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int b = e / O, o = e - b * O;
    const size_t base = (size_t)b * tiles * O + o;
    float best = pv[base];
    int bi = pi[base];
    for (int t = 1; t < tiles; ++t) {
        const float c = pv[base + (size_t)t * O];
        if (c > best) { best = c; bi = pi[base + (size_t)t * O]; }
    }
    best += bias ? bias[o] : 0.0f;
    out[e] = relu ? fmaxf(best, 0.0f) : best;
    idx[e] = bi;
}

// Backward, part 1, workgroup (chunk of 64 positions, cloud b): the pooled output (b, o) passes its gradient to ONE
// position idx[b][o].  The workgroup first lists the channels o whose position falls into its chunk, in ascending o
// (a ballot / prefix compaction over the cloud's idx row: walking all O channels with a dependent scalar load each
// took 147 us), then walks the list -- entries alternately by threads 0-127 and 128-255, one LDS accumulator tile each,
// added at the end: fixed orders, no atomics -- adding w[o][:] * gp to that position's column of g_x and copying the
// position's input column to xsel[b][o][:] for part 2.
// LDS: xt [64][C + 1] | acc [2][64][C + 1] | hits [O] ints | the hits' gradients [O]
__global__ __launch_bounds__(256) void pw_pool_grad_points_kernel(int C, int O, int N, const float *__restrict__ x,
                                                                  const float *__restrict__ w,
                                                                  const float *__restrict__ g_out,
                                                                  const float *__restrict__ out,
                                                                  const int *__restrict__ idx, int relu,
                                                                  float *__restrict__ xsel, float *__restrict__ g_x) {
    extern __shared__ float pool_sm[];
    __shared__ int wave_count[4], nhits;
    const int ld = C + 1, n0 = blockIdx.x * 64, b = blockIdx.y, tid = threadIdx.x;
    float *xt = pool_sm, *acc = pool_sm + 64 * ld;
    int *hits = reinterpret_cast<int *>(pool_sm + 3 * 64 * ld);
    float *hitg = pool_sm + 3 * 64 * ld + O;
    const int p = tid & 63, cg = tid >> 6;
    const float *xb = x + (size_t)b * C * N;
    for (int c = cg; c < C; c += 4) xt[p * ld + c] = n0 + p < N ? xb[(size_t)c * N + n0 + p] : 0.0f;
    for (int e = tid; e < 2 * 64 * ld; e += 256) acc[e] = 0.0f;
    if (tid == 0) nhits = 0;
    __syncthreads();
    for (int base = 0; base < O; base += 256) {
        const int o = base + tid;
        const int at = o < O ? idx[(size_t)b * O + o] - n0 : -1;
        const bool hit = at >= 0 && at < 64;
        const unsigned long long m = __ballot(hit);
        if (p == 0) wave_count[cg] = __popcll(m);
        __syncthreads();
        int before = nhits;
        for (int k = 0; k < cg; ++k) before += wave_count[k];
        if (hit) {
            // the entry's gradient through the ReLU, fetched here by the thread that found it (all entries at once)
            const int slot = before + __popcll(m & ((1ull << p) - 1ull));
            const float gv = g_out[(size_t)b * O + o];
            hits[slot] = o | (at << 24);
            hitg[slot] = (!relu || out[(size_t)b * O + o] > 0.0f) ? gv : 0.0f;
        }
        __syncthreads();
        if (tid == 0) nhits += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        __syncthreads();
    }
    const int half = __builtin_amdgcn_readfirstlane(tid >> 7), c = tid & 127, total = nhits;
    float *mine = acc + half * 64 * ld;
    // eight list entries per round: their loads (gradient, weight row) are independent and issued together -- every
    // load unconditional, entries past the list read entry 0's row and are dropped -- the LDS updates follow in list
    // order (one entry at a time paid a global round trip each: 110 us; four per round under conditions: 80-112 us)
    constexpr int PE = 8;
    const int cw = c < C ? c : 0;
    for (int k0 = half; k0 < total; k0 += 2 * PE) {
        int o[PE], at[PE];
        float gp[PE], wv[PE];
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            const int k = k0 + 2 * j;
            const int h = hits[k < total ? k : 0];
            o[j] = k < total ? (h & 0xffffff) : -1;
            at[j] = h >> 24;
            gp[j] = hitg[k < total ? k : 0];
            wv[j] = w[(size_t)(h & 0xffffff) * C + cw];
        }
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            if (o[j] >= 0 && c < C) {
                xsel[((size_t)b * O + o[j]) * C + c] = xt[at[j] * ld + c];
                if (g_x) mine[at[j] * ld + c] = __builtin_fmaf(wv[j], gp[j], mine[at[j] * ld + c]);
            }
        }
    }
    if (!g_x) return;
    __syncthreads();
    float *gb = g_x + (size_t)b * C * N;
    if (n0 + p < N)
        for (int cc = cg; cc < C; cc += 4) gb[(size_t)cc * N + n0 + p] = acc[p * ld + cc] + acc[64 * ld + p * ld + cc];
}

// part 2: g_w[o][c] = sum_b gp[b][o] xsel[b][o][c], g_bias[o] = sum_b gp[b][o]: workgroup o, thread c, clouds in order
__global__ __launch_bounds__(128) void pw_pool_grad_weight_kernel(int B, int C, int O, const float *__restrict__ xsel,
                                                                  const float *__restrict__ g_out,
                                                                  const float *__restrict__ out, int relu,
                                                                  float *__restrict__ g_w, float *__restrict__ g_bias) {
    const int o = blockIdx.x, c = threadIdx.x;
    float a = 0.0f, sb = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float gv = g_out[(size_t)b * O + o];
        const float gp = (!relu || out[(size_t)b * O + o] > 0.0f) ? gv : 0.0f;
        sb += gp;
        if (c < C) a = __builtin_fmaf(gp, xsel[((size_t)b * O + o) * C + c], a);
    }
    if (c < C) g_w[(size_t)o * C + c] = a;
    if (c == 0 && g_bias) g_bias[o] = sb;
}

static int pw_channel_splits(int b, int c) {
    // (c, s) workgroups: enough of them to fill the chip, never more than clouds
    int s = (2048 + c - 1) / c;
    if (s > b) s = b;
    return s < 1 ? 1 : s;
}

static int pw_weight_splits(int b, int c_in, int c_out, int n) {
    const long long tiles = (long long)((c_out + PW_T - 1) / PW_T) * ((c_in + PW_T - 1) / PW_T);
    const long long total = (long long)b * ((n + PW_KC - 1) / PW_KC);
    // tiles * s workgroups, ONE of which a CU holds at a time (87 / 127 KB of LDS with two / three planes): at most one
    // round of the 256 CUs -- with two rounds (512 / tiles, from when two tiles shared a CU) the contraction took as long
    // and the fold read twice the shares
    long long s = 256 / tiles;
    // (a lone workgroup walks its chunks at ~2 us each: the short contractions are spread down to two chunks per
    // share, the long ones keep at least eight and a light fold)
    const long long cap = total >= 64 ? total / 8 : total / 2;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}

// k-contiguous operand (rows x K, element (i, k) at i * ld + k) readable as aligned float4 along k at 32-bit byte offsets
static bool pw_vec(const float *p, long long batch, int ld, int K, int rows) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (batch % 4 == 0) && (ld % 4 == 0) && (K % 4 == 0) &&
           (long long)rows * ld < (1ll << 29);
}
// the same for a row-contiguous operand (K x rows, element (i, k) at k * ld + i): float4 along its rows, whole quads
static bool pw_vec_rows(const float *p, long long batch, int ld, int rows, int K) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (batch % 4 == 0) && (ld % 4 == 0) && (rows % 4 == 0) &&
           ((long long)K + PW_KC) * ld < (1ll << 29);
}


}  // namespace apn

#ifdef PW_STAMPS
// (diagnostic builds only) attach a buffer of 8 stamps per workgroup for the next launches of the contraction kernel
extern "C" __attribute__((visibility("default"))) int apn_pw_debug_stamps(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(apn::d_pw_stamps), &buf, sizeof(buf));
}
#endif

extern "C" int apn_pw_conv_tiles(int b, int n) { return b * ((n + apn::PW_T - 1) / apn::PW_T); }

template <bool AK, bool BK, int NS>
static int pw_launch(dim3 grid, const apn::PwGemm &g, hipStream_t stream) {
    using namespace apn;
    static DynLdsOnce configured;        // per instantiation and device (apn_common.h)
    if (hipError_t e = set_dyn_lds(configured, (const void *)pw_gemm_kernel<AK, BK, NS>, pw_lds_bytes<NS>())) return (int)e;
    hipLaunchKernelGGL((pw_gemm_kernel<AK, BK, NS>), grid, dim3(512), pw_lds_bytes<NS>(), stream, g);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

#define PW_LAUNCH(AK, BK, grid, g)                                                                              \
    do {                                                                                                        \
        const int rc__ = precision == 3 ? pw_launch<AK, BK, 3>(grid, g, (hipStream_t)stream)                   \
                                        : pw_launch<AK, BK, 2>(grid, g, (hipStream_t)stream);                  \
        if (rc__) return rc__;                                                                                  \
    } while (0)

extern "C" int apn_pw_conv_forward(int b, int c_in, int c_out, int n, int precision, const float *x, const float *w,
                                   float *y, float *part, void *stream) {
    using namespace apn;
    if (b < 0 || c_in <= 0 || c_out <= 0 || n < 0 || b > 65535 || (precision != 2 && precision != 3)) return APN_EINVAL;
    if (b == 0 || n == 0) return APN_OK;
    if (!x || !w || !y) return APN_EINVAL;
    PwGemm g{};
    g.A = PwOperand{w, 0, c_in};
    g.B = PwOperand{x, (long long)c_in * n, n};
    g.D = y; g.d_batch = (long long)c_out * n; g.ldd = n;
    g.R = c_out; g.Q = n; g.K = c_in;
    g.cpb = (c_in + PW_KC - 1) / PW_KC; g.cps = g.cpb; g.total = b * g.cpb;
    g.a_vec = pw_vec(w, 0, c_in, c_in, c_out); g.b_vec = pw_vec_rows(x, g.B.batch, n, n, c_in);
    g.part = part;
    PW_LAUNCH(true, false, dim3((n + PW_T - 1) / PW_T, (c_out + PW_T - 1) / PW_T, b), g);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_pw_bn_act(int b, int c, int n, const float *y, const float *part, int tiles, const float *gamma,
                             const float *beta, float eps, float momentum, int training, int relu, float *run_mean,
                             float *run_var, long long *batches, float *stat, float *out, void *stream) {
    using namespace apn;
    if (b < 0 || c <= 0 || n < 0) return APN_EINVAL;
    if (b == 0 || n == 0) return APN_OK;
    if (!y || !stat || !out || (training && (!part || tiles <= 0)) || (!training && (!run_mean || !run_var)))
        return APN_EINVAL;
    hipLaunchKernelGGL(pw_bn_act_kernel, dim3(c, pw_channel_splits(b, c)), dim3(256), 0, (hipStream_t)stream, b, c, n,
                       y, part, tiles, gamma, beta, eps, momentum, training, relu, run_mean, run_var, batches, stat, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_pw_bn_act_grad_splits(int b, int c) { return apn::pw_channel_splits(b, c); }

extern "C" int apn_pw_bn_act_grad(int b, int c, int n, const float *g, const float *y, const float *stat, int training,
                                  int relu, float *part_b, float *gy, float *g_gamma, float *g_beta, void *stream) {
    using namespace apn;
    if (b < 0 || c <= 0 || n < 0) return APN_EINVAL;
    if (b == 0 || n == 0) return APN_OK;
    if (!g || !y || !stat || !part_b || !gy) return APN_EINVAL;
    const bool aligned = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
    if ((n & 3) == 0 && (long long)b * n <= 1024ll * 4 * PW_BV && aligned) {       // a channel's values fit one workgroup's registers
        hipLaunchKernelGGL(pw_bwd_fused_kernel, dim3(c), dim3(1024), 0, (hipStream_t)stream, b, c, n, g, y, stat, relu, training,
                           gy, g_gamma, g_beta);
        APN_LAUNCH_CHECK();
        return APN_OK;
    }
    const int s = pw_channel_splits(b, c);
    hipLaunchKernelGGL(pw_bwd_sums_kernel, dim3(c, s), dim3(256), 0, (hipStream_t)stream, b, c, n, g, y, stat, relu,
                       part_b);
    APN_LAUNCH_CHECK();
    hipLaunchKernelGGL(pw_bwd_apply_kernel, dim3(c, s), dim3(256), 0, (hipStream_t)stream, b, c, n, g, y, stat, relu,
                       training, part_b, s, gy, g_gamma, g_beta);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_pw_conv_grad_input(int b, int c_in, int c_out, int n, int precision, const float *gy, const float *w,
                                      float *gx, void *stream) {
    using namespace apn;
    if (b < 0 || c_in <= 0 || c_out <= 0 || n < 0 || b > 65535 || (precision != 2 && precision != 3)) return APN_EINVAL;
    if (b == 0 || n == 0) return APN_OK;
    if (!gy || !w || !gx) return APN_EINVAL;
    PwGemm g{};
    g.A = PwOperand{w, 0, c_in};                      // W^T: element (c, k = o) at w[k * c_in + c]
    g.B = PwOperand{gy, (long long)c_out * n, n};
    g.D = gx; g.d_batch = (long long)c_in * n; g.ldd = n;
    g.R = c_in; g.Q = n; g.K = c_out;
    g.cpb = (c_out + PW_KC - 1) / PW_KC; g.cps = g.cpb; g.total = b * g.cpb;
    g.a_vec = pw_vec_rows(w, 0, c_in, c_in, c_out); g.b_vec = pw_vec_rows(gy, g.B.batch, n, n, c_out);
    PW_LAUNCH(false, false, dim3((n + PW_T - 1) / PW_T, (c_in + PW_T - 1) / PW_T, b), g);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_pw_conv_grad_weight_splits(int b, int c_in, int c_out, int n) {
    return apn::pw_weight_splits(b, c_in, c_out, n);
}

extern "C" int apn_pw_conv_grad_weight(int b, int c_in, int c_out, int n, int precision, const float *gy,
                                       const float *x, float *scratch, float *gw, void *stream) {
    using namespace apn;
    if (b <= 0 || c_in <= 0 || c_out <= 0 || n <= 0 || !gy || !x || !scratch || !gw || (precision != 2 && precision != 3))
        return APN_EINVAL;
    const int s = pw_weight_splits(b, c_in, c_out, n);
    PwGemm g{};
    g.A = PwOperand{gy, (long long)c_out * n, n};     // (o, k = position)
    g.B = PwOperand{x, (long long)c_in * n, n};       // (k = position, c) at x[c * n + k]
    g.D = scratch; g.d_batch = (long long)c_out * c_in; g.ldd = c_in;
    g.R = c_out; g.Q = c_in; g.K = n;
    g.cpb = (n + PW_KC - 1) / PW_KC; g.total = b * g.cpb; g.cps = (g.total + s - 1) / s;
    g.a_vec = pw_vec(gy, g.A.batch, n, n, c_out); g.b_vec = pw_vec(x, g.B.batch, n, n, c_in);
    PW_LAUNCH(true, true, dim3((c_in + PW_T - 1) / PW_T, (c_out + PW_T - 1) / PW_T, s), g);
    APN_LAUNCH_CHECK();
    const size_t ne = (size_t)c_out * c_in;
    hipLaunchKernelGGL(pw_fold_kernel, dim3((unsigned)((ne + 63) / 64)), dim3(256), 0, (hipStream_t)stream, scratch, s,
                       ne, gw);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_pw_conv_max_tiles(int n) { return (n + apn::PW_T - 1) / apn::PW_T; }

extern "C" int apn_pw_conv_max_forward(int b, int c_in, int c_out, int n, int precision, const float *x, const float *w,
                                       const float *bias, int relu, float *tile_val, int *tile_idx, float *out, int *idx,
                                       void *stream) {
    using namespace apn;
    if (b <= 0 || c_in <= 0 || c_out <= 0 || n <= 0 || b > 65535 || (precision != 2 && precision != 3) || !x || !w ||
        !tile_val || !tile_idx || !out || !idx)
        return APN_EINVAL;
    const int tiles = (n + PW_T - 1) / PW_T;
    PwGemm g{};
    g.A = PwOperand{w, 0, c_in};
    g.B = PwOperand{x, (long long)c_in * n, n};
    g.R = c_out; g.Q = n; g.K = c_in;
    g.cpb = (c_in + PW_KC - 1) / PW_KC; g.cps = g.cpb; g.total = b * g.cpb;
    g.a_vec = pw_vec(w, 0, c_in, c_in, c_out); g.b_vec = pw_vec_rows(x, g.B.batch, n, n, c_in);
    g.pool_val = tile_val; g.pool_idx = tile_idx;
    PW_LAUNCH(true, false, dim3(tiles, (c_out + PW_T - 1) / PW_T, b), g);
    const int total = b * c_out;
    hipLaunchKernelGGL(pw_pool_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, total, c_out,
                       tiles, tile_val, tile_idx, bias, relu, out, idx);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_pw_conv_max_backward(int b, int c_in, int c_out, int n, const float *g_out, const float *out,
                                        const int *idx, const float *x, const float *w, int relu, float *xsel, float *g_x,
                                        float *g_w, float *g_bias, void *stream) {
    using namespace apn;
    if (b <= 0 || c_in <= 0 || c_in > 128 || c_out <= 0 || n <= 0 || b > 65535 || !g_out || !out || !idx || !x || !w ||
        !xsel || !g_w)
        return APN_EINVAL;
    if (c_out >= (1 << 24)) return APN_EINVAL;
    const size_t lds = ((size_t)3 * 64 * (c_in + 1) + 2 * (size_t)c_out) * sizeof(float);
    if (hipError_t e = hipFuncSetAttribute((const void *)pw_pool_grad_points_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
        return (int)e;
    hipLaunchKernelGGL(pw_pool_grad_points_kernel, dim3((n + 63) / 64, b), dim3(256), lds, (hipStream_t)stream, c_in,
                       c_out, n, x, w, g_out, out, idx, relu, xsel, g_x);
    APN_LAUNCH_CHECK();
    hipLaunchKernelGGL(pw_pool_grad_weight_kernel, dim3(c_out), dim3(128), 0, (hipStream_t)stream, b, c_in, c_out, xsel,
                       g_out, out, relu, g_w, g_bias);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

// The contraction kernel itself, for the small dense products around the fused set-abstraction blocks (their
// PyTorch forms are skinny or oddly shaped GEMMs the vendor library serves at 40-120 us apiece):
//   splits == 0:  D[z] (R x Q) = A[z] (R x K) B[z] (K x Q) for z < nbatch  (a batch stride of 0 shares an operand)
//   splits  > 0:  D (R x Q) = sum_z A[z] B[z], the (z, k) range cut into `splits` shares (scratch [splits][R][Q]) that
//                 are added in a fixed order.
// An operand is k-contiguous (element (i, k) at i * ld + k) or row-contiguous (k * ld + i).
extern "C" int apn_pw_contract_splits(int nbatch, int r, int q, int k) { return apn::pw_weight_splits(nbatch, q, r, k); }

extern "C" int apn_pw_contract(int nbatch, int r, int q, int k, const float *a, long long a_batch, int lda, int a_kcont,
                               const float *b, long long b_batch, int ldb, int b_kcont, float *d, long long d_batch,
                               int ldd, int splits, float *scratch, int precision, void *stream) {
    using namespace apn;
    if (nbatch <= 0 || r <= 0 || q <= 0 || k <= 0 || nbatch > 65535 || !a || !b || !d || (precision != 2 && precision != 3) ||
        splits < 0 || (splits > 0 && (!scratch || ldd != q)))
        return APN_EINVAL;
    PwGemm g{};
    g.A = PwOperand{a, a_batch, lda};
    g.B = PwOperand{b, b_batch, ldb};
    g.R = r; g.Q = q; g.K = k;
    g.cpb = (k + PW_KC - 1) / PW_KC;
    g.total = nbatch * g.cpb;
    g.a_vec = a_kcont ? pw_vec(a, a_batch, lda, k, r) : pw_vec_rows(a, a_batch, lda, r, k);
    g.b_vec = b_kcont ? pw_vec(b, b_batch, ldb, k, q) : pw_vec_rows(b, b_batch, ldb, q, k);
    int nz = nbatch;
    if (splits > 0) {
        g.D = scratch; g.d_batch = (long long)r * q; g.ldd = q;
        g.cps = (g.total + splits - 1) / splits;
        nz = splits;
    } else {
        g.D = d; g.d_batch = d_batch; g.ldd = ldd;
        g.cps = g.cpb;
    }
    const dim3 grid((q + PW_T - 1) / PW_T, (r + PW_T - 1) / PW_T, nz);
    if (a_kcont && b_kcont) PW_LAUNCH(true, true, grid, g);
    else if (a_kcont) PW_LAUNCH(true, false, grid, g);
    else if (b_kcont) PW_LAUNCH(false, true, grid, g);
    else PW_LAUNCH(false, false, grid, g);
    if (splits > 0) {
        const size_t ne = (size_t)r * q;
        hipLaunchKernelGGL(pw_fold_kernel, dim3((unsigned)((ne + 63) / 64)), dim3(256), 0, (hipStream_t)stream, scratch,
                           splits, ne, d);
        APN_LAUNCH_CHECK();
    }
    return APN_OK;
}

// ---- layout change between the per-point layers (B, C, N) and the grouper / attention side (B, N, C) ----------------
// out[z][j][i] = in[z][i][j] for in (nbatch, r, c): 64 x 64 tiles through LDS, both sides in whole 256-byte lines.
// (The generator made these copies with PyTorch's strided copy kernel: 25 us for 16 MB, eight times per step.)
namespace apn {
__global__ __launch_bounds__(256) void pw_transpose_kernel(int r, int c, const float *__restrict__ in, float *__restrict__ out) {
    __shared__ float tile[64][65];
    const int z = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    in += (size_t)z * r * c;
    out += (size_t)z * r * c;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {                          // all sixteen loads requested together (clamped addresses)
        const int rr = r0 + ty * 16 + i, cc = c0 + tx;
        v[i] = in[(size_t)(rr < r ? rr : r - 1) * c + (cc < c ? cc : c - 1)];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[ty * 16 + i][tx] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = c0 + ty * 16 + i, rr = r0 + tx;
        if (cc < c && rr < r) out[(size_t)cc * r + rr] = tile[tx][ty * 16 + i];
    }
}

// three_nn's squared distances -> the interpolation weights of upsampling.py:97-100 in one launch:
// w_j = (1 / (sqrt(d2_j) + 1e-8)) / sum_j (1 / (sqrt(d2_j) + 1e-8)), the sum taken left to right.
__global__ __launch_bounds__(256) void idw3_kernel(long long n, const float *__restrict__ d2, float *__restrict__ w) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = 1.0f / (__fsqrt_rn(d2[3 * i + 0]) + 1e-8f);
    const float b = 1.0f / (__fsqrt_rn(d2[3 * i + 1]) + 1e-8f);
    const float c = 1.0f / (__fsqrt_rn(d2[3 * i + 2]) + 1e-8f);
    const float sum = (a + b) + c;
    w[3 * i + 0] = a / sum;
    w[3 * i + 1] = b / sum;
    w[3 * i + 2] = c / sum;
}
}  // namespace apn

extern "C" int apn_pw_transpose(int nbatch, int r, int c, const float *in, float *out, void *stream) {
    if (nbatch < 0 || r < 0 || c < 0) return APN_EINVAL;
    if (nbatch == 0 || r == 0 || c == 0) return APN_OK;
    if (!in || !out || nbatch > 65535 || (r + 63) / 64 > 65535) return APN_EINVAL;
    hipLaunchKernelGGL(apn::pw_transpose_kernel, dim3((c + 63) / 64, (r + 63) / 64, nbatch), dim3(256), 0, (hipStream_t)stream,
                       r, c, in, out);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_three_nn_weights(long long n, const float *dist2, float *weight, void *stream) {
    if (n < 0) return APN_EINVAL;
    if (n == 0) return APN_OK;
    if (!dist2 || !weight || (n + 255) / 256 > 0x7FFFFFFF) return APN_EINVAL;
    hipLaunchKernelGGL(apn::idw3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, dist2, weight);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
