// sa_fused.hip -- the grouped shared-MLP of a PointNeXt set-abstraction block as
// one MFMA pipeline per pass, for gfx950.
//
// What it replaces.  After FPS and ball query the reference block
// (openpoints/models/backbone/pointnext.py:157-168 with QueryAndGroup,
// openpoints/models/layers/group.py:235-255,323-335) materialises
//     fj = cat[(xyz[idx]-new_xyz)/r , f[idx]]         (B, 3+C, M, K)   73 MB @ B=32
//     y1 = Conv2d(3+C -> C1)(fj); a1 = ReLU(BN(y1))   (B, C1,  M, K)   67 MB x3
//     y2 = Conv2d(C1 -> C2)(a1);  z  = BN(y2)         (B, C2,  M, K)  134 MB x2
//     o  = max_K z                                    (B, C2,  M)
// in HBM (~0.5 GB moved forward, ~1 GB forward+backward).  Here no (.,M,K) tensor
// ever exists: every pass re-gathers the neighbourhood rows of a point-major bf16
// copy of f (4 MB, L2 resident), and runs the whole chain in registers.  A tile is
// one query = K = 32 positions; one wave owns a tile at a time.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16, f32 accumulate).  Operand lane maps
// (cdna_hip_programming.md section 3): lane l = (r = l&31, h = l>>5) holds
// A[row r][k = 8h+j] / B[k = 8h+j][col r], j = 0..7; C/D: col = lane&31,
// row(reg,h) = (reg&3) + 8*(reg>>2) + 4*h.  Chosen so that NO data ever moves
// between lanes inside the chain:
//   * the gather is per position, so lane (pos, h) naturally holds 8 consecutive
//     input channels of its position: X is an A operand (rows = positions) or, with
//     the two builtin arguments swapped, a B operand (cols = positions);
//   * conv1 is issued "transposed"  Y1^T = W1 * X^T : accumulator lane = position,
//     register = mid channel row(reg,h).  After BN+ReLU, registers 8s..8s+7 packed
//     to bf16 ARE the A fragment of k-step s of conv2 (the "accumulator as next
//     operand" rule), with k order row(8s+j,h); W2's fragments are built in that
//     same order once per wave;
//   * conv2 is issued "straight"  Y2 = A1 * W2^T : accumulator lane = output
//     channel, register = position -- so BatchNorm statistics (sum over positions)
//     and the max over the K neighbours are sums/maxima over a lane's OWN registers
//     plus one exchange between the two lane halves.
//
// BatchNorm in training mode needs batch statistics between a conv and its
// activation, so the forward is two passes over the tiles:
//   pass 1  gather, conv1                      -> per-channel sum / sum-of-squares of y1
//   pass 2  gather, conv1, BN1+ReLU, conv2     -> sum / sumsq of y2, and per (query, channel)
//                                                 the extreme of y2 over K and where it is
// BN2 is affine per channel, so max_K BN2(y2) = scale*ext_K(y2)+shift with ext = max
// when gamma2 >= 0 and min otherwise: no third pass.  Statistics leave the kernels
// as one partial row per workgroup; the small consumer kernels of sa_glue.hip sum the
// rows in float64 (deterministic).  At world_size > 1 the summed rows are what the
// SyncBatchNorm all-reduce runs on.
//
// Roofline: compulsory HBM traffic of pass 2 at B=32 is xyz 0.4 + idx 2.1 + ft 2.1
// + out 2 x 4.2 MB ~ 13 MB (1.6 us at 8 TB/s); bf16 MFMA work 7 x 32 cycles per tile.
// Both are far below the per-tile VALU/gather latency, which is what bounds it.
#include "apn_common.h"
#include "apn_mfma.h"
#include <type_traits>
#include "sa_chain.h"

namespace apn {

constexpr int SA_C = 32;     // feature channels in
constexpr int SA_C1 = 32;    // mid channels
constexpr int SA_C2 = 64;    // out channels
constexpr int SA_K = 32;     // neighbours per query = positions per tile
constexpr int SA_WAVES = 4;  // waves per workgroup

// Constant operand fragments kept in LDS instead of registers: fragment f of lane l is the
// 16-byte word [(f * NS + part) * 64 + l] -- one conflict-free ds_read_b128 per use.  The
// per-wave register footprint of the chain's constants (weights in both orientations, BN
// folds: ~130 VGPRs in the backward pass) is what pushed the kernels past 256 registers, i.e.
// to one wave per SIMD with a quarter of the vector instructions moving values to and from
// accumulator registers.
template <int NS>
__device__ __forceinline__ void put_frag(uint4 *base, int f, int lane, const Frag<NS> &v) {
#pragma unroll
    for (int p = 0; p < NS; ++p) base[(f * NS + p) * 64 + lane] = __builtin_bit_cast(uint4, v.p[p]);
}

template <int NS>
__device__ __forceinline__ Frag<NS> get_frag(const uint4 *base, int f, int lane) {
    Frag<NS> v;
#pragma unroll
    for (int p = 0; p < NS; ++p) v.p[p] = __builtin_bit_cast(bf16x8, base[(f * NS + p) * 64 + lane]);
    return v;
}

// The operand value the MFMA sees: rounded to bf16, or to hi + lo bf16 parts (split mode).
template <int NS>
__device__ __forceinline__ float eff(float v) {
    const __bf16 hi = (__bf16)v;
    float r = (float)hi;
    if (NS == 2) r += (float)(__bf16)(v - r);
    return r;
}

struct SaArgs {
    int b, n, m;                 // clouds, support points, queries
    const float *xyz;            // (B,N,3)
    const float *new_xyz;        // (B,M,3)
    const __bf16 *ft;            // (B,N,32) bf16
    const __bf16 *ft_lo;         // (B,N,32) bf16 remainders (split-operand mode), else null
    const int *idx;              // (B,M,32)
    const float *w1;             // (32, 35): columns [dp(3), f(32)] as the reference's cat([dp, fj])
    float radius;
    const int *tmap;             // distinct-hit tile map (csrc/sa_wide_glue.hip: apn_sa_wide_tilemap) or null
    unsigned long long *stamps;  // diagnostics (apn_sa_debug_stamps): per wave 8 wall-clock stamps, else null
};

// wall-clock stamp k of this wave (100 MHz counter, chip-wide), when a stamp buffer is attached
__device__ __forceinline__ void stamp(const SaArgs &a, int wave, int k) {
    if (a.stamps && (threadIdx.x & 63) == 0)
        a.stamps[((size_t)blockIdx.x * SA_WAVES + wave) * 16 + k] = wall_clock64();
}

// With a tile map (CP) a tile is 32 ROWS = (query, distinct neighbour, multiplicity) instead of one query's 32
// slots: the ball query's fill copies of slot 0 are folded into one row (3.7x fewer tiles at stage 1), whole
// queries packed in order.  Per row: rowinfo = qlocal | slot << 8 | mult << 16 (| queries of the tile << 24 in
// row 0; padding rows: mult 0, qlocal 255), rownn = the neighbour; per tile: tq0 = its first query.
__device__ __forceinline__ int tm_tiles(const SaArgs &a) { return a.tmap[0]; }
__device__ __forceinline__ const int *tm_tq0(const SaArgs &a) { return a.tmap + 4; }
__device__ __forceinline__ const unsigned *tm_rows(const SaArgs &a) {
    return reinterpret_cast<const unsigned *>(a.tmap + 4 + ((a.b * a.m + 3) & ~3));
}
__device__ __forceinline__ const int *tm_nn(const SaArgs &a) {
    return a.tmap + 4 + ((a.b * a.m + 3) & ~3) + (size_t)32 * a.b * a.m;
}
__device__ __forceinline__ unsigned ri_q(unsigned info) { return info & 0xffu; }
__device__ __forceinline__ unsigned ri_slot(unsigned info) { return (info >> 8) & 0xffu; }
__device__ __forceinline__ unsigned ri_mult(unsigned info) { return (info >> 16) & 0xffu; }
// the records of this lane's 16 accumulator rows acc_row(i, h) (row p's record sits in lane p)
__device__ __forceinline__ void row_meta(unsigned info, int h, unsigned (&meta)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) meta[i] = (unsigned)__builtin_amdgcn_ds_bpermute(acc_row(i, h) << 2, (int)info);
}

// Per-wave constant operand fragments of conv1: lane (r = mid channel, h), step s,
// element j  <->  input channel k = 16 s + 8 h + j: features 0..31, then dp x,y,z.
template <int NS>
__device__ __forceinline__ void load_w1_frags(const float *__restrict__ w1, int r, int h,
                                              Frag<NS> (&wf)[3]) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            t[j] = s < 2 ? w1[r * 35 + 3 + 16 * s + 8 * h + j]
                         : ((h == 0 && j < 3) ? w1[r * 35 + j] : 0.0f);
        wf[s] = make_frag<NS>(t);
    }
}

// Forward launch 1 of 3: the point-major operand table AND BatchNorm-1's batch statistics, per POINT.
//   ft[b][n][0..31] = bf16(f[b][:, n])  (+ the remainder table in split mode): one 64-byte row per point, so that
//   a neighbour gather is two 16-byte loads per lane instead of 32 strided 4-byte loads;
//   part[workgroup][64] = {sum y1, sum y1^2}[32] over the positions that gather the workgroup's points, from the
//   index stage's occurrence statistics (sa_geo.hip): with F = W1f f_n (operands as the MFMA sees them),
//   sum y1 = occ F + W1p D,  sum y1^2 = occ F^2 + 2 F W1p D  (+ W1p DD W1p^T once, added by workgroup 0).
// No pass over the positions: this replaced sa_prep_features + sa_fwd_stats1 (6.7 + 8.8 us at B = 32).
// A wave owns 32 points: their operand fragments (the hi / lo split of 16 channels per lane) ARE the table rows it
// writes, and F = X W1f^T is two MFMA k-steps (accumulator: lane = mid channel, register = point), so the
// statistics are 16 fused multiply-adds per lane.  At most PS_MAX_ROWS workgroups of 16 waves, each walking
// tiles of 512 points: few enough partial rows that every workgroup of the next launch folds them itself
// (no fold launch, fixed order: deterministic).
constexpr int PS_PTS = 512, PS_MAX_ROWS = 64, PS_NT = 2 * PS_PTS;   // points per tile, rows, threads (256 / 128 / 512: the per-point pass 4 us shorter, the forward pass's fold of 128 rows 2 us longer, the step 1 % slower)
constexpr double GEO_INV_UNIT = 1.0 / 68719476736.0;       // sa_geo.hip: D in units of 2^-36

#ifdef APN_WG_STAMPS
// (diagnostic builds only, scripts/stamp_glue.py) wall-clock stamps of a workgroup's phases, 16 per workgroup
__device__ unsigned long long *d_wg_stamps_f = nullptr;
__device__ __forceinline__ void wg_stamp(int k) {
    unsigned long long *st = d_wg_stamps_f;
    if (st && threadIdx.x == 0) st[(size_t)blockIdx.x * 16 + k] = wall_clock64();
}
#else
__device__ __forceinline__ void wg_stamp(int) {}
#endif

template <int NS>
__global__ __launch_bounds__(PS_NT) void sa_prep_stats_kernel(
    int n, int tiles_per_cloud, int total_tiles, const float *__restrict__ f, const long long *__restrict__ geo,
    const double *__restrict__ dd, int dd_rows, const float *__restrict__ w1, int stats, __bf16 *__restrict__ ft,
    __bf16 *__restrict__ ft_lo, float *__restrict__ part, unsigned long long *__restrict__ zero, long long zero_words) {
    extern __shared__ float sf_raw[];                            // [32][PS_PTS + 1] the tile, channel-major
    float (*sf)[PS_PTS + 1] = reinterpret_cast<float (*)[PS_PTS + 1]>(sf_raw);
    __shared__ float4 sgeo[PS_PTS];                              // {occ, Dx, Dy, Dz} of the tile's points
    __shared__ float red[PS_NT / 64][64];
    __shared__ double sdd[6];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    critical_stream_priority();
    wg_stamp(0);
    // accumulators of LATER launches (BatchNorm-2's sums) are cleared here
    for (long long e = (long long)blockIdx.x * PS_NT + tid; e < zero_words; e += (long long)gridDim.x * PS_NT) zero[e] = 0ull;
    Frag<NS> wf[3];                                              // W1f^T as B operand: lane = mid channel r (step 2 unused)
    load_w1_frags<NS>(w1, r, h, wf);
    const float wp0 = eff<NS>(w1[r * 35]), wp1 = eff<NS>(w1[r * 35 + 1]), wp2 = eff<NS>(w1[r * 35 + 2]);
    float s1 = 0.0f, s2 = 0.0f;                                  // this lane's mid channel r, its half's points
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int cloud = tile / tiles_per_cloud, n0 = (tile % tiles_per_cloud) * PS_PTS;
        {
            const int pt = tid & (PS_PTS - 1), cg = tid / PS_PTS;
            const bool in = n0 + pt < n;
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = in ? f[((size_t)cloud * SA_C + cg + 2 * k) * n + n0 + pt] : 0.0f;
            float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (stats && in && cg == 0) {
                const longlong2 *gp = reinterpret_cast<const longlong2 *>(geo + ((size_t)cloud * n + n0 + pt) * 4);
                const longlong2 g0 = gp[0], g1 = gp[1];
                gv = make_float4((float)g0.x, (float)((double)g0.y * GEO_INV_UNIT), (float)((double)g1.x * GEO_INV_UNIT),
                                 (float)((double)g1.y * GEO_INV_UNIT));
            }
            wg_stamp(1);
            __syncthreads();                                        // the previous tile's readers
#pragma unroll
            for (int k = 0; k < 16; ++k) sf[cg + 2 * k][pt] = v[k];
            if (cg == 0) sgeo[pt] = gv;
            wg_stamp(2);
        }
        __syncthreads();
        wg_stamp(3);
        const int pt = 32 * wave + r;                               // this lane's point: channels 8h.., 16 + 8h..
        float x0[8], x1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { x0[j] = sf[8 * h + j][pt]; x1[j] = sf[16 + 8 * h + j][pt]; }
        const Frag<NS> a0 = make_frag<NS>(x0), a1 = make_frag<NS>(x1);
        if (n0 + pt < n) {
            uint4 *row = reinterpret_cast<uint4 *>(ft + ((size_t)cloud * n + n0 + pt) * SA_C);
            row[h] = __builtin_bit_cast(uint4, a0.p[0]);
            row[2 + h] = __builtin_bit_cast(uint4, a1.p[0]);
            if (NS == 2) {
                uint4 *rl = reinterpret_cast<uint4 *>(ft_lo + ((size_t)cloud * n + n0 + pt) * SA_C);
                rl[h] = __builtin_bit_cast(uint4, a0.p[NS - 1]);
                rl[2 + h] = __builtin_bit_cast(uint4, a1.p[NS - 1]);
            }
        }
        wg_stamp(4);
        if (stats) {
            f32x16 y = {0};
            y = mfma<NS>(a0, wf[0], y);                             // F: lane = mid channel r, register = point
            y = mfma<NS>(a1, wf[1], y);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4 g = sgeo[32 * wave + acc_row(i, h)];
                const float PD = __builtin_fmaf(wp2, g.w, __builtin_fmaf(wp1, g.z, wp0 * g.y));
                const float oF = g.x * y[i];
                s1 += oF + PD;
                s2 += __builtin_fmaf(oF, y[i], 2.0f * y[i] * PD);
            }
        }
    }
    wg_stamp(5);
    if (!stats) return;
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) { red[wave][r] = s1; red[wave][32 + r] = s2; }
    if (blockIdx.x == 0 && wave >= 1 && wave < 7) {          // the batch's second moments: its clouds' shares
        // one wave per moment: lane l sums rows l, l + 64, ... in order, then a fixed butterfly -- the same bits in every
        // run (a single thread walking the rows one dependent load at a time held workgroup 0 back by 2.4 us)
        double s = 0.0;
        for (int e = lane; e < dd_rows; e += 64) s += dd[(size_t)e * 6 + (wave - 1)];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) sdd[wave - 1] = s;
    }
    __syncthreads();
    if (tid < 64) {
        float val = 0.0f;
#pragma unroll
        for (int w = 0; w < PS_NT / 64; ++w) val += red[w][tid];
        if (tid >= 32 && blockIdx.x == 0) {
            // the per-position term of sum y1^2: (W1p d)^2 summed over all positions = W1p DD W1p^T
            const double wx = wp0, wy = wp1, wz = wp2;           // tid & 31 == r
            const double t = wx * wx * sdd[0] + 2.0 * wx * wy * sdd[1] + 2.0 * wx * wz * sdd[2] + wy * wy * sdd[3] +
                             2.0 * wy * wz * sdd[4] + wz * wz * sdd[5];
            val += (float)t;
        }
        part[(size_t)blockIdx.x * 64 + tid] = val;
    }
    wg_stamp(6);
}

// Gathering one tile.  Lane (pos, h) fetches 2 x 16 bytes of its neighbour's bf16 row (and of
// the remainder row in split mode) and the neighbour's coordinates.  The loads of tile t+1
// are ISSUED (fetch_tile) before the arithmetic of tile t and only consumed (build_frags) one
// iteration later, and the neighbour indices are read two tiles ahead: the dependent
// idx -> row chain (two L2 round trips) is off the critical path of the tile loop.
template <int NS>
struct TileRaw {
    uint4 f[2], fl[NS];   // feature-row slices (fl: remainder row, split mode)
    float px, py, pz;     // neighbour coordinates
    float qx, qy, qz;     // query coordinates (wave-uniform without a tile map)
    int nb;               // neighbour index of this lane's position
    unsigned info;        // tile map: this lane's row record
    int q0;               // tile map: the tile's first query
};

// what is read two tiles ahead: the neighbour of this lane's position (and, with a tile map, its record)
struct TileHead {
    int nb, q0;
    unsigned info;
    int tile;             // the tile's number (wave-uniform)
};
template <bool CP>
__device__ __forceinline__ TileHead load_head(const SaArgs &a, int tile, int r) {
    TileHead t;
    t.tile = tile;
    if (CP) {
        t.nb = tm_nn(a)[(size_t)tile * SA_K + r];
        t.info = tm_rows(a)[(size_t)tile * SA_K + r];
        t.q0 = tm_tq0(a)[tile];
    } else {
        t.nb = a.idx[(size_t)tile * SA_K + r];
        t.info = 0x10000u | ((unsigned)r << 8);      // one query, slot r, multiplicity 1
        t.q0 = tile;
    }
    return t;
}

// Addresses are a kernel argument (scalar registers) plus a 32-BIT byte offset per lane: the compiler then uses the
// scalar-base + vector-offset form of the load; with 64-bit element indices it kept whole 64-bit addresses (base + lane
// part) in vector registers across the tile loop -- eight registers, spilled to scratch in the backward pass.  The
// entry points check that every table is smaller than 4 GB (sa_check).
template <typename T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

template <int NS>
__device__ __forceinline__ void fetch_tile(const SaArgs &a, const TileHead &hd, int h, TileRaw<NS> &t) {
    const int nb = hd.nb;
    const unsigned cloud = (unsigned)(hd.q0 / a.m);
    const unsigned pt = cloud * (unsigned)a.n + (unsigned)nb;
    const unsigned rowoff = pt * (unsigned)(SA_C * 2) + 16u * (unsigned)h;        // bytes into the bf16 table
    t.f[0] = ld_off<uint4>(a.ft, rowoff);
    t.f[1] = ld_off<uint4>(a.ft, rowoff + 32u);
    if (NS == 2) {
        t.fl[0] = ld_off<uint4>(a.ft_lo, rowoff);
        t.fl[NS - 1] = ld_off<uint4>(a.ft_lo, rowoff + 32u);
    }
    struct xyz3 { float x, y, z; };                                                 // one 12-byte load
    const xyz3 pp = ld_off<xyz3>(a.xyz, pt * 12u);
    t.px = pp.x; t.py = pp.y; t.pz = pp.z;
    const unsigned q = (unsigned)hd.q0 + (ri_mult(hd.info) ? ri_q(hd.info) : 0u);
    const xyz3 qq = ld_off<xyz3>(a.new_xyz, q * 12u);
    t.qx = qq.x; t.qy = qq.y; t.qz = qq.z;
    t.nb = nb;
    t.info = hd.info;
    t.q0 = hd.q0;
}

// Operand fragments of a fetched tile.  deff (optional) receives, in BOTH halves, the relative
// position as the MFMA sees it (rounded to the operand precision).
template <int NS>
__device__ __forceinline__ void build_frags(const SaArgs &a, const TileRaw<NS> &t, int h,
                                            Frag<NS> (&x)[3], float *deff = nullptr) {
    x[0].p[0] = __builtin_bit_cast(bf16x8, t.f[0]);
    x[1].p[0] = __builtin_bit_cast(bf16x8, t.f[1]);
    if (NS == 2) {
        x[0].p[NS - 1] = __builtin_bit_cast(bf16x8, t.fl[0]);
        x[1].p[NS - 1] = __builtin_bit_cast(bf16x8, t.fl[NS - 1]);
    }
    float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (h == 0 || deff) {
        // group.py:250-253: (grouped_xyz - query) then /= radius
        d[0] = (t.px - t.qx) / a.radius;
        d[1] = (t.py - t.qy) / a.radius;
        d[2] = (t.pz - t.qz) / a.radius;
    }
    if (deff) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const __bf16 hi = (__bf16)d[j];
            deff[j] = (float)hi;
            if (NS == 2) deff[j] += (float)(__bf16)(d[j] - (float)hi);
        }
        if (h != 0) d[0] = d[1] = d[2] = 0.0f;
    }
    x[2] = make_frag<NS>(d);
}

// The tile loop of a wave, software-pipelined as described above.  BODY(tile, raw) consumes one
// fetched tile.
// PROLOGUE() is the kernel's workgroup-wide start-up (constant fragments into LDS, the BatchNorm fold: barriers
// inside, so EVERY wave runs it, also one without tiles).  It runs AFTER the wave's first tile has been
// requested: the head of that tile is read speculatively, before the tile count is known (any tile number below
// the map's capacity B*M addresses allocated rows), so the start-up's own round trips (partial rows, weights)
// overlap the dependent head -> rows chain instead of preceding it.
// PRE() runs at the top of every iteration BEFORE the next tile's loads are issued, and once more
// after the last tile: the place for stores / atomics deferred from the previous tile.  The
// memory counter is in order, so anything issued between a prefetch and the wait for it is
// waited for too; deferred to here, the atomics of tile t have the whole of tile t+1 to retire.
// Make a tile head's three words ARRIVE here.  The wait the compiler places before a dependent use is vmcnt(0)
// whenever the number of memory operations issued since the load is not a compile-time constant (loops over a
// tile's queries, stores under wave-uniform branches): placed where only older loads are in flight it costs
// nothing; placed behind a tile's scatter atomics / pooled stores -- where the next prefetch used to meet it -- it
// drained those first (measured: 2.4 us between two tiles of the backward pass).
__device__ __forceinline__ void touch(const TileHead &hd) {
    asm volatile("" ::"v"(hd.nb), "v"(hd.info), "v"(hd.q0));
}

// LATE (backward pass over the tile map): ONE register set for the tile's rows.  The pass runs at the register
// file's limit (two waves of 256 registers per SIMD) and the second set did not fit: the compiler spilled the next
// tile's rows to scratch AS THEY WERE LOADED -- load, wait for it, store it, four to six times in a row at the top of
// every tile, 2.3 us of serial round trips (ISA; stamps).  Here the next tile's rows are requested at the END of a
// tile's arithmetic, into the registers its own rows left at the tile's start: one batch in flight during the tile's
// stores and atomics.  BODY must not read its `raw` argument after it has called `before_stores`.
// AHEAD(head) (LATE only) is called right before a tile's rows are requested -- one tile ahead of its arithmetic: the place
// for asynchronous copies into LDS of whatever else the tile needs (the memory counter is in order: once the rows have
// arrived, so has everything requested before them).
struct NoAhead {
    __device__ __forceinline__ void operator()(const TileHead &) const {}
};
template <int NS, bool CP, bool LATE = false, typename Pro, typename Pre, typename Body, typename Ahead = NoAhead>
__device__ __forceinline__ void for_each_tile(const SaArgs &a, int wave, int r, int h, Pro prologue, Pre pre, Body body,
                                              Ahead ahead = Ahead()) {
    // XCD-aware tile numbers: workgroups w and w + 8 share an XCD (round-robin placement: a speed assumption, never a
    // correctness one) and its L2.  Dealt out in launch order, a cloud's ~130 tiles went to all eight XCDs and every
    // XCD pulled every cloud's rows through its own L2 (PMC: 35 MB fetched for a 4 MB table).  Tile number
    // t = 8 u + x is therefore mapped to tile (x, u) of a layout in which XCD x owns the x-th EIGHTH of the tiles -- four
    // whole clouds at B = 32 -- so a cloud's rows are fetched by one L2 only.  (gridDim.x a multiple of 8: else identity.)
    const int cap = a.b * a.m, stride = gridDim.x * SA_WAVES;
    const int tiles = CP ? tm_tiles(a) : cap;
    const bool xcd = (gridDim.x & 7) == 0;
    const int per_x = (tiles + 7) >> 3;                         // tiles of one XCD's eighth
    auto remap = [&](int t) {                                    // launch-order number -> tile (>= tiles: none)
        if (!xcd) return t;
        const int wg = t / SA_WAVES, wv = t - wg * SA_WAVES;     // t = (workgroup-slot, wave)
        const int x = wg & 7, u = (wg >> 3) * SA_WAVES + wv;     // XCD, and the slot's number inside it
        return u < per_x ? x * per_x + u : tiles;
    };
    int slot = blockIdx.x * SA_WAVES + wave;                     // advances by `stride`
    int tile = remap(slot);
    // the heads of the wave's first TWO tiles, both before anything else (the second was requested after the
    // prologue: the first iteration then stalled a full memory round trip on it before its own tile's arithmetic)
    const TileHead hd0 = load_head<CP>(a, tile < tiles ? tile : 0, r);
    TileHead hd_nxt = load_head<CP>(a, remap(slot + stride) < tiles ? remap(slot + stride) : 0, r);
    TileRaw<NS> cur, nxt;
    const bool any = tile < tiles;                                 // wave-uniform
    stamp(a, wave, 1);
    if (any) {
        ahead(hd0);
        fetch_tile<NS>(a, hd0, h, cur);
    }
    prologue();
    stamp(a, wave, 2);
    if (!any) return;
    int nst = 3;
    for (; tile < tiles; slot += stride, tile = remap(slot)) {
        const bool more = remap(slot + stride) < tiles;        // wave-uniform
        const int tile2 = remap(slot + 2 * stride);
        if (LATE) {
            const TileHead hd_nxt2 = load_head<CP>(a, tile2 < tiles ? tile2 : tile, r);
            body(tile, cur, nst == 4, [&] {
                // (a scheduling fence: hoisted above the tile's last arithmetic these loads find no free registers and are
                // spilled to scratch as they arrive -- load, wait, store, one after the other)
                asm volatile("" ::: "memory");
                touch(hd_nxt);
                if (more) {
                    ahead(hd_nxt);
                    fetch_tile<NS>(a, hd_nxt, h, cur);
                }
            });
            stamp(a, wave, nst < 5 ? nst : 5);
            ++nst;
            hd_nxt = hd_nxt2;
            continue;
        }
        touch(hd_nxt);
        pre();
        if (more) fetch_tile<NS>(a, hd_nxt, h, nxt);
        const TileHead hd_nxt2 = load_head<CP>(a, tile2 < tiles ? tile2 : tile, r);
        body(tile, cur, nst == 4, [&] { touch(hd_nxt2); });
        stamp(a, wave, nst < 5 ? nst : 5);
        ++nst;
        if (more) cur = nxt;
        hd_nxt = hd_nxt2;
    }
    pre();
    stamp(a, wave, 6);
}

// Workgroup-level fold of per-lane statistics: vals[i] belongs to channel (i*32 + r) of this lane's half;
// halves and waves add.  Thread e < NV*32 returns the workgroup's sum of column e (the others 0), which it then
// adds into the launch's accumulator set (apn_common.h: acc_add -- integer atomics, order-independent).
template <int NV>
__device__ __forceinline__ float fold_partials(float (&vals)[NV], int lane, int wave) {
    __shared__ float red[SA_WAVES][NV][32];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float other = __shfl_xor(vals[i], 32);
        if (lane < 32) red[wave][i][lane] = vals[i] + other;
    }
    __syncthreads();
    float s = 0.0f;
    const int e = threadIdx.x;
    if (e < NV * 32) {
#pragma unroll
        for (int w = 0; w < SA_WAVES; ++w) s += red[w][e >> 5][e & 31];
    }
    return s;
}

// The BatchNorm fold inside its consumer: float64 column sums of part[rows][2 C] (C = 32: rows of 64 floats; a few
// dozen rows from sa_prep_stats_kernel), or already reduced sums[2 C + 2] = {sums, global count, world} (the
// SyncBatchNorm path), then the C channel constants by threads 0..C-1.  For a workgroup of 256 threads; fixed
// summation order: every workgroup computes the same bits.  Workgroup 0 also leaves pack[4][C] and updates the
// running buffers.  scale/shift are valid in threads 0..C-1 on return (no trailing barrier).
__device__ __forceinline__ void fold_rows32(const float *__restrict__ part, int rows, const double *__restrict__ sums,
                                            const BnArgs &bn, float *__restrict__ pack, float &scale, float &shift,
                                            double *scratch /* LDS, 17 x 64 doubles */) {
    double (*fred)[64] = reinterpret_cast<double (*)[64]>(scratch);
    double *ftot = scratch + 16 * 64;
    const int tid = threadIdx.x;
    double count = bn.count;
    double gpre = 1.0, bpre = 0.0;                 // requested with the partial rows, used behind the barriers
    if (tid < 32) {
        if (bn.gamma) gpre = (double)bn.gamma[tid];
        if (bn.beta) bpre = (double)bn.beta[tid];
    }
    if (bn.training) {
        if (part) {
            const int quad = tid & 15, rg = tid >> 4;
            const float4 *__restrict__ p4 = reinterpret_cast<const float4 *>(part) + quad;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int rr = rg;
            for (; rr + 7 * 16 < rows; rr += 8 * 16) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p4[(size_t)(rr + 16 * u) * 16];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w;
                }
            }
            for (; rr + 3 * 16 < rows; rr += 4 * 16) {       // (64 rows: exactly one batch of four loads per thread)
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = p4[(size_t)(rr + 16 * u) * 16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w;
                }
            }
            for (; rr < rows; rr += 16) {
                const float4 v = p4[(size_t)rr * 16];
                s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
            }
            fred[rg][4 * quad] = s0; fred[rg][4 * quad + 1] = s1; fred[rg][4 * quad + 2] = s2; fred[rg][4 * quad + 3] = s3;
            __syncthreads();
            if (tid < 64) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) s += fred[k][tid];
                ftot[tid] = s;
            }
        } else if (tid < 64) {
            ftot[tid] = sums[tid];
        }
        if (!part) count = sums[64];
        __syncthreads();
    }
    if (tid < 32) {
        if (tid == 0 && blockIdx.x == 0 && bn.training && bn.nbt) *bn.nbt += 1;
        bn_channel(bn, 32, tid, bn.training ? ftot[tid] : 0.0, bn.training ? ftot[32 + tid] : 0.0, count,
                   blockIdx.x == 0, pack, scale, shift, gpre, bpre);
    }
}

// Forward launch 2 of 3.  Prologue: BatchNorm-1 folded from the partial rows of sa_prep_stats_kernel (or from reduced
// sums); then per tile a1 = relu(bn1(conv1)), y2 = conv2(a1).  Outputs ysel/ksel (B,M,64): the extreme of y2 over K
// (max where gamma2 >= 0, else min) and its position; {sum[64], sumsq[64]} of y2 into the accumulator set acc2
// (cleared by the launch before); pack1 (workgroup 0) for the backward.
template <int NS, bool CP>
__global__ __launch_bounds__(SA_WAVES * 64, 3) void sa_fwd_main_kernel(
    SaArgs a, const float *__restrict__ w2, BnArgs bn1, const float *__restrict__ part1, int rows1,
    const double *__restrict__ sums1, float *__restrict__ pack1, const float *__restrict__ gamma2,
    float *__restrict__ ysel, unsigned char *__restrict__ ksel, unsigned long long *__restrict__ acc2) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    enum { F_W1 = 0, F_W2 = 3, F_COUNT = 7 };
    __shared__ uint4 cfrag[F_COUNT * NS * 64];
    __shared__ __attribute__((aligned(16))) float bn1v[2][2][16];   // {scale, shift}[h][register]
    // wave-private tiles [out-channel half][row][32 + 1] of the signed y2, for the pool per query over a tile map
    // (one tile per query: only the prologue's fold scratch lives here)
    constexpr int PT_WAVE = CP ? 2 * 32 * 33 : 17 * 64 * 2 / SA_WAVES;        // floats per wave
    __shared__ __attribute__((aligned(16))) float ptile[SA_WAVES * PT_WAVE];
    __shared__ __attribute__((aligned(16))) unsigned sinfo[SA_WAVES][32];
    static_assert(SA_WAVES * PT_WAVE * 4 >= 17 * 64 * 8, "fold scratch");
    critical_stream_priority();
    stamp(a, wave, 0);
    const float sg[2] = {(!gamma2 || gamma2[r] >= 0.0f) ? 1.0f : -1.0f, (!gamma2 || gamma2[32 + r] >= 0.0f) ? 1.0f : -1.0f};
    float st[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // sum t0, sum t1, sumsq t0, sumsq t1

    auto prologue = [&]() {
        if (wave == 0) {
            Frag<NS> w1f[3];
            load_w1_frags<NS>(a.w1, r, h, w1f);
#pragma unroll
            for (int s = 0; s < 3; ++s) put_frag<NS>(cfrag, F_W1 + s, lane, w1f[s]);
        }
        if (wave == 1 || wave == 2) {
            // conv2 B fragments: lane (out channel 32 t + r, h), step s, element j <-> mid channel row(8s+j, h)
            const int t = wave - 1;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float tmp[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tmp[j] = w2[(32 * t + r) * SA_C1 + acc_row(8 * s + j, h)];
                put_frag<NS>(cfrag, F_W2 + 2 * t + s, lane, make_frag<NS>(tmp));
            }
        }
        float sc, sh;
        fold_rows32(part1, rows1, sums1, bn1, pack1, sc, sh, reinterpret_cast<double *>(ptile));
        if (threadIdx.x < 32) {          // channel i sits in half (i >> 2) & 1, register (i & 3) + 4 (i >> 3)
            const int i = threadIdx.x;
            bn1v[0][(i >> 2) & 1][(i & 3) + 4 * (i >> 3)] = sc;
            bn1v[1][(i >> 2) & 1][(i & 3) + 4 * (i >> 3)] = sh;
        }
        __syncthreads();
    };

    for_each_tile<NS, CP>(a, wave, r, h, prologue, [] {}, [&](int tile, const TileRaw<NS> &raw, bool, auto before_stores) {
        int lane_o = lane;                       // see sa_bwd_kernel: fragments are read per use
        asm volatile("" : "+v"(lane_o));
        Frag<NS> x[3];
        build_frags<NS>(a, raw, h, x);
        const int q0 = __builtin_amdgcn_readfirstlane(raw.q0);
        const int nq = __builtin_amdgcn_readfirstlane((int)(raw.info >> 24));       // lane 0 holds row 0
        // tile map: multiplicities of this lane's sixteen accumulator rows (the row records go through a wave-private
        // LDS line: one write, four 16-byte broadcast reads) and where the queries' rows start
        float multf[16];
        unsigned starts = 1u;
        float *pt = ptile + wave * PT_WAVE;
        if (CP) {
            if (lane < 32) sinfo[wave][lane] = raw.info;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&sinfo[wave][8 * gq + 4 * (lane_o >> 5)]);
                multf[4 * gq] = (float)ri_mult(v.x); multf[4 * gq + 1] = (float)ri_mult(v.y);
                multf[4 * gq + 2] = (float)ri_mult(v.z); multf[4 * gq + 3] = (float)ri_mult(v.w);
            }
            const unsigned prev = (unsigned)__shfl_up((int)raw.info, 1);
            starts = (unsigned)__ballot(lane < 32 && (lane == 0 || ri_q(raw.info) != ri_q(prev)));
        }
        f32x16 y1 = {0};
#pragma unroll
        for (int s = 0; s < 3; ++s) y1 = mfma<NS>(get_frag<NS>(cfrag, F_W1 + s, lane_o), x[s], y1);  // Y1^T: lane = position
        {
            const float4 *scv = reinterpret_cast<const float4 *>(bn1v[0][lane_o >> 5]);
            const float4 *shv = reinterpret_cast<const float4 *>(bn1v[1][lane_o >> 5]);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const float4 sc = scv[i4], sh = shv[i4];
                y1[4 * i4] = __builtin_fmaxf(__builtin_fmaf(y1[4 * i4], sc.x, sh.x), 0.0f);
                y1[4 * i4 + 1] = __builtin_fmaxf(__builtin_fmaf(y1[4 * i4 + 1], sc.y, sh.y), 0.0f);
                y1[4 * i4 + 2] = __builtin_fmaxf(__builtin_fmaf(y1[4 * i4 + 2], sc.z, sh.z), 0.0f);
                y1[4 * i4 + 3] = __builtin_fmaxf(__builtin_fmaf(y1[4 * i4 + 3], sc.w, sh.w), 0.0f);
            }
        }
        const Frag<NS> a0 = pack8<NS>(y1, 0), a1 = pack8<NS>(y1, 8);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 y2 = {0};
            y2 = mfma<NS>(a0, get_frag<NS>(cfrag, F_W2 + 2 * t, lane_o), y2);  // Y2: lane = out channel 32 t + r, register = position
            y2 = mfma<NS>(a1, get_frag<NS>(cfrag, F_W2 + 2 * t + 1, lane_o), y2);
            if (CP) {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float wy = multf[i] * y2[i];
                    s1 += wy;
                    s2 += wy * y2[i];
                    pt[(t * 32 + acc_row(i, h)) * 33 + r] = sg[t] * y2[i];
                }
                st[t] += s1;
                st[2 + t] += s2;
                continue;
            }
            float best = sg[t] * y2[0];
            int bpos = 0;
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s1 += y2[i];
                s2 += y2[i] * y2[i];
                const float v = sg[t] * y2[i];
                if (i > 0 && v > best) { best = v; bpos = i; }
            }
            st[t] += s1;
            st[2 + t] += s2;
            int kpos = acc_row(bpos, h);
            const float obest = __shfl_xor(best, 32);
            const int okpos = __shfl_xor(kpos, 32);
            if (obest > best || (obest == best && okpos < kpos)) { best = obest; kpos = okpos; }
            if (h == 0) {
                ysel[(size_t)tile * SA_C2 + 32 * t + r] = sg[t] * best;
                ksel[(size_t)tile * SA_C2 + 32 * t + r] = (unsigned char)kpos;
            }
        }
        if (CP) {
            // The pool, query by query: lane (r, h) owns out channel 32 h + r and walks the query's rows of ITS tile in
            // ascending order (rows ascend with the slot, a row's slot = its offset from the query's first row: the
            // first maximum is the lowest slot).  ~5 instructions per row where the compare/select form over the
            // accumulator registers took ~85 per query and channel half.
            asm volatile("" ::: "memory");
            const float *mine = pt + h * (32 * 33) + r;
            const float sgn = h ? sg[1] : sg[0];
            float *ys = ysel + (size_t)q0 * SA_C2 + lane;
            unsigned char *ks = ksel + (size_t)q0 * SA_C2 + lane;
            before_stores();
            // all 32 rows are read, sixteen at a time into registers (independent LDS reads, one wait per half: a loop
            // over a query's rows paid one LDS round trip per row, ~1.2 us per tile); a row that starts a query
            // (wave-uniform bit of `starts`; the first padding row starts a segment of its own, never stored) first
            // flushes the query before it
            int seg = -1, ra = 0, kpos = 0;
            float best = 0.0f;
#pragma unroll
            for (int half16 = 0; half16 < 2; ++half16) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = mine[(16 * half16 + i) * 33];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rho = 16 * half16 + i;
                    if ((starts >> rho) & 1u) {
                        if (seg >= 0 && seg < nq) {
                            ys[(size_t)seg * SA_C2] = sgn * best;
                            ks[(size_t)seg * SA_C2] = (unsigned char)kpos;
                        }
                        ++seg;
                        best = -__builtin_inff();
                        kpos = 0;
                        ra = rho;
                    }
                    const bool up = v[i] > best;
                    best = up ? v[i] : best;
                    kpos = up ? rho - ra : kpos;
                }
            }
            if (seg >= 0 && seg < nq) {
                ys[(size_t)seg * SA_C2] = sgn * best;
                ks[(size_t)seg * SA_C2] = (unsigned char)kpos;
            }
            asm volatile("" ::: "memory");
        }
    });
    const float tot = fold_partials<4>(st, lane, wave);
    if (threadIdx.x < 128) acc_add(acc2, 128, blockIdx.x % ACC_COPIES, threadIdx.x, tot);
    stamp(a, wave, 7);
}

// ---------------------------------------------------------------------------
// Backward.  Given g_out = dL/d(max_K BN2(y2)) the caller forms, from (B,M,64)
// tensors only, BN2's two reduction terms S1 = sum g, S2 = sum g*yhat_sel and the
// per-channel constants of
//     dL/dy2 = goa[q,c] * [pos == ksel[q,c]]  +  y2 * D2[c] + E2[c]
// (goa = g * gamma2*invstd2; the dense part is BN's mean/variance feedback).  ONE
// pass re-runs the chain per tile and produces
//   dL/dW2's three ingredients (see below),
//   BN1's reduction terms T1 = sum g_u, T2 = sum g_u * yhat1, g_u = (dL/da1) * [a1 > 0],
//   A (B,N,32)  = g_u summed per SOURCE POINT (float atomics, de-duplicated),
//   geo (B,N,4) = {count, sum of relative positions} of each source point's occurrences,
//   HA, HB (B,M,32) = g_u and yhat1 summed per QUERY.
// dL/dy1 = g_u*ca + yhat1*cb + cc needs T1, T2 of the WHOLE batch (ca, cb, cc), but only its
// sums per point (G) and per query (H) are ever used -- everything downstream is linear in
// it: dL/df = G W1f, dL/dW1f = G^T f, the coordinate columns from G, H and the
// coordinates.  Those sums are linear in {g_u, yhat1, 1} too, and yhat1 is affine in the
// tile's inputs, so
//   G[n] = ca*A[n] + cb*inv1*(W1 [geo_xyz[n]; count[n] f_n] - count[n] mean1) + cc*count[n]
//   H[q] = ca*HA[q] + cb*HB[q] + cc*K
// are formed per point / per query by the consumer (sa_glue.hip: bwd_point_grads) once the
// batch constants exist: no second pass over the positions.
// dL/dW2[c][mid] = sum_pos dL/dy2[pos][c] a1[pos][mid] does not need y2 at all: with y2 = a1 W2^T,
//   sum_pos (y2 D2 + E2)[c] a1[mid] = D2[c] (W2 Gram)[c][mid] + E2[c] suma[mid],
//   Gram = sum_pos a1^T a1 (32x32: ONE MFMA product per tile, both operands the same registers),
//   suma = sum_pos a1, and the sparse part is goa[q][c] * a1[q, ksel[q][c], :] -- lane c fetches
//   that row of a1 from the lanes that hold it (ds_bpermute) into 32 exact f32 accumulators.
// The consumer (sa_glue.hip: bwd_consts1) assembles dL/dW2 from the three.  This replaced the
// recomputation of y2 and a (64 x 32 x positions) MFMA product: 42 instead of 60 MFMAs and about
// 30 % fewer vector instructions per tile (no y2 epilogue, no hi/lo split of y2).
// dL/da1 = dL/dy2 * W2 never needs y2 transposed: the dense part folds to
// a1 * (W2^T diag(D2) W2) + E2*W2 (a 32x32 matrix Qm and a vector, built by the
// caller), and the sparse part is a one-hot-weighted (pos x channel) operand built
// from ksel/goa -- both land as MFMAs in the [lane = mid channel, register = position]
// layout that the ReLU mask and BN1 terms already use.
struct SaBwdArgs {
    const float *w2;        // (64,32)
    const float *scale1, *shift1, *mean1, *inv1;   // BN1 fold and statistics [32] (pack1 of the forward)
    const float *pack2;     // BN2 {scale, shift, mean, invstd}[64] (the forward's)
    const unsigned long long *accS;   // accumulator set {S1 = sum g, S2 = sum g*yhat_sel}[64] of the backward entry, or
    const double *sumsS;              // ... the reduced sums {S[128], global count, world} (SyncBatchNorm)
    double count;           // positions on this rank
    int train2;
    const float *goa;       // (B,M,64)
    const unsigned char *ksel;  // (B,M,64)
    const int *rowdst;      // the row map (apn_sa_rowmap_many): the place of every tile-map row in the point-sorted order
};

// Backward launch 2 of 4.  Prologue (every workgroup, same bits): the per-channel constants of
//   dL/dy2 = goa*[pos==ksel] + y2*D2 + E2,  D2 = -scale2 inv2 S2 / P,  E2 = -scale2 S1 / P + scale2 mean2 inv2 S2 / P
// and their images through W2: Qm = W2^T diag(D2) W2 (32x32), evec = E2 W2 -- formerly a launch of its own.
// Epilogue: the workgroup's share of dL/dW2 = sparse part + D2 (W2 Gram) + E2 (x) suma as ONE partial row
// partW2[workgroup][64*32] (summed in float64, in a fixed order, by the last launch): no float atomics on dL/dW2.
// Round 5: NO float atomics.  A row's g_u (its 32 mid channels: one 128-byte line) is STORED at the row's place in the
// point-sorted order (g.rowdst, index-stage data like the tile map): the rows of a support point are then contiguous in GU
// and the per-point kernel sums them in ascending row order -- bit-reproducible gradients by construction, plain stores
// (rounds 1-4 added them into A (B,N,32) with memory-side float atomics: 15.9 MB of atomic traffic per launch, the
// accumulation target cleared by the forward's last launch; knocked out, the step was 6 us shorter, most of it BEHIND the
// kernel: the atomics' retirement and the next kernel's cold reads of their lines -- profiles/r05_scatter_knockouts.txt).
// The pass runs over the tile map only (CP): without one the caller builds it (adaptpoint_amd/fused.py).
template <int NS, bool CP>
__global__ __launch_bounds__(SA_WAVES * 64, 2) void sa_bwd_kernel(SaArgs a, SaBwdArgs g,
                                                               unsigned long long *__restrict__ accT,
                                                               float *__restrict__ partW2,
                                                               float *__restrict__ GU,
                                                               float *__restrict__ HA,
                                                               float *__restrict__ HB) {
    static_assert(CP, "the backward pass runs over a tile map");
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // constant fragments, built once per workgroup (wave w builds every fourth) into LDS
    enum { F_W1 = 0, F_QM = 3, F_W2T = 5, F_COUNT = 9 };
    __shared__ uint4 cfrag[F_COUNT * NS * 64];
    __shared__ __attribute__((aligned(16))) float bn1v[2][2][16];   // {scale, shift}[h][register]
    __shared__ __attribute__((aligned(16))) float sw2[SA_C2][SA_C1 + 4];   // W2 (prologue products, epilogue row)
    __shared__ float sD[SA_C2], sE[SA_C2];
    __shared__ __attribute__((aligned(16))) float sqm[SA_C1 + 1][SA_C1 + 4];   // Qm[k][mid]; row 32 = evec; later Gram | suma
    critical_stream_priority();
    stamp(a, wave, 0);
    // BatchNorm-1's constants of channel r and evec[r]: kept in LDS and READ PER TILE (five registers less across the loop:
    // the pass runs at the register file's limit, see for_each_tile)
    __shared__ float chan[5][32];                 // {scale1, shift1, mean1, inv1, evec}[channel]
    // sparse part of dL/dW2, S^T a1 over the tile's positions, as MFMA accumulators: dsp[t] holds
    // dL/dW2[c = 32 t + acc_row(i, h)][mid = r]  (round 3: 32 f32 accumulators per lane fed by 32 ds_bpermute + 32 FMAs
    // PER QUERY -- ~260 LDS-crossbar and vector instructions per tile; now 16 transposed LDS reads + 12 MFMAs per tile)
    f32x16 dsp[2] = {{0}, {0}};
    f32x16 gram = {0};        // sum_pos a1^T a1: row mid' = acc_row(i, h), column mid = r
    float suma = 0.0f;        // sum_pos a1[pos][mid = r] (this half's positions)
    float st[2] = {0.0f, 0.0f};

    // wave-private LDS image of the sparse operand: [hi | lo] tiles of 32 rows x 72 bf16
    // (64 channels + 8 pad: 144-byte rows keep the 16-byte fragment reads conflict-free)
    constexpr int SP_ROW = 72, SP_TILE = 32 * SP_ROW;
    // (tile map: the region also holds two [32][33] float tiles between the image's uses)
    constexpr int SP_WAVE = (CP && NS * SP_TILE * 2 < 2 * 32 * 33 * 4) ? 2 * 32 * 33 * 4 : NS * SP_TILE * 2;   // bytes
    constexpr int SP_BYTES = SA_WAVES * SP_WAVE, WRED_BYTES = SA_WAVES * SA_C2 * SA_C1 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char sp_raw[SP_BYTES > WRED_BYTES ? SP_BYTES : WRED_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned sinfo[SA_WAVES][32];
    // (tile map) goa / ksel of a tile's first four queries, copied here one tile ahead by two asynchronous global -> LDS
    // loads (four consecutive queries are 1024 / 256 contiguous bytes): no registers held across the previous tile
    __shared__ __attribute__((aligned(16))) float qgoa[CP ? SA_WAVES : 1][4 * SA_C2];
    __shared__ __attribute__((aligned(16))) unsigned char qksel[CP ? SA_WAVES : 1][4 * SA_C2];
    __bf16 *sp_img = reinterpret_cast<__bf16 *>(sp_raw + wave * SP_WAVE);

    // the places of a tile's rows (g.rowdst), copied here one tile ahead by an asynchronous global -> LDS load; two lines per
    // wave: the next tile's copy is issued before this tile's stores have read theirs
    __shared__ __attribute__((aligned(16))) int sdst[SA_WAVES][2][64];
    int ahead_n = 0, body_n = 0;              // tiles requested / processed by this wave (their parity picks the line)
    // GU[place][mid] = v  (`place`: a 32-bit row number; the address is the scalar base + a 32-bit byte offset: B * M * 32 * 128
    // < 2^32 is checked at the entry)
    auto gu_store = [&](int place, float v) {
        *reinterpret_cast<float *>(reinterpret_cast<char *>(GU) + (((unsigned)place << 7) + ((unsigned)r << 2))) = v;
    };

    auto prologue = [&]() {
        const int tid = threadIdx.x;
        {   // requested first: everything the constants need
            Frag<NS> w1f[3];
            if (wave == 0) load_w1_frags<NS>(a.w1, r, h, w1f);
            float wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wv[k] = g.w2[tid + 256 * k];
            float c1[4] = {0.f, 0.f, 0.f, 0.f};              // BatchNorm-1's constants of channel tid (threads 128..159)
            if (tid >= 128 && tid < 160) {
                c1[0] = g.scale1[tid - 128]; c1[1] = g.shift1[tid - 128]; c1[2] = g.mean1[tid - 128]; c1[3] = g.inv1[tid - 128];
            }
            double sv = 0.0, count = g.count;                // S[tid] (threads 0..127: one column each)
            float p_sc = 0.0f, p_mu = 0.0f, p_iv = 0.0f;
            if (tid < 128) sv = g.sumsS ? g.sumsS[tid] : acc_read(g.accS, 128, tid);
            if (g.sumsS) count = g.sumsS[128];
            if (tid < 64) { p_sc = g.pack2[tid]; p_mu = g.pack2[128 + tid]; p_iv = g.pack2[192 + tid]; }
            if (wave == 0) {
#pragma unroll
                for (int s = 0; s < 3; ++s) put_frag<NS>(cfrag, F_W1 + s, lane, w1f[s]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) sw2[(tid + 256 * k) >> 5][tid & 31] = wv[k];
            if (tid >= 128 && tid < 160) {
#pragma unroll
                for (int k = 0; k < 4; ++k) chan[k][tid - 128] = c1[k];
            }
            // S2 (threads 64..127) meets S1 (threads 0..63) in the same wave's other half: lanes l and l + 64 are
            // different waves, so it goes through LDS (sqm is free until the barrier below)
            double *sS2 = reinterpret_cast<double *>(&sqm[0][0]);
            if (tid >= 64 && tid < 128) sS2[tid - 64] = sv;
            __syncthreads();
            if (tid < 64) {
                const double sc = p_sc, mu = p_mu, iv = p_iv, s1 = sv, s2 = sS2[tid];
                double d = 0.0, e = 0.0;
                if (g.train2) {
                    const double rc = 1.0 / count;
                    d = -sc * iv * s2 * rc;
                    e = (-sc * s1 + sc * mu * iv * s2) * rc;
                }
                sD[tid] = (float)d;
                sE[tid] = (float)e;
            }
            if (tid >= 64 && tid < 96) {
                const int i = tid - 64;          // channel i sits in half (i >> 2) & 1, register (i & 3) + 4 (i >> 3)
                const float sci = g.scale1[i], shi = g.shift1[i];
                bn1v[0][(i >> 2) & 1][(i & 3) + 4 * (i >> 3)] = sci;
                bn1v[1][(i >> 2) & 1][(i & 3) + 4 * (i >> 3)] = shi;
            }
        }
        __syncthreads();
        {
            // term 3's B fragments of W2 with k = output channel in natural order 16 s + 8 h + j -- and, from the same
            // values, Qm = W2^T diag(D2) W2 and evec = E2 W2 as two small MFMA products (A = the D2-weighted /
            // the E2-in-row-0 image of W2): Qm lands as [lane = mid, register = mid' in accumulator-row order], which IS
            // the fragment order of term 1's B operand (the accumulator-as-next-operand rule): no LDS round trip.
            // (A 64-iteration multiply-add loop over LDS by all four waves took ~1.2 us of every workgroup's start.)
            // Round 5: k-step s of the two products by WAVE s -- its shares through the wave's own image region (free until
            // the first tile), summed by wave 2 in a fixed order: as ONE wave's four steps (8 LDS reads, three operand
            // splits and six MFMAs each, one behind the other) they were 2.2 of the prologue's 5 us with three waves idle.
            const int s = wave;
            float tmp[8], tmpd[8], tmpe[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * s + 8 * h + j;
                tmp[j] = sw2[c][r];
                tmpd[j] = tmp[j] * sD[c];
                tmpe[j] = r == 0 ? sE[c] : 0.0f;
            }
            const Frag<NS> wt = make_frag<NS>(tmp);
            put_frag<NS>(cfrag, F_W2T + s, lane, wt);
            const f32x16 zero16 = {0};
            const f32x16 qp = mfma<NS>(make_frag<NS>(tmpd), wt, zero16);
            const f32x16 ep = mfma<NS>(make_frag<NS>(tmpe), wt, zero16);
            float *scr = reinterpret_cast<float *>(sp_img);
#pragma unroll
            for (int e = 0; e < 16; ++e) scr[e * 64 + lane] = qp[e];
            if (h == 0) scr[16 * 64 + r] = ep[0];                 // row 0 of the product = this step's share of evec
        }
        __syncthreads();
        if (wave == 2) {
            f32x16 qacc = {0};
            float ev = 0.0f;
#pragma unroll
            for (int w = 0; w < SA_WAVES; ++w) {
                const float *pw = reinterpret_cast<const float *>(sp_raw + w * SP_WAVE);
#pragma unroll
                for (int e = 0; e < 16; ++e) qacc[e] += pw[e * 64 + lane];
                if (h == 0) ev += pw[16 * 64 + r];
            }
            put_frag<NS>(cfrag, F_QM, lane, pack8<NS>(qacc, 0));
            put_frag<NS>(cfrag, F_QM + 1, lane, pack8<NS>(qacc, 8));
            if (h == 0) chan[4][r] = ev;
        }
        __syncthreads();
        {   // the wave's image region starts cleared (behind the barrier: wave 2 has read the shares)
            uint4 *z = reinterpret_cast<uint4 *>(sp_img);
            for (int e = lane; e < SP_WAVE / 16; e += 64) z[e] = make_uint4(0u, 0u, 0u, 0u);
        }
    };

    auto scatter_pending = [] {};
    auto ahead = [&](const TileHead &hd) {
        if (!CP) return;
        typedef __attribute__((address_space(3))) void lds_void;
        typedef __attribute__((address_space(1))) const void glb_void;
        const int q0n = __builtin_amdgcn_readfirstlane(hd.q0);
        const int qbn = q0n < a.b * a.m - 4 ? q0n : a.b * a.m - 4;
        int l_o = lane;                                  // (opaque: keeps base + 16 * lane out of two hoisted register pairs)
        asm volatile("" : "+v"(l_o));
        const char *gb = reinterpret_cast<const char *>(g.goa) + (size_t)qbn * (SA_C2 * 4);
        const char *kb = reinterpret_cast<const char *>(g.ksel) + (size_t)qbn * SA_C2;
        __builtin_amdgcn_global_load_lds((glb_void *)(gb + (unsigned)(16 * l_o)), (lds_void *)&qgoa[wave][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(kb + (unsigned)(4 * l_o)), (lds_void *)&qksel[wave][0], 4, 0, 0);
        // the tile's 32 places (lanes 32..63 repeat them: never past the array's end)
        const char *db = reinterpret_cast<const char *>(g.rowdst) + (size_t)__builtin_amdgcn_readfirstlane(hd.tile) * (SA_K * 4);
        __builtin_amdgcn_global_load_lds((glb_void *)(db + (unsigned)(4 * (l_o & 31))), (lds_void *)&sdst[wave][ahead_n & 1][0], 4, 0, 0);
        ++ahead_n;
    };
    for_each_tile<NS, CP, CP>(a, wave, r, h, prologue, scatter_pending, [&](int tile, const TileRaw<NS> &raw, bool probe, auto before_stores) {
        auto st2 = [&](int k) { if (probe) stamp(a, wave, 8 + k); };
        st2(0);
        // the constant fragments are READ PER USE: an opaque copy of the lane id keeps the
        // compiler from hoisting these loop-invariant LDS reads back into ~130 registers
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        Frag<NS> x[3];
        build_frags<NS>(a, raw, h, x);

        const int q0 = __builtin_amdgcn_readfirstlane(raw.q0);
        const int nq = CP ? __builtin_amdgcn_readfirstlane((int)(raw.info >> 24)) : 1;
        const bool live_row = !CP || ri_mult(raw.info) != 0;
        const float mrow = CP ? (float)ri_mult(raw.info) : 1.0f;       // multiplicity of position r
        // (tile map) the gradients and pooled slots of the tile's first four queries are requested NOW, one batch
        // of loads: a query-by-query chain paid one memory round trip per query (~4 per tile)
        // (over the tile map they were copied into the wave's LDS buffer one tile ahead, with the tile's rows: `ahead` below)
        float gv4[4];
        int kc4[4];
        const int qb = q0 < a.b * a.m - 4 ? q0 : a.b * a.m - 4;        // first query of the copied block of four
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (CP) {
                gv4[u] = 0.0f;
                kc4[u] = 0;
            } else {
                gv4[u] = 0.0f;
                kc4[u] = 0;
                if (u == 0) {
                    gv4[u] = g.goa[(size_t)(q0 + u) * SA_C2 + lane];
                    kc4[u] = g.ksel[(size_t)(q0 + u) * SA_C2 + lane];
                }
            }
        }
        // tile map: multiplicities of this lane's sixteen accumulator rows acc_row(i, h) (the row records go through
        // a wave-private LDS line: one write, four 16-byte broadcast reads), and where the queries' rows start
        // (kept PACKED, four 8-bit multiplicities per register, and converted at each use -- one v_cvt_f32_ubyteN each:
        // sixteen floats held across the whole tile were what pushed the pass over 256 registers)
        unsigned mpk[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
        unsigned starts = 1u;                                          // bit = a query's first row (wave-uniform)
        if (CP) {
            if (lane < 32) sinfo[wave][lane] = raw.info;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&sinfo[wave][8 * gq + 4 * (lane_o >> 5)]);
                mpk[gq] = ri_mult(v.x) | (ri_mult(v.y) << 8) | (ri_mult(v.z) << 16) | (ri_mult(v.w) << 24);
            }
            const unsigned prev = (unsigned)__shfl_up((int)raw.info, 1);
            starts = (unsigned)__ballot(lane < 32 && (lane == 0 || ri_q(raw.info) != ri_q(prev)));
        }
        auto multf = [&](int i) { return (float)((mpk[i >> 2] >> (8 * (i & 3))) & 0xffu); };   // of accumulator row acc_row(i, h)
        // conv1 in both layouts (3 + 3 k-steps on the same fragments)
        f32x16 yT = {0}, y1 = {0};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const Frag<NS> w = get_frag<NS>(cfrag, F_W1 + s, lane_o);
            yT = mfma<NS>(w, x[s], yT);   // lane = position, register = mid channel
            y1 = mfma<NS>(x[s], w, y1);   // lane = mid channel, register = position
        }
        {
            const float4 *scv = reinterpret_cast<const float4 *>(bn1v[0][lane_o >> 5]);
            const float4 *shv = reinterpret_cast<const float4 *>(bn1v[1][lane_o >> 5]);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const float4 sc = scv[i4], sh = shv[i4];
                yT[4 * i4] = __builtin_fmaxf(__builtin_fmaf(yT[4 * i4], sc.x, sh.x), 0.0f);
                yT[4 * i4 + 1] = __builtin_fmaxf(__builtin_fmaf(yT[4 * i4 + 1], sc.y, sh.y), 0.0f);
                yT[4 * i4 + 2] = __builtin_fmaxf(__builtin_fmaf(yT[4 * i4 + 2], sc.z, sh.z), 0.0f);
                yT[4 * i4 + 3] = __builtin_fmaxf(__builtin_fmaf(yT[4 * i4 + 3], sc.w, sh.w), 0.0f);
            }
        }
        st2(1);
        // operand of the Qm product: a1, weighted by the row's multiplicity (the dense part of dL/dy2 reaches
        // every one of the positions the row stands for)
        Frag<NS> a0, a1;
        if (CP) {
            f32x16 yw;
#pragma unroll
            for (int i = 0; i < 16; ++i) yw[i] = yT[i] * mrow;
            a0 = pack8<NS>(yw, 0);
            a1 = pack8<NS>(yw, 8);
        } else {
            a0 = pack8<NS>(yT, 0);
            a1 = pack8<NS>(yT, 8);
        }

        // One-hot-weighted operand of the sparse part, S[pos][c] = goa[c] * [ksel[c] == pos]
        // (64 nonzeros in a 32 x 64 tile).  Built through a wave-private LDS image instead of
        // 2048 compare/selects: lane c drops its value into row ksel[c], every lane reads its
        // fragment rows back (ds_read_b128, conflict-free with 144-byte rows).  LDS executes one wave's
        // instructions in order, so the image needs no barrier; it starts zeroed and is left zeroed (one tile per
        // query: lane c zeroes its element again; tile map: the region is reused below and cleared whole).  With a
        // tile map the tile holds several queries: each drops into ITS rows (row0 of the query + the pooled slot).
        Frag<NS> sp[4];
        {
            unsigned todo = starts;                                     // consumed query by query
            if (CP) {
                // the block of four queries copied ahead starts at qb <= q0 (== q0 except at the very end of the tensor)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (older than the tile's rows: long arrived)
                const int dq = q0 - qb;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = (u + dq < 4 ? u + dq : 3) * SA_C2 + lane_o;
                    gv4[u] = qgoa[wave][e];
                    kc4[u] = qksel[wave][e];
                }
            }
#pragma unroll 1
            for (int jb = 0; jb < nq; jb += 4) {
                if (jb) {                                               // (rare) the tile's queries beyond the first four
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (jb + u < nq) {
                            gv4[u] = g.goa[(size_t)(q0 + jb + u) * SA_C2 + lane];
                            kc4[u] = g.ksel[(size_t)(q0 + jb + u) * SA_C2 + lane];
                        }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (jb + u < nq) {                                  // wave-uniform
                        const float gv = gv4[u];                        // lane = out channel c
                        int kc = kc4[u];
                        if (CP) {
                            kc += __builtin_ctz(todo);                  // first row of this query in the tile
                            todo &= todo - 1u;
                        }
                        const __bf16 ghi = (__bf16)gv;
                        const __bf16 glo = (__bf16)(gv - (float)ghi);
                        __bf16 *cell = sp_img + kc * SP_ROW + lane;
                        cell[0] = ghi;
                        if (NS == 2) cell[SP_TILE] = glo;
                        if (!CP) kc4[u] = kc;
                    }
                }
            }
        }
        asm volatile("" ::: "memory");      // (the image's cells were written through another type)
        {
            // a1 in the [lane = mid, register = position] layout: both operands of the Gram product and the B operand of
            // the sparse part of dL/dW2 (the k index -- positions in accumulator-row order -- pairs the same registers)
            f32x16 an;
            const float sc1 = chan[0][lane_o & 31], sh1 = chan[1][lane_o & 31];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                an[i] = __builtin_fmaxf(__builtin_fmaf(y1[i], sc1, sh1), 0.0f);
                if (CP) suma = __builtin_fmaf(multf(i), an[i], suma);
                else suma += an[i];
            }
            const Frag<NS> b0 = pack8<NS>(an, 0), b1 = pack8<NS>(an, 8);
            // Gram += a1^T a1 (one side weighted by the multiplicity)
            if (CP) {
                f32x16 aw;
#pragma unroll
                for (int i = 0; i < 16; ++i) aw[i] = an[i] * multf(i);
                gram = mfma<NS>(pack8<NS>(aw, 0), b0, gram);
                gram = mfma<NS>(pack8<NS>(aw, 8), b1, gram);
            } else {
                gram = mfma<NS>(b0, b0, gram);
                gram = mfma<NS>(b1, b1, gram);
            }
            // dL/dW2's sparse part += S^T a1: the A operand (rows = output channels, k = positions) is the image read
            // TRANSPOSED (ds_read_b64_tr_b16: a 16-lane group reads a block of 4 position rows x 16 channels and every
            // lane receives its channel's 4 positions; lane 4q + p of a group addresses row q, channels 4p .. 4p + 3).
            // Fragment (t, s) of lane (r, h): channel 32 t + r, positions acc_row(8 s + j, h) = rows 16 s + 4 h + j (j < 4)
            // and 16 s + 8 + 4 h + (j - 4): two reads per part, immediate offsets from one base address.
            typedef short tr_s4 __attribute__((ext_vector_type(4)));
            typedef short tr_s8 __attribute__((ext_vector_type(8)));
            typedef __attribute__((address_space(3))) tr_s4 tr_lds;
            const int li = lane_o & 15, gq = li >> 2, gp = li & 3, grh = (lane_o >> 4) & 1;
            const __bf16 *tb = sp_img + (4 * (lane_o >> 5) + gq) * SP_ROW + 16 * grh + 4 * gp;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    Frag<NS> at;
#pragma unroll
                    for (int pp = 0; pp < NS; ++pp) {
                        const __bf16 *src = tb + pp * SP_TILE + (16 * sk) * SP_ROW + 32 * t;
                        const tr_s4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds *)(src));
                        const tr_s4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds *)(src + 8 * SP_ROW));
                        const tr_s8 v8 = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                        at.p[pp] = __builtin_bit_cast(bf16x8, v8);
                    }
                    dsp[t] = mfma<NS>(at, sk ? b1 : b0, dsp[t]);
                }
            }
        }
        st2(2);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const __bf16 *src = sp_img + r * SP_ROW + 16 * s + 8 * h;
            sp[s].p[0] = *reinterpret_cast<const bf16x8 *>(src);
            if (NS == 2) sp[s].p[NS - 1] = *reinterpret_cast<const bf16x8 *>(src + SP_TILE);
        }
        asm volatile("" ::: "memory");      // (the region is accessed through several types: keep the phases in program order)
        if (!CP) {                                                      // one query per tile: its 64 cells again
            __bf16 *cell = sp_img + kc4[0] * SP_ROW + lane;
            cell[0] = (__bf16)0.0f;
            if (NS == 2) cell[SP_TILE] = (__bf16)0.0f;
        }
        // dL/da1 [lane = mid, register = position]
        f32x16 ga;
        const float ev = chan[4][lane_o & 31];
#pragma unroll
        for (int i = 0; i < 16; ++i) ga[i] = CP ? ev * multf(i) : ev;
        ga = mfma<NS>(a0, get_frag<NS>(cfrag, F_QM, lane_o), ga);
        ga = mfma<NS>(a1, get_frag<NS>(cfrag, F_QM + 1, lane_o), ga);
#pragma unroll
        for (int s = 0; s < 4; ++s) ga = mfma<NS>(sp[s], get_frag<NS>(cfrag, F_W2T + s, lane_o), ga);

        st2(3);
        // (tile map) g_u and multiplicity * yhat1 of every row also go to two wave-private LDS tiles [row][mid] (the
        // image's region: the image is dead until the next tile), for the sums per query below
        float *gt = reinterpret_cast<float *>(sp_img);
        asm volatile("" ::: "memory");
        float s1 = 0.0f, s2 = 0.0f, hb = 0.0f;
        const float sc1l = chan[0][lane_o & 31], sh1l = chan[1][lane_o & 31], iv1 = chan[3][lane_o & 31];
        const float nmi = -chan[2][lane_o & 31] * iv1;                  // yhat = y1 inv1 - mean1 inv1
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float u = __builtin_fmaf(y1[i], sc1l, sh1l);
            const float yhat = __builtin_fmaf(y1[i], iv1, nmi);
            ga[i] = u > 0.0f ? ga[i] : 0.0f;   // g_u
            s1 += ga[i];
            s2 += ga[i] * yhat;
            if (CP) {
                gt[acc_row(i, h) * 33 + r] = ga[i];
                gt[32 * 33 + acc_row(i, h) * 33 + r] = multf(i) * yhat;
            } else {
                hb += yhat;
            }
        }
        st[0] += s1;
        st[1] += s2;

        st2(4);
        before_stores();
        {   // sums per query and per source point
            int live;
            if (CP) {
                // a query's rows are consecutive: half 0 adds g_u (HA), half 1 multiplicity * yhat1 (HB), row by row
                // in ascending order (fixed: reproducible), each lane its mid channel
                asm volatile("" ::: "memory");
                const float *mine = gt + h * (32 * 33) + r;
                float *dst = (h ? HB : HA) + (size_t)q0 * SA_C1 + r;
                // (all 32 rows, sixteen at a time into registers: independent LDS reads, one wait per half -- a loop over
                // a query's rows paid one LDS round trip per row; a row that starts a query flushes the one before it)
                int seg = -1;
                float sum = 0.0f;
#pragma unroll
                for (int half16 = 0; half16 < 2; ++half16) {
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = mine[(16 * half16 + i) * 33];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if ((starts >> (16 * half16 + i)) & 1u) {
                            if (seg >= 0 && seg < nq) dst[(size_t)seg * SA_C1] = sum;
                            ++seg;
                            sum = 0.0f;
                        }
                        sum += v[i];
                    }
                }
                if (seg >= 0 && seg < nq) dst[(size_t)seg * SA_C1] = sum;
                // the image's region: cleared for the next tile
                asm volatile("" ::: "memory");
                {
                    uint4 *z = reinterpret_cast<uint4 *>(sp_img);
#pragma unroll
                    for (int e = 0; e < (NS * SP_TILE * 2 / 16 + 63) / 64; ++e)
                        if (e * 64 + lane < NS * SP_TILE * 2 / 16) z[e * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
                }
                asm volatile("" ::: "memory");
                // rows in use are a prefix of the tile, every one a distinct point of its query: no folding
                live = __popcll(__ballot(lane < 32 && live_row));
            } else {
                const float ha = s1 + __shfl_xor(s1, 32);
                hb += __shfl_xor(hb, 32);
                if (h == 0) {
                    HA[(size_t)tile * SA_C1 + r] = ha;
                    HB[(size_t)tile * SA_C1 + r] = hb;
                }
                live = 0;                                         // (not instantiated: the pass runs over a tile map)
            }

            st2(5);
            st2(6);
            if (CP) {
                // the tile's rows of g_u, STORED at their places in the point-sorted order at the END of the tile: behind every
                // load of this iteration and ahead of the next iteration's prefetch.  The places of this lane's sixteen rows
                // come back from the wave's LDS line (copied there one tile ahead, `ahead`): four 16-byte reads.  A wave
                // instruction writes two whole 128-byte lines (lanes 0..31 / 32..63: the 32 mid channels of two rows).
                const int *dl = &sdst[wave][body_n & 1][0];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (8 * gq < live) {                          // wave-uniform: is any lane's position live?
                        const int4 dd = *reinterpret_cast<const int4 *>(&dl[8 * gq + 4 * (lane_o >> 5)]);
                        const int base = 8 * gq + 4 * h;
                        if (base + 0 < live) gu_store(dd.x, ga[4 * gq + 0]);
                        if (base + 1 < live) gu_store(dd.y, ga[4 * gq + 1]);
                        if (base + 2 < live) gu_store(dd.z, ga[4 * gq + 2]);
                        if (base + 3 < live) gu_store(dd.w, ga[4 * gq + 3]);
                    }
                }
            }
            ++body_n;
        }
    }, ahead);
    {   // BatchNorm-1's reduction terms {T1, T2}[32] into their accumulator set
        const float tot = fold_partials<2>(st, lane, wave);
        if (threadIdx.x < 64) acc_add(accT, 64, blockIdx.x % ACC_COPIES, threadIdx.x, tot);
    }
    // The workgroup's share of dL/dW2: fold the four waves' sparse parts in LDS (kept in registers: thread t owns
    // elements t + 256 j), then Gram and suma, then  row[c][mid] = sparse + D2[c] (W2 Gram)[c][mid] + E2[c] suma[mid].
    float (*wred)[SA_C2 * SA_C1] = reinterpret_cast<float (*)[SA_C2 * SA_C1]>(sp_raw);   // images are dead
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        wred[wave][acc_row(i, h) * SA_C1 + r] = dsp[0][i];
        wred[wave][(32 + acc_row(i, h)) * SA_C1 + r] = dsp[1][i];
    }
    __syncthreads();
    // thread (c = tid >> 2, eight mids from mid0 = 8 (tid & 3)): its eight elements of the workgroup's row
    const int ec = threadIdx.x >> 2, emid = (threadIdx.x & 3) * 8;
    float sp8[8];
    {
        float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), hi4 = lo4;
#pragma unroll
        for (int w = 0; w < SA_WAVES; ++w) {
            const float4 a4 = *reinterpret_cast<const float4 *>(&wred[w][ec * SA_C1 + emid]);
            const float4 b4 = *reinterpret_cast<const float4 *>(&wred[w][ec * SA_C1 + emid + 4]);
            lo4.x += a4.x; lo4.y += a4.y; lo4.z += a4.z; lo4.w += a4.w;
            hi4.x += b4.x; hi4.y += b4.y; hi4.z += b4.z; hi4.w += b4.w;
        }
        sp8[0] = lo4.x; sp8[1] = lo4.y; sp8[2] = lo4.z; sp8[3] = lo4.w;
        sp8[4] = hi4.x; sp8[5] = hi4.y; sp8[6] = hi4.z; sp8[7] = hi4.w;
    }
    __syncthreads();
    suma += __shfl_xor(suma, 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) wred[wave][acc_row(i, h) * SA_C1 + r] = gram[i];
    if (h == 0) wred[wave][SA_C1 * SA_C1 + r] = suma;
    __syncthreads();
    for (int e = threadIdx.x; e < SA_C1 * SA_C1 + SA_C1; e += SA_WAVES * 64)       // Gram[mid'][mid] | suma: sqm is free
        sqm[e >> 5][e & 31] = (wred[0][e] + wred[1][e]) + (wred[2][e] + wred[3][e]);
    __syncthreads();
    {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll 8
        for (int k = 0; k < SA_C1; ++k) {
            const float w = sw2[ec][k];
            const float4 g0 = *reinterpret_cast<const float4 *>(&sqm[k][emid]);
            const float4 g1 = *reinterpret_cast<const float4 *>(&sqm[k][emid + 4]);
            acc[0] = __builtin_fmaf(w, g0.x, acc[0]); acc[1] = __builtin_fmaf(w, g0.y, acc[1]);
            acc[2] = __builtin_fmaf(w, g0.z, acc[2]); acc[3] = __builtin_fmaf(w, g0.w, acc[3]);
            acc[4] = __builtin_fmaf(w, g1.x, acc[4]); acc[5] = __builtin_fmaf(w, g1.y, acc[5]);
            acc[6] = __builtin_fmaf(w, g1.z, acc[6]); acc[7] = __builtin_fmaf(w, g1.w, acc[7]);
        }
        const float d = sD[ec], e = sE[ec];
        const float4 s0 = *reinterpret_cast<const float4 *>(&sqm[32][emid]);
        const float4 s1 = *reinterpret_cast<const float4 *>(&sqm[32][emid + 4]);
        float4 *row = reinterpret_cast<float4 *>(partW2 + (size_t)blockIdx.x * (SA_C2 * SA_C1) + ec * SA_C1 + emid);
        row[0] = make_float4(sp8[0] + d * acc[0] + e * s0.x, sp8[1] + d * acc[1] + e * s0.y,
                             sp8[2] + d * acc[2] + e * s0.z, sp8[3] + d * acc[3] + e * s0.w);
        row[1] = make_float4(sp8[4] + d * acc[4] + e * s1.x, sp8[5] + d * acc[5] + e * s1.y,
                             sp8[6] + d * acc[6] + e * s1.z, sp8[7] + d * acc[7] + e * s1.w);
    }
    stamp(a, wave, 7);
}

static int sa_grid(int tiles, bool compact = false) {
    // two workgroups of 4 waves per CU when there is enough work: 2 waves per SIMD; three over a tile map (a
    // wave then has ~1.5 tiles instead of 2.2: +3 % on the step, 512 / 768 / 1024 workgroups measured)
    const int cap = compact ? 768 : 512;
    int g = (tiles + SA_WAVES - 1) / SA_WAVES;
    return g < cap ? (g < 1 ? 1 : g) : cap;
}

// The backward pass fits two workgroups per CU (241 registers, 57 KB of LDS) and, alone, is
// fastest with 512; beside the sampler of the other stream (bench.py's pipeline) the best total
// is 448 -- most CUs doubly occupied, which hides the waits of this latency-bound kernel, with
// some room left on the CUs the sampler lives on (256 / 448 / 512 workgroups: 161.5k / 166.5k /
// 153k clouds/s, five interleaved runs each).
static int sa_grid_bwd(int tiles) {
    int g = sa_grid(tiles);
    return g < 448 ? g : 448;      // (round 3, over the tile map: 320 / 384 / 448 / 512 workgroups -> 224 / 231 / 233 / 230 k clouds/s)
}

}  // namespace apn

extern "C" int apn_sa_grid_blocks(int b, int m) { return apn::sa_grid(b * m); }
extern "C" int apn_sa_grid_rows(int b, int m, int with_tile_map) { return apn::sa_grid(b * m, with_tile_map != 0); }
extern "C" int apn_sa_bwd_main_rows(int b, int m) { return apn::sa_grid_bwd(b * m); }
extern "C" int apn_sa_acc_words(int ncol) { return ncol > 0 ? apn::acc_words(ncol) : 0; }

extern "C" int apn_sa_prep_rows(int b, int n) {
    const long long tiles = (long long)b * ((n + apn::PS_PTS - 1) / apn::PS_PTS);
    return (int)(tiles < apn::PS_MAX_ROWS ? (tiles < 1 ? 1 : tiles) : apn::PS_MAX_ROWS);
}

// ft holds `precision` tables of (B,N,32) bf16 back to back: [hi] or [hi][lo].
extern "C" int apn_sa_prep_stats(int b, int n, const float *f, const void *geo, const void *dd, const float *w1,
                                 int precision, int stats, void *ft, float *part1, void *zero, long long zero_words,
                                 void *stream) {
    using namespace apn;
    if (b <= 0 || n <= 0 || b > 65535 || (precision != 1 && precision != 2) || !f || !ft || !w1) return APN_EINVAL;
    if (stats && (!geo || !dd || !part1 || ((uintptr_t)geo & 15) || ((uintptr_t)part1 & 15))) return APN_EINVAL;
    if (zero_words < 0 || (zero_words && !zero)) return APN_EINVAL;
    __bf16 *hi = (__bf16 *)ft;
    __bf16 *lo = precision == 2 ? hi + (size_t)b * n * SA_C : nullptr;
    const int tpc = (n + PS_PTS - 1) / PS_PTS;
    auto kern = precision == 2 ? sa_prep_stats_kernel<2> : sa_prep_stats_kernel<1>;
    const int lds = SA_C * (PS_PTS + 1) * (int)sizeof(float);
    static DynLdsOnce lds_set[2];                             // (per precision and device: apn_common.h)
    if (hipError_t e = set_dyn_lds(lds_set[precision - 1], (const void *)kern, lds)) return (int)e;
    hipLaunchKernelGGL(kern, dim3(apn_sa_prep_rows(b, n)), dim3(PS_NT), lds, (hipStream_t)stream, n, tpc, b * tpc, f,
                       (const long long *)geo, (const double *)dd, b * (apn_sa_geo_dd_doubles(n) / 6), w1, stats, hi, lo, part1,
                       (unsigned long long *)zero, zero_words);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

#ifdef APN_WG_STAMPS
extern "C" __attribute__((visibility("default"))) int apn_sa_debug_wg_stamps_fused(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(apn::d_wg_stamps_f), &buf, sizeof(buf));
}
#endif

static unsigned long long *g_stamps = nullptr;
// Diagnostics: attach (or detach: NULL) a buffer of 16 x (workgroups x 4) 64-bit stamps; the next launches of the two
// tile passes record per wave the wall clock at {0: entry, 1: tile count known, 2: prologue done, 3..5: tiles done,
// 6: loop done, 7: kernel end} (scripts/stamp_passes.py).  Not for concurrent use.
extern "C" int apn_sa_debug_stamps(void *buf) { g_stamps = (unsigned long long *)buf; return APN_OK; }

static apn::SaArgs sa_args(int b, int n, int m, const float *xyz, const float *new_xyz, const void *ft,
                           int precision, const int *idx, const float *w1, float radius, const int *tmap) {
    const __bf16 *hi = (const __bf16 *)ft;
    return apn::SaArgs{b, n, m, xyz, new_xyz, hi,
                       precision == 2 ? hi + (size_t)b * n * apn::SA_C : nullptr, idx, w1, radius, tmap, g_stamps};
}

static int sa_check(int b, int n, int m, int precision) {
    if (precision != 1 && precision != 2) return APN_EINVAL;
    if (b <= 0 || n <= 0 || m <= 0) return APN_EINVAL;
    if ((long long)b * m > 0x7fffffffLL / 64) return APN_EINVAL;
    if ((long long)b * n > 0xffffffffLL / 256) return APN_EINVAL;     // 32-bit byte offsets: bf16 tables (fetch_tile)
    if ((long long)b * m > 0xffffffffLL / (32 * 128)) return APN_EINVAL;   // ... and GU's rows (sa_bwd_kernel: gu_store)
    return APN_OK;
}

extern "C" int apn_sa_fwd_main(int b, int n, int m, int precision, float radius, const float *xyz,
                               const float *new_xyz, const void *ft, const int *idx, const int *tmap, const float *w1,
                               const float *w2, const float *g1, const float *b1, float *rm1, float *rv1, void *nbt1,
                               float eps1, float mom1, int train1, double count, const float *part1, int rows1,
                               const double *sums1, float *pack1, const float *gamma2, float *ysel, void *ksel,
                               void *acc2, void *stream) {
    using namespace apn;
    if (int e = sa_check(b, n, m, precision)) return e;
    if (!xyz || !new_xyz || !ft || !idx || !w1 || !w2 || !pack1 || !ysel || !ksel || !acc2) return APN_EINVAL;
    if (train1 && !part1 && !sums1) return APN_EINVAL;
    if (!train1 && (!rm1 || !rv1)) return APN_EINVAL;
    if (part1 && (((uintptr_t)part1 & 15) || rows1 <= 0)) return APN_EINVAL;
    SaArgs a = sa_args(b, n, m, xyz, new_xyz, ft, precision, idx, w1, radius, tmap);
    BnArgs bn{g1, b1, rm1, rv1, (long long *)nbt1, eps1, mom1, train1, count};
    auto kern = precision == 2 ? (tmap ? sa_fwd_main_kernel<2, true> : sa_fwd_main_kernel<2, false>)
                               : (tmap ? sa_fwd_main_kernel<1, true> : sa_fwd_main_kernel<1, false>);
    hipLaunchKernelGGL(kern, dim3(sa_grid(b * m, tmap != nullptr)), dim3(SA_WAVES * 64), 0, (hipStream_t)stream, a, w2, bn,
                       sums1 ? nullptr : part1, rows1, sums1, pack1, gamma2, ysel, (unsigned char *)ksel,
                       (unsigned long long *)acc2);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

extern "C" int apn_sa_bwd_main(int b, int n, int m, int precision, float radius, const float *xyz,
                               const float *new_xyz, const void *ft, const int *idx, const int *tmap, const float *w1,
                               const float *w2, const float *pack1, const float *pack2, const void *accS,
                               const double *sumsS, double count, int train2, const float *goa, const void *ksel,
                               void *accT, float *partW2, const int *rowdst, float *GU, float *HA, float *HB, void *stream) {
    using namespace apn;
    if (int e = sa_check(b, n, m, precision)) return e;
    if (!w2 || !pack1 || !pack2 || (!accS && !sumsS) || !goa || !ksel || !accT || !partW2 || !GU || !HA || !HB)
        return APN_EINVAL;
    if (!tmap || !rowdst) return APN_EINVAL;         // the pass runs over the tile map and stores through its row map
    SaArgs a = sa_args(b, n, m, xyz, new_xyz, ft, precision, idx, w1, radius, tmap);
    SaBwdArgs g;
    g.w2 = w2;
    g.scale1 = pack1; g.shift1 = pack1 + 32; g.mean1 = pack1 + 64; g.inv1 = pack1 + 96;
    g.pack2 = pack2;
    g.accS = (const unsigned long long *)accS; g.sumsS = sumsS;
    g.count = count; g.train2 = train2;
    g.goa = goa; g.ksel = (const unsigned char *)ksel;
    g.rowdst = rowdst;
    auto kern = precision == 2 ? sa_bwd_kernel<2, true> : sa_bwd_kernel<1, true>;
    hipLaunchKernelGGL(kern, dim3(sa_grid_bwd(b * m)), dim3(SA_WAVES * 64), 0, (hipStream_t)stream, a, g,
                       (unsigned long long *)accT, partW2, GU, HA, HB);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
