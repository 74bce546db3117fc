// fps.hip -- furthest point sampling for gfx950 (one workgroup per cloud).
//
// Replaces furthest_point_sampling_kernel<block_size> and its launcher
// (openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:101-260).
//
// FPS is a serial chain: M-1 dependent steps, each "update N running minimum
// distances, then arg-max".  Only B workgroups exist, so neither HBM nor MFMA
// bounds it -- the latency of ONE step does.  The design removes everything a
// step can do without:
//   * the cloud (x, y, z), the running min-distances and the point ids live in
//     VGPRs for the whole kernel (S "slots" per lane); global memory is touched
//     once on entry and once on exit;
//   * the arg-max is an order-free integer max over the float bit patterns
//     (min-distances are >= 0, so uint order == float order): six DPP steps per
//     wave, no LDS;
//   * waves meet through ONE raw s_barrier per step and a double-buffered
//     16-byte LDS record per wave {max, x, y, z} -- the winner's coordinates
//     travel with its distance, so the next step needs no dependent lookup;
//     they are then held in SGPRs (v_readlane) and feed the VALU as scalar
//     operands.
//
// Tie rule.  The reference resolves equal maxima through its block tree
// (sampling_gpu.cu:93-98,146-210): among equal values the winner is the thread
// with the smallest bit-reversed id, then the lowest k inside that thread
// (strict > at :143-144).  Here points are laid out by that priority: position
// q = tid*S + slot holds the point of priority rank q, so "lowest position
// wins" -- which ballot + find-first-set gives for free at every level --
// reproduces the reference order exactly, with no key compares in the loop.
#include "apn_common.h"
#include "ball_query_body.h"

#include <cmath>
#include <cstdlib>

namespace apn {

// Geometry of the reference launch for n points: block size bs = 2^L
// (cuda_utils.h:10-14), every reference thread t owns points t, t+bs, ...:
// `full` of them, plus one more when t < rem.
struct FpsOrder {
    int n, bs, L, full, rem;
};

// 1e10f, the value callers pre-fill `temp` with (subsample.py:94).
constexpr unsigned FPS_BIG_BITS = 0x501502F9u;

// Number of points owned by reference threads whose L-bit reversed id is < R.
__device__ __forceinline__ int fps_rank_start(const FpsOrder &o, int R) {
    int g = 0;
    if (o.rem > 0) {
        // R' < R first differs from R at a set bit i of R; its i low bits are
        // free and land, reversed, in the top i bits of bitrev(R').
        for (int i = o.L - 1; i >= 0; --i) {
            if ((R >> i) & 1) {
                const int hi = R >> (i + 1);
                const int hb = o.L - 1 - i;
                const int c = hb > 0 ? (int)(__brev((unsigned)hi) >> (32 - hb)) : 0;
                if (o.rem > c) {
                    int cnt = (o.rem - c + (1 << (o.L - i)) - 1) >> (o.L - i);
                    const int cap = 1 << i;
                    g += cnt < cap ? cnt : cap;
                }
            }
        }
    }
    return R * o.full + g;
}

// Point id that has priority rank q (0 <= q < n).
__device__ __forceinline__ int fps_rank_to_point(const FpsOrder &o, int q) {
    int R, k;
    if (o.rem == 0) {
        R = q / o.full;
        k = q - R * o.full;
    } else {
        int lo = 0, hi = o.bs - 1;  // largest R with start(R) <= q
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (fps_rank_start(o, mid) <= q) lo = mid; else hi = mid - 1;
        }
        R = lo;
        k = q - fps_rank_start(o, R);
    }
    const int t = o.L > 0 ? (int)(__brev((unsigned)R) >> (32 - o.L)) : 0;
    return t + k * o.bs;
}

template <int W>
__device__ __forceinline__ unsigned group_max_u32(unsigned v) {
    // max over aligned groups of W lanes (W <= 16), valid in every lane.
    if (W >= 2) v = dpp_max_u32<DPP_QUAD_XOR1>(v);
    if (W >= 4) v = dpp_max_u32<DPP_QUAD_XOR2>(v);
    if (W >= 8) v = dpp_max_u32<DPP_ROW_HALF_MIRROR>(v);
    if (W >= 16) v = dpp_max_u32<DPP_ROW_MIRROR>(v);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// W waves per workgroup, S point slots per lane; requires n <= 64*W*S.
template <int W, int S>
__global__ __launch_bounds__(W * 64) void fps_reg_kernel(FpsOrder o, int m,
                                                         const float *__restrict__ xyz,
                                                         float *__restrict__ temp,
                                                         int *__restrict__ idxs,
                                                         float *__restrict__ new_xyz) {
    const int n = o.n;
    const int cloud = blockIdx.x;
    xyz += (size_t)cloud * n * 3;
    if (temp) temp += (size_t)cloud * n;   // null: start from 1e10 everywhere, no write-back
    idxs += (size_t)cloud * m;
    if (new_xyz) new_xyz += (size_t)cloud * m * 3;   // optional: coordinates of the picks

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    float px[S], py[S], pz[S];
    unsigned dmin[S];
    int pid[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = tid * S + s;
        if (q < n) {
            const int p = fps_rank_to_point(o, q);
            pid[s] = p;
            px[s] = xyz[p * 3 + 0];
            py[s] = xyz[p * 3 + 1];
            pz[s] = xyz[p * 3 + 2];
            dmin[s] = temp ? __float_as_uint(temp[p]) : FPS_BIG_BITS;
        } else {
            // Padding: min-distance stays +0.0 and sits at the highest
            // positions, so it can only tie, and a tie goes to a real point.
            pid[s] = 0;
            px[s] = py[s] = pz[s] = 0.0f;
            dmin[s] = 0u;
        }
    }

    __shared__ float4 rec[2][W];
    __shared__ int rec_pid[2][W];

    float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];  // old = 0 (sampling_gpu.cu:118-122)
    if (tid == 0) {
        idxs[0] = 0;
        if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
    }

    for (int j = 1; j < m; ++j) {
        unsigned best;
        float bx, by, bz;
        int bp;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float d = dist2(px[s] - x1, py[s] - y1, pz[s] - z1);
            // min on the bit patterns: both operands are >= +0 (or a NaN, whose
            // pattern is above every finite one), so this equals fminf(d, dmin)
            // and needs no canonicalising v_max in front of a v_min_f32.
            const unsigned du = __float_as_uint(d);
            dmin[s] = du < dmin[s] ? du : dmin[s];
            if (s == 0) {
                best = dmin[0]; bx = px[0]; by = py[0]; bz = pz[0]; bp = pid[0];
            } else {
                const bool g = dmin[s] > best;  // strict: the lower slot keeps a tie
                best = g ? dmin[s] : best;
                bx = g ? px[s] : bx;
                by = g ? py[s] : by;
                bz = g ? pz[s] : bz;
                bp = g ? pid[s] : bp;
            }
        }
        const unsigned wmax = wave_max_u32(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int wl = (int)__builtin_ctzll(cand);  // lowest lane = highest priority
        int old;
        if (W == 1) {
            x1 = readlane_f(bx, wl);
            y1 = readlane_f(by, wl);
            z1 = readlane_f(bz, wl);
            old = __builtin_amdgcn_readlane(bp, wl);
        } else {
            const int buf = j & 1;
            if (lane == wl) {
                rec[buf][wave] = make_float4(__uint_as_float(wmax), bx, by, bz);
                rec_pid[buf][wave] = bp;
            }
            // One rendezvous per step.  A raw barrier: __syncthreads() would
            // also drain the idxs store below (vmcnt) on every step.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const float4 e = rec[buf][lane & (W - 1)];
            const int ep = rec_pid[buf][lane & (W - 1)];
            const unsigned ev = __float_as_uint(e.x);
            const unsigned gmax = group_max_u32<W>(ev);
            const unsigned long long cw = __ballot(ev == gmax);
            const int ws = (int)__builtin_ctzll(cw);  // lowest wave = highest priority
            x1 = readlane_f(e.y, ws);
            y1 = readlane_f(e.z, ws);
            z1 = readlane_f(e.w, ws);
            old = __builtin_amdgcn_readlane(ep, ws);
        }
        if (tid == 0) idxs[j] = old;
        // the winner's coordinates are already wave-uniform: the gather of the sampled points
        // (pointnext.py:147) is one 12-byte store, issued by the LAST wave so that wave 0,
        // which already stores the index, is not the straggler at the next barrier
        if (new_xyz && tid >= (W - 1) * 64 && tid < (W - 1) * 64 + 3) {
            const int d = tid - (W - 1) * 64;
            new_xyz[j * 3 + d] = d == 0 ? x1 : d == 1 ? y1 : z1;
        }
    }

    // The reference leaves the final min-distances in temp (sampling_gpu.cu:141-142).
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = tid * S + s;
        if (temp && q < n) temp[pid[s]] = __uint_as_float(dmin[s]);
    }
}

// Second-generation step for n <= 4096: the cross-wave arg-max is ONE LDS 64-bit atomic
// max instead of per-wave records + a second reduction.
//   key = (min-distance bits << 32) | (0xFFFFFFFF - priority rank)
// so the maximum key is the largest distance and, among equals, the smallest rank: the
// reference's tie rule falls out of the integer compare, across lanes AND waves, with no
// election (every lane that holds its wave's maximum issues the atomic; normally one).
// After the barrier every lane reads the winning key (broadcast ds_read_b64) and then the
// winner's {x,y,z,id} from a rank-indexed LDS table (broadcast ds_read_b128): the next
// step's coordinates arrive in VGPRs, with no ballot / find-first / readlane on the
// critical path.  Three key slots rotate; slot (j+1)%3 is zeroed during step j, after
// barrier j-1 proved that every wave finished reading it and before barrier j lets
// anyone use it again.
// NEST (the index pyramids: block k + 1 samples from block k's samples).  FPS is PROGRESSIVE: run on the first sample's
// picks, in pick order, it returns picks 0, 1, 2, ... of that sample again -- pick j is the point of largest distance to
// picks 0 .. j-1 over ALL points, it lies in the sample, so it is also the largest over the sample, and the running
// minima are the same float operations on the same coordinates.  That holds whenever every arg-max was UNIQUE; among
// equal maxima the winner depends on the tie rule, which is positional (bit-reversed thread id) and differs between the
// two arrays.  So the step also records the FIRST step at which its maximum was not unique (or was 0: the cloud is
// exhausted): tie_out[cloud].  A following level of m' <= that many picks IS the prefix 0 .. m'-1 and costs a copy
// (tie_prev given and tie_prev[cloud] >= m); any other cloud runs the full sampler, bit-exact as ever.
template <int W, int S, bool NEST = false>
__device__ __forceinline__ void fps_atomic_body(const FpsOrder &o, int m,
                                                const float *__restrict__ xyz,
                                                float *__restrict__ temp,
                                                int *__restrict__ idxs,
                                                float *__restrict__ new_xyz, int cloud,
                                                float4 *tab /* LDS [n] {x, y, z, id} by priority rank */,
                                                const int *__restrict__ tie_prev = nullptr,
                                                int *__restrict__ tie_out = nullptr) {
    __shared__ unsigned long long slot[3];
    __shared__ int tie_first;
    const int n = o.n;
    if (NEST && tie_prev && tie_prev[cloud] >= m) {       // workgroup-uniform: the prefix of the previous level's picks
        const float *src = xyz + (size_t)cloud * n * 3;
        for (int i = threadIdx.x; i < m; i += W * 64) idxs[(size_t)cloud * m + i] = i;
        for (int i = threadIdx.x; i < m * 3; i += W * 64) new_xyz[(size_t)cloud * m * 3 + i] = src[i];
        if (threadIdx.x == 0 && tie_out) tie_out[cloud] = tie_prev[cloud];
        return;
    }
    int my_tie = 0x7fffffff;                                // wave-uniform: this wave's first ambiguous step
    xyz += (size_t)cloud * n * 3;
    if (temp) temp += (size_t)cloud * n;   // null: start from 1e10 everywhere, no write-back
    idxs += (size_t)cloud * m;
    if (new_xyz) new_xyz += (size_t)cloud * m * 3;
    const int tid = threadIdx.x;

    float px[S], py[S], pz[S];
    unsigned dmin[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = tid * S + s;
        if (q < n) {
            const int p = fps_rank_to_point(o, q);
            px[s] = xyz[p * 3 + 0];
            py[s] = xyz[p * 3 + 1];
            pz[s] = xyz[p * 3 + 2];
            dmin[s] = temp ? __float_as_uint(temp[p]) : FPS_BIG_BITS;
            tab[q] = make_float4(px[s], py[s], pz[s], __int_as_float(p));
        } else {
            px[s] = py[s] = pz[s] = 0.0f;      // padding: distance stays +0.0 at the highest ranks
            dmin[s] = 0u;
        }
    }
    constexpr unsigned long long FPS_KEY_NONE = 0x00000000FFFFFFFFull;       // (distance +0.0, rank 0)
    if (tid < 3) slot[tid] = FPS_KEY_NONE;
    if (NEST && tid == 0) tie_first = 0x7fffffff;
    float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
    if (tid == 0) {
        idxs[0] = 0;
        if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
    }
    __syncthreads();

    // LDS byte offsets of the three key slots (ds_max_u64 / ds_write_b64 take raw offsets)
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    const unsigned slot0 = (unsigned)(size_t)(lds_u64 *)&slot[0];
    int cur = 1, nxt = 2;                      // j % 3 and (j + 1) % 3
    for (int j = 1; j < m; ++j) {
        unsigned best = 0u;
        int bslot = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float d = dist2(px[s] - x1, py[s] - y1, pz[s] - z1);
            const unsigned du = __float_as_uint(d);
            dmin[s] = du < dmin[s] ? du : dmin[s];
            if (s == 0) {
                best = dmin[0];
            } else {
                const bool g = dmin[s] > best;   // strict: the lower slot keeps a tie
                best = g ? dmin[s] : best;
                bslot = g ? s : bslot;
            }
        }
        const unsigned wmax = wave_max_u32(best);
        bool wave_tie = false;                             // this wave holds its maximum more than once (wave-uniform)
        if (NEST) {
            int c = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) c += dmin[s] == wmax ? 1 : 0;
            const unsigned long long once = __ballot(c >= 1), twice = __ballot(c >= 2);
            wave_tie = twice != 0ull || (once & (once - 1ull)) != 0ull;
        }
        if (best == wmax && best != 0u) {
            // hand-issued so that the compiler's atomic optimizer does not wrap the (almost
            // always single-lane) atomic in a scalar reduction loop.  A wave whose maximum is +0.0 stays out: every
            // one of its lanes ties there (points already picked, duplicates of picked points -- the augmentor
            // moves its masked points to the origin, half of a generated cloud), and 64 lanes x 8 waves on ONE LDS
            // address cost microseconds per step (1024 -> 512 on such clouds: 276 us against 162).  Such a wave can
            // only win when EVERY distance is zero, and then the winner is rank 0 -- which is what an untouched slot
            // holds (FPS_KEY_NONE = the key of distance +0.0 at rank 0).
            const unsigned long long key =
                ((unsigned long long)best << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(tid * S + bslot));
            asm volatile("ds_max_u64 %0, %1" :: "v"(slot0 + 8u * (unsigned)cur), "v"(key) : "memory");
        }
        if (tid == 0) slot[nxt] = FPS_KEY_NONE;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned long long k = slot[cur];
        const unsigned rank = 0xFFFFFFFFu - (unsigned)k;               // (an untouched slot holds FPS_KEY_NONE: rank 0)
        const float4 w = tab[rank];
        x1 = w.x; y1 = w.y; z1 = w.z;
        if (NEST) {
            // the winning distance was held more than once: inside its owner's wave, or by another wave too; or it is 0
            const unsigned wdist = (unsigned)(k >> 32);
            const int owner_wave = (int)((rank / (unsigned)S) >> 6);
            const bool amb = wdist == 0u || (wmax == wdist && (wave_tie || owner_wave != (tid >> 6)));
            if (amb && j < my_tie) my_tie = j;
        }
        if (tid == 0) idxs[j] = __float_as_int(w.w);
        if (new_xyz && tid >= (W - 1) * 64 && tid < (W - 1) * 64 + 3) {
            const int d = tid - (W - 1) * 64;
            new_xyz[j * 3 + d] = d == 0 ? x1 : d == 1 ? y1 : z1;
        }
        cur = nxt;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = tid * S + s;
        if (temp && q < n) temp[__float_as_int(tab[q].w)] = __uint_as_float(dmin[s]);
    }
    if (NEST && tie_out) {
        if ((tid & 63) == 0 && my_tie != 0x7fffffff) atomicMin(&tie_first, my_tie);
        __syncthreads();
        if (tid == 0) tie_out[cloud] = tie_first;
    }
}

// The pyramid's sampler (see NEST above): level 1 with tie_prev == null records the first ambiguous step; deeper levels
// take the previous level's record and are a copy wherever it allows.
template <int W, int S>
__global__ __launch_bounds__(W * 64) void fps_nested_kernel(FpsOrder o, int m, const float *__restrict__ xyz,
                                                            const int *__restrict__ tie_prev, int *__restrict__ idxs,
                                                            float *__restrict__ new_xyz, int *__restrict__ tie_out) {
    extern __shared__ float4 tab[];
    fps_atomic_body<W, S, true>(o, m, xyz, nullptr, idxs, new_xyz, blockIdx.x, tab, tie_prev, tie_out);
}

template <int W, int S>
__global__ __launch_bounds__(W * 64) void fps_atomic_kernel(FpsOrder o, int m,
                                                            const float *__restrict__ xyz,
                                                            float *__restrict__ temp,
                                                            int *__restrict__ idxs,
                                                            float *__restrict__ new_xyz) {
    extern __shared__ float4 tab[];
    fps_atomic_body<W, S>(o, m, xyz, temp, idxs, new_xyz, blockIdx.x, tab);
}

// The index stage of the fused set-abstraction block as ONE launch with two roles: workgroups
// [0, b_fps) run farthest point sampling of one batch (one cloud each, ~160 us of dependent
// steps on b_fps CUs), the rest run the ball query of ANOTHER batch whose samples already exist
// (~20 us on the other CUs).  Back to back on one queue the two kernels cost their sum; here
// the search hides inside the sampler's latency.  Workgroups are dispatched in index order, so
// the samplers start first.
struct BallArgs {
    int b, n, m, nsample, q_per_block, blocks_x;
    float radius2;
    const float *new_xyz, *xyz;
    int *idx;
};

template <int S>
__global__ __launch_bounds__(512) void fps_ball_kernel(FpsOrder o, int b_fps, int m,
                                                       const float *__restrict__ xyz,
                                                       int *__restrict__ idxs,
                                                       float *__restrict__ new_xyz, BallArgs q) {
    extern __shared__ float4 dyn4[];
    if ((int)blockIdx.x < b_fps) {
        fps_atomic_body<8, S>(o, m, xyz, nullptr, idxs, new_xyz, blockIdx.x, dyn4);
    } else {
        const int t = blockIdx.x - b_fps;
        ball_query_body<8>(q.n, q.m, q.radius2, q.nsample, q.q_per_block, 1, q.new_xyz, q.xyz, q.idx,
                           t / q.blocks_x, t % q.blocks_x, reinterpret_cast<float *>(dyn4));
    }
}

// Diagnostic twin of fps_reg_kernel<8,2> (n = 1024): identical step, with s_memtime
// stamps around its segments accumulated by wave 0 into dbg[0..5] (cycles summed over the
// m-1 steps).  Never used by the product path; its run time is not representative
// (the stamps serialise), only the SHARES are (cdna_hip_programming.md section 7).
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

__global__ __launch_bounds__(512) void fps_stamp_kernel(FpsOrder o, int m, const float *__restrict__ xyz,
                                                        float *__restrict__ temp,
                                                        int *__restrict__ idxs,
                                                        unsigned long long *__restrict__ dbg) {
    constexpr int W = 8, S = 2;
    const int n = o.n;
    const int cloud = blockIdx.x;
    xyz += (size_t)cloud * n * 3;
    if (temp) temp += (size_t)cloud * n;   // null: start from 1e10 everywhere, no write-back
    idxs += (size_t)cloud * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float px[S], py[S], pz[S];
    unsigned dmin[S];
    int pid[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = tid * S + s;
        const int p = q < n ? fps_rank_to_point(o, q) : 0;
        pid[s] = p;
        px[s] = q < n ? xyz[p * 3 + 0] : 0.f;
        py[s] = q < n ? xyz[p * 3 + 1] : 0.f;
        pz[s] = q < n ? xyz[p * 3 + 2] : 0.f;
        dmin[s] = q < n ? (temp ? __float_as_uint(temp[p]) : FPS_BIG_BITS) : 0u;
    }
    __shared__ float4 rec[2][W];
    __shared__ int rec_pid[2][W];
    float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
    if (tid == 0) idxs[0] = 0;
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 1; j < m; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t0 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        unsigned best; float bx, by, bz; int bp;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float d = dist2(px[s] - x1, py[s] - y1, pz[s] - z1);
            const unsigned du = __float_as_uint(d);
            dmin[s] = du < dmin[s] ? du : dmin[s];
            if (s == 0) { best = dmin[0]; bx = px[0]; by = py[0]; bz = pz[0]; bp = pid[0]; }
            else {
                const bool g = dmin[s] > best;
                best = g ? dmin[s] : best; bx = g ? px[s] : bx; by = g ? py[s] : by;
                bz = g ? pz[s] : bz; bp = g ? pid[s] : bp;
            }
        }
        asm volatile("" :: "v"(best));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned wmax = wave_max_u32(best);
        asm volatile("" :: "s"(wmax));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long cand = __ballot(best == wmax);
        const int wl = (int)__builtin_ctzll(cand);
        const int buf = j & 1;
        if (lane == wl) {
            rec[buf][wave] = make_float4(__uint_as_float(wmax), bx, by, bz);
            rec_pid[buf][wave] = bp;
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t3 = stamp();   // includes the lgkmcnt(0) wait for the ds_write
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t4 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        const float4 e = rec[buf][lane & (W - 1)];
        const int ep = rec_pid[buf][lane & (W - 1)];
        asm volatile("" :: "v"(e.x), "v"(ep));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t5 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned ev = __float_as_uint(e.x);
        const unsigned gmax = group_max_u32<W>(ev);
        const unsigned long long cw = __ballot(ev == gmax);
        const int ws = (int)__builtin_ctzll(cw);
        x1 = readlane_f(e.y, ws); y1 = readlane_f(e.z, ws); z1 = readlane_f(e.w, ws);
        const int old = __builtin_amdgcn_readlane(ep, ws);
        if (tid == 0) idxs[j] = old;
        asm volatile("" :: "s"(x1), "s"(y1), "s"(z1));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t6 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2;
        acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t6 - t5;
    }
    if (tid == 0 && cloud == 0)
        for (int i = 0; i < 6; ++i) dbg[i] = acc[i];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = tid * S + s;
        if (temp && q < n) temp[pid[s]] = __uint_as_float(dmin[s]);
    }
}

// Any-n fallback (n > 16384): the reference's thread structure with the
// distances kept in `temp` (global / L2) and a halving tree in LDS whose merge
// keeps the left slot on ties, exactly the reference's __update order.
__global__ __launch_bounds__(1024) void fps_stream_kernel(FpsOrder o, int m,
                                                          const float *__restrict__ xyz,
                                                          float *__restrict__ temp,
                                                          int *__restrict__ idxs) {
    const int n = o.n, bs = o.bs;
    const int cloud = blockIdx.x;
    xyz += (size_t)cloud * n * 3;
    temp += (size_t)cloud * n;
    idxs += (size_t)cloud * m;
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const int tid = threadIdx.x;
    int old = 0;
    if (tid == 0) idxs[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        float best = -1.0f;
        int besti = 0;
        for (int k = tid; k < n; k += bs) {
            const float d = dist2(xyz[k * 3 + 0] - x1, xyz[k * 3 + 1] - y1, xyz[k * 3 + 2] - z1);
            const float d2 = __builtin_fminf(d, temp[k]);
            temp[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
        sv[tid] = best;
        si[tid] = besti;
        __syncthreads();
        for (int half = bs >> 1; half >= 1; half >>= 1) {
            if (tid < half) {
                const float a = sv[tid], c = sv[tid + half];
                if (c > a) { sv[tid] = c; si[tid] = si[tid + half]; }
            }
            __syncthreads();
        }
        old = si[0];
        __syncthreads();  // everyone has read si[0] before the next step rewrites it
        if (tid == 0) idxs[j] = old;
    }
}

// algo: 0 = LDS-atomic step when it applies; 1 = per-wave records (first generation).  A per-call
// argument (the tuned entry point below), never process state: the operators are re-entrant.
template <int W, int S>
static int launch_reg(const FpsOrder &o, int b, int m, const float *xyz, float *temp, int *idxs,
                      float *new_xyz, int algo, hipStream_t st) {
    if (W > 1 && o.n <= 4096 && algo == 0) {
        hipLaunchKernelGGL((fps_atomic_kernel<W, S>), dim3(b), dim3(W * 64), sizeof(float4) * o.n, st,
                           o, m, xyz, temp, idxs, new_xyz);
        APN_LAUNCH_CHECK();
        return APN_OK;
    }
    hipLaunchKernelGGL((fps_reg_kernel<W, S>), dim3(b), dim3(W * 64), 0, st, o, m, xyz, temp, idxs,
                       new_xyz);
    APN_LAUNCH_CHECK();
    return APN_OK;
}

template <int W>
static int dispatch_slots(const FpsOrder &o, int b, int m, const float *xyz, float *temp,
                          int *idxs, float *new_xyz, int algo, hipStream_t st) {
    const int need = (o.n + W * 64 - 1) / (W * 64);
#define APN_FPS_CASE(SS) \
    if (need <= SS) return launch_reg<W, SS>(o, b, m, xyz, temp, idxs, new_xyz, algo, st);
    APN_FPS_CASE(1)
    APN_FPS_CASE(2)
    APN_FPS_CASE(3)
    APN_FPS_CASE(4)
    APN_FPS_CASE(6)
    APN_FPS_CASE(8)
    APN_FPS_CASE(12)
    APN_FPS_CASE(16)
#undef APN_FPS_CASE
    return APN_EINVAL;
}

}  // namespace apn

static apn::FpsOrder fps_order(int n) {
    apn::FpsOrder o;
    o.n = n;
    {   // cuda_utils.h:10-14, same double arithmetic
        const int pow_2 = (int)(std::log((double)n) / std::log(2.0));
        int v = 1 << pow_2;
        if (v > 1024) v = 1024;
        if (v < 1) v = 1;
        o.bs = v;
    }
    o.L = 0;
    while ((1 << o.L) < o.bs) ++o.L;
    o.full = n / o.bs;
    o.rem = n % o.bs;
    return o;
}

// Which step the operator entries run: the LDS-atomic step (algo 0; 314 against 370 ns per step at 1024 points).
// History worth keeping (profiles/r04_packed_fp32_op_sel.md): BESIDE MFMA-heavy kernels (the benches'
// index stream next to the MLP stream, two lanes of a captured training step) this step returned wrong picks for ~2 % of the
// clouds -- one spurious arg-max per affected cloud -- while every isolated test was bit-exact.  The cause was not in this
// file's logic: the compiler's SLP vectoriser had turned the distance update into packed-FP32 instructions, among them
// `v_pk_add_f32 v[6:7], v[12:13], v[6:7] op_sel:[0,1]` (py[0..1] - y1 with the centre's y in the pair's HIGH register), and
// that form -- an op_sel bit on a VGPR pair -- intermittently computes its low lane as `py[0] - 0` while another stream's MFMA
// kernels are resident.  (Round 3 suspected the z update's `op_sel_hi:[1,0]` form; round 4 edited the failing build
// instruction by instruction: with only the 132 op_sel forms of this file replaced by scalar pairs it passes, with only the
// 642 op_sel_hi forms replaced it fails; a synthetic probe reproduces it.)  The library is compiled with -fno-slp-vectorize
// (adaptpoint_amd/build.py) and tests/test_host_cpu.py asserts that no kernel holds the form.
// APN_FPS_RECORDS=1 (read once) selects the per-wave-record step, which never had packed instructions (its centre lives in SGPRs).
static int fps_default_algo() {
    static const int algo = [] { const char *e = getenv("APN_FPS_RECORDS"); return (e && atoi(e) == 1) ? 1 : 0; }();
    return algo;
}

// waves: 0 = heuristic, else the number of waves per cloud (1, 2, 4, 8, 16); algo: see launch_reg.
static int fps_impl(int b, int n, int m, const float *xyz, float *temp, int *idxs, float *new_xyz,
                    int waves, int algo, void *stream) {
    using namespace apn;
    if (b < 0) return APN_EINVAL;
    if (b == 0 || m <= 0) return APN_OK;  // sampling_gpu.cu:110
    if (n <= 0 || !xyz || !idxs) return APN_EINVAL;
    if (!temp && (!new_xyz || n > 16384)) return APN_EINVAL;   // only the _xyz entry may omit temp
    hipStream_t st = (hipStream_t)stream;

    const FpsOrder o = fps_order(n);

    if (n > 16384) {
        hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(o.bs), 0, st, o, m, xyz, temp, idxs);
        APN_LAUNCH_CHECK();
        return APN_OK;
    }
    if (waves != 0 && waves != 1 && waves != 2 && waves != 4 && waves != 8 && waves != 16)
        return APN_EINVAL;
    if (algo != 0 && algo != 1) return APN_EINVAL;
    int w = waves;
    if (w == 0) {
        w = n <= 128 ? 1 : n <= 256 ? 2 : n <= 512 ? 4 : n <= 4096 ? 8 : 16;
        // stacked index stages (hundreds of clouds in one launch: every CU holds several chains): half the waves per
        // cloud -- fewer waves at each step's rendezvous, 395 instead of 457 ns per step at 640 clouds of 1024 points
        // (202 vs 234 us per launch), and half the wave slots taken from the kernels of the other stream
        if (b >= 256 && w == 8 && n <= 2048) w = 4;
    }
    while (w < 16 && (n + w * 64 - 1) / (w * 64) > 16) w *= 2;
    switch (w) {
    case 1: return dispatch_slots<1>(o, b, m, xyz, temp, idxs, new_xyz, algo, st);
    case 2: return dispatch_slots<2>(o, b, m, xyz, temp, idxs, new_xyz, algo, st);
    case 4: return dispatch_slots<4>(o, b, m, xyz, temp, idxs, new_xyz, algo, st);
    case 8: return dispatch_slots<8>(o, b, m, xyz, temp, idxs, new_xyz, algo, st);
    default: return dispatch_slots<16>(o, b, m, xyz, temp, idxs, new_xyz, algo, st);
    }
}

extern "C" int apn_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                           int *idxs, void *stream) {
    return fps_impl(b, n, m, xyz, temp, idxs, nullptr, 0, fps_default_algo(), stream);
}

// Tuning / diagnostic entry (not part of the reference boundary): the same sampler with the
// number of waves per cloud (0 = heuristic; 1, 2, 4, 8, 16) and the step algorithm (0 = default,
// 1 = first-generation per-wave records) chosen per call.  Results do not depend on either.
extern "C" int apn_furthest_point_sampling_tuned(int b, int n, int m, const float *xyz, float *temp,
                                                 int *idxs, int waves, int algo, void *stream) {
    return fps_impl(b, n, m, xyz, temp, idxs, nullptr, waves, algo, stream);
}

// FPS that also writes the sampled coordinates new_xyz (B,M,3) = xyz[idx]
// (pointnext.py:146-147 in one launch).  n <= 16384 only (the register-resident kernel).
extern "C" int apn_furthest_point_sampling_xyz(int b, int n, int m, const float *xyz, float *temp,
                                               int *idxs, float *new_xyz, void *stream) {
    if (n > 16384 || !new_xyz) return APN_EINVAL;
    static const int env_waves = [] { const char *e = getenv("APN_FPS_WAVES"); return e ? atoi(e) : 0; }();
    return fps_impl(b, n, m, xyz, temp, idxs, new_xyz, env_waves, fps_default_algo(), stream);
}

namespace apn {
__global__ __launch_bounds__(256) void fps_fill_int_kernel(int *__restrict__ p, int v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
template <int W>
static int launch_nested(const FpsOrder &o, int b, int m, const float *xyz, const int *tie_prev, int *idxs, float *new_xyz,
                         int *tie_out, hipStream_t st) {
    const int need = (o.n + W * 64 - 1) / (W * 64);
#define APN_FPSN_CASE(SS)                                                                                              \
    if (need <= SS) {                                                                                                  \
        hipLaunchKernelGGL((fps_nested_kernel<W, SS>), dim3(b), dim3(W * 64), sizeof(float4) * o.n, st, o, m, xyz,     \
                           tie_prev, idxs, new_xyz, tie_out);                                                          \
        APN_LAUNCH_CHECK();                                                                                            \
        return APN_OK;                                                                                                 \
    }
    APN_FPSN_CASE(1)
    APN_FPSN_CASE(2)
    APN_FPSN_CASE(4)
    APN_FPSN_CASE(8)
#undef APN_FPSN_CASE
    return APN_EINVAL;
}
}  // namespace apn

// FPS of an index PYRAMID's level (no reference counterpart: the reference runs the full sampler at every level,
// pointnext.py:146 per block / generator_component4_15.py:406 per stage; the picks are the same, see NEST in the step).
// xyz (B,n,3): level 1: the cloud; deeper: the previous level's sampled coordinates IN PICK ORDER.  tie_prev (B) or null:
// the previous level's record; tie_out (B): this level's (first step whose arg-max was not unique, or INT_MAX).
// Outside the LDS-atomic step's range (n <= 128, n > 4096) the full sampler runs and the record says 0 (never a prefix).
extern "C" int apn_furthest_point_sampling_nested(int b, int n, int m, const float *xyz, const int *tie_prev, int *idxs,
                                                  float *new_xyz, int *tie_out, void *stream) {
    using namespace apn;
    if (b < 0) return APN_EINVAL;
    if (b == 0 || m <= 0) return APN_OK;
    if (n <= 0 || m > n || !xyz || !idxs || !new_xyz || !tie_out) return APN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n <= 128 || n > 4096 || fps_default_algo() != 0) {
        if (int rc = apn_furthest_point_sampling_xyz(b, n, m, xyz, nullptr, idxs, new_xyz, stream)) return rc;
        hipLaunchKernelGGL(fps_fill_int_kernel, dim3((b + 255) / 256), dim3(256), 0, st, tie_out, 0, b);
        APN_LAUNCH_CHECK();
        return APN_OK;
    }
    const FpsOrder o = fps_order(n);
    int w = n <= 256 ? 2 : n <= 512 ? 4 : 8;
    if (b >= 256 && w == 8 && n <= 2048) w = 4;
    switch (w) {
    case 2: return launch_nested<2>(o, b, m, xyz, tie_prev, idxs, new_xyz, tie_out, st);
    case 4: return launch_nested<4>(o, b, m, xyz, tie_prev, idxs, new_xyz, tie_out, st);
    default: return launch_nested<8>(o, b, m, xyz, tie_prev, idxs, new_xyz, tie_out, st);
    }
}

// FPS (+ sampled coordinates) of batch A and, in the same launch, the zero-filling ball query of
// batch B (whose new_xyz_b already exists).  Either half may be absent (xyz_a / xyz_b null).
// Shapes the fused kernel does not cover fall back to the two launches back to back.
extern "C" int apn_sa_sample_overlap(int b, int n, int m, float radius, int nsample,
                                     const float *xyz_a, int *fidx_a, float *new_xyz_a,
                                     const float *xyz_b, const float *new_xyz_b, int *idx_b,
                                     void *stream) {
    using namespace apn;
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return APN_EINVAL;
    if (xyz_a && (!fidx_a || !new_xyz_a)) return APN_EINVAL;
    if (xyz_b && (!new_xyz_b || !idx_b)) return APN_EINVAL;
    const int need = (n + 511) / 512;               // slots per lane with 8 waves
    // (the two-role launch runs the LDS-atomic step: only when that step is asked for, see fps_default_algo)
    const bool fusable = xyz_a && xyz_b && n > 512 && n <= 4096 && need <= 8 && fps_default_algo() == 0;
    if (!fusable) {
        if (xyz_a)
            if (int rc = apn_furthest_point_sampling_xyz(b, n, m, xyz_a, nullptr, fidx_a, new_xyz_a, stream))
                return rc;
        if (xyz_b) return apn_ball_query_zero(b, n, m, radius, nsample, new_xyz_b, xyz_b, idx_b, stream);
        return APN_OK;
    }
    const FpsOrder o = fps_order(n);
    BallArgs q;
    q.b = b; q.n = n; q.m = m; q.nsample = nsample;
    q.q_per_block = 32;
    q.blocks_x = (m + q.q_per_block - 1) / q.q_per_block;
    q.radius2 = radius * radius;                    // ball_query_gpu.cu:29 (float32 product)
    q.new_xyz = new_xyz_b; q.xyz = xyz_b; q.idx = idx_b;
    const size_t dyn_fps = sizeof(float4) * (size_t)n;
    const size_t dyn_bq = sizeof(float) * 3 * (size_t)(n < BQ_CHUNK ? n : BQ_CHUNK) + sizeof(int) * 2 * q.q_per_block;
    const size_t dyn = dyn_fps > dyn_bq ? dyn_fps : dyn_bq;
    const dim3 grid(b + b * q.blocks_x);
    hipStream_t st = (hipStream_t)stream;
#define APN_FB_CASE(SS)                                                                          \
    if (need <= SS) {                                                                            \
        hipLaunchKernelGGL((fps_ball_kernel<SS>), grid, dim3(512), dyn, st, o, b, m, xyz_a, fidx_a, \
                           new_xyz_a, q);                                                        \
        APN_LAUNCH_CHECK();                                                                      \
        return APN_OK;                                                                           \
    }
    APN_FB_CASE(2)
    APN_FB_CASE(4)
    APN_FB_CASE(8)
#undef APN_FB_CASE
    return APN_EINVAL;
}

// Diagnostic only (see fps_stamp_kernel): n must be 1024; dbg receives 6 cycle sums.
extern "C" int apn_fps_debug_stamps(int b, int n, int m, const float *xyz, float *temp, int *idxs,
                                    unsigned long long *dbg, void *stream) {
    using namespace apn;
    if (n != 1024 || b <= 0 || m <= 1 || !dbg) return APN_EINVAL;
    FpsOrder o{1024, 1024, 10, 1, 0};
    hipLaunchKernelGGL(fps_stamp_kernel, dim3(b), dim3(512), 0, (hipStream_t)stream, o, m, xyz, temp,
                       idxs, dbg);
    APN_LAUNCH_CHECK();
    return APN_OK;
}
