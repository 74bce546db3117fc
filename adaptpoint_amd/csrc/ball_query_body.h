// ball_query_body.h -- the radius search of ball_query.hip as a device function, so that it can
// also run as one ROLE of the fused index-stage launch (fps.hip: fps_ball_kernel).
#pragma once
#include "apn_common.h"

namespace apn {

constexpr int BQ_CHUNK = 4096;             // support points staged per LDS pass (48 KiB)
constexpr int BQ_QPW = 4;                  // queries a wave tests each staged point against (8 for full tiles: ball_query.hip)

// One workgroup of WAVES waves: tile `bx` of cloud `cloud`; each wave owns queries
// q0 + wave, q0 + wave + WAVES, ... of the tile.  s_dyn: three coordinate planes of
// min(n, BQ_CHUNK) floats, then 2 * q_per_block ints.
template <int WAVES, int QPW = BQ_QPW>
__device__ __forceinline__ void ball_query_body(
    int n, int m, float radius2, int nsample, int q_per_block, int zero_empty,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int *__restrict__ idx,
    int cloud, int bx, float *s_dyn) {
    // Dynamic LDS: three coordinate planes of `chunk` floats, then per query of
    // the tile its hit count so far and its first hit.
    const int chunk = min(n, BQ_CHUNK);
    float *sx = s_dyn, *sy = s_dyn + chunk, *sz = s_dyn + 2 * chunk;
    int *s_cnt = reinterpret_cast<int *>(s_dyn + 3 * chunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q_begin = bx * q_per_block;
    const int q_end = min(q_begin + q_per_block, m);

    xyz += (size_t)cloud * n * 3;
    new_xyz += (size_t)cloud * m * 3;
    idx += (size_t)cloud * m * nsample;

    int *cnt_of = s_cnt;
    int *first_of = s_cnt + q_per_block;
    for (int i = tid; i < q_per_block; i += WAVES * 64) { cnt_of[i] = 0; first_of[i] = 0; }

    for (int base = 0; base < n; base += BQ_CHUNK) {
        const int len = min(BQ_CHUNK, n - base);
        __syncthreads();  // previous pass done with the planes (and cnt init visible)
        // Coalesced staging: the chunk is 3*len consecutive floats.
        for (int i = tid; i < 3 * len; i += WAVES * 64) {
            const float v = xyz[(size_t)base * 3 + i];
            const int p = i / 3, c = i - p * 3;
            (c == 0 ? sx : c == 1 ? sy : sz)[p] = v;
        }
        __syncthreads();

        // A wave takes QPW consecutive queries per pass over the chunk: a point's coordinates are read from LDS
        // once for all of them (one query per pass spent most of its instructions on the three LDS reads and the
        // loop; the per-query part -- distance, ballot, slot arithmetic -- and the order of the hits are unchanged)
        for (int qb = q_begin + wave * QPW; qb < q_end; qb += WAVES * QPW) {
            int cnt[QPW], first[QPW];
            float qx[QPW], qy[QPW], qz[QPW];
            bool open = false;
#pragma unroll
            for (int j = 0; j < QPW; ++j) {
                const int q = qb + j;
                const bool in = q < q_end;
                cnt[j] = in ? __builtin_amdgcn_readfirstlane(cnt_of[q - q_begin]) : nsample;   // wave-uniform
                first[j] = in ? __builtin_amdgcn_readfirstlane(first_of[q - q_begin]) : 0;
                const int qc = in ? q : q_begin;
                qx[j] = new_xyz[qc * 3 + 0];
                qy[j] = new_xyz[qc * 3 + 1];
                qz[j] = new_xyz[qc * 3 + 2];
                open = open || cnt[j] < nsample;
            }
            if (!open) continue;
            for (int k0 = 0; k0 < len; k0 += 64) {
                const int k = k0 + lane;
                const bool inside = k < len;
                const float px = inside ? sx[k] : 0.0f, py = inside ? sy[k] : 0.0f, pz = inside ? sz[k] : 0.0f;
                bool any_open = false;
#pragma unroll
                for (int j = 0; j < QPW; ++j) {
                    if (cnt[j] >= nsample) continue;              // wave-uniform
                    const bool hit = inside && dist2(qx[j] - px, qy[j] - py, qz[j] - pz) < radius2;
                    const unsigned long long mask = __ballot(hit);
                    if (mask != 0ull) {
                        if (cnt[j] == 0) first[j] = base + k0 + (int)__builtin_ctzll(mask);
                        const int slot = cnt[j] + (int)__builtin_amdgcn_mbcnt_hi(
                                                      (unsigned)(mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (hit && slot < nsample) idx[(size_t)(qb + j) * nsample + slot] = base + k;
                        cnt[j] += (int)__builtin_popcountll(mask);
                    }
                    any_open = any_open || cnt[j] < nsample;
                }
                if (!any_open) break;
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < QPW; ++j)
                    if (qb + j < q_end) { cnt_of[qb + j - q_begin] = cnt[j]; first_of[qb + j - q_begin] = first[j]; }
            }
        }
    }

    __syncthreads();   // the counts were left by the wave that scanned the query's group of QPW, read below per query
    // Tail of each row: slots the scan never reached repeat the first hit
    // (ball_query_gpu.cu:41-45).  Rows of empty balls stay untouched.
    for (int q = q_begin + wave; q < q_end; q += WAVES) {
        const int ql = q - q_begin;
        const int cnt = __builtin_amdgcn_readfirstlane(cnt_of[ql]);
        if ((cnt == 0 && !zero_empty) || cnt >= nsample) continue;
        const int first = __builtin_amdgcn_readfirstlane(first_of[ql]);   // 0 for an empty ball
        int *row = idx + (size_t)q * nsample;
        for (int l = cnt + lane; l < nsample; l += 64) row[l] = first;
    }
}


}  // namespace apn
