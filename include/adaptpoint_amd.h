/*
 * adaptpoint_amd.h -- C ABI of libadaptpoint_amd.so (MI355X / gfx950).
 *
 * The drop-in boundary for the set-abstraction hot path of AdaptPoint /
 * OpenPoints.  Each entry point replaces one function that the reference binds
 * with pybind11 in openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24
 * (module `pointnet2_batch_cuda`), and takes what that function's C++ wrapper
 * hands to its kernel launcher: plain device pointers and sizes, plus the HIP
 * stream to launch on.  No torch types, no allocation, no global state.
 *
 * Contract shared by every entry point (reference: SURVEY.md section 8b):
 *   - all pointers are DEVICE pointers on the current HIP device, float32 /
 *     int32, dense row-major ("contiguous") in the layout given below;
 *   - outputs and scratch are allocated, and where stated pre-initialised, by
 *     the caller; kernels never allocate or free;
 *   - `stream` is a hipStream_t (NULL = the default stream); the call only
 *     enqueues work, it never synchronises the device;
 *   - return value: 0 on success; a positive value is the hipError_t of the
 *     failed launch; APN_EINVAL for an argument the kernels cannot handle.
 *     Nothing ever calls exit() (the reference launchers do, e.g.
 *     sampling_gpu.cu:46-50).
 *   - sizes of zero are no-ops that return 0.
 */
#ifndef ADAPTPOINT_AMD_H
#define ADAPTPOINT_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APN_OK 0
#define APN_EINVAL (-1)

#if defined(__GNUC__)
#define APN_API __attribute__((visibility("default")))
#else
#define APN_API
#endif

/* Library / ABI version: major*10000 + minor*100 + patch. */
APN_API int apn_version(void);

/* Text for a return code of any function below (static storage). */
APN_API const char *apn_error_string(int code);

/* The compiler flag line the library was built with (static storage).  No reference counterpart: the
 * reference's setup.py (openpoints/cpp/pointnet2_batch/setup.py) fixes its nvcc flags; here two flags are
 * part of correctness (-ffp-contract=off pins the float rounding of every distance, -fno-slp-vectorize
 * keeps compiler-made packed-FP32 out), and the loader checks for them. */
APN_API const char *apn_build_flags(void);

/* Replaces furthest_point_sampling_wrapper (pointnet2_api.cpp:18,
 * sampling.cpp:39-48 -> sampling_gpu.cu:101-260).
 *   xyz  (B,N,3) in; temp (B,N) in/out, pre-filled by the caller (1e10,
 *   subsample.py:94), holds the final min squared distances on return;
 *   idxs (B,M) out.  idxs[:,0] = 0; ties follow the reference's block-tree
 *   order (smallest bit-reversed thread id, then lowest index). */
APN_API int apn_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                int *idxs, void *stream);

/* Replaces ball_query_wrapper (pointnet2_api.cpp:11, ball_query.cpp:29-39 ->
 * ball_query_gpu.cu:15-73).
 *   new_xyz (B,M,3) queries; xyz (B,N,3) support; idx (B,M,nsample) out,
 *   pre-zeroed by the caller (group.py:194): rows of empty balls are not
 *   written.  Strict d2 < radius*radius in float32. */
APN_API int apn_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, void *stream);

/* Replaces group_points_wrapper (pointnet2_api.cpp:12, group_points.cpp:25-35 ->
 * group_points_gpu.cu:53-92).  points (B,C,N), idx (B,npoints,nsample) ->
 * out (B,C,npoints,nsample). */
APN_API int apn_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int *idx, float *out, void *stream);

/* Replaces group_points_grad_wrapper (pointnet2_api.cpp:13, group_points.cpp:13-23
 * -> group_points_gpu.cu:14-50).  grad_out (B,C,npoints,nsample), idx ->
 * grad_points (B,C,N) += scatter; the caller zeroes grad_points (group.py:111). */
APN_API int apn_group_points_grad(int b, int c, int n, int npoints, int nsample,
                          const float *grad_out, const int *idx, float *grad_points,
                          void *stream);

/* Replaces gather_points_wrapper (pointnet2_api.cpp:15, sampling.cpp:16-24 ->
 * sampling_gpu.cu:15-51).  points (B,C,N), idx (B,npoints) -> out (B,C,npoints). */
APN_API int apn_gather_points(int b, int c, int n, int npoints, const float *points,
                      const int *idx, float *out, void *stream);

/* Replaces gather_points_grad_wrapper (pointnet2_api.cpp:16, sampling.cpp:27-36 ->
 * sampling_gpu.cu:53-90).  grad_points (B,C,N) += scatter of grad_out (B,C,npoints);
 * caller-zeroed (subsample.py:136). */
APN_API int apn_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int *idx, float *grad_points, void *stream);

/* The training loop's resampler (examples/classification/train_autoaug.py:493-498): points
 * (B,N,C) rows (3 <= C <= 8), fidx (B,p_all) FPS picks, choice (s_cnt) a subset of 0..p_all-1
 * shared by the batch -> pos (B,s_cnt,3) = points[b, fidx[b, choice[s]], :3] and
 * x (B,cx,s_cnt) = the first cx channels, channel-major (what `data['pos']`, `data['x']`
 * become at :500-501), one launch.  choice values must lie in [0, p_all). */
APN_API int apn_resample_points(int b, int n, int c, int p_all, int s_cnt, int cx,
                                const float *points, const int *fidx, const int *choice,
                                float *pos, float *x, void *stream);

/* Replaces three_nn_wrapper (pointnet2_api.cpp:20, interpolate.cpp:20-28 ->
 * interpolate_gpu.cu:16-81).  unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3)
 * SQUARED distances ascending, idx (B,n,3).  m < 3 leaves +inf / index 0 in
 * the unused slots, as the reference does. */
APN_API int apn_three_nn(int b, int n, int m, const float *unknown, const float *known,
                 float *dist2, int *idx, void *stream);

/* Replaces three_interpolate_wrapper (pointnet2_api.cpp:21, interpolate.cpp:31-43 ->
 * interpolate_gpu.cu:84-124).  points (B,C,M), idx/weight (B,N,3) -> out (B,C,N).
 * NOTE the reference's argument order: (b, c, m, n). */
APN_API int apn_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, void *stream);

/* Replaces three_interpolate_grad_wrapper (pointnet2_api.cpp:22,
 * interpolate.cpp:45-57 -> interpolate_gpu.cu:127-168).  grad_out (B,C,N) ->
 * grad_points (B,C,M) += scatter; caller-zeroed (upsampling.py:82).
 * NOTE the reference's argument order: (b, c, n, m). */
APN_API int apn_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int *idx, const float *weight, float *grad_points,
                               void *stream);

/* ------------------------------------------------------------------------
 * Fused set-abstraction block (no single reference entry point: the reference runs
 * this chain as PyTorch ops over materialised (B,C,M,K) tensors --
 * openpoints/models/layers/group.py:235-255,323-335 and
 * openpoints/models/backbone/pointnext.py:146-168).  Shapes supported by this
 * build: c_in = 32 features (+3 relative xyz), c_mid = 32, c_out = 64,
 * nsample = 32; anything else returns APN_EINVAL and callers use the unfused ops.
 *   xyz (B,N,3) f32, new_xyz (B,M,3) f32, ft (B,N,32) bf16 point-major copy of the
 *   features, idx (B,M,32) i32, w1 (32,35) f32 with columns [dp(3), f(32)],
 *   w2 (64,32) f32.  Per-channel sums cross workgroups WITHOUT fold launches: BatchNorm-1's statistics
 *   leave apn_sa_prep_stats as a few dozen partial rows that every workgroup of the next launch sums itself;
 *   every other per-channel sum goes through an ACCUMULATOR SET -- apn_sa_acc_words(ncol) unsigned 64-bit
 *   words, zeroed by an earlier launch, added to with integer atomics on a two-limb fixed-point value
 *   (csrc/apn_common.h: order-independent, the same bits every run) -- which the consumer kernel reads in
 *   its prologue; with reduced (e.g. all-reduced) float64 sums `sums*` the consumers take those instead.
 *   Weight gradients leave their producers as partial rows summed in float64 by apn_sa_bwd_finalize; G and
 *   gip are float atomic adds into zeroed buffers.
 *   "pack" = {scale, shift, mean, invstd}[C] of a BatchNorm folded to y*scale+shift.
 * Everything below only enqueues kernels (graph-capturable).
 * ------------------------------------------------------------------------ */

/* FPS that also writes new_xyz (B,M,3) = xyz[idx] (pointnext.py:146-147); n <= 16384.
 * temp == NULL: start from 1e10 everywhere (what callers fill temp with, subsample.py:94)
 * and leave no min-distances behind. */
APN_API int apn_furthest_point_sampling_xyz(int b, int n, int m, const float *xyz, float *temp,
                                            int *idxs, float *new_xyz, void *stream);

/* apn_ball_query that writes zeros into the rows of empty balls (no pre-zeroed idx needed). */
APN_API int apn_ball_query_zero(int b, int n, int m, float radius, int nsample,
                                const float *new_xyz, const float *xyz, int *idx, void *stream);

/* Workgroups the fused passes launch for B clouds x M queries (without / with a tile map). */
APN_API int apn_sa_grid_blocks(int b, int m);
APN_API int apn_sa_grid_rows(int b, int m, int with_tile_map);
/* workgroups (= partial rows partW2) of the backward pass */
APN_API int apn_sa_bwd_main_rows(int b, int m);
/* unsigned 64-bit words of an accumulator set of ncol columns */
APN_API int apn_sa_acc_words(int ncol);
/* 16-byte granular zero fill by a kernel (graph-capturable) */
APN_API int apn_zero_fill(void *base, long long bytes, void *stream);
/* Diagnostic: stamps[slot] = the device's 100 MHz wall clock when the launch runs (one thread; graph-capturable:
 * phase boundaries of a replayed step on every branch of the graph).  stamps: unsigned 64-bit words. */
APN_API int apn_debug_stamp(void *stamps, int slot, void *stream);
/* Diagnostic: `blocks` workgroups of 256 lanes hold 24 register values each for `turns` barrier-and-LDS-atomic turns and
 * check them: bad[i] (i < 24) = lanes whose i-th value changed, bad[24 .. 31] samples (index << 32 | value found).
 * bad: 32 unsigned 64-bit words, zeroed by the caller.  (Does register state survive beside other kernels?) */
APN_API int apn_debug_vgpr_hold(int blocks, int turns, unsigned long long *bad, void *stream);
/* Diagnostic: a packed-FP32 instruction with operand selection, checked result by result beside whatever else runs.
 * form 0: v_pk_add_f32 ... op_sel_hi:[1,0] (a pair's LOW register feeds both lanes); form 1: ... op_sel:[0,1] (its HIGH
 * register feeds both lanes: the form blamed in profiles/r04_packed_fp32_op_sel.md); form 2: form 1 behind `s_nop 7`.
 * bad[0] results of the lane that crosses halves wrong, bad[1] of those computed with the pair's OTHER register, bad[2]
 * other results wrong, bad[3] results checked, bad[7] samples taken, bad[8 + 3k ..] sample k (k < 8): {turn << 32 | block
 * << 8 | lane}, {found << 32 | wanted}, {operand << 32 | the other lane's result}.  bad: 32 unsigned 64-bit words,
 * zeroed by the caller. */
APN_API int apn_debug_vpk_probe(int blocks, int turns, int form, unsigned long long *bad, void *stream);

/* `precision` (every function that takes ft): 1 = operands rounded to bf16; 2 = operands split
 * into hi + lo bf16 parts, each product three MFMAs (hi*hi + hi*lo + lo*hi, "bf16x3"): fp32-grade
 * results (~1e-5) at 3x the (small) MFMA cost.  ft then holds `precision` tables of (B,N,32)
 * bf16 back to back: [hi] or [hi][lo]. */

/* Index stage, last piece (csrc/sa_geo.hip): occurrence statistics of the neighbourhoods.  For every support
 * point n: geo[b][n] = {occ, Dx, Dy, Dz} as four int64 -- the number of positions (query, slot) with
 * idx == n and the sum over them of d = (xyz[n] - new_xyz[query]) / radius (group.py:250-253) in units of
 * 2^-36 (integer sums: reproducible); dd[b][apn_sa_geo_dd_doubles(n)] (float64) = the cloud's shares of
 * sum d d^T over its positions, six values {xx, xy, xz, yy, yz, zz} per slab of 2048 points.  One launch, LDS
 * accumulation, nothing to clear; nsample must be 32. */
APN_API int apn_sa_geo_dd_doubles(int n);
APN_API int apn_sa_point_geo(int b, int n, int m, int nsample, float radius, const float *xyz,
                             const float *new_xyz, const int *idx, void *geo, void *dd, void *stream);

/* Forward launch 1 of 3: f (B,32,N) f32 -> ft, and (stats != 0) BatchNorm-1's batch statistics WITHOUT a pass
 * over the positions: part1[apn_sa_prep_rows(b, n)][64] = {sum, sumsq}[32] of y1 = conv1(cat(dp, f[idx])) from geo / dd
 * (y1 is affine in per-point and per-position terms; operands as the MFMA sees them).  Also clears `zero_words`
 * 64-bit words at `zero` (the accumulator set of the next launch). */
APN_API int apn_sa_prep_rows(int b, int n);
APN_API int apn_sa_prep_stats(int b, int n, const float *f, const void *geo, const void *dd, const float *w1,
                              int precision, int stats, void *ft, float *part1, void *zero, long long zero_words,
                              void *stream);

/* out[0..ncol) = float64 column sums of part[rows][ncol] (ncol <= 128, a power of two) /
 * the totals of an accumulator set; out[ncol] = count (this rank's positions), out[ncol + 1] = 1.
 * The SyncBatchNorm path reduces, all-reduces the ncol + 2 values over ranks (sums, GLOBAL count, world size),
 * then calls the consumer with `sums`: it takes the count from the reduced vector (its `count` argument is
 * ignored) and dL/dgamma, dL/dbeta are reported as global sum / world (what SyncBatchNorm +
 * DistributedDataParallel leave in .grad). */
APN_API int apn_sa_reduce_rows(const float *part, int rows, int ncol, double count, double *out,
                               void *stream);
APN_API int apn_sa_reduce_acc(const void *acc, int ncol, double count, double *out, void *stream);

/* BatchNorm fold as a launch of its own (the width-generic family, csrc/sa_wide*.hip): {sum, sumsq}[C] over
 * `count` positions (part[rows][2C], or sums[2C] when part == NULL) -> pack[4][C]; updates the running
 * buffers / num_batches_tracked when training (torch.nn.BatchNorm semantics); uses the running buffers when
 * not training.  Rider: sgn_out[i] = sign(sgn_gamma[i]), i < sgn_c.  C a multiple of 4, <= 1024. */
APN_API int apn_sa_bn_fold(const float *part, int rows, const double *sums, int c, double count,
                           const float *gamma, const float *beta, float eps, float momentum,
                           float *running_mean, float *running_var, void *num_batches_tracked,
                           int training, float *pack, const float *sgn_gamma, int sgn_c,
                           float *sgn_out, void *stream);

/* Forward launch 2 of 3.  Prologue: BatchNorm-1 (g1, b1, running buffers, eps, momentum, training flag; `count`
 * positions) folded from part1[rows1][64] -- or from sums1 -- by every workgroup (workgroup 0 writes pack1
 * [4][32] and updates the running buffers).  Then a1 = relu(bn1(y1)); y2 = conv2(a1);
 * ysel/ksel (B,M,64): per (query, channel) the extreme of y2 over the K neighbours (max where gamma2 >= 0,
 * min otherwise) and the neighbour slot holding it; acc2 (accumulator set, 128 columns, zeroed by
 * apn_sa_prep_stats) += {sum[64], sumsq[64]} of y2. */
APN_API int apn_sa_fwd_main(int b, int n, int m, int precision, float radius, const float *xyz,
                            const float *new_xyz, const void *ft, const int *idx, const int *tmap, const float *w1,
                            const float *w2, const float *g1, const float *b1, float *rm1, float *rv1, void *nbt1,
                            float eps1, float mom1, int train1, double count, const float *part1, int rows1,
                            const double *sums1, float *pack1, const float *gamma2, float *ysel, void *ksel,
                            void *acc2, void *stream);

/* Forward launch 3 of 3.  Prologue: BatchNorm-2 folded from acc2 (or sums2) by every workgroup (the first
 * writes pack2 [4][64] and updates the running buffers).
 * out (B,64,M) = act(ysel*scale2 + shift2 + Ws f[:, fidx] + bs); ws/bs/ft/fidx may be null
 * (no skip branch), relu = 0/1.  f is read from the point-major table(s) ft; fidx (B,M), ws (64,32), bs (64).
 * (zero_base, zero_floats: optional region -- the backward's atomically accumulated gip | accS | accT --
 * cleared by this launch, the forward's last, so that apn_sa_backward_seq(zero_bytes = 0) needs no fill launch) */
APN_API int apn_sa_fwd_out(int b, int n, int m, const float *ysel, const void *acc2, const double *sums2,
                           const float *g2, const float *b2, float *rm2, float *rv2, void *nbt2, float eps2,
                           float mom2, int train2, double count, float *pack2, const void *ft, int precision,
                           const int *fidx, const float *ws, const float *bs, int relu, float *out,
                           float *zero_base, long long zero_floats, void *stream);

/* Backward launch 1 of 4: g = g_out * [out > 0] (relu) ; goa (B,M,64) = g * scale2;
 * g_out (B,64,M) is read with element strides (gs_b, gs_c, gs_m) -- a broadcast upstream
 * gradient (stride 0, e.g. from loss = out.sum()) needs no materialised copy;
 * accS (accumulator set, 128 columns, zeroed) += {S1 = sum g, S2 = sum g*yhat_sel}[64]; with the skip branch
 * partWs[apn_sa_bwd_prep_rows(b, m)][64*32] = dL/dWs per block of 64 queries and gip (B,N,32) += Ws^T g at the
 * sampled points (zeroed): float atomic adds, or -- dup (int32[b], may be NULL; the tail of the row map's blob,
 * apn_sa_rowmap_many) says 0 for a cloud, i.e. its picks are m different points -- plain stores. */
APN_API int apn_sa_bwd_prep_rows(int b, int m);
APN_API int apn_sa_bwd_prep(int b, int n, int m, const float *g_out, long long gs_b,
                            long long gs_c, long long gs_m, const float *out, int relu,
                            const float *ysel, const float *pack2, const void *ft, int precision,
                            const int *fidx, const float *ws, float *goa, void *accS,
                            float *partWs, float *gip, const int *dup, void *stream);

/* Backward launch 2 of 4: the pass over the positions.  Prologue (every workgroup): the constants of
 * dL/dy2 = goa*[pos==ksel] + y2*D2 + E2 from accS (or sumsS) and pack2, Qm = W2^T diag(D2) W2, evec = E2 W2.
 * accT (accumulator set, 64 columns, zeroed) += {sum g_u, sum g_u*yhat1}[32] (g_u = dL/da1 * [a1 > 0]);
 * partW2[apn_sa_bwd_main_rows(b, m)][64*32] = the workgroup's share of dL/dW2 (sparse arg-max part +
 * D2 (W2 Gram) + E2 (x) sum a1);  GU[place][32] = g_u of every live tile-map row, stored at the row's place in the
 * point-sorted order (rowdst: apn_sa_rowmap_many; 32 b m rows of 32 floats, nothing to clear) -- round 5: NO float
 * atomics; a point's rows are consecutive and apn_sa_bwd_point_grads sums them in ascending row order, so every
 * gradient of the chain is bit-reproducible (rounds 1-4: A (B,N,32) += g_u by memory-side float atomics, or 64-bit
 * fixed-point integer atomics in a separate "deterministic" mode).  tmap and rowdst are required: the pass runs over the
 * tile map.  HA, HB (B,M,32) = g_u and yhat1 summed per query.  pack1 = BN1's [4][32] of the forward. */
APN_API int apn_sa_bwd_main(int b, int n, int m, int precision, float radius, const float *xyz,
                            const float *new_xyz, const void *ft, const int *idx, const int *tmap, const float *w1,
                            const float *w2, const float *pack1, const float *pack2, const void *accS,
                            const double *sumsS, double count, int train2, const float *goa, const void *ksel,
                            void *accT, float *partW2, const int *rowdst, float *GU, float *HA, float *HB,
                            void *stream);

/* Backward launch 3 of 4.  Prologue: the batch constants of dL/dy1 = g_u*ca + yhat1*cb + cc from accT (or sumsT).
 * dL/dy1 summed per source point (G) and per query (H), formed from the point's rows of GU (pcnt_poff: int32[2 b n],
 * how many rows gather a point and the place of the first -- apn_sa_rowmap_many), geo (apn_sa_point_geo), HA, HB, and
 * everything linear in them, one workgroup per 64-point tile: g_f (B,32,N) = G W1[:,3:] (+ gip); optional
 * g_p (B,N,3) += G W1[:,:3]/r and g_newp (B,M,3) = -H W1[:,:3]/r; partW[apn_sa_bwd_weight_rows(b, n)][32*38] =
 * per-block products for dL/dW1 (sa_glue.hip). */
APN_API int apn_sa_bwd_weight_rows(int b, int n);
APN_API int apn_sa_bwd_point_grads(int b, int n, int m, const float *GU, const int *pcnt_poff, const void *geo,
                                   const float *HA, const float *HB, const void *accT, const double *sumsT,
                                   double count, int train1, const float *pack1, const void *ft, int precision,
                                   const float *xyz, const float *new_xyz, const float *w1,
                                   const float *gip, float radius, float *partW, float *g_f,
                                   float *g_p, float *g_newp, void *stream);

/* Backward launch 4 of 4: column sums in float64 -> g_w1 (32,35) from partW, g_w2 (64,32) from partW2, optional
 * g_ws (64,32) from partWs; g_bs [64] = S1 of this rank (accS); g_b2 / g_g2 = S1 / S2 and g_b1 / g_g1 = T1 / T2
 * from the accumulator sets or, with reduced sums, global / world.  Any gradient pointer may be null. */
APN_API int apn_sa_bwd_finalize(const float *partW, int rows_w, float radius, float *g_w1,
                                const float *partWs, int rows_s, float *g_ws, const float *partW2, int rows_2,
                                float *g_w2, const void *accS, const double *sumsS,
                                const void *accT, const double *sumsT, float *g_bs, float *g_g2, float *g_b2,
                                float *g_g1, float *g_b1, void *stream);

/* Diagnostics: attach (NULL: detach) a buffer of 8 x 4 x workgroups 64-bit wall-clock stamps that the next
 * launches of the two tile passes fill per wave (scripts/stamp_passes.py).  Not for concurrent use. */
APN_API int apn_sa_debug_stamps(void *buf);

/* Whole-direction launch sequences (csrc/sa_seq.hip): the same kernels as above, enqueued
 * back-to-back by ONE call so that an eager step stays GPU-bound.  `phases` (bit mask
 * 1|2|4) selects the part to enqueue, so a caller can all-reduce the BatchNorm sums between
 * phases (SyncBatchNorm): forward 1 = prep + BatchNorm-1 sums, 2 = main pass, 4 = output; backward
 * 1 = (zero +) entry, 2 = main pass, 4 = point gradients + finalize.
 * sums* (float64, reduced over ranks) replace the partial rows / accumulator sets when non-NULL. */
APN_API int apn_sa_forward_seq(
    int phases, int precision, int b, int n, int m, float radius, const float *xyz, const float *new_xyz,
    const float *f, const int *idx, const int *tmap, const int *fidx, const void *geo, const void *dd,
    const float *w1, const float *w2, const float *ws, const float *bs,
    const float *g1, const float *b1, float *rm1, float *rv1, void *nbt1, float eps1, float mom1,
    int train1,
    const float *g2, const float *b2, float *rm2, float *rv2, void *nbt2, float eps2, float mom2,
    int train2,
    double count, int relu, void *ft, float *part1, const double *sums1, const double *sums2,
    float *pack1, float *pack2, void *acc2, float *ysel, void *ksel,
    float *out, float *zero_base, long long zero_floats, void *stream);
APN_API int apn_sa_backward_seq(
    int phases, int precision, int b, int n, int m, float radius, const float *xyz, const float *new_xyz,
    const int *idx, const int *tmap, const int *fidx, const void *geo, const float *w1, const float *w2,
    const float *ws, const void *ft, const float *pack1, const float *pack2, const float *ysel,
    const void *ksel, const float *out, int relu, int train1, int train2, double count,
    const float *g_out, long long gs_b, long long gs_c, long long gs_m,
    void *zero_base, long long zero_bytes, float *gip, void *accS, void *accT,
    const int *pcnt_poff, const int *rowdst, float *GU,   /* the row map (apn_sa_rowmap_many) and the rows' scratch */
    float *goa, float *partWs, float *partW2, float *partW, const double *sumsS, const double *sumsT,
    float *HA, float *HB,
    float *g_f, float *g_p, float *g_newp, float *g_w1, float *g_w2, float *g_g1, float *g_b1, float *g_g2,
    float *g_b2, float *g_ws, float *g_bs, void *stream);
/* Index stages of consecutive batches, overlapped: FPS (+ sampled coordinates) of batch A and,
 * in the SAME launch, the zero-filling ball query of batch B, whose new_xyz_b an earlier call
 * produced (csrc/fps.hip: fps_ball_kernel; 512 < n <= 4096, else the two launches back to
 * back).  Either half may be absent: xyz_a == NULL or xyz_b == NULL. */
APN_API int apn_sa_sample_overlap(int b, int n, int m, float radius, int nsample,
                                  const float *xyz_a, int *fidx_a, float *new_xyz_a,
                                  const float *xyz_b, const float *new_xyz_b, int *idx_b,
                                  void *stream);
/* FPS of one level of an index PYRAMID (no reference counterpart: the reference runs the full sampler at every level --
 * pointnext.py:146 per block, models_adaptpoint/generator_component4_15.py:406 per stage -- with the same result).
 * FPS is progressive: run on a sample's picks in pick order it returns picks 0, 1, 2, ... again whenever every arg-max
 * was unique.  xyz (B,n,3): the cloud (level 1) or the previous level's sampled coordinates in pick order; tie_prev (B)
 * or null: the previous level's record; tie_out (B): the first step whose arg-max was not unique (INT_MAX: none; 0: not
 * recorded).  A cloud with tie_prev[cloud] >= m gets idxs = 0 .. m-1 and a copy of its first m rows; any other cloud
 * the full sampler.  idxs (B,m) int32, new_xyz (B,m,3). */
APN_API int apn_furthest_point_sampling_nested(int b, int n, int m, const float *xyz, const int *tie_prev, int *idxs,
                                               float *new_xyz, int *tie_out, void *stream);
/* apn_sa_sample_seq (below) with the nested sampler: one level of an index pyramid. */
APN_API int apn_sa_sample_seq_nested(int b, int n, int m, float radius, int nsample, const float *xyz,
                                     const int *tie_prev, int *tie_out, int *fidx, float *new_xyz, int *idx, void *geo,
                                     void *dd, void *stream);
/* Index stage: temp := 1e10, FPS (+ sampled coordinates), ball query (zero-filling) and, with geo != NULL,
 * apn_sa_point_geo.  temp may be NULL (no min-distances kept).  n <= 16384 (the register-resident sampler; larger
 * clouds: apn_furthest_point_sampling + apn_ball_query). */
APN_API int apn_sa_sample_seq(int b, int n, int m, float radius, int nsample, const float *xyz,
                              float *temp, int *fidx, float *new_xyz, int *idx, void *geo, void *dd,
                              void *stream);

/* ------------------------------------------------------------------------
 * the grouping stage of the imitator's PointsetGrouper
 * (openpoints/models_adaptpoint/generator_component4_15.py:394-431, normalize = "anchor"):
 *   out[b][ch][q] = max_k ( alpha[ch] * (points[b][idx[b][q][k]][ch] - points[b][fidx[b][q]][ch]) + beta[ch] )
 * without the (B, np, K, C) intermediates.  points (B,N,C) f32 point-major as in the reference,
 * idx (B,M,K) from apn_ball_query, fidx (B,M) from apn_furthest_point_sampling, alpha/beta [C];
 * out (B,C,M); ksel (B,M,C) uint8 = position of the (first) maximum, kept for the backward.
 * C a power of two in 4..1024, K <= 255.
 * Backward: g_points (B,N,C) = alpha*g at the selected neighbour - alpha*g at the anchor, summed over the
 * queries: caller-zeroed; where a cloud's slice [N][8..32 channels] fits in LDS (N <= 4096) it is accumulated
 * there (LDS atomics) and stored whole, otherwise added with global float atomics;
 * part[apn_pointset_group_rows(b, m, c)][2C] = partial rows of {sum g*(x_sel - anchor), sum g} = dL/dalpha,
 * dL/dbeta (summed by the caller).
 * ------------------------------------------------------------------------ */
APN_API int apn_pointset_group_rows(int b, int m, int c);
APN_API int apn_pointset_group_max(int b, int n, int m, int c, int k, const float *points,
                                   const int *idx, const int *fidx, const float *alpha,
                                   const float *beta, float *out, void *ksel, void *stream);
APN_API int apn_pointset_group_max_grad(int b, int n, int m, int c, int k, const float *points,
                                        const int *idx, const int *fidx, const float *alpha,
                                        const void *ksel, const float *g_out, float *g_points,
                                        float *part, void *stream);

/* ------------------------------------------------------------------------
 * multi-head self-attention over the M points of a cloud with
 * head_dim 16 (the imitator's Anchor_selfattention,
 * openpoints/models_adaptpoint/generator_component4_15.py:467-474) without the (B,H,M,M)
 * score tensor:  out = softmax(q k^T / 4) v  per head.  q, k, v, out: (B, M, heads*16) f32 (the
 * layout the reference holds them in before its reshape/permute); M % 32 == 0.
 * apn_attention_prep writes the bf16 hi|lo operand images into `images`: 3 (forward only) or 6
 * (for_backward != 0) blocks of b*heads*m*32 bf16 = 64 bytes per point and head each.
 * apn_attention_fwd also returns lse (B,heads,M), the log2-domain log-sum-exp of every query.
 * apn_attention_bwd: g_out = dL/d out -> dq, dk, dv (B,M,heads*16); scratch = 2 * b*heads*m*32
 * bf16 + b*heads*m floats.
 * ------------------------------------------------------------------------ */
/* The same attention for few points (0 < m <= apn_attention_small_max() = 32; the imitator's 4-anchor head): one wave
 * per (cloud, head) in float32; the backward recomputes the probabilities from q, k, v. */
APN_API int apn_attention_small_max(void);
APN_API int apn_attention_small_fwd(int b, int m, int heads, const float *q, const float *k, const float *v,
                                    float *out, void *stream);
APN_API int apn_attention_small_bwd(int b, int m, int heads, const float *q, const float *k, const float *v,
                                    const float *g_out, float *dq, float *dk, float *dv, void *stream);
APN_API int apn_attention_prep(int b, int m, int heads, const float *q, const float *k,
                               const float *v, void *images, int for_backward, void *stream);
APN_API int apn_attention_fwd(int b, int m, int heads, const void *images, float *out, float *lse,
                              void *stream);
APN_API int apn_attention_bwd(int b, int m, int heads, const void *images, const float *out,
                              const float *lse, const float *g_out, void *scratch, float *dq,
                              float *dk, float *dv, void *stream);

/* ---- width-generic fused grouped MLP (csrc/sa_wide.hip): C_mid = H in {32,64,128,256}, C_out = 2H,
 * K = 32.  conv1 is hoisted to the points by the caller: U (B,N,H) = W1f f + W1p p / r per point,
 * V (B,M,H) = W1p new_p / r per query, so that y1[q,k] = U[idx[q,k]] - V[q]
 * (openpoints/models/backbone/pointnext.py:157-166 with group.py:248-254).  pack1 = BatchNorm-1's
 * {scale, shift, mean, invstd}[H].  Weight operands are "B images" (bf16 hi/lo parts in MFMA
 * fragment order, adaptpoint_amd/fused_wide.py::mfma_b_image). ---- */
APN_API int apn_sa_wide_grid(int b, int m);          /* workgroups = partial rows of the three passes */
/* Tile map (distinct-hit packing; part of the index stage: it depends on idx only).  A ball query lists a
 * query's cnt distinct hits and fills the other 32 - cnt slots with copies of slot 0
 * (ball_query_gpu.cu:41-48), and every copy computes the same y1, a1, y2: the passes below run over ROWS
 * = (query, distinct slot, multiplicity), whole queries packed in order into 32-row MFMA tiles.
 * tmap: int32[apn_sa_wide_tilemap_ints(b, m)], device memory:
 *   [0] tiles in use; [4, 4 + b m) first query of each tile; [4 + roundup4(b m), ... + 32 b m) the rows,
 *   qlocal | slot << 8 | mult << 16 | (row 0: queries in the tile) << 24, mult = 0 for padding; then
 *   32 b m ints: every row's neighbour idx[query][slot]; scratch.
 * mode 1: fold the copies of slot 0 of every index row that has the ball-query structure (checked row
 * by row; any other row is kept whole); mode 0: one tile per query, 32 rows of multiplicity 1. */
APN_API int apn_sa_wide_tilemap_ints(int b, int m);
APN_API int apn_sa_wide_tilemap(int b, int m, int mode, const int *idx, int *tmap, void *stream);
/* count tile maps in one pair of launches (the batches of a stacked index stage): map z reads idx + z * b * m * 32
 * and fills tmap + z * apn_sa_wide_tilemap_ints(b, m). */
APN_API int apn_sa_wide_tilemap_many(int count, int b, int m, int mode, const int *idx, int *tmap, void *stream);
/* The ROW MAP of `count` stacked tile maps (round 5; index-stage data, a pure function of the neighbour indices): per batch z
 *   pcnt_poff + z * apn_sa_rowmap_ints(b, n)  int32[2 b n + b]: per support point, the number of tile-map rows that gather it
 *                          and the place of the first of them in the point-sorted order (a cloud's places lie in its own
 *                          range of row ids); then per cloud dup = 0 iff its m picks fidx (z-th block of b m ints; may be
 *                          NULL: dup = 1) are m different points -- apn_sa_bwd_prep then STORES the skip branch's gradient
 *                          rows instead of adding them with float atomics;
 *   rowdst + z * 32 b m    int32[32 b m]: the place of every live row of map z = tmap + z * apn_sa_wide_tilemap_ints(b, m).
 * A point's rows occupy consecutive places in ascending row order; GU needs apn_sa_rowmap_places(b, n, m) rows of 32 floats.
 * Replaces the scatter-add of the reference's grouping backward (group_points_grad_kernel_fast,
 * openpoints/cpp/pointnet2_batch/src/group_points_gpu.cu:14-46, atomicAdd per element) inside the fused chain by a store +
 * an ordered sum.  scratch: int32[32 b m] (multi-launch path only). */
APN_API int apn_sa_rowmap_places(int b, int n, int m);
APN_API int apn_sa_rowmap_ints(int b, int n);
APN_API int apn_sa_rowmap_many(int count, int b, int n, int m, const int *tmap, const int *fidx, int *pcnt_poff, int *rowdst, int *scratch,
                               void *stream);
/* out[ncol] (float64) = column sums of part[rows][ncol] (float32) in a fixed order; two passes
 * through scratch[apn_sa_wide_colsum_chunks(rows, ncol)][ncol] (float64) when there is more than one chunk */
APN_API int apn_sa_wide_colsum_chunks(int rows, int ncol);
APN_API int apn_sa_wide_colsum(const float *part, int rows, int ncol, double *scratch, double *out,
                               void *stream);
/* part[grid][2H] = {sum, sumsq} of y1 */
APN_API int apn_sa_wide_stats1(int b, int n, int m, int c_mid, const float *U, const float *V,
                               const int *idx, const int *tmap, float *part, void *stream);
/* a1 = relu(scale1 y1 + shift1), y2 = a1 W2^T (w2_image: B image of W2^T, H x O, min(4, O/32) column
 * tiles per block) -> ysel/ksel (B,M,O): per (query, channel) the extreme of y2 over the K slots
 * (max where sgn2 = +1, min where -1; first slot among equals) and that slot;
 * part[grid][2*O] = {sum, sumsq} of y2 */
APN_API int apn_sa_wide_fwd_main(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                 const int *idx, const int *tmap, const void *w2_image, const float *pack1,
                                 const float *sgn2, float *ysel, void *ksel, float *part, void *stream);
/* dL/da1 = S W2 + a1 Qm + evec (z_image: B image of [W2 ; Qm], (O+H) x H, min(4, H/32) column tiles
 * per block; S[pos,c] = goa[q,c] [ksel[q,c] == pos]), g_u = dL/da1 [a1 > 0]:
 * GU (32 b m, H): row tile * 32 + r = g_u summed over the positions the tile map's row stands for (rows in
 * use only; no atomics -- apn_sa_wide_point_grads sums them per point through the inverse map),
 * HA (B,M,H) = sum_k g_u, HB (B,M,H) = sum_k yhat1, part[grid][2H] = {sum g_u, sum g_u yhat1}.
 * evec == NULL: BatchNorm-2 on running statistics (D2 = E2 = 0): the a1 Qm part of the chain is skipped. */
APN_API int apn_sa_wide_bwd_main(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                                 const int *idx, const int *tmap, const void *z_image, const float *pack1,
                                 const float *evec, const float *goa, const void *ksel, float *GU,
                                 float *HA, float *HB, float *part, float *r_part, void *stream);
/* 1 when apn_sa_wide_bwd_main also leaves the weight-gradient products, as r_part[apn_sa_wide_grid(b, m)]
 * [(O+H) H + H] = the workgroups' shares of {[S^T ; a1^T] a1, sum a1} (c_mid = 32); 0: r_part is ignored and
 * apn_sa_wide_wgrad computes them */
APN_API int apn_sa_wide_wgrad_fused(int c_mid);
/* The small kernels around those passes (csrc/sa_wide_glue.hip).
 * image: B image of Bm (kd x nc): rows < k0 from src0 (row-major kd x nc, or nc x k0 read transposed when
 *   trans0), the rest from src1 ((kd - k0) x nc); ct column tiles per block.
 * bwd_prep: g (B,O,M; element strides gs_*), zeroed where out_act (B,O,M; NULL: no mask) is not positive
 *   -> goa (B,M,O) = g scale2, gpre (B,M,O; NULL: not wanted) = g, and
 *   part_s[apn_sa_wide_bwd_prep_rows(b, m)][2*O] = {sum g, sum g yhat_sel}.
 * consts2 / consts1: the BatchNorm backward constants from those rows (or from `sums`: float64
 *   {S[2C], global count, world} all-reduced over ranks): d2e2 = {D2, E2}[O]; cabc = {ca, cb, cc}[H];
 *   dgamma, dbeta (global / world with `sums`).
 * point_terms: G (B,N,H) = dL/dU = ca (GU rows of the point, summed through the inverse map of apn_sa_wide_csr)
 *   + cb inv1 (occ (U - mean1) - SP . W1p / r) + cc occ, with geo = {occ, SP} of apn_sa_wide_csr;
 *   HA <- -dL/dV = ca HA + cb HB + 32 cc (B,M,H); w1 (H x ldw) with W1p in its first three columns. */
APN_API int apn_sa_wide_image(const float *src0, int k0, int trans0, const float *src1, int kd, int nc, int ct,
                              void *image, void *stream);
APN_API int apn_sa_wide_bwd_prep_rows(int b, int m);
APN_API int apn_sa_wide_bwd_prep(int b, int m, int c_out, const float *g, long long gs_b, long long gs_c,
                                 long long gs_m, const float *ysel, const float *pack2, const float *out_act,
                                 float *gpre, float *goa, float *part_s, void *stream);
APN_API int apn_sa_wide_consts2(const float *part_s, int rows, const double *sums, int c_out,
                                const float *pack2, double count, int training, float *d2e2,
                                float *g_gamma2, float *g_beta2, void *stream);
APN_API int apn_sa_wide_consts1(const float *part_t, int rows, const double *sums, int c_mid,
                                const float *pack1, double count, int training, float *cabc,
                                float *g_gamma1, float *g_beta1, void *stream);
APN_API int apn_sa_wide_point_terms(int b, int n, int m, int c_mid, const float *cabc, const float *pack1,
                                    const float *U, const float *geo, const float *w1, int ldw, float radius,
                                    const float *GU, const int *pcnt_poff, const int *plist, float *G, float *HA,
                                    const float *HB, void *stream);
/* r_part[splits][(O+H) H + H] = partial {[S^T ; a1^T] a1 (rows < O: the sparse part of dL/dW2; the rest:
 * the Gram matrix of a1), sum of a1}; the caller sums the splits (apn_sa_wide_colsum) */
APN_API int apn_sa_wide_wgrad_splits(int b, int m, int c_mid);
APN_API int apn_sa_wide_wgrad(int b, int n, int m, int c_mid, int c_out, const float *U, const float *V,
                              const int *idx, const int *tmap, const float *pack1, const float *goa,
                              const void *ksel, int splits, float *r_part, void *stream);

/* Inverse map of the tile map (index stage): pcnt_poff int32[2 b n] = for every support point the number of
 * rows that gather it and where its list starts in plist int32[32 b m] (row ids tile * 32 + r, ascending);
 * geo float[4 b n] (may be NULL) = {occurrences, sum of the gathering queries' coordinates} per point; with fidx (b,m) = the
 * point every query is (FPS picks) also fq int32[b n] = the query a point is, or -1 (both may be NULL). */
APN_API int apn_sa_wide_csr(int b, int n, int m, const int *idx, const float *new_xyz, const int *tmap,
                            int *pcnt_poff, int *plist, float *geo, const int *fidx, int *fq, void *stream);
/* The dense kernels of the path (csrc/sa_wide_dense.hip), one launch each:
 * fwd_prep: U (B,N,H) = W1f f + W1p p / r, V (B,M,H) = W1p new_p / r (w1: H x (C+3), coordinates first, as the
 *   reference's cat([dp, fj])), and the B image of W2^T (w2: O x H); with fq (B,N; apn_sa_wide_csr) also
 *   fs (B,M,C) = the sampled points' own features f[:, :, fidx] as query-major rows (fs NULL: not wanted);
 *   with geo (B,N,4; apn_sa_wide_csr) also part1[apn_sa_wide_fwd_prep_rows(b, n, m)][2H] = partial {sum, sumsq}
 *   of y1 over ALL positions -- BatchNorm-1's batch statistics from per-point and per-query terms alone
 *   (sum y1 = sum_n occ U - 32 sum_q V, ...), in place of apn_sa_wide_stats1's pass (part1 NULL: not wanted).
 * out: out (B,O,M) = act(ysel (B,M,O) scale2 + shift2 + skip) (pack2 = {scale, shift, mean, invstd}[O]);
 *   skip = ws (O x C) fs + bs, the residual branch on the sampled points' features fs (B,M,C) of fwd_prep
 *   (ws NULL: none; bs may be NULL); act = ReLU when relu.
 * bwd_mid: from part_s (or `sums`, as consts2): d2e2, dgamma2, dbeta2, evec[H] = E2 W2 and the B image of
 *   [W2 ; Qm], Qm = W2^T diag(D2) W2.
 * bwd_fin: from part_t (or `sums`, as consts1): cabc, dgamma1, dbeta1; and g_w2 (O,H) = R_S + D2 (W2 Gram)
 *   + E2 (x) suma, with R float64[(O+H) H + H] = {R_S ; Gram ; suma} (apn_sa_wide_wgrad's splits summed).
 * point_grads (C, H <= 64): G = dL/dU per point (GU rows summed through the inverse map, plus BatchNorm-1's
 *   mean/variance terms), g_f (B,C,N) = G W1f, g_p (B,N,3) = G W1p / r, g_q (B,M,3) = -Hq W1p / r (either may
 *   be NULL); with a residual branch (c_skip = O > 0: gpre (B,M,O) from bwd_prep, fq (B,N) = the query a point
 *   is or -1, fs (B,M,C) of fwd_prep, ws (O x C)) g_f also receives ws^T gpre at the sampled points;
 *   w_part[apn_sa_wide_point_grads_rows(b, n)][apn_sa_wide_point_grads_cols(C, H, O)] = the workgroups' shares
 *   of {dL/dW1 (H x (C+3)), dL/dws (O x C), dL/dbs (O)}; w_part NULL: no weight takes a gradient, the shares are
 *   not formed (the frozen classifier of the GAN's feedback pass).
 * colsum_f32: out[ncol] (float32) = column sums (in float64, fixed order) of part[rows][ncol]. */
APN_API int apn_sa_wide_fwd_prep(int b, int c_in, int n, int m, int c_mid, int c_out, float radius, const float *f,
                                 const float *p, const float *new_p, const float *w1, const float *w2, float *U,
                                 float *V, void *w2_image, const int *fq, float *fs, const float *geo, float *part1,
                                 void *stream);
APN_API int apn_sa_wide_fwd_prep_rows(int b, int n, int m);
APN_API int apn_sa_wide_out(int b, int m, int c_out, const float *ysel, const float *pack2, int c_in,
                            const float *fs, const float *ws, const float *bs, int relu, float *out, void *stream);
APN_API int apn_sa_wide_bwd_mid(const float *part_s, int rows, const double *sums, int c_mid, int c_out,
                                const float *pack2, double count, int training, const float *w2, float *d2e2,
                                float *g_gamma2, float *g_beta2, float *evec, void *z_image, void *stream);
APN_API int apn_sa_wide_bwd_fin(const float *part_t, int rows, const double *sums, int c_mid, int c_out,
                                const float *pack1, double count, int training, float *cabc, float *g_gamma1,
                                float *g_beta1, const double *R, const float *d2e2, const float *w2, float *g_w2,
                                void *stream);
APN_API int apn_sa_wide_point_grads_rows(int b, int n);
APN_API int apn_sa_wide_point_grads(int b, int c_in, int n, int m, int c_mid, float radius, const float *GU,
                                    const int *pcnt_poff, const int *plist, const float *geo, const float *U,
                                    const float *f, const float *p, const float *new_p, const float *HA,
                                    const float *HB, const float *cabc, const float *pack1, const float *w1,
                                    int c_skip, const float *gpre, const int *fq, const float *fs, const float *ws,
                                    float *g_f, float *g_p, float *g_q, float *w_part, void *stream);
APN_API int apn_sa_wide_point_grads_cols(int c_in, int c_mid, int c_skip);
APN_API int apn_sa_wide_colsum_f32(const float *part, int rows, int ncol, float *out, void *stream);

/* ------------------------------------------------------------------------
 * the imitator's per-point MLP layer `ConvBNReLU1D` =
 * Conv1d(kernel 1, no bias) + BatchNorm1d (+ ReLU)
 * (openpoints/models_adaptpoint/generator_component4_15.py:92-104; instantiated as the embedding, the four
 * extract_feat_list layers :588-657 and the `fuse` layer of PointNetFeaturePropagation :330-366), which the
 * reference runs as cuDNN convolution / batch-norm calls.  Tensors are channels-first contiguous float32 as
 * torch.nn.Conv1d takes them: x (B,c_in,N), w (c_out,c_in), y / out / g / gy (B,c_out,N).  Any sizes.
 * precision = 2: operands split into two bf16 planes (three MFMAs per product, ~4e-6 of an fp32 contraction);
 *   3: three planes, six MFMAs, fp32-class (~2e-7).  f32 accumulation either way.
 * conv_forward: y = w x per cloud; part (may be NULL)
 *   [apn_pw_conv_tiles(b, n)][2][c_out] = each 128-position tile's {sum, M2 = sum of squared deviations from the
 *   tile's own mean} of y per channel (combined over tiles in float64 by bn_act: no E[y^2] - mean^2 cancellation).
 * bn_act: out = [relu](gamma (y - mean) invstd + beta); training: batch statistics folded (float64) from `part`
 *   (tiles rows), running_mean / running_var updated with `momentum` (unbiased variance) and batches[0] += 1
 *   (each may be NULL); otherwise the running statistics are used.  stat [4][c] = {mean, invstd, scale, shift}.
 * bn_act_grad: gy = dL/dy from g = dL/dout (BatchNorm's batch-statistics gradient when training), g_gamma,
 *   g_beta [c] (may be NULL); part_b = scratch [apn_pw_bn_act_grad_splits(b, c)][2][c].  One launch when a channel's b * n
 *   values fit one workgroup's registers (b * n <= 32768, n % 4 == 0), else two.
 * conv_grad_input: gx (B,c_in,N) = w^T gy.   conv_grad_weight: gw (c_out,c_in) = sum_b gy_b x_b^T, split over
 *   apn_pw_conv_grad_weight_splits(...) ranges of (cloud, position) whose shares (scratch [splits][c_out][c_in])
 *   are added in a fixed order: bit-reproducible. */
APN_API int apn_pw_conv_tiles(int b, int n);
APN_API int apn_pw_conv_forward(int b, int c_in, int c_out, int n, int precision, const float *x, const float *w,
                                float *y, float *part, void *stream);
APN_API int apn_pw_bn_act(int b, int c, int n, const float *y, const float *part, int tiles, const float *gamma,
                          const float *beta, float eps, float momentum, int training, int relu, float *run_mean,
                          float *run_var, long long *batches, float *stat, float *out, void *stream);
APN_API int apn_pw_bn_act_grad_splits(int b, int c);
APN_API int apn_pw_bn_act_grad(int b, int c, int n, const float *g, const float *y, const float *stat, int training,
                               int relu, float *part_b, float *gy, float *g_gamma, float *g_beta, void *stream);
APN_API int apn_pw_conv_grad_input(int b, int c_in, int c_out, int n, int precision, const float *gy, const float *w,
                                   float *gx, void *stream);
APN_API int apn_pw_conv_grad_weight_splits(int b, int c_in, int c_out, int n);
APN_API int apn_pw_conv_grad_weight(int b, int c_in, int c_out, int n, int precision, const float *gy, const float *x,
                                    float *scratch, float *gw, void *stream);

/* The contraction kernel of the per-point layers as such (csrc/pointwise.hip), for the small dense products around
 * the fused set-abstraction blocks (conv1 at the points, dL/df, dL/dW1 of the wide blocks):
 *   splits == 0: d[z] (r x q) = a[z] (r x k) b[z] (k x q), z < nbatch (a batch stride of 0 shares the operand);
 *   splits  > 0: d (r x q, ldd == q) = sum_z a[z] b[z] in `splits` shares (apn_pw_contract_splits; scratch
 *                [splits][r][q]) added in a fixed order.
 * a_kcont / b_kcont: the operand's element (i, k) lies at i * ld + k (1) or at k * ld + i (0). */
APN_API int apn_pw_contract_splits(int nbatch, int r, int q, int k);
APN_API int apn_pw_contract(int nbatch, int r, int q, int k, const float *a, long long a_batch, int lda, int a_kcont,
                            const float *b, long long b_batch, int ldb, int b_kcont, float *d, long long d_batch, int ldd,
                            int splits, float *scratch, int precision, void *stream);
/* out[z][j][i] = in[z][i][j] for in (nbatch, r, c) float32, contiguous: the layout change between the per-point layers'
 * (B, C, N) and the grouper's / attention's (B, N, C) (generator_component4_15.py:650-657 permutes and lets PyTorch copy). */
APN_API int apn_pw_transpose(int nbatch, int r, int c, const float *in, float *out, void *stream);
/* three_nn's squared distances (n, 3) -> the inverse-distance weights of openpoints/models/layers/upsampling.py:97-100,
 * w_j = (1 / (sqrt(d2_j) + 1e-8)) / sum_j (...): one launch for the five elementwise ones of the PyTorch form. */
APN_API int apn_three_nn_weights(long long n, const float *dist2, float *weight, void *stream);

/* the per-anchor transforms of AdaptPoint_Augmentor.local_transformaton
 * (openpoints/models_adaptpoint/generator_component4_15.py:236-297): prob (n,9) the imitator's numbers per anchor,
 * keep (n,3) / axes (n,3) the call's random switches as floats -> lin (n,3,3) = R diag(s), off (n,3); formulas in
 * csrc/augment.hip.  _grad: g_prob (n,9) from g_lin (n,3,3) and g_off (n,3) (either may be NULL = zero). */
APN_API int apn_anchor_transforms(int n, const float *prob, const float *keep, const float *axes, float r_range,
                                  float s_range, float t_range, float *lin, float *off, void *stream);
APN_API int apn_anchor_transforms_grad(int n, const float *prob, const float *keep, const float *axes, float r_range,
                                       float s_range, float t_range, const float *g_lin, const float *g_off,
                                       float *g_prob, void *stream);

/* The deformation that consumes them (generator_component4_15.py:156-232, 313-327 and the mask at :180): kernel
 * regression weights towards the m <= 8 anchors, blend of the anchors' affine maps, unit-sphere normalisation, mask:
 *   xyz (B,n,3), anchors (B,m,3), lin (B,m,3,3), off (B,m,3), axes (B,3) floats, mask (B,n) or NULL ->
 *   out (B,n,3); z (B,n,3) and stat (B,8) are kept for the backward.  n <= 4096.  One workgroup per cloud.
 * _backward: g_lin (B,m,3,3), g_off (B,m,3), g_mask (B,n) (may be NULL) from g_out (B,n,3). */
APN_API int apn_deform_forward(int b, int n, int m, const float *xyz, const float *anchors, const float *lin,
                               const float *off, const float *axes, const float *mask, float sigma, float *z, float *stat,
                               float *out, void *stream);
APN_API int apn_deform_backward(int b, int n, int m, const float *xyz, const float *anchors, const float *axes,
                                const float *mask, float sigma, const float *z, const float *stat, const float *g_out,
                                float *g_lin, float *g_off, float *g_mask, void *stream);

/* The last layer of the discriminator's group-all stage with its pooling
 * (openpoints/models_adaptpoint/point_discriminator.py:183-189: conv -> ReLU -> max over the cloud's points), fused:
 *   out (B,c_out) = [relu](max_n (w x_b)[o][n] + bias[o]),  idx (B,c_out) int32 = the position of that maximum (the
 *   lowest one among equals); the (B,c_out,N) activation is never written.  bias may be NULL.  Scratch: tile_val /
 *   tile_idx [b][apn_pw_conv_max_tiles(n)][c_out].
 * Backward (c_in <= 128): every (b, o) passes its gradient to position idx[b][o] alone -- g_x (B,c_in,N) (fully
 *   written; may be NULL), g_w (c_out,c_in), g_bias [c_out] (may be NULL); xsel = scratch (B,c_out,c_in).  No
 *   atomics: fixed orders, bit-reproducible. */
APN_API int apn_pw_conv_max_tiles(int n);
APN_API int apn_pw_conv_max_forward(int b, int c_in, int c_out, int n, int precision, const float *x, const float *w,
                                    const float *bias, int relu, float *tile_val, int *tile_idx, float *out, int *idx,
                                    void *stream);
APN_API int apn_pw_conv_max_backward(int b, int c_in, int c_out, int n, const float *g_out, const float *out,
                                     const int *idx, const float *x, const float *w, int relu, float *xsel, float *g_x,
                                     float *g_w, float *g_bias, void *stream);

/* ------------------------------------------------------------------------
 * spectral normalisation, the parametrisation of every layer of PointDiscriminator1
 * (openpoints/models_adaptpoint/point_discriminator.py:17-73, 149-191: torch.nn.utils.spectral_norm, which the
 * reference evaluates as ~13 small PyTorch launches per training-mode forward and ~7 per backward).
 *   w (rows x cols) the weight as a matrix; u [rows], v [cols] the module's power-iteration buffers.
 * spectral_norm (3 launches; 2 when !training): training: one power iteration u = normalize(w v),
 *   v = normalize(w^T u) written INTO u, v; sigma[0] = u^T w v; w_normalized = w / sigma; u_used / v_used = the
 *   vectors sigma was formed with (kept for the backward, as PyTorch clones them); scratch [rows + cols] floats.
 * spectral_norm_grad (2 launches): g_w = g / sigma - (sum(g o w_normalized) / sigma) u_used v_used^T;
 *   part = scratch of apn_spectral_norm_blocks(rows, cols) doubles.  Fixed summation orders: reproducible. */
APN_API int apn_spectral_norm_blocks(int rows, int cols);
APN_API int apn_spectral_norm(int rows, int cols, const float *w, int training, float eps, float *u, float *v,
                              float *scratch, float *u_used, float *v_used, float *sigma, float *w_normalized,
                              void *stream);
APN_API int apn_spectral_norm_grad(int rows, int cols, const float *g, const float *w_normalized, const float *sigma,
                                   const float *u_used, const float *v_used, double *part, float *g_w, void *stream);

/* The same for ALL layers of a network in one set of launches (3 forward, 2 backward; the layer rides on blockIdx.y): arrays
 * of n_layers (<= 8) entries with the meaning of the arguments above; scratch[i]: rows[i] + cols[i] floats; part[i]:
 * apn_spectral_norm_blocks(rows[i], cols[i]) doubles.  (PointDiscriminator1 holds seven spectral-normalised layers and is
 * evaluated three times per train_gan iteration, point_discriminator.py:17-73, train_autoaug.py:150-196.) */
APN_API int apn_spectral_norm_many(int n_layers, const int *rows, const int *cols, const float *const *w, int training,
                                   float eps, float *const *u, float *const *v, float *const *scratch,
                                   float *const *u_used, float *const *v_used, float *const *sigma,
                                   float *const *w_normalized, void *stream);
APN_API int apn_spectral_norm_grad_many(int n_layers, const int *rows, const int *cols, const float *const *g,
                                        const float *const *w_normalized, const float *const *sigma,
                                        const float *const *u_used, const float *const *v_used, double *const *part,
                                        float *const *g_w, void *stream);

/* Tuning / diagnostic entry, NOT part of the reference boundary: apn_furthest_point_sampling
 * with the number of wavefronts that cooperate on one cloud (1, 2, 4, 8 or 16; 0 = the built-in
 * heuristic) and the step algorithm (0 = one LDS 64-bit atomic max per step for n <= 4096: what the operator
 * entries run; 1 = per-wave records + second reduction) chosen PER CALL.  No process-wide state: every entry
 * point of this library is re-entrant.  Results do not depend on either argument. */
APN_API int apn_furthest_point_sampling_tuned(int b, int n, int m, const float *xyz, float *temp,
                                              int *idx, int waves, int algo, void *stream);

/* Diagnostic only: the FPS step of the n = 1024 geometry with s_memtime stamps; dbg[0..5]
 * = cycles summed over the m-1 steps for {update, wave max, pick+LDS write, barrier,
 * LDS read, group max + broadcast}.  Not part of the product path. */
APN_API int apn_fps_debug_stamps(int b, int n, int m, const float *xyz, float *temp, int *idxs,
                                 unsigned long long *dbg, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ADAPTPOINT_AMD_H */
