"""Width-generic fused grouped MLP of a PointNeXt set-abstraction block (csrc/sa_wide.hip).

    out (B,O,M) = max_K bn2(conv2(relu(bn1(conv1(cat[(p[idx] - new_p) / r, f[idx]])))))
    (openpoints/models/backbone/pointnext.py:157-166 over QueryAndGroup, group.py:235-255)

for every block of PointNeXt-S: C_in = 32..256, C_mid = H in {32, 64, 128, 256}, C_out = 2H, K = 32.
How the work is split (see the header of csrc/sa_wide.hip):

  * conv1 is hoisted to the POINTS: U = W1f f + W1p p / r (B,N,H), V = W1p new_p / r (B,M,H), so
    that y1[q,k] = U[idx[q,k]] - V[q].  U, V and every other product whose contraction runs over
    points or channels only (dL/df = G W1f, dL/dW1 = G^T [p, f], the BatchNorm constants, ...) are
    plain dense algebra on small tensors and stay PyTorch/rocBLAS fp32 here;
  * everything that runs over the B*M*K POSITIONS is a hand-written kernel: statistics of y1,
    the MFMA contraction a1 W2^T with its statistics and the pool, the backward through the
    pool / BatchNorm-2 / conv2 / ReLU with the per-point scatter, and the weight-gradient
    products.  No (B,C,M,K) tensor is ever materialised, forward or backward.

BatchNorm follows the modules' training flag; with sync_bn the float64 sums are all-reduced
over ranks at the four points where statistics leave the kernels (SyncBatchNorm semantics,
gradients of gamma/beta reported as global / world -- see adaptpoint_amd.fused).
"""
import torch

from . import _lib, pointwise
from . import fused as _fz
from .fused import _call

K_NS = 32
LEAN_MAX = 64        # widest C_in / C_mid whose dense products run in csrc/sa_wide_dense.hip
WIDTHS = (32, 64, 128, 256)


def supported(p, f, idx_or_k, conv1, conv2, bns=(), npoint=None):
    """npoint: the number of queries when `idx_or_k` is the neighbourhood size alone (the kernels pack query ids
    into 24 bits: B * npoint is bounded)."""
    k = idx_or_k.shape[2] if torch.is_tensor(idx_or_k) else int(idx_or_k)
    m = idx_or_k.shape[1] if torch.is_tensor(idx_or_k) else (p.shape[1] if npoint is None else int(npoint))
    H, O = conv1.weight.shape[0], conv2.weight.shape[0]
    return (f.is_cuda and f.dtype == torch.float32 and p.dtype == torch.float32 and k == K_NS
            and H in WIDTHS and O == 2 * H and conv1.weight.shape[1] == f.shape[1] + 3
            and conv2.weight.shape[1] == H and conv1.bias is None and conv2.bias is None
            and p.shape[0] * m < 2 ** 24
            and (64 * (H + 1) + (f.shape[1] + 3) * 65) * 4 <= 160 * 1024        # LDS of the per-point gradient kernel
            and all(bn.momentum is not None for bn in bns))


def mfma_b_image(Bm, ct):
    """B operand image of a contraction y[row, col] = sum_k A[row, k] Bm[k, col] for the wave-per-tile
    MFMA kernels: Bm (Kd, Nc) fp32 with Kd % 32 == 0 and Nc % (32 ct) == 0 -> bf16 tensor
    [col block][k chunk of 32][col tile j < ct][k-step s < 2][part hi/lo][lane = 32 h + r][8]:
    lane (r, h) of fragment (j, s) holds Bm[32 kc + 16 s + 8 h + e][32 (ct cb + j) + r], e < 8
    (the B lane map of v_mfma_f32_32x32x16_bf16, csrc/apn_mfma.h); every (col block, k chunk) is
    one contiguous 1024 ct -byte ... block the kernels copy into LDS verbatim."""
    Kd, Nc = Bm.shape
    ncb, nkc = Nc // (32 * ct), Kd // 32
    hi = Bm.to(torch.bfloat16)
    lo = (Bm - hi.float()).to(torch.bfloat16)
    x = torch.stack([hi, lo], 0).view(2, nkc, 2, 2, 8, ncb, ct, 32)      # p, kc, s, h, e, cb, j, r
    return x.permute(5, 1, 6, 2, 0, 3, 7, 4).contiguous()                # cb, kc, j, s, p, h, r, e


def _colsum(rows2d):
    """float64 column sums of a float32 (rows, ncol) tensor by the extension's own fixed-order kernel.
    (torch's multi-block reductions rely on semaphores that did not survive hipGraph replay here:
    `x.double().sum(0)` returned stale values from the second replay on.)"""
    rows2d = rows2d.contiguous()
    rows, ncol = rows2d.shape
    chunks = _lib.load().apn_sa_wide_colsum_chunks(rows, ncol)
    buf = torch.empty((chunks + 1) * ncol, dtype=torch.float64, device=rows2d.device)
    out = buf[:ncol]
    _call("apn_sa_wide_colsum", rows2d.device, rows2d.data_ptr(), rows, ncol, buf[ncol:].data_ptr(), out.data_ptr())
    return out


def _bn_pack(part, rows, C, count, bn, dev, training, sync, sgn_from=None, sgn_c=0):
    """BatchNorm fold by the extension's `apn_sa_bn_fold` (csrc/sa_glue.hip): the partial rows
    {sum[C], sumsq[C]} -> pack {scale, shift, mean, invstd}[C] (float32, (4C,)), running buffers and
    num_batches_tracked updated as torch.nn.BatchNorm does; with `sync` the float64 sums, the position
    count and a 1 are all-reduced over ranks first.  Rider: sgn (sign of another BatchNorm's gamma)."""
    pack = torch.empty(4 * C, dtype=torch.float32, device=dev)
    sgn = torch.empty(sgn_c, dtype=torch.float32, device=dev) if sgn_c else None
    sums = None
    if training and sync:
        sums = torch.cat([_colsum(part), torch.full((1,), float(count), dtype=torch.float64, device=dev),
                          torch.ones(1, dtype=torch.float64, device=dev)])
        _fz._allreduce_sum_(sums)
    args = _fz._bn_args(bn)      # (gamma, beta, running_mean, running_var, nbt, eps, momentum, training)
    _call("apn_sa_bn_fold", dev, _fz._ptr(part) if (training and sums is None) else None, rows, _fz._ptr(sums), C,
          float(count), args[0], args[1], args[5], args[6], args[2], args[3], args[4], 1 if training else 0,
          pack.data_ptr(), _fz._ptr(sgn_from), sgn_c, _fz._ptr(sgn))
    return pack, sgn, sums


def _image(src0, k0, trans0, src1, kd, nc, ct):
    """B image (bf16 hi/lo, MFMA fragment order) of the (kd x nc) operand whose first k0 rows come from
    src0 and the rest from src1 -- one launch (`mfma_b_image` is the tensor-op statement of the layout)."""
    img = torch.empty(nc // (32 * ct), kd // 32, ct, 2, 2, 64, 8, dtype=torch.bfloat16, device=src0.device)
    _call("apn_sa_wide_image", src0.device, src0.data_ptr(), k0, 1 if trans0 else 0, _fz._ptr(src1), kd, nc, ct,
          img.data_ptr())
    return img


class NeighbourIndex:
    """What the fused passes need to know about the neighbour indices idx (B,M,32) beyond idx itself -- all of
    it a function of the coordinates only, so it belongs to the INDEX stage and can be built once per batch,
    off the feature stream (`neighbour_index`, `Sampling.index`, `index_pyramid`):
      tmap       distinct-hit tile map (csrc/sa_wide_glue.hip): rows = (query, distinct slot, multiplicity)
                 packed 32 to an MFMA tile;
      pcnt_poff  inverse map: per support point the number of rows that gather it and where its list starts;
      plist      the lists (row ids, ascending);   geo (B,N,4): occurrences, sum of the gathering queries' coordinates;
      fidx / fq  (optional) the point every query is (FPS picks) and its inverse (the query a point is, or -1).
    """
    __slots__ = ("idx", "tmap", "pcnt_poff", "plist", "geo", "n_points", "fidx", "fq")

    def __init__(self, idx, tmap, pcnt_poff, plist, geo, n_points, fidx=None, fq=None):
        self.idx, self.tmap, self.pcnt_poff, self.plist, self.geo, self.n_points = idx, tmap, pcnt_poff, plist, geo, n_points
        self.fidx, self.fq = fidx, fq


def tile_map(idx, fold=True, out=None):
    """The tile map alone (see NeighbourIndex).  fold=False: one 32-row tile per query (no use made of the
    ball-query structure)."""
    B, M, K = idx.shape
    assert K == K_NS and idx.dtype == torch.int32 and idx.is_contiguous()
    tmap = out if out is not None else torch.empty(_lib.load().apn_sa_wide_tilemap_ints(B, M), dtype=torch.int32,
                                                  device=idx.device)
    _call("apn_sa_wide_tilemap", idx.device, B, M, 1 if fold else 0, idx.data_ptr(), tmap.data_ptr())
    return tmap


def tile_maps(idx_all, count, fold=True, out=None):
    """The tile maps of `count` consecutive batches of one stacked index stage -- idx_all (count * B, M, 32) -- in ONE
    pair of launches: a tensor (count, ints), row z the map of batch z (what `tile_map` gives for idx_all[z*B:(z+1)*B])."""
    SB, M, K = idx_all.shape
    assert K == K_NS and idx_all.dtype == torch.int32 and idx_all.is_contiguous() and SB % count == 0
    B = SB // count
    ints = _lib.load().apn_sa_wide_tilemap_ints(B, M)
    maps = out if out is not None else torch.empty(count, ints, dtype=torch.int32, device=idx_all.device)
    _call("apn_sa_wide_tilemap_many", idx_all.device, count, B, M, 1 if fold else 0, idx_all.data_ptr(), maps.data_ptr())
    return maps


def row_maps(maps, count, B, N, M, out=None, fidx=None):
    """The ROW MAPS of `count` stacked tile maps (`tile_maps`; count = 1: `maps` may be one map) in one launch:
    (pcnt_poff (count, 2 B N + B) int32, rowdst (count, 32 B M) int32) -- per support point how many tile-map rows gather it and
    the place of the first of them in the point-sorted order, then per cloud whether its picks `fidx` ((count B, M) int32, the
    FPS indices; None: not examined) hit a point twice; per row its place.  Index-stage data (a pure function of the
    neighbour indices), consumed by the register-resident backward pass (csrc/sa_fused.hip: the rows of g_u are stored at
    their places, so a point's rows are contiguous: no float atomics) and its per-point kernel."""
    dev = maps.device
    if out is None:
        out = (torch.empty(count, 2 * B * N + B, dtype=torch.int32, device=dev),
               torch.empty(count, 32 * B * M, dtype=torch.int32, device=dev))
    pcnt_poff, rowdst = out
    assert pcnt_poff.numel() == count * _lib.load().apn_sa_rowmap_ints(B, N) and rowdst.numel() == count * 32 * B * M
    assert maps.is_contiguous() and (fidx is None or (fidx.is_contiguous() and fidx.dtype == torch.int32
                                                      and fidx.numel() == count * B * M))
    scratch = torch.empty(32 * B * M, dtype=torch.int32, device=dev)        # (the multi-launch path's lists)
    _call("apn_sa_rowmap_many", dev, count, B, N, M, maps.data_ptr(), fidx.data_ptr() if fidx is not None else None,
          pcnt_poff.data_ptr(), rowdst.data_ptr(), scratch.data_ptr())
    return out


def row_map(tmap, B, N, M, out=None, fidx=None):
    """`row_maps` for one tile map: (pcnt_poff (2 B N + B), rowdst (32 B M))."""
    if out is not None:
        out = (out[0].view(1, -1), out[1].view(1, -1))
    pcnt_poff, rowdst = row_maps(tmap, 1, B, N, M, out=out, fidx=fidx)
    return pcnt_poff.view(-1), rowdst.view(-1)


@torch.no_grad()
def neighbour_index(idx, new_p, n_points, fold=True, fidx=None, out=None):
    """Build the NeighbourIndex of idx (B,M,32) over n_points support points; new_p (B,M,3): the queries;
    fidx (B,M) int32 (optional): the FPS picks, for blocks with a residual branch.  out: a NeighbourIndex of
    the same shapes to refill (its buffers are kept: a captured hipGraph reads the same addresses)."""
    assert idx.is_contiguous() and new_p.is_contiguous() and (fidx is None or fidx.is_contiguous())
    B, M, _ = idx.shape
    dev = idx.device
    if out is None:
        out = NeighbourIndex(idx, None, torch.empty(2 * B * n_points, dtype=torch.int32, device=dev),
                             torch.empty(32 * B * M, dtype=torch.int32, device=dev),
                             torch.empty(B, n_points, 4, dtype=torch.float32, device=dev), n_points, fidx,
                             None if fidx is None else torch.empty(B, n_points, dtype=torch.int32, device=dev))
    else:
        assert out.n_points == n_points and out.plist.numel() == 32 * B * M and (fidx is None) == (out.fq is None)
        out.idx, out.fidx = idx, fidx
    out.tmap = tile_map(idx, fold, out=out.tmap)
    _call("apn_sa_wide_csr", dev, B, n_points, M, idx.data_ptr(), new_p.data_ptr(), out.tmap.data_ptr(),
          out.pcnt_poff.data_ptr(), out.plist.data_ptr(), out.geo.data_ptr(), _fz._ptr(fidx), _fz._ptr(out.fq))
    return out


def _training(bn):
    return bn.training or not bn.track_running_stats


def lean(C, H):
    """Shapes whose dense products (conv1 at the points, Qm, dL/df, dL/dW1) run in the path's own fused
    kernels (csrc/sa_wide_dense.hip); wider ones hand them to library GEMMs."""
    return C <= LEAN_MAX and H <= LEAN_MAX and C % 4 == 0


class _WideBlock(torch.autograd.Function):
    """max_K bn2(conv2(relu(bn1(conv1(grouped))))) [+ ws f[fidx] + bs] [ReLU] -- forward and backward."""

    @staticmethod
    def forward(ctx, p, new_p, f, w1, g1, b1, w2, g2, b2, ws, bs, mods):
        radius, bn1, bn2, sync_bn, nbr, relu = mods
        p, new_p, f = p.contiguous(), new_p.contiguous(), f.contiguous()
        idx, tmap = nbr.idx, nbr.tmap
        dev = f.device
        B, C, N = f.shape
        M = new_p.shape[1]
        H, O = w1.shape[0], w2.shape[0]
        fusedd = lean(C, H)
        assert ws is None or (fusedd and nbr.fq is not None)
        sync = sync_bn and (_fz._world(True) > 1 or _fz.FORCE_PHASED)
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.no_grad():
            W1 = w1.detach().reshape(H, C + 3).contiguous()
            W2 = w2.detach().reshape(O, H).contiguous()
            Ws = None if ws is None else ws.detach().reshape(O, C).contiguous()
            # conv1 at the points: one row per support point, one per query; the image of W2^T
            part1, rows1 = None, 0
            if fusedd:
                U = torch.empty(B, N, H, **f32)
                V = torch.empty(B, M, H, **f32)
                ct = min(4, O // 32)
                w2img = torch.empty(O // (32 * ct), H // 32, ct, 2, 2, 64, 8, dtype=torch.bfloat16, device=dev)
                # (with a residual branch: also the sampled points' own features as rows fs (B,M,C))
                fs = torch.empty(B, M, C, **f32) if Ws is not None else None
                # BatchNorm-1's batch statistics come out of the same launch (per-point / per-query terms weighted
                # by the index stage's occurrence counts: no pass over the positions)
                if _training(bn1) and K_NS == 32:
                    rows1 = _lib.load().apn_sa_wide_fwd_prep_rows(B, N, M)
                    part1 = torch.empty(rows1, 2 * H, **f32)
                _call("apn_sa_wide_fwd_prep", dev, B, C, N, M, H, O, float(radius), f.data_ptr(), p.data_ptr(),
                      new_p.data_ptr(), W1.data_ptr(), W2.data_ptr(), U.data_ptr(), V.data_ptr(), w2img.data_ptr(),
                      _fz._ptr(nbr.fq if Ws is not None else None), _fz._ptr(fs),
                      _fz._ptr(nbr.geo if part1 is not None else None), _fz._ptr(part1))
            else:
                # conv1's input per point, channels-first as W1's columns order it: [p / r ; f]  (B, C + 3, N); then
                # U[b] (N x H) = X[b]^T W1^T on the per-point contraction kernel, operands read where they lie
                # (as PyTorch ops: a skinny matmul, a division and a baddbmm of 40 us)
                X = torch.cat([p.transpose(1, 2) / radius, f], 1)
                U = torch.empty(B, N, H, **f32)
                pointwise.contract(B, N, H, C + 3, X, (C + 3) * N, N, False, W1, 0, C + 3, True, U, d_batch=N * H, ldd=H)
                V = (torch.matmul(new_p, W1[:, :3].t()) / radius).contiguous()
                w2img = _image(W2, H, True, None, H, O, min(4, O // 32))          # W2^T (H x O)
            grid = _lib.load().apn_sa_wide_grid(B, M)
            count = float(B * M * K_NS)
            tr1, tr2 = _training(bn1), _training(bn2)
            if tr1 and part1 is None:
                rows1 = grid
                part1 = torch.empty(grid, 2 * H, **f32)
                _call("apn_sa_wide_stats1", dev, B, N, M, H, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
                      tmap.data_ptr(), part1.data_ptr())
            pack1, sgn2, _ = _bn_pack(part1, rows1, H, count, bn1, dev, tr1, sync, sgn_from=g2, sgn_c=O)
            ysel = torch.empty(B, M, O, **f32)
            ksel = torch.empty(B, M, O, dtype=torch.uint8, device=dev)
            part2 = torch.empty(grid, 2 * O, **f32)
            _call("apn_sa_wide_fwd_main", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
                  tmap.data_ptr(), w2img.data_ptr(), pack1.data_ptr(), sgn2.data_ptr(), ysel.data_ptr(), ksel.data_ptr(),
                  part2.data_ptr())
            pack2, _, _ = _bn_pack(part2 if tr2 else None, grid, O, count, bn2, dev, tr2, sync)
            out = torch.empty(B, O, M, **f32)
            _call("apn_sa_wide_out", dev, B, M, O, ysel.data_ptr(), pack2.data_ptr(), C,
                  _fz._ptr(fs if Ws is not None else None), _fz._ptr(Ws),
                  _fz._ptr(None if bs is None else bs.detach()), 1 if relu else 0, out.data_ptr())
        ctx.save_for_backward(p, new_p, f, U, V, pack1, pack2, ysel, ksel, W1, W2, Ws, out if relu else None,
                              fs if Ws is not None else None, None if fusedd else X)
        ctx.nbr = nbr
        ctx.cfg = (radius, tr1, tr2, sync, count, relu, g1 is not None, b1 is not None, g2 is not None, b2 is not None,
                   bs is not None)
        ctx.need = (p.requires_grad, new_p.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        p, new_p, f, U, V, pack1, pack2, ysel, ksel, W1, W2, Ws, out_act, fs, X = ctx.saved_tensors
        nbr = ctx.nbr
        idx, tmap = nbr.idx, nbr.tmap
        radius, tr1, tr2, sync, count, relu, a1, a2, a3, a4, a5 = ctx.cfg
        need_p, need_q = ctx.need
        need_w = any(ctx.needs_input_grad[3:11])       # any weight of the block (frozen in the GAN feedback pass)
        dev = f.device
        B, C, N = f.shape
        M = new_p.shape[1]
        H, O = W1.shape[0], W2.shape[0]
        fusedd = lean(C, H)
        lib = _lib.load()
        if g.dtype != torch.float32:
            g = g.float()
        f32 = dict(dtype=torch.float32, device=dev)

        def reduced(part):
            """SyncBatchNorm: the rows summed, with {count, 1} appended, all-reduced over ranks."""
            v = torch.cat([_colsum(part), torch.full((1,), count, dtype=torch.float64, device=dev),
                           torch.ones(1, dtype=torch.float64, device=dev)])
            _fz._allreduce_sum_(v)
            return v
        # the upstream gradient (through the block's ReLU) in query-major layout, BatchNorm-2's row sums
        prow = lib.apn_sa_wide_bwd_prep_rows(B, M)
        goa = torch.empty(B, M, O, **f32)
        gpre = torch.empty(B, M, O, **f32) if Ws is not None else None
        partS = torch.empty(prow, 2 * O, **f32)
        gs = g.stride()
        _call("apn_sa_wide_bwd_prep", dev, B, M, O, g.data_ptr(), gs[0], gs[1], gs[2], ysel.data_ptr(),
              pack2.data_ptr(), _fz._ptr(out_act), _fz._ptr(gpre), goa.data_ptr(), partS.data_ptr())
        small = torch.empty(2 * O + O + O + H + 3 * H + 2 * H, **f32)
        o = 0
        d2e2 = small[o:o + 2 * O]; o += 2 * O
        g_gamma2 = small[o:o + O]; o += O
        g_beta2 = small[o:o + O]; o += O
        evec = small[o:o + H]; o += H
        cabc = small[o:o + 3 * H]; o += 3 * H
        g_gamma1 = small[o:o + H]; o += H
        g_beta1 = small[o:o + H]
        sS = reduced(partS) if sync else None
        if fusedd:
            # BatchNorm-2 backward constants, Qm = W2^T diag(D2) W2, evec = E2 W2, image of [W2 ; Qm]: one launch
            ctz = min(4, H // 32)
            zimg = torch.empty(H // (32 * ctz), (O + H) // 32, ctz, 2, 2, 64, 8, dtype=torch.bfloat16, device=dev)
            _call("apn_sa_wide_bwd_mid", dev, None if sync else partS.data_ptr(), prow, _fz._ptr(sS), H, O,
                  pack2.data_ptr(), count, 1 if tr2 else 0, W2.data_ptr(), d2e2.data_ptr(), g_gamma2.data_ptr(),
                  g_beta2.data_ptr(), evec.data_ptr(), zimg.data_ptr())
        else:
            _call("apn_sa_wide_consts2", dev, None if sync else partS.data_ptr(), prow, _fz._ptr(sS), O,
                  pack2.data_ptr(), count, 1 if tr2 else 0, d2e2.data_ptr(), g_gamma2.data_ptr(), g_beta2.data_ptr())
            W2t = W2.t().contiguous()
            Qm = pointwise.matmul_nt(W2t * d2e2[:O], W2t)                         # W2^T diag(D2) W2  (H,H)
            torch.mv(W2.t(), d2e2[O:], out=evec)
            zimg = _image(W2, O, False, Qm, O + H, H, min(4, H // 32))           # [W2 ; Qm]  ((O+H) x H)
        grid = lib.apn_sa_wide_grid(B, M)
        GU = torch.empty(B * M * K_NS, H, **f32)          # one row per tile-map row (upper bound; rows in use written)
        HA = torch.empty(B, M, H, **f32)
        HB = torch.empty(B, M, H, **f32)
        partT = torch.empty(grid, 2 * H, **f32)
        # weight-gradient products over the positions, {[S^T ; a1^T] a1, sum a1}: one partial row per workgroup of
        # the backward pass itself (H = 32), or per split of their own pass
        rows = O + H
        wg_fused = bool(lib.apn_sa_wide_wgrad_fused(H))
        if wg_fused:
            splits = grid
        else:
            groups = (rows // 32 + 7) // 8
            splits = max(1, min(512 // groups, (B * M) // 4, (64 << 20) // (rows * H * 4)))
        Rpart = torch.empty(splits, rows * H + H, **f32)
        # (eval-mode BatchNorm-2: Qm = 0 and evec = 0 -- the kernel skips that third of its chain on a NULL evec)
        _call("apn_sa_wide_bwd_main", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
              tmap.data_ptr(), zimg.data_ptr(), pack1.data_ptr(), evec.data_ptr() if tr2 else None, goa.data_ptr(),
              ksel.data_ptr(),
              GU.data_ptr(), HA.data_ptr(), HB.data_ptr(), partT.data_ptr(), Rpart.data_ptr())
        if not wg_fused and need_w:
            _call("apn_sa_wide_wgrad", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
                  tmap.data_ptr(), pack1.data_ptr(), goa.data_ptr(), ksel.data_ptr(), splits, Rpart.data_ptr())
        # (no weight needs a gradient: the weight-gradient pass and its fold are skipped, dL/dW2 below is unused)
        R = _colsum(Rpart) if need_w else torch.zeros(rows * H + H, dtype=torch.float64, device=dev)
        sT = reduced(partT) if sync else None
        g_ws = g_bs = None
        if fusedd:
            # BatchNorm-1 backward constants and dL/dW2 = R_S + D2 (W2 Gram) + E2 (x) suma: one launch
            g_w2 = torch.empty(O, H, **f32)
            _call("apn_sa_wide_bwd_fin", dev, None if sync else partT.data_ptr(), grid, _fz._ptr(sT), H, O,
                  pack1.data_ptr(), count, 1 if tr1 else 0, cabc.data_ptr(), g_gamma1.data_ptr(), g_beta1.data_ptr(),
                  R.data_ptr(), d2e2.data_ptr(), W2.data_ptr(), g_w2.data_ptr())
            # per point: dL/dU (rows summed through the inverse map, fixed order), dL/df, dL/dp; per query
            # dL/dnew_p; the workgroups' shares of dL/dW1 (and of the residual branch's dL/dws, dL/dbs): one launch
            Os = O if Ws is not None else 0
            wrows, wcols = lib.apn_sa_wide_point_grads_rows(B, N), lib.apn_sa_wide_point_grads_cols(C, H, Os)
            g_f = torch.empty(B, C, N, **f32)
            g_p = torch.empty(B, N, 3, **f32) if need_p else None
            g_q = torch.empty(B, M, 3, **f32) if need_q else None
            Wpart = torch.empty(wrows, wcols, **f32) if need_w else None     # (no weight takes a gradient: no shares)
            _call("apn_sa_wide_point_grads", dev, B, C, N, M, H, float(radius), GU.data_ptr(), nbr.pcnt_poff.data_ptr(),
                  nbr.plist.data_ptr(), nbr.geo.data_ptr(), U.data_ptr(), f.data_ptr(), p.data_ptr(), new_p.data_ptr(),
                  HA.data_ptr(), HB.data_ptr(), cabc.data_ptr(), pack1.data_ptr(), W1.data_ptr(), Os, _fz._ptr(gpre),
                  _fz._ptr(nbr.fq if Os else None), _fz._ptr(fs), _fz._ptr(Ws), g_f.data_ptr(),
                  _fz._ptr(g_p), _fz._ptr(g_q), _fz._ptr(Wpart))
            g_w1 = None
            if need_w:
                wsum = torch.empty(wcols, **f32)
                _call("apn_sa_wide_colsum_f32", dev, Wpart.data_ptr(), wrows, wcols, wsum.data_ptr())
                g_w1 = wsum[:H * (C + 3)].view(H, C + 3, 1, 1)
                if Os:
                    g_ws = wsum[H * (C + 3):H * (C + 3) + O * C].view(O, C, 1)
                    g_bs = wsum[H * (C + 3) + O * C:] if a5 else None
        else:
            D2, E2 = d2e2[:O], d2e2[O:]
            Rm, suma = R[:rows * H].view(rows, H), R[rows * H:]
            g_w2 = ((Rm[:O] + D2.double()[:, None] * (W2.double() @ Rm[O:]) + E2.double()[:, None] * suma[None, :]).float()
                    if need_w else None)
            _call("apn_sa_wide_consts1", dev, None if sync else partT.data_ptr(), grid, _fz._ptr(sT), H,
                  pack1.data_ptr(), count, 1 if tr1 else 0, cabc.data_ptr(), g_gamma1.data_ptr(), g_beta1.data_ptr())
            G = torch.empty(B, N, H, **f32)
            _call("apn_sa_wide_point_terms", dev, B, N, M, H, cabc.data_ptr(), pack1.data_ptr(), U.data_ptr(),
                  nbr.geo.data_ptr(), W1.data_ptr(), C + 3, float(radius), GU.data_ptr(), nbr.pcnt_poff.data_ptr(),
                  nbr.plist.data_ptr(), G.data_ptr(), HA.data_ptr(), HB.data_ptr())
            Hq = HA
            W1p = W1[:, :3]
            # dL/df[b] (C x N) = W1f^T G[b]^T and dL/dW1 = sum_b G[b]^T X[b]^T on the contraction kernel (channels-first
            # results without a transposed copy; the PyTorch forms were five GEMMs of 25-50 us, three of them 3 wide)
            g_f = torch.empty(B, C, N, **f32)
            pointwise.contract(B, C, N, H, W1[:, 3:], 0, C + 3, False, G, N * H, H, True, g_f, d_batch=C * N, ldd=N)
            g_p = g_q = None
            if need_p or need_q:        # (B, n, H) x (H, 3): as PyTorch GEMMs 40 us each
                W1pr = W1p * (1.0 / radius)
                if need_p:
                    g_p = torch.empty(B, N, 3, **f32)
                    pointwise.contract(B, N, 3, H, G, N * H, H, True, W1pr, 0, 3, False, g_p, d_batch=N * 3, ldd=3)
                if need_q:
                    g_q = torch.empty(B, M, 3, **f32)
                    pointwise.contract(B, M, 3, H, Hq, M * H, H, True, W1pr, 0, 3, False, g_q, d_batch=M * 3, ldd=3)
                    g_q.neg_()
            g_w1 = None
            if need_w:
                g_w1 = torch.empty(H, C + 3, **f32)
                pointwise.contract(B, H, C + 3, N, G, N * H, H, False, X, (C + 3) * N, N, True, g_w1, reduce=True)
                gq = torch.empty(H, 3, **f32)          # the queries' share of the coordinate columns
                pointwise.contract(B, H, 3, M, Hq, M * H, H, False, new_p, M * 3, 3, False, gq, reduce=True)
                g_w1[:, :3].sub_(gq, alpha=1.0 / radius)
                g_w1 = g_w1.view(H, C + 3, 1, 1)
        if not need_w:
            return (g_p, g_q, g_f) + (None,) * 9
        return (g_p, g_q, g_f, g_w1, g_gamma1 if a1 else None, g_beta1 if a2 else None,
                g_w2.view(O, H, 1, 1), g_gamma2 if a3 else None, g_beta2 if a4 else None, g_ws, g_bs, None)


def grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False, index=None):
    """out (B,O,M) = max_K bn2(conv2(relu(bn1(conv1(cat[(p[idx]-new_p)/r, f[idx]]))))), any PointNeXt-S width.
    index: the `NeighbourIndex` of idx built ahead of time (default: built here, on the calling stream)."""
    if index is None:
        index = neighbour_index(idx.contiguous(), new_p.detach().contiguous(), p.shape[1])
    return _WideBlock.apply(p, new_p, f, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight,
                            bn2.bias, None, None, (float(radius), bn1, bn2, sync_bn, index, False))


def block(p, new_p, f, index, radius, conv1, bn1, conv2, bn2, skip_conv=None, relu=False, sync_bn=False):
    """A whole set-abstraction block after its index stage (pointnext.py:150-168):
        out (B,O,M) = act(max_K bn2(conv2(relu(bn1(conv1(grouped))))) + skip_conv(f[:, :, fidx])),
    skip_conv: a 1x1 Conv1d on the sampled points' own features (None: no residual branch), act = ReLU when
    `relu`.  index: the NeighbourIndex of the block's neighbours, built with `fidx` when there is a skip branch.
    Only for shapes with `lean(C, H)`; wider blocks take `grouped_mlp_max` and apply the branch outside."""
    ws = bs = None
    if skip_conv is not None:
        assert index.fq is not None, "the residual branch needs the NeighbourIndex built with fidx"
        ws, bs = skip_conv.weight, skip_conv.bias
    return _WideBlock.apply(p, new_p, f, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias,
                            ws, bs, (float(radius), bn1, bn2, sync_bn, index, bool(relu)))
