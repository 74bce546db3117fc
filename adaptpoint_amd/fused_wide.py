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

from . import _lib
from . import fused as _fz
from .fused import _call

K_NS = 32
WIDTHS = (32, 64, 128, 256)
_DEBUG = None        # tests may set a dict: intermediates of the last call are stashed in it


def supported(p, f, idx_or_k, conv1, conv2, bns=()):
    k = idx_or_k.shape[2] if torch.is_tensor(idx_or_k) else int(idx_or_k)
    H, O = conv1.weight.shape[0], conv2.weight.shape[0]
    return (f.is_cuda and f.dtype == torch.float32 and p.dtype == torch.float32 and k == K_NS
            and H in WIDTHS and O == 2 * H and conv1.weight.shape[1] == f.shape[1] + 3
            and conv2.weight.shape[1] == H and conv1.bias is None and conv2.bias is None
            and p.shape[0] * (idx_or_k.shape[1] if torch.is_tensor(idx_or_k) else 1) < 2 ** 24
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


def _stats_to_pack(sums, count, bn, C, training, sync):
    """{sum[C], sumsq[C]} (float64) -> pack {scale, shift, mean, invstd}[C] (float32 (4C,)), with the
    running-buffer update of torch.nn.BatchNorm when training; the running buffers when not."""
    dev = bn.weight.device if bn.weight is not None else bn.running_mean.device
    if training:
        if sync:
            vec = torch.cat([sums, torch.full((1,), float(count), dtype=torch.float64, device=dev),
                             torch.ones(1, dtype=torch.float64, device=dev)])
            _fz._allreduce_sum_(vec)
            sums, count_t = vec[:2 * C], vec[2 * C]
        else:
            count_t = torch.full((), float(count), dtype=torch.float64, device=dev)   # (a fill: graph-capturable)
        mean = sums[:C] / count_t
        var = (sums[C:] / count_t - mean * mean).clamp_min(0.0)
        if bn.track_running_stats and bn.training:
            with torch.no_grad():
                bn.num_batches_tracked += 1
                mom = bn.momentum
                unbiased = var * (count_t / (count_t - 1).clamp_min(1.0))
                bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
                bn.running_var.mul_(1 - mom).add_(unbiased.to(bn.running_var.dtype), alpha=mom)
    else:
        mean, var = bn.running_mean.double(), bn.running_var.double()
        count_t = torch.full((), float(count), dtype=torch.float64, device=dev)
    inv = torch.rsqrt(var + bn.eps)
    gamma = bn.weight.detach().double() if bn.weight is not None else torch.ones_like(mean)
    beta = bn.bias.detach().double() if bn.bias is not None else torch.zeros_like(mean)
    scale = gamma * inv
    return torch.cat([scale, beta - mean * scale, mean, inv]).float().contiguous(), count_t


def _training(bn):
    return bn.training or not bn.track_running_stats


class _WideMlpMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, new_p, f, idx, w1, g1, b1, w2, g2, b2, mods):
        radius, bn1, bn2, sync_bn = mods
        p, new_p, f, idx = p.contiguous(), new_p.contiguous(), f.contiguous(), idx.contiguous()
        dev = f.device
        B, C, N = f.shape
        M = new_p.shape[1]
        H, O = w1.shape[0], w2.shape[0]
        sync = sync_bn and (_fz._world(True) > 1 or _fz.FORCE_PHASED)
        with torch.no_grad():
            W1 = w1.detach().reshape(H, C + 3)
            W1p, W1f = W1[:, :3], W1[:, 3:]
            W2 = w2.detach().reshape(O, H)
            # conv1 at the points: one row per support point, one per query
            U = torch.baddbmm(torch.matmul(p, W1p.t()) / radius, f.transpose(1, 2), W1f.t().expand(B, C, H))
            V = (torch.matmul(new_p, W1p.t()) / radius).contiguous()
            U = U.contiguous()
            grid = _lib.load().apn_sa_wide_grid(B, M)
            count = float(B * M * K_NS)
            tr1, tr2 = _training(bn1), _training(bn2)
            sums1 = None
            if tr1:
                part1 = torch.empty(grid, 2 * H, dtype=torch.float32, device=dev)
                _call("apn_sa_wide_stats1", dev, B, N, M, H, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
                      part1.data_ptr())
                sums1 = _colsum(part1)
            pack1, cnt1 = _stats_to_pack(sums1, count, bn1, H, tr1, sync)
            sgn2 = (torch.where(g2.detach() >= 0, 1.0, -1.0).float() if g2 is not None
                    else torch.ones(O, device=dev))
            w2img = mfma_b_image(W2.t(), min(4, O // 32))
            ysel = torch.empty(B, M, O, dtype=torch.float32, device=dev)
            ksel = torch.empty(B, M, O, dtype=torch.uint8, device=dev)
            part2 = torch.empty(grid, 2 * O, dtype=torch.float32, device=dev)
            _call("apn_sa_wide_fwd_main", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
                  w2img.data_ptr(), pack1.data_ptr(), sgn2.data_ptr(), ysel.data_ptr(), ksel.data_ptr(),
                  part2.data_ptr())
            pack2, cnt2 = _stats_to_pack(_colsum(part2) if tr2 else None, count, bn2, O, tr2, sync)
            out = torch.addcmul(pack2[O:2 * O], ysel, pack2[:O]).transpose(1, 2).contiguous()
            if _DEBUG is not None:
                _DEBUG.update(U=U, V=V, pack1=pack1, w2img=w2img, ysel=ysel, ksel=ksel, part2=part2, pack2=pack2,
                              sgn2=sgn2, sums1=sums1)
        ctx.save_for_backward(p, new_p, f, idx, U, V, pack1, pack2, ysel, ksel, W1, W2)
        ctx.cfg = (radius, tr1, tr2, sync, cnt1, cnt2, g1 is not None, b1 is not None, g2 is not None,
                   b2 is not None)
        ctx.need = (p.requires_grad, new_p.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        p, new_p, f, idx, U, V, pack1, pack2, ysel, ksel, W1, W2 = ctx.saved_tensors
        radius, tr1, tr2, sync, cnt1, cnt2, a1, a2, a3, a4 = ctx.cfg
        need_p, need_q = ctx.need
        dev = f.device
        B, C, N = f.shape
        M = new_p.shape[1]
        H, O = W1.shape[0], W2.shape[0]
        W1p, W1f = W1[:, :3], W1[:, 3:]
        lib = _lib.load()
        gz = g.transpose(1, 2).contiguous().float()                        # (B,M,O)
        scale2, mean2, inv2 = pack2[:O], pack2[2 * O:3 * O], pack2[3 * O:]
        yh_sel = (ysel - mean2) * inv2
        # BatchNorm-2 backward: only the pooled positions carry upstream gradient
        s = torch.stack([_colsum(gz.view(B * M, O)), _colsum((gz * yh_sel).view(B * M, O))])       # (2,O)
        world = 1.0
        if sync:
            vec = torch.cat([s.reshape(-1), torch.ones(1, dtype=torch.float64, device=dev)])
            _fz._allreduce_sum_(vec)
            s, world = vec[:2 * O].view(2, O), vec[2 * O]
        g_gamma2, g_beta2 = (s[1] / world).float(), (s[0] / world).float()
        if tr2:
            D2 = -scale2.double() * inv2.double() * s[1] / cnt2
            E2 = -scale2.double() * s[0] / cnt2 + scale2.double() * mean2.double() * inv2.double() * s[1] / cnt2
        else:
            D2 = torch.zeros(O, dtype=torch.float64, device=dev)
            E2 = torch.zeros(O, dtype=torch.float64, device=dev)
        goa = (gz * scale2).contiguous()
        W2d = W2.double()
        Qm = (W2d.t() * D2) @ W2d                                            # W2^T diag(D2) W2  (H,H)
        evec = (E2 @ W2d).float().contiguous()
        zimg = mfma_b_image(torch.cat([W2, Qm.float()], 0), min(4, H // 32))
        grid = lib.apn_sa_wide_grid(B, M)
        A = torch.zeros(B, N, H, dtype=torch.float32, device=dev)
        HA = torch.empty(B, M, H, dtype=torch.float32, device=dev)
        HB = torch.empty(B, M, H, dtype=torch.float32, device=dev)
        partT = torch.empty(grid, 2 * H, dtype=torch.float32, device=dev)
        _call("apn_sa_wide_bwd_main", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
              zimg.data_ptr(), pack1.data_ptr(), evec.data_ptr(), goa.data_ptr(), ksel.data_ptr(),
              A.data_ptr(), HA.data_ptr(), HB.data_ptr(), partT.data_ptr())
        # weight-gradient products over the positions
        rows = O + H
        groups = (rows // 32 + 7) // 8
        splits = max(1, min(512 // groups, (B * M) // 4, (64 << 20) // (rows * H * 4)))
        Rpart = torch.empty(splits, rows, H, dtype=torch.float32, device=dev)
        sumapart = torch.empty(splits, H, dtype=torch.float32, device=dev)
        _call("apn_sa_wide_wgrad", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(),
              pack1.data_ptr(), goa.data_ptr(), ksel.data_ptr(), splits, Rpart.data_ptr(),
              sumapart.data_ptr())
        R = _colsum(Rpart.view(splits, rows * H)).view(rows, H)
        suma = _colsum(sumapart)
        if _DEBUG is not None:
            _DEBUG.update(goa=goa, zimg=zimg, A=A, HA=HA, HB=HB, partT=partT, Rpart=Rpart, sumapart=sumapart, R=R,
                          suma=suma, D2=D2, E2=E2, s=s, gz=gz, evec=evec)
        g_w2 = (R[:O] + D2[:, None] * (W2d @ R[O:]) + E2[:, None] * suma[None, :]).float()
        # BatchNorm-1 backward constants
        T = _colsum(partT)
        if sync:
            vec = torch.cat([T, torch.ones(1, dtype=torch.float64, device=dev)])
            _fz._allreduce_sum_(vec)
            T_glob = vec[:2 * H]
        else:
            T_glob = T
        g_gamma1, g_beta1 = (T_glob[H:] / world).float(), (T_glob[:H] / world).float()
        scale1, mean1, inv1 = pack1[:H], pack1[2 * H:3 * H], pack1[3 * H:]
        ca = scale1
        if tr1:
            cb = (-scale1.double() * T_glob[H:] / cnt1).float()
            cc = (-scale1.double() * T_glob[:H] / cnt1).float()
        else:
            cb = torch.zeros_like(scale1)
            cc = torch.zeros_like(scale1)
        # how often, and from which queries, every point is gathered (coordinates only)
        flat = idx.view(B, M * K_NS).long()
        occ = torch.zeros(B, N, dtype=torch.float32, device=dev).scatter_add_(
            1, flat, torch.ones(B, M * K_NS, dtype=torch.float32, device=dev))
        SP = torch.zeros(B, N, 3, dtype=torch.float32, device=dev).scatter_add_(
            1, flat.unsqueeze(-1).expand(-1, -1, 3), new_p.repeat_interleave(K_NS, dim=1))
        # dL/dU per point and -dL/dV per query: dL/dy1 = ca g_u + cb yhat1 + cc summed over the positions
        yh_pts = inv1 * (occ.unsqueeze(-1) * (U - mean1) - torch.matmul(SP, W1p.t()) / radius)
        G = ca * A + cb * yh_pts + cc * occ.unsqueeze(-1)                    # (B,N,H)
        Hq = ca * HA + cb * HB + float(K_NS) * cc                            # (B,M,H)
        g_f = torch.matmul(G, W1f).transpose(1, 2).contiguous()              # (B,C,N)
        g_p = torch.matmul(G, W1p) / radius if need_p else None
        g_q = -torch.matmul(Hq, W1p) / radius if need_q else None
        g_w1f = torch.einsum('bnh,bcn->hc', G, f)
        g_w1p = (torch.einsum('bnh,bnd->hd', G, p) - torch.einsum('bmh,bmd->hd', Hq, new_p)) / radius
        g_w1 = torch.cat([g_w1p, g_w1f], 1).view(H, C + 3, 1, 1)
        return (g_p, g_q, g_f, None, g_w1, g_gamma1 if a1 else None, g_beta1 if a2 else None,
                g_w2.view(O, H, 1, 1), g_gamma2 if a3 else None, g_beta2 if a4 else None, None)


def grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False):
    """out (B,O,M) = max_K bn2(conv2(relu(bn1(conv1(cat[(p[idx]-new_p)/r, f[idx]]))))), any PointNeXt-S width."""
    return _WideMlpMax.apply(p, new_p, f, idx, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight,
                             bn2.bias, (float(radius), bn1, bn2, sync_bn))
