"""Fused grouped shared-MLP of a set-abstraction block (csrc/sa_fused.hip).

`grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)` computes, for the
shapes the fused kernels support,

    dp  = (p[idx] - new_p) / radius ; fj = f[idx]                 group.py:248-254
    y1  = conv1(cat[dp, fj]) ; a1 = relu(bn1(y1))                 pointnext.py:119-128,166
    y2  = conv2(a1) ; out = max_K bn2(y2)                         pointnext.py:166

without materialising any (B, C, M, K) tensor.  BatchNorm runs in training mode
(batch statistics, running buffers updated) exactly when the modules are in
training mode; the MFMA contraction is bf16 x bf16 -> f32.
"""
import torch
import torch.distributed as dist

from . import _lib


def supported(p, f, idx, conv1, conv2):
    return (f.is_cuda and f.dtype == torch.float32 and f.shape[1] == 32 and idx.shape[2] == 32
            and conv1.weight.shape[:2] == (32, 35) and conv2.weight.shape[:2] == (64, 32)
            and conv1.bias is None and conv2.bias is None)


def _call(name, dev, *args):
    lib = _lib.load()
    with torch.cuda.device(dev):
        code = getattr(lib, name)(*args, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(code, name)


def _allreduce_(t, sync):
    if sync and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t


def _bn_fold(sum_, sumsq, count, bn, training):
    """Per-channel (scale, shift, mean, invstd) of a BatchNorm given batch sums."""
    if training or not bn.track_running_stats:
        mean = sum_ / count
        var = (sumsq / count - mean * mean).clamp_min_(0.0)
        if training and bn.track_running_stats:
            with torch.no_grad():
                mom = bn.momentum if bn.momentum is not None else 0.1
                unbiased = var * (count / max(count - 1.0, 1.0))
                bn.running_mean.mul_(1 - mom).add_(mean.float(), alpha=mom)
                bn.running_var.mul_(1 - mom).add_(unbiased.float(), alpha=mom)
                bn.num_batches_tracked += 1
    else:
        mean, var = bn.running_mean.double(), bn.running_var.double()
    invstd = torch.rsqrt(var + bn.eps)
    gamma = bn.weight.double() if bn.weight is not None else torch.ones_like(mean)
    beta = bn.bias.double() if bn.bias is not None else torch.zeros_like(mean)
    scale = gamma * invstd
    shift = beta - mean * scale
    return scale, shift, mean, invstd


class FusedForward:
    """Forward of the fused chain; returns what the backward needs as well."""

    def __init__(self, p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False):
        dev = f.device
        B, C, N = f.shape
        M = new_p.shape[1]
        K = idx.shape[2]
        self.dims = (B, N, M, C, 32, 64, K)
        self.radius = float(radius)
        lib = _lib.load()
        rows = lib.apn_sa_grid_blocks(B, M)
        w1 = conv1.weight.detach().reshape(32, 35).contiguous()
        w2 = conv2.weight.detach().reshape(64, 32).contiguous()
        ft = torch.empty(B, N, C, dtype=torch.bfloat16, device=dev)
        _call("apn_sa_prep_features", dev, B, C, N, f.data_ptr(), ft.data_ptr())
        hdr = (B, N, M, C, 32, 64, K, self.radius, p.data_ptr(), new_p.data_ptr(), ft.data_ptr(),
               idx.data_ptr(), w1.data_ptr())
        training1 = bn1.training
        count = float(B * M * K)
        if training1:
            part1 = torch.empty(rows, 64, dtype=torch.float32, device=dev)
            _call("apn_sa_fwd_stats1", dev, *hdr, part1.data_ptr())
            s = _allreduce_(part1.double().sum(0), sync_bn)
            if sync_bn and dist.is_initialized():
                count *= dist.get_world_size()
            sum1, sq1 = s[:32], s[32:]
        else:
            sum1 = sq1 = None
        scale1, shift1, mean1, inv1 = _bn_fold(sum1, sq1, count, bn1, training1)
        gamma2 = bn2.weight.detach() if bn2.weight is not None else torch.ones(64, device=dev)
        sgn2 = torch.where(gamma2 >= 0, 1.0, -1.0).float().contiguous()
        ysel = torch.empty(B, M, 64, dtype=torch.float32, device=dev)
        ksel = torch.empty(B, M, 64, dtype=torch.uint8, device=dev)
        part2 = torch.empty(rows, 128, dtype=torch.float32, device=dev)
        sc1f, sh1f = scale1.float().contiguous(), shift1.float().contiguous()
        _call("apn_sa_fwd_main", dev, *hdr, w2.data_ptr(), sc1f.data_ptr(), sh1f.data_ptr(),
              sgn2.data_ptr(), ysel.data_ptr(), ksel.data_ptr(), part2.data_ptr())
        training2 = bn2.training
        if training2:
            s = _allreduce_(part2.double().sum(0), sync_bn)
            sum2, sq2 = s[:64], s[64:]
        else:
            sum2 = sq2 = None
        scale2, shift2, mean2, inv2 = _bn_fold(sum2, sq2, count, bn2, training2)
        # max_K bn2(y2) = scale2 * ext_K(y2) + shift2  (ext = max where gamma2 >= 0, else min)
        self.out = (ysel * scale2.float() + shift2.float()).transpose(1, 2).contiguous()  # (B,64,M)
        self.saved = dict(ft=ft, w1=w1, w2=w2, scale1=sc1f, shift1=sh1f, mean1=mean1, inv1=inv1,
                          mean2=mean2, inv2=inv2, scale2=scale2, ysel=ysel, ksel=ksel, count=count)


class GroupedMlpMax(torch.autograd.Function):
    """autograd wrapper: out (B,64,M) = max_K bn2(conv2(relu(bn1(conv1(cat[dp, f[idx]])))))."""

    @staticmethod
    def forward(ctx, p, new_p, f, idx, radius, w1, g1, b1, w2, g2, b2, conv1, bn1, conv2, bn2,
                sync_bn):
        fw = FusedForward(p.contiguous(), new_p.contiguous(), f.contiguous(), idx.contiguous(),
                          radius, conv1, bn1, conv2, bn2, sync_bn)
        ctx.fw = fw
        ctx.tensors = (p, new_p, idx)
        ctx.bn = (bn1, bn2)
        ctx.sync_bn = sync_bn
        ctx.need_xyz_grad = p.requires_grad or new_p.requires_grad
        return fw.out

    @staticmethod
    def backward(ctx, g_out):
        fw, sv = ctx.fw, ctx.fw.saved
        p, new_p, idx = ctx.tensors
        bn1, bn2 = ctx.bn
        B, N, M, C, C1, C2, K = fw.dims
        dev = g_out.device
        P = sv["count"]
        go = g_out.transpose(1, 2).contiguous().float()                     # (B,M,64)
        mean2, inv2, scale2 = sv["mean2"], sv["inv2"], sv["scale2"]
        yhat_sel = (sv["ysel"].double() - mean2) * inv2
        S = torch.stack([go.double().sum((0, 1)), (go.double() * yhat_sel).sum((0, 1))])
        _allreduce_(S, ctx.sync_bn)
        S1, S2 = S[0], S[1]
        w1, w2 = sv["w1"], sv["w2"]
        w2d = w2.double()
        D2 = -scale2 * inv2 * S2 / P
        E2 = -scale2 * S1 / P + scale2 * mean2 * inv2 * S2 / P
        goa = (go * scale2.float()).contiguous()
        qm = (w2d.t() @ (D2.unsqueeze(1) * w2d)).float().contiguous()      # (32,32)
        evec = (E2 @ w2d).float().contiguous()
        d2e2 = torch.stack([D2, E2]).float().contiguous()
        bn1pack = torch.stack([sv["scale1"].double(), sv["shift1"].double(), sv["mean1"],
                               sv["inv1"]]).float().contiguous()
        lib = _lib.load()
        rows = lib.apn_sa_grid_blocks(B, M)
        hdr = (B, N, M, C, C1, C2, K, fw.radius, p.data_ptr(), new_p.data_ptr(),
               sv["ft"].data_ptr(), idx.data_ptr(), w1.data_ptr(), w2.data_ptr(),
               bn1pack.data_ptr(), qm.data_ptr(), evec.data_ptr())
        part = torch.empty(rows, 64, dtype=torch.float32, device=dev)
        gw2p = torch.empty(rows, C2 * C1, dtype=torch.float32, device=dev)
        _call("apn_sa_bwd_pass1", dev, *hdr, d2e2.data_ptr(), goa.data_ptr(),
              sv["ksel"].data_ptr(), part.data_ptr(), gw2p.data_ptr())
        T = _allreduce_(part.double().sum(0), ctx.sync_bn)
        T1, T2 = T[:32], T[32:]
        sc1 = sv["scale1"].double()
        cabc = torch.stack([sc1, -sc1 * T2 / P, -sc1 * T1 / P]).float().contiguous()
        G = torch.zeros(B, N, C1, dtype=torch.float32, device=dev)
        H = torch.empty(B, M, C1, dtype=torch.float32, device=dev)
        _call("apn_sa_bwd_pass2", dev, *hdr, goa.data_ptr(), sv["ksel"].data_ptr(),
              cabc.data_ptr(), G.data_ptr(), H.data_ptr())
        # everything downstream of dL/dy1 is linear: three small GEMMs
        w1f, w1p = w1[:, 3:], w1[:, :3]
        g_f = torch.matmul(G, w1f).transpose(1, 2).contiguous()             # (B,32,N)
        g_w1f = torch.einsum('bnm,bni->mi', G, sv["ft"].float())
        Gd, Hd = G.double(), H.double()
        g_w1p = (torch.einsum('bnm,bnd->md', Gd, p.double())
                 - torch.einsum('bqm,bqd->md', Hd, new_p.double())) / fw.radius
        g_w1 = torch.cat([g_w1p.float(), g_w1f], 1).reshape(C1, C + 3, 1, 1)
        g_w2 = gw2p.sum(0).reshape(C2, C1, 1, 1)
        g_p = g_newp = None
        if ctx.need_xyz_grad:
            g_p = torch.matmul(G, w1p) / fw.radius
            g_newp = -torch.matmul(H, w1p) / fw.radius
        g_g1 = T2.float() if bn1.weight is not None else None
        g_b1 = T1.float() if bn1.bias is not None else None
        g_g2 = S2.float() if bn2.weight is not None else None
        g_b2 = S1.float() if bn2.bias is not None else None
        ctx.fw = None
        return (g_p, g_newp, g_f, None, None, g_w1, g_g1, g_b1, g_w2, g_g2, g_b2,
                None, None, None, None, None)


def grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False):
    return GroupedMlpMax.apply(p, new_p, f, idx, radius, conv1.weight, bn1.weight, bn1.bias,
                               conv2.weight, bn2.weight, bn2.bias, conv1, bn1, conv2, bn2, sync_bn)
