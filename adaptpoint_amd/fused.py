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
