"""Fused grouped shared-MLP of a set-abstraction block (csrc/sa_fused.hip, sa_glue.hip).

`grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)` computes, for the
shapes the fused kernels support,

    dp  = (p[idx] - new_p) / radius ; fj = f[idx]                 group.py:248-254
    y1  = conv1(cat[dp, fj]) ; a1 = relu(bn1(y1))                 pointnext.py:119-128,166
    y2  = conv2(a1) ; out = max_K bn2(y2)                         pointnext.py:166

without materialising any (B, C, M, K) tensor, forward and backward.  BatchNorm follows
the modules' training flag (batch statistics + running-buffer update, or the running
buffers); the MFMA contractions are bf16 x bf16 -> f32, every statistic is f32 partials
summed in f64.  The whole op is a fixed sequence of kernel launches on the current
stream -- no host reads -- so it can be captured in a HIP graph.  With sync_bn=True the
per-channel sums are all-reduced across ranks (SyncBatchNorm semantics) at the three
points where statistics leave the kernels.
"""
import torch
import torch.distributed as dist

from . import _lib


def supported(p, f, idx, conv1, conv2):
    return (f.is_cuda and f.dtype == torch.float32 and f.shape[1] == 32 and idx.shape[2] == 32
            and tuple(conv1.weight.shape[:2]) == (32, 35) and tuple(conv2.weight.shape[:2]) == (64, 32)
            and conv1.bias is None and conv2.bias is None)


def _call(name, dev, *args):
    lib = _lib.load()
    with torch.cuda.device(dev):
        code = getattr(lib, name)(*args, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(code, name)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _world(sync):
    if sync and dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def _reduce(part, ncol, dev, sync):
    """float partial rows -> float64 column sums (all-reduced over ranks when sync)."""
    out = torch.empty(ncol, dtype=torch.float64, device=dev)
    _call("apn_sa_reduce_rows", dev, part.data_ptr(), part.shape[0], ncol, out.data_ptr())
    if _world(sync) > 1:
        dist.all_reduce(out)
    return out


def _fold(sums, c, count, bn, dev):
    pack = torch.empty(4, c, dtype=torch.float32, device=dev)
    training = bn.training or not bn.track_running_stats
    mom = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats
    _call("apn_sa_bn_fold", dev, _ptr(sums), c, float(count), _ptr(bn.weight), _ptr(bn.bias),
          float(bn.eps), float(mom), _ptr(bn.running_mean) if track else None,
          _ptr(bn.running_var) if track else None,
          _ptr(bn.num_batches_tracked) if (track and bn.training) else None,
          1 if training else 0, pack.data_ptr())
    return pack, training


class FusedForward:
    """Forward of the fused chain; keeps what the backward needs."""

    def __init__(self, p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False):
        dev = f.device
        B, C, N = f.shape
        M = new_p.shape[1]
        K = idx.shape[2]
        self.dims = (B, N, M, C, 32, 64, K)
        self.radius = float(radius)
        self.sync = sync_bn
        lib = _lib.load()
        rows = lib.apn_sa_grid_blocks(B, M)
        w1 = conv1.weight.detach().reshape(32, 35)
        w2 = conv2.weight.detach().reshape(64, 32)
        w1 = w1 if w1.is_contiguous() else w1.contiguous()
        w2 = w2 if w2.is_contiguous() else w2.contiguous()
        ft = torch.empty(B, N, C, dtype=torch.bfloat16, device=dev)
        _call("apn_sa_prep_features", dev, B, C, N, f.data_ptr(), ft.data_ptr())
        hdr = (B, N, M, C, 32, 64, K, self.radius, p.data_ptr(), new_p.data_ptr(), ft.data_ptr(),
               idx.data_ptr(), w1.data_ptr())
        count = float(B * M * K) * _world(sync_bn)
        sums1 = None
        if bn1.training or not bn1.track_running_stats:
            part1 = torch.empty(rows, 64, dtype=torch.float32, device=dev)
            _call("apn_sa_fwd_stats1", dev, *hdr, part1.data_ptr())
            sums1 = _reduce(part1, 64, dev, sync_bn)
        pack1, self.train1 = _fold(sums1, 32, count, bn1, dev)
        sgn2 = torch.empty(64, dtype=torch.float32, device=dev)
        _call("apn_sa_sign", dev, _ptr(bn2.weight), 64, sgn2.data_ptr())
        ysel = torch.empty(B, M, 64, dtype=torch.float32, device=dev)
        ksel = torch.empty(B, M, 64, dtype=torch.uint8, device=dev)
        part2 = torch.empty(rows, 128, dtype=torch.float32, device=dev)
        _call("apn_sa_fwd_main", dev, *hdr, w2.data_ptr(), pack1.data_ptr(),
              pack1.data_ptr() + 4 * 32, sgn2.data_ptr(), ysel.data_ptr(), ksel.data_ptr(),
              part2.data_ptr())
        sums2 = None
        if bn2.training or not bn2.track_running_stats:
            sums2 = _reduce(part2, 128, dev, sync_bn)
        pack2, self.train2 = _fold(sums2, 64, count, bn2, dev)
        # max_K bn2(y2) = scale2 * ext_K(y2) + shift2  (ext = max where gamma2 >= 0, else min)
        out = torch.empty(B, 64, M, dtype=torch.float32, device=dev)
        _call("apn_sa_fwd_out", dev, B, M, ysel.data_ptr(), pack2.data_ptr(), out.data_ptr())
        self.out = out
        self.saved = dict(ft=ft, w1=w1, w2=w2, pack1=pack1, pack2=pack2, ysel=ysel, ksel=ksel,
                          count=count)


class GroupedMlpMax(torch.autograd.Function):
    """autograd wrapper: out (B,64,M) = max_K bn2(conv2(relu(bn1(conv1(cat[dp, f[idx]])))))."""

    @staticmethod
    def forward(ctx, p, new_p, f, idx, radius, w1, g1, b1, w2, g2, b2, conv1, bn1, conv2, bn2,
                sync_bn):
        fw = FusedForward(p.contiguous(), new_p.contiguous(), f.contiguous(), idx.contiguous(),
                          radius, conv1, bn1, conv2, bn2, sync_bn)
        ctx.fw = fw
        ctx.tensors = (p.contiguous(), new_p.contiguous(), idx.contiguous())
        ctx.has_affine = (g1 is not None, b1 is not None, g2 is not None, b2 is not None)
        ctx.need_xyz_grad = (p.requires_grad, new_p.requires_grad)
        return fw.out

    @staticmethod
    def backward(ctx, g_out):
        fw, sv = ctx.fw, ctx.fw.saved
        p, new_p, idx = ctx.tensors
        B, N, M, C, C1, C2, K = fw.dims
        dev = g_out.device
        P, sync = sv["count"], fw.sync
        lib = _lib.load()
        f32 = dict(dtype=torch.float32, device=dev)
        g_out = g_out.contiguous()
        w1, w2, pack1, pack2 = sv["w1"], sv["w2"], sv["pack1"], sv["pack2"]

        # BN2 reduction terms from (B,M,64) tensors only
        goa = torch.empty(B, M, C2, **f32)
        prow = lib.apn_sa_bwd_prep_rows(B, M)
        partS = torch.empty(prow, 128, **f32)
        _call("apn_sa_bwd_prep", dev, B, M, g_out.data_ptr(), sv["ysel"].data_ptr(),
              pack2.data_ptr(), goa.data_ptr(), partS.data_ptr())
        S = _reduce(partS, 128, dev, sync)
        d2e2 = torch.empty(2, C2, **f32)
        qm = torch.empty(C1, C1, **f32)
        evec = torch.empty(C1, **f32)
        g_g2 = torch.empty(C2, **f32)
        g_b2 = torch.empty(C2, **f32)
        _call("apn_sa_bwd_consts2", dev, S.data_ptr(), pack2.data_ptr(), w2.data_ptr(), float(P),
              1 if fw.train2 else 0, d2e2.data_ptr(), qm.data_ptr(), evec.data_ptr(),
              g_g2.data_ptr(), g_b2.data_ptr())

        rows = lib.apn_sa_grid_blocks(B, M)
        hdr = (B, N, M, C, C1, C2, K, fw.radius, p.data_ptr(), new_p.data_ptr(),
               sv["ft"].data_ptr(), idx.data_ptr(), w1.data_ptr(), w2.data_ptr(),
               pack1.data_ptr(), qm.data_ptr(), evec.data_ptr())
        part = torch.empty(rows, 64, **f32)
        gw2p = torch.empty(rows, C2 * C1, **f32)
        _call("apn_sa_bwd_pass1", dev, *hdr, d2e2.data_ptr(), goa.data_ptr(),
              sv["ksel"].data_ptr(), part.data_ptr(), gw2p.data_ptr())
        T = _reduce(part, 64, dev, sync)
        gw2d = _reduce(gw2p, C2 * C1, dev, False)      # weight grads are reduced by DDP, not here
        g_w2 = torch.empty(C2, C1, 1, 1, **f32)
        _call("apn_sa_cast_d2f", dev, gw2d.data_ptr(), C2 * C1, g_w2.data_ptr())
        cabc = torch.empty(3, C1, **f32)
        g_g1 = torch.empty(C1, **f32)
        g_b1 = torch.empty(C1, **f32)
        _call("apn_sa_bwd_consts1", dev, T.data_ptr(), pack1.data_ptr(), float(P),
              1 if fw.train1 else 0, cabc.data_ptr(), g_g1.data_ptr(), g_b1.data_ptr())

        G = torch.zeros(B, N, C1, **f32)
        H = torch.empty(B, M, C1, **f32)
        _call("apn_sa_bwd_pass2", dev, *hdr, goa.data_ptr(), sv["ksel"].data_ptr(),
              cabc.data_ptr(), G.data_ptr(), H.data_ptr())

        # everything downstream of dL/dy1 is linear in G (per source point) and H (per query)
        g_f = torch.empty(B, C, N, **f32)
        need_p, need_q = ctx.need_xyz_grad
        g_p = torch.zeros(B, N, 3, **f32) if need_p else None
        g_newp = torch.empty(B, M, 3, **f32) if need_q else None
        _call("apn_sa_bwd_input_grad", dev, B, N, M, G.data_ptr(), H.data_ptr(), w1.data_ptr(),
              fw.radius, g_f.data_ptr(), _ptr(g_p), _ptr(g_newp))
        wrows = lib.apn_sa_bwd_weight_rows(B, N)
        partW = torch.empty(wrows, 32 * 38, **f32)
        _call("apn_sa_bwd_weight_grad", dev, B, N, M, G.data_ptr(), H.data_ptr(),
              sv["ft"].data_ptr(), p.data_ptr(), new_p.data_ptr(), partW.data_ptr())
        sW = _reduce(partW, 32 * 38, dev, False)
        g_w1 = torch.empty(C1, C + 3, 1, 1, **f32)
        _call("apn_sa_bwd_w1_final", dev, sW.data_ptr(), fw.radius, g_w1.data_ptr())

        a1, a2, a3, a4 = ctx.has_affine
        ctx.fw = None
        return (g_p, g_newp, g_f, None, None, g_w1, g_g1 if a1 else None, g_b1 if a2 else None,
                g_w2, g_g2 if a3 else None, g_b2 if a4 else None, None, None, None, None, None)


def grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False):
    return GroupedMlpMax.apply(p, new_p, f, idx, radius, conv1.weight, bn1.weight, bn1.bias,
                               conv2.weight, bn2.weight, bn2.bias, conv1, bn1, conv2, bn2, sync_bn)
