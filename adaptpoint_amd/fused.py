"""Fused set-abstraction block (csrc/sa_fused.hip, sa_glue.hip, fps.hip, ball_query.hip).

Two entry points over the same kernels:

`fused_set_abstraction(p, f, npoint, radius, conv1, bn1, conv2, bn2, skip_conv, relu)`
    the whole block of openpoints/models/backbone/pointnext.py:140-170 for the shapes the
    kernels support -- FPS (+ the gather of the sampled points), ball query, then

        dp  = (p[idx] - new_p) / radius ; fj = f[idx]                 group.py:248-254
        y1  = conv1(cat[dp, fj]) ; a1 = relu(bn1(y1))                 pointnext.py:119-128,166
        y2  = conv2(a1) ; o = max_K bn2(y2)                           pointnext.py:166
        out = relu(o + skip_conv(f[:, fps_idx]))                      pointnext.py:157-168

    returning (new_p, out);

`grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)`
    only the grouped MLP + pool `o`, for callers that sample / query themselves.

No (B, C, M, K) tensor is materialised, forward or backward.  BatchNorm follows the
modules' training flag (batch statistics + running-buffer update, or the running
buffers); the MFMA contractions are bf16 x bf16 -> f32 accumulate, every statistic is an
f32 partial summed in f64.  Forward is 8 kernel launches, backward 9 (+ one memset),
all on the current stream with no host reads, so a step can be captured in a HIP graph.
With sync_bn=True the per-channel float64 sums are all-reduced across ranks
(SyncBatchNorm semantics) at the four points where statistics leave the kernels.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib

C_IN, C_MID, C_OUT, K_NS = 32, 32, 64, 32


def supported(p, f, idx_or_k, conv1, conv2):
    k = idx_or_k.shape[2] if torch.is_tensor(idx_or_k) else int(idx_or_k)
    return (f.is_cuda and f.dtype == torch.float32 and p.dtype == torch.float32
            and f.shape[1] == C_IN and k == K_NS
            and tuple(conv1.weight.shape[:2]) == (C_MID, C_IN + 3)
            and tuple(conv2.weight.shape[:2]) == (C_OUT, C_MID)
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


def _rows_or_sums(part, ncol, dev, sync):
    """(part_ptr, rows, sums_ptr, keepalive): the consumer kernels sum the partial rows
    themselves; with SyncBatchNorm on they get float64 sums all-reduced over ranks."""
    if part is None:
        return None, 0, None, None
    if _world(sync) > 1:
        sums = torch.empty(ncol, dtype=torch.float64, device=dev)
        _call("apn_sa_reduce_rows", dev, part.data_ptr(), part.shape[0], ncol, sums.data_ptr())
        dist.all_reduce(sums)
        return None, 0, sums.data_ptr(), sums
    return part.data_ptr(), part.shape[0], None, part


def _fold(part, c, count, bn, dev, sync, sgn_gamma=None, sgn_out=None):
    pack = torch.empty(4, c, dtype=torch.float32, device=dev)
    training = bn.training or not bn.track_running_stats
    mom = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats
    pp, rows, sp, _keep = _rows_or_sums(part if training else None, 2 * c, dev, sync)
    _call("apn_sa_bn_fold", dev, pp, rows, sp, c, float(count),
          _ptr(bn.weight), _ptr(bn.bias), float(bn.eps), float(mom),
          _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None,
          _ptr(bn.num_batches_tracked) if (track and bn.training) else None,
          1 if training else 0, pack.data_ptr(), _ptr(sgn_gamma),
          0 if sgn_out is None else sgn_out.numel(), _ptr(sgn_out))
    return pack, training


def _mat(w, rows, cols):
    w = w.detach().reshape(rows, cols)
    return w if w.is_contiguous() else w.contiguous()


class _Forward:
    """Runs the forward launches and keeps what the backward needs."""

    def __init__(self, p, f, new_p, idx, fidx, radius, conv1, bn1, conv2, bn2, skip_conv, relu,
                 sync_bn):
        dev = f.device
        B, C, N = f.shape
        M = new_p.shape[1]
        self.dims = (B, N, M)
        self.radius = float(radius)
        self.sync = sync_bn
        self.relu = 1 if relu else 0
        w1 = _mat(conv1.weight, C_MID, C_IN + 3)
        w2 = _mat(conv2.weight, C_OUT, C_MID)
        ws = bs = None
        if skip_conv is not None:
            ws = _mat(skip_conv.weight, C_OUT, C_IN)
            bs = skip_conv.bias.detach() if skip_conv.bias is not None else None
        ft = torch.empty(B, N, C, dtype=torch.bfloat16, device=dev)
        _call("apn_sa_prep_features", dev, B, C, N, f.data_ptr(), ft.data_ptr())
        hdr = (B, N, M, C_IN, C_MID, C_OUT, K_NS, self.radius, p.data_ptr(), new_p.data_ptr(),
               ft.data_ptr(), idx.data_ptr(), w1.data_ptr())
        count = float(B * M * K_NS) * _world(sync_bn)
        rows = _lib.load().apn_sa_grid_blocks(B, M)
        part1 = None
        if bn1.training or not bn1.track_running_stats:
            part1 = torch.empty(rows, 64, dtype=torch.float32, device=dev)
            _call("apn_sa_fwd_stats1", dev, *hdr, part1.data_ptr())
        sgn2 = torch.empty(C_OUT, dtype=torch.float32, device=dev)
        pack1, self.train1 = _fold(part1, C_MID, count, bn1, dev, sync_bn, bn2.weight, sgn2)
        ysel = torch.empty(B, M, C_OUT, dtype=torch.float32, device=dev)
        ksel = torch.empty(B, M, C_OUT, dtype=torch.uint8, device=dev)
        part2 = torch.empty(rows, 128, dtype=torch.float32, device=dev)
        _call("apn_sa_fwd_main", dev, *hdr, w2.data_ptr(), pack1.data_ptr(),
              pack1.data_ptr() + 4 * C_MID, sgn2.data_ptr(), ysel.data_ptr(), ksel.data_ptr(),
              part2.data_ptr())
        pack2, self.train2 = _fold(part2, C_OUT, count, bn2, dev, sync_bn)
        # max_K bn2(y2) = scale2 * ext_K(y2) + shift2  (ext = max where gamma2 >= 0, else min)
        out = torch.empty(B, C_OUT, M, dtype=torch.float32, device=dev)
        _call("apn_sa_fwd_out", dev, B, N, M, ysel.data_ptr(), pack2.data_ptr(),
              f.data_ptr() if ws is not None else None, _ptr(fidx) if ws is not None else None,
              _ptr(ws), _ptr(bs), self.relu, out.data_ptr())
        self.out = out
        self.saved = dict(p=p, f=f, new_p=new_p, idx=idx, fidx=fidx, ft=ft, w1=w1, w2=w2, ws=ws,
                          has_bs=bs is not None, pack1=pack1, pack2=pack2, ysel=ysel, ksel=ksel,
                          count=count)


def _backward(fw, g_out, need_p, need_newp):
    """All gradients of the fused chain from g_out (B,64,M)."""
    sv = fw.saved
    B, N, M = fw.dims
    dev = g_out.device
    P, sync = sv["count"], fw.sync
    f32 = dict(dtype=torch.float32, device=dev)
    g_out = g_out.contiguous()
    w1, w2, ws, pack1, pack2 = sv["w1"], sv["w2"], sv["ws"], sv["pack1"], sv["pack2"]
    p, new_p, idx, ft = sv["p"], sv["new_p"], sv["idx"], sv["ft"]
    has_skip = ws is not None

    lib = _lib.load()
    # one zero-fill for every atomically accumulated buffer of the backward
    nf = C_OUT * C_MID + B * N * C_MID * (2 if has_skip else 1)
    accf = torch.zeros(nf, **f32)
    g_w2 = accf[:C_OUT * C_MID]
    o = C_OUT * C_MID
    G = accf[o:o + B * N * C_MID]
    o += B * N * C_MID
    gip = accf[o:o + B * N * C_MID] if has_skip else None

    goa = torch.empty(B, M, C_OUT, **f32)
    prow = lib.apn_sa_bwd_prep_rows(B, M)
    partS = torch.empty(prow, 128, **f32)
    partWs = torch.empty(prow, C_OUT * C_IN, **f32) if has_skip else None
    _call("apn_sa_bwd_prep", dev, B, N, M, g_out.data_ptr(), fw.out.data_ptr(), fw.relu,
          sv["ysel"].data_ptr(), pack2.data_ptr(), sv["f"].data_ptr() if has_skip else None,
          _ptr(sv["fidx"]) if has_skip else None, _ptr(ws), goa.data_ptr(), partS.data_ptr(),
          _ptr(partWs), _ptr(gip))
    d2e2 = torch.empty(2, C_OUT, **f32)
    qm = torch.empty(C_MID, C_MID, **f32)
    evec = torch.empty(C_MID, **f32)
    g_g2 = torch.empty(C_OUT, **f32)
    g_b2 = torch.empty(C_OUT, **f32)
    pp, nr, sp, _k1 = _rows_or_sums(partS, 128, dev, sync)
    _call("apn_sa_bwd_consts2", dev, pp, nr, sp, pack2.data_ptr(), w2.data_ptr(), float(P),
          1 if fw.train2 else 0, d2e2.data_ptr(), qm.data_ptr(), evec.data_ptr(),
          g_g2.data_ptr(), g_b2.data_ptr())
    hdr = (B, N, M, C_IN, C_MID, C_OUT, K_NS, fw.radius, p.data_ptr(), new_p.data_ptr(),
           ft.data_ptr(), idx.data_ptr(), w1.data_ptr(), w2.data_ptr(), pack1.data_ptr(),
           qm.data_ptr(), evec.data_ptr())
    rows = lib.apn_sa_grid_blocks(B, M)
    partT = torch.empty(rows, 64, **f32)
    _call("apn_sa_bwd_pass1", dev, *hdr, d2e2.data_ptr(), goa.data_ptr(), sv["ksel"].data_ptr(),
          partT.data_ptr(), g_w2.data_ptr())
    cabc = torch.empty(3, C_MID, **f32)
    g_g1 = torch.empty(C_MID, **f32)
    g_b1 = torch.empty(C_MID, **f32)
    pp, nr, sp, _k2 = _rows_or_sums(partT, 64, dev, sync)
    _call("apn_sa_bwd_consts1", dev, pp, nr, sp, pack1.data_ptr(), float(P),
          1 if fw.train1 else 0, cabc.data_ptr(), g_g1.data_ptr(), g_b1.data_ptr())
    H = torch.empty(B, M, C_MID, **f32)
    _call("apn_sa_bwd_pass2", dev, *hdr, goa.data_ptr(), sv["ksel"].data_ptr(), cabc.data_ptr(),
          G.data_ptr(), H.data_ptr())
    # everything downstream of dL/dy1 is linear in G (per source point) and H (per query)
    g_f = torch.empty(B, C_IN, N, **f32)
    g_p = torch.zeros(B, N, 3, **f32) if need_p else None
    g_newp = torch.empty(B, M, 3, **f32) if need_newp else None
    _call("apn_sa_bwd_input_grad", dev, B, N, M, G.data_ptr(), H.data_ptr(), w1.data_ptr(),
          _ptr(gip), fw.radius, g_f.data_ptr(), _ptr(g_p), _ptr(g_newp))
    wrows = lib.apn_sa_bwd_weight_rows(B, N)
    partW = torch.empty(wrows, 32 * 38, **f32)
    _call("apn_sa_bwd_weight_grad", dev, B, N, M, G.data_ptr(), H.data_ptr(), ft.data_ptr(),
          p.data_ptr(), new_p.data_ptr(), partW.data_ptr())
    g_w1 = torch.empty(C_MID, C_IN + 3, 1, 1, **f32)
    g_ws = torch.empty(C_OUT, C_IN, 1, **f32) if has_skip else None
    g_bs = torch.empty(C_OUT, **f32) if (has_skip and sv["has_bs"]) else None
    # weight gradients are per-rank sums here; DistributedDataParallel averages them
    _call("apn_sa_bwd_finalize", dev, partW.data_ptr(), wrows, fw.radius, g_w1.data_ptr(),
          _ptr(partWs), prow, _ptr(g_ws), partS.data_ptr(), _ptr(g_bs))
    return dict(f=g_f, p=g_p, new_p=g_newp, w1=g_w1, w2=g_w2.view(C_OUT, C_MID, 1, 1),
                g1=g_g1, b1=g_b1, g2=g_g2, b2=g_b2,
                ws=g_ws, bs=g_bs)


class _GroupedMlpMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, new_p, f, idx, w1, g1, b1, w2, g2, b2, mods):
        radius, conv1, bn1, conv2, bn2, sync_bn = mods
        fw = _Forward(p.contiguous(), f.contiguous(), new_p.contiguous(), idx.contiguous(), None,
                      radius, conv1, bn1, conv2, bn2, None, False, sync_bn)
        ctx.fw = fw
        ctx.flags = (p.requires_grad, new_p.requires_grad,
                     g1 is not None, b1 is not None, g2 is not None, b2 is not None)
        return fw.out

    @staticmethod
    def backward(ctx, g_out):
        need_p, need_q, a1, a2, a3, a4 = ctx.flags
        g = _backward(ctx.fw, g_out, need_p, need_q)
        ctx.fw = None
        return (g["p"], g["new_p"], g["f"], None, g["w1"], g["g1"] if a1 else None,
                g["b1"] if a2 else None, g["w2"], g["g2"] if a3 else None,
                g["b2"] if a4 else None, None)


def grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False):
    """out (B,64,M) = max_K bn2(conv2(relu(bn1(conv1(cat[(p[idx]-new_p)/r, f[idx]])))))."""
    return _GroupedMlpMax.apply(p, new_p, f, idx, conv1.weight, bn1.weight, bn1.bias,
                                conv2.weight, bn2.weight, bn2.bias,
                                (radius, conv1, bn1, conv2, bn2, sync_bn))


class _SetAbstraction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, f, w1, g1, b1, w2, g2, b2, ws, bs, mods):
        npoint, radius, conv1, bn1, conv2, bn2, skip_conv, relu, sync_bn = mods
        p = p.contiguous()
        f = f.contiguous()
        dev = f.device
        B, N, _ = p.shape
        # FPS + gather of the sampled points in one launch (pointnext.py:146-147)
        fidx = torch.empty(B, npoint, dtype=torch.int32, device=dev)
        new_p = torch.empty(B, npoint, 3, dtype=torch.float32, device=dev)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=dev)
        _call("apn_furthest_point_sampling_xyz", dev, B, N, npoint, p.data_ptr(), temp.data_ptr(),
              fidx.data_ptr(), new_p.data_ptr())
        idx = torch.empty(B, npoint, K_NS, dtype=torch.int32, device=dev)
        _call("apn_ball_query_zero", dev, B, N, npoint, float(radius), K_NS, new_p.data_ptr(),
              p.data_ptr(), idx.data_ptr())
        fw = _Forward(p, f, new_p, idx, fidx, radius, conv1, bn1, conv2, bn2, skip_conv, relu,
                      sync_bn)
        ctx.fw = fw
        ctx.flags = (p.requires_grad, g1 is not None, b1 is not None, g2 is not None,
                     b2 is not None, ws is not None, bs is not None)
        ctx.mark_non_differentiable(new_p) if not p.requires_grad else None
        return new_p, fw.out

    @staticmethod
    def backward(ctx, g_newp_in, g_out):
        need_p, a1, a2, a3, a4, a5, a6 = ctx.flags
        fw = ctx.fw
        g = _backward(fw, g_out, need_p, need_p)
        g_p = g["p"]
        if need_p:
            # new_p = p[fidx]: its gradient (from the chain and from downstream users) returns to p
            gq = g["new_p"] if g_newp_in is None else g["new_p"] + g_newp_in
            g_p = g_p.scatter_add(1, fw.saved["fidx"].long().unsqueeze(-1).expand(-1, -1, 3), gq)
        ctx.fw = None
        return (g_p, g["f"], g["w1"], g["g1"] if a1 else None, g["b1"] if a2 else None, g["w2"],
                g["g2"] if a3 else None, g["b2"] if a4 else None, g["ws"] if a5 else None,
                g["bs"] if a6 else None, None)


def fused_set_abstraction(p, f, npoint, radius, conv1, bn1, conv2, bn2, skip_conv=None, relu=True,
                          sync_bn=False):
    """(new_p (B,npoint,3), out (B,64,npoint)) of a PointNeXt set-abstraction block."""
    assert skip_conv is None or isinstance(skip_conv, nn.Conv1d)
    ws = skip_conv.weight if skip_conv is not None else None
    bs = skip_conv.bias if skip_conv is not None else None
    return _SetAbstraction.apply(p, f, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight,
                                 bn2.bias, ws, bs,
                                 (npoint, radius, conv1, bn1, conv2, bn2, skip_conv, relu, sync_bn))
