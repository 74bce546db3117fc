"""The imitator's per-point MLP layer on the gfx950 kernels of csrc/pointwise.hip.

`ConvBNReLU1D` (openpoints/models_adaptpoint/generator_component4_15.py:92-104) is
Conv1d(kernel 1) + BatchNorm1d + ReLU on channels-first (B, C, N) tensors; the reference runs it as
three cuDNN / elementwise calls forward and five or more backward.  Here a layer is

    forward   y = W x            one MFMA contraction (operands in bf16 planes, f32 accumulate) whose epilogue
                                  leaves BatchNorm's batch statistics as partial rows,
              out = relu(bn(y))   one pass that folds those rows (float64) and updates the running statistics;
    backward  gy = dL/dy          two passes over (g, y) (BatchNorm's two sums, then the apply),
              gx = W^T gy         the same contraction kernel, W read transposed in place,
              gW = sum gy x^T     the same kernel with both operands position-contiguous, split over
                                  (cloud, position) ranges whose shares are added in a fixed order

-- 2 launches forward, 6 backward (the weight gradient's shares are folded by one more), bit-reproducible
gradients, no transposed copies.  `conv_bn_act` takes the torch modules (their parameters and buffers are used and
updated in place), so `state_dict`s stay the reference's.  Also here: `conv_bias_act` and `conv_max` (the
discriminator's per-point layers, the last one fused with its max-pool), `run_block` (a 1x1 `convblock` of the
PointNeXt mirror), `matmul_nt` / `contract` (the contraction kernel on raw operands).
"""
import torch

from . import _lib
from .fused import _call

# bf16 planes per operand of the contractions: 3 = six MFMAs per product, fp32-class (~2e-7: what a stack of ~40
# training-mode BatchNorm layers over few points needs to stay within a few 1e-3 of the reference's gradients);
# 2 = three MFMAs, ~4e-6 per contraction, as the fused set-abstraction kernels use
PRECISION = 3


def supported(x, conv, bn):
    """Shapes / modes the kernels serve: float32 CUDA, contiguous (B, C, N), kernel 1, no bias, a BatchNorm1d with a
    numeric momentum (the cumulative-average mode of momentum=None stays on PyTorch)."""
    return (_pointwise_conv(x, conv) and conv.bias is None
            and (bn.momentum is not None or not bn.training)
            and (bn.training or bn.track_running_stats))


def _pointwise_conv(x, conv):
    """x (B, C, N) float32 on the GPU and `conv` a plain 1x1 convolution (Conv1d, or Conv2d as the reference's
    grouped stages hold them: a (B, C, 1, N) tensor is the same memory)."""
    one = lambda t, v: all(k == v for k in t)
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and 0 < x.shape[0] <= 65535 and x.shape[2] > 0
            and one(conv.kernel_size, 1) and one(conv.stride, 1) and one(conv.padding, 0) and one(conv.dilation, 1)
            and conv.groups == 1 and conv.weight.dtype == torch.float32 and conv.in_channels == x.shape[1])


def run_block(x, block, allow=True):
    """One `convblock` of the PointNeXt mirror -- Sequential(conv[, BatchNorm][, ReLU]) with a 1x1 convolution -- on
    x (B, C, N): through the contraction kernels where they apply, else through the modules themselves."""
    mods = list(block)
    conv = mods[0]
    bn = mods[1] if len(mods) > 1 and isinstance(mods[1], (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)) else None
    relu = isinstance(mods[-1], torch.nn.ReLU)
    known = len(mods) == 1 + (bn is not None) + relu
    if allow and known and _pointwise_conv(x, conv):
        if bn is not None and supported(x, conv, bn):
            return _ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, bn, relu)
        if bn is None:
            return conv_bias_act(x, conv.weight, conv.bias, relu)
    if isinstance(conv, torch.nn.Conv2d):
        return block(x.unsqueeze(2)).squeeze(2)
    return block(x)


class _ConvBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, relu):
        dev = x.device
        x = x.contiguous()
        B, C, N = x.shape
        O = weight.shape[0]
        w = weight.detach().reshape(O, C).contiguous()
        training = bn.training or not bn.track_running_stats
        lib = _lib.load()
        y = torch.empty(B, O, N, device=dev)
        tiles = lib.apn_pw_conv_tiles(B, N)
        part = torch.empty(tiles, 2, O, device=dev) if training else None
        _call("apn_pw_conv_forward", dev, B, C, O, N, PRECISION, x.data_ptr(), w.data_ptr(), y.data_ptr(),
              part.data_ptr() if training else None)
        out = torch.empty_like(y)
        stat = torch.empty(4, O, device=dev)
        track = bn.track_running_stats and bn.running_mean is not None
        _call("apn_pw_bn_act", dev, B, O, N, y.data_ptr(), part.data_ptr() if training else None, tiles,
              gamma.data_ptr() if gamma is not None else None, beta.data_ptr() if beta is not None else None,
              float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.0), int(training), int(relu),
              bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
              bn.num_batches_tracked.data_ptr() if (track and training and bn.num_batches_tracked is not None) else None,
              stat.data_ptr(), out.data_ptr())
        ctx.save_for_backward(x, w, y, stat)
        ctx.cfg = (training, relu, gamma is not None, beta is not None, weight.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, y, stat = ctx.saved_tensors
        training, relu, has_gamma, has_beta, wshape = ctx.cfg
        dev = x.device
        B, C, N = x.shape
        O = w.shape[0]
        lib = _lib.load()
        g = g.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        part_b = torch.empty(lib.apn_pw_bn_act_grad_splits(B, O), 2, O, device=dev)
        gy = torch.empty_like(y)
        ggb = torch.empty(2, O, device=dev)
        _call("apn_pw_bn_act_grad", dev, B, O, N, g.data_ptr(), y.data_ptr(), stat.data_ptr(), int(training), int(relu),
              part_b.data_ptr(), gy.data_ptr(), ggb[0].data_ptr(), ggb[1].data_ptr())
        gx = gw = None
        if need_x:
            gx = torch.empty_like(x)
            _call("apn_pw_conv_grad_input", dev, B, C, O, N, PRECISION, gy.data_ptr(), w.data_ptr(), gx.data_ptr())
        if need_w:
            splits = lib.apn_pw_conv_grad_weight_splits(B, C, O, N)
            scratch = torch.empty(splits, O, C, device=dev)
            gw = torch.empty(O, C, device=dev)
            _call("apn_pw_conv_grad_weight", dev, B, C, O, N, PRECISION, gy.data_ptr(), x.data_ptr(), scratch.data_ptr(),
                  gw.data_ptr())
            gw = gw.view(wshape)
        return gx, gw, (ggb[0] if has_gamma else None), (ggb[1] if has_beta else None), None, None


class _NoNorm:
    """What `_ConvBNAct` reads of a BatchNorm module, set so that the normalisation is the identity in eval mode
    (mean 0, variance 1, eps 0, no scale): the layer is then convolution + bias (+ ReLU)."""
    training = False
    track_running_stats = True
    momentum = 0.0
    eps = 0.0
    num_batches_tracked = None
    _cache = {}

    def __init__(self, channels, device):
        key = (channels, str(device))
        if key not in _NoNorm._cache:
            _NoNorm._cache[key] = (torch.zeros(channels, device=device), torch.ones(channels, device=device))
        self.running_mean, self.running_var = _NoNorm._cache[key]


def conv_bias_act(x, weight, bias=None, relu=True):
    """[relu](weight x + bias) for x (B, C, N) and a kernel-1 weight (O, C[, 1[, 1]]) on the contraction kernels
    (the discriminator's per-point layers, point_discriminator.py:183-187, which carry no BatchNorm)."""
    return _ConvBNAct.apply(x, weight, None, bias, _NoNorm(weight.shape[0], x.device), relu)


def conv_bn_act(x, conv, bn, relu=True):
    """relu(bn(conv(x))) for a kernel-1 bias-free `conv` (nn.Conv1d) and `bn` (nn.BatchNorm1d) on x (B, C, N)."""
    return _ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, bn, relu)


class _ConvMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        dev = x.device
        x = x.contiguous()
        B, C, N = x.shape
        O = weight.shape[0]
        w = weight.detach().reshape(O, C).contiguous()
        tiles = _lib.load().apn_pw_conv_max_tiles(N)
        tile_val = torch.empty(B, tiles, O, device=dev)
        tile_idx = torch.empty(B, tiles, O, dtype=torch.int32, device=dev)
        out = torch.empty(B, O, device=dev)
        idx = torch.empty(B, O, dtype=torch.int32, device=dev)
        _call("apn_pw_conv_max_forward", dev, B, C, O, N, PRECISION, x.data_ptr(), w.data_ptr(),
              bias.data_ptr() if bias is not None else None, int(relu), tile_val.data_ptr(), tile_idx.data_ptr(),
              out.data_ptr(), idx.data_ptr())
        ctx.save_for_backward(x, w, out, idx)
        ctx.cfg = (relu, bias is not None, weight.shape)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, g, _):
        x, w, out, idx = ctx.saved_tensors
        relu, has_bias, wshape = ctx.cfg
        dev = x.device
        B, C, N = x.shape
        O = w.shape[0]
        g = g.contiguous()
        xsel = torch.empty(B, O, C, device=dev)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty(O, C, device=dev)
        gb = torch.empty(O, device=dev) if has_bias else None
        _call("apn_pw_conv_max_backward", dev, B, C, O, N, g.data_ptr(), out.data_ptr(), idx.data_ptr(), x.data_ptr(),
              w.data_ptr(), int(relu), xsel.data_ptr(), gx.data_ptr() if gx is not None else None, gw.data_ptr(),
              gb.data_ptr() if gb is not None else None)
        return gx, gw.view(wshape), gb, None


def conv_max_supported(x, c_in):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and 0 < x.shape[0] <= 65535 and x.shape[2] > 0
            and c_in == x.shape[1] and c_in <= 128)


def conv_max(x, weight, bias=None, relu=True):
    """(B, O) = [relu](max over the N points of (weight x + bias)) for x (B, C, N), weight (O, C[, 1, 1]): the last
    layer of the discriminator's group-all stage with its pooling (point_discriminator.py:183-189), without the
    (B, O, N) activation.  Also returns nothing else: the arg-max positions stay inside the autograd node."""
    return _ConvMax.apply(x, weight, bias, relu)[0]


def conv_then_bn(x, block, allow=True):
    """Sequential(Conv1d(k = 1[, bias]), BatchNorm1d) on x (B, C, N).  Without a convolution bias the whole block runs on
    the contraction kernels (`_ConvBNAct`); with one, the convolution + bias does (its weight gradient is a split-K
    contraction instead of MIOpen's `igemm_wrw`, 130 us for 3 -> 64 channels over 32 x 1024 points) and the BatchNorm
    module follows on PyTorch -- the bias is NOT dropped although a training-mode BatchNorm cancels it: it still moves
    the running mean, matters in eval mode, and its (numerically tiny) gradient is what the reference hands Adam."""
    conv, bn = block[0], block[1]
    plain = not (block._forward_pre_hooks or conv._forward_hooks or conv._forward_pre_hooks or bn._forward_pre_hooks)
    if allow and plain and len(block) == 2 and _pointwise_conv(x, conv):
        out = None
        if conv.bias is None and supported(x, conv, bn) and not bn._forward_hooks:
            out = _ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, bn, False)
        elif conv.bias is not None:
            out = bn(conv_bias_act(x, conv.weight, conv.bias, relu=False))
        if out is not None:
            for hook in block._forward_hooks.values():      # the block's own forward hooks see the fused result
                res = hook(block, (x,), out)
                if res is not None:
                    out = res
            return out
    return block(x)


class _LinearNoBias(torch.autograd.Function):
    """y (P, O) = x (P, C) W^T with the weight gradient on the contraction kernel's split-K form: W.grad (O, C) =
    g^T x contracts over the P rows (32768 of them for the imitator's `to_qkv`): the GEMM library ran that product as
    a handful of workgroups (134 us); the forward and the input gradient are well-shaped library GEMMs."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return x @ weight.t()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            P, C = x.shape
            O = weight.shape[0]
            gw = torch.empty(O, C, device=x.device)
            contract(1, O, C, P, g, 0, O, False, x, 0, C, False, gw, reduce=True)
        return gx, gw


def linear_nobias(x, linear):
    """`linear(x)` for a bias-free nn.Linear and x (..., C) float32 on the GPU (else the module itself)."""
    if (linear.bias is not None or not x.is_cuda or x.dtype != torch.float32 or x.shape[-1] != linear.in_features
            or x.numel() // x.shape[-1] < 4096):
        return linear(x)
    lead = x.shape[:-1]
    return _LinearNoBias.apply(x.reshape(-1, x.shape[-1]).contiguous(), linear.weight).reshape(*lead, linear.out_features)


def matmul_nt(a, b):
    """a (R, K) @ b (Q, K)^T -> (R, Q) on the contraction kernel in its split-K form (both operands contiguous along
    the contracted index; fixed-order fold).  For the small products of the fused blocks' backward passes whose
    shapes PyTorch's GEMM library serves badly (a 256 x 512 x 256 product ran as ONE workgroup, 122 us)."""
    a, b = a.contiguous(), b.contiguous()
    R, K = a.shape
    Q = b.shape[0]
    lib = _lib.load()
    splits = lib.apn_pw_conv_grad_weight_splits(1, Q, R, K)
    scratch = torch.empty(splits, R, Q, device=a.device)
    out = torch.empty(R, Q, device=a.device)
    _call("apn_pw_conv_grad_weight", a.device, 1, Q, R, K, PRECISION, a.data_ptr(), b.data_ptr(), scratch.data_ptr(),
          out.data_ptr())
    return out


def contract(nbatch, R, Q, K, a, a_batch, lda, a_kcont, b, b_batch, ldb, b_kcont, out, d_batch=0, ldd=None, reduce=False):
    """csrc/pointwise.hip's contraction kernel on raw operand descriptions (include/adaptpoint_amd.h,
    apn_pw_contract): out[z] = a[z] b[z] per batch entry, or (reduce) out = sum_z a[z] b[z] in fixed-order shares.
    a / b: tensors whose data_ptr() is the operand's first element (a storage offset is a column offset)."""
    lib = _lib.load()
    splits, scratch = 0, None
    if reduce:
        splits = lib.apn_pw_contract_splits(nbatch, R, Q, K)
        scratch = torch.empty(splits, R, Q, device=out.device)
    _call("apn_pw_contract", out.device, nbatch, R, Q, K, a.data_ptr(), a_batch, lda, int(a_kcont), b.data_ptr(), b_batch,
          ldb, int(b_kcont), out.data_ptr(), d_batch, Q if ldd is None else ldd, splits,
          scratch.data_ptr() if scratch is not None else None, PRECISION)
    return out


class _Transpose12(torch.autograd.Function):
    """(B, R, C) -> (B, C, R), both contiguous: one tiled launch each way (csrc/pointwise.hip, apn_pw_transpose)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, R, C = x.shape
        out = torch.empty(B, C, R, device=x.device)
        _call("apn_pw_transpose", x.device, B, R, C, x.data_ptr(), out.data_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, C, R = g.shape
        out = torch.empty(B, R, C, device=g.device)
        _call("apn_pw_transpose", g.device, B, C, R, g.data_ptr(), out.data_ptr())
        return out


def transpose12(x):
    """`x.permute(0, 2, 1).contiguous()` for a float32 (B, R, C) tensor on the GPU, as ONE tiled copy each way (the
    generator switches between the per-point layers' (B, C, N) and the grouper's / attention's (B, N, C) eight times per
    forward pass; PyTorch's strided copy took 25 us for 16 MB).  Anything else takes PyTorch's path."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[0] <= 65535 and x.numel() > 0:
        return _Transpose12.apply(x)
    return x.permute(0, 2, 1).contiguous()
