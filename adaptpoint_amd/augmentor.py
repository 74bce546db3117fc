"""The AdaptPoint imitator (generator): anchors -> predictor network -> per-anchor rigid/scale
deformation -> kernel-regressed blend -> unit sphere -> point mask.

Host-side mirror of `AdaptPoint_Augmentor`
(openpoints/models_adaptpoint/generator_component4_15.py:119-327).  The predictor network is
`adaptpoint_amd.imitator.SAComponent` (attribute `predict_prob_layer`, so a reference state_dict
loads unchanged); what this file adds is the geometry that consumes its two outputs:

    anchors  a_m   = x[FPS(x, M)]                                            (:150-151)
    factors  R_m, s_m, t_m from prob (B,M,9) and the step's random draws     (:236-297)
    moved    y_mn  = (x_n - a_m) R_m diag(s_m) + t_m + a_m                   (:295, :166)
    weights  w_mn  = exp(-|(a_m - x_n) * axis|^2 / (2 sigma^2))              (:204-234)
    blend    z_n   = sum_m w_mn y_mn / sum_m w_mn                            (:231-232)
    output   mask_n * unit_sphere(z)_n                                       (:313-327, :173)

It is written around ONE set of random draws per call (`Noise`), because that is what has to be
pinned for parity: the reference draws them from the CPU generator in a fixed order
(`torch.Tensor(B,M,3).uniform_` -> `torch.bernoulli` -> two `torch.randint(1, 8, ...)`, :245-247,
:218, :308) after the Gumbel noise of the mask (:714).  `draw_noise` makes the same calls in the
same order, so with the same CPU seed the mirror consumes the generator exactly like the
reference on CPU; on the GPU the Gumbel noise comes from the device generator (as it does in the
reference), or the whole `Noise` is passed in (tests do that to replay a CPU golden).
"""
import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .fused import _call
from .imitator import SAComponent, index_points
from .layers import furthest_point_sample


@dataclass
class Noise:
    """The random draws of one generator call.
    gumbel_expo (B,N,2): Exp(1) samples behind the mask's Gumbel noise (None: drawn on the logits'
    device at use); keep (B,M,3) in {0,1}: which of rotation / scaling / translation each anchor
    applies; axes (B,M,3) in {0,1}: the axes its scaling and translation act on; kernel_axes
    (B,1,3) in {0,1}: the axes the kernel-regression distance is measured along."""
    keep: torch.Tensor
    axes: torch.Tensor
    kernel_axes: torch.Tensor
    gumbel_expo: Optional[torch.Tensor] = None

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return Noise(mv(self.keep), mv(self.axes), mv(self.kernel_axes), mv(self.gumbel_expo))


def _axis_bits(code):
    """1..7 -> its three bits, least significant first: (..., 3) int32 (:299-311)."""
    return ((code.unsqueeze(-1) >> torch.arange(3)) & 1).int()


def draw_noise(batch, n_points, n_anchor, with_gumbel=False):
    """The reference's CPU-generator calls, in its order.  with_gumbel=True also draws the mask's
    Exp(1) samples first -- where `F.gumbel_softmax` draws them when the logits live on the CPU."""
    expo = torch.empty(batch, n_points, 2).exponential_() if with_gumbel else None
    keep = torch.bernoulli(torch.empty(batch, n_anchor, 3).uniform_(0, 1))
    axes = _axis_bits(torch.randint(1, 8, (batch, n_anchor)))
    kernel_axes = _axis_bits(torch.randint(1, 8, (batch, 1)))
    return Noise(keep, axes, kernel_axes, expo)


def draw_noise_on(device, batch, n_points, n_anchor):
    """The same distributions drawn from the DEVICE generator (no host-to-device copy: a captured
    hipGraph replays fresh draws).  Not the reference's CPU stream of numbers -- use `draw_noise`
    (or the module's default) when the reference's draws are to be reproduced."""
    expo = torch.empty(batch, n_points, 2, device=device).exponential_()
    keep = torch.bernoulli(torch.rand(batch, n_anchor, 3, device=device))
    bits = torch.arange(3, device=device)
    axes = ((torch.randint(1, 8, (batch, n_anchor), device=device).unsqueeze(-1) >> bits) & 1).int()
    kernel_axes = ((torch.randint(1, 8, (batch, 1), device=device).unsqueeze(-1) >> bits) & 1).int()
    return Noise(keep, axes, kernel_axes, expo)


class _AnchorTransforms(torch.autograd.Function):
    """csrc/augment.hip: one launch forward, one backward, for what the composed form below spends ~40 + ~40 on."""

    @staticmethod
    def forward(ctx, prob, keep, axes, ranges):
        prob = prob.contiguous()
        n = prob.numel() // 9
        lin = torch.empty(*prob.shape[:-1], 3, 3, device=prob.device)
        off = torch.empty(*prob.shape[:-1], 3, device=prob.device)
        _call("apn_anchor_transforms", prob.device, n, prob.data_ptr(), keep.data_ptr(), axes.data_ptr(), *ranges,
              lin.data_ptr(), off.data_ptr())
        ctx.save_for_backward(prob, keep, axes)
        ctx.ranges = ranges
        return lin, off

    @staticmethod
    def backward(ctx, g_lin, g_off):
        prob, keep, axes = ctx.saved_tensors
        g = torch.empty_like(prob)
        _call("apn_anchor_transforms_grad", prob.device, prob.numel() // 9, prob.data_ptr(), keep.data_ptr(),
              axes.data_ptr(), *ctx.ranges, g_lin.contiguous().data_ptr() if g_lin is not None else None,
              g_off.contiguous().data_ptr() if g_off is not None else None, g.data_ptr())
        return g, None, None, None


def anchor_transforms(prob, noise, r_range, s_range, t_range, fused=True):
    """prob (B,M,9) -> A (B,M,3,3) = R diag(s) and t (B,M,3)   (:236-297): the extension's kernel on the GPU, the
    composed PyTorch form elsewhere."""
    if fused and prob.is_cuda and prob.dtype == torch.float32:
        keep = noise.keep.to(prob.dtype).contiguous()
        axes = noise.axes.to(prob.dtype).contiguous()
        return _AnchorTransforms.apply(prob, keep, axes, (float(r_range), float(s_range), float(t_range)))
    return anchor_transforms_composed(prob, noise, r_range, s_range, t_range)


def anchor_transforms_composed(prob, noise, r_range, s_range, t_range):
    """prob (B,M,9) -> A (B,M,3,3) = R diag(s) and t (B,M,3)   (:236-297).
    Rotation angles tanh(.) * r_range degrees, scales 1 + sigmoid(.) * (s_range - 1), offsets
    tanh(.) * t_range; each of the three switched per anchor by `keep`, scale and offset confined
    to the anchor's `axes` (a scale of 0 means "axis not scaled": it becomes 1)."""
    keep, axes = noise.keep.to(prob.dtype), noise.axes.to(prob.dtype)
    # (the reference multiplies by a float32 tensor holding pi; a Python scalar is rounded to float32 by the
    # same multiplication and needs no host-to-device copy, so the step stays hipGraph-capturable)
    ang = math.pi * (torch.tanh(prob[..., 0:3]) * r_range) / 180.0 * keep[..., 0:1]
    s = (torch.sigmoid(prob[..., 3:6]) * (s_range - 1) + 1) * keep[..., 1:2] * axes
    s = s + (s == 0)
    t = torch.tanh(prob[..., 6:9]) * t_range * keep[..., 2:3] * axes
    sx, sy, sz = torch.sin(ang).unbind(-1)
    cx, cy, cz = torch.cos(ang).unbind(-1)
    # rows of the rotation the reference composes (:288-290; its centre entry is sz*sy*sx + cz*cy)
    rot = torch.stack([cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                       sz * cy, sz * sy * sx + cz * cy, sz * sy * cx - cz * sx,
                       -sy, cy * sx, cy * cx], dim=-1).unflatten(-1, (3, 3))
    return rot * s.unsqueeze(-2), t


def kernel_weights(x, anchors, kernel_axes, sigma):
    """w (B,M,N) = exp(-|(a_m - x_n) * axes|^2 / (2 sigma^2))   (:204-229).  The reference takes a
    square root and squares it again; the squared distance is used directly here (same value to an
    ulp, and no 0/0 in the derivative at the anchors themselves)."""
    d = (anchors.unsqueeze(2) - x.unsqueeze(1)) * kernel_axes.to(x.dtype).unsqueeze(2)
    return torch.exp(-0.5 * d.square().sum(-1) / sigma ** 2)


def deform(x, anchors, lin, off, w):
    """z (B,N,3) = sum_m w_mn ((x_n - a_m) A_m + t_m + a_m) / sum_m w_mn   (:156-167, :231-232)."""
    moved = torch.matmul(x.unsqueeze(1) - anchors.unsqueeze(2), lin) + (off + anchors).unsqueeze(2)
    return (w.unsqueeze(-1) * moved).sum(1) / w.sum(1).unsqueeze(-1)


class _DeformNormaliseMask(torch.autograd.Function):
    """csrc/augment.hip: kernel regression + blend + unit sphere + mask, one workgroup per cloud, one launch each way
    (the composed form below -- kernel_weights, deform, unit_sphere and the mask product -- is ~30 launches forward and
    ~50 backward).  Gradients reach the per-anchor transforms and the mask; the cloud and its anchors are data."""

    @staticmethod
    def forward(ctx, x, anchors, lin, off, axes, mask0, sigma):
        B, N, _ = x.shape
        M = anchors.shape[1]
        x, anchors, lin, off = x.contiguous(), anchors.contiguous(), lin.contiguous(), off.contiguous()
        z = torch.empty_like(x)
        stat = torch.empty(B, 8, device=x.device)
        out = torch.empty_like(x)
        _call("apn_deform_forward", x.device, B, N, M, x.data_ptr(), anchors.data_ptr(), lin.data_ptr(), off.data_ptr(),
              axes.data_ptr(), mask0.data_ptr(), float(sigma), z.data_ptr(), stat.data_ptr(), out.data_ptr())
        ctx.save_for_backward(x, anchors, axes, mask0, z, stat)
        ctx.sigma = float(sigma)
        return out

    @staticmethod
    def backward(ctx, g):
        x, anchors, axes, mask0, z, stat = ctx.saved_tensors
        B, N, _ = x.shape
        M = anchors.shape[1]
        g = g.contiguous()
        g_lin = torch.empty(B, M, 3, 3, device=x.device)
        g_off = torch.empty(B, M, 3, device=x.device)
        g_mask = torch.empty(B, N, device=x.device) if ctx.needs_input_grad[5] else None
        _call("apn_deform_backward", x.device, B, N, M, x.data_ptr(), anchors.data_ptr(), axes.data_ptr(),
              mask0.data_ptr(), ctx.sigma, z.data_ptr(), stat.data_ptr(), g.data_ptr(), g_lin.data_ptr(),
              g_off.data_ptr(), g_mask.data_ptr() if g_mask is not None else None)
        return None, None, g_lin, g_off, None, g_mask, None


def deform_normalise_mask(x, anchors, lin, off, kernel_axes, mask0, sigma, fused=True):
    """x (B,N,3), per-anchor (lin, off), kernel_axes (B,1,3), mask0 (B,N) -> the augmented cloud (B,N,3)
    (:156-232, 313-327, 180).  The extension's kernel where it applies (GPU, float32, <= 8 anchors, N <= 4096, no
    gradient asked for the cloud itself), the composed form otherwise."""
    if (fused and x.is_cuda and x.dtype == torch.float32 and anchors.shape[1] <= 8 and x.shape[1] <= 4096
            and not x.requires_grad and not anchors.requires_grad):
        axes = kernel_axes.to(x.dtype).reshape(x.shape[0], 3).contiguous()
        return _DeformNormaliseMask.apply(x, anchors, lin, off, axes, mask0.contiguous(), sigma)
    w = kernel_weights(x, anchors, kernel_axes, sigma)
    return unit_sphere(deform(x, anchors, lin, off, w)) * mask0.unsqueeze(-1)


def unit_sphere(z):
    """Centre each cloud and scale it just inside the unit sphere (:313-327)."""
    z = z - z.mean(dim=-2, keepdim=True)
    r = z.square().sum(-1).sqrt().amax(dim=-1)
    return z * ((1 / r) * 0.999999).view(-1, 1, 1)


class AdaptPointAugmentor(nn.Module):
    """generator_component4_15.py:119-181 (`AdaptPoint_Augmentor`; cfg keys of
    cfgs/scanobjectnn/pointnext-s_adaptpoint_1.yaml:50-56)."""

    def __init__(self, w_num_anchor=4, w_sigma=0.5, w_R_range=10, w_S_range=3, w_T_range=0.25, fused=True):
        super().__init__()
        self.num_anchor = w_num_anchor
        self.sigma = w_sigma
        self.w_R_range, self.w_S_range, self.w_T_range = w_R_range, w_S_range, w_T_range
        self.fused = fused
        self.predict_prob_layer = SAComponent(fused=fused)

    def forward(self, xyz, noise: Optional[Noise] = None):
        """xyz (B,N,3) -> (xyz, augmented (B,N,3)).  `noise`: the call's random draws (default:
        drawn here the way the reference draws them)."""
        B, N, _ = xyz.shape
        xyz = xyz.contiguous()
        anchor_idx = furthest_point_sample(xyz, self.num_anchor).long()
        anchors = index_points(xyz, anchor_idx)
        prob, logits = self.predict_prob_layer(xyz, anchor_idx, return_logits=True)
        if noise is None:
            # the reference's order: Gumbel noise of the mask (device generator), then the CPU draws
            expo = torch.empty_like(logits, memory_format=torch.contiguous_format).exponential_()
            noise = draw_noise(B, N, self.num_anchor)
            noise.gumbel_expo = expo
        noise = noise.to(xyz.device)
        mask = self.predict_prob_layer.hard_mask(logits, noise.gumbel_expo)
        lin, off = anchor_transforms(prob, noise, self.w_R_range, self.w_S_range, self.w_T_range, self.fused)
        out = deform_normalise_mask(xyz, anchors, lin, off, noise.kernel_axes, mask[:, :, 0], self.sigma, self.fused)
        return xyz, out


AdaptPoint_Augmentor = AdaptPointAugmentor          # the reference's registry name
