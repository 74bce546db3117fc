"""The AdaptPoint imitator's PointsetGrouper over the gfx950 operators.

Host-side mirror of `PointsetGrouper`
(openpoints/models_adaptpoint/generator_component4_15.py:368-431): FPS, gather of the sampled
points, ball query, then -- for normalize="anchor", the only mode the reference instantiates
(generator_component4_15.py:611-612) -- `max_k(alpha * (points[idx] - points[fps]) + beta)`.
The reference materialises four (B, np, K, C) tensors for that last line (~200 MB each at
B=32, N=1024, C=128); `group_max` computes it in one pass (csrc/pointset_group.hip).
Parameter names (`affine_alpha`, `affine_beta`, shape (1,1,1,C)) match the reference, so its
state_dict loads unchanged.
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .fused import _call
from .layers import ball_query, furthest_point_sample


class _GroupMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, fidx, alpha, beta):
        if not points.is_cuda:
            raise RuntimeError("adaptpoint_amd.pointset.group_max needs CUDA/HIP tensors: the product "
                               "path has no CPU fallback")
        points = points.contiguous().float()
        idx = idx.contiguous().int()
        fidx = fidx.contiguous().int()
        B, N, C = points.shape
        M, K = idx.shape[1], idx.shape[2]
        al = alpha.detach().reshape(-1).contiguous().float()
        be = beta.detach().reshape(-1).contiguous().float()
        assert al.numel() == C and be.numel() == C and fidx.shape == (B, M)
        out = torch.empty(B, C, M, dtype=torch.float32, device=points.device)
        ksel = torch.empty(B, M, C, dtype=torch.uint8, device=points.device)
        _call("apn_pointset_group_max", points.device, B, N, M, C, K, points.data_ptr(),
              idx.data_ptr(), fidx.data_ptr(), al.data_ptr(), be.data_ptr(), out.data_ptr(),
              ksel.data_ptr())
        ctx.save_for_backward(points, idx, fidx, al, ksel)
        ctx.shapes = (alpha.shape, beta.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        points, idx, fidx, al, ksel = ctx.saved_tensors
        B, N, C = points.shape
        M, K = idx.shape[1], idx.shape[2]
        g = g.contiguous().float()
        g_points = ops.zeros(*points.shape, dtype=points.dtype, device=points.device)
        rows = _lib.load().apn_pointset_group_rows(B, M, C)
        part = torch.empty(rows, 2 * C, dtype=torch.float32, device=points.device)
        _call("apn_pointset_group_max_grad", points.device, B, N, M, C, K, points.data_ptr(),
              idx.data_ptr(), fidx.data_ptr(), al.data_ptr(), ksel.data_ptr(), g.data_ptr(),
              g_points.data_ptr(), part.data_ptr())
        from .fused_wide import _colsum           # fixed-order float64 column sums (hipGraph-replay safe)
        sums = _colsum(part).float()
        return (g_points, None, None, sums[:C].reshape(ctx.shapes[0]), sums[C:].reshape(ctx.shapes[1]))


def group_max(points, idx, fidx, alpha, beta):
    """out (B,C,M) = max_k(alpha * (points[idx] - points[fidx]) + beta); points (B,N,C),
    idx (B,M,K) int, fidx (B,M) int, alpha/beta with C elements.  The forward is bit-identical to the reference's
    composition; the gradient w.r.t. `points` is accumulated with float LDS atomics and is therefore equal only up to
    the order of its additions from run to run (csrc/pointset_group.hip)."""
    return _GroupMax.apply(points, idx, fidx, alpha, beta)


def group_max_supported(points, k):
    c = points.shape[-1]
    return points.is_cuda and c >= 4 and c <= 1024 and (c & (c - 1)) == 0 and 0 < k <= 255


class PointsetGrouper(nn.Module):
    """generator_component4_15.py:368-431."""

    def __init__(self, channel, reduce, kneighbors, radi, normalize="anchor", fused=True, **kwargs):
        super().__init__()
        self.fused = fused            # False: the reference's composition (on the GPU) after FPS / ball query
        self.reduce = reduce
        self.kneighbors = kneighbors
        self.radi = radi
        self.normalize = normalize.lower() if normalize is not None else None
        if self.normalize not in ("center", "anchor"):
            self.normalize = None
        if self.normalize is not None:
            self.affine_alpha = nn.Parameter(torch.ones([1, 1, 1, channel]))
            self.affine_beta = nn.Parameter(torch.zeros([1, 1, 1, channel]))

    @torch.no_grad()
    def index(self, xyz, ties=None, nested=False):
        """The stage's index work -- a function of the coordinates alone: (fps_idx (B,np), new_xyz (B,np,3),
        idx (B,np,K)).  `forward(..., index=...)` takes it, so a caller can run it ahead of the features.
        nested=True (the imitator's four stages sample from each other's samples): the nested sampler
        (`layers.furthest_point_sample_nested`; same picks, a copy of the previous stage's first np picks wherever its
        record `ties` allows) -- the tuple then carries this stage's record as a fourth element."""
        xyz = xyz.contiguous()
        if nested and xyz.is_cuda and xyz.shape[1] <= 4096:
            from .layers import furthest_point_sample_nested
            fps_idx, new_xyz, rec = furthest_point_sample_nested(xyz, xyz.shape[1] // self.reduce, ties)
            return fps_idx, new_xyz, ball_query(self.radi, self.kneighbors, xyz, new_xyz), rec
        fps_idx = furthest_point_sample(xyz, xyz.shape[1] // self.reduce)               # :406
        new_xyz = torch.gather(xyz, 1, fps_idx.long().unsqueeze(-1).expand(-1, -1, 3))   # :407
        idx = ball_query(self.radi, self.kneighbors, xyz, new_xyz)                       # :412
        return fps_idx, new_xyz, idx

    def forward(self, xyz, points, index=None):
        """xyz (B,N,3), points (B,N,C) -> new_xyz (B,np,3), new_points (B,C,np)."""
        xyz = xyz.contiguous()
        if index is not None and not xyz.requires_grad:
            fps_idx, new_xyz, idx = index[:3]
        else:
            fps_idx = furthest_point_sample(xyz, xyz.shape[1] // self.reduce)               # :406
            new_xyz = torch.gather(xyz, 1, fps_idx.long().unsqueeze(-1).expand(-1, -1, 3))   # :407
            idx = ball_query(self.radi, self.kneighbors, xyz, new_xyz)                       # :412
        if self.fused and self.normalize == "anchor" and group_max_supported(points, self.kneighbors):
            return new_xyz, group_max(points, idx, fps_idx, self.affine_alpha, self.affine_beta)
        # the other modes, as the reference composes them (:413-429)
        B, N, C = points.shape
        bi = torch.arange(B, device=points.device).view(B, 1, 1)
        grouped = points[bi, idx.long(), :]
        if self.normalize is not None:
            if self.normalize == "center":
                mean = grouped.mean(dim=2, keepdim=True)
            else:
                mean = torch.gather(points, 1, fps_idx.long().unsqueeze(-1).expand(-1, -1, C)).unsqueeze(-2)
            grouped = self.affine_alpha * (grouped - mean) + self.affine_beta
        return new_xyz, grouped.max(dim=2)[0].permute(0, 2, 1)
