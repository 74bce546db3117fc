"""Operator layer of the hot path: tensors in, tensors out, autograd where the operator has a
gradient -- over the nine C-ABI operators in `adaptpoint_amd.ops`.

The callable NAMES and argument orders are the contract the reference's models are written against
(openpoints/models/layers/__init__.py:8-14: `furthest_point_sample(xyz, npoint)`,
`ball_query(radius, nsample, xyz, new_xyz)`, `grouping_operation(features, idx)`,
`gather_operation(features, idx)`, `three_nn(unknown, known)`,
`three_interpolate(features, idx, weight)`, `three_interpolation(unknown_xyz, known_xyz, feat)`);
what they compute follows subsample.py:76-144, group.py:76-200, upsampling.py:11-102.  Everything
else here is this build's own shape:

  * the three index operators have no gradient, so they are plain no-grad functions, not
    autograd Functions;
  * the two copy operators (rows selected by an index tensor, gradient = scatter-add) share one
    autograd Function;
  * neighbourhood grouping is two small modules, `BallGrouper` / `KnnGrouper`, that return the
    neighbour indices separately from the grouped tensors, because the fused kernels
    (adaptpoint_amd.fused, adaptpoint_amd.pointset) consume the indices and never materialise
    the (B,C,M,K) tensors the unfused path needs.

Buffers are allocated on the inputs' device and pre-initialised as the extension's contract asks
(temp = 1e10, ball-query indices = 0, gradient targets = 0).  Calls go through the
module attribute `ops.<wrapper>` at call time, so a test can stand the CPU oracle in for the
extension underneath this whole layer.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops


def _alloc(like, *shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=like.device)


def _need_contiguous(**tensors):
    for name, t in tensors.items():
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous (the extension reads raw rows)")


# ---------------------------------------------------------------- index operators (no gradient)
@torch.no_grad()
def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) -> (B,npoint) int32 indices, starting from point 0   (subsample.py:78-98)."""
    _need_contiguous(xyz=xyz)
    B, N = xyz.shape[:2]
    picks = _alloc(xyz, B, npoint, dtype=torch.int32)
    running_min = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
    ops.furthest_point_sampling_wrapper(B, N, npoint, xyz, running_min, picks)
    return picks


@torch.no_grad()
def furthest_point_sample_nested(xyz, npoint, ties=None):
    """One level of a sampling PYRAMID (a block / stage that samples from the previous one's samples): FPS is
    progressive -- on a sample's picks, in pick order, it returns picks 0, 1, 2, ... again as long as every arg-max was
    unique -- so a deeper level is a copy wherever the previous level's record `ties` (B,) int32 says so, the full
    sampler elsewhere (csrc/fps.hip, NEST): the same indices as `furthest_point_sample`, bit for bit.
    xyz (B,N,3): the cloud (ties=None) or the previous level's `new_xyz`.  -> (picks (B,npoint) int32, new_xyz
    (B,npoint,3) = xyz[picks], this level's record)."""
    from .fused import _call
    _need_contiguous(xyz=xyz)
    if not xyz.is_cuda:
        raise RuntimeError("adaptpoint_amd.layers needs CUDA/HIP tensors: the product path has no CPU fallback")
    B, N = xyz.shape[:2]
    picks = _alloc(xyz, B, npoint, dtype=torch.int32)
    new_xyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device)
    rec = torch.empty(B, dtype=torch.int32, device=xyz.device)
    _call("apn_furthest_point_sampling_nested", xyz.device, B, N, npoint, xyz.data_ptr(),
          None if ties is None else ties.data_ptr(), picks.data_ptr(), new_xyz.data_ptr(), rec.data_ptr())
    return picks, new_xyz, rec


@torch.no_grad()
def ball_query(radius, nsample, xyz, new_xyz):
    """The first `nsample` points of xyz (B,N,3) inside the ball around each new_xyz (B,M,3),
    in index order, padded with the first hit; all-zero rows for empty balls: (B,M,nsample) int32
    (group.py:179-196)."""
    _need_contiguous(xyz=xyz, new_xyz=new_xyz)
    B, N = xyz.shape[:2]
    M = new_xyz.shape[1]
    nbr = ops.zeros(B, M, nsample, dtype=torch.int32, device=xyz.device)
    ops.ball_query_wrapper(B, N, M, radius, nsample, new_xyz, xyz, nbr)
    return nbr


@torch.no_grad()
def three_nn(unknown, known):
    """The three nearest `known` (B,m,3) points of every `unknown` (B,n,3) point:
    (distance (B,n,3) -- the square ROOT of what the kernel writes --, index (B,n,3) int32)
    (upsampling.py:14-35)."""
    _need_contiguous(unknown=unknown, known=known)
    B, n = unknown.shape[:2]
    d2 = _alloc(unknown, B, n, 3)
    nearest = _alloc(unknown, B, n, 3, dtype=torch.int32)
    ops.three_nn_wrapper(B, n, known.shape[1], unknown, known, d2, nearest)
    return d2.sqrt_(), nearest


# ---------------------------------------------------------------- copy operators (gradient = scatter-add)
class _TakeRows(Function):
    """out[b, c, ...] = features[b, c, idx[b, ...]] for idx (B,M) ("gather", subsample.py:108-141)
    or idx (B,M,K) ("group", group.py:76-114).  float32 even under autocast, as the reference's
    custom_fwd(cast_inputs=float32)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, idx):
        _need_contiguous(features=features, idx=idx)
        B, C, N = features.shape
        out = _alloc(features, B, C, *idx.shape[1:])
        if idx.dim() == 2:
            ops.gather_points_wrapper(B, C, N, idx.shape[1], features, idx, out)
        else:
            ops.group_points_wrapper(B, C, N, idx.shape[1], idx.shape[2], features, idx, out)
        ctx.save_for_backward(idx)
        ctx.n_rows = N
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, C = g.shape[:2]
        acc = ops.zeros(B, C, ctx.n_rows, dtype=torch.float32, device=g.device)
        g = g.contiguous()
        if idx.dim() == 2:
            ops.gather_points_grad_wrapper(B, C, ctx.n_rows, idx.shape[1], g, idx, acc)
        else:
            ops.group_points_grad_wrapper(B, C, ctx.n_rows, idx.shape[1], idx.shape[2], g, idx, acc)
        return acc, None


def gather_operation(features, idx):
    """features (B,C,N), idx (B,M) int32 -> (B,C,M)."""
    if idx.dim() != 2:
        raise RuntimeError("gather_operation takes idx of shape (B, M)")
    return _TakeRows.apply(features, idx)


def grouping_operation(features, idx):
    """features (B,C,N), idx (B,M,K) int32 -> (B,C,M,K)."""
    if idx.dim() != 3:
        raise RuntimeError("grouping_operation takes idx of shape (B, M, K)")
    return _TakeRows.apply(features, idx)


class _Blend3(Function):
    """out[b,c,i] = sum_j weight[b,i,j] * features[b,c,idx[b,i,j]], j < 3   (upsampling.py:43-86)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, idx, weight):
        _need_contiguous(features=features, idx=idx, weight=weight)
        B, C, m = features.shape
        n = idx.shape[1]
        out = _alloc(features, B, C, n)
        ops.three_interpolate_wrapper(B, C, m, n, features, idx, weight, out)
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, g):
        idx, weight = ctx.saved_tensors
        B, C, n = g.shape
        acc = ops.zeros(B, C, ctx.m, dtype=torch.float32, device=g.device)
        ops.three_interpolate_grad_wrapper(B, C, n, ctx.m, g.contiguous(), idx, weight, acc)
        return acc, None, None


three_interpolate = _Blend3.apply


def inverse_distance_weights(dist, eps=1e-8):
    """(B,n,3) distances -> weights proportional to 1 / (dist + eps), summing to 1 (upsampling.py:97-100)."""
    inv = 1.0 / (dist + eps)
    return inv / inv.sum(dim=2, keepdim=True)


@torch.no_grad()
def three_nn_weights(unknown, known):
    """`three_nn` + `inverse_distance_weights` for the feature-propagation stages: (nearest (B,n,3) int32, weights
    (B,n,3)) with the weights formed from the kernel's squared distances in ONE launch (`apn_three_nn_weights`: the
    square root, the reciprocals, the sum and the division of upsampling.py:97-100 in that order) instead of five."""
    _need_contiguous(unknown=unknown, known=known)
    B, n = unknown.shape[:2]
    d2 = _alloc(unknown, B, n, 3)
    nearest = _alloc(unknown, B, n, 3, dtype=torch.int32)
    ops.three_nn_wrapper(B, n, known.shape[1], unknown, known, d2, nearest)
    if not d2.is_cuda:
        return nearest, inverse_distance_weights(d2.sqrt_())
    from .fused import _call
    w = _alloc(unknown, B, n, 3)
    _call("apn_three_nn_weights", d2.device, B * n, d2.data_ptr(), w.data_ptr())
    return nearest, w


def three_interpolation(unknown_xyz, known_xyz, know_feat):
    """Features (B,C,m) known at known_xyz (B,m,3), carried to unknown_xyz (B,n,3) by
    inverse-distance weighting over the three nearest known points -> (B,C,n) (upsampling.py:92-102)."""
    nearest, weights = three_nn_weights(unknown_xyz.contiguous(), known_xyz.contiguous())
    return three_interpolate(know_feat.contiguous(), nearest, weights)


# ---------------------------------------------------------------- neighbourhood grouping
def _relative_positions(support_xyz, query_xyz, idx):
    """(B,3,M,K): neighbour positions minus their query (group.py:248-251)."""
    rows = grouping_operation(support_xyz.transpose(1, 2).contiguous(), idx)
    return rows - query_xyz.transpose(1, 2).unsqueeze(-1)


class BallGrouper(nn.Module):
    """Ball-query neighbourhoods (`QueryAndGroup`, group.py:206-255, for relative positions):
    forward -> (dp (B,3,M,K), fj (B,C,M,K) | None); `normalize_dp` divides dp by the radius.
    `neighbours` alone gives the (B,M,K) indices the fused kernels take."""

    def __init__(self, radius, nsample, normalize_dp=False):
        super().__init__()
        self.radius, self.nsample, self.normalize_dp = radius, nsample, normalize_dp

    def neighbours(self, query_xyz, support_xyz):
        return ball_query(self.radius, self.nsample, support_xyz, query_xyz)

    def forward(self, query_xyz, support_xyz, features=None, idx=None):
        """`idx`: neighbours computed ahead of time (the index stage depends on coordinates only)."""
        if idx is None:
            idx = self.neighbours(query_xyz, support_xyz)
        dp = _relative_positions(support_xyz, query_xyz, idx)
        if self.normalize_dp:
            dp = dp / self.radius
        return dp, (None if features is None else grouping_operation(features, idx))


class KnnGrouper(nn.Module):
    """k-nearest-neighbour neighbourhoods (`KNNGroup` / `KNN`, group.py:12-28, 275-320): the same
    two outputs; `normalize_dp` divides dp by each cloud's largest neighbour distance."""

    def __init__(self, nsample, normalize_dp=False):
        super().__init__()
        self.nsample, self.normalize_dp = nsample, normalize_dp

    @torch.no_grad()
    def neighbours(self, query_xyz, support_xyz):
        # distances laid out (B,N,M) and the k smallest taken along N, as the reference does: the
        # order among near-equal distances then is the same
        nearest = torch.cdist(support_xyz, query_xyz).topk(self.nsample, dim=1, largest=False).indices
        return nearest.transpose(1, 2).contiguous().int()

    def forward(self, query_xyz, support_xyz, features=None):
        idx = self.neighbours(query_xyz, support_xyz)
        dp = _relative_positions(support_xyz, query_xyz, idx)
        if self.normalize_dp:
            dp = dp / dp.square().sum(1).sqrt().amax(dim=(1, 2)).view(-1, 1, 1, 1)
        return dp, (None if features is None else grouping_operation(features, idx))


class GroupAll(nn.Module):
    """One neighbourhood holding every point (group.py:258-272): dp (B,3,1,N), fj (B,C,1,N)."""

    def forward(self, query_xyz, support_xyz, features=None):
        dp = support_xyz.transpose(1, 2).unsqueeze(2)
        return dp, (None if features is None else features.unsqueeze(2))


def make_grouper(group_args):
    """The cfgs' `group_args` (NAME ballquery | knn, radius, nsample, normalize_dp;
    cfgs/scanobjectnn/pointnext-s.yaml:22-24) -> a grouper; nsample None -> GroupAll
    (what pointnext.py:130-133 asks for in its last stage)."""
    args = dict(group_args)
    kind = args.get('NAME', 'ballquery')
    if args.get('nsample', 20) is None:
        return GroupAll()
    if kind == 'ballquery':
        return BallGrouper(args.get('radius', 0.1), args.get('nsample', 20), args.get('normalize_dp', False))
    if kind == 'knn':
        return KnnGrouper(args.get('nsample', 20), args.get('normalize_dp', False))
    raise NotImplementedError(f"grouper '{kind}' is outside the hot-path build")
