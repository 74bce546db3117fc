"""Host-side mirror of the reference's operator layer
(openpoints/models/layers/subsample.py:76-144, group.py:76-272,322-353,
upsampling.py:11-102): the same public names, argument meaning and autograd
behaviour, over the gfx950 operators in `adaptpoint_amd.ops`.

Outputs are allocated with torch.empty/zeros on the input's device (the
reference uses the legacy torch.cuda.FloatTensor constructors, which allocate
on the *current* device).
"""
import copy
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops


class FurthestPointSampling(Function):
    """subsample.py:76-102."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        ops.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """subsample.py:108-141 / group.py:140-171."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty(B, C, npoint, dtype=torch.float32, device=features.device)
        ops.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        ops.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class BallQuery(Function):
    """group.py:177-200."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor,
                new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros(B, npoint, nsample, dtype=torch.int32, device=xyz.device)
        ops.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    """group.py:76-114 (float32 even under autocast, as custom_fwd(cast_inputs) there)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty(B, C, nfeatures, nsample, dtype=torch.float32, device=features.device)
        ops.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, None]:
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        ops.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx,
                                      grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class ThreeNN(Function):
    """upsampling.py:11-37: returns (sqrt(dist2), idx)."""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor):
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, N, 3, dtype=torch.int32, device=unknown.device)
        ops.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """upsampling.py:43-86."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor):
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        ops.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros(B, c, m, dtype=torch.float32, device=grad_out.device)
        ops.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight,
                                           grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


def three_interpolation(unknown_xyz, known_xyz, know_feat):
    """upsampling.py:92-102: inverse-distance weights over the 3 nearest known points."""
    dist, idx = three_nn(unknown_xyz, known_xyz)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    return three_interpolate(know_feat, idx, weight)


class QueryAndGroup(nn.Module):
    """group.py:206-255 (ball query, relative positions, optional /radius)."""

    def __init__(self, radius: float, nsample: int, relative_xyz=True, normalize_dp=False,
                 normalize_by_std=False, normalize_by_allstd=False, normalize_by_allstd2=False,
                 return_only_idx=False, **kwargs):
        super().__init__()
        self.radius, self.nsample = radius, nsample
        self.normalize_dp = normalize_dp
        self.normalize_by_std = normalize_by_std
        self.normalize_by_allstd = normalize_by_allstd
        self.normalize_by_allstd2 = normalize_by_allstd2
        assert self.normalize_dp + self.normalize_by_std + self.normalize_by_allstd < 2
        self.relative_xyz = relative_xyz
        self.return_only_idx = return_only_idx

    def forward(self, query_xyz, support_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, support_xyz, query_xyz)
        if self.return_only_idx:
            return idx
        xyz_trans = support_xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, nsample)
        if self.relative_xyz:
            grouped_xyz = grouped_xyz - query_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_dp:
                grouped_xyz /= self.radius
        grouped_features = grouping_operation(features, idx) if features is not None else None
        return grouped_xyz, grouped_features


class GroupAll(nn.Module):
    """group.py:258-272."""

    def forward(self, new_xyz, xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        grouped_features = features.unsqueeze(2) if features is not None else None
        return grouped_xyz, grouped_features


class KNN(nn.Module):
    """group.py:12-28 (cdist + topk; pure torch in the reference too)."""

    def __init__(self, neighbors, transpose_mode=True):
        super().__init__()
        self.neighbors = neighbors

    @torch.no_grad()
    def forward(self, support, query):
        dist = torch.cdist(support, query)
        k_dist = dist.topk(k=self.neighbors, dim=1, largest=False)
        return k_dist.values, k_dist.indices.transpose(1, 2).contiguous().int()


class KNNGroup(nn.Module):
    """group.py:275-320."""

    def __init__(self, nsample: int, relative_xyz=True, normalize_dp=False,
                 return_only_idx=False, **kwargs):
        super().__init__()
        self.nsample = nsample
        self.knn = KNN(nsample, transpose_mode=True)
        self.relative_xyz = relative_xyz
        self.normalize_dp = normalize_dp
        self.return_only_idx = return_only_idx

    def forward(self, query_xyz, support_xyz, features=None):
        _, idx = self.knn(support_xyz, query_xyz)
        if self.return_only_idx:
            return idx
        idx = idx.int()
        xyz_trans = support_xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        if self.relative_xyz:
            grouped_xyz -= query_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_dp:
            grouped_xyz /= torch.amax(torch.sqrt(torch.sum(grouped_xyz ** 2, dim=1)),
                                      dim=(1, 2)).view(-1, 1, 1, 1)
        if features is not None:
            return grouped_xyz, grouping_operation(features, idx)
        return grouped_xyz, None


def get_aggregation_feautres(p, dp, f, fj, feature_type='dp_fj'):
    """group.py:323-335 (the reference's spelling is kept)."""
    if feature_type == 'dp_fj':
        fj = torch.cat([dp, fj], 1)
    elif feature_type == 'dp_fj_df':
        df = fj - f.unsqueeze(-1)
        fj = torch.cat([dp, fj, df], 1)
    elif feature_type == 'pi_dp_fj_df':
        df = fj - f.unsqueeze(-1)
        fj = torch.cat([p.transpose(1, 2).unsqueeze(-1).expand(-1, -1, -1, df.shape[-1]),
                        dp, fj, df], 1)
    elif feature_type == 'dp_df':
        df = fj - f.unsqueeze(-1)
        fj = torch.cat([dp, df], 1)
    return fj


# Channels entering a block's first conv for the feature types that
# get_aggregation_feautres builds (cf. layers/local_aggregation.py:13-29).
CHANNEL_MAP = {
    'dp_fj': lambda x: 3 + x,
    'dp_fj_df': lambda x: 2 * x + 3,
    'pi_dp_fj_df': lambda x: 2 * x + 6,
    'dp_df': lambda x: x + 3,
}


def create_grouper(group_args):
    """group.py:338-353."""
    group_args_copy = copy.deepcopy(dict(group_args))
    method = group_args_copy.pop('NAME', 'ballquery')
    radius = group_args_copy.pop('radius', 0.1)
    nsample = group_args_copy.pop('nsample', 20)
    if nsample is not None:
        if method == 'ballquery':
            return QueryAndGroup(radius, nsample, **group_args_copy)
        if method == 'knn':
            return KNNGroup(nsample, **group_args_copy)
        raise NotImplementedError(f"grouper {method}")
    return GroupAll()
