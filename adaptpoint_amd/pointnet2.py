"""PointNet++ blocks over the gfx950 operators (second half).

Host-side mirrors of the two modules PointNet++ / ASSANet are assembled from in the reference
(openpoints/models/backbone/pointnetv2.py:17-150):

  `SetAbstractionMSG`    `PointNetSAModuleMSG` (:17-106): FPS once, then one neighbourhood query and one shared MLP
                         + max-pool per scale (`ConvPool`, openpoints/models/layers/local_aggregation.py:140-239:
                         ball query, group xyz and features, 'dp_fj', Conv2d-BN-ReLU blocks, max over K, optional
                         residual branch on the sampled points' own features), results concatenated;
  `FeaturePropagation2`  `PointNetFPModule` (:108-150): three-nearest inverse-distance interpolation of the coarse
                         features, concatenation with the dense ones, Conv1d-BN-ReLU blocks.

Sub-module names and nesting are the reference's (`local_aggregations.<i>.SA_CONFIG_operator.convs.<j>.0/1`,
`...skipconv.0`, `convs.<j>.0/1`), so its state_dict loads unchanged.  The sampled coordinates and the residual
branch's features are selected with `gather_operation` (the extension's gather kernel and its scatter-add
gradient) where the reference writes `torch.gather` -- the same values (subsample.py:176-185
checks exactly that equality), which makes these blocks the module-level consumer of those two operators.
"""
import torch
import torch.nn as nn

from .layers import furthest_point_sample, gather_operation, make_grouper, three_interpolation
from .set_abstraction import _act, convblock


class ConvPool(nn.Module):
    """local_aggregation.py:140-239 for feature_type 'dp_fj' and max reduction."""

    def __init__(self, channels, conv_args=None, norm_args=None, act_args=None, group_args=None, use_res=False):
        super().__init__()
        channels = list(channels)
        conv_args = dict(conv_args or {})
        self.use_res = use_res
        if use_res:
            self.skipconv = (convblock(channels[0], channels[-1], 1, norm_args=None, act_args=None, **conv_args)
                             if channels[0] != channels[-1] else nn.Identity())
        channels[0] += 3                                           # 'dp_fj': relative positions + neighbour features
        convs = [convblock(channels[i], channels[i + 1], 2, norm_args=norm_args, act_args=act_args, **conv_args)
                 for i in range(len(channels) - 2)]
        convs.append(convblock(channels[-2], channels[-1], 2, norm_args=norm_args,
                               act_args=None if use_res else act_args, **conv_args))
        self.convs = nn.Sequential(*convs)
        self.act = _act(act_args)
        self.grouper = make_grouper(group_args)

    def forward(self, query_xyz, support_xyz, features, query_idx=None):
        dp, fj = self.grouper(query_xyz, support_xyz, features)
        identity = 0
        if self.use_res:
            if query_idx is not None and query_xyz.shape[1] != support_xyz.shape[1]:
                features = gather_operation(features.contiguous(), query_idx)      # the sampled points' own features
            identity = self.skipconv(features)
        pooled = torch.max(self.convs(torch.cat([dp, fj], 1)), dim=-1)[0]
        return self.act(pooled + identity) if self.use_res else pooled


class _LocalAggregation(nn.Module):
    """The reference's wrapper level (`LocalAggregation`, local_aggregation.py:246-286), kept for its parameter names."""

    def __init__(self, channels, conv_args, norm_args, act_args, group_args, use_res):
        super().__init__()
        self.SA_CONFIG_operator = ConvPool(channels, conv_args, norm_args, act_args, group_args, use_res)

    def forward(self, query_xyz, support_xyz, features, query_idx=None):
        return self.SA_CONFIG_operator(query_xyz, support_xyz, features, query_idx)


class SetAbstractionMSG(nn.Module):
    """pointnetv2.py:17-106 (sampler 'fps')."""

    def __init__(self, stride, radii, nsamples, channel_list, group_args, conv_args=None, norm_args=None,
                 act_args=None, use_res=False, query_as_support=False):
        super().__init__()
        self.stride, self.query_as_support = stride, query_as_support
        channel_list = [list(c) for c in channel_list]
        blocks = []
        for i, (radius, nsample) in enumerate(zip(radii, nsamples)):
            if i > 0 and query_as_support:
                channel_list[i][0] = channel_list[i - 1][-1]
            args = dict(group_args, radius=radius, nsample=nsample)
            blocks.append(_LocalAggregation(channel_list[i], conv_args, norm_args, act_args, args, use_res))
        self.local_aggregations = nn.ModuleList(blocks)

    def forward(self, support_xyz, support_features=None, query_xyz=None):
        idx = None
        if query_xyz is None and self.stride > 1:
            idx = furthest_point_sample(support_xyz.contiguous(), support_xyz.shape[1] // self.stride)
            # coordinates of the picks: the gather operator on the (B,3,N) view (== torch.gather along N)
            query_xyz = gather_operation(support_xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        elif query_xyz is None:
            query_xyz = support_xyz
        outs = []
        for blk in self.local_aggregations:
            new = blk(query_xyz, support_xyz, support_features, query_idx=idx)
            outs.append(new)
            if self.query_as_support:
                support_xyz, support_features, idx = query_xyz, new, None
        return query_xyz, torch.cat(outs, dim=1)


class FeaturePropagation2(nn.Module):
    """pointnetv2.py:108-150."""

    def __init__(self, mlp, norm_args=None, act_args=None):
        super().__init__()
        norm_args = {'norm': 'bn1d'} if norm_args is None else norm_args
        act_args = {'act': 'relu'} if act_args is None else act_args
        self.convs = nn.Sequential(*[convblock(mlp[i], mlp[i + 1], 1, norm_args=norm_args, act_args=act_args)
                                     for i in range(len(mlp) - 1)])

    def forward(self, unknown, known, unknown_feats, known_feats):
        if known is not None:
            up = three_interpolation(unknown, known, known_feats)
        else:
            up = known_feats.expand(*known_feats.shape[:2], unknown.shape[1])
        return self.convs(up if unknown_feats is None else torch.cat([unknown_feats, up], dim=1))
