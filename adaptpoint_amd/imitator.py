"""The AdaptPoint imitator's predictor network `SAComponent` over the gfx950 operators.

Host-side mirror of `SAComponent` and its sub-modules
(openpoints/models_adaptpoint/generator_component4_15.py:92-104, 330-366, 533-587, 588-712): an
embedding, four `ConvBNReLU1D + PointsetGrouper` stages (SURVEY section 8f row 1,
adaptpoint_amd.pointset), four feature-propagation decoders (three_nn + three_interpolate, the
operators of section 8a), the anchor head, and the point-masking branch with
`Anchor_selfattention` over all N points (row 2, adaptpoint_amd.attention).  Sub-module names and
nesting match the reference, so its state_dict loads unchanged.  The deformation that consumes
`prob` / `masking` (AdaptPoint_Augmentor, :115-327) is plain elementwise PyTorch and stays the
reference's.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import AnchorSelfAttention
from . import graphs
from .graphs import mark, mark_grad
from . import pointwise
from .layers import inverse_distance_weights, three_interpolate, three_interpolation, three_nn
from .pointset import PointsetGrouper


# The four grouper stages sample from each other's samples: deeper stages through the nested sampler (a copy of the previous
# stage's first picks wherever every arg-max was unique).  False: every stage runs the full sampler, as the reference does.
NESTED_SAMPLING = True


def index_points(points, idx):
    """points (B,N,C), idx (B,S[,K]) -> (B,S[,K],C)   (:74-90)."""
    B = points.shape[0]
    view = [B] + [1] * (idx.dim() - 1)
    return points[torch.arange(B, device=points.device).view(view), idx, :]


def knn_point(nsample, xyz, new_xyz):
    """(:59-72) -- the matmul-form squared distance and an unsorted top-k, as the reference."""
    dist = -2 * torch.matmul(new_xyz, xyz.permute(0, 2, 1))
    dist += torch.sum(new_xyz ** 2, -1).unsqueeze(-1)
    dist += torch.sum(xyz ** 2, -1).unsqueeze(1)
    return torch.topk(dist, nsample, dim=-1, largest=False, sorted=False)[1]


class ConvBNReLU1D(nn.Module):
    """(:92-104).  `fused`: the layer on csrc/pointwise.hip (adaptpoint_amd.pointwise) where its shape and mode
    are served -- kernel 1, no bias, float32 on the GPU -- and as the reference's three modules otherwise."""

    def __init__(self, in_channels, out_channels, kernel_size=1, bias=True, activation='relu', fused=True):
        super().__init__()
        if activation.lower() != 'relu':
            raise NotImplementedError("only the ReLU activation is instantiated by the imitator")
        self.act = nn.ReLU(inplace=True)
        self.net = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size, bias=bias),
                                 nn.BatchNorm1d(out_channels), self.act)
        self.fused = fused

    def forward(self, x):
        if self.fused and pointwise.supported(x, self.net[0], self.net[1]):
            return pointwise.conv_bn_act(x, self.net[0], self.net[1], relu=True)
        return self.net(x)


class PointNetFeaturePropagation(nn.Module):
    """(:330-366)"""

    def __init__(self, in_channel, out_channel, blocks=1, groups=1, res_expansion=1.0, bias=False,
                 activation='relu', fused=True):
        super().__init__()
        self.fuse = ConvBNReLU1D(in_channel, out_channel, 1, bias=bias, fused=fused)

    def forward(self, xyz1, xyz2, points1, points2, nearest=None):
        """nearest: (indices, weights) of `three_nn` + `inverse_distance_weights` computed ahead (index work)."""
        if nearest is None:
            interpolated = three_interpolation(xyz1, xyz2, points2)
        else:
            interpolated = three_interpolate(points2.contiguous(), nearest[0], nearest[1])
        new_points = interpolated if points1 is None else torch.cat([points1, interpolated], dim=1)
        return self.fuse(new_points)


class Producefactor(nn.Module):
    """(:533-587)"""

    def __init__(self, kneighbors, out_channels):
        super().__init__()
        self.keighbors = kneighbors
        self.out_channels = out_channels
        self.global_layer = nn.Sequential(nn.Conv1d(3, out_channels, 1, bias=False), nn.BatchNorm1d(out_channels))
        self.prob_head = nn.Sequential(nn.Conv1d(out_channels * 2, 3 * 3, 1, bias=False), nn.BatchNorm1d(3 * 3))
        self.anchor_selfattention = AnchorSelfAttention(dim=out_channels, head_num=4)

    def forward(self, a_points, sa_x, sa_xyz, xyz_raw, idx_knn=None):
        num_anchor = a_points.shape[1]
        if idx_knn is None:
            idx_knn = knn_point(self.keighbors, sa_xyz, a_points)
        local_feat = torch.max(index_points(sa_x, idx_knn), dim=2)[0]
        local_feat = local_feat + self.anchor_selfattention(x=local_feat, xyz=a_points)
        on = a_points.is_cuda
        global_feat = pointwise.conv_then_bn(a_points.permute(0, 2, 1).contiguous(), self.global_layer, allow=on).permute(0, 2, 1)
        global_feat = torch.max(global_feat, dim=1, keepdim=True)[0]
        feat = torch.cat([local_feat, global_feat.repeat(1, num_anchor, 1)], dim=-1)
        return pointwise.conv_then_bn(feat.permute(0, 2, 1).contiguous(), self.prob_head, allow=on).permute(0, 2, 1)


class SAComponent(nn.Module):
    """(:588-712)"""

    def __init__(self, in_channel=3, embed_dim=64, res_expansion=1.0, activation="relu", bias=False,
                 normalize="anchor", dim_expansion=(2, 2, 2, 2), radii=(0.1, 0.2, 0.4, 0.8),
                 k_neighbors=(24, 24, 24, 24), reducers=(2, 2, 2, 2), fused=True, **kwargs):
        super().__init__()
        self.stages = len(dim_expansion)
        self.embedding = ConvBNReLU1D(in_channel, embed_dim, bias=bias, activation=activation, fused=fused)
        self.extract_feat_list = nn.ModuleList()
        self.pointset_grouper_list = nn.ModuleList()
        last = embed_dim
        channels = [embed_dim]
        for i in range(self.stages):
            out = last * dim_expansion[i]
            self.extract_feat_list.append(ConvBNReLU1D(last, out, kernel_size=1, bias=bias, activation=activation,
                                                       fused=fused))
            self.pointset_grouper_list.append(PointsetGrouper(channel=out, reduce=reducers[i],
                                                              kneighbors=k_neighbors[i], radi=radii[i],
                                                              normalize=normalize, fused=fused))
            last = out
            channels.append(out)
        self.head = Producefactor(kneighbors=24, out_channels=last)
        self.decode_list = nn.ModuleList(
            PointNetFeaturePropagation(channels[-(i + 1)] + channels[-(i + 2)], channels[-(i + 2)],
                                       blocks=1, groups=1, res_expansion=res_expansion, bias=bias,
                                       activation=activation, fused=fused) for i in range(self.stages))
        self.localfeat_mask_selfattention = AnchorSelfAttention(dim=embed_dim, head_num=4, fused=fused)
        self.extract_local_feat_masking = nn.Sequential(nn.Conv1d(embed_dim, 3, 1, bias=False), nn.BatchNorm1d(3))
        self.extract_global_feat_masking = nn.Sequential(nn.Conv1d(last, 3, 1, bias=False), nn.BatchNorm1d(3))
        self.fuse_masking = nn.Sequential(nn.Conv1d(6, 2, 1, bias=False), nn.BatchNorm1d(2))

    def masking_logits(self, x0, x_last, xyz):
        """(:704-713) up to the Gumbel soft-max: (B,N,2)."""
        N = x0.shape[-1]
        x0t = pointwise.transpose12(x0)
        local = self.localfeat_mask_selfattention(x=x0t, xyz=xyz) + x0t
        on = x0.is_cuda and self.embedding.fused
        masking_local = pointwise.conv_then_bn(pointwise.transpose12(local), self.extract_local_feat_masking, allow=on)
        masking_global = torch.max(pointwise.conv_then_bn(x_last, self.extract_global_feat_masking, allow=on), dim=2,
                                   keepdim=True)[0]
        masking = torch.cat([masking_local, masking_global.repeat(1, 1, N)], dim=1)
        return pointwise.conv_then_bn(masking, self.fuse_masking, allow=on).permute(0, 2, 1)

    def forward(self, x, a_index=None, return_logits=False):
        """x (B,N,3), a_index (B,M) anchor indices -> prob (B,M,9), masking (B,N,2) one-hot.

        Under `graphs.overlapping()` (and for coordinates that need no gradient) the step's independent parts run as
        a second lane (ONE side stream): the anchor head -- which reads the last stage's output, not the decoders' --
        beside the decoders and the masking branch; the backward pass inherits the split."""
        a_points = index_points(x, a_index)
        xyz = x
        dev = x.device
        overlap = graphs.overlap_enabled() and x.is_cuda and not x.requires_grad
        f = self.embedding(x.permute(0, 2, 1).contiguous())
        mark("imitator: embedding done")
        mark_grad(f, "backward: embedding output gradient formed")
        xyz_list, x_list = [xyz], [f]
        ties = None
        for i in range(self.stages):
            f = self.extract_feat_list[i](f)
            part = None
            if NESTED_SAMPLING and xyz.is_cuda and not xyz.requires_grad:
                # stage i + 1 samples from stage i's samples: the nested sampler (same picks; csrc/fps.hip, NEST)
                part = self.pointset_grouper_list[i].index(xyz, ties=ties, nested=True)
                ties = part[3] if len(part) > 3 else None
            xyz, f = self.pointset_grouper_list[i](xyz, pointwise.transpose12(f), index=part)
            xyz_list.append(xyz)
            x_list.append(f)
            mark(f"imitator: stage {i + 1} done")
            mark_grad(f, f"backward: imitator stage {i + 1} output gradient formed")
        idx_knn = None
        if overlap:
            s_head = graphs.fork(graphs.LANE2, dev, a_points, f, xyz)
            with torch.cuda.stream(s_head):
                prob = self.head(a_points=a_points, sa_x=f.permute(0, 2, 1), sa_xyz=xyz, xyz_raw=x, idx_knn=idx_knn)
                mark("imitator: anchor head done (side stream)")
        for i in range(self.stages):
            x_list[-(i + 2)] = self.decode_list[i](xyz1=xyz_list[-(i + 2)], xyz2=xyz_list[-(i + 1)],
                                                   points1=x_list[-(i + 2)], points2=x_list[-(i + 1)])
        mark("imitator: decoders done")
        mark_grad(x_list[0], "backward: decoders' output gradient formed (masking branch done)")
        if not overlap:
            prob = self.head(a_points=a_points, sa_x=f.permute(0, 2, 1), sa_xyz=xyz, xyz_raw=x, idx_knn=idx_knn)
            mark("imitator: anchor head done")
        mark_grad(prob, "backward: anchor head output gradient formed")
        logits = self.masking_logits(x_list[0], x_list[-1], xyz_list[0])
        mark("imitator: masking logits done")
        if overlap:
            graphs.join(s_head, prob)
        if return_logits:
            return prob, logits
        return prob, self.hard_mask(logits)

    @staticmethod
    def hard_mask(logits, expo=None, tau=0.1):
        """(:714) `F.gumbel_softmax(logits, tau=0.1, hard=True)`: one-hot forward, soft-max
        gradient.  `expo` (B,N,2): the Exp(1) samples behind the Gumbel noise; drawn from the
        logits' device generator when absent -- exactly what torch.nn.functional does."""
        if expo is None:
            return F.gumbel_softmax(logits, tau=tau, hard=True, eps=1e-10, dim=-1)
        y = ((logits - expo.log()) / tau).softmax(-1)
        hard = torch.zeros_like(logits).scatter_(-1, y.argmax(-1, keepdim=True), 1.0)
        return hard - y.detach() + y
