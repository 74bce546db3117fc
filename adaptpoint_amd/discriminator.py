"""The AdaptPoint discriminator: a group-all PointNet with spectral-normalised layers.

Host-side mirror of `PointDiscriminator1` and its `PointNetSetAbstraction_SpectralNorm` stage
(openpoints/models_adaptpoint/point_discriminator.py:17-73, 129-191; cfg
cfgs/scanobjectnn/pointnext-s_adaptpoint_1.yaml:58-61): per-point 1x1 convolutions
3 -> 64 -> 128 -> 1024 with ReLU, max over the N points, then 1024 -> 512 -> 256 -> num_classes -> 1
with ReLU / dropout 0.4 and a final sigmoid.  No BatchNorm; every weight carries
`torch.nn.utils.parametrizations.spectral_norm` (one power iteration per training-mode forward,
state in the `_u` / `_v` buffers) -- registered through `adaptpoint_amd.spectral`, which evaluates it in three
launches instead of ~13 on the GPU.  It uses none of the extension operators: the "group all"
stage is the identity grouping, so the per-point MLP runs directly on (B,3,N).

Module names and the convolution type (Conv2d with 1x1 kernels) are the reference's, so its
state_dict -- `sa1.mlp_convs.<i>.parametrizations.weight.original`, `..._u`, `..._v`, `fc1...`,
`prob_head.0...` -- loads unchanged (800,671 parameters at num_classes = 15).
"""
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.parametrizations import spectral_norm as _torch_spectral_norm

from . import pointwise
from .spectral import spectral_norm as _fused_spectral_norm    # torch's parametrisation on csrc/spectral.hip


class _GroupAllStage(nn.Module):
    """point_discriminator.py:149-191 with group_all=True: the shared MLP over every point of
    the cloud and a max over them.  `mlp_bns` exists (empty) because the reference registers it."""

    def __init__(self, in_channel, mlp, fused=True):
        super().__init__()
        self.fused = fused
        spectral_norm = _fused_spectral_norm if fused else _torch_spectral_norm
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        for out_channel in mlp:
            self.mlp_convs.append(spectral_norm(nn.Conv2d(in_channel, out_channel, 1)))
            in_channel = out_channel

    def forward(self, xyz):
        """xyz (B,3,N) -> (B,C_last): the reference's (B,3,N,1) layout is (B,3,N) with a unit
        trailing axis; the max over the "nsample" axis (:189) is the max over the points."""
        last = self.mlp_convs[-1]
        if self.fused and pointwise.conv_max_supported(xyz, xyz.shape[1]):
            # every layer on the contraction kernels of csrc/pointwise.hip (each spectral-normalised `weight` is
            # read ONCE: a read in training mode is a power iteration); the last one fused with the pooling
            x = xyz
            for conv in self.mlp_convs[:-1]:
                x = pointwise.conv_bias_act(x, conv.weight, conv.bias, relu=True)
            if pointwise.conv_max_supported(x, last.in_channels):
                return pointwise.conv_max(x, last.weight, last.bias, relu=True)
            return F.relu(last(x.unsqueeze(-1))).amax(dim=2).squeeze(-1)
        x = xyz.unsqueeze(-1)
        for conv in self.mlp_convs[:-1]:
            x = F.relu(conv(x))
        if self.fused and pointwise.conv_max_supported(x.squeeze(-1), last.in_channels):
            # convolution + ReLU + max in one pass: the (B, 1024, N) activation is never written, the backward
            # touches one position per (cloud, channel).  (`last.weight` is read ONCE: every read of a
            # spectral-normalised weight in training mode is a power iteration.)
            return pointwise.conv_max(x.squeeze(-1), last.weight, last.bias, relu=True)
        return F.relu(last(x)).amax(dim=2).squeeze(-1)


class PointDiscriminator1(nn.Module):
    """point_discriminator.py:17-73."""

    def __init__(self, num_classes=40, normal_channel=False, fused=True, **kwargs):
        super().__init__()
        if normal_channel:
            raise NotImplementedError("normal channels are not used by the AdaptPoint configs")
        self.normal_channel = False
        self.sa1 = _GroupAllStage(3, [64, 128, 1024], fused=fused)
        spectral_norm = _fused_spectral_norm if fused else _torch_spectral_norm
        self.fc1 = spectral_norm(nn.Linear(1024, 512))
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = spectral_norm(nn.Linear(512, 256))
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = spectral_norm(nn.Linear(256, num_classes))
        self.prob_head = nn.Sequential(spectral_norm(nn.Linear(num_classes, 1)), nn.Sigmoid())

    def _forward_all_weights_at_once(self, xyz, wn):
        """The same forward with the seven spectral-normalised weights handed in (spectral.normalise_many)."""
        convs = self.sa1.mlp_convs
        x = xyz.permute(0, 2, 1).contiguous()
        for conv, w in zip(convs[:-1], wn[:len(convs) - 1]):
            x = pointwise.conv_bias_act(x, w, conv.bias, relu=True)
        x = pointwise.conv_max(x, wn[len(convs) - 1], convs[-1].bias, relu=True)
        w1, w2, w3, wh = wn[len(convs):]
        x = self.drop1(F.relu(F.linear(x, w1, self.fc1.bias)))
        x = self.drop2(F.relu(F.linear(x, w2, self.fc2.bias)))
        x = F.linear(x, w3, self.fc3.bias)
        return self.prob_head[1](F.linear(x, wh, self.prob_head[0].bias))

    def forward(self, xyz):
        """xyz (B,N,3) -> probability of "real" (B,1)."""
        if self.sa1.fused and xyz.is_cuda and len(self.sa1.mlp_convs) == 3:
            x0 = xyz.permute(0, 2, 1)
            if pointwise.conv_max_supported(x0, 3) and self.sa1.mlp_convs[-1].in_channels <= 128:
                # every layer's spectral normalisation (a power iteration each in training mode) in one set of launches
                from .spectral import normalise_many
                wn = normalise_many([*self.sa1.mlp_convs, self.fc1, self.fc2, self.fc3, self.prob_head[0]])
                if wn is not None:
                    return self._forward_all_weights_at_once(xyz, wn)
        x = self.sa1(xyz.permute(0, 2, 1).contiguous())
        x = self.drop1(F.relu(self.fc1(x)))
        x = self.drop2(F.relu(self.fc2(x)))
        return self.prob_head(self.fc3(x))
