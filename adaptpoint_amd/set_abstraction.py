"""PointNeXt set-abstraction block over the gfx950 operators.

Host-side mirror of `SetAbstraction` in the reference
(openpoints/models/backbone/pointnext.py:82-170) restricted to what the configs
in scope instantiate (cfgs/scanobjectnn/pointnext-s.yaml:5-36): conv-norm-act
order, BatchNorm, ReLU, FPS sampler, ball-query or group-all grouping, max
pooling, optional residual.  Sub-module names and nesting match the reference
(`convs.<i>.0` conv, `convs.<i>.1` norm, `skipconv.0`), so a reference
state_dict loads unchanged.
"""
import logging

import torch
import torch.nn as nn

from . import layers, pointwise
from .layers import BallGrouper, make_grouper

_log = logging.getLogger("adaptpoint_amd")

# fused=True requests that fall back to the unfused operators, by reason (bench.py reports the
# total; each distinct reason is logged once).
FUSED_FALLBACKS = {}


# True: the width-generic kernels (csrc/sa_wide.hip) also take the 32 -> 32 -> 64 shape that the
# register-resident kernels of csrc/sa_fused.hip specialise in (A/B switch for benchmarks and tests).
PREFER_WIDE = False
MAX_SAMPLE_SEQ_POINTS = 16384   # apn_sa_sample_seq / apn_furthest_point_sampling_xyz: clouds kept in registers / LDS
COMPACT_RESIDENT = True      # the register-resident kernels run over the distinct-hit tile map when a Sampling carries one


def fused_wide_first():
    return PREFER_WIDE


def _note_fallback(reason):
    if reason not in FUSED_FALLBACKS:
        _log.warning("SetAbstraction(fused=True) runs UNFUSED: %s", reason)
    FUSED_FALLBACKS[reason] = FUSED_FALLBACKS.get(reason, 0) + 1


def _ranks():
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _norm(norm_args, channels, dim):
    if norm_args is None:
        return None
    name = norm_args.get('norm', None)
    if name is None:
        return None
    if name in ('bn', 'bn2d', 'bn1d'):
        # create_norm (layers/norm.py): 'bn' resolves by the block's dimension
        return nn.BatchNorm2d(channels) if dim == 2 else nn.BatchNorm1d(channels)
    raise NotImplementedError(f"norm '{name}' is outside the hot-path build")


def _act(act_args):
    if act_args is None:
        return None
    name = act_args.get('act', 'relu')
    if name == 'relu':
        return nn.ReLU(inplace=act_args.get('inplace', True))
    raise NotImplementedError(f"activation '{name}' is outside the hot-path build")


def convblock(cin, cout, dim, norm_args=None, act_args=None, order='conv-norm-act', bias=True):
    """create_convblock1d/2d (layers/conv.py:24-104) for the conv-norm-act order:
    1x1 conv (bias dropped when a norm follows), norm, activation."""
    if order != 'conv-norm-act':
        raise NotImplementedError(f"conv order '{order}' is outside the hot-path build")
    norm = _norm(norm_args, cout, dim)
    conv_cls = nn.Conv2d if dim == 2 else nn.Conv1d
    layers = [conv_cls(cin, cout, 1, bias=bias and norm is None)]
    if norm is not None:
        layers.append(norm)
    act = _act(act_args)
    if act is not None:
        layers.append(act)
    return nn.Sequential(*layers)


class SetAbstraction(nn.Module):
    """pointnext.py:82-170."""

    def __init__(self, in_channels, out_channels, layers=1, stride=1,
                 group_args=None, norm_args=None, act_args=None, conv_args=None,
                 sampler='fps', feature_type='dp_fj', use_res=False, is_head=False, fused=False,
                 sync_bn=False, **kwargs):
        super().__init__()
        group_args = dict(group_args or {'NAME': 'ballquery', 'radius': 0.1, 'nsample': 16})
        norm_args = {'norm': 'bn1d'} if norm_args is None else norm_args
        act_args = {'act': 'relu'} if act_args is None else act_args
        conv_args = dict(conv_args or {})
        self.stride = stride
        self.is_head = is_head
        self.all_aggr = not is_head and stride == 1
        self.use_res = use_res and not self.all_aggr and not self.is_head
        self.feature_type = feature_type
        # fused=True routes group -> conv/BN/ReLU -> conv/BN -> max through csrc/sa_fused.hip
        # when the shapes allow; sync_bn=True all-reduces its BatchNorm statistics.
        self.fused = fused
        self.sync_bn = sync_bn

        mid_channel = out_channels // 2 if stride > 1 else out_channels
        channels = [in_channels] + [mid_channel] * (layers - 1) + [out_channels]
        if feature_type != 'dp_fj':
            raise NotImplementedError("only the 'dp_fj' aggregation (relative positions + neighbour "
                                      "features; the cfgs in scope) is on the hot path")
        channels[0] = in_channels if is_head else 3 + channels[0]

        if self.use_res:
            self.skipconv = (convblock(in_channels, channels[-1], 1)
                             if in_channels != channels[-1] else nn.Identity())
            self.act = _act(act_args)

        dim = 1 if is_head else 2
        convs = []
        for i in range(len(channels) - 1):
            last = i == len(channels) - 2
            convs.append(convblock(channels[i], channels[i + 1], dim,
                                   norm_args=norm_args if not is_head else None,
                                   act_args=None if last and (self.use_res or is_head) else act_args,
                                   **conv_args))
        self.convs = nn.Sequential(*convs)
        if not is_head:
            if self.all_aggr:
                group_args['nsample'] = None
                group_args['radius'] = None
            self.grouper = make_grouper(group_args)
            if sampler.lower() != 'fps':
                raise NotImplementedError("only the FPS sampler is on the hot path")

    def _fused_parts(self):
        """(conv1, bn1, conv2, bn2, relu_after_bn2) if the MLP has the fused kernels' structure."""
        g = self.grouper
        if not (isinstance(g, BallGrouper) and g.normalize_dp and len(self.convs) == 2):
            return None
        blk1, blk2 = self.convs[0], self.convs[1]
        if not (len(blk1) == 3 and isinstance(blk1[1], nn.BatchNorm2d) and isinstance(blk1[2], nn.ReLU)
                and len(blk2) in (2, 3) and isinstance(blk2[1], nn.BatchNorm2d)
                and (len(blk2) == 2 or isinstance(blk2[2], nn.ReLU))):
            return None
        return blk1[0], blk1[1], blk2[0], blk2[1], len(blk2) == 3

    def _resident(self):
        """Whether this block runs on the register-resident fused kernels (32 -> 32 -> 64, K = 32)."""
        if not self.fused or self.is_head or self.all_aggr or fused_wide_first():
            return False
        parts = self._fused_parts()
        if parts is None:
            return False
        w = parts[0].weight
        return w.shape[0] == 32 and w.shape[1] == 35 and parts[2].weight.shape[0] == 64 and self.grouper.nsample == 32

    def sample(self, p, out=None):
        """The block's index stage alone (FPS + ball query; for the register-resident kernels also the
        neighbourhoods' occurrence statistics) -> adaptpoint_amd.fused.Sampling."""
        from . import fused
        return fused.sample_and_query(p, p.shape[1] // self.stride, self.grouper.radius,
                                      self.grouper.nsample, out=out, geo=self._resident())

    def wide_shapes(self, c_in):
        """(uses the width-generic kernels for c_in input channels, has a fused residual branch there)."""
        from . import fused_wide
        parts = self._fused_parts() if (self.fused and not self.is_head and not self.all_aggr) else None
        if parts is None:
            return False, False
        H = parts[0].weight.shape[0]
        resident = c_in == 32 and H == 32 and not fused_wide_first()
        return (not resident) and H in fused_wide.WIDTHS, self._skip_conv1d() is not None and fused_wide.lean(c_in, H)

    def index_for(self, smp, n_points, c_in, out=None):
        """The NeighbourIndex (tile map + inverse map; adaptpoint_amd.fused_wide) of a Sampling of this block:
        index-stage work, to be run where the Sampling is made.  Stored as smp.index and returned."""
        from . import fused_wide
        wide, skip = self.wide_shapes(c_in)
        if not wide:
            # the register-resident kernels take the tile map alone
            if self.fused and not self.is_head and not self.all_aggr and self._fused_parts() is not None and COMPACT_RESIDENT:
                smp.tmap = fused_wide.tile_map(smp.idx, out=smp.tmap)
                # ... and the map's row map, for their backward pass (the rows of g_u stored in point-sorted order)
                B, M = smp.idx.shape[0], smp.idx.shape[1]
                smp.rowmap = fused_wide.row_map(smp.tmap, B, n_points, M, out=getattr(smp, "rowmap", None), fidx=smp.fidx)
            return None
        smp.index = fused_wide.neighbour_index(smp.idx, smp.new_p, n_points, fidx=smp.fidx if skip else None, out=out)
        return smp.index

    def _skip_conv1d(self):
        if not self.use_res:
            return None
        if (isinstance(self.skipconv, nn.Sequential) and len(self.skipconv) == 1
                and isinstance(self.skipconv[0], nn.Conv1d) and isinstance(self.act, nn.ReLU)):
            return self.skipconv[0]
        return None

    def _wide_block(self, p, f, sampling=None):
        """The block through the width-generic kernels (adaptpoint_amd.fused_wide), residual branch and final ReLU
        included when the shape has them fused; None when the configuration is not covered."""
        from . import fused, fused_wide
        parts = self._fused_parts()
        if parts is None or self.all_aggr:
            return None
        conv1, bn1, conv2, bn2, relu_after = parts
        g = self.grouper
        if not fused_wide.supported(p, f, g.nsample, conv1, conv2, bns=(bn1, bn2), npoint=p.shape[1] // self.stride):
            return None
        if sampling is None and p.shape[1] > MAX_SAMPLE_SEQ_POINTS:
            # the fused index stage (apn_sa_sample_seq) keeps a cloud resident: larger clouds take the unfused operators,
            # whose FPS is the streaming sampler (csrc/fps.hip: fps_stream_kernel)
            _note_fallback(f"N={p.shape[1]} > {MAX_SAMPLE_SEQ_POINTS}: index stage beyond the resident samplers")
            return None
        C, H = f.shape[1], conv1.weight.shape[0]
        skip = self._skip_conv1d()
        if self.use_res and skip is None:
            return None
        smp = sampling if sampling is not None else self.sample(p.detach())
        fuse_skip = skip is not None and fused_wide.lean(C, H)
        nbr = smp.index
        if nbr is None or (fuse_skip and nbr.fq is None):
            nbr = fused_wide.neighbour_index(smp.idx, smp.new_p, p.shape[1], fidx=smp.fidx if fuse_skip else None)
        new_p = smp.new_p
        if p.requires_grad:            # the sampled coordinates stay differentiable (the AdaptPoint feedback path)
            new_p = torch.gather(p, 1, smp.fidx.long().unsqueeze(-1).expand(-1, -1, 3))
        if fuse_skip or skip is None:
            out = fused_wide.block(p, new_p, f, nbr, g.radius, conv1, bn1, conv2, bn2, skip_conv=skip,
                                   relu=(skip is not None) or relu_after, sync_bn=self.sync_bn)
            return new_p, out
        pooled = fused_wide.block(p, new_p, f, nbr, g.radius, conv1, bn1, conv2, bn2, relu=False, sync_bn=self.sync_bn)
        identity = pointwise.run_block(torch.gather(f, -1, smp.fidx.long().unsqueeze(1).expand(-1, C, -1)),
                                       self.skipconv)
        return new_p, self.act(pooled + identity)

    def sample_many(self, ps, outs=None):
        """Index stages of several batches, FPS of one sharing its launch with the ball query of
        the previous one (adaptpoint_amd.fused.sample_and_query_many)."""
        from . import fused
        return fused.sample_and_query_many(ps, ps[0].shape[1] // self.stride, self.grouper.radius,
                                           self.grouper.nsample, outs=outs)

    def _fused_block(self, p, f, sampling=None):
        """The whole block (FPS, ball query, grouped MLP, pool, skip, ReLU) through
        adaptpoint_amd.fused, or None when the configuration / shapes are not covered."""
        from . import fused
        parts = self._fused_parts()
        if parts is None or self.all_aggr:
            return None
        conv1, bn1, conv2, bn2, relu_after = parts
        g = self.grouper
        if (not fused.supported(p, f, g.nsample, conv1, conv2, bns=(bn1, bn2),
                               npoint=p.shape[1] // self.stride) or p.shape[1] > MAX_SAMPLE_SEQ_POINTS
                or fused_wide_first()):
            return None
        skip = None
        if self.use_res:
            if not (isinstance(self.skipconv, nn.Sequential) and len(self.skipconv) == 1
                    and isinstance(self.skipconv[0], nn.Conv1d) and isinstance(self.act, nn.ReLU)):
                return None
            skip, relu = self.skipconv[0], True
        else:
            relu = relu_after
        return fused.fused_set_abstraction(p, f, p.shape[1] // self.stride, g.radius, conv1, bn1,
                                           conv2, bn2, skip, relu, sync_bn=self.sync_bn,
                                           sampling=sampling)

    def _fused_forward(self, new_p, p, f, idx=None):
        """max_K convs(cat[dp, f[idx]]) through the fused kernels, or None if unsupported."""
        from . import fused
        parts = self._fused_parts()
        if parts is None:
            return None
        conv1, bn1, conv2, bn2, relu_after = parts
        g = self.grouper
        from . import fused_wide
        if idx is None:
            idx = g.neighbours(new_p, p)
        if fused.supported(p, f, idx, conv1, conv2, bns=(bn1, bn2)) and not fused_wide_first():
            out = fused.grouped_mlp_max(p, new_p, f, idx, g.radius, conv1, bn1, conv2, bn2,
                                        sync_bn=self.sync_bn)
        elif fused_wide.supported(p, f, idx, conv1, conv2, bns=(bn1, bn2)):
            out = fused_wide.grouped_mlp_max(p, new_p, f, idx, g.radius, conv1, bn1, conv2, bn2,
                                             sync_bn=self.sync_bn)
        else:
            return None
        if relu_after:          # activation after the last BN commutes with the max
            out = self.convs[1][2](out)
        return out

    @staticmethod
    def pool(x):
        return torch.max(x, dim=-1, keepdim=False)[0]

    def forward(self, pf, sampling=None):
        p, f = pf
        if self.is_head:
            for blk in self.convs:                      # the stem: a 1x1 convolution (csrc/pointwise.hip when fused)
                f = pointwise.run_block(f, blk, self.fused)
            return p, f
        if self.fused and not self.all_aggr:
            res = self._fused_block(p, f, sampling)
            if res is None:
                res = self._wide_block(p, f, sampling)
            if res is not None:
                return res
        idx = None
        if self.all_aggr:
            new_p = p
        elif sampling is not None:          # index stage computed ahead (adaptpoint_amd.fused.Sampling)
            picks, idx = sampling.fidx.long(), sampling.idx
            new_p = (torch.gather(p, 1, picks.unsqueeze(-1).expand(-1, -1, 3)) if p.requires_grad
                     else sampling.new_p)
        else:
            picks = layers.furthest_point_sample(p, p.shape[1] // self.stride).long()
            new_p = torch.gather(p, 1, picks.unsqueeze(-1).expand(-1, -1, 3))
        identity = None
        if self.use_res:                 # the skip branch sees the sampled points' own features
            identity = self.skipconv(torch.gather(f, -1, picks.unsqueeze(1).expand(-1, f.shape[1], -1)))
        pooled = self._fused_forward(new_p, p, f, idx) if (self.fused and not self.all_aggr) else None
        if pooled is None:
            if self.fused and not self.all_aggr:
                why = (f"C_in={f.shape[1]} -> {[c[0].out_channels for c in self.convs]}, "
                       f"K={getattr(self.grouper, 'nsample', None)}: no fused kernel for this shape")
                if self.sync_bn and _ranks() > 1:
                    # workloads.sync_batchnorm_ left this block's BatchNorms unconverted because the block exchanges its own
                    # sums; the composed path below would normalise with RANK-LOCAL statistics -- silently not the reference's
                    # SyncBatchNorm (train_autoaug.py:275-282).  Loud instead.
                    raise RuntimeError("SetAbstraction(fused=True, sync_bn=True) cannot run its fused kernels (" + why +
                                       ") and its BatchNorm modules are plain ones: convert them "
                                       "(adaptpoint_amd.dp.convert_sync_batchnorm) or build the block with fused=False")
                _note_fallback(why)
            dp, fj = self.grouper(new_p, p, f, idx) if idx is not None else self.grouper(new_p, p, f)
            x = torch.cat([dp, fj], 1)                                   # 'dp_fj' (group.py:325-326)
            if self.fused and self.all_aggr and x.shape[2] == 1:
                # group-all: the "grouped" tensor is (B, C, 1, N) = the points themselves; its 1x1 layers run on
                # the per-point contraction kernels
                x = x.squeeze(2)
                for blk in self.convs:
                    x = pointwise.run_block(x, blk)
                pooled = self.pool(x).unsqueeze(-1)
            else:
                pooled = self.pool(self.convs(x))
        if identity is not None:
            pooled = self.act(pooled + identity)
        return new_p, pooled
