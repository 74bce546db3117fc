"""PointNeXt-S classifier over the gfx950 set-abstraction blocks.

Host-side mirror of the model `cfgs/scanobjectnn/pointnext-s.yaml:5-36` builds in the reference:
`BaseCls` (openpoints/models/classification/cls_base.py:13-39) = `PointNextEncoder`
(openpoints/models/backbone/pointnext.py:312-441; blocks [1]*6, so every stage is one
SetAbstraction and no InvResMLP exists) + `ClsHead` (cls_base.py:78-136).  Module nesting and
names match the reference (`encoder.encoder.<stage>.0....`, `prediction.head.<i>....`), so a
reference state_dict loads unchanged.  Stage 1 (1024 -> 512 points, 32 -> 64 channels) has the
shape the fused kernels cover; the other stages run the unfused extension operators + PyTorch.
"""
import torch
import torch.nn as nn

from .set_abstraction import MAX_SAMPLE_SEQ_POINTS, SetAbstraction

# forward_cls_feat without a pyramid handed in builds one itself (nested sampler: blocks 2-4 sample from the previous
# block's samples, which FPS returns as a prefix).  False: every block runs its own full sampler, as the reference does.
NESTED_PYRAMID = True


class PointNextEncoderS(nn.Module):
    """pointnext.py:338-441 for sa_layers=2, sa_use_res=True, one block per stage."""

    def __init__(self, in_channels=4, width=32, strides=(1, 2, 2, 2, 2, 1), radius=0.15,
                 radius_scaling=1.5, nsample=32, fused=False, sync_bn=False):
        super().__init__()
        self.strides = list(strides)
        # _to_full_list (pointnext.py:389-407): the radius grows after every down-sampling stage
        radii, r = [], radius
        for s in self.strides:
            radii.append(r)
            if s != 1:
                r *= radius_scaling
        channels = []
        for s in self.strides:
            if s != 1:
                width *= 2
            channels.append(width)
        stages, cin = [], in_channels
        for i, (s, c) in enumerate(zip(self.strides, channels)):
            is_head = i == 0 and s == 1
            sa = SetAbstraction(cin, c, layers=1 if is_head else 2, stride=s,
                                group_args={'NAME': 'ballquery', 'radius': radii[i], 'nsample': nsample,
                                            'normalize_dp': True},
                                norm_args={'norm': 'bn'}, act_args={'act': 'relu'},
                                conv_args={'order': 'conv-norm-act'}, sampler='fps',
                                feature_type='dp_fj', use_res=True, is_head=is_head,
                                fused=fused, sync_bn=sync_bn)
            stages.append(nn.Sequential(sa))
            cin = c
        self.encoder = nn.Sequential(*stages)
        self.out_channels = channels[-1]
        self.radii = radii

    @torch.no_grad()
    def index_pyramid(self, p0, out=None, events=False):
        """The index stages of ALL blocks -- FPS (+ sampled coordinates) and ball query per
        down-sampling block -- for a batch of coordinates p0 (B,N,3).  They depend on coordinates
        only (block k samples from block k-1's samples), so the whole pyramid can be computed ahead of
        the feature path: on another stream, for the next batch (scripts/bench_pointnext.py --pipeline).
        Returns one `adaptpoint_amd.fused.Sampling` (or None) per block; `out`: buffers to fill.  events: record an
        event behind every block's index stage (`Sampling.ready`); `forward_cls_feat` then waits block by block, so
        a pyramid running on a side stream is consumed level by level instead of as a whole."""
        from . import fused
        res, p = [], p0.contiguous()
        ties = None          # block k + 1 samples from block k's samples: the nested sampler (csrc/fps.hip, NEST)
        for i, stage in enumerate(self.encoder):
            sa = stage[0]
            if sa.is_head or sa.all_aggr:
                res.append(None)
                continue
            smp = fused.sample_and_query(p, p.shape[1] // sa.stride, sa.grouper.radius, sa.grouper.nsample,
                                         out=None if out is None else out[i], geo=sa._resident(), nested=True, ties=ties)
            ties = smp.ties
            # tile map + inverse map of the neighbourhoods, for the blocks on the width-generic kernels
            sa.index_for(smp, p.shape[1], sa.convs[0][0].in_channels - 3, out=smp.index)
            if events:
                from . import graphs
                smp.ready = graphs.ready_event()
            res.append(smp)
            p = smp.new_p
        return res

    def forward_seg_feat(self, p0, f0=None):
        """Every level's (points, features): what the segmentation decoder consumes (pointnext.py:443-453)."""
        if hasattr(p0, 'keys'):
            p0, f0 = p0['pos'], p0.get('x', None)
        if f0 is None:
            f0 = p0.clone().transpose(1, 2).contiguous()
        p, f = [p0], [f0]
        for stage in self.encoder:
            _p, _f = stage[0]([p[-1], f[-1]])
            p.append(_p)
            f.append(_f)
        return p, f

    def forward_cls_feat(self, p0, f0=None, pyramid=None):
        if hasattr(p0, 'keys'):
            p0, f0 = p0['pos'], p0.get('x', None)
        if f0 is None:
            f0 = p0.clone().transpose(1, 2).contiguous()
        if pyramid is None and p0.is_cuda and NESTED_PYRAMID and p0.shape[1] <= MAX_SAMPLE_SEQ_POINTS:
            # (larger clouds: every block samples itself -- the unfused operators' streaming sampler; the nested copy only
            # pays off for N <= 4096 anyway)
            # the index stages of all blocks first (coordinates only; gradients reach p0 through the blocks' own
            # differentiable gathers): deeper levels then cost a copy instead of a sampler chain
            pyramid = self.index_pyramid(p0.detach())
        for i, stage in enumerate(self.encoder):
            smp = None if pyramid is None else pyramid[i]
            if smp is not None and getattr(smp, 'ready', None) is not None:
                torch.cuda.current_stream(p0.device).wait_event(smp.ready)
            p0, f0 = stage[0]([p0, f0], sampling=smp) if smp is not None else stage[0]([p0, f0])
        return f0.squeeze(-1)


class FeaturePropagation(nn.Module):
    """PointNet++ feature propagation (`FeaturePropogation`, pointnext.py:173-226):
    features known at the coarse points p2 are carried to the dense points p1 by inverse-distance
    weighting over the three nearest coarse points (`three_nn` + `three_interpolate`, the operators of
    section 8a), concatenated with the dense level's own features and passed through Conv1d-BN-ReLU
    blocks.  `upsample=False` is the global variant (mean-pooled feature broadcast back).  Sub-module
    names (`convs.<i>.0/1`, `linear1`, `linear2`) are the reference's."""

    def __init__(self, mlp, upsample=True, norm_args=None, act_args=None):
        super().__init__()
        from .set_abstraction import convblock
        norm_args = {'norm': 'bn1d'} if norm_args is None else norm_args
        act_args = {'act': 'relu'} if act_args is None else act_args
        mlp = list(mlp)
        self.upsample = upsample
        if not upsample:
            self.linear2 = nn.Sequential(nn.Linear(mlp[0], mlp[1]), nn.ReLU(inplace=True))
            mlp[1] *= 2
            self.linear1 = nn.Sequential(*[convblock(mlp[i], mlp[i + 1], 1, norm_args=norm_args, act_args=act_args)
                                           for i in range(1, len(mlp) - 1)])
        else:
            self.convs = nn.Sequential(*[convblock(mlp[i], mlp[i + 1], 1, norm_args=norm_args, act_args=act_args)
                                         for i in range(len(mlp) - 1)])

    def forward(self, pf1, pf2=None):
        from .layers import three_interpolation
        if pf2 is None:
            _, f = pf1
            g = self.linear2(f.mean(dim=-1))
            return self.linear1(torch.cat((f, g.unsqueeze(-1).expand(-1, -1, f.shape[-1])), dim=1))
        (p1, f1), (p2, f2) = pf1, pf2
        up = three_interpolation(p1, p2, f2)
        return self.convs(up if f1 is None else torch.cat((f1, up), dim=1))


class PointNextDecoder(nn.Module):
    """`PointNextDecoder` (pointnext.py:461-500) for decoder_layers Conv1d blocks per level: walks the
    encoder's levels from coarse to dense, one FeaturePropagation per level (`decoder.<i>.0`)."""

    def __init__(self, encoder_channel_list, decoder_layers=2, decoder_stages=4, in_channels=3):
        super().__init__()
        chans = list(encoder_channel_list)
        cur = chans[-1]
        skip = chans[:-1]
        if len(skip) < decoder_stages:
            skip.insert(0, in_channels)
        fp = chans[:decoder_stages]
        stages = [None] * len(fp)
        for i in range(-1, -len(fp) - 1, -1):
            stages[i] = nn.Sequential(FeaturePropagation([skip[i] + cur] + [fp[i]] * decoder_layers))
            cur = fp[i]
        self.decoder = nn.Sequential(*stages)
        self.out_channels = fp[-len(fp)]

    def forward(self, p, f):
        f = list(f)
        for i in range(-1, -len(self.decoder) - 1, -1):
            f[i - 1] = self.decoder[i][0]([p[i - 1], f[i - 1]], [p[i], f[i]])
        return f[-len(self.decoder) - 1]


class ClsHead(nn.Module):
    """cls_base.py:78-136 with mlps [512, 256], BatchNorm1d, ReLU, dropout 0.5."""

    def __init__(self, num_classes=15, in_channels=512, mlps=(512, 256), dropout=0.5):
        super().__init__()
        dims = [in_channels] + list(mlps) + [num_classes]
        heads = []
        for i in range(len(dims) - 2):
            heads.append(nn.Sequential(nn.Linear(dims[i], dims[i + 1], bias=False),
                                       nn.BatchNorm1d(dims[i + 1]), nn.ReLU(inplace=True)))
            if dropout:
                heads.append(nn.Dropout(dropout))
        heads.append(nn.Sequential(nn.Linear(dims[-2], dims[-1])))
        self.head = nn.Sequential(*heads)

    def forward(self, x):
        return self.head(x)


class SmoothCrossEntropy(nn.Module):
    """openpoints/loss/build.py:12-66 (label_smoothing > 0 branch, no ignore_index/weight)."""

    def __init__(self, label_smoothing=0.3):
        super().__init__()
        self.label_smoothing = label_smoothing

    def forward(self, pred, gt):
        n_class = pred.size(1)
        one_hot = torch.zeros_like(pred).scatter(1, gt.view(-1, 1), 1)
        one_hot = one_hot * (1 - self.label_smoothing) + (1 - one_hot) * self.label_smoothing / (n_class - 1)
        return -(one_hot * torch.log_softmax(pred, dim=1)).sum(dim=1).mean()


class PointNextSClassifier(nn.Module):
    """BaseCls for cfgs/scanobjectnn/pointnext-s.yaml (+ criterion of cfgs/scanobjectnn/default.yaml:36-38)."""

    def __init__(self, num_classes=15, fused=False, sync_bn=False):
        super().__init__()
        self.encoder = PointNextEncoderS(fused=fused, sync_bn=sync_bn)
        self.prediction = ClsHead(num_classes, self.encoder.out_channels)
        self.criterion = SmoothCrossEntropy(0.3)

    def forward(self, data, pyramid=None):
        return self.prediction(self.encoder.forward_cls_feat(data, pyramid=pyramid))

    def get_logits_loss(self, data, gt, pyramid=None):
        logits = self.forward(data, pyramid=pyramid)
        return logits, self.criterion(logits, gt.long())


def fill_parameters_by_name(model, scale=0.08):
    """Deterministic, construction-order independent weights: every tensor of the state_dict is
    drawn from a generator seeded by its NAME.  Used so that the reference model (imported in the
    build container to make goldens) and this mirror hold identical weights without shipping a
    5 MB state_dict."""
    import zlib
    sd = model.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            if name.endswith('num_batches_tracked'):
                continue
            if name.endswith('running_var'):
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif name.endswith('running_mean'):
                v = 0.1 * torch.randn(t.shape, generator=g)
            elif t.dim() == 1 and name.endswith('weight'):          # norm gamma
                v = 0.75 + 0.5 * torch.rand(t.shape, generator=g)
            elif t.dim() == 1:                                        # biases
                v = 0.05 * torch.randn(t.shape, generator=g)
            else:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (1.0 / fan_in) ** 0.5
            t.copy_(v.to(t.dtype))
    return model
