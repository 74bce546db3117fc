"""Regenerates the fixtures in tests/golden/ (run in the BUILD container only).

    python tests/golden/make_golden.py

It imports the reference's Python from /root/reference *in memory* (nothing is
copied), stubbing the third-party packages this image lacks and the two compiled
modules, and uses two kinds of reference code to pin the oracle:

  * the reference's independent pure-torch restatements
    (openpoints/models/backbone/pointmlp.py:85-128 farthest_point_sample /
    query_ball_point, openpoints/models/layers/group.py:120-137
    torch_grouping_operation), on inputs where their semantics coincide with the
    CUDA kernels' (start index forced to 0, no exact ties, no point at distance
    exactly r) -- they must agree with oracle/ on every index;
  * the reference's own modules (QueryAndGroup, SetAbstraction of
    openpoints/models/backbone/pointnext.py) executed on CPU with the five
    extension-backed symbols monkey-patched to oracle/ -- their outputs and
    gradients become module-level goldens for the host-side mirror in
    adaptpoint_amd/.

Every index golden is also required to be identical under all four
squared-distance roundings of the oracle (the CUDA compiler's contraction is not
observable here), else the seed is rejected.

Fixtures store OUTPUTS (and small state_dicts); inputs are regenerated from
seeds by tests/golden_inputs.py.

    python tests/golden/make_golden.py [all | pointnet2 | adaptpoint | pins | classifier]

`pointnet2` writes pointnet2_golden.npz (G1-G8: operators, SetAbstraction, the PointNeXt-S
classifier, the imitator's grouper / attention / predictor network); `adaptpoint` writes
adaptpoint_golden.npz (G9-G13: the generator with pinned random draws, the discriminator, one
`train_gan` iteration and one `train_one_epoch` iteration replayed statement for statement over
the reference's modules, three_interpolation and KNNGroup).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
import golden_inputs as GI  # noqa: E402

REF = "/root/reference"


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _MM:
        """multimethod stand-in: keeps every overload, dispatches dict vs list."""
        def __init__(self, fn):
            self.fns = [fn]
            self.__name__ = fn.__name__
        def register(self, fn):
            self.fns.append(fn)
            return self
        def __get__(self, obj, objtype=None):
            def call(*a, **k):
                arg = a[0] if a else None
                fn = self.fns[0] if isinstance(arg, dict) or len(self.fns) == 1 else self.fns[-1]
                return fn(obj, *a, **k)
            return call
    _registry = {}
    def multimethod(fn):
        key = fn.__qualname__
        if key in _registry:
            return _registry[key].register(fn)
        _registry[key] = _MM(fn)
        return _registry[key]
    mod("multimethod", multimethod=multimethod)
    mod("termcolor", colored=lambda s, *a, **k: s)
    mod("shortuuid", uuid=lambda: "0")
    mod("wandb")
    mod("h5py")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in {**(d or {}), **kw}.items():
                self[k] = v
        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)
        __setattr__ = __setitem__
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
    mod("easydict", EasyDict=EasyDict)
    tb = mod("torch.utils.tensorboard", SummaryWriter=object)
    torch.utils.tensorboard = tb
    mod("pointnet2_batch_cuda")
    mod("chamfer")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class _OracleOps:
    """autograd-capable CPU stand-ins for the extension-backed symbols, built on oracle/."""

    @staticmethod
    def furthest_point_sample(xyz, npoint):
        return _t(O.furthest_point_sampling(xyz.detach().numpy(), npoint))

    @staticmethod
    def ball_query(radius, nsample, xyz, new_xyz):
        return _t(O.ball_query(radius, nsample, xyz.detach().numpy(), new_xyz.detach().numpy()))

    class _Group(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, idx):
            ctx.save_for_backward(idx)
            ctx.n = features.shape[2]
            return _t(O.group_points(features.detach().numpy(), idx.numpy()))

        @staticmethod
        def backward(ctx, g):
            (idx,) = ctx.saved_tensors
            return _t(O.group_points_grad(g.contiguous().numpy(), idx.numpy(), ctx.n)), None

    @staticmethod
    def three_nn(unknown, known):
        d2, idx = O.three_nn(unknown.detach().numpy(), known.detach().numpy())
        return torch.sqrt(_t(d2)), _t(idx)

    class _Interp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, idx, weight):
            ctx.save_for_backward(idx, weight)
            ctx.m = features.shape[2]
            return _t(O.three_interpolate(features.detach().numpy(), idx.numpy(),
                                          weight.detach().numpy()))

        @staticmethod
        def backward(ctx, g):
            idx, w = ctx.saved_tensors
            return _t(O.three_interpolate_grad(g.contiguous().numpy(), idx.numpy(),
                                               w.detach().numpy(), ctx.m)), None, None


_imported = []


def import_reference():
    if _imported:
        return _imported[0]
    _stub_modules()
    sys.path.insert(0, REF)
    import openpoints.models.layers.group as ref_group
    import openpoints.models.backbone.pointnext as ref_pointnext
    import openpoints.models.backbone.pointmlp as ref_pointmlp
    ref_group.ball_query = _OracleOps.ball_query
    ref_group.grouping_operation = _OracleOps._Group.apply
    ref_pointnext.furthest_point_sample = _OracleOps.furthest_point_sample
    _imported.append((ref_group, ref_pointnext, ref_pointmlp))
    return _imported[0]


def all_variants_equal(fn):
    ref = fn(O.DIST_PINNED)
    for v in O.ALL_DIST_VARIANTS:
        out = fn(v)
        if isinstance(ref, tuple):
            ok = all(np.array_equal(a, b) for a, b in zip(ref, out))
        else:
            ok = np.array_equal(ref, out)
        if not ok:
            raise SystemExit(f"rounding variant {v} changes an index golden: reject the seed")
    return ref


def main():
    ref_group, ref_pointnext, ref_pointmlp = import_reference()
    out = {}

    # ---- G1: config 1 (BASELINE.json configs[0]) ---------------------------------
    xyz = GI.config1_xyz()                                   # (2,1024,3)
    fps512 = all_variants_equal(lambda v: O.furthest_point_sampling(xyz, 512, v))
    q512 = GI.take_points(xyz, fps512)
    bq1 = all_variants_equal(lambda v: O.ball_query(0.15, 32, xyz, q512, v))
    fps256 = all_variants_equal(lambda v: O.furthest_point_sampling(q512, 256, v))
    q256 = GI.take_points(q512, fps256)
    r2 = 0.15 * 1.5
    bq2 = all_variants_equal(lambda v: O.ball_query(r2, 32, q512, q256, v))
    out.update(g1_fps512=fps512, g1_bq_r015=bq1, g1_fps256=fps256, g1_bq_r0225=bq2)

    # pin against the reference's pure-torch restatements (pointmlp.py:85-128)
    real_randint = torch.randint
    torch.randint = lambda *a, **k: torch.zeros(a[2], dtype=k.get("dtype", torch.long))
    try:
        ref_fps = ref_pointmlp.farthest_point_sample(_t(xyz), 512).numpy()
    finally:
        torch.randint = real_randint
    assert np.array_equal(ref_fps, fps512), "oracle FPS != reference pure-torch FPS"
    ref_bq = ref_pointmlp.query_ball_point(0.15, 32, _t(xyz), _t(q512)).numpy()
    has_hit = ref_bq[..., 0] < xyz.shape[1]
    assert has_hit.all(), "config-1 has empty balls; pick the comparison accordingly"
    assert np.array_equal(ref_bq, bq1), "oracle ball_query != reference pure-torch query_ball_point"
    print("G1: oracle == reference pure-torch FPS / query_ball_point on config 1")

    # ---- G2: per-op float paths ---------------------------------------------------
    feats = GI.seeded_normal((2, 32, 1024), seed=11)
    grouped = O.group_points(feats, bq1)
    ref_grouped = ref_group.torch_grouping_operation(_t(feats), _t(bq1).long()).numpy()
    assert np.array_equal(grouped, ref_grouped), "oracle group_points != torch_grouping_operation"
    print("G2: oracle group_points == reference torch_grouping_operation")
    gout = GI.seeded_normal((2, 32, 512, 32), seed=12)
    out["g2_group_grad"] = O.group_points_grad(gout, bq1, 1024)
    out["g2_gather"] = O.gather_points(feats, fps512)
    out["g2_gather_grad"] = O.gather_points_grad(GI.seeded_normal((2, 32, 512), seed=13), fps512, 1024)
    i3 = all_variants_equal(lambda v: O.three_nn(xyz, q512, v)[1])   # indices: every rounding
    d2 = O.three_nn(xyz, q512)[0]                                    # distances: pinned rounding
    out.update(g2_three_nn_dist2=d2, g2_three_nn_idx=i3)
    w = GI.three_nn_weights(d2)
    f512 = GI.seeded_normal((2, 64, 512), seed=14)
    out["g2_three_interp"] = O.three_interpolate(f512, i3, w)
    out["g2_three_interp_grad"] = O.three_interpolate_grad(
        GI.seeded_normal((2, 64, 1024), seed=15), i3, w, 512)

    # ---- G3: ties, ragged sizes, empty balls -----------------------------------
    for name, cloud, m in GI.tie_cases():
        out[f"g3_fps_{name}"] = all_variants_equal(
            lambda v, c=cloud, mm=m: O.furthest_point_sampling(c, mm, v))
    tiny_q = GI.take_points(xyz, fps512)[:, :64]
    out["g3_bq_empty"] = all_variants_equal(lambda v: O.ball_query(1e-4, 16, xyz, tiny_q + 0.5, v))
    out["g3_bq_k8_big"] = all_variants_equal(lambda v: O.ball_query(0.6, 8, xyz, tiny_q, v))
    kd2, kidx = O.three_nn(xyz[:, :50], xyz[:, :2])
    out.update(g3_three_nn_m2_dist2=kd2, g3_three_nn_m2_idx=kidx)

    # ---- G4: module level, reference modules over oracle ops ----------------------
    torch.manual_seed(0)
    from easydict import EasyDict
    sa = ref_pointnext.SetAbstraction(
        32, 64, layers=2, stride=2,
        group_args=EasyDict(NAME='ballquery', radius=0.15, nsample=32, normalize_dp=True),
        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
        sampler='fps', feature_type='dp_fj', use_res=True)
    sa.train()
    p = _t(GI.unit_sphere_cloud(2, 1024, seed=3))
    f = _t(GI.seeded_normal((2, 32, 1024), seed=4)).requires_grad_(True)
    state = {k: v.detach().clone().numpy() for k, v in sa.state_dict().items()}
    new_p, fo = sa([p, f])
    wts = _t(GI.seeded_normal(tuple(fo.shape), seed=5))
    (fo * wts).sum().backward()
    out["g4_sa_new_p"] = new_p.detach().numpy()
    out["g4_sa_out"] = fo.detach().numpy()
    out["g4_sa_grad_f"] = f.grad.numpy()
    for k, v in state.items():
        out["g4_sa_state/" + k] = v
    for k, prm in sa.named_parameters():
        out["g4_sa_grad/" + k] = prm.grad.numpy()

    qg = ref_group.QueryAndGroup(0.15, 32, normalize_dp=True)
    dp, fj = qg(_t(q512), _t(xyz), _t(feats))
    out["g4_qg_dp"] = dp.numpy()
    out["g4_qg_fj_checksum"] = np.array([fj.double().sum().item(), fj.double().abs().sum().item()])

    # ---- G5: the full PointNeXt-S classifier (BASELINE configs[2]) ---------------------
    import openpoints.models.classification.cls_base as ref_cls
    import openpoints.models.backbone.pointnext as ref_pn
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    enc = ref_pn.PointNextEncoder(in_channels=4, width=32, blocks=[1, 1, 1, 1, 1, 1],
                                  strides=[1, 2, 2, 2, 2, 1], sa_layers=2, sa_use_res=True,
                                  radius=0.15, radius_scaling=1.5, nsample=32, expansion=4,
                                  aggr_args={'feature_type': 'dp_fj', 'reduction': 'max'},
                                  group_args=EasyDict(NAME='ballquery', normalize_dp=True),
                                  conv_args={'order': 'conv-norm-act'}, act_args={'act': 'relu'},
                                  norm_args={'norm': 'bn'})
    head = ref_cls.ClsHead(num_classes=15, in_channels=enc.out_channels, mlps=[512, 256],
                           norm_args={'norm': 'bn1d'})
    class _Cls(torch.nn.Module):          # BaseCls without the registry/config machinery
        def __init__(self):
            super().__init__()
            self.encoder, self.prediction = enc, head
        def forward(self, data):
            return self.prediction(self.encoder.forward_cls_feat(data))
    ref_model = fill_parameters_by_name(_Cls())
    assert sum(q.numel() for q in ref_model.parameters()) == 1367119     # pointnext-s.yaml:1-3
    out["g5_state_keys"] = np.array(sorted(ref_model.state_dict().keys()))
    pos = _t(GI.unit_sphere_cloud(2, 1024, seed=31))
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    ref_model.eval()
    with torch.no_grad():
        out["g5_logits_eval"] = ref_model({'pos': pos, 'x': x}).numpy()
    ref_model.train()
    for mod in ref_model.modules():       # dropout off: the only RNG in the forward
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    pos_t = pos.clone()
    xt = x.clone().requires_grad_(True)
    logits = ref_model({'pos': pos_t, 'x': xt})
    logits.square().sum().backward()
    out["g5_logits_train"] = logits.detach().numpy()
    out["g5_grad_x_checksum"] = np.array([xt.grad.double().sum().item(), xt.grad.double().abs().sum().item()])
    out["g5_grad_stem_w"] = ref_model.encoder.encoder[0][0].convs[0][0].weight.grad.numpy()

    # ---- G6: SURVEY 8(f) row 1 -- the imitator's PointsetGrouper, reference module over oracle ops
    import openpoints.models_adaptpoint.generator_component4_15 as ref_gen
    ref_gen.furthest_point_sample = _OracleOps.furthest_point_sample
    ref_gen.ball_query = _OracleOps.ball_query
    grouper = ref_gen.PointsetGrouper(channel=64, reduce=2, kneighbors=24, radi=0.2, normalize="anchor")
    with torch.no_grad():
        grouper.affine_alpha.copy_(_t(GI.seeded_normal((1, 1, 1, 64), seed=61)))   # both signs
        grouper.affine_beta.copy_(_t(GI.seeded_normal((1, 1, 1, 64), seed=62)))
    g6_xyz = _t(GI.unit_sphere_cloud(2, 512, seed=63))
    g6_pts = _t(GI.seeded_normal((2, 512, 64), seed=64)).requires_grad_(True)
    g6_newxyz, g6_out = grouper(g6_xyz, g6_pts)
    g6_w = _t(GI.seeded_normal(tuple(g6_out.shape), seed=65))
    (g6_out * g6_w).sum().backward()
    out["g6_pg_new_xyz"] = g6_newxyz.detach().numpy()
    out["g6_pg_out"] = g6_out.detach().numpy()
    out["g6_pg_grad_points"] = g6_pts.grad.numpy()
    out["g6_pg_grad_alpha"] = grouper.affine_alpha.grad.numpy()
    out["g6_pg_grad_beta"] = grouper.affine_beta.grad.numpy()

    # ---- G7: SURVEY 8(f) row 2 -- the imitator's Anchor_selfattention (plain PyTorch in the reference)
    torch.manual_seed(7)
    att = ref_gen.Anchor_selfattention(dim=64, head_num=4)
    att.train()
    for k_, v_ in att.state_dict().items():
        out["g7_att_state/" + k_] = v_.detach().clone().numpy()
    g7_x = _t(GI.seeded_normal((2, 64, 64), seed=71)).requires_grad_(True)
    g7_xyz = _t(GI.unit_sphere_cloud(2, 64, seed=72))
    g7_out = att(g7_x, g7_xyz)
    (g7_out * _t(GI.seeded_normal(tuple(g7_out.shape), seed=73))).sum().backward()
    out["g7_att_out"] = g7_out.detach().numpy()
    out["g7_att_grad_x"] = g7_x.grad.numpy()
    out["g7_att_grad_qkv_w"] = att.to_qkv.weight.grad.numpy()

    # ---- G8: the imitator's predictor network SAComponent (rows 1, 2 and the 8a operators together)
    ref_gen.three_nn = _OracleOps.three_nn
    ref_gen.three_interpolate = _OracleOps._Interp.apply
    sac = fill_parameters_by_name(ref_gen.SAComponent())
    assert sum(q.numel() for q in sac.parameters()) == 5998062
    sac.train()
    g8_logits = {}
    sac.fuse_masking.register_forward_hook(lambda mod, inp, outp: g8_logits.__setitem__("v", outp))
    g8_x = _t(GI.unit_sphere_cloud(2, 512, seed=81))
    g8_anchor = _t(O.furthest_point_sampling(g8_x.numpy(), 4)).long()
    torch.manual_seed(0)
    g8_prob, g8_mask = sac(g8_x, g8_anchor)
    (g8_prob.sum() + (g8_logits["v"] * _t(GI.seeded_normal(tuple(g8_logits["v"].shape), seed=82))).sum()).backward()
    out["g8_sac_prob"] = g8_prob.detach().numpy()
    out["g8_sac_mask_logits"] = g8_logits["v"].detach().permute(0, 2, 1).numpy()
    out["g8_sac_grad_embed_w"] = sac.embedding.net[0].weight.grad.numpy()
    out["g8_sac_grad_alpha0"] = sac.pointset_grouper_list[0].affine_alpha.grad.numpy()

    path = os.path.join(HERE, "pointnet2_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


# =====================================================================================
# G9-G13: the AdaptPoint half (BASELINE configs[3]) -> tests/golden/adaptpoint_golden.npz
# =====================================================================================
MASK_MARGIN = 5e-3     # every point's Gumbel decision must clear this, so that a GPU run whose mask
                       # logits differ by rounding picks the same mask (seeds that do not are rejected)


def _patch_generator_ops():
    import openpoints.models_adaptpoint.generator_component4_15 as ref_gen
    ref_gen.furthest_point_sample = _OracleOps.furthest_point_sample
    ref_gen.ball_query = _OracleOps.ball_query
    ref_gen.three_nn = _OracleOps.three_nn
    ref_gen.three_interpolate = _OracleOps._Interp.apply
    return ref_gen


def _mask_margin(logits, seed):
    """min over points of |(l0 + g0) - (l1 + g1)| for the Gumbel noise the reference draws FIRST
    after torch.manual_seed(seed) (generator_component4_15.py:714)."""
    torch.manual_seed(seed)
    expo = torch.empty(logits.shape).exponential_()
    z = logits - expo.log()
    return (z[..., 0] - z[..., 1]).abs().min().item()


def _seed_with_margin(run_logits, first_seed):
    """Smallest seed >= first_seed whose mask decisions all clear MASK_MARGIN."""
    for seed in range(first_seed, first_seed + 200):
        if _mask_margin(run_logits(), seed) > MASK_MARGIN:
            return seed
    raise SystemExit("no seed clears the mask margin")


def height_channel(pos):
    """the 4th input channel of ScanObjectNN clouds: y - min y (dataset/scanobjectnn/scanobjectnn.py:95-96)."""
    return pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]


def _reference_classifier():
    from easydict import EasyDict
    import openpoints.models.classification.cls_base as ref_cls
    import openpoints.models.backbone.pointnext as ref_pn
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    enc = ref_pn.PointNextEncoder(in_channels=4, width=32, blocks=[1, 1, 1, 1, 1, 1],
                                  strides=[1, 2, 2, 2, 2, 1], sa_layers=2, sa_use_res=True,
                                  radius=0.15, radius_scaling=1.5, nsample=32, expansion=4,
                                  aggr_args={'feature_type': 'dp_fj', 'reduction': 'max'},
                                  group_args=EasyDict(NAME='ballquery', normalize_dp=True),
                                  conv_args={'order': 'conv-norm-act'}, act_args={'act': 'relu'},
                                  norm_args={'norm': 'bn'})
    head = ref_cls.ClsHead(num_classes=15, in_channels=enc.out_channels, mlps=[512, 256],
                           norm_args={'norm': 'bn1d'})
    import openpoints.loss.build as ref_loss

    class _Cls(torch.nn.Module):          # BaseCls (cls_base.py:13-39) without the registry/config machinery
        def __init__(self):
            super().__init__()
            self.encoder, self.prediction = enc, head
            self.criterion = ref_loss.SmoothCrossEntropy(label_smoothing=0.3)
        def forward(self, data):
            return self.prediction(self.encoder.forward_cls_feat(data))
        def get_logits_loss(self, data, gt):
            logits = self.forward(data)
            return logits, self.criterion(logits, gt.long())
    return fill_parameters_by_name(_Cls())


def main_classifier_b8():
    """G17 / G18: the classifier at B = 8 in TRAINING mode (dropout off), per-parameter gradients -- the sharp model-level
    pins (round 3's G5 training golden sat at B = 2, where the head's BatchNorm1d maps every feature to +-gamma and the
    fused path could only be compared with the build's own unfused mirror).  G17: forward + SmoothCE + backward on
    1024-point clouds (BASELINE configs[2]).  G18: one whole `train_one_epoch` iteration (train_autoaug.py:471-512) on
    2048-point clouds: resampler, forward, loss, backward, clip, AdamW."""
    import_reference()
    out = {}

    def grads_of(model, prefix):
        for name, q in model.named_parameters():
            g = q.grad.detach().numpy().reshape(-1)
            out[f"{prefix}_grad/{name}"] = g[GI.gradient_sample_index(name, g.size)].copy()
            out[f"{prefix}_gnorm/{name}"] = np.array(np.linalg.norm(g.astype(np.float64)))

    def no_dropout(model):
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return model.train()
    # ---- G17
    cls = no_dropout(_reference_classifier())
    pos = _t(GI.unit_sphere_cloud(8, 1024, seed=171))
    x = torch.cat([pos, height_channel(pos)], -1).transpose(1, 2).contiguous().requires_grad_(True)
    target = torch.tensor([5, 14, 0, 7, 3, 11, 9, 2])
    logits, loss = cls.get_logits_loss({'pos': pos, 'x': x}, target)
    loss.backward()
    out.update(g17_logits=logits.detach().numpy(), g17_loss=np.array(loss.item()), g17_grad_x=x.grad.numpy(),
               g17_target=target.numpy())
    grads_of(cls, "g17")
    for name, b in cls.named_buffers():
        if name.endswith("running_mean"):
            out[f"g17_bn/{name}"] = b.numpy().copy()
    print(f"G17: classifier fwd+bwd at B=8 (training mode): loss {loss.item():.5f}")
    # ---- G18
    cls = no_dropout(_reference_classifier())
    pos2k = _t(GI.unit_sphere_cloud(8, 2048, seed=181))
    points = torch.cat([pos2k, height_channel(pos2k)], -1)
    opt = torch.optim.AdamW(cls.parameters(), lr=2e-3, weight_decay=0.05)       # default.yaml:41-46
    npoints, point_all = 1024, 1200
    np.random.seed(18)
    fps_idx = _OracleOps.furthest_point_sample(points[:, :, :3].contiguous(), point_all)
    choice = np.random.choice(point_all, npoints, False)
    fps_idx = fps_idx[:, choice]
    points = torch.gather(points, 1, fps_idx.unsqueeze(-1).long().expand(-1, -1, points.shape[-1]))
    data = {'pos': points[:, :, :3].contiguous(), 'x': points[:, :, :4].transpose(1, 2).contiguous()}
    logits, loss = cls.get_logits_loss(data, target)
    loss.backward()
    grads_of(cls, "g18")                                                        # before clipping
    gnorm = torch.nn.utils.clip_grad_norm_(cls.parameters(), 10, norm_type=2)
    before = {n: q.detach().clone() for n, q in cls.named_parameters()}
    opt.step()
    for name, q in cls.named_parameters():                                      # the step each sampled weight took
        d = (q.detach() - before[name]).numpy().reshape(-1)
        out[f"g18_step/{name}"] = d[GI.gradient_sample_index(name, d.size)].copy()
    out.update(g18_choice=choice.astype(np.int32), g18_logits=logits.detach().numpy(), g18_loss=np.array(loss.item()),
               g18_grad_norm=np.array(gnorm.item()), g18_target=target.numpy())
    for name, b in cls.named_buffers():
        if name.endswith("running_mean"):
            out[f"g18_bn/{name}"] = b.numpy().copy()
    print(f"G18: train_one_epoch iteration at B=8: loss {loss.item():.5f}, grad norm {gnorm.item():.4f}")
    path = os.path.join(HERE, "classifier_b8_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main_adaptpoint():
    import_reference()
    from easydict import EasyDict
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    ref_gen = _patch_generator_ops()
    import openpoints.models_adaptpoint.point_discriminator as ref_dis
    import openpoints.function_adaptpoint.ganloss_cls as ref_fb
    import openpoints.models.layers.upsampling as ref_up
    import openpoints.models.layers.group as ref_group
    out = {}

    # ---- G9: the generator (AdaptPoint_Augmentor.forward, :134-181) with pinned RNG ---------------
    gen = fill_parameters_by_name(ref_gen.AdaptPoint_Augmentor(w_num_anchor=4, w_sigma=0.5, w_R_range=10,
                                                               w_S_range=3, w_T_range=0.25))
    assert sum(q.numel() for q in gen.parameters()) == 5998062
    gen.train()
    g9_x = _t(GI.unit_sphere_cloud(2, 512, seed=91))
    grabbed = {}
    gen.predict_prob_layer.fuse_masking.register_forward_hook(
        lambda mod, inp, outp: grabbed.__setitem__("logits", outp.detach().permute(0, 2, 1).clone()))

    def g9_logits():
        state = {k: v.clone() for k, v in gen.state_dict().items()}     # BN running buffers move in train mode
        with torch.no_grad():
            gen(g9_x)
        gen.load_state_dict(state)
        return grabbed["logits"]
    g9_seed = _seed_with_margin(g9_logits, 9)
    torch.manual_seed(g9_seed)
    _, g9_new = gen(g9_x)
    (g9_new * _t(GI.seeded_normal(tuple(g9_new.shape), seed=92))).sum().backward()
    sac = gen.predict_prob_layer
    out.update(g9_seed=np.array(g9_seed), g9_gen_out=g9_new.detach().numpy(),
               g9_mask_logits=grabbed["logits"].numpy(),
               g9_grad_embed_w=sac.embedding.net[0].weight.grad.numpy(),
               g9_grad_prob_head_w=sac.head.prob_head[0].weight.grad.numpy(),
               g9_grad_mask_local_w=sac.extract_local_feat_masking[0].weight.grad.numpy())
    print(f"G9: generator golden at seed {g9_seed}; masked points: "
          f"{int((g9_new.detach().abs().sum(-1) == 0).sum())} of {g9_new.shape[0] * g9_new.shape[1]}")

    # ---- G10: the discriminator (PointDiscriminator1, point_discriminator.py:17-73) ------------------
    dis = fill_parameters_by_name(ref_dis.PointDiscriminator1(num_classes=15, normal_channel=False))
    assert sum(q.numel() for q in dis.parameters()) == 800671
    out["g10_state_keys"] = np.array(sorted(dis.state_dict().keys()))
    for mod in dis.modules():              # dropout draws from the device generator: off for the goldens
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    dis.eval()
    g10_x = _t(GI.unit_sphere_cloud(2, 512, seed=101))
    with torch.no_grad():
        out["g10_dis_eval"] = dis(g10_x).numpy()
    dis.train()
    xg = g10_x.clone().requires_grad_(True)
    d_out = dis(xg)                        # one power iteration per spectral-normed layer
    (d_out * _t(np.array([[1.0], [-2.0]], np.float32))).sum().backward()
    out.update(g10_dis_train=d_out.detach().numpy(), g10_grad_x=xg.grad.numpy(),
               g10_grad_conv0=dis.sa1.mlp_convs[0].parametrizations.weight.original.grad.numpy(),
               g10_u_fc1=dis.fc1.parametrizations.weight[0]._u.detach().numpy())

    # ---- G11: one train_gan iteration (train_autoaug.py:133-204) at B=2 ------------------------------
    gen = fill_parameters_by_name(ref_gen.AdaptPoint_Augmentor())
    dis = fill_parameters_by_name(ref_dis.PointDiscriminator1(num_classes=15, normal_channel=False))
    for mod in dis.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    cls = _reference_classifier()
    label = torch.tensor([3, 11])
    cfg = EasyDict(criterion_args=dict(NAME='SmoothCrossEntropy', label_smoothing=0.3),
                   adaptpoint_params=dict(hardratio_s=3, hardratio=3), epochs=300,
                   model=dict(in_channels=4))

    class _Quiet:                          # SummaryWriter / Summary stand-ins (ganloss_cls.py:61-64)
        train_iter_num = 0
        def add_scalar(self, *a, **k):
            pass
    opt_g = torch.optim.Adam(gen.parameters(), lr=1e-4, betas=(0.5, 0.999))      # pointnext-s_adaptpoint_1.yaml:63-71
    opt_d = torch.optim.Adam(dis.parameters(), lr=4e-4, betas=(0.5, 0.999))
    bce = torch.nn.BCELoss()
    gen.train(); dis.train(); cls.eval()
    gen.predict_prob_layer.fuse_masking.register_forward_hook(
        lambda mod, inp, outp: grabbed.__setitem__("logits", outp.detach().permute(0, 2, 1).clone()))
    # The anchor head selects 24 of the coarsest level's points per anchor (knn_point, top-k unsorted) and takes a
    # max over their features (generator_component4_15.py:565-567): a near-tie in either is decided by 1e-7-level
    # rounding upstream and switches whole gradient entries (round 2's golden had one: the head gradient had two
    # outcomes, 1e-5 or 2.8e-3 from the reference).  The golden's INPUT is therefore chosen among seeds >= 111
    # such that the 24th / 25th distances differ by > 1e-3 relative and every max over the 24 leads its runner-up by
    # > 2e-5 of the feature scale.
    gen.predict_prob_layer.head.register_forward_pre_hook(
        lambda mod, inp, kw: grabbed.__setitem__("head_in", (inp, kw)), with_kwargs=True)

    def g11_head_margins():
        inp, kw = grabbed["head_in"]
        names = ("a_points", "sa_x", "sa_xyz")
        a_points, sa_x, sa_xyz = [kw[n] if n in kw else inp[i] for i, n in enumerate(names)]
        d = ref_gen.square_distance(a_points, sa_xyz).sort(dim=-1)[0]
        knn_gap = float(((d[..., 24] - d[..., 23]) / d[..., 24]).min())
        feats = ref_gen.index_points(sa_x, ref_gen.knn_point(24, sa_xyz, a_points))          # (B,4,24,C)
        top2 = feats.topk(2, dim=2)[0]
        lead = (top2[:, :, 0] - top2[:, :, 1]) / feats.abs().max()
        # (EXACT ties are structural -- a grouper output whose maximum is the query's own row is exactly its beta, for
        # every such query -- and bit-identical in any implementation: the first index wins in torch and in the
        # kernels alike; what must not occur is a maximum decided by rounding)
        max_gap = float(lead[lead > 0].min())
        return knn_gap, max_gap

    def g11_input(seed):
        pos = _t(GI.unit_sphere_cloud(2, 512, seed=seed))
        return torch.cat([pos, height_channel(pos)], -1)                        # (2,512,4)

    def g11_logits():
        state = {k: v.clone() for k, v in gen.state_dict().items()}
        with torch.no_grad():
            gen(points[:, :, :3].contiguous())
        gen.load_state_dict(state)
        return grabbed["logits"]
    # (8192 maxima over 24 candidates each: the smallest lead of any input is a few 1e-6 of the feature scale --
    # ~50 float32 ulps, against 1e-7-level differences between implementations; the input with the LARGEST smallest
    # lead among seeds 111..399 whose top-24 gap exceeds 1e-3 is taken)
    scan = []
    for cand in range(111, 400):
        points = g11_input(cand)
        g11_logits()
        knn_gap, max_gap = g11_head_margins()
        if knn_gap > 1e-3:
            scan.append((max_gap, knn_gap, cand))
    max_gap, knn_gap, g11_pos_seed = max(scan)
    if max_gap < 2e-6:
        raise SystemExit("G11: no input seed clears the anchor head's margins")
    points = g11_input(g11_pos_seed)
    g11_seed = _seed_with_margin(g11_logits, 11)
    torch.manual_seed(g11_seed)
    # -- the body of the loop, statement for statement (without .cuda(), PointWOLF's dump and logging)
    points_clone = points.clone()
    input_pointcloud = points[:, :, :3].contiguous()
    real_label = torch.full((2, 1), 0.9)
    fake_label = torch.full((2, 1), 0.1)
    _, gen_imgs = gen(input_pointcloud)
    g_loss_raw = bce(dis(gen_imgs), real_label)
    points[:, :, :3] = gen_imgs
    data_fake = {'pos': points[:, :, :3].contiguous(), 'y': label,
                 'x': points[:, :, :4].transpose(1, 2).contiguous()}
    data_real = {'pos': points_clone[:, :, :3].contiguous(), 'y': label,
                 'x': points_clone[:, :, :4].transpose(1, 2).contiguous()}
    feedback = ref_fb.get_feedback_loss_ver1(cfg=cfg, model_pointcloud=cls, data_real=data_real,
                                             data_fake=data_fake, epoch=7, summary=_Quiet(), writer=_Quiet())
    g_loss = g_loss_raw + feedback * 1
    opt_g.zero_grad()
    g_loss.backward()
    g11_grad_embed = gen.predict_prob_layer.embedding.net[0].weight.grad.clone()
    g11_grad_head = gen.predict_prob_layer.head.prob_head[0].weight.grad.clone()
    # (round 4) EVERY parameter's gradient of the generator step, sampled as the B = 8 classifier goldens are
    for name, q in gen.named_parameters():
        if q.grad is not None:
            gq = q.grad.detach().numpy().reshape(-1)
            out[f"g11_gen_grad/{name}"] = gq[GI.gradient_sample_index(name, gq.size)].copy()
            out[f"g11_gen_gnorm/{name}"] = np.array(np.linalg.norm(gq.astype(np.float64)))
    opt_g.step()
    real_loss = bce(dis(input_pointcloud), real_label)
    fake_loss = bce(dis(gen_imgs.detach()), fake_label)
    d_loss = (real_loss + fake_loss) / 2
    opt_d.zero_grad()
    d_loss.backward()
    g11_grad_fc3 = dis.fc3.parametrizations.weight.original.grad.clone()
    for name, q in dis.named_parameters():
        if q.grad is not None:
            gq = q.grad.detach().numpy().reshape(-1)
            out[f"g11_dis_grad/{name}"] = gq[GI.gradient_sample_index(name, gq.size)].copy()
            out[f"g11_dis_gnorm/{name}"] = np.array(np.linalg.norm(gq.astype(np.float64)))
    opt_d.step()
    out.update(g11_seed=np.array(g11_seed), g11_pos_seed=np.array(g11_pos_seed),
               g11_head_margins=np.array([knn_gap, max_gap]), g11_gen=gen_imgs.detach().numpy(),
               g11_losses=np.array([g_loss_raw.item(), feedback.item(), g_loss.item(), d_loss.item()]),
               g11_grad_embed_w=g11_grad_embed.numpy(), g11_grad_prob_head_w=g11_grad_head.numpy(),
               g11_grad_fc3=g11_grad_fc3.numpy(),
               g11_embed_w_after=gen.predict_prob_layer.embedding.net[0].weight.detach().numpy(),
               g11_fc3_after=dis.fc3.parametrizations.weight.original.detach().numpy())
    print(f"G11: joint step golden at input seed {g11_pos_seed} (top-24 gap {knn_gap:.2e}, max gap {max_gap:.2e}), "
          f"draw seed {g11_seed}: g_raw {g_loss_raw.item():.5f} feedback "
          f"{feedback.item():.5f} d {d_loss.item():.5f}")

    # ---- G12: a12 three_interpolation (upsampling.py:92-102) and a14 KNNGroup (group.py:275-320) -----
    ref_up.three_nn = _OracleOps.three_nn
    ref_up.three_interpolate = _OracleOps._Interp.apply
    xyz = GI.config1_xyz()
    known = GI.take_points(xyz, O.furthest_point_sampling(xyz, 256))
    feat = _t(GI.seeded_normal((2, 48, 256), seed=121)).requires_grad_(True)
    up = ref_up.three_interpolation(_t(xyz), _t(known), feat)
    (up * _t(GI.seeded_normal(tuple(up.shape), seed=122))).sum().backward()
    out.update(g12_interp=up.detach().numpy(), g12_interp_grad=feat.grad.numpy())
    knn = ref_group.KNNGroup(nsample=8, relative_xyz=True, normalize_dp=True)
    feats = _t(GI.seeded_normal((2, 16, 1024), seed=123))
    dpk, fjk = knn(_t(known[:, :64]), _t(xyz), feats)
    out.update(g12_knn_dp=dpk.numpy(), g12_knn_fj=fjk.numpy())

    # ---- G13: one train_one_epoch iteration (train_autoaug.py:471-512): resampler + classifier step ---
    import openpoints.models.backbone.pointnext as ref_pn     # furthest_point_sample already patched
    cls = _reference_classifier()
    for mod in cls.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    cls.train()
    pos2k = _t(GI.unit_sphere_cloud(2, 2048, seed=131))
    points = torch.cat([pos2k, height_channel(pos2k)], -1)                    # (2,2048,4)
    target = torch.tensor([5, 14])
    opt = torch.optim.AdamW(cls.parameters(), lr=2e-3, weight_decay=0.05)       # default.yaml:41-46
    npoints, point_all = 1024, 1200
    np.random.seed(13)
    fps_idx = _OracleOps.furthest_point_sample(points[:, :, :3].contiguous(), point_all)
    choice = np.random.choice(point_all, npoints, False)
    fps_idx = fps_idx[:, choice]
    points = torch.gather(points, 1, fps_idx.unsqueeze(-1).long().expand(-1, -1, points.shape[-1]))
    data = {'pos': points[:, :, :3].contiguous(), 'x': points[:, :, :4].transpose(1, 2).contiguous()}
    logits, loss = cls.get_logits_loss(data, target)
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(cls.parameters(), 10, norm_type=2)
    opt.step()
    cls.zero_grad()
    out.update(g13_choice=choice.astype(np.int32), g13_pos=data['pos'].numpy(), g13_x=data['x'].numpy(),
               g13_logits=logits.detach().numpy(), g13_loss=np.array(loss.item()),
               g13_grad_norm=np.array(gnorm.item()),
               g13_head_w_after=cls.prediction.head[-1][0].weight.detach().numpy(),
               g13_bn1_mean_after=cls.encoder.encoder[1][0].convs[0][1].running_mean.numpy())
    print(f"G13: classifier step golden: loss {loss.item():.5f}, grad norm {gnorm.item():.4f}")

    # ---- G14: SURVEY 8f row 4 -- FeaturePropogation and the segmentation decoder (pointnext.py:173-226, 461-500) ----
    import openpoints.models.backbone.pointnext as ref_pn2
    ref_pn2.three_interpolation = ref_up.three_interpolation          # the reference function over the oracle ops
    fp = fill_parameters_by_name(ref_pn2.FeaturePropogation([64 + 32, 32, 32]))
    fp.train()
    p1 = GI.unit_sphere_cloud(2, 512, seed=141)
    p2 = GI.take_points(p1, O.furthest_point_sampling(p1, 128))
    f1 = _t(GI.seeded_normal((2, 32, 512), seed=142)).requires_grad_(True)
    f2 = _t(GI.seeded_normal((2, 64, 128), seed=143)).requires_grad_(True)
    fo = fp([_t(p1), f1], [_t(p2), f2])
    (fo * _t(GI.seeded_normal(tuple(fo.shape), seed=144))).sum().backward()
    out.update(g14_fp_out=fo.detach().numpy(), g14_fp_grad_f1=f1.grad.numpy(), g14_fp_grad_f2=f2.grad.numpy(),
               g14_fp_grad_w0=fp.convs[0][0].weight.grad.numpy())
    fpg = fill_parameters_by_name(ref_pn2.FeaturePropogation([32, 32, 24], upsample=False))
    fpg.train()
    out["g14_fp_global_out"] = fpg([None, _t(GI.seeded_normal((2, 32, 100), seed=145))]).detach().numpy()
    dec = fill_parameters_by_name(ref_pn2.PointNextDecoder(encoder_channel_list=[32, 64, 128, 256, 512], decoder_layers=2,
                                                           decoder_stages=4))
    dec.train()
    out["g14_dec_keys"] = np.array(sorted(dec.state_dict().keys()))
    pl = [GI.unit_sphere_cloud(2, 256, seed=146)]
    for m in (128, 64, 32, 16):
        pl.append(GI.take_points(pl[-1], O.furthest_point_sampling(pl[-1], m)))
    fl = [_t(GI.seeded_normal((2, c, n), seed=147 + i)) for i, (c, n) in enumerate(zip((32, 64, 128, 256, 512), (256, 128, 64, 32, 16)))]
    out["g14_dec_out"] = dec([_t(q) for q in pl], fl).detach().numpy()
    print("G14: FeaturePropogation / PointNextDecoder goldens")

    path = os.path.join(HERE, "adaptpoint_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


def _fp_pyramid(seed):
    """The four feature-propagation levels of the imitator at N = 1024: (unknown, known) clouds."""
    cloud = GI.unit_sphere_cloud(2, 1024, seed=seed)
    levels = [cloud]
    for m in (512, 256, 128, 64):
        levels.append(GI.take_points(levels[-1], O.furthest_point_sampling(levels[-1], m)))
    return [(levels[i], levels[i + 1]) for i in range(4)]


def main_pins():
    """G15 / G16 (round 3): the parts of the oracle that no reference-EXECUTED result pinned so far.

    G15: the interpolation half and the scatter-add gradients, against reference-held pure-torch code --
      * three nearest + inverse-distance interpolation: `square_distance` (curvenet.py:213-222) and the forward of
        `PointNetFeaturePropagation` (curvenet.py:428-455: sort, first three, 1 / (d + 1e-8) weights on the SQUARED
        distances, index_points * weight) -> oracle.three_nn must give the same indices (a case is rejected unless the
        four smallest distances of every point are separated by 1e-4 relative: the matmul form of the distance and
        the kernel's difference form may otherwise order near-ties differently), oracle.three_interpolate with the
        reference's weights the same values (1e-6), and autograd through the reference's forward the same gradient
        as oracle.three_interpolate_grad (1e-6);
      * autograd through `torch_grouping_operation` (group.py:120-137) and through `torch.gather` (what
        `GatherOperation`'s own self-check compares with, subsample.py:176-185) -> oracle.group_points_grad /
        oracle.gather_points_grad (1e-6).
      The reference's outputs are committed (tests/golden/pins_golden.npz) so that the CPU suite re-checks the oracle
      and the GPU suite the kernels against them.
    G16: SURVEY 8f row 4, second half -- PointNet++ through the boundary: the reference's `PointNetSAModuleMSG` (one
      plain multi-scale stage and one residual stage, pointnetv2.py:17-106 over ConvPool, local_aggregation.py:140-239)
      and `PointNetFPModule` (pointnetv2.py:108-150) over the oracle operators."""
    ref_group, ref_pointnext, ref_pointmlp = import_reference()
    import openpoints.models.backbone.curvenet as ref_curve
    import openpoints.models.backbone.pointnetv2 as ref_pn2
    import openpoints.models.layers.upsampling as ref_up
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    from easydict import EasyDict
    out = {}

    # ---- G15a: three nearest + interpolation --------------------------------------------------------------
    cases = [("cfg1", GI.config1_xyz(), GI.take_points(GI.config1_xyz(), O.furthest_point_sampling(GI.config1_xyz(), 512)))]
    for lv, (unk, kn) in enumerate(_fp_pyramid(seed=151)):
        cases.append((f"fp{lv}", unk, kn))
    for name, unk, kn in cases:
        d_ref = ref_curve.square_distance(_t(unk), _t(kn))                       # (B,n,m), reference function
        ds, order = d_ref.sort(dim=-1)                                           # curvenet.py:448-449
        # the matmul form |a|^2 + |b|^2 - 2ab carries an ABSOLUTE error of a few eps * (|a|^2 + |b|^2) ~ 5e-7 here, the
        # kernel's difference form a relative one: where two of a point's four smallest distances lie closer than
        # 4e-6 the two forms may order them differently.  Such points are excluded from the index comparison (the
        # mask is committed); everywhere else the indices must be identical, and the excluded share small.
        exact = ((_t(unk).double().unsqueeze(2) - _t(kn).double().unsqueeze(1)) ** 2).sum(-1)      # (B,n,m) float64
        four = exact.sort(dim=-1)[0][:, :, :4]
        safe = ((four[:, :, 1:] - four[:, :, :-1]).min(-1)[0] > 4e-6).numpy()
        assert safe.mean() > 0.97, f"G15 {name}: too many near-ties ({1 - safe.mean():.3f})"
        idx_ref = order[:, :, :3].numpy()
        o_d2, o_idx = O.three_nn(unk, kn)
        assert np.array_equal(o_idx[safe], idx_ref[safe]), f"G15 {name}: oracle three_nn indices != reference sort"
        assert np.allclose(o_d2[safe], ds[:, :, :3].numpy()[safe], rtol=0, atol=2e-6), "distances (matmul form vs differences)"
        for v in O.ALL_DIST_VARIANTS:                                             # every rounding of the oracle
            assert np.array_equal(O.three_nn(unk, kn, v)[1][safe], idx_ref[safe]), f"G15 {name}: rounding variant {v}"
        out[f"g15_{name}_safe"] = safe
        o_idx = idx_ref                                  # the interpolation checks below run on the REFERENCE's choice
        c = 32 if name != "fp3" else 96
        pts = _t(GI.seeded_normal((2, c, kn.shape[1]), seed=152)).requires_grad_(True)
        fp_mod = ref_curve.PointNetFeaturePropagation(in_channel=c, mlp=[])       # forward = the interpolation alone
        got = fp_mod(_t(unk).transpose(1, 2), _t(kn).transpose(1, 2), None, pts)  # (B,c,n)
        w_ref = (1.0 / (ds[:, :, :3] + 1e-8))
        w_ref = (w_ref / w_ref.sum(2, keepdim=True)).numpy()                      # curvenet.py:452-454
        mine = O.three_interpolate(pts.detach().numpy(), o_idx, w_ref)
        assert np.allclose(mine, got.detach().numpy(), rtol=1e-6, atol=1e-6), f"G15 {name}: three_interpolate"
        gout = _t(GI.seeded_normal(tuple(got.shape), seed=153))
        (got * gout).sum().backward()
        mine_g = O.three_interpolate_grad(gout.numpy(), o_idx, w_ref, kn.shape[1])
        assert np.allclose(mine_g, pts.grad.numpy(), rtol=1e-6, atol=1e-6), f"G15 {name}: three_interpolate_grad"
        out[f"g15_{name}_idx"] = idx_ref.astype(np.int32)
        if name in ("cfg1", "fp3"):
            out[f"g15_{name}_weight"] = w_ref.astype(np.float32)
            out[f"g15_{name}_interp"] = got.detach().numpy()
            out[f"g15_{name}_interp_grad"] = pts.grad.numpy()
    print("G15a: oracle three_nn / three_interpolate / its gradient == the reference's pure-torch interpolation on",
          [n for n, _, _ in cases])

    # ---- G15b: scatter-add gradients ------------------------------------------------------------------------
    xyz = GI.config1_xyz()
    fps512 = O.furthest_point_sampling(xyz, 512)
    bq = O.ball_query(0.15, 32, xyz, GI.take_points(xyz, fps512))
    feats = _t(GI.seeded_normal((2, 32, 1024), seed=11)).requires_grad_(True)
    grouped = ref_group.torch_grouping_operation(feats, _t(bq).long())
    gout = _t(GI.seeded_normal((2, 32, 512, 32), seed=12))
    (grouped * gout).sum().backward()
    mine = O.group_points_grad(gout.numpy(), bq, 1024)
    assert np.allclose(mine, feats.grad.numpy(), rtol=1e-6, atol=1e-5), "group_points_grad"
    out["g15_group_grad"] = feats.grad.numpy()
    feats2 = _t(GI.seeded_normal((2, 32, 1024), seed=11)).requires_grad_(True)
    gathered = torch.gather(feats2, 2, _t(fps512).long().unsqueeze(1).expand(-1, 32, -1))   # subsample.py:181-183
    assert np.array_equal(gathered.detach().numpy(), O.gather_points(feats2.detach().numpy(), fps512))
    gout2 = _t(GI.seeded_normal((2, 32, 512), seed=13))
    (gathered * gout2).sum().backward()
    assert np.allclose(O.gather_points_grad(gout2.numpy(), fps512, 1024), feats2.grad.numpy(), rtol=1e-6, atol=1e-6)
    out["g15_gather_grad"] = feats2.grad.numpy()
    print("G15b: oracle group_points_grad / gather_points_grad == autograd through the reference's pure-torch forms")

    # ---- G16: PointNet++ blocks (pointnetv2.py:17-150) over the oracle operators ----------------------------
    ref_pn2.furthest_point_sample = _OracleOps.furthest_point_sample
    ref_up.three_nn = _OracleOps.three_nn
    ref_up.three_interpolate = _OracleOps._Interp.apply
    ref_pn2.three_interpolation = ref_up.three_interpolation
    common = dict(aggr_args={'feature_type': 'dp_fj', 'reduction': 'max'}, conv_args={'order': 'conv-norm-act'},
                  norm_args={'norm': 'bn'}, act_args={'act': 'relu'})
    sa1 = fill_parameters_by_name(ref_pn2.PointNetSAModuleMSG(
        stride=4, radii=[0.1, 0.2], nsamples=[16, 32], channel_list=[[4, 16, 32], [4, 16, 32]],
        group_args=EasyDict(NAME='ballquery', normalize_dp=False), use_res=False, **common))
    sa2 = fill_parameters_by_name(ref_pn2.PointNetSAModuleMSG(
        stride=4, radii=[0.4], nsamples=[32], channel_list=[[64, 64, 96]],
        group_args=EasyDict(NAME='ballquery', normalize_dp=True), use_res=True, **common))
    fp = fill_parameters_by_name(ref_pn2.PointNetFPModule([96 + 64, 64, 48]))
    for mod in (sa1, sa2, fp):
        mod.train()
    p0 = _t(GI.unit_sphere_cloud(2, 1024, seed=161))
    f0 = _t(GI.seeded_normal((2, 4, 1024), seed=162)).requires_grad_(True)
    p1, f1 = sa1(p0, f0)                      # (2,256,3), (2,64,256)
    p2, f2 = sa2(p1, f1)                      # (2,64,3),  (2,96,64)
    up = fp(p1, p2, f1, f2)                   # (2,48,256)
    (up * _t(GI.seeded_normal(tuple(up.shape), seed=163))).sum().backward()
    out.update(g16_p1=p1.numpy(), g16_f1=f1.detach().numpy(), g16_p2=p2.numpy(), g16_f2=f2.detach().numpy(),
               g16_up=up.detach().numpy(), g16_grad_f0=f0.grad.numpy(),
               g16_grad_sa1_w=sa1.local_aggregations[1].SA_CONFIG_operator.convs[0][0].weight.grad.numpy(),
               g16_grad_sa2_skip=sa2.local_aggregations[0].SA_CONFIG_operator.skipconv[0].weight.grad.numpy(),
               g16_grad_fp_w=fp.convs[1][0].weight.grad.numpy())
    for tag, mod in (("sa1", sa1), ("sa2", sa2), ("fp", fp)):
        out[f"g16_{tag}_keys"] = np.array(sorted(mod.state_dict().keys()))
    print("G16: PointNet++ SA (multi-scale, residual) + FP goldens:", tuple(f1.shape), tuple(f2.shape), tuple(up.shape))

    path = os.path.join(HERE, "pins_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "pointnet2"):
        main()
    if what in ("all", "adaptpoint"):
        main_adaptpoint()
    if what in ("all", "pins"):
        main_pins()
    if what in ("all", "classifier"):
        main_classifier_b8()
