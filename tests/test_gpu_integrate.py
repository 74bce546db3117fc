"""GPU: `adaptpoint_amd.integrate` dispatching into the FUSED kernels.  The real reference cannot travel to the GPU box
(tests/test_integrate_reference_cpu.py runs against it in the build container, on CPU tensors); here stand-in classes
with the reference's attribute layout -- what `openpoints/models/backbone/pointnext.py:82-170` and
`models_adaptpoint/generator_component4_15.py:93-105, 368-480` construct: `convs`, `skipconv`, `act`, `grouper`
(`QueryAndGroup` attributes), `sample_fn`; `net`; `reduce / kneighbors / radi / affine_*`; `to_qkv / pos_embedding / res`
-- are patched exactly as the reference's classes are, and must then (1) run on the fused kernels (counted), (2) equal this
package's own fused modules holding the same weights, bit for bit where the kernels are deterministic."""
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _fake_reference_modules():
    """Two modules named like the reference's, holding classes with the reference's constructor-time attributes and a
    `forward` that raises: after the patch nothing may reach it for the covered configurations."""
    pn = types.ModuleType("openpoints.models.backbone.pointnext")
    gen = types.ModuleType("openpoints.models_adaptpoint.generator_component4_15")

    def furthest_point_sample(*a):
        raise AssertionError("reference forward reached")

    class QueryAndGroup(nn.Module):
        def __init__(self, radius, nsample, normalize_dp=True):
            super().__init__()
            self.radius, self.nsample, self.normalize_dp = radius, nsample, normalize_dp
            self.relative_xyz, self.normalize_by_std, self.normalize_by_allstd = True, False, False
            self.normalize_by_allstd2, self.return_only_idx = False, False

    class SetAbstraction(nn.Module):
        def __init__(self, cin, cout, radius=0.15, nsample=32, feature_type='dp_fj'):
            super().__init__()
            self.stride, self.is_head, self.all_aggr, self.use_res, self.feature_type = 2, False, False, True, feature_type
            mid = cout // 2
            self.skipconv = nn.Sequential(nn.Conv1d(cin, cout, 1))
            self.act = nn.ReLU(inplace=True)
            self.convs = nn.Sequential(nn.Sequential(nn.Conv2d(3 + cin, mid, 1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(inplace=True)),
                                       nn.Sequential(nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout)))
            self.grouper = QueryAndGroup(radius, nsample)
            self.sample_fn = furthest_point_sample

        def forward(self, pf):
            raise AssertionError("reference forward reached")

    class ConvBNReLU1D(nn.Module):
        def __init__(self, cin, cout, bias=False):
            super().__init__()
            self.act = nn.ReLU(inplace=True)
            self.net = nn.Sequential(nn.Conv1d(cin, cout, 1, bias=bias), nn.BatchNorm1d(cout), self.act)

        def forward(self, x):
            raise AssertionError("reference forward reached")

    class PointsetGrouper(nn.Module):
        def __init__(self, channel, reduce, kneighbors, radi):
            super().__init__()
            self.reduce, self.kneighbors, self.radi, self.normalize = reduce, kneighbors, radi, "anchor"
            self.affine_alpha = nn.Parameter(torch.ones([1, 1, 1, channel]))
            self.affine_beta = nn.Parameter(torch.zeros([1, 1, 1, channel]))

        def forward(self, xyz, points):
            raise AssertionError("reference forward reached")

    class Anchor_selfattention(nn.Module):
        def __init__(self, dim, head_num):
            super().__init__()
            self.dim, self.head_num, self.head_dim = dim, head_num, dim // head_num
            self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
            self.pos_embedding = nn.Sequential(nn.Conv1d(3, dim, 1), nn.BatchNorm1d(dim))
            self.res = nn.Sequential(nn.Conv1d(dim, dim, 1), nn.BatchNorm1d(dim))

        def forward(self, x, xyz=None):
            raise AssertionError("reference forward reached")
    pn.SetAbstraction = SetAbstraction
    gen.ConvBNReLU1D, gen.PointsetGrouper, gen.Anchor_selfattention = ConvBNReLU1D, PointsetGrouper, Anchor_selfattention
    return pn, gen


def test_patched_classes_run_on_the_fused_kernels_and_equal_the_package_modules(dev, monkeypatch):
    from adaptpoint_amd import integrate
    from adaptpoint_amd import set_abstraction as SA
    from adaptpoint_amd.attention import AnchorSelfAttention
    from adaptpoint_amd.imitator import ConvBNReLU1D
    from adaptpoint_amd.pointset import PointsetGrouper
    pn, gen = _fake_reference_modules()
    monkeypatch.setitem(sys.modules, integrate.TARGET_MODULES[0], pn)
    monkeypatch.setitem(sys.modules, integrate.TARGET_MODULES[1], gen)
    integrate.COUNTS.clear()
    assert set(integrate.patch_openpoints(lazy=True)) == set(integrate.TARGET_MODULES)
    try:
        torch.manual_seed(0)
        B = 4
        p = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=7)).to(dev)
        f = torch.from_numpy(GI.seeded_normal((B, 32, 1024), seed=8)).to(dev)
        # --- SetAbstraction: the headline block's shape, training mode
        ref = pn.SetAbstraction(32, 64).to(dev).train()
        mine = SA.SetAbstraction(32, 64, layers=2, stride=2, fused=True,
                                 group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                                 norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                                 use_res=True).to(dev).train()
        mine.load_state_dict(ref.state_dict())                       # same key names: the reference's
        before = sum(SA.FUSED_FALLBACKS.values())
        fa, fb = f.clone().requires_grad_(True), f.clone().requires_grad_(True)
        pa, oa = ref([p, fa])
        pb, ob = mine([p, fb])
        oa.square().sum().backward()
        ob.square().sum().backward()
        assert sum(SA.FUSED_FALLBACKS.values()) == before            # the fused launches, not the unfused mirror
        assert torch.equal(pa, pb) and float((oa - ob).abs().max()) <= 1e-5 * float(ob.abs().max())
        assert float((fa.grad - fb.grad).abs().max()) <= 1e-4 * float(fb.grad.abs().max())
        for (k, qa), (_, qb) in zip(ref.named_parameters(), mine.named_parameters()):
            assert float((qa.grad - qb.grad).abs().max()) <= 1e-4 * float(qb.grad.abs().max()) + 1e-6, k
        other = pn.SetAbstraction(32, 64, feature_type='dp_df').to(dev)
        with pytest.raises(AssertionError, match="reference forward reached"):
            other([p, f])                                            # uncovered configuration -> the reference's own forward
        # --- the imitator's three classes
        cr, cm = gen.ConvBNReLU1D(64, 128).to(dev).train(), ConvBNReLU1D(64, 128, bias=False).to(dev).train()
        cm.load_state_dict(cr.state_dict())
        x = torch.randn(B, 64, 1024, device=dev)
        assert torch.equal(cr(x), cm(x))
        gr, gm = gen.PointsetGrouper(64, 2, 24, 0.2).to(dev), PointsetGrouper(64, 2, 24, 0.2).to(dev)
        gm.load_state_dict(gr.state_dict())
        pts = torch.randn(B, 1024, 64, device=dev)
        (xa, ya), (xb, yb) = gr(p, pts), gm(p, pts)
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
        ar, am = gen.Anchor_selfattention(64, 4).to(dev).train(), AnchorSelfAttention(64, 4).to(dev).train()
        am.load_state_dict(ar.state_dict())
        xq = torch.randn(B, 1024, 64, device=dev)
        assert torch.equal(ar(xq, p), am(xq, p))
        c = dict(integrate.COUNTS)
        assert c.get("SetAbstraction.fused") == 1 and c.get("SetAbstraction.reference: feature_type 'dp_df'") == 1, c
        assert c.get("ConvBNReLU1D.fused") == 1 and c.get("PointsetGrouper.fused") == 1 and c.get("Anchor_selfattention.fused") == 1, c
        assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    finally:
        integrate.unpatch_openpoints()


def test_clouds_beyond_the_resident_samplers_go_to_the_reference_forward(dev, monkeypatch):
    """the adapter is chosen from the module's structure, the point count is a property of the
    CALL.  A patched, unmodified tree on S3DIS-sized clouds (N > 16384: beyond apn_sa_sample_seq) must reach the
    reference's own forward, counted by reason -- never raise APN_EINVAL out of the fused index stage; other dtypes
    likewise.  The package's own module takes the unfused index stage (streaming sampler) for such clouds."""
    from adaptpoint_amd import integrate
    from adaptpoint_amd import set_abstraction as SA
    pn, gen = _fake_reference_modules()
    monkeypatch.setitem(sys.modules, integrate.TARGET_MODULES[0], pn)
    monkeypatch.setitem(sys.modules, integrate.TARGET_MODULES[1], gen)
    integrate.COUNTS.clear()
    integrate.patch_openpoints(lazy=True)
    try:
        torch.manual_seed(0)
        N = 20000
        p = torch.from_numpy(GI.unit_sphere_cloud(1, N, seed=11)).to(dev)
        f = torch.from_numpy(GI.seeded_normal((1, 32, N), seed=12)).to(dev)
        for cin, cout in ((32, 64), (64, 128)):                  # the resident shape and a width-generic one
            ref = pn.SetAbstraction(cin, cout).to(dev).train()
            fx = f if cin == 32 else torch.cat([f, f], 1)
            with pytest.raises(AssertionError, match="reference forward reached"):
                ref([p, fx])
            with pytest.raises(AssertionError, match="reference forward reached"):
                ref([p[:, :1024].double(), fx[:, :, :1024].double()])
        c = dict(integrate.COUNTS)
        assert c.get(f"SetAbstraction.reference: N > {integrate.MAX_FUSED_POINTS}") == 2, c
        assert c.get("SetAbstraction.reference: dtype torch.float64 / torch.float64") == 2, c
        assert "SetAbstraction.fused" not in c, c
        # the package's own block: fused=True on a 20000-point cloud runs (unfused index stage + the fused grouped MLP)
        mine = SA.SetAbstraction(64, 128, layers=2, stride=4, fused=True,
                                 group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                                 norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                                 use_res=True).to(dev).train()
        plain = SA.SetAbstraction(64, 128, layers=2, stride=4, fused=False,
                                  group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                                  norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                                  use_res=True).to(dev).train()
        plain.load_state_dict(mine.state_dict())
        f64 = torch.cat([f, f], 1)
        (pa, oa), (pb, ob) = mine([p, f64]), plain([p, f64])
        assert pa.shape == (1, N // 4, 3) and torch.equal(pa, pb)
        assert float((oa - ob).abs().max()) <= 2e-3 * float(ob.abs().max())
    finally:
        integrate.unpatch_openpoints()
