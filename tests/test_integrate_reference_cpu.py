"""Build-container CPU test (needs /root/reference; skipped where the reference is absent, e.g. on the GPU box): the
zero-edit opt-in `adaptpoint_amd.integrate.patch_openpoints()` against the REAL reference classes, imported in memory
with the stand-ins of tests/golden/make_golden.py for the packages this image lacks.  After the patch the reference's
SetAbstraction / PointsetGrouper / Anchor_selfattention / ConvBNReLU1D keep their parameters and state_dict keys, their
forwards dispatch into adaptpoint_amd (on CPU tensors the package's host-side mirrors over the oracle stand in for the
fused kernels), and the results equal what the unpatched reference computes over the same oracle operators."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "openpoints")), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ref_group, ref_pointnext, ref_pointmlp = mg.import_reference()        # oracle-backed FPS / ball query / grouping
    import openpoints.models_adaptpoint.generator_component4_15 as ref_gen
    ref_gen.furthest_point_sample = mg._OracleOps.furthest_point_sample
    ref_gen.ball_query = mg._OracleOps.ball_query
    return ref_pointnext, ref_gen


def _clone_inputs(*ts):
    return [t.detach().clone().requires_grad_(t.requires_grad) for t in ts]


def test_patched_reference_classes_dispatch_to_this_package_and_agree(ref, cpu_mirrors):
    import golden_inputs as GI
    from easydict import EasyDict
    from adaptpoint_amd import integrate
    ref_pointnext, ref_gen = ref
    torch.manual_seed(0)
    sa = ref_pointnext.SetAbstraction(
        32, 64, layers=2, stride=2,
        group_args=EasyDict(NAME='ballquery', radius=0.15, nsample=32, normalize_dp=True),
        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
        sampler='fps', feature_type='dp_fj', use_res=True).train()
    head = ref_pointnext.SetAbstraction(4, 32, layers=1, stride=1, group_args=EasyDict(NAME='ballquery', radius=0.15, nsample=32),
                                        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                                        is_head=True).train()
    odd = ref_pointnext.SetAbstraction(32, 64, layers=2, stride=2,
                                       group_args=EasyDict(NAME='ballquery', radius=0.15, nsample=16, normalize_dp=True),
                                       norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                                       sampler='fps', feature_type='dp_df', use_res=True).train()
    grp = ref_gen.PointsetGrouper(channel=16, reduce=2, kneighbors=24, radi=0.2)
    with torch.no_grad():
        grp.affine_alpha.uniform_(0.5, 1.5)
        grp.affine_beta.uniform_(-0.2, 0.2)
    att = ref_gen.Anchor_selfattention(32, 2).train()
    cbr = ref_gen.ConvBNReLU1D(16, 32, bias=False).train()
    mods = {"sa": sa, "head": head, "odd": odd, "grp": grp, "att": att, "cbr": cbr}
    keys = {k: list(m.state_dict().keys()) for k, m in mods.items()}
    params = {k: [id(q) for q in m.parameters()] for k, m in mods.items()}

    p = torch.from_numpy(GI.unit_sphere_cloud(2, 256, seed=3))
    f = torch.from_numpy(GI.seeded_normal((2, 32, 256), seed=4)).requires_grad_(True)
    x4 = torch.from_numpy(GI.seeded_normal((2, 4, 256), seed=5))
    pts = torch.from_numpy(GI.seeded_normal((2, 256, 16), seed=6)).requires_grad_(True)
    xa = torch.from_numpy(GI.seeded_normal((2, 64, 32), seed=7))
    xc = torch.from_numpy(GI.seeded_normal((2, 16, 256), seed=8))

    def run_all():
        out = {}
        (f1,) = _clone_inputs(f)
        new_p, o = sa([p, f1])
        (o * o).sum().backward()
        out["sa"] = [new_p, o.detach(), f1.grad] + [q.grad.clone() for q in sa.parameters()]
        sa.zero_grad()
        out["head"] = [head([p, x4])[1].detach()]
        out["odd"] = [odd([p, f.detach()])[1].detach()]
        (pt1,) = _clone_inputs(pts)
        nx, g = grp(p, pt1)
        g.square().sum().backward()
        out["grp"] = [nx, g.detach(), pt1.grad, grp.affine_alpha.grad.clone()]
        grp.zero_grad()
        out["att"] = [att(xa, p[:, :64]).detach()]
        out["cbr"] = [cbr(xc).detach()]
        return out

    def bn_state(m):
        return {k: v.clone() for k, v in m.state_dict().items()}
    start = {k: bn_state(m) for k, m in mods.items()}
    want = run_all()                                   # the reference's own forwards (over the oracle operators)
    after_ref = {k: bn_state(m) for k, m in mods.items()}
    for k, m in mods.items():
        m.load_state_dict(start[k])                    # same BatchNorm running statistics for the second pass

    integrate.COUNTS.clear()
    patched = integrate.patch_openpoints()
    try:
        assert set(patched) == set(integrate.TARGET_MODULES)
        assert ref_pointnext.SetAbstraction.forward is integrate._sa_forward
        got = run_all()
        for k, m in mods.items():
            assert list(m.state_dict().keys()) == keys[k], k          # nothing added, renamed or re-registered
            assert [id(q) for q in m.parameters()] == params[k], k
        counts = dict(integrate.COUNTS)
    finally:
        integrate.unpatch_openpoints()
    assert ref_pointnext.SetAbstraction.forward is not integrate._sa_forward and "fused" not in ref_gen.PointsetGrouper.__dict__
    # dispatch: the covered configurations went into adaptpoint_amd, the uncovered one to the reference's forward, counted
    assert counts.get("SetAbstraction.fused") == 2, counts                                  # the block and the stem
    assert counts.get("SetAbstraction.reference: feature_type 'dp_df'") == 1, counts
    assert counts.get("PointsetGrouper.composed") == 1 and counts.get("Anchor_selfattention.composed") == 1, counts
    assert counts.get("ConvBNReLU1D.reference: CPU tensor") == 1, counts
    for k in want:
        for i, (a, b) in enumerate(zip(got[k], want[k])):
            scale = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 2e-5 * scale, (k, i, float((a - b).abs().max()), scale)
    for k, m in mods.items():                                                               # running statistics advanced alike
        for name, v in m.state_dict().items():
            np.testing.assert_allclose(v.double().numpy(), after_ref[k][name].double().numpy(), rtol=1e-5, atol=1e-6, err_msg=f"{k}.{name}")


def test_environment_switch_patches_lazily(monkeypatch):
    """APN_PATCH_OPENPOINTS=1: the drop-in module installs a post-import hook; unset, nothing is touched."""
    from adaptpoint_amd import integrate
    monkeypatch.delenv("APN_PATCH_OPENPOINTS", raising=False)
    assert not integrate.requested_by_environment()
    monkeypatch.setenv("APN_PATCH_OPENPOINTS", "1")
    assert integrate.requested_by_environment()
    try:
        integrate.patch_openpoints(lazy=True)
        assert any(isinstance(f, integrate._PostImport) for f in sys.meta_path)
    finally:
        integrate.unpatch_openpoints()
    assert not any(isinstance(f, integrate._PostImport) for f in sys.meta_path)
