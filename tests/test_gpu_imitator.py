"""The imitator's predictor network (SAComponent, generator_component4_15.py:588-712) over the fused
operators, against the reference module's own outputs (tests/golden/make_golden.py, G8: the
reference run on CPU over the oracle ops with name-seeded weights)."""
import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _run(dev, oracle, fused):
    from adaptpoint_amd.imitator import SAComponent
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    m = fill_parameters_by_name(SAComponent(fused=fused)).to(dev)
    assert sum(q.numel() for q in m.parameters()) == 5998062
    m.train()
    xyz = GI.unit_sphere_cloud(2, 512, seed=81)
    x = torch.from_numpy(xyz).to(dev)
    anchor = torch.from_numpy(oracle.furthest_point_sampling(xyz, 4)).long().to(dev)
    prob, logits = m(x, anchor, return_logits=True)
    w = torch.from_numpy(GI.seeded_normal((2, 2, 512), seed=82)).to(dev).permute(0, 2, 1)
    (prob.sum() + (logits * w).sum()).backward()
    res = {"g8_sac_prob": prob, "g8_sac_mask_logits": logits,
           "g8_sac_grad_embed_w": m.embedding.net[0].weight.grad,
           "g8_sac_grad_alpha0": m.pointset_grouper_list[0].affine_alpha.grad}
    with torch.no_grad():
        _, mask = m(x, anchor)
    assert mask.shape == (2, 512, 2) and torch.all(mask.sum(-1) == 1)      # hard Gumbel soft-max: one-hot
    return {k: v.detach().cpu().numpy() for k, v in res.items()}


def _err(a, ref):
    return float(np.abs(a - ref).max() / max(1.0, np.abs(ref).max()))


def test_sa_component_matches_reference_golden(dev, golden, oracle):
    """Two comparisons.  (1) Against the reference module (run on CPU in the build container): ~40
    fp32 layers with training-mode BatchNorm over as few as 2 x 32 points; PyTorch-CPU and MIOpen
    (whose algorithm choice is tuned per machine) differ by ~1e-4 in the outputs and a few 1e-3 of
    their scale in gradients through the whole stack -- bars 1e-2 / 5e-2.  (2) The fused operators
    against the composed forms of the same mirror ON THE SAME GPU (same MIOpen kernels either side):
    the grouping stage is bit-exact, the attention core 1e-5 -- bars 1e-4 / 1e-3 / 2e-2."""
    fused = _run(dev, oracle, True)
    composed = _run(dev, oracle, False)
    for key, tol in (("g8_sac_prob", 1e-2), ("g8_sac_mask_logits", 1e-2),
                     ("g8_sac_grad_embed_w", 5e-2), ("g8_sac_grad_alpha0", 5e-2)):
        assert _err(fused[key], golden[key]) <= tol, (key, "fused vs reference", _err(fused[key], golden[key]))
        assert _err(composed[key], golden[key]) <= tol, (key, "composed vs reference")
    # (gradient bars 2e-2: the grouper's gradient sums with LDS float atomics whose order varies from run to run, and ~40
    # training-mode BatchNorm layers over few points amplify that -- typically 2e-3, once 1.27e-2 in a full-suite run)
    for key, tol in (("g8_sac_prob", 1e-4), ("g8_sac_mask_logits", 1e-3),
                     ("g8_sac_grad_embed_w", 2e-2), ("g8_sac_grad_alpha0", 2e-2)):
        assert _err(fused[key], composed[key]) <= tol, (key, "fused vs composed", _err(fused[key], composed[key]))
