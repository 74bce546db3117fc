"""The imitator's predictor network (SAComponent, generator_component4_15.py:588-712) over the fused
operators, against the reference module's own outputs (tests/golden/make_golden.py, G8: the
reference run on CPU over the oracle ops with name-seeded weights)."""
import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [True, False])
def test_sa_component_matches_reference_golden(dev, golden, oracle, fused):
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
    # fp32 network of ~40 layers with training-mode BatchNorm over as few as 2 x 32 points:
    # PyTorch-CPU vs MIOpen alone differ by ~1e-4 in the outputs; gradients through the whole stack
    # (max-pool arg-max ties, BatchNorm of tiny batches) amplify that to a few 1e-3 of their scale
    for got, key, tol in ((prob, "g8_sac_prob", 2e-3), (logits, "g8_sac_mask_logits", 2e-3),
                          (m.embedding.net[0].weight.grad, "g8_sac_grad_embed_w", 1e-2),
                          (m.pointset_grouper_list[0].affine_alpha.grad, "g8_sac_grad_alpha0", 1e-2)):
        ref = golden[key]
        err = np.abs(got.detach().cpu().numpy() - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), (key, err)
    prob2, mask = m(x, anchor)
    assert mask.shape == (2, 512, 2) and torch.all(mask.sum(-1) == 1)      # hard Gumbel soft-max: one-hot
