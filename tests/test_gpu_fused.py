"""GPU tests of the fused grouped-MLP kernels (csrc/sa_fused.hip) against plain PyTorch.

Tolerances (floating point, stated here as section 3 of the task asks):
  * against the reference that rounds where the bf16 MFMA path rounds
    (tests/fused_reference.py, float64 accumulation): max |err| <= 2e-2, mean |err| <= 5e-4
    on BN-normalised outputs of O(1) -- what is left is f32-vs-f64 accumulation order and the
    occasional one-ulp flip of a bf16 rounding;
  * against the plain fp32 chain: max |err| <= 1.5e-1, mean |err| <= 1e-2 (bf16 inputs have
    8 significant bits; BASELINE.json's north_star asks for the bf16 MFMA contraction).
"""
import numpy as np
import pytest
import torch

import golden_inputs as GI
from fused_reference import chain

pytestmark = pytest.mark.gpu


def _setup(dev, B=4, seed=0, neg_gamma=False):
    from adaptpoint_amd.layers import ball_query, furthest_point_sample
    p = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=seed)).to(dev)
    f = torch.from_numpy(GI.seeded_normal((B, 32, 1024), seed=seed + 1)).to(dev)
    fidx = furthest_point_sample(p, 512).long()
    new_p = torch.gather(p, 1, fidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = ball_query(0.15, 32, p, new_p)
    torch.manual_seed(seed)
    conv1 = torch.nn.Conv2d(35, 32, 1, bias=False).to(dev)
    conv2 = torch.nn.Conv2d(32, 64, 1, bias=False).to(dev)
    bn1 = torch.nn.BatchNorm2d(32).to(dev)
    bn2 = torch.nn.BatchNorm2d(64).to(dev)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        if neg_gamma:
            bn2.weight[::3] *= -1
            bn1.weight[::5] *= -1
    return p, new_p, f, idx, conv1, bn1, conv2, bn2


@pytest.mark.parametrize("neg_gamma", [False, True])
def test_fused_forward_matches_pytorch(dev, neg_gamma):
    from adaptpoint_amd.fused import FusedForward, supported
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = _setup(dev, neg_gamma=neg_gamma)
    assert supported(p, f, idx, conv1, conv2)
    fw = FusedForward(p, new_p, f, idx, 0.15, conv1, bn1, conv2, bn2)
    torch.cuda.synchronize()
    args = (p, new_p, f, idx, 0.15, conv1.weight.view(32, 35), bn1.weight, bn1.bias,
            conv2.weight.view(64, 32), bn2.weight, bn2.bias)
    ref_bf, _ = chain(*args, emulate_bf16=True)
    ref_32, _ = chain(*args, emulate_bf16=False)
    e_bf = (fw.out.double() - ref_bf).abs()
    e_32 = (fw.out.double() - ref_32).abs()
    print("fused fwd err vs bf16-emulation max %.3e mean %.3e | vs fp32 max %.3e mean %.3e"
          % (e_bf.max(), e_bf.mean(), e_32.max(), e_32.mean()))
    assert e_bf.max() <= 2e-2 and e_bf.mean() <= 5e-4
    assert e_32.max() <= 1.5e-1 and e_32.mean() <= 1e-2


def test_fused_forward_updates_running_stats(dev):
    from adaptpoint_amd.fused import FusedForward
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = _setup(dev, B=2, seed=3)
    FusedForward(p, new_p, f, idx, 0.15, conv1, bn1, conv2, bn2)
    _, mid = chain(p, new_p, f, idx, 0.15, conv1.weight.view(32, 35), bn1.weight, bn1.bias,
                   conv2.weight.view(64, 32), bn2.weight, bn2.bias, emulate_bf16=True)
    n = 2 * 512 * 32
    np.testing.assert_allclose(bn1.running_mean.cpu().numpy(), 0.1 * mid["m1"].flatten().cpu().numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(bn1.running_var.cpu().numpy(),
                               (0.9 + 0.1 * mid["v1"].flatten() * n / (n - 1)).cpu().numpy(), rtol=2e-3)
    np.testing.assert_allclose(bn2.running_mean.cpu().numpy(), 0.1 * mid["m2"].flatten().cpu().numpy(), rtol=5e-3, atol=5e-4)
    assert int(bn1.num_batches_tracked) == 1 and int(bn2.num_batches_tracked) == 1
