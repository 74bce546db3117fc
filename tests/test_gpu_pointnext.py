"""GPU: the full PointNeXt-S classifier (BASELINE configs[2]) through the extension, unfused
(fp32: matches the reference goldens to 1e-4) and with the fused bf16 stage 1."""
import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _inputs(dev, b=2, seed=31):
    pos = torch.from_numpy(GI.unit_sphere_cloud(b, 1024, seed=seed)).to(dev)
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    return pos, x


def test_classifier_unfused_matches_reference_goldens(dev, golden):
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    m = fill_parameters_by_name(PointNextSClassifier()).to(dev)
    pos, x = _inputs(dev)
    m.eval()
    with torch.no_grad():
        logits = m({'pos': pos, 'x': x})
    # fp32 throughout, but MIOpen runs these 1x1 convolutions with Winograd-class kernels
    # (miopenSp3AsmConv_*_f3x2): 1e-3-level differences from the CPU reference over 12 layers
    np.testing.assert_allclose(logits.cpu().numpy(), golden["g5_logits_eval"], rtol=5e-3, atol=3e-3)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    xt = x.clone().requires_grad_(True)
    lt = m({'pos': pos, 'x': xt})
    lt.square().sum().backward()
    np.testing.assert_allclose(lt.detach().cpu().numpy(), golden["g5_logits_train"], rtol=1e-2, atol=5e-3)
    chk = np.array([xt.grad.double().sum().item(), xt.grad.double().abs().sum().item()])
    np.testing.assert_allclose(chk[1], golden["g5_grad_x_checksum"][1], rtol=2e-2)
    # the signed sum too (round 3 computed it and never asserted it): measured against the sum of magnitudes, since the
    # signed sum of this gradient is two orders smaller than that
    assert abs(chk[0] - golden["g5_grad_x_checksum"][0]) <= 2e-2 * golden["g5_grad_x_checksum"][1]


# Bars of the two tests below, by path: (logits, loss, input gradient, worst parameter gradient).  Measured values are
# printed by the tests and recorded in DESIGN.md section 3.
# Measured (round 4): unfused logits 1.2e-5 / loss 8e-7 / input gradient 6.5e-4 / parameter gradients median 6e-4, worst
# 1.6e-3 (MIOpen's convolution gradients); every stage fused: 2.3e-4 / 5.4e-6 / 1.3e-2 / median 9.4e-3, worst 1.8e-2 -- the
# split-operand contraction moves ~1e-5 of the ReLU gates and pool winners of four stacked blocks, and at B = 8 every
# BatchNorm spreads each switched gate over its whole channel.
# (unfused gradient bars leave room for MIOpen's per-box solver choice: 1.1e-2 was seen on the sibling test G18)
G17_BARS = {False: dict(logits=1e-4, loss=1e-5, grad_x=1e-2, grads=3e-2), True: dict(logits=1e-3, loss=5e-5, grad_x=4e-2, grads=5e-2)}


@pytest.mark.parametrize("fused", [False, True])
def test_training_mode_at_b8_against_the_reference_parameter_by_parameter(dev, golden_b8, fused):
    """G17 on the GPU, unfused (nine operators + PyTorch fp32) and with every stage on the fused kernels: logits, loss,
    input gradient and EVERY parameter's gradient in relative L2 against the REFERENCE classifier's golden at B = 8 in
    training mode (not against the build's own mirror)."""
    import classifier_b8_checks as K
    from adaptpoint_amd import set_abstraction as SA
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    before = sum(SA.FUSED_FALLBACKS.values())
    r = K.run_g17(fill_parameters_by_name(PointNextSClassifier(fused=fused)), dev, golden_b8)
    print("G17 on the GPU (fused=%s):" % fused, {k: "%.2e" % r[k] for k in ("logits", "loss", "grad_x", "bn")},
          "worst parameter gradients:", [("%.2e" % v, n) for v, n in K.worst(r["grads"], 4)],
          "median %.2e" % float(np.median(list(r["grads"].values()))))
    bar = G17_BARS[fused]
    assert sum(SA.FUSED_FALLBACKS.values()) == before
    assert r["logits"] < bar["logits"] and r["loss"] < bar["loss"] and r["grad_x"] < bar["grad_x"] and r["bn"] < 1e-4
    assert max(r["grads"].values()) < bar["grads"], K.worst(r["grads"])
    assert max(r["norms"].values()) < bar["grads"], K.worst(r["norms"])
