"""CPU suite: the PointNeXt-S classifier mirror (adaptpoint_amd/pointnext.py) against the
REFERENCE model (openpoints BaseCls = PointNextEncoder + ClsHead) captured in the goldens by
tests/golden/make_golden.py; both hold the same name-seeded weights."""
import numpy as np
import torch

import golden_inputs as GI


def _inputs():
    pos = torch.from_numpy(GI.unit_sphere_cloud(2, 1024, seed=31))
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    return pos, x


def _model():
    from oracle import cpu_block as CB
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    with CB.CpuOps():
        m = PointNextSClassifier()
    return fill_parameters_by_name(m)


def test_structure_matches_reference(golden):
    m = _model()
    assert sum(q.numel() for q in m.parameters()) == 1367119          # pointnext-s.yaml:1-3
    assert sorted(m.state_dict().keys()) == list(golden["g5_state_keys"])
    np.testing.assert_allclose(m.encoder.radii, [0.15, 0.15, 0.225, 0.3375, 0.50625, 0.759375], rtol=1e-12)


def test_logits_and_grads_match_reference(golden):
    from oracle import cpu_block as CB
    m = _model()
    pos, x = _inputs()
    with CB.CpuOps():
        m.eval()
        with torch.no_grad():
            logits = m({'pos': pos, 'x': x})
        np.testing.assert_allclose(logits.numpy(), golden["g5_logits_eval"], rtol=1e-4, atol=1e-5)
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        xt = x.clone().requires_grad_(True)
        lt = m({'pos': pos, 'x': xt})
        lt.square().sum().backward()
    np.testing.assert_allclose(lt.detach().numpy(), golden["g5_logits_train"], rtol=1e-4, atol=1e-5)
    chk = np.array([xt.grad.double().sum().item(), xt.grad.double().abs().sum().item()])
    np.testing.assert_allclose(chk, golden["g5_grad_x_checksum"], rtol=1e-4)
    np.testing.assert_allclose(m.encoder.encoder[0][0].convs[0][0].weight.grad.numpy(),
                               golden["g5_grad_stem_w"], rtol=1e-3, atol=1e-5)


def test_smooth_cross_entropy():
    from adaptpoint_amd.pointnext import SmoothCrossEntropy
    torch.manual_seed(0)
    pred = torch.randn(6, 15)
    gt = torch.randint(0, 15, (6,))
    want = torch.nn.functional.cross_entropy(pred, gt, label_smoothing=0.0)
    assert torch.allclose(SmoothCrossEntropy(0.0 + 1e-12)(pred, gt), want, atol=1e-5)
    # smoothing spreads eps/(n-1) over the wrong classes (build.py:55)
    eps, n = 0.3, 15
    oh = torch.nn.functional.one_hot(gt, n).float()
    tgt = oh * (1 - eps) + (1 - oh) * eps / (n - 1)
    assert torch.allclose(SmoothCrossEntropy(0.3)(pred, gt), -(tgt * torch.log_softmax(pred, 1)).sum(1).mean())


def test_training_mode_at_b8_matches_the_reference_parameter_by_parameter(golden_b8):
    """G17: forward + SmoothCE + backward at B = 8 in training mode (dropout off) against the reference classifier run
    over the oracle operators: logits, loss, input gradient and EVERY parameter's gradient in relative L2."""
    from oracle import cpu_block as CB
    import classifier_b8_checks as K
    with CB.CpuOps():
        r = K.run_g17(_model(), torch.device("cpu"), golden_b8)
    print("G17 on CPU:", {k: "%.2e" % r[k] for k in ("logits", "loss", "grad_x", "bn")}, "worst gradients:", K.worst(r["grads"]))
    # measured: 0.0 everywhere (the mirror issues the same torch-CPU calls in the same order over the same oracle operators)
    assert r["logits"] < 1e-6 and r["loss"] < 1e-6 and r["grad_x"] < 1e-6 and r["bn"] < 1e-6
    assert max(r["grads"].values()) < 1e-6, K.worst(r["grads"])
    assert max(r["norms"].values()) < 1e-6, K.worst(r["norms"])


def test_training_iteration_at_b8_matches_the_reference_trainer(golden_b8):
    """G18: one `train_one_epoch` iteration at B = 8 (resampler, forward, loss, backward, clip, AdamW): pre-clip gradients
    per parameter, and every sampled weight with a gradient above the noise takes exactly the reference's step."""
    from oracle import cpu_block as CB
    import classifier_b8_checks as K
    with CB.CpuOps():
        r = K.run_g18(_model(), torch.device("cpu"), golden_b8, grad_bar=2e-4)
    print("G18 on CPU:", {k: ("%.2e" % r[k] if isinstance(r[k], float) else r[k]) for k in ("logits", "loss", "bn", "steps_checked", "step_mismatch")},
          "worst gradients:", K.worst(r["grads"]))
    assert r["logits"] < 1e-6 and r["loss"] < 1e-6 and r["bn"] < 1e-6            # measured: 0.0
    assert max(r["grads"].values()) < 1e-6, K.worst(r["grads"])
    assert r["steps_checked"] > 100000 and r["step_mismatch"] == 0
