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
# Measured (rounds 4 and 5, the same): unfused logits 1.2e-5 / loss 8e-7 / input gradient 6.5e-4 / parameter gradients median 6e-4, worst
# 1.6e-3 (MIOpen's convolution gradients); every stage fused: 2.3e-4 / 5.4e-6 / 1.3e-2 / median 9.4e-3, worst 1.8e-2 -- the
# split-operand contraction moves ~1e-5 of the ReLU gates and pool winners of four stacked blocks, and at B = 8 every
# BatchNorm spreads each switched gate over its whole channel.
# (unfused gradient bars leave room for MIOpen's per-box solver choice: 1.1e-2 was seen on the sibling test G18)
# Round 5: the fused bars are 2 x measured -- what the residual IS is held by
# test_fused_gradient_residual_is_gate_noise_not_arithmetic below.
G17_BARS = {False: dict(logits=1e-4, loss=1e-5, grad_x=1e-2, grads=3e-2), True: dict(logits=5e-4, loss=2e-5, grad_x=2.6e-2, grads=3.6e-2)}


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


def _g17_gradients(model, dev, golden, hooks=None):
    """{name: gradient} (+ 'x': the input gradient, 'logits') of the G17 training pass on the GPU."""
    import classifier_b8_checks as K
    model = K.no_dropout(model).to(dev).train()
    handles = hooks(model) if hooks else []
    pos = torch.from_numpy(GI.unit_sphere_cloud(8, 1024, seed=171)).to(dev)
    x = torch.cat([pos, K.height_channel(pos)], -1).transpose(1, 2).contiguous().requires_grad_(True)
    target = torch.from_numpy(golden["g17_target"]).to(dev)
    logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, target)
    loss.backward()
    for h in handles:
        h.remove()
    out = {n: q.grad.detach().double() for n, q in model.named_parameters() if q.grad is not None}
    out["x"] = x.grad.detach().double()
    out["logits"] = logits.detach().double()
    return out


def test_fused_gradient_residual_is_gate_noise_not_arithmetic(dev, golden_b8):
    """the fused classifier's gradients sit 1e-2 from the reference (relative L2) where the
    unfused path sits at 6e-4, and a bar of 5e-2 would let a real kernel regression of a few 1e-3 through.  Two
    measurements on the G17 inputs that make the bar mean something -- the classifier-level counterpart of
    tests/test_gpu_fused.py::test_discontinuities_explain_the_gradient_residual:
      (a) NOISE FLOOR: the UNFUSED fp32 classifier against ITSELF with every convolution / linear output perturbed by
          N(0, 3e-6 rms) -- the size of the split-operand contraction's forward error (its logits then move by about what the
          fused path's do).  The perturbation switches a few 1e-6 of the ReLU gates and pool winners of four stacked blocks;
          at B = 8 every BatchNorm spreads each switched gate over its channel.  Measured (round 5): parameter gradients
          2-6e-2 in relative L2 for such a perturbation (median 5e-2, logits 2.7e-4), the fused path 0.02-1.9e-2 (median
          1.0e-2, logits 2.3e-4): at most 0.73 of the floor.  The fused path's
          residual against the unfused one must be of THAT size (<= 3 x the floor, per parameter): it is the chain's own
          discontinuities, not arithmetic;
      (b) SYSTEMATIC ERROR: switched gates are noise of random sign, a wrong or missing term is not.  The PROJECTION of
          the fused gradient's error on the reference gradient, <g_fused - g_ref, g_ref> / |g_ref|^2 -- the relative error of
          the gradient's SCALE along the true direction -- averages the gate noise down with the tensor's size (not by its
          square root: BatchNorm correlates a channel's entries -- measured: 1e-2 at 1,000 entries under the 1e-5 noise,
          2-3e-3 above 8,000): it must stay below 1e-3 for every parameter of 8,192 entries or more (measured: <= 3.7e-4
          over thirteen tensors, against 3.4e-3 for the noise floor's own; the noise floor's own projection is printed beside it).  A backward kernel that drops a
          term or mis-scales one by a few 1e-3 fails here although it would pass the relative-L2 bars."""
    import classifier_b8_checks as K
    from adaptpoint_amd import set_abstraction as SA
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name

    def perturb(model):
        gen = torch.Generator(device=dev).manual_seed(4242)

        def hook(mod, inp, out):
            return out + torch.randn(out.shape, device=out.device, generator=gen) * (3e-6 * out.detach().pow(2).mean().sqrt())
        return [m.register_forward_hook(hook) for m in model.modules()
                if isinstance(m, (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Linear))]

    mk = lambda fused: fill_parameters_by_name(PointNextSClassifier(fused=fused))
    before = sum(SA.FUSED_FALLBACKS.values())
    ref = _g17_gradients(mk(False), dev, golden_b8)
    noisy = _g17_gradients(mk(False), dev, golden_b8, hooks=perturb)
    fused = _g17_gradients(mk(True), dev, golden_b8)
    assert sum(SA.FUSED_FALLBACKS.values()) == before
    floor_l2 = 1e-3 * float(np.median([float(golden_b8[k]) for k in golden_b8.files if k.startswith("g17_gnorm/")]))
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-300))
    proj = lambda a, b: float(((a - b) * b).sum() / (b * b).sum().clamp_min(1e-300))
    rows = []
    for name, g in ref.items():
        if name == "logits" or float(g.norm()) < floor_l2:           # (analytic zeros: classifier_b8_checks.norm_floor)
            continue
        rows.append((name, g.numel(), rel(fused[name], g), rel(noisy[name], g), proj(fused[name], g), proj(noisy[name], g)))
    worst_ratio = max(r[2] / max(r[3], 1e-12) for r in rows)
    big = [r for r in rows if r[1] >= 8192]
    print("fused vs unfused on the G17 inputs: relative L2 median %.2e (noise floor %.2e), worst ratio to the floor %.2f; "
          "projection on the reference gradient: worst |.| %.2e over %d tensors of >= 8192 entries (noise floor's: %.2e); "
          "logits %.2e (floor %.2e)"
          % (float(np.median([r[2] for r in rows])), float(np.median([r[3] for r in rows])), worst_ratio,
             max(abs(r[4]) for r in big), len(big), max(abs(r[5]) for r in big),
             rel(fused["logits"], ref["logits"]), rel(noisy["logits"], ref["logits"])))
    for name, n, e_f, e_n, p_f, p_n in rows:
        assert e_f <= 3.0 * e_n + 1e-3, (name, e_f, e_n)             # (a) of the noise floor's size
    for name, n, e_f, e_n, p_f, p_n in big:
        assert abs(p_f) <= 1e-3, (name, n, p_f, p_n)                 # (b) no systematic error along the true gradient
