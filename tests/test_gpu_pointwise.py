"""The per-point MLP layer (csrc/pointwise.hip, adaptpoint_amd.pointwise) against a float64 evaluation of the same
three modules -- Conv1d(kernel 1, no bias) + BatchNorm1d + ReLU, `ConvBNReLU1D` of
openpoints/models_adaptpoint/generator_component4_15.py:92-104 -- forward, every gradient, the running statistics."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

# (B, C_in, C_out, N): the imitator's layers at B = 2 (embedding, extract_feat 1 and 4, decoders 1 and 4), then
# sizes that are multiples of nothing (tile edges in all three contractions, the scalar loaders)
SHAPES = [(2, 3, 64, 1024), (2, 64, 128, 1024), (2, 512, 1024, 128), (2, 1536, 512, 256), (2, 192, 64, 1024),
          (3, 130, 70, 77), (1, 5, 33, 257), (5, 36, 200, 130),
          (40, 16, 24, 1024)]      # B * N above 32768: BatchNorm's gradient as two launches (below: one, csrc/pointwise.hip)

# planes per operand -> (bar on out, bar on gradients), relative L2 against float64: two bf16 planes keep 16 bits
# of each operand (~4e-6 measured), three keep all 24 (5e-8 .. 6e-7 measured, growing with the contraction length up to
# 1536: f32 accumulation -- fp32-class, the default)
TOL = {2: (3e-5, 2e-4), 3: (2e-6, 1e-5)}


def _rel(a, ref):
    return float((a.double() - ref).norm() / ref.norm().clamp_min(1e-30))


def _layer(C, O, dev, seed):
    g = torch.Generator().manual_seed(seed)
    conv = nn.Conv1d(C, O, 1, bias=False)
    bn = nn.BatchNorm1d(O)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(O, C, 1, generator=g) / C ** 0.5)
        bn.weight.copy_(0.5 + torch.rand(O, generator=g))
        bn.bias.copy_(0.3 * torch.randn(O, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(O, generator=g))
        bn.running_var.copy_(0.5 + torch.rand(O, generator=g))
    return conv.to(dev), bn.to(dev)


def _reference(conv, bn, x, gout, relu):
    """float64 modules.  With the ReLU, `gout` is zeroed IN PLACE where the pre-activation lies within 1e-4 of
    zero: there a 1e-5 difference in y decides the mask, and one flipped position (3 of 130,000 in the last
    shape) would move every gradient by ~1e-3 -- the comparison is of everything else."""
    conv64, bn64 = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    x64 = x.detach().double().requires_grad_(True)
    out = bn64(conv64(x64))
    if relu:
        gout.mul_((out.detach().abs() > 1e-4).to(gout.dtype))
        out = torch.relu(out)
    out.backward(gout.double())
    return out.detach(), x64.grad, conv64.weight.grad, bn64.weight.grad, bn64.bias.grad, bn64


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("relu", [True, False], ids=["relu", "linear"])
@pytest.mark.parametrize("planes", [3, 2])
def test_layer_matches_float64_modules(dev, shape, relu, planes, monkeypatch):
    from adaptpoint_amd import pointwise
    monkeypatch.setattr(pointwise, "PRECISION", planes)
    TOL_OUT, TOL_GRAD = TOL[planes]
    B, C, O, N = shape
    conv, bn = _layer(C, O, dev, seed=B + C + O + N)
    g = torch.Generator(dev).manual_seed(1)
    x = torch.randn(B, C, N, device=dev, generator=g).requires_grad_(True)
    gout = torch.randn(B, O, N, device=dev, generator=g)
    ref_out, ref_gx, ref_gw, ref_gg, ref_gb, bn64 = _reference(conv, bn, x, gout, relu)
    assert pointwise.supported(x, conv, bn)
    out = pointwise.conv_bn_act(x, conv, bn, relu=relu)
    out.backward(gout)
    assert _rel(out, ref_out) < TOL_OUT
    assert float((out.double() - ref_out).abs().max()) < 10 * TOL_OUT
    assert _rel(x.grad, ref_gx) < TOL_GRAD
    assert _rel(conv.weight.grad, ref_gw) < TOL_GRAD
    assert _rel(bn.weight.grad, ref_gg) < TOL_GRAD
    assert _rel(bn.bias.grad, ref_gb) < TOL_GRAD
    assert _rel(bn.running_mean, bn64.running_mean) < 1e-5
    assert _rel(bn.running_var, bn64.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1


def test_layer_with_running_statistics(dev):
    """eval(): the running statistics normalise, BatchNorm's gradient has no batch terms, nothing is updated."""
    from adaptpoint_amd import pointwise
    B, C, O, N = 3, 96, 160, 300
    conv, bn = _layer(C, O, dev, seed=5)
    bn.eval()
    g = torch.Generator(dev).manual_seed(2)
    x = torch.randn(B, C, N, device=dev, generator=g).requires_grad_(True)
    gout = torch.randn(B, O, N, device=dev, generator=g)
    ref_out, ref_gx, ref_gw, ref_gg, ref_gb, _ = _reference(conv, bn, x, gout, True)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    out = pointwise.conv_bn_act(x, conv, bn)
    out.backward(gout)
    TOL_OUT, TOL_GRAD = TOL[pointwise.PRECISION]
    assert _rel(out, ref_out) < TOL_OUT and _rel(x.grad, ref_gx) < TOL_GRAD and _rel(conv.weight.grad, ref_gw) < TOL_GRAD
    assert _rel(bn.weight.grad, ref_gg) < TOL_GRAD and _rel(bn.bias.grad, ref_gb) < TOL_GRAD
    assert torch.equal(bn.running_mean, rm) and torch.equal(bn.running_var, rv) and int(bn.num_batches_tracked) == 0


def test_gradients_are_bit_reproducible(dev):
    from adaptpoint_amd import pointwise
    B, C, O, N = 8, 192, 64, 1024
    conv, bn = _layer(C, O, dev, seed=9)
    g = torch.Generator(dev).manual_seed(3)
    x0 = torch.randn(B, C, N, device=dev, generator=g)
    gout = torch.randn(B, O, N, device=dev, generator=g)
    runs = []
    for _ in range(2):
        x = x0.clone().requires_grad_(True)
        conv.weight.grad = bn.weight.grad = bn.bias.grad = None
        out = pointwise.conv_bn_act(x, conv, bn)
        out.backward(gout)
        runs.append((out.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(),
                     bn.bias.grad.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_module_falls_back_where_the_kernels_do_not_apply(dev):
    """`ConvBNReLU1D(fused=True)` with a bias keeps the reference's three modules; without one it runs the
    kernels and agrees with its own unfused twin (same state_dict)."""
    from adaptpoint_amd.imitator import ConvBNReLU1D
    from adaptpoint_amd import pointwise
    x = torch.randn(2, 16, 200, device=dev, generator=torch.Generator(dev).manual_seed(4))
    biased = ConvBNReLU1D(16, 32, bias=True).to(dev)
    assert not pointwise.supported(x, biased.net[0], biased.net[1])
    assert biased(x).shape == (2, 32, 200)
    fused = ConvBNReLU1D(16, 32, bias=False, fused=True).to(dev)
    plain = ConvBNReLU1D(16, 32, bias=False, fused=False).to(dev)
    plain.load_state_dict(fused.state_dict())
    a, b = fused(x), plain(x)
    assert _rel(a, b.double()) < 1e-5
    assert _rel(fused.net[1].running_var, plain.net[1].running_var.double()) < 1e-5


@pytest.mark.parametrize("shape", [(2, 128, 1024, 1024), (3, 128, 1024, 2048), (2, 37, 70, 333), (4, 5, 130, 64)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("relu", [True, False], ids=["relu", "linear"])
def test_conv_max_matches_float64_composition(dev, shape, relu):
    """convolution + bias (+ ReLU) + max over the points, fused (the discriminator's last group-all layer,
    point_discriminator.py:183-189), against the composed float64 evaluation: output, arg-max consistency, and the
    three gradients (the pooled gradient reaches one position per (cloud, channel))."""
    from adaptpoint_amd import pointwise
    B, C, O, N = shape
    g = torch.Generator(dev).manual_seed(B + C + O + N)
    x = torch.randn(B, C, N, device=dev, generator=g).requires_grad_(True)
    w = (torch.randn(O, C, 1, 1, device=dev, generator=g) / C ** 0.5).requires_grad_(True)
    bias = (0.3 * torch.randn(O, device=dev, generator=g)).requires_grad_(True)
    gout = torch.randn(B, O, device=dev, generator=g)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, bias))
    y = torch.einsum("oc,bcn->bon", w64.view(O, C), x64) + b64.view(1, O, 1)
    ref = (torch.relu(y) if relu else y).amax(dim=2)
    # near-ties of the maximum (and of the ReLU kink) decide where the gradient goes: keep them out of the comparison
    top2 = y.detach().topk(2, dim=2)[0]
    clear = ((top2[..., 0] - top2[..., 1]) > 1e-4) & ((top2[..., 0].abs() > 1e-4) | (not relu))
    gout = gout * clear
    ref.backward(gout.double())
    assert pointwise.conv_max_supported(x, C)
    out = pointwise.conv_max(x, w, bias, relu=relu)
    out.backward(gout)
    assert _rel(out, ref.detach()) < 1e-6
    assert _rel(x.grad, x64.grad) < 5e-6 and _rel(w.grad, w64.grad) < 5e-6 and _rel(bias.grad, b64.grad) < 5e-6


@pytest.mark.parametrize("shape", [(32, 128, 1024), (3, 70, 77), (1, 1, 5), (2, 1024, 128), (5, 64, 1), (4, 65, 129)])
def test_transpose12_is_the_permuted_copy_both_ways(dev, shape):
    """`pointwise.transpose12` (apn_pw_transpose): bit-identical to `permute(0, 2, 1).contiguous()`, gradient included."""
    from adaptpoint_amd import pointwise
    torch.manual_seed(3)
    x = torch.randn(*shape, device=dev, requires_grad=True)
    y = pointwise.transpose12(x)
    assert y.is_contiguous() and torch.equal(y, x.detach().permute(0, 2, 1).contiguous())
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(x.grad, g.permute(0, 2, 1).contiguous())


@pytest.mark.parametrize("n,m", [(1024, 512), (512, 256), (77, 5), (128, 64)])
def test_three_nn_weights_equal_the_composed_form(dev, n, m):
    """`layers.three_nn_weights` (one launch for the weights) against `three_nn` + `inverse_distance_weights` (the
    reference's five elementwise operators, upsampling.py:97-100): same neighbours, weights to the last bit or one ulp."""
    from adaptpoint_amd import layers
    torch.manual_seed(4)
    unknown = torch.rand(3, n, 3, device=dev)
    known = torch.rand(3, m, 3, device=dev)
    known[0, :3] = unknown[0, :3]                      # exact hits: a zero distance (weight ~1, the others ~1e-8 x)
    dist, nearest = layers.three_nn(unknown, known)
    want = layers.inverse_distance_weights(dist)
    got_nearest, got = layers.three_nn_weights(unknown, known)
    assert torch.equal(got_nearest, nearest)
    assert float((got - want).abs().max()) <= 2e-7 and float((got.sum(-1) - 1).abs().max()) <= 3e-7
