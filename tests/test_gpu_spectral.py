"""csrc/spectral.hip (adaptpoint_amd.spectral) against torch.nn.utils.parametrizations.spectral_norm itself -- the
parametrisation the reference's discriminator layers carry (point_discriminator.py:17-73, 149-191) -- on the same
weights and the same power-iteration state: normalised weight, updated `_u` / `_v`, gradient, eval mode, and the GAN
pattern of two forwards before one backward."""
import copy

import pytest
import torch
import torch.nn as nn
from torch.nn.utils.parametrizations import spectral_norm as torch_spectral_norm

pytestmark = pytest.mark.gpu

# the discriminator's seven layers (3 -> 64 -> 128 -> 1024 convolutions, 1024 -> 512 -> 256 -> 15 -> 1 linears)
LAYERS = [("conv", 3, 64), ("conv", 64, 128), ("conv", 128, 1024), ("lin", 1024, 512), ("lin", 512, 256),
          ("lin", 256, 15), ("lin", 15, 1), ("lin", 77, 130)]


def _pair(kind, cin, cout, dev, seed):
    from adaptpoint_amd.spectral import spectral_norm
    torch.manual_seed(seed)
    base = nn.Conv2d(cin, cout, 1) if kind == "conv" else nn.Linear(cin, cout)
    ref = torch_spectral_norm(copy.deepcopy(base))
    mine = spectral_norm(copy.deepcopy(base))
    mine.load_state_dict(ref.state_dict())                      # same original weight, same _u / _v
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    return ref.to(dev), mine.to(dev)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "%s%dx%d" % l)
def test_training_forward_backward_and_state(dev, layer):
    ref, mine = _pair(*layer, dev, seed=3)
    # (every read of `.weight` in training mode is a power iteration: shapes come from `.original`)
    g = torch.randn(ref.parametrizations.weight.original.shape, device=dev, generator=torch.Generator(dev).manual_seed(1))
    for m in (ref, mine):
        m.train()
        (m.weight * g).sum().backward()                         # one power iteration, then the gradient
    pr, pm = ref.parametrizations.weight, mine.parametrizations.weight
    assert _rel(pm[0]._u, pr[0]._u) < 1e-5 and _rel(pm[0]._v, pr[0]._v) < 1e-5
    assert _rel(pm.original.grad, pr.original.grad) < 1e-5
    with torch.no_grad():                                       # a second iteration from the updated state
        assert _rel(mine.weight, ref.weight) < 1e-5
    for m in (ref, mine):
        m.eval()
    u0 = pm[0]._u.clone()
    with torch.no_grad():
        assert _rel(mine.weight, ref.weight) < 1e-5
    assert torch.equal(pm[0]._u, u0)                            # eval: no power iteration


def test_two_forwards_then_one_backward(dev):
    """loss = f(W after iteration 1) + f(W after iteration 2): each forward's gradient uses the vectors of ITS
    iteration (PyTorch clones them for exactly this GAN pattern)."""
    ref, mine = _pair("lin", 256, 64, dev, seed=5)
    gen = torch.Generator(dev).manual_seed(2)
    shape = ref.parametrizations.weight.original.shape
    g1 = torch.randn(shape, device=dev, generator=gen)
    g2 = torch.randn(shape, device=dev, generator=gen)
    for m in (ref, mine):
        m.train()
        ((m.weight * g1).sum() + (m.weight * g2).sum()).backward()
    assert _rel(mine.parametrizations.weight.original.grad, ref.parametrizations.weight.original.grad) < 1e-5
    assert _rel(mine.parametrizations.weight[0]._v, ref.parametrizations.weight[0]._v) < 1e-5


def test_results_are_bit_reproducible(dev):
    """fixed summation orders in every kernel: the same state gives the same bits"""
    _, mine = _pair("lin", 1024, 512, dev, seed=7)
    state = copy.deepcopy(mine.state_dict())
    g = torch.randn(512, 1024, device=dev, generator=torch.Generator(dev).manual_seed(3))
    runs = []
    for _ in range(2):
        mine.load_state_dict(state)
        mine.zero_grad(set_to_none=True)
        w = mine.weight
        (w * g).sum().backward()
        p = mine.parametrizations.weight
        runs.append((w.detach().clone(), p.original.grad.clone(), p[0]._u.clone(), p[0]._v.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("training", [True, False])
def test_all_layers_at_once_equal_layer_by_layer(dev, training):
    """`spectral.normalise_many` (3 + 2 launches for the discriminator's seven layers) against the per-layer kernels on the
    same state: normalised weights, the updated `_u` / `_v`, the gradients -- bit for bit (same arithmetic, same orders);
    two passes, so that the second starts from the first one's power-iteration state; a layer whose gradient is not asked
    for gets none."""
    from adaptpoint_amd.spectral import normalise_many
    one = [_pair(*layer, dev, seed=11 + i)[1] for i, layer in enumerate(LAYERS[:7])]
    many = [copy.deepcopy(m) for m in one]
    gen = torch.Generator(dev).manual_seed(9)
    for rep in range(2):
        gs = [torch.randn(m.parametrizations.weight.original.shape, device=dev, generator=gen) for m in one]
        for m in one + many:
            m.train(training)
            m.zero_grad(set_to_none=True)
        w_one = [m.weight for m in one]
        w_many = normalise_many(many)
        assert w_many is not None and len(w_many) == 7
        skip = 4                                                 # this layer's weight takes no part in the loss
        sum((w * g).sum() for i, (w, g) in enumerate(zip(w_one, gs)) if i != skip).backward()
        sum((w * g).sum() for i, (w, g) in enumerate(zip(w_many, gs)) if i != skip).backward()
        for i, (a, b) in enumerate(zip(one, many)):
            pa, pb = a.parametrizations.weight, b.parametrizations.weight
            assert torch.equal(w_one[i], w_many[i]) and w_many[i].shape == pa.original.shape
            assert torch.equal(pa[0]._u, pb[0]._u) and torch.equal(pa[0]._v, pb[0]._v)
            if i == skip:
                assert pa.original.grad is None and pb.original.grad is None
            else:
                assert torch.equal(pa.original.grad, pb.original.grad)


def test_discriminator_forward_uses_one_set_of_launches_and_matches(dev):
    """PointDiscriminator1(fused=True) normalises its seven weights at once; output, gradient w.r.t. the cloud, weight
    gradients and the buffers equal the layer-by-layer path of the same module (normalise_many switched off)."""
    from adaptpoint_amd import spectral
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    torch.manual_seed(0)
    a = fill_parameters_by_name(PointDiscriminator1(num_classes=15, fused=True)).to(dev).train()
    b = copy.deepcopy(a)
    for m in list(a.modules()) + list(b.modules()):
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    x1 = torch.rand(4, 512, 3, device=dev).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    ya = a(x1)
    keep = spectral.normalise_many
    spectral.normalise_many = lambda mods, name="weight": None
    try:
        yb = b(x2)
    finally:
        spectral.normalise_many = keep
    ya.sum().backward()
    yb.sum().backward()
    assert torch.equal(ya, yb) and torch.equal(x1.grad, x2.grad)
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and torch.equal(pa.grad, pb.grad), na
    for (na, ba), (nb, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(ba, bb), na
