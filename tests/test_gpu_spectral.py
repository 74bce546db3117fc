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
