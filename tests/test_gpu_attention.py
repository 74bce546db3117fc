"""the imitator's Anchor_selfattention core fused (csrc/attention.hip) --
GPU parity against the float64 oracle and against the reference module's own outputs (G7)."""
import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _mirror(golden, dev):
    from adaptpoint_amd.attention import AnchorSelfAttention
    m = AnchorSelfAttention(dim=64, head_num=4).to(dev)
    state = {k.split("/", 1)[1]: torch.from_numpy(np.asarray(golden[k]))
             for k in golden.files if k.startswith("g7_att_state/")}
    m.load_state_dict(state)          # the reference's parameter / buffer names
    m.train()
    return m


@pytest.mark.parametrize("B,M,H", [(2, 64, 4), (3, 1024, 4), (1, 2048, 2), (2, 288, 1), (2, 32, 4)])
def test_attention_matches_float64_oracle(dev, oracle, B, M, H):
    """Split-operand MFMA products: outputs within 5e-5, gradients within 5e-5 relative L2 of a
    float64 evaluation (PyTorch's float32 composition sits at ~2e-6 / 4e-7)."""
    from adaptpoint_amd.attention import attention
    q, k, v, w = (GI.seeded_normal((B, M, H * 16), seed=80 + i) for i in range(4))
    want = oracle.attention(q, k, v, H)
    dq, dk, dv = oracle.attention_grad(q, k, v, H, w)
    tq, tk, tv = (torch.from_numpy(t).to(dev).requires_grad_(True) for t in (q, k, v))
    out = attention(tq, tk, tv, H)
    assert np.abs(out.detach().cpu().numpy() - want).max() <= 5e-5
    out.backward(torch.from_numpy(w).to(dev))
    for got, ref in ((tq.grad, dq), (tk.grad, dk), (tv.grad, dv)):
        got = got.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(got - ref) <= 5e-5 * np.linalg.norm(ref)


def test_attention_large_scores_and_fallback_shapes(dev, oracle):
    from adaptpoint_amd.attention import attention, supported
    # scores of magnitude ~40: the running-max recurrence must not overflow / lose the tail
    q, k, v = (GI.seeded_normal((2, 128, 64), seed=90 + i) for i in range(3))
    q = (q * 6).astype(np.float32)
    out = attention(*(torch.from_numpy(t).to(dev) for t in (q, k, v)), 4)
    assert np.abs(out.cpu().numpy() - oracle.attention(q, k, v, 4)).max() <= 5e-4
    # M not a multiple of 32: the reference's composition runs instead (same result)
    q, k, v = (torch.from_numpy(GI.seeded_normal((2, 50, 64), seed=95 + i)).to(dev) for i in range(3))
    assert not supported(q, 4)
    from adaptpoint_amd import attention as A
    before = sum(A.COMPOSED_CALLS.values())
    out = attention(q, k, v, 4)
    assert sum(A.COMPOSED_CALLS.values()) == before + 1 and any("M=50" in k_ for k_ in A.COMPOSED_CALLS)   # counted, not silent
    assert np.abs(out.cpu().numpy() - oracle.attention(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), 4)).max() <= 1e-5
    with pytest.raises(RuntimeError):
        attention(q.cpu(), k.cpu(), v.cpu(), 4)


@pytest.mark.parametrize("B,M,H", [(32, 4, 4), (3, 1, 2), (2, 7, 4), (5, 31, 1), (2, 24, 16)])
def test_attention_for_few_points_matches_float64(dev, B, M, H):
    """m <= 32 that is no multiple of 32 (the imitator's 4-anchor head, generator_component4_15.py:572): the one-wave
    float32 kernel, forward and the three gradients, against the composition in float64; not counted as composed;
    two runs bit-identical (fixed-order sums)."""
    from adaptpoint_amd import attention as A
    q, k, v, w = (torch.from_numpy(GI.seeded_normal((B, M, 16 * H), seed=300 + i)).to(dev) for i in range(4))
    q = q * 2.0
    before = sum(A.COMPOSED_CALLS.values())
    outs = []
    for _ in range(2):
        a = [t.clone().requires_grad_(True) for t in (q, k, v)]
        o = A.attention(*a, H)
        (o * w).sum().backward()
        outs.append([o.detach()] + [t.grad for t in a])
    assert sum(A.COMPOSED_CALLS.values()) == before
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    d = [t.double().clone().requires_grad_(True) for t in (q, k, v)]
    ref = A._reference(*d, H)
    (ref * w.double()).sum().backward()
    for got, want in zip(outs[0], [ref.detach()] + [t.grad for t in d]):
        assert float((got.double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


def test_anchor_self_attention_module_matches_reference_golden(dev, golden):
    """The mirror module on the GPU (fused attention core) against the reference's
    Anchor_selfattention run on CPU (tests/golden/make_golden.py, G7)."""
    m = _mirror(golden, dev)
    x = torch.from_numpy(GI.seeded_normal((2, 64, 64), seed=71)).to(dev).requires_grad_(True)
    xyz = torch.from_numpy(GI.unit_sphere_cloud(2, 64, seed=72)).to(dev)
    out = m(x, xyz)
    (out * torch.from_numpy(GI.seeded_normal(tuple(out.shape), seed=73)).to(dev)).sum().backward()
    for got, key in ((out, "g7_att_out"), (x.grad, "g7_att_grad_x"), (m.to_qkv.weight.grad, "g7_att_grad_qkv_w")):
        ref = golden[key]
        assert np.abs(got.detach().cpu().numpy() - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), key
