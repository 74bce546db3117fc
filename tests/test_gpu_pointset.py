"""the imitator's PointsetGrouper grouping stage, fused
(adaptpoint_amd/csrc/pointset_group.hip) -- GPU parity against the numpy oracle and against the
reference module's own outputs (tests/golden, G6)."""
import os
import sys

import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _case(B, N, M, K, C, seed):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((B, N, C)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, K)).astype(np.int32)
    idx[:, :, K // 2:] = idx[:, :, :1]                     # ball-query style fill: repeats of slot 0
    fidx = rng.integers(0, N, (B, M)).astype(np.int32)
    alpha = rng.standard_normal(C).astype(np.float32)
    alpha[::7] = 0.0                                        # all positions tie -> position 0
    beta = rng.standard_normal(C).astype(np.float32)
    return pts, idx, fidx, alpha, beta


@pytest.mark.parametrize("B,N,M,K,C", [(4, 1024, 512, 24, 128), (2, 128, 64, 24, 1024), (3, 200, 37, 5, 16),
                                       (2, 64, 16, 24, 4), (1, 512, 256, 255, 64)])
def test_group_max_forward_is_exact_and_backward_matches(dev, oracle, B, N, M, K, C):
    from adaptpoint_amd.pointset import group_max
    pts, idx, fidx, alpha, beta = _case(B, N, M, K, C, seed=B * 1000 + C)
    want, want_k = oracle.pointset_group_max(pts, idx, fidx, alpha, beta)
    P = torch.from_numpy(pts).to(dev).requires_grad_(True)
    al = torch.from_numpy(alpha).to(dev).requires_grad_(True)
    be = torch.from_numpy(beta).to(dev).requires_grad_(True)
    out = group_max(P, torch.from_numpy(idx).to(dev), torch.from_numpy(fidx).to(dev), al, be)
    assert np.array_equal(out.detach().cpu().numpy(), want)         # bit-exact
    w = GI.seeded_normal(tuple(want.shape), seed=5)
    (out * torch.from_numpy(w).to(dev)).sum().backward()
    gp, ga, gb = oracle.pointset_group_max_grad(pts, idx, fidx, alpha, want_k, w)
    # float atomics / float partial sums vs float64: 1e-5 of the largest entry
    for got, ref in ((P.grad, gp), (al.grad, ga), (be.grad, gb)):
        got = got.detach().cpu().numpy().astype(np.float64).reshape(ref.shape)
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_group_max_rejects_cpu_and_unsupported(dev):
    from adaptpoint_amd.pointset import group_max, group_max_supported
    pts = torch.zeros(1, 8, 8)
    with pytest.raises(RuntimeError):
        group_max(pts, torch.zeros(1, 4, 2, dtype=torch.int32), torch.zeros(1, 4, dtype=torch.int32),
                  torch.ones(8), torch.zeros(8))
    assert not group_max_supported(torch.zeros(1, 8, 24, device=dev), 24)      # C not a power of two
    with pytest.raises(RuntimeError):
        group_max(torch.zeros(1, 8, 24, device=dev), torch.zeros(1, 4, 2, dtype=torch.int32, device=dev),
                  torch.zeros(1, 4, dtype=torch.int32, device=dev), torch.ones(24, device=dev),
                  torch.zeros(24, device=dev))


def test_pointset_grouper_module_matches_reference_golden(dev, golden):
    """The mirror module on the GPU (FPS, ball query, fused group-max) against the reference's
    PointsetGrouper run on CPU over the oracle ops (tests/golden/make_golden.py, G6)."""
    from adaptpoint_amd.pointset import PointsetGrouper
    g = PointsetGrouper(channel=64, reduce=2, kneighbors=24, radi=0.2, normalize="anchor").to(dev)
    with torch.no_grad():
        g.affine_alpha.copy_(torch.from_numpy(GI.seeded_normal((1, 1, 1, 64), seed=61)))
        g.affine_beta.copy_(torch.from_numpy(GI.seeded_normal((1, 1, 1, 64), seed=62)))
    xyz = torch.from_numpy(GI.unit_sphere_cloud(2, 512, seed=63)).to(dev)
    pts = torch.from_numpy(GI.seeded_normal((2, 512, 64), seed=64)).to(dev).requires_grad_(True)
    new_xyz, out = g(xyz, pts)
    assert np.array_equal(new_xyz.detach().cpu().numpy(), golden["g6_pg_new_xyz"])
    assert np.array_equal(out.detach().cpu().numpy(), golden["g6_pg_out"])
    w = torch.from_numpy(GI.seeded_normal(tuple(out.shape), seed=65)).to(dev)
    (out * w).sum().backward()
    for got, key in ((pts.grad, "g6_pg_grad_points"), (g.affine_alpha.grad, "g6_pg_grad_alpha"),
                     (g.affine_beta.grad, "g6_pg_grad_beta")):
        ref = golden[key]
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), key
