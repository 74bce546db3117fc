"""Seeded input generators shared by tests/golden/make_golden.py and the tests.
Fixtures store outputs only; these functions regenerate the matching inputs."""
import numpy as np
import torch

from adaptpoint_amd.synthetic import (seeded_normal, seeded_uniform, sphere_surface_cloud,  # noqa: F401
                                      unit_sphere_cloud)


def config1_xyz():
    """BASELINE.json configs[0]: torch.manual_seed(0); rand(2,1024,3)*2-1."""
    g = torch.Generator().manual_seed(0)
    return (torch.rand(2, 1024, 3, generator=g) * 2 - 1).numpy()


def take_points(xyz, idx):
    """(B,N,3), (B,M) -> (B,M,3)"""
    return np.take_along_axis(xyz, idx[..., None].astype(np.int64).repeat(3, -1), 1).copy()


def three_nn_weights(dist2):
    """upsampling.py:97-100."""
    dist = np.sqrt(dist2)
    r = 1.0 / (dist + 1e-8)
    return (r / r.sum(-1, keepdims=True)).astype(np.float32)


def tie_cases():
    """(name, cloud (B,N,3) float32, m): exact-tie and ragged-size FPS cases."""
    cases = []
    base = seeded_uniform((2, 1024, 3), 21).astype(np.float32)
    dup = base.copy()
    dup[:, 512:] = dup[:, :512]                       # every point duplicated once
    cases.append(("dup", dup, 600))
    half0 = base.copy()
    half0[:, ::2] = 0.0                               # AdaptPoint mask pattern: half at the origin
    cases.append(("half_origin", half0, 700))
    cases.append(("all_same", np.full((2, 300, 3), 0.25, np.float32), 40))
    lattice = np.stack(np.meshgrid(*[np.arange(8, dtype=np.float32)] * 3, indexing="ij"), -1)
    cases.append(("lattice512", np.tile(lattice.reshape(1, 512, 3), (2, 1, 1)) * 0.125, 256))
    cases.append(("n1200", seeded_uniform((2, 1200, 3), 22).astype(np.float32), 400))
    n2048 = seeded_uniform((2, 2048, 3), 23).astype(np.float32)
    cases.append(("n2048_m1200", n2048, 1200))
    q = np.round(seeded_uniform((2, 2048, 3), 24) * 4).astype(np.float32) / 4   # coarse grid: many ties
    cases.append(("n2048_grid", q, 300))
    cases.append(("n1000_grid", q[:, :1000].copy(), 1000))
    cases.append(("n100_m_gt_n", seeded_uniform((2, 100, 3), 25).astype(np.float32), 130))
    cases.append(("n5", seeded_uniform((3, 5, 3), 26).astype(np.float32), 5))
    cases.append(("n1", seeded_uniform((2, 1, 3), 27).astype(np.float32), 3))
    cases.append(("n3000_grid", np.round(seeded_uniform((1, 3000, 3), 28) * 8).astype(np.float32) / 8, 200))  # dyadic: exact in every rounding
    return cases


def gradient_sample_index(name, numel, keep=8192):
    """Which entries of a parameter's gradient the B = 8 classifier goldens store (tests/golden/classifier_b8_golden.npz):
    all of them up to `keep`, else `keep` entries drawn without replacement by a generator seeded with the parameter's
    NAME (sorted) -- the relative L2 error over such a subset is an unbiased estimate of the whole tensor's, and the
    fixture stays below 2 MB instead of 11."""
    import zlib
    if numel <= keep:
        return np.arange(numel)
    rng = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return np.sort(rng.choice(numel, keep, replace=False))
