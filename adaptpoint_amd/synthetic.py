"""Seeded synthetic inputs of the benchmark workloads: point clouds of the two
distributions every measured number is quoted on, and seeded normal / uniform tensors.  Pure functions of
their seed (torch CPU generators), so every rank, test and profile run sees the same data."""
import numpy as np
import torch


def seeded_normal(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).numpy()


def seeded_uniform(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1).numpy()


def unit_sphere_cloud(b, n, seed):
    """D1 of SURVEY.md section 8d: uniform cube, centred, scaled into the unit sphere
    (the arithmetic of PointCloudCenterAndNormalize, openpoints/transforms/point_transformer_gpu.py:55-60)."""
    x = seeded_uniform((b, n, 3), seed).astype(np.float32)
    x = x - x.mean(axis=1, keepdims=True)
    m = np.sqrt((x ** 2).sum(-1, keepdims=True)).max(axis=1, keepdims=True)
    return (x / m).astype(np.float32)


def sphere_surface_cloud(b, n, seed):
    """D2: unit-sphere surface + N(0, 0.01) jitter (scan-like)."""
    x = seeded_normal((b, n, 3), seed).astype(np.float32)
    x = x / np.sqrt((x ** 2).sum(-1, keepdims=True))
    return (x + 0.01 * seeded_normal((b, n, 3), seed + 1000)).astype(np.float32)
