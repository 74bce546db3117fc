"""oracle/cpu_block.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

A CPU set-abstraction block for the `cpu_baseline` leg of bench.py and for
tests: the SAME host-side module as the product (adaptpoint_amd.set_abstraction
.SetAbstraction, a mirror of openpoints/models/backbone/pointnext.py:82-170)
with its five extension-backed symbols swapped for autograd wrappers around the
C oracle, and torch-CPU doing conv / BatchNorm / ReLU / max -- i.e. what the
reference's Python would execute if its extension had a CPU build.
"""
import numpy as np
import torch

from . import oracle as O


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def furthest_point_sample(xyz, npoint):
    return _t(O.furthest_point_sampling(xyz.detach().numpy(), npoint))


def ball_query(radius, nsample, xyz, new_xyz):
    return _t(O.ball_query(radius, nsample, xyz.detach().numpy(), new_xyz.detach().numpy()))


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return _t(O.group_points(features.detach().numpy(), idx.numpy()))

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _t(O.group_points_grad(g.contiguous().numpy(), idx.numpy(), ctx.n)), None


grouping_operation = _Group.apply


class CpuOps:
    """Context manager: route adaptpoint_amd.layers' extension-backed symbols to the oracle."""

    def __enter__(self):
        from adaptpoint_amd import layers, set_abstraction
        self._saved = (layers.ball_query, layers.grouping_operation,
                       set_abstraction.furthest_point_sample)
        layers.ball_query = ball_query
        layers.grouping_operation = grouping_operation
        set_abstraction.furthest_point_sample = furthest_point_sample
        return self

    def __exit__(self, *exc):
        from adaptpoint_amd import layers, set_abstraction
        layers.ball_query, layers.grouping_operation, set_abstraction.furthest_point_sample = self._saved
        return False


def build_cpu_block(make_block):
    """make_block() -> SetAbstraction; returns it bound to the oracle ops (CPU)."""
    with CpuOps():
        blk = make_block()
        blk.sample_fn = furthest_point_sample
    return blk


def run_step(blk, p, f):
    """One forward+backward of the block on CPU tensors, ops from the oracle."""
    with CpuOps():
        f = f.detach().requires_grad_(True)
        for prm in blk.parameters():
            prm.grad = None
        new_p, out = blk([p, f])
        out.sum().backward()
    return new_p, out, f.grad
