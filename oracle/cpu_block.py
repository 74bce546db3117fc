"""oracle/cpu_block.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

`CpuOps` stands the C oracle in for the HIP extension at the level of the
`adaptpoint_amd.ops.*_wrapper` functions (the Python face of the C ABI): inside the context
every host-side module of the product -- the operator layer, SetAbstraction, the PointNeXt-S
classifier, the imitator, the discriminator, the training steps -- runs on CPU tensors exactly as
written, with torch-CPU doing conv / BatchNorm / ReLU / max.  That is what the reference's
Python would execute if its extension had a CPU build; it is how the CPU suite checks the
mirrors against the reference-generated goldens, and it is the `cpu_baseline` leg of bench.py.
Only tests/, __graft_entry__.smoke() and that leg import this file.
"""
import numpy as np
import torch

from . import oracle as O


def _np(t):
    return t.detach().numpy()


def _put(dst, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).view_as(dst))


# The nine wrappers of openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:11-23 over the oracle:
# same positional signatures, caller-owned and caller-initialised buffers.
def _ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx):
    hits = O.ball_query(radius, nsample, _np(xyz), _np(new_xyz))
    # rows of empty balls: the oracle front-end starts from zeros, as every caller of this layer does (group.py:194)
    _put(idx, hits)
    return 1


def _group_points(b, c, n, npoints, nsample, points, idx, out):
    _put(out, O.group_points(_np(points), _np(idx)))
    return 1


def _group_points_grad(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    grad_points.add_(torch.from_numpy(O.group_points_grad(_np(grad_out), _np(idx), n)))
    return 1


def _gather_points(b, c, n, npoints, points, idx, out):
    _put(out, O.gather_points(_np(points), _np(idx)))
    return 1


def _gather_points_grad(b, c, n, npoints, grad_out, idx, grad_points):
    grad_points.add_(torch.from_numpy(O.gather_points_grad(_np(grad_out), _np(idx), n)))
    return 1


def _furthest_point_sampling(b, n, m, points, temp, idx):
    picks, running_min = O.furthest_point_sampling(_np(points), m, return_temp=True)
    _put(idx, picks)
    _put(temp, running_min)
    return 1


def _three_nn(b, n, m, unknown, known, dist2, idx):
    d2, nearest = O.three_nn(_np(unknown), _np(known))
    _put(dist2, d2)
    _put(idx, nearest)


def _three_interpolate(b, c, m, n, points, idx, weight, out):
    _put(out, O.three_interpolate(_np(points), _np(idx), _np(weight)))


def _three_interpolate_grad(b, c, n, m, grad_out, idx, weight, grad_points):
    grad_points.add_(torch.from_numpy(O.three_interpolate_grad(_np(grad_out), _np(idx), _np(weight), m)))


def _resample_points(b, n, c, p_all, s_cnt, cx, points, fidx, choice, pos, x):
    po, xo = O.resample_points(_np(points), _np(fidx), _np(choice), cx)
    _put(pos, po)
    _put(x, xo)
    return 1


_WRAPPERS = {
    "resample_points_wrapper": _resample_points,
    "ball_query_wrapper": _ball_query,
    "group_points_wrapper": _group_points,
    "group_points_grad_wrapper": _group_points_grad,
    "gather_points_wrapper": _gather_points,
    "gather_points_grad_wrapper": _gather_points_grad,
    "furthest_point_sampling_wrapper": _furthest_point_sampling,
    "three_nn_wrapper": _three_nn,
    "three_interpolate_wrapper": _three_interpolate,
    "three_interpolate_grad_wrapper": _three_interpolate_grad,
}


class CpuOps:
    """Context manager: `adaptpoint_amd.ops`' nine wrappers -> the oracle (CPU tensors)."""

    def __enter__(self):
        from adaptpoint_amd import ops
        self._saved = {k: getattr(ops, k) for k in _WRAPPERS}
        for k, fn in _WRAPPERS.items():
            setattr(ops, k, fn)
        return self

    def __exit__(self, *exc):
        from adaptpoint_amd import ops
        for k, fn in self._saved.items():
            setattr(ops, k, fn)
        return False


def build_cpu_block(make_block):
    """make_block() -> SetAbstraction (kept for bench.py's cpu_baseline leg)."""
    return make_block()


def run_step(blk, p, f):
    """One forward+backward of the block on CPU tensors, ops from the oracle."""
    with CpuOps():
        f = f.detach().requires_grad_(True)
        for prm in blk.parameters():
            prm.grad = None
        new_p, out = blk([p, f])
        out.sum().backward()
    return new_p, out, f.grad
