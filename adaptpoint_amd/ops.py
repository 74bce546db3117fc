"""The nine operators of the reference extension `pointnet2_batch_cuda`
(openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24), same names,
positional arguments and buffer-ownership rules, running hand-written gfx950
kernels through the C ABI of libadaptpoint_amd.so.

The caller allocates every output and pre-initialises it where the reference's
Python layer does (temp = 1e10, ball-query idx = 0, gradient targets = 0).
Differences from the reference, all on the error path: argument problems raise
RuntimeError instead of calling exit(-1) (ball_query.cpp:14-26,
sampling_gpu.cu:46-50); kernels are launched on PyTorch's current stream of the
tensors' device rather than on the legacy default stream.
"""
import torch

from . import _lib

__all__ = [
    "ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper",
    "gather_points_wrapper", "gather_points_grad_wrapper",
    "furthest_point_sampling_wrapper", "three_nn_wrapper",
    "three_interpolate_wrapper", "three_interpolate_grad_wrapper",
    "resample_points_wrapper", "zeros",
]


def _chk(t, name, dtype, numel, dev=None):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (got {t.device}); "
                           "the extension has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.numel() < numel:
        raise RuntimeError(f"{name} has {t.numel()} elements, the sizes passed need {numel}")
    if dev is not None and t.device != dev:
        raise RuntimeError(f"{name} is on {t.device}, expected {dev}")
    return t.device


def _launch(fn_name, dev, *args):
    lib = _lib.load()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        code = getattr(lib, fn_name)(*args, stream)
    _lib.check(code, fn_name)


f32, i32 = torch.float32, torch.int32


def zeros(*shape, dtype=torch.float32, device=None):
    """A zero-filled tensor whose fill is a KERNEL on the tensor's current stream.  `torch.zeros` lowers to
    hipMemsetAsync for buffers of a few megabytes; captured into a hipGraph that is a memset node, which this stack
    replays correctly once and with a garbage pattern afterwards (adaptpoint_amd/graphs.py) -- the caller-zeroed
    buffers of the extension's contract (ball-query indices, gradient targets) must survive replay."""
    t = torch.empty(*shape, dtype=dtype, device=device)
    if not t.is_cuda:
        return t.zero_()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return t
    if nbytes % 16 or t.data_ptr() % 16:
        return t.fill_(0)
    _launch("apn_zero_fill", t.device, t.data_ptr(), nbytes)
    return t


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """pointnet2_api.cpp:11 / ball_query.cpp:29-39.  idx (B,M,nsample) pre-zeroed."""
    dev = _chk(new_xyz, "new_xyz", f32, b * m * 3)
    _chk(xyz, "xyz", f32, b * n * 3, dev)
    _chk(idx, "idx", i32, b * m * nsample, dev)
    _launch("apn_ball_query", dev, b, n, m, float(radius), nsample,
            new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr())
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    """pointnet2_api.cpp:12 / group_points.cpp:25-35."""
    dev = _chk(points, "points", f32, b * c * n)
    _chk(idx, "idx", i32, b * npoints * nsample, dev)
    _chk(out, "out", f32, b * c * npoints * nsample, dev)
    _launch("apn_group_points", dev, b, c, n, npoints, nsample,
            points.data_ptr(), idx.data_ptr(), out.data_ptr())
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    """pointnet2_api.cpp:13 / group_points.cpp:13-23.  grad_points pre-zeroed."""
    dev = _chk(grad_out, "grad_out", f32, b * c * npoints * nsample)
    _chk(idx, "idx", i32, b * npoints * nsample, dev)
    _chk(grad_points, "grad_points", f32, b * c * n, dev)
    _launch("apn_group_points_grad", dev, b, c, n, npoints, nsample,
            grad_out.data_ptr(), idx.data_ptr(), grad_points.data_ptr())
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    """pointnet2_api.cpp:15 / sampling.cpp:16-24."""
    dev = _chk(points, "points", f32, b * c * n)
    _chk(idx, "idx", i32, b * npoints, dev)
    _chk(out, "out", f32, b * c * npoints, dev)
    _launch("apn_gather_points", dev, b, c, n, npoints,
            points.data_ptr(), idx.data_ptr(), out.data_ptr())
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    """pointnet2_api.cpp:16 / sampling.cpp:27-36.  grad_points pre-zeroed."""
    dev = _chk(grad_out, "grad_out", f32, b * c * npoints)
    _chk(idx, "idx", i32, b * npoints, dev)
    _chk(grad_points, "grad_points", f32, b * c * n, dev)
    _launch("apn_gather_points_grad", dev, b, c, n, npoints,
            grad_out.data_ptr(), idx.data_ptr(), grad_points.data_ptr())
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    """pointnet2_api.cpp:18 / sampling.cpp:39-48.  temp (B,N) pre-filled with 1e10."""
    dev = _chk(points, "points", f32, b * n * 3)
    _chk(temp, "temp", f32, b * n, dev)
    _chk(idx, "idx", i32, b * m, dev)
    _launch("apn_furthest_point_sampling", dev, b, n, m,
            points.data_ptr(), temp.data_ptr(), idx.data_ptr())
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    """pointnet2_api.cpp:20 / interpolate.cpp:20-28.  Writes SQUARED distances."""
    dev = _chk(unknown, "unknown", f32, b * n * 3)
    _chk(known, "known", f32, b * m * 3, dev)
    _chk(dist2, "dist2", f32, b * n * 3, dev)
    _chk(idx, "idx", i32, b * n * 3, dev)
    _launch("apn_three_nn", dev, b, n, m,
            unknown.data_ptr(), known.data_ptr(), dist2.data_ptr(), idx.data_ptr())


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    """pointnet2_api.cpp:21 / interpolate.cpp:31-43.  Argument order (b, c, m, n)."""
    dev = _chk(points, "points", f32, b * c * m)
    _chk(idx, "idx", i32, b * n * 3, dev)
    _chk(weight, "weight", f32, b * n * 3, dev)
    _chk(out, "out", f32, b * c * n, dev)
    _launch("apn_three_interpolate", dev, b, c, m, n,
            points.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr())


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    """pointnet2_api.cpp:22 / interpolate.cpp:45-57.  Argument order (b, c, n, m)."""
    dev = _chk(grad_out, "grad_out", f32, b * c * n)
    _chk(idx, "idx", i32, b * n * 3, dev)
    _chk(weight, "weight", f32, b * n * 3, dev)
    _chk(grad_points, "grad_points", f32, b * c * m, dev)
    _launch("apn_three_interpolate_grad", dev, b, c, n, m,
            grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), grad_points.data_ptr())


# ---- operators of this build that the reference extension does not have (same calling style) ----
def resample_points_wrapper(b, n, c, p_all, s_cnt, cx, points, fidx, choice, pos, x):
    """The training loop's resampler (examples/classification/train_autoaug.py:493-501) in one
    launch: pos (B,s_cnt,3) = points[b, fidx[b, choice[s]], :3], x (B,cx,s_cnt) = its first cx
    channels, channel-major.  points (B,N,C), fidx (B,p_all) int32, choice (s_cnt) int32."""
    dev = _chk(points, "points", f32, b * n * c)
    _chk(fidx, "fidx", i32, b * p_all, dev)
    _chk(choice, "choice", i32, s_cnt, dev)
    _chk(pos, "pos", f32, b * s_cnt * 3, dev)
    _chk(x, "x", f32, b * cx * s_cnt, dev)
    _launch("apn_resample_points", dev, b, n, c, p_all, s_cnt, cx, points.data_ptr(), fidx.data_ptr(),
            choice.data_ptr(), pos.data_ptr(), x.data_ptr())
    return 1
