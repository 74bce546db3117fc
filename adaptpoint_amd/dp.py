"""Data-parallel plumbing shared by bench.py and the tests: one process per GPU,
clouds sharded by rank (no data-path collective: every operator of the hot path is
independent per cloud), gradient all-reduce through
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).
"""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend, device=None, force=False):
    world, rank, _ = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if device is not None and device.type == "cuda":
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return world, rank


def all_reduce_sum_(t):
    """dist.all_reduce(t) (SUM, in place) on whatever backend the group has.  Device tensors over a gloo group (two ranks
    sharing ONE GPU in tests/test_gpu_syncbn_two_ranks.py: RCCL refuses two ranks on one device) are staged through the
    host -- the collective itself is `torch.distributed.all_reduce` either way, so `workloads.count_collectives` counts it."""
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host)
        t.copy_(host)
    else:
        dist.all_reduce(t)
    return t


def fence(device=None):
    """barrier + device synchronize: brackets the timed region on both sides."""
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device=None):
    if not dist.is_initialized():
        return seconds
    dev = device if device is not None and device.type == "cuda" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(step, steps, warmup, device=None):
    """W untimed warm-up steps, then exactly `steps` timed ones between two fences;
    returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step()
    fence(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence(device)
    return max_over_ranks(time.perf_counter() - t0, device)


def shard_seed(base_seed, rank):
    """Each rank draws its own clouds: global batch = world * per-rank batch."""
    return base_seed + 1000 * rank


def host_threads(cap=None):
    """Threads this process may actually use (affinity / cgroup aware)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if cap:
        n = min(n, cap)
    return max(1, n)


def _common_span(tensors):
    """One 1-D view covering all `tensors` when they are contiguous views of ONE storage packed
    closely together (the fused block's backward carves every parameter gradient out of a
    single buffer), else None."""
    t0 = tensors[0]
    base = t0.untyped_storage().data_ptr()
    for t in tensors:
        if (t.untyped_storage().data_ptr() != base or not t.is_contiguous() or t.dtype != t0.dtype
                or t.device != t0.device):
            return None
    lo = min(t.storage_offset() for t in tensors)
    hi = max(t.storage_offset() + t.numel() for t in tensors)
    payload = sum(t.numel() for t in tensors)
    # the span is reduced (and scaled) WHOLE: only the alignment padding of the producer's carve (64 elements per
    # view, adaptpoint_amd.fused._carve) may lie between the views -- anything else living in that storage
    # would be averaged with them
    if hi - lo > payload + 64 * len(tensors):
        return None
    return torch.empty(0, dtype=t0.dtype, device=t0.device).set_(t0.untyped_storage(), lo, (hi - lo,))


def allreduce_mean_(tensors):
    """Average a list of gradient tensors over ranks in place with ONE collective (RCCL over
    xGMI on ROCm).  The whole parameter set of a set-abstraction block is ~22 KB, so a single
    latency-bound message per step is the right shape for point-to-point xGMI links (no
    bucketing, no overlap machinery).  Gradients that already share one buffer are reduced in
    place (no flatten / copy-back launches); otherwise: flatten, all-reduce, copy back."""
    if not dist.is_initialized() or not tensors:
        return
    world = dist.get_world_size()
    span = _common_span(tensors)
    if span is not None:
        if dist.get_backend() == "nccl":
            dist.all_reduce(span, op=dist.ReduceOp.AVG)
        else:
            all_reduce_sum_(span)
            if world > 1:
                span.div_(world)
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    all_reduce_sum_(flat)
    if world > 1:
        flat.div_(world)
    off = 0
    views = []
    for t in tensors:
        views.append(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    torch._foreach_copy_(list(tensors), views)


# ---------------------------------------------------------------------------------------------
# SyncBatchNorm for the UNFUSED path, built for latency-bound links.
#
# The reference forces torch.nn.SyncBatchNorm at world_size > 1 (examples/classification/main.py:27,
# train_autoaug.py:275-276).  That module all-GATHERS (mean, invstd, count) per layer and is
# GPU-only.  The statistics exchange of a BatchNorm layer is two tiny vectors, so here it is ONE
# all-reduce(SUM) of [sum, sum of squares, count] in the forward and ONE of [sum dy, sum dy*(x-mean)]
# in the backward -- the same exchange the fused block performs (adaptpoint_amd.fused) -- on
# whatever backend the process group has (RCCL over xGMI; gloo in the CPU tests).  Gradient
# semantics are torch.nn.SyncBatchNorm's: dL/dx uses the global sums; dL/dgamma, dL/dbeta stay
# rank-local and are averaged with the other parameters' gradients.
# ---------------------------------------------------------------------------------------------
FORCE_COLLECTIVES = False      # testing hook: issue the statistics all-reduces at world size 1 too (one-GPU RCCL runs)


def _sum_over_ranks_(t):
    if dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES):
        all_reduce_sum_(t)
    return t


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        xd = x.double()
        stat = torch.cat([xd.sum(dims), (xd * xd).sum(dims),
                          torch.full((1,), float(x.numel() // C), dtype=torch.float64, device=x.device)])
        _sum_over_ranks_(stat)
        count = stat[-1]
        mean = stat[:C] / count
        var = (stat[C:2 * C] / count - mean * mean).clamp_min_(0.0)
        invstd = torch.rsqrt(var + eps)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = ((xd - mean.view(shape)) * invstd.view(shape)).to(x.dtype)
        ctx.save_for_backward(xhat, weight, invstd.to(x.dtype))
        ctx.count, ctx.dims, ctx.shape = count, dims, shape
        y = xhat if weight is None else xhat * weight.view(shape)
        if bias is not None:
            y = y + bias.view(shape)
        ctx.mark_non_differentiable(mean, var, count)
        return y, mean, var, count

    @staticmethod
    def backward(ctx, dy, *unused):
        xhat, weight, invstd = ctx.saved_tensors
        C = xhat.shape[1]
        shape, dims = ctx.shape, ctx.dims
        dyd = dy.double()
        local = torch.cat([dyd.sum(dims), (dyd * xhat.double()).sum(dims)])
        g_weight = local[C:].to(dy.dtype) if weight is not None else None
        g_bias = local[:C].to(dy.dtype)
        glob = _sum_over_ranks_(local.clone())
        m_dy = (glob[:C] / ctx.count).to(dy.dtype).view(shape)
        m_dyx = (glob[C:] / ctx.count).to(dy.dtype).view(shape)
        scale = invstd if weight is None else invstd * weight
        dx = (dy - m_dy - xhat * m_dyx) * scale.view(shape)
        return dx, g_weight, g_bias, None


class SyncBatchNormAllReduce(torch.nn.modules.batchnorm._BatchNorm):
    """Drop-in for BatchNorm1d/2d whose batch statistics span all ranks (see the note above)."""

    def _check_input_dim(self, x):
        if x.dim() < 2:
            raise ValueError(f"expected at least 2D input (got {x.dim()}D)")

    def forward(self, x):
        if not (self.training or not self.track_running_stats):
            return torch.nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight,
                                                  self.bias, False, 0.0, self.eps)
        y, mean, var, count = _SyncBNFn.apply(x, self.weight, self.bias, self.eps)
        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (count / (count - 1).clamp_min(1.0))
                self.running_mean.mul_(1 - mom).add_(mean.to(self.running_mean.dtype), alpha=mom)
                self.running_var.mul_(1 - mom).add_(unbiased.to(self.running_var.dtype), alpha=mom)
        return y


def convert_sync_batchnorm(module):
    """Replace every BatchNorm1d/2d/3d under `module` by SyncBatchNormAllReduce (parameters and
    buffers are shared, names unchanged), as torch.nn.SyncBatchNorm.convert_sync_batchnorm does."""
    out = module
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(module, SyncBatchNormAllReduce):
        out = SyncBatchNormAllReduce(module.num_features, module.eps, module.momentum, module.affine,
                                     module.track_running_stats)
        if module.affine:
            out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var = module.running_mean, module.running_var
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        out.add_module(name, convert_sync_batchnorm(child))
    return out
