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
    if hi - lo > 2 * payload + 4096:
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
            dist.all_reduce(span)
            if world > 1:
                span.div_(world)
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat)
    if world > 1:
        flat.div_(world)
    off = 0
    views = []
    for t in tensors:
        views.append(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    torch._foreach_copy_(list(tensors), views)
