"""Helper of tests/test_gpu_ddp_single.py (own process: it creates a world_size-1 RCCL group): the SyncBatchNorm step
of bench.py's N>1 path -- four statistics all-reduces inside the fused block, one flat gradient all-reduce -- run (a)
eagerly around the collectives and (b) with the collectives captured into a hipGraph (thread-local capture mode),
from the same weights on the same clouds.  Prints one JSON line with the largest relative difference of every gradient."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN  # noqa: E402
from adaptpoint_amd import dp, fused, graphs  # noqa: E402


def main():
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dp.init("nccl", dev, force=True)
    fused.FORCE_PHASED = True
    torch.manual_seed(0)
    blk = BN.make_block(fused=True, sync_bn=True).to(dev).train()
    p, f = BN.make_inputs(32, seed=3)
    p, f = p.to(dev), f.to(dev).requires_grad_(True)
    params = list(blk.parameters())
    wts = torch.randn(32, 64, 512, device=dev, generator=torch.Generator(dev).manual_seed(1))
    state = {k: v.clone() for k, v in blk.state_dict().items()}

    def step():
        for q in params:
            q.grad = None
        f.grad = None
        _, out = blk([p, f])
        torch.autograd.backward([out], [wts])
        dp.allreduce_mean_([q.grad for q in params])

    def grads():
        return [f.grad.clone()] + [q.grad.clone() for q in params]

    def restore():
        with torch.no_grad():
            own = blk.state_dict()
            for k, v in state.items():
                own[k].copy_(v)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    restore()
    step()
    torch.cuda.synchronize()
    eager = grads()
    restore()
    dist.barrier()
    torch.cuda.synchronize()
    g, _, census = graphs.capture(step, what="the SyncBatchNorm step with its collectives", capture_error_mode="thread_local")
    g.replay()
    torch.cuda.synchronize()
    captured = grads()
    worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(captured, eager))
    print(json.dumps({"max_rel_diff": worst, "graph_nodes": census}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
