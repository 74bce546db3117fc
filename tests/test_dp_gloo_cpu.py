"""CPU suite: the N>1 path of bench.py at world_size 2 over gloo -- clouds sharded by
rank, DistributedDataParallel gradient all-reduce, barrier + max-over-ranks timing.
The block runs on CPU over the oracle ops (tests may use the oracle; SyncBatchNorm
is GPU-only in PyTorch, so it stays off here)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _block():
    from adaptpoint_amd.set_abstraction import SetAbstraction
    return SetAbstraction(8, 16, layers=2, stride=2,
                          group_args={'NAME': 'ballquery', 'radius': 0.3, 'nsample': 8, 'normalize_dp': True},
                          norm_args={'norm': 'bn'}, act_args={'act': 'relu'},
                          conv_args={'order': 'conv-norm-act'}, use_res=True)


def _inputs(seed):
    import golden_inputs as GI
    p = torch.from_numpy(GI.unit_sphere_cloud(2, 128, seed=seed))
    f = torch.from_numpy(GI.seeded_normal((2, 8, 128), seed=seed + 7))
    return p, f


def _local_grads(seed):
    from oracle import cpu_block as CB
    torch.manual_seed(0)
    blk = CB.build_cpu_block(_block)
    blk.train()
    p, f = _inputs(seed)
    CB.run_step(blk, p, f)
    return [q.grad.clone() for q in blk.parameters()]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from adaptpoint_amd import dp
    from oracle import cpu_block as CB
    from oracle import oracle as O
    O.set_threads(1)
    w, r = dp.init("gloo")
    assert (w, r) == (world, rank)
    torch.manual_seed(0)
    blk = CB.build_cpu_block(_block)
    blk.train()
    model = torch.nn.parallel.DistributedDataParallel(blk)
    p, f = _inputs(dp.shard_seed(0, rank))
    f.requires_grad_(True)

    def step():
        for q in blk.parameters():
            q.grad = None
        with CB.CpuOps():
            _, out = model([p, f])
            out.sum().backward()
    elapsed = dp.timed_steps(step, steps=2, warmup=1)
    grads = [q.grad.clone() for q in blk.parameters()]
    # the bench's other exchange: local backward (no DDP) + one flat-bucket all-reduce
    torch.manual_seed(0)
    blk2 = CB.build_cpu_block(_block)
    blk2.train()
    CB.run_step(blk2, p, f.detach())
    flat_grads = [q.grad for q in blk2.parameters()]
    dp.allreduce_mean_(flat_grads)
    # gradients carved out of ONE buffer (what the fused block's backward produces): reduced in
    # place, padding between the views included
    buf = torch.full((300,), float(rank + 1))
    shared = [buf[0:100].view(10, 10), buf[128:178], buf[192:292].view(4, 25)]
    assert dp._common_span(shared) is not None and dp._common_span(flat_grads) is None
    dp.allreduce_mean_(shared)
    torch.save({"elapsed": elapsed, "grads": grads, "flat": [g.clone() for g in flat_grads],
                "shared": buf.clone()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_gloo_shards_and_allreduces(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from adaptpoint_amd import dp
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    # one clock for the job: the max over ranks, identical everywhere
    assert res[0]["elapsed"] == res[1]["elapsed"] > 0
    # DDP leaves the SAME averaged gradient on every rank ...
    for a, b in zip(res[0]["grads"], res[1]["grads"]):
        assert torch.equal(a, b)
    # ... equal to the mean of the per-shard gradients computed without any collective
    want = [(_a + _b) / 2 for _a, _b in zip(_local_grads(dp.shard_seed(0, 0)), _local_grads(dp.shard_seed(0, 1)))]
    for got, w in zip(res[0]["grads"], want):
        np.testing.assert_allclose(got.numpy(), w.numpy(), rtol=1e-4, atol=1e-5)  # f32 sums in a different order
    for r in range(world):     # mean of (1, 2) over the whole span 0..292, untouched beyond it
        assert torch.all(res[r]["shared"][:292] == 1.5) and torch.all(res[r]["shared"][292:] == r + 1)
    # the flat-bucket all-reduce leaves the same averaged gradients as DDP, on every rank
    for a, b, w in zip(res[0]["flat"], res[1]["flat"], want):
        assert torch.equal(a, b)
        np.testing.assert_allclose(a.numpy(), w.numpy(), rtol=1e-4, atol=1e-5)


def test_shards_are_distinct():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from adaptpoint_amd import dp
    a, _ = _inputs(dp.shard_seed(0, 0))
    b, _ = _inputs(dp.shard_seed(0, 1))
    assert not torch.equal(a, b)
    assert dp.host_threads() >= 1 and dp.host_threads(cap=1) == 1
