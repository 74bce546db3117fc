"""CPU suite, world_size 2 over gloo: the launchable training-step workloads of BASELINE configs[2]-[4]
(`adaptpoint_amd.workloads.build`, the entry point of `bench.py --workload ...`) under data parallelism, set up as the
reference sets its networks up (examples/classification/main.py:27, train_autoaug.py:98-102, 275-282): SyncBatchNorm +
gradient averaging for the classifier, gradient averaging alone for generator and discriminator.  Property: after one
step on different shards every rank holds bit-identical weights and running statistics, the step did move them, and
the collectives it issued are the ones the bench line will report."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NPTS, PER_RANK = 512, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, workload):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import cpu_block as CB
    from adaptpoint_amd import attention as A, dp, workloads
    dp.init("gloo")
    job = workloads.build(workload, torch.device("cpu"), batch=PER_RANK, npoints=NPTS, fused=False, name_seeded=True,
                          dropout=False, record_grads=True)
    assert job.distributed and job.world == 2 and job.rank == rank
    before = {n: {k: v.detach().clone() for k, v in net.state_dict().items()} for n, net in job.nets.items()}
    core = A.attention
    A.attention = A._reference              # the attention core refuses CPU tensors (conftest.cpu_mirrors)
    try:
        with CB.CpuOps(), workloads.count_collectives() as c:
            torch.manual_seed(500 + rank)   # the generator's CPU draws differ per rank, as its clouds do
            job.step()
    finally:
        A.attention = core
    torch.save({"state": {n: {k: v.detach().clone() for k, v in net.state_dict().items()} for n, net in job.nets.items()},
                "before": before, "collectives": dict(c.calls), "syncbn": job.syncbn, "points": job.points,
                "taps": [(n, [g.clone() for g in gs]) for n, gs in job.taps]}, os.path.join(out_dir, f"{workload}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload", ["classifier", "adaptpoint"])
def test_world2_training_step_workloads_keep_the_ranks_in_step(tmp_path, workload):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), workload), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"{workload}{r}.pt")) for r in (0, 1))
    assert not torch.equal(a["points"], b["points"])                        # every rank drew its own shard
    trained = {"classifier": ["classifier"], "adaptpoint": ["classifier", "generator", "discriminator"]}[workload]
    for net in trained:
        moved = 0
        for k, v in a["state"][net].items():
            if net == "generator" and ("running_" in k or "num_batches" in k):
                continue                                                    # G / D keep per-rank BatchNorm statistics
            if net == "discriminator" and (k.endswith("_u") or k.endswith("_v") or "running_" in k or "num_batches" in k or "parametrizations" in k and "original" not in k):
                continue                                                    # power-iteration state follows the local batch
            assert torch.equal(v, b["state"][net][k]), (net, k)
            moved += int(v.dtype.is_floating_point and not torch.equal(v, a["before"][net][k]))
        assert moved > 10, (net, moved)
    # SyncBatchNorm on the classifier only: all 12 BatchNorm modules converted (no fused blocks on CPU)
    assert a["syncbn"] == {"fused_blocks_exchanging_their_own_sums": 0, "modules_converted": 12}
    # every synchronisation point left the same averaged gradients on both ranks
    assert [n for n, _ in a["taps"]] == ({"classifier": ["classifier"],
                                          "adaptpoint": ["generator", "discriminator", "classifier"]}[workload])
    for (n, ga), (_, gb) in zip(a["taps"], b["taps"]):
        for x, y in zip(ga, gb):
            assert torch.equal(x, y), n
    # 12 layers x (forward + backward) statistics exchanges + one flat gradient all-reduce per trained network
    want = 24 + 1 + (2 if workload == "adaptpoint" else 0)
    assert a["collectives"]["all_reduce"] == want == b["collectives"]["all_reduce"], a["collectives"]


def _fallback_worker(rank, world, port, out_dir):
    """A fused block whose BatchNorms `sync_batchnorm_` left alone (it exchanges its own sums) but whose fused kernels cannot
    run (CPU tensors here; on a GPU: an unsupported width): the composed path must refuse, not normalise rank-locally."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import golden_inputs as GI
    from oracle import cpu_block as CB
    from adaptpoint_amd import dp, workloads
    from adaptpoint_amd.set_abstraction import SetAbstraction
    dp.init("gloo")
    blk = SetAbstraction(32, 64, layers=2, stride=2, fused=True, use_res=True,
                         group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                         norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'}).train()
    model, n_fused, n_conv = workloads.sync_batchnorm_(torch.nn.Sequential(blk))
    p = torch.from_numpy(GI.unit_sphere_cloud(2, 256, seed=40 + rank))
    f = torch.from_numpy(GI.seeded_normal((2, 32, 256), seed=50 + rank))
    msg = ""
    try:
        with CB.CpuOps():
            blk([p, f])
    except RuntimeError as exc:
        msg = str(exc)
    torch.save({"fused": n_fused, "converted": n_conv, "sync_bn": blk.sync_bn, "error": msg}, os.path.join(out_dir, f"fb{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world2_fused_block_that_falls_back_under_syncbn_raises(tmp_path):
    """`sync_batchnorm_` leaves the BatchNorm modules of a fused set-abstraction block unconverted
    because the block exchanges its own statistics; if the block then cannot run its fused kernels it must not silently
    normalise with rank-local statistics (not the reference's SyncBatchNorm, train_autoaug.py:275-282)."""
    mp.spawn(_fallback_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        d = torch.load(os.path.join(tmp_path, f"fb{r}.pt"))
        assert d["fused"] == 1 and d["converted"] == 0 and d["sync_bn"] is True, d
        assert "cannot run its fused kernels" in d["error"] and "convert_sync_batchnorm" in d["error"], d["error"]
