"""Run by tests/test_gpu_syncbn_two_ranks.py, one fresh process per role, all on cuda:0:

    syncbn_two_ranks_helper.py rank <r> <world> <port> <out_dir>    one rank of a gloo process group (device tensors are
                                                                    staged through the host: RCCL refuses two ranks on one
                                                                    device), 16 of the 32 clouds, EAGER
    syncbn_two_ranks_helper.py whole <out_dir>                      no torch.distributed: all 32 clouds in one process

Both build the PointNeXt-S classifier with every stage on the fused kernels through `adaptpoint_amd.workloads.build`
-- under data parallelism that makes the four fused blocks exchange their own BatchNorm sums in four all-reduces each
(`SetAbstraction.sync_bn`: csrc/sa_fused.hip's phases with the reduced sums between them) and converts the other
BatchNorm modules -- run ONE `train_one_epoch` iteration (train_autoaug.py:471-512; the reference forces SyncBatchNorm +
DistributedDataParallel there: examples/classification/main.py:27, train_autoaug.py:275-282) and save logits, loss, the
gradients the optimizer consumed (after the exchange), the BatchNorm buffers and the updated parameters."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PER_RANK, NPTS, SEED = 16, 1024, 0


def shard(rank):
    from adaptpoint_amd import dp, workloads
    pts = workloads._clouds(PER_RANK, NPTS, dp.shard_seed(SEED, rank))
    label = (torch.arange(PER_RANK) + 3 * rank) % 15
    return pts, label


def run(job, pts, label, dev):
    from adaptpoint_amd import workloads
    job.points, job.label = pts.to(dev), label.to(dev)
    with workloads.count_collectives() as c:
        logits, loss = job.cls_step(job.points, job.label, job.choice)
    torch.cuda.synchronize()
    C = job.nets["classifier"]
    (name, grads), = job.taps
    return {"logits": logits.cpu(), "loss": float(loss), "grads": [g.cpu() for g in grads],
            "buffers": {k: v.detach().cpu() for k, v in C.named_buffers()},
            "params": [q.detach().cpu() for q in C.parameters()], "collectives": dict(c.calls), "syncbn": job.syncbn}


def main():
    from adaptpoint_amd import _lib, dp, workloads
    from adaptpoint_amd import set_abstraction as SA
    _lib.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    role = sys.argv[1]
    if role == "rank":
        rank, world, port, out_dir = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
        os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        dp.init("gloo")
        assert torch.distributed.get_world_size() == world
        job = workloads.build("classifier", dev, batch=PER_RANK, npoints=NPTS, fused=True, distributed=True, seed=SEED,
                              name_seeded=True, dropout=False, record_grads=True)
        res = run(job, *shard(rank), dev)
        res["fused_fallbacks"] = dict(SA.FUSED_FALLBACKS)
        torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    else:
        out_dir = sys.argv[2]
        job = workloads.build("classifier", dev, batch=2 * PER_RANK, npoints=NPTS, fused=True, distributed=False, seed=SEED,
                              name_seeded=True, dropout=False, record_grads=True)
        (p0, l0), (p1, l1) = shard(0), shard(1)
        res = run(job, torch.cat([p0, p1]), torch.cat([l0, l1]), dev)
        res["fused_fallbacks"] = dict(SA.FUSED_FALLBACKS)
        torch.save(res, os.path.join(out_dir, "whole.pt"))
    print("done", role, flush=True)


if __name__ == "__main__":
    main()
