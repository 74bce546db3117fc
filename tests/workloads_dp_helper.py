"""Run by tests/test_gpu_workloads_dp.py in a process of its own (a world-size-1 RCCL process group): the training-step
workloads built for data parallelism (SyncBatchNorm exchanges + flat gradient all-reduces issued through RCCL) against
the same workloads built without torch.distributed, from identical weights, clouds and draws -- the gradients each
optimizer step consumed, network by network.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from adaptpoint_amd import _lib, dp, fused, workloads
    _lib.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dp.init("nccl", dev, force=True)
    out = {}
    for workload in sys.argv[1:] or ["classifier", "gan"]:
        taps = {}
        for mode in ("single", "distributed"):
            fused.FORCE_PHASED = dp.FORCE_COLLECTIVES = mode == "distributed"     # issue every exchange at world size 1 too
            torch.manual_seed(7)
            torch.cuda.manual_seed(7)
            job = workloads.build(workload, dev, batch=8, npoints=1024, fused=True, distributed=(mode == "distributed"),
                                  name_seeded=True, dropout=False, record_grads=True)
            torch.cuda.manual_seed(11)                                            # the generator's device draws
            with workloads.count_collectives() as c:
                job.step()
            torch.cuda.synchronize()
            taps[mode] = job.taps
            if mode == "distributed":
                out[workload] = {"collectives": dict(c.calls), "syncbn": job.syncbn}
        dev_rel = {}
        for (n, ga), (m, gb) in zip(taps["single"], taps["distributed"]):
            assert n == m and len(ga) == len(gb)
            num = sum(float(((x - y).double() ** 2).sum()) for x, y in zip(ga, gb))
            den = sum(float((x.double() ** 2).sum()) for x in ga)
            dev_rel[n] = (num / max(den, 1e-300)) ** 0.5
            out[workload].setdefault("tensors", {})[n] = len(ga)
        out[workload]["relative_l2_of_gradient_difference"] = dev_rel
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
