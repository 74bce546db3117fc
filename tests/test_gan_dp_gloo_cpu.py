"""CPU suite, world_size 2 over gloo: one AdaptPoint joint step (`train_gan`, BASELINE configs[4] = configs[3]
sharded by cloud) under data parallelism.  The reference wraps the generator and the discriminator in
DistributedDataParallel WITHOUT converting their BatchNorm (train_autoaug.py:98-102; only the classifier gets
SyncBatchNorm, :276): every rank runs the step on its own clouds and the gradients are averaged before each Adam
step.  Properties checked: (1) both ranks leave the step with bit-identical generator and discriminator
parameters; (2) those parameters are what ONE process obtains by computing the two shards' gradients separately,
averaging them and stepping once (`GanStep.grad_sync` is the only difference between the two runs)."""
import copy
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NPTS, PER_RANK = 512, 2      # 512: the imitator head takes 24 neighbours among N / 16 points


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    """networks with name-seeded weights, the batch, and one set of random draws per shard"""
    import golden_inputs as GI
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    G = fill_parameters_by_name(AdaptPointAugmentor(fused=False))
    D = fill_parameters_by_name(PointDiscriminator1(num_classes=15))
    D.drop1.p = D.drop2.p = 0.0
    C = fill_parameters_by_name(PointNextSClassifier())
    pos = torch.from_numpy(GI.unit_sphere_cloud(2 * PER_RANK, NPTS, seed=31))
    points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1)
    label = torch.tensor([2, 9, 5, 14])
    noises = []
    for s in range(2):
        torch.manual_seed(100 + s)
        noises.append(draw_noise(PER_RANK, NPTS, 4, with_gumbel=True))
    return G, D, C, points, label, noises


def _run(G, D, C, points, label, noise, grad_sync):
    from oracle import cpu_block as CB
    from adaptpoint_amd import attention as A
    from adaptpoint_amd.gan import GanStep
    from adaptpoint_amd.pointnext import SmoothCrossEntropy
    core = A.attention
    A.attention = A._reference              # the attention core refuses CPU tensors (conftest.cpu_mirrors)
    try:
        with CB.CpuOps():
            GanStep(G, D, C, SmoothCrossEntropy(0.3), grad_sync=grad_sync)(points, label, noise=noise)
    finally:
        A.attention = core
    state = lambda net: ({k: v.detach().clone() for k, v in net.named_parameters()},
                         {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
    return state(G), state(D)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from adaptpoint_amd import dp
    dp.init("gloo")
    G, D, C, points, label, noises = _setup()
    sl = slice(PER_RANK * rank, PER_RANK * (rank + 1))
    g, d = _run(G, D, C, points[sl].contiguous(), label[sl], noises[rank], dp.allreduce_mean_)
    torch.save({"G": g, "D": d}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_world2_joint_step_equals_averaged_shard_gradients(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in (0, 1))
    for net in ("G", "D"):
        for part in (0, 1):                                             # parameters, then the averaged gradients
            for k in r0[net][part]:
                assert torch.equal(r0[net][part][k], r1[net][part][k]), (net, k)   # (1) the ranks stay in step

    # (2) one process: record each shard's gradients at the two synchronisation points, then step with their mean
    G, D, C, points, label, noises = _setup()
    recorded = []
    for s in range(2):
        calls = []
        sl = slice(PER_RANK * s, PER_RANK * (s + 1))
        _run(copy.deepcopy(G), copy.deepcopy(D), C, points[sl].contiguous(), label[sl], copy.deepcopy(noises[s]),
             lambda grads, calls=calls: calls.append([g.clone() for g in grads]))
        recorded.append(calls)                                           # [generator grads, discriminator grads]
    turn = iter(range(2))

    def inject(grads):
        i = next(turn)
        for g, a, b in zip(grads, recorded[0][i], recorded[1][i]):
            g.copy_((a + b) / 2)
    g, d = _run(copy.deepcopy(G), copy.deepcopy(D), C, points[:PER_RANK].contiguous(), label[:PER_RANK],
                copy.deepcopy(noises[0]), inject)
    for got, want in ((r0["G"], g), (r0["D"], d)):
        for k in want[1]:                                               # the gradients each Adam step consumed
            scale = float(want[1][k].abs().max()) + 1e-12
            assert float((got[1][k] - want[1][k]).abs().max()) <= 1e-4 * scale + 1e-7, k   # (2 threads vs all: summation orders; some gradients are exact zeros + noise)
        # Adam's first step moves a weight by lr * sign(gradient): equal wherever the gradient is not rounding noise
        same = sum(int(((got[0][k] - want[0][k]).abs() < 2e-6).sum()) for k in want[0])
        assert same >= 0.999 * sum(v.numel() for v in want[0].values())
