"""Experiment: how much do two independent MLP chains (two replicas of the headline block, each on its own stream,
index stages precomputed) overlap on one GPU?  The single chain is a sequence of latency-bound launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import copy
import torch
import bench as BN

dev = torch.device("cuda:0")
torch.manual_seed(0)
from adaptpoint_amd import fused as _f
_f.PRECISION = "bf16x3"
SPG = 20
blocks = [BN.make_block(fused=True).to(dev).train()]
blocks.append(copy.deepcopy(blocks[0]))
chains = []
for c, blk in enumerate(blocks):
    pf = [BN.make_inputs(BN.B_PER_GPU, seed=100 * c + i) for i in range(SPG)]
    ps = [a.to(dev) for a, _ in pf]
    fs = [b.to(dev).requires_grad_(True) for _, b in pf]
    smp = []
    for p in ps:
        s = blk.sample(p)
        blk.index_for(s, BN.N_PTS, BN.C_IN)
        smp.append(s)
    chains.append((blk, ps, fs, smp))
ones = torch.ones(1, 1, 1, device=dev)


def steps(chain):
    blk, ps, fs, smp = chain
    for i in range(SPG):
        for f in fs:
            f.grad = None
        for q in blk.parameters():
            q.grad = None
        _, out = blk([ps[i], fs[i]], sampling=smp[i])
        torch.autograd.backward([out], [ones.expand_as(out)])


graphs, streams = [], [torch.cuda.Stream(), torch.cuda.Stream()]
for c, chain in enumerate(chains):
    with torch.cuda.stream(streams[c]):
        for _ in range(2):
            steps(chain)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[c]):
        steps(chain)
    graphs.append(g)
torch.cuda.synchronize()


def run(which, reps=50):
    for _ in range(5):
        for c in which:
            with torch.cuda.stream(streams[c]):
                graphs[c].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for c in which:
            with torch.cuda.stream(streams[c]):
                graphs[c].replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nsteps = reps * SPG * len(which)
    return el / nsteps * 1e3, BN.B_PER_GPU * nsteps / el


a = run([0])
b = run([0, 1])
print(f"one chain (index stages precomputed): {a[0]:.4f} ms/step, {a[1]:.0f} clouds/s")
print(f"two chains side by side:              {b[0]:.4f} ms/step, {b[1]:.0f} clouds/s  ({b[1] / a[1]:.2f}x)")
