"""Experiment: what on the index stream slows the MLP chain?  One chain of 20 captured MLP steps (index stages
precomputed) replayed alone, then beside a second stream that replays, per chain replay, (a) the stacked FPS, (b) the
stacked ball query, (c) both, (d) FPS with fewer waves per cloud."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as BN
from adaptpoint_amd import _lib
from adaptpoint_amd.fused import _call, Sampling

dev = torch.device("cuda:0")
torch.manual_seed(0)
from adaptpoint_amd import fused as _f
_f.PRECISION = "bf16x3"
SPG = 20
blk = BN.make_block(fused=True).to(dev).train()
pf = [BN.make_inputs(BN.B_PER_GPU, seed=i) for i in range(SPG)]
p_all = torch.cat([a for a, _ in pf]).to(dev)
ps = [p_all[i * BN.B_PER_GPU:(i + 1) * BN.B_PER_GPU] for i in range(SPG)]
fs = [b.to(dev).requires_grad_(True) for _, b in pf]
smp = []
for p in ps:
    s = blk.sample(p)
    blk.index_for(s, BN.N_PTS, BN.C_IN)
    smp.append(s)
ones = torch.ones(1, 1, 1, device=dev)


def steps():
    for i in range(SPG):
        for f in fs:
            f.grad = None
        for q in blk.parameters():
            q.grad = None
        _, out = blk([ps[i], fs[i]], sampling=smp[i])
        torch.autograd.backward([out], [ones.expand_as(out)])


main, side = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(main):
    for _ in range(2):
        steps()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main):
    steps()

SB, N, M, K = SPG * BN.B_PER_GPU, BN.N_PTS, BN.NPOINT, BN.NSAMPLE
temp = torch.empty(SB, N, device=dev)
fidx = torch.empty(SB, M, dtype=torch.int32, device=dev)
newp = torch.empty(SB, M, 3, device=dev)
idx = torch.zeros(SB, M, K, dtype=torch.int32, device=dev)


def fps(waves):
    temp.fill_(1e10)
    _call("apn_furthest_point_sampling_tuned", dev, SB, N, M, p_all.data_ptr(), temp.data_ptr(), fidx.data_ptr(), waves, 0)


def ballq():
    _call("apn_ball_query", dev, SB, N, M, 0.15, K, newp.data_ptr(), p_all.data_ptr(), idx.data_ptr())


newp.copy_(torch.gather(p_all, 1, torch.zeros(SB, M, 1, dtype=torch.long, device=dev).expand(-1, -1, 3) + torch.arange(M, device=dev).view(1, M, 1)))
loads = {"nothing": None, "FPS (8 waves per cloud)": lambda: fps(0), "FPS, 4 waves per cloud": lambda: fps(4),
         "FPS, 2 waves per cloud": lambda: fps(2), "ball query": ballq, "FPS + ball query": lambda: (fps(0), ballq())}
for name, fn in loads.items():
    sg = None
    if fn is not None:
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        sg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(sg, stream=side):
            fn()
    def run(reps):
        for _ in range(reps):
            if sg is not None:
                with torch.cuda.stream(side):
                    sg.replay()
            with torch.cuda.stream(main):
                g.replay()
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 40
    side_us = 0.0
    if sg is not None:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            sg.replay()
        torch.cuda.synchronize()
        side_us = (time.perf_counter() - t1) / 20 * 1e6
    print(f"chain beside {name:26s}: {el * 1e3 / SPG:.4f} ms/step  (replay {el * 1e3:.3f} ms; that load alone {side_us:.0f} us)")
    if name == "FPS + ball query":
        # the same with bench.py's event structure: the side replay waits for the previous chain replay, the chain
        # replay for the previous side replay
        mlp_done, box = torch.cuda.Event(), [torch.cuda.Event()]
        mlp_done.record(main); box[0].record(side)
        def run2(reps):
            for _ in range(reps):
                side.wait_event(mlp_done)
                with torch.cuda.stream(side):
                    sg.replay()
                    nxt = torch.cuda.Event(); nxt.record(side)
                main.wait_event(box[0])
                with torch.cuda.stream(main):
                    g.replay()
                    mlp_done.record(main)
                box[0] = nxt
        run2(5); torch.cuda.synchronize()
        t0 = time.perf_counter(); run2(40); torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 40
        print(f"   ... with the two-event hand-over per replay: {el * 1e3 / SPG:.4f} ms/step  (replay {el * 1e3:.3f} ms)")
