"""The imitator's predictor network SAComponent (generator_component4_15.py:588-712; the network
half of BASELINE configs[3]) forward + backward, B=32, one MI355X: the mirror over the fused
operators against the same mirror grouping / attending the way the reference composes them in
PyTorch (fused=False: materialised (B,np,K,C) and (B,H,N,N) tensors).  Same weights and inputs.

    python scripts/bench_imitator.py [--points 1024|2048]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import golden_inputs as GI
from adaptpoint_amd.imitator import SAComponent
from adaptpoint_amd.layers import furthest_point_sample
from adaptpoint_amd.pointnext import fill_parameters_by_name


def time_us(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--graph", action="store_true", help="replay forward+backward from a hipGraph")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    xyz = torch.from_numpy(GI.unit_sphere_cloud(a.batch, a.points, seed=0)).to(dev)
    anchor = furthest_point_sample(xyz, 4).long()                  # AdaptPoint_Augmentor.forward, :150
    res = {"B": a.batch, "N": a.points, "params": 5998062, "launch": "hipGraph replay" if a.graph else "eager"}
    outs = {}
    for name, fused in (("fused", True), ("composed", False)):
        model = fill_parameters_by_name(SAComponent(fused=fused)).to(dev).train()

        def step():
            for q in model.parameters():
                q.grad = None
            prob, logits = model(xyz, anchor, return_logits=True)
            (prob.sum() + logits.sum()).backward()
        if a.graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            for q in model.parameters():
                q.grad = None
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                prob, logits = model(xyz, anchor, return_logits=True)
                (prob.sum() + logits.sum()).backward()
            step = graph.replay
        torch.cuda.reset_peak_memory_stats()
        res[name + "_fwd_bwd_ms"] = round(time_us(step) / 1e3, 3)
        res[name + "_peak_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
        with torch.no_grad():
            outs[name] = model(xyz, anchor, return_logits=True)
    res["max_abs_diff_prob"] = float((outs["fused"][0] - outs["composed"][0]).abs().max())
    res["max_abs_diff_logits"] = float((outs["fused"][1] - outs["composed"][1]).abs().max())
    res["speedup"] = round(res["composed_fwd_bwd_ms"] / res["fused_fwd_bwd_ms"], 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
