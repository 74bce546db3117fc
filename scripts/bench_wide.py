"""Per-kernel timing of the width-generic fused path (csrc/sa_wide.hip) at the four PointNeXt-S
stage shapes, B=32, against the register-resident stage-1 kernels; HIP events, eager.

    python scripts/bench_wide.py [--batch 64] [--residual] [--frozen] [--clustered]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_fused_wide as T
from adaptpoint_amd import fused, fused_wide

dev = torch.device("cuda:0")


def time_calls(run, names, iters=20):
    """Wrap fused._call so that every extension launch carries an event pair; returns mean us per name."""
    acc = {}
    orig = fused._call

    def timed(name, d, *a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(name, d, *a, **k)
        e.record()
        acc.setdefault(name.replace("apn_sa_", ""), []).append((s, e))
        return r
    for _ in range(3):
        run()
    fused._call = fused_wide._call = timed
    try:
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
    finally:
        fused._call = fused_wide._call = orig
    return {k: round(1e3 * sum(s.elapsed_time(e) for s, e in v) / iters, 1) for k, v in acc.items()}


def total_us(run, iters=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run()
    e.record()
    torch.cuda.synchronize()
    return round(1e3 * s.elapsed_time(e) / iters, 1)


BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 32
RESIDUAL = "--residual" in sys.argv          # the whole block: residual branch + ReLU (fused_wide.block), as the classifier runs it
FROZEN = "--frozen" in sys.argv              # no weight takes a gradient, the points do (the GAN's feedback pass)
CLUSTERED = "--clustered" in sys.argv
for (cin, N, M, radius) in T.STAGES:
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = T._setup(dev, cin, N, M, radius, B=BATCH, seed=5, clustered=CLUSTERED)
    f.requires_grad_(True)
    params = [conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias]
    skip = nbr = None
    if RESIDUAL:
        from adaptpoint_amd.layers import furthest_point_sample
        skip = torch.nn.Conv1d(cin, conv2.weight.shape[0], 1).to(dev)
        nbr = fused_wide.neighbour_index(idx, new_p, N, fidx=furthest_point_sample(p, M))
        params += [skip.weight, skip.bias]
    if FROZEN:
        for q in params:
            q.requires_grad_(False)
        p.requires_grad_(True)
    for name, fn in (("wide", fused_wide.grouped_mlp_max), ("register-resident", fused.grouped_mlp_max)):
        if name != "wide" and (cin != 32 or RESIDUAL):
            continue

        def run():
            f.grad = p.grad = None
            for q in params:
                q.grad = None
            if RESIDUAL:
                out = fused_wide.block(p, new_p, f, nbr, radius, conv1, bn1, conv2, bn2, skip_conv=skip, relu=True)
            else:
                out = fn(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)
            out.sum().backward()
        print(json.dumps({"stage_C_in": cin, "N": N, "M": M, "batch": BATCH, "kernels": name, "eager_fwd_bwd_us": total_us(run),
                          "per_launch_us": time_calls(run, None)}), flush=True)
