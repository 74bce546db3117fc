"""the PointsetGrouper grouping stage at the imitator's four stage shapes
(generator_component4_15.py:589-612: embed 64, x2 per stage, reduce 2, K=24), B=32, one MI355X.
Times forward and forward+backward of the fused op against the reference's composition
(index_points gather -> subtract anchor -> affine -> max) run with PyTorch on the same GPU and
the same indices.  Prints one JSON line per stage with algorithmic bytes and GB/s.

    python scripts/bench_pointset.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import golden_inputs as GI
from adaptpoint_amd.layers import ball_query, furthest_point_sample
from adaptpoint_amd.pointset import group_max


def time_us(fn, iters=30, warm=5):
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


def composed(points, idx, fidx, alpha, beta):
    B, N, C = points.shape
    bi = torch.arange(B, device=points.device).view(B, 1, 1)
    grouped = points[bi, idx.long(), :]
    mean = torch.gather(points, 1, fidx.long().unsqueeze(-1).expand(-1, -1, C)).unsqueeze(-2)
    grouped = alpha.view(1, 1, 1, -1) * (grouped - mean) + beta.view(1, 1, 1, -1)
    return grouped.max(dim=2)[0].permute(0, 2, 1)


def main():
    dev = torch.device("cuda:0")
    B, K = 32, 24
    xyz = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=0)).to(dev)
    for stage, (C, radius) in enumerate([(128, 0.1), (256, 0.2), (512, 0.4), (1024, 0.8)]):
        N = xyz.shape[1]
        M = N // 2
        fidx = furthest_point_sample(xyz, M)
        new_xyz = torch.gather(xyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3))
        idx = ball_query(radius, K, xyz, new_xyz)
        pts = torch.randn(B, N, C, device=dev, requires_grad=True)
        alpha = torch.randn(C, device=dev, requires_grad=True)
        beta = torch.randn(C, device=dev, requires_grad=True)
        w = torch.randn(B, C, M, device=dev)

        def run(fn, backward):
            def go():
                out = fn(pts, idx, fidx, alpha, beta)
                if backward:
                    pts.grad = alpha.grad = beta.grad = None
                    out.backward(w)
            return go
        res = {"stage": stage + 1, "B": B, "N": N, "np": M, "K": K, "C": C}
        with torch.no_grad():
            res["fused_fwd_us"] = round(time_us(run(group_max, False)), 1)
            res["torch_fwd_us"] = round(time_us(run(composed, False)), 1)
        res["fused_fwd_bwd_us"] = round(time_us(run(group_max, True)), 1)
        res["torch_fwd_bwd_us"] = round(time_us(run(composed, True)), 1)
        # compulsory bytes of the forward: table + indices in, output + selection out
        alg = B * (N * C * 4 + M * K * 4 + M * 4 + C * M * 4 + M * C)
        res["fwd_algorithmic_MB"] = round(alg / 1e6, 1)
        res["fwd_GBps"] = round(alg / res["fused_fwd_us"] / 1e3, 1)
        res["materialised_by_reference_MB"] = round(4 * B * M * K * C * 4 / 1e6, 1)
        print(json.dumps(res), flush=True)
        xyz = new_xyz.detach()


if __name__ == "__main__":
    main()
