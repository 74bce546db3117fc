"""The AdaptPoint imitator's encoder and masking attention (SAComponent,
openpoints/models_adaptpoint/generator_component4_15.py:588-712) reduced to the parts SURVEY 8(f)
rows 1-2 name: embedding -> 4 x (ConvBNReLU1D + PointsetGrouper) -> Anchor_selfattention on the
embedded points; forward + backward, B=32, one MI355X.  `fused` uses
adaptpoint_amd.pointset / adaptpoint_amd.attention; `composed` keeps the same FPS / ball-query
operators but groups and attends the way the reference composes them in PyTorch (materialised
(B,np,K,C) and (B,H,N,N) tensors).  Same weights, same inputs.

    python scripts/bench_imitator_core.py [--points 1024|2048]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn as nn

import golden_inputs as GI
from adaptpoint_amd.attention import AnchorSelfAttention
from adaptpoint_amd.pointset import PointsetGrouper


def conv_bn_relu(cin, cout):                     # ConvBNReLU1D, generator_component4_15.py:92-104
    return nn.Sequential(nn.Conv1d(cin, cout, 1, bias=False), nn.BatchNorm1d(cout), nn.ReLU(inplace=True))


class ImitatorCore(nn.Module):
    def __init__(self, fused, embed=64, radii=(0.1, 0.2, 0.4, 0.8)):
        super().__init__()
        self.embedding = conv_bn_relu(3, embed)
        self.extract = nn.ModuleList()
        self.groupers = nn.ModuleList()
        c = embed
        for r in radii:                          # :604-614
            self.extract.append(conv_bn_relu(c, 2 * c))
            self.groupers.append(PointsetGrouper(channel=2 * c, reduce=2, kneighbors=24, radi=r, fused=fused))
            c *= 2
        self.attention = AnchorSelfAttention(dim=embed, head_num=4, fused=fused)     # :644

    def forward(self, xyz):
        x0 = self.embedding(xyz.permute(0, 2, 1).contiguous())
        x, p = x0, xyz
        for ext, grp in zip(self.extract, self.groupers):                              # :677-684
            x = ext(x)
            p, x = grp(p, x.permute(0, 2, 1).contiguous())
        att = self.attention(x=x0.permute(0, 2, 1), xyz=xyz)                          # :706
        return x, att


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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    xyz = torch.from_numpy(GI.unit_sphere_cloud(a.batch, a.points, seed=0)).to(dev)
    torch.manual_seed(0)
    fused = ImitatorCore(True).to(dev).train()
    composed = ImitatorCore(False).to(dev).train()
    composed.load_state_dict(fused.state_dict())
    res = {"B": a.batch, "N": a.points}
    outs = {}
    for name, model in (("fused", fused), ("composed", composed)):
        def step():
            for q in model.parameters():
                q.grad = None
            x, att = model(xyz)
            (x.sum() + att.sum()).backward()
        torch.cuda.reset_peak_memory_stats()
        res[name + "_fwd_bwd_ms"] = round(time_us(step) / 1e3, 3)
        res[name + "_peak_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
        with torch.no_grad():
            outs[name] = model(xyz)
    res["max_abs_diff_features"] = float((outs["fused"][0] - outs["composed"][0]).abs().max())
    res["max_abs_diff_attention"] = float((outs["fused"][1] - outs["composed"][1]).abs().max())
    res["speedup"] = round(res["composed_fwd_bwd_ms"] / res["fused_fwd_bwd_ms"], 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
