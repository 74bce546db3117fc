"""Per-layer timing of the imitator's per-point MLP layers (ConvBNReLU1D) at B = 32: csrc/pointwise.hip with two
and three bf16 planes against the reference's three modules on PyTorch (MIOpen / Tensile fp32), fwd + bwd, hipGraph
replay of 20 layers' worth per graph so that launch overheads are the GPU's, not Python's."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from adaptpoint_amd import pointwise

LAYERS = [("embedding", 3, 64, 1024), ("extract1", 64, 128, 1024), ("extract2", 128, 256, 512),
          ("extract3", 256, 512, 256), ("extract4", 512, 1024, 128), ("decode1", 1536, 512, 256),
          ("decode2", 768, 256, 512), ("decode3", 384, 128, 1024), ("decode4", 192, 64, 1024)]


def timed(fn, iters):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * 10) * 1e3       # us per fwd+bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--scale", type=int, default=1, help="multiply N (1024 -> 2048 points: 2)")
    ap.add_argument("--layers", default="", help="comma-separated subset")
    ap.add_argument("--only", default="", help="planes2 | planes3 | torch")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tot = {"planes2": 0.0, "planes3": 0.0, "torch": 0.0}
    for name, C, O, N in LAYERS:
        if a.layers and name not in a.layers.split(","):
            continue
        N *= a.scale
        conv = nn.Conv1d(C, O, 1, bias=False).to(dev); bn = nn.BatchNorm1d(O).to(dev)
        x = torch.randn(a.batch, C, N, device=dev, requires_grad=C != 3)
        gout = torch.randn(a.batch, O, N, device=dev)

        def fused():
            conv.weight.grad = bn.weight.grad = bn.bias.grad = None
            x.grad = None
            pointwise.conv_bn_act(x, conv, bn).backward(gout)

        def plain():
            conv.weight.grad = bn.weight.grad = bn.bias.grad = None
            x.grad = None
            torch.relu(bn(conv(x))).backward(gout)

        row = {"layer": name, "c_in": C, "c_out": O, "n": N, "gflop_fwd_bwd": round(6e-9 * a.batch * N * C * O, 2)}
        for planes in (2, 3):
            if a.only and a.only != f"planes{planes}":
                continue
            pointwise.PRECISION = planes
            row[f"planes{planes}_us"] = round(timed(fused, a.iters), 1)
            tot[f"planes{planes}"] += row[f"planes{planes}_us"]
        if not a.only or a.only == "torch":
            row["torch_us"] = round(timed(plain, a.iters), 1)
            tot["torch"] += row["torch_us"]
        print(json.dumps(row), flush=True)
    print(json.dumps({"layer": "all nine", **{k + "_us": round(v, 1) for k, v in tot.items()}}))


if __name__ == "__main__":
    main()
