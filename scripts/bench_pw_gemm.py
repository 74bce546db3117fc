"""The three contractions of one per-point layer (csrc/pointwise.hip) timed one by one through the C ABI: forward
y = W x, input gradient gx = W^T gy, weight gradient gW = sum gy x^T (with its fold), ten launches per hipGraph.

    python scripts/bench_pw_gemm.py [--layer decode1] [--planes 3]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adaptpoint_amd import _lib
from adaptpoint_amd.fused import _call

LAYERS = {"embedding": (3, 64, 1024), "extract1": (64, 128, 1024), "extract2": (128, 256, 512), "extract3": (256, 512, 256),
          "extract4": (512, 1024, 128), "decode1": (1536, 512, 256), "decode2": (768, 256, 512), "decode3": (384, 128, 1024),
          "decode4": (192, 64, 1024)}


def timed(fn, iters=20):
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
    return e0.elapsed_time(e1) / (iters * 10) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="decode1")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--planes", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    out = {}
    for name in a.layer.split(","):
        C, O, N = LAYERS[name]
        B = a.batch
        x = torch.randn(B, C, N, device=dev); w = torch.randn(O, C, device=dev) * 0.05
        gy = torch.randn(B, O, N, device=dev)
        y = torch.empty(B, O, N, device=dev); gx = torch.empty_like(x); gw = torch.empty(O, C, device=dev)
        part = torch.empty(lib.apn_pw_conv_tiles(B, N), 2, O, device=dev)
        scratch = torch.empty(lib.apn_pw_conv_grad_weight_splits(B, C, O, N), O, C, device=dev)
        fwd = lambda: _call("apn_pw_conv_forward", dev, B, C, O, N, a.planes, x.data_ptr(), w.data_ptr(), y.data_ptr(), part.data_ptr())
        dgr = lambda: _call("apn_pw_conv_grad_input", dev, B, C, O, N, a.planes, gy.data_ptr(), w.data_ptr(), gx.data_ptr())
        wgr = lambda: _call("apn_pw_conv_grad_weight", dev, B, C, O, N, a.planes, gy.data_ptr(), x.data_ptr(), scratch.data_ptr(), gw.data_ptr())
        row = {"layer": name, "forward_us": round(timed(fwd), 1), "input_grad_us": round(timed(dgr), 1), "weight_grad_us": round(timed(wgr), 1)}
        if os.environ.get("APN_LIB_PATH") is None:
            ref = torch.einsum("oc,bcn->bon", w.double(), x.double())
            row["forward_err"] = float((y.double() - ref).abs().max() / ref.abs().max())
            ref = torch.einsum("oc,bon->bcn", w.double(), gy.double())
            row["input_grad_err"] = float((gx.double() - ref).abs().max() / ref.abs().max())
            ref = torch.einsum("bon,bcn->oc", gy.double(), x.double())
            row["weight_grad_err"] = float((gw.double() - ref).abs().max() / ref.abs().max())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
