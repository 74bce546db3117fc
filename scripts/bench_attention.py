"""the imitator's Anchor_selfattention core, softmax(q k^T / 4) v with 4 heads of
16 dims over all M points of a cloud (generator_component4_15.py:460-474), B=32, one MI355X.
Fused kernels (csrc/attention.hip) against the reference's composition in PyTorch on the same GPU
(which materialises the (B,H,M,M) scores) and against PyTorch's own fused SDPA.

    python scripts/bench_attention.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from adaptpoint_amd import attention as A


def time_us(fn, iters=20, warm=3):
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


def sdpa(q, k, v, heads):
    B, M, C = q.shape
    qh, kh, vh = (t.reshape(B, M, heads, C // heads).permute(0, 2, 1, 3) for t in (q, k, v))
    return F.scaled_dot_product_attention(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, M, C)


def main():
    dev = torch.device("cuda:0")
    B, H = 32, 4
    for M in (1024, 2048):
        q, k, v = (torch.randn(B, M, H * 16, device=dev, requires_grad=True) for _ in range(3))
        w = torch.randn(B, M, H * 16, device=dev)
        res = {"B": B, "M": M, "heads": H, "head_dim": 16}
        for name, fn in (("fused", A.attention), ("torch_composed", A._reference), ("torch_sdpa_f32", sdpa)):
            try:
                with torch.no_grad():
                    res[name + "_fwd_us"] = round(time_us(lambda: fn(q, k, v, H)), 1)

                def fb():
                    q.grad = k.grad = v.grad = None
                    fn(q, k, v, H).backward(w)
                res[name + "_fwd_bwd_us"] = round(time_us(fb), 1)
            except Exception as exc:                 # e.g. no SDPA kernel for this dtype/shape
                res[name + "_error"] = type(exc).__name__
        flops = 4 * B * H * M * M * 16
        res["fwd_algorithmic_GFLOP"] = round(flops / 1e9, 2)
        res["fused_fwd_TFLOPs"] = round(flops / res["fused_fwd_us"] / 1e6, 1)
        res["scores_materialised_by_reference_MB"] = round(2 * B * H * M * M * 4 / 1e6)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
