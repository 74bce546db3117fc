"""Do the GEMM library's calls stay correct when two branches of ONE replayed hipGraph run them concurrently?
(PyTorch hands rocBLAS / hipBLASLt a workspace per (handle, stream) when it launches eagerly; what a captured graph's
branches share is decided at capture.)  Chains of library products of the shapes the joint step's two lanes hold,
captured on two forked streams, replayed, compared with the same products run eagerly one after the other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(64, 512, 512), (64, 512, 256), (64, 256, 15), (128, 1024, 3072), (32768, 64, 192), (32, 1024, 512),
          (2048, 256, 512), (8192, 131, 128), (4096, 259, 256)]
A = {s: torch.randn(s[0], s[1], device=dev) for s in SHAPES}
W = {s: torch.randn(s[2], s[1], device=dev) / s[1] ** 0.5 for s in SHAPES}
bias = {s: torch.randn(s[2], device=dev) for s in SHAPES}


def chain(which, rounds):
    out = []
    for r in range(rounds):
        for i, s in enumerate(SHAPES):
            if i % 2 == which:
                out.append(torch.nn.functional.linear(A[s], W[s], bias[s] if r % 2 else None))
                out.append(A[s].t() @ out[-1])            # a weight-gradient shaped product
    return out


ref0, ref1 = chain(0, 4), chain(1, 4)
torch.cuda.synchronize()
s1 = torch.cuda.Stream()
warm = torch.cuda.Stream()
warm.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(warm):
    chain(0, 1); chain(1, 1)
    s1.wait_stream(warm)
    with torch.cuda.stream(s1):
        chain(1, 1)
    warm.wait_stream(s1)
torch.cuda.current_stream().wait_stream(warm)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    with torch.cuda.stream(s1):
        o1 = chain(1, 4)
    o0 = chain(0, 4)
    main.wait_stream(s1)
bad = 0
for it in range(20):
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(o0 + o1, ref0 + ref1):
        if not torch.equal(a, b):
            err = float((a - b).abs().max() / b.abs().max())
            if err > 1e-5:
                bad += 1
                if bad <= 5:
                    print("replay", it, "shape", tuple(a.shape), "relative deviation", err)
print("products that differ from the eager results over 20 replays:", bad)
