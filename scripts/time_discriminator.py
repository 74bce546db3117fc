"""Device time of the discriminator alone (B=32): forward, and forward+backward w.r.t. weights and input."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adaptpoint_amd.discriminator import PointDiscriminator1

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
D = PointDiscriminator1(num_classes=15).to(dev)
x = torch.randn(32, N, 3, device=dev, requires_grad=True)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def fwd():
    with torch.no_grad():
        D(x)


def fwdbwd():
    D.zero_grad(set_to_none=True)
    x.grad = None
    D(x).sum().backward()


print(f"N={N}: forward {timed(fwd):.3f} ms, forward+backward {timed(fwdbwd):.3f} ms (eager)")
