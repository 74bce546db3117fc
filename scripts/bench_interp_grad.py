"""three_interpolate_grad at the imitator's four decoder shapes and the PointNeXt segmentation ones: us per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pointnet2_batch_cuda as ext
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (b, c, n, m) in ((32, 1024, 128, 64), (32, 512, 256, 128), (32, 256, 512, 256), (32, 128, 1024, 512), (32, 64, 1024, 512),
                     (8, 64, 4096, 1024), (2, 64, 15000, 3750)):
    u = torch.rand(b, n, 3, device=dev)
    kn = u[:, ::n // m][:, :m].contiguous()
    d2 = torch.empty(b, n, 3, device=dev)
    idx = torch.empty(b, n, 3, dtype=torch.int32, device=dev)
    ext.three_nn_wrapper(b, n, m, u, kn, d2, idx)
    w = 1.0 / (d2.sqrt() + 1e-8)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    g = torch.randn(b, c, n, device=dev)
    gp = torch.zeros(b, c, m, device=dev)
    for _ in range(3):
        ext.three_interpolate_grad_wrapper(b, c, n, m, g, idx, w, gp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ext.three_interpolate_grad_wrapper(b, c, n, m, g, idx, w, gp)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    mb = (g.numel() + 2 * gp.numel()) * 4 / 1e6
    print(f"b={b} c={c} n={n} m={m}: {us:7.1f} us  ({mb:.1f} MB moved -> {mb / us * 1e-3 * 1e3:.0f} GB/s)")
