import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_fused_wide as T
from adaptpoint_amd.fused_wide import grouped_mlp_max
from adaptpoint_amd import fused_wide as FW
dev = torch.device("cuda:0")
for (cin, N, M, radius) in T.STAGES:
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = T._setup(dev, cin, N, M, radius, B=32, seed=5)
    f.requires_grad_(True)
    params = [conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias]
    def fb():
        f.grad = None
        for q in params: q.grad = None
        out = grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)
        out.square().sum().backward()
        return out
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fb()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    MODE = os.environ.get("MODE", "a")
    g = torch.cuda.CUDAGraph()
    DBG = {}
    FW._DEBUG = DBG
    with torch.cuda.graph(g, stream=side if MODE == "b" else None):
        out = fb()
    res = []
    FW._DEBUG = None
    snaps = []
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
        snaps.append({k: v.detach().clone() for k, v in DBG.items() if torch.is_tensor(v)})
        if i > 0:
            d = {k: float((snaps[i][k].double() - snaps[0][k].double()).abs().max()) for k in snaps[0]}
            print("C=%d DIFF replay %d vs 0:" % (cin, i), {k: "%.1e" % v for k, v in d.items() if v > 0})
        res.append([out.detach().clone(), f.grad.clone()] + [q.grad.clone() for q in params])
    ref = [fb().detach().clone(), f.grad.clone()] + [q.grad.clone() for q in params]
    torch.cuda.synchronize()
    for i in (0, 1, 2):
        errs = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(res[i], ref)]
        fin = [bool(torch.isfinite(a).all()) for a in res[i]]
        print("C=%d replay %d vs eager:" % (cin, i), " ".join("%.1e" % e for e in errs), "finite:", fin, flush=True)
