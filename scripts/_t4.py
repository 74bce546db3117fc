import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import _lib
from adaptpoint_amd.fused import _call
from adaptpoint_amd.layers import ball_query, furthest_point_sample
dev = torch.device("cuda:0")
lib = _lib.load()
H, N, M, radius, B = 32, 1024, 512, 0.15, 32
O = 2 * H
p = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=1)).to(dev)
fidx = furthest_point_sample(p, M).long()
new_p = torch.gather(p, 1, fidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = ball_query(radius, 32, p, new_p)
g = torch.Generator(dev).manual_seed(0)
U = torch.randn(B, N, H, device=dev, generator=g); V = 0.3 * torch.randn(B, M, H, device=dev, generator=g)
pack1 = torch.cat([0.5 + torch.rand(H, device=dev, generator=g), 0.2 * torch.randn(H, device=dev, generator=g),
                   0.1 * torch.randn(H, device=dev, generator=g), 0.5 + torch.rand(H, device=dev, generator=g)]).contiguous()
goa = torch.randn(B, M, O, device=dev, generator=g)
ksel = torch.randint(0, 32, (B, M, O), device=dev, generator=g).to(torch.uint8)
rows = O + H
splits = 512
def run(Rpart, sp):
    _call("apn_sa_wide_wgrad", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(), pack1.data_ptr(),
          goa.data_ptr(), ksel.data_ptr(), splits, Rpart.data_ptr(), sp.data_ptr())
Rref = torch.empty(splits, rows, H, device=dev); sref = torch.empty(splits, H, device=dev)
run(Rref, sref); torch.cuda.synchronize()
mode = os.environ.get("MODE", "k")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    Rw = torch.empty(splits, rows, H, device=dev); sw = torch.empty(splits, H, device=dev)
    run(Rw, sw)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    Rp = torch.empty(splits, rows, H, device=dev); sp = torch.empty(splits, H, device=dev)
    run(Rp, sp)
    if mode == "s":
        R = Rp.double().sum(0)
for i in range(3):
    gr.replay(); torch.cuda.synchronize()
    print("mode", mode, "replay", i, "Rpart max diff", float((Rp - Rref).abs().max()), "suma", float((sp - sref).abs().max()),
          ("Rsum diff %.3e" % float((R - Rref.double().sum(0)).abs().max())) if mode == "s" else "")
