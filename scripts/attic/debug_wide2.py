import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_fused_wide as T
from fused_reference import chain_grad
from adaptpoint_amd.fused_wide import grouped_mlp_max
dev = torch.device("cuda:0")
for (cin, N, M, radius) in T.STAGES:
  for seed, B, neg in ((5, 4, True), (5, 4, False), (7, 2, True), (11, 4, True)):
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = T._setup(dev, cin, N, M, radius, B=B, seed=seed, neg_gamma=neg)
    H, O = conv1.weight.shape[0], conv2.weight.shape[0]
    wts = torch.randn(B, O, M, device=dev, generator=torch.Generator(dev).manual_seed(9))
    def ref(noise=None):
        leaves = [t.detach().clone().requires_grad_(True) for t in T._chain_args(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)
                  if torch.is_tensor(t) and t.is_floating_point()]
        out, mid = chain_grad(leaves[0], leaves[1], leaves[2], idx, radius, *leaves[3:], y1_noise=noise)
        (out * wts.double()).sum().backward()
        return {k: t.grad for k, t in zip(("p", "newp", "f", "w1", "g1", "b1", "w2", "g2", "b2"), leaves)}, mid
    want, mid = ref()
    noise = torch.randn(mid["y1"].shape, device=dev, dtype=torch.float64, generator=torch.Generator(dev).manual_seed(1)) * 1e-6
    pert, _ = ref(noise)
    p.requires_grad_(True); new_p.requires_grad_(True); f.requires_grad_(True)
    out = grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)
    (out * wts).sum().backward()
    got = dict(f=f.grad, p=p.grad, newp=new_p.grad, w1=conv1.weight.grad.view(H, -1), w2=conv2.weight.grad.view(O, H),
               g1=bn1.weight.grad, b1=bn1.bias.grad, g2=bn2.weight.grad, b2=bn2.bias.grad)
    pre = ((mid["y1"] - mid["m1"]) / torch.sqrt(mid["v1"] + 1e-5) * bn1.weight.double().view(1, -1, 1, 1) + bn1.bias.double().view(1, -1, 1, 1))
    near = float((pre.abs() < 1e-6).double().mean())
    print(f"C={cin} seed={seed} B={B} neg={neg} gates<1e-6: {near:.1e} | kernel:", {k: "%.0e" % T._rel_l2(got[k], want[k]) for k in ("f", "w1", "w2", "g1")},
          "| chain+1e-6 noise:", {k: "%.0e" % T._rel_l2(pert[k], want[k]) for k in ("f", "w1", "w2", "g1")}, flush=True)
