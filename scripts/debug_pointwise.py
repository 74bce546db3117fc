"""Debug aid: csrc/pointwise.hip layer against float64 modules, error per quantity and shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_pointwise import SHAPES, _layer, _reference, _rel
from adaptpoint_amd import pointwise

dev = torch.device("cuda:0")
for relu in (True, False):
    for (B, C, O, N) in SHAPES:
        conv, bn = _layer(C, O, dev, seed=B + C + O + N)
        g = torch.Generator(dev).manual_seed(1)
        x = torch.randn(B, C, N, device=dev, generator=g).requires_grad_(True)
        gout = torch.randn(B, O, N, device=dev, generator=g)
        ref = _reference(conv, bn, x, gout, relu)
        out = pointwise.conv_bn_act(x, conv, bn, relu=relu)
        out.backward(gout)
        flips = int(((out > 0) != (ref[0] > 0)).sum()) if relu else 0
        print(f"relu={relu} {B}x{C}x{O}x{N}: out {_rel(out.detach(), ref[0]):.1e} gx {_rel(x.grad, ref[1]):.1e} "
              f"gw {_rel(conv.weight.grad, ref[2]):.1e} gg {_rel(bn.weight.grad, ref[3]):.1e} gb {_rel(bn.bias.grad, ref[4]):.1e} "
              f"mask flips {flips}")
