"""Stage-shape forward+backward of the width-generic path in a loop, for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide -- python3 scripts/prof_wide.py 0
(argument: index into the four PointNeXt-S stage shapes; the neighbour index is built once, outside the loop)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_fused_wide as T
from adaptpoint_amd import fused_wide

dev = torch.device("cuda:0")
cin, N, M, radius = T.STAGES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
p, new_p, f, idx, conv1, bn1, conv2, bn2 = T._setup(dev, cin, N, M, radius, B=batch, seed=5)
f.requires_grad_(True)
params = [conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias]
nbr = fused_wide.neighbour_index(idx, new_p, N)
ones = torch.ones(1, 1, 1, device=dev)
for _ in range(iters):
    f.grad = None
    for q in params:
        q.grad = None
    out = fused_wide.grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, index=nbr)
    torch.autograd.backward([out], [ones.expand_as(out)])
torch.cuda.synchronize()
print("tiles in use:", int(nbr.tmap[0]), "of", idx.shape[0] * idx.shape[1])
