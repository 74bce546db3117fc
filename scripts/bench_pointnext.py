"""BASELINE configs[2]: full PointNeXt-S classifier training step (fwd + bwd + AdamW) on
random 15-class clouds, B=32, N=1024, one MI355X.  Not the headline metric (bench.py is);
a second data point for DESIGN.md.

    python scripts/bench_pointnext.py [--fused] [--steps 50]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import golden_inputs as GI
from adaptpoint_amd.pointnext import PointNextSClassifier


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", action="store_true", help="fused bf16x3 stage 1")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = PointNextSClassifier(fused=a.fused).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05)   # cfgs/scanobjectnn/default.yaml
    pos = torch.from_numpy(GI.unit_sphere_cloud(a.batch, 1024, seed=0)).to(dev)
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    gt = torch.randint(0, 15, (a.batch,), device=dev, generator=torch.Generator(dev).manual_seed(0))

    def step():
        opt.zero_grad(set_to_none=True)
        logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)     # train_autoaug.py:505-508
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"config": "PointNeXt-S classifier train step, B=%d N=1024" % a.batch,
                      "stage1": "fused bf16x3" if a.fused else "unfused ops + PyTorch fp32",
                      "ms_per_step": round(1e3 * el / a.steps, 3),
                      "clouds_per_s": round(a.batch * a.steps / el, 1), "loss": float(loss)}))


if __name__ == "__main__":
    main()
