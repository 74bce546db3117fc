import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import set_abstraction as SA
from adaptpoint_amd.pointnext import PointNextSClassifier
dev = torch.device("cuda:0")
pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
gt = torch.randint(0, 15, (32,), device=dev, generator=torch.Generator(dev).manual_seed(0))
for wide in (False, True):
    SA.PREFER_WIDE = wide
    for graph in (False, True):
        torch.manual_seed(0)
        model = PointNextSClassifier(fused=True).to(dev).train()
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, capturable=True)
        def step():
            opt.zero_grad(set_to_none=True)
            logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
            opt.step()
            return loss
        losses = []
        if graph:
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    losses.append(step().item())
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                lg = step()
            for _ in range(12):
                g.replay(); losses.append(lg.item())
        else:
            for _ in range(16):
                losses.append(step().item())
        print("wide-first" if wide else "stage1-old", "graph" if graph else "eager", " ".join("%.4f" % v for v in losses), flush=True)
