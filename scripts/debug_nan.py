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
for name, fused, wide in (("unfused", False, False), ("stage1-old", True, False), ("wide-first", True, True)):
    SA.PREFER_WIDE = wide
    torch.manual_seed(0)
    model = PointNextSClassifier(fused=fused).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05)
    losses = []
    for it in range(150):
        opt.zero_grad(set_to_none=True)
        logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
        opt.step()
        losses.append(loss.item())
        if not torch.isfinite(loss) or not torch.isfinite(gn):
            bad = [k for k, q in model.named_parameters() if q.grad is not None and not torch.isfinite(q.grad).all()]
            print(name, "non-finite at step", it, "loss", loss.item(), "gn", gn.item(), bad[:6])
            break
    print(name, " ".join("%.3f" % v for v in losses[::10]), flush=True)
