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
for name, fused, wide in (("stage1-old", True, False), ("wide-first", True, True)):
    SA.PREFER_WIDE = wide
    torch.manual_seed(0)
    model = PointNextSClassifier(fused=fused).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, capturable=True)
    gn_box = []
    def step():
        opt.zero_grad(set_to_none=True)
        logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
        opt.step()
        return loss, gn
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        lg, gng = step()
    out = []
    for it in range(120):
        g.replay()
        l, n = lg.item(), gng.item()
        out.append("%.3f/%.1f" % (l, n))
        if not (l == l and n == n and n < 1e30):
            bad = [k for k, q in model.named_parameters() if not torch.isfinite(q).all()]
            badb = [k for k, q in model.named_buffers() if not torch.isfinite(q.float()).all()]
            print(name, "non-finite at replay", it, bad[:5], badb[:5])
            break
    print(name, " ".join(out[::4]), flush=True)
