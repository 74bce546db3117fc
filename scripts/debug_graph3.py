import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import set_abstraction as SA
from adaptpoint_amd.pointnext import PointNextSClassifier
dev = torch.device("cuda:0")
keep_gn = "--keep-gn" in sys.argv
pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
gt = torch.randint(0, 15, (32,), device=dev, generator=torch.Generator(dev).manual_seed(0))
SA.PREFER_WIDE = "--wide" in sys.argv
torch.manual_seed(0)
model = PointNextSClassifier(fused="--unfused" not in sys.argv).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, capturable=True)
box = []
def step():
    opt.zero_grad(set_to_none=True)
    logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
    if keep_gn:
        box.append(gn)
    opt.step()
    return loss
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    lg = step()
out = []
every = int(sys.argv[sys.argv.index("--every") + 1]) if "--every" in sys.argv else 1
for it in range(70):
    g.replay()
    if it % every == every - 1:
        out.append("%.3f" % lg.item())
bad = [k for k, q in model.named_parameters() if not torch.isfinite(q).all()]
print(sys.argv[1:], " ".join(out[-12:]), "nonfinite params:", bad[:4], flush=True)
