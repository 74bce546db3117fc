import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd.pointnext import PointNextSClassifier
def main():
    V = sys.argv[1:]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    from adaptpoint_amd import set_abstraction as SA
    SA.PREFER_WIDE = True
    if "dataFirst" in V:
        pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
        x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
        gt = torch.randint(0, 15, (32,), device=dev, generator=torch.Generator(dev).manual_seed(0))
    model = PointNextSClassifier(fused=True).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, capturable=True)
    if "dataFirst" not in V:
        pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
        x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
        gt = torch.randint(0, 15, (32,), device=dev, generator=torch.Generator(dev).manual_seed(0))
    def step():
        opt.zero_grad(set_to_none=True)
        logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
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
    for it in range(30):
        g.replay()
        if it % 5 == 0:
            out.append("%.3f" % lg.item())
    print(V, " ".join(out), flush=True)

main()
