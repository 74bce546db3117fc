import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd.pointnext import PointNextSClassifier
T = os.environ.get("T", "")
def main():
    from adaptpoint_amd import set_abstraction as SA, fused_wide
    if "w" in T: SA.PREFER_WIDE = True
    if "n" in T: fused_wide.WIDTHS = ()
    if "z" in T:
        _e = torch.empty
        torch.empty = lambda *a, **k: torch.zeros(*a, **k)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = PointNextSClassifier(fused=True).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, capturable=True)
    pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    gt = torch.randint(0, 15, (32,), device=dev, generator=torch.Generator(dev).manual_seed(0))
    def step():
        opt.zero_grad(set_to_none=True)
        logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
        opt.step()
        return (loss, gn) if "g" in T else loss
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        res = step()
    P = dict(model.named_parameters())
    n = "prediction.head.4.0.weight"
    out = []
    for i in range(4):
        graph.replay(); torch.cuda.synchronize()
        l = res[0] if "g" in T else res
        out.append("%.3f/%.2e" % (float(l.detach()), P[n].grad.norm().item()))
        bad = [k for k, q in P.items() if q.grad is not None and not torch.isfinite(q.grad).all()]
        badp = [k for k, q in P.items() if not torch.isfinite(q).all()]
        badb = [k for k, q in model.named_buffers() if not torch.isfinite(q.float()).all()]
        if i < 2:
            print("replay", i, "nonfinite grads:", bad, "params:", badp[:3], "buffers:", badb[:3])
    print("T=%r" % T, " ".join(out))
main()
