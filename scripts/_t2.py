import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd.pointnext import PointNextSClassifier
from adaptpoint_amd import set_abstraction as SA, fused_wide
T = os.environ.get("T", "")
def main():
    if "w" in T: SA.PREFER_WIDE = True
    if "n" in T: fused_wide.WIDTHS = ()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = PointNextSClassifier(fused=True).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, capturable=True)
    pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    gt = torch.randint(0, 15, (32,), device=dev, generator=torch.Generator(dev).manual_seed(0))
    def fb():
        opt.zero_grad(set_to_none=True)
        logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, gt)
        loss.backward()
        return loss
    def upd():
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
        opt.step()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fb(); upd()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph(); opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        lg = fb()
        if "u" in T:
            upd()
    P = dict(model.named_parameters())
    for i in range(3):
        graph.replay(); torch.cuda.synchronize()
        norms = {k: q.grad.norm().item() for k, q in P.items()}
        bad = [k for k, v in norms.items() if not (v == v and v < 1e30)]
        big = sorted(norms.items(), key=lambda kv: -kv[1] if kv[1] == kv[1] else -1e99)[:3]
        print("T=%r replay %d loss %.4f bad=%s top=%s" % (T, i, lg.item(), bad[:4], [(k[16:], "%.2e" % v) for k, v in big]))
        if "u" not in T:
            upd()
main()
