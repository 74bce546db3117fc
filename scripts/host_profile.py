"""cProfile of the eager bench step (host side) on the GPU box."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = bench.make_block(fused=True).to(dev).train()
p, f = bench.make_inputs(32, 0)
p = p.to(dev); f = f.to(dev).requires_grad_(True)
params = list(blk.parameters())
def step():
    f.grad = None
    for q in params: q.grad = None
    _, out = blk([p, f]); out.sum().backward()
for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time {t_issue/200*1e6:.0f} us/step, wall {t_all/200*1e6:.0f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(18); print(sio.getvalue()[:3500])
