"""Which PyTorch operators (name, shapes) the classifier training step still launches: one eager step of
scripts/bench_pointnext.py's loop under torch.profiler."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.profiler import profile, ProfilerActivity
import golden_inputs as GI
from adaptpoint_amd.pointnext import PointNextSClassifier

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = PointNextSClassifier(fused=True).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, capturable=True)
pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
gt = torch.randint(0, 15, (32,), device=dev)
data = {'pos': pos, 'x': x}


def step():
    opt.zero_grad(set_to_none=True)
    logits, loss = model.get_logits_loss(data, gt)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    rows.append((dt, e.count, e.key, str(e.input_shapes)[:120]))
rows.sort(reverse=True)
print(f"aten ops with device time: {sum(r[0] for r in rows) / 1e3:.2f} ms over {sum(r[1] for r in rows)} calls")
for dt, cnt, key, shp in rows[:70]:
    print(f"{dt / 1e3:8.3f} ms {cnt:4d}x  {key[:30]:30s} {shp}")
