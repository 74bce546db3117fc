"""Which PyTorch operators (by call site) launch the small kernels of the joint step: one eager `train_gan`
iteration under torch.profiler, device time and launch count per (operator, python frame)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.profiler import profile, ProfilerActivity
import golden_inputs as GI
from adaptpoint_amd.augmentor import AdaptPointAugmentor
from adaptpoint_amd.discriminator import PointDiscriminator1
from adaptpoint_amd.gan import GanStep
from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy

dev = torch.device("cuda:0")
B, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=0))
points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).to(dev)
label = (torch.arange(B) % 15).to(dev)
torch.manual_seed(0)
G, D, C = AdaptPointAugmentor().to(dev), PointDiscriminator1(num_classes=15).to(dev), PointNextSClassifier(fused=True).to(dev)
step = GanStep(G, D, C, SmoothCrossEntropy(0.3), batched_feedback=True, capturable=True)
for _ in range(3):
    step(points, label, device_noise=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(points, label, device_noise=True)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print(f"aten ops with device time: {sum(r[0] for r in rows) / 1e3:.2f} ms over {sum(r[1] for r in rows)} calls")
for dt, cnt, key, shp in rows[:90]:
    print(f"{dt / 1e3:8.3f} ms {cnt:4d}x  {key[:30]:30s} {shp}")
