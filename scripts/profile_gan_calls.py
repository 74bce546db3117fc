"""Every extension launch of one eager joint step between its own HIP events: time, entry point, leading integer arguments.

    python scripts/profile_gan_calls.py [rows to print] [--order]
"""
import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import fused, fused_wide, pointwise, pointset, attention
from adaptpoint_amd.augmentor import AdaptPointAugmentor
from adaptpoint_amd.discriminator import PointDiscriminator1
from adaptpoint_amd.gan import GanStep
from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy
dev = torch.device("cuda:0")
B, N = 32, 1024
pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=0))
points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).to(dev)
label = (torch.arange(B) % 15).to(dev)
torch.manual_seed(0)
G, D, C = AdaptPointAugmentor().to(dev), PointDiscriminator1(num_classes=15).to(dev), PointNextSClassifier(fused=True).to(dev)
step = GanStep(G, D, C, SmoothCrossEntropy(0.3), batched_feedback=True, capturable=True)
for _ in range(3):
    step(points, label, device_noise=True)
torch.cuda.synchronize()
acc = []
orig = fused._call
def timed(name, d, *a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig(name, d, *a, **k); e.record()
    acc.append((name, tuple(x for x in a[:6] if isinstance(x, (int, float))), s, e))
    return r
mods = [fused, fused_wide, pointwise, pointset, attention]
import adaptpoint_amd.layers as L, adaptpoint_amd.spectral as SP, adaptpoint_amd.augmentor as AU
mods += [L, SP, AU]
for m in mods:
    if hasattr(m, "_call"): m._call = timed
step(points, label, device_noise=True)
torch.cuda.synchronize()
rows = [(s.elapsed_time(e) * 1e3, n, a) for n, a, s, e in acc]
tot = sum(r[0] for r in rows)
print(f"{len(rows)} extension calls, {tot/1e3:.2f} ms between their own events")
args = [x for x in sys.argv[1:] if not x.startswith("--")]
top = int(args[0]) if args else 45
if "--order" in sys.argv:            # in launch order, with the time since the step's first launch
    t0 = acc[0][2]
    for n, a, s, e in acc:
        print(f"{t0.elapsed_time(s) * 1e3:9.1f} us  +{s.elapsed_time(e) * 1e3:7.1f}  {n:36s} {a}")
else:
    for us, n, a in sorted(rows, reverse=True)[:top]:
        print(f"{us:8.1f} us  {n:36s} {a}")
