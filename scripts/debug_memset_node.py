"""Which part of the joint step puts a MEMSET node into its hipGraph at N = 2048 (adaptpoint_amd/graphs.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptpoint_amd import graphs, synthetic as GI  # noqa: E402
from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise_on  # noqa: E402
from adaptpoint_amd.discriminator import PointDiscriminator1  # noqa: E402
from adaptpoint_amd.gan import feedback_loss  # noqa: E402
from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy  # noqa: E402

dev = torch.device("cuda:0")
B, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
G, D = AdaptPointAugmentor(fused=True).to(dev), PointDiscriminator1(num_classes=15, fused=True).to(dev)
C = PointNextSClassifier(fused=True).to(dev)
pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=3)).to(dev)
points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1)
label = torch.arange(B, device=dev) % 15
noise = draw_noise_on(dev, B, N, 4)
crit, bce = SmoothCrossEntropy(0.3), torch.nn.BCELoss()
real_t = torch.full((B, 1), 0.9, device=dev)


def census(name, fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = graphs.new_graph()
    with torch.cuda.graph(g):
        fn()
    print(name, graphs.node_census(g), flush=True)


G.train(); D.train(); C.eval()
xyz = points[:, :, :3].contiguous()
census("G forward", lambda: G(xyz, noise)[1].sum())
census("G forward + backward", lambda: G(xyz, noise)[1].sum().backward())
gen = G(xyz, noise)[1].detach()
census("D forward + backward", lambda: bce(D(gen), real_t).backward())


def fb():
    g = gen.clone().requires_grad_(True)
    fake = {'pos': g, 'x': torch.cat([g, points[:, :, 3:4]], -1).transpose(1, 2).contiguous()}
    real = {'pos': xyz, 'x': points[:, :, :4].transpose(1, 2).contiguous()}
    feedback_loss(C, crit, real, fake, label, 3.0, True, frozen=True)[0].backward()


census("feedback forward + backward", fb)

# which operator issues the memsets: an eager G forward under the profiler
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    G(xyz, noise)[1].sum()
    torch.cuda.synchronize()
evs = prof.events()
mems = [e for e in evs if "emset" in e.name]
print("memset events:", [(e.name, e.device_type) for e in mems][:6])
for m in mems:
    t0 = m.time_range.start
    # the CPU op whose interval contains the memset's launch (runtime call hipMemsetAsync precedes it)
    cands = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and e.time_range.start <= t0 <= e.time_range.end]
    print("  memset under:", [(c.name, c.input_shapes) for c in sorted(cands, key=lambda e: e.time_range.end - e.time_range.start)[:4]])
rt = [e for e in evs if "hipMemset" in e.name]
for m in rt:
    t0 = m.time_range.start
    cands = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and e is not m and e.time_range.start <= t0 <= e.time_range.end]
    print("  hipMemset call under:", [(c.name, c.input_shapes) for c in sorted(cands, key=lambda e: e.time_range.end - e.time_range.start)[:3]])
