"""FPS 1024 -> 512: time per launch against the number of clouds (one workgroup per cloud), and for clouds whose
masked points were moved to the origin (what the augmentor hands the feedback pass)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adaptpoint_amd.layers import furthest_point_sample
dev = torch.device("cuda:0")
torch.manual_seed(0)


def t(x, m, it=20):
    for _ in range(3):
        furthest_point_sample(x, m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        furthest_point_sample(x, m)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


for n, m in ((1024, 512), (512, 256)):
    for b in (8, 16, 32, 48, 64, 96, 128, 256):
        x = torch.rand(b, n, 3, device=dev) * 2 - 1
        print(f"n={n} m={m} b={b:4d}: {t(x, m):7.1f} us", flush=True)
x = torch.rand(64, 1024, 3, device=dev) * 2 - 1
mask = (torch.rand(64, 1024, 1, device=dev) > 0.3).float()
print(f"b=64 with 30% of the points at the origin: {t((x * mask).contiguous(), 512):7.1f} us")

# the fused blocks' sampler entry (also writes the sampled coordinates; no min-distance buffer)
from adaptpoint_amd.fused import _call


def t_xyz(x, m, it=20):
    b, n, _ = x.shape
    fidx = torch.empty(b, m, dtype=torch.int32, device=dev)
    newp = torch.empty(b, m, 3, device=dev)
    f = lambda: _call("apn_furthest_point_sampling_xyz", dev, b, n, m, x.data_ptr(), None, fidx.data_ptr(), newp.data_ptr())
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


for b in (32, 64, 128):
    x = torch.rand(b, 1024, 3, device=dev) * 2 - 1
    print(f"sampler + coordinates entry, n=1024 m=512 b={b}: {t_xyz(x, 512):7.1f} us")
x = torch.rand(64, 1024, 3, device=dev) * 2 - 1
print(f"  b=64, 30% of the points at the origin: {t_xyz((x * mask).contiguous(), 512):7.1f} us")
xs = x / x.norm(dim=-1, keepdim=True)
print(f"  b=64, points on the unit sphere: {t_xyz(xs.contiguous(), 512):7.1f} us")
half = torch.cat([(xs[:32] * mask[:32]), xs[32:]], 0).contiguous()
print(f"  b=64, first half masked sphere, second half sphere: {t_xyz(half, 512):7.1f} us")

# the joint step's own data: the augmentor's output stacked on the real clouds
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as GI
from adaptpoint_amd.augmentor import AdaptPointAugmentor
pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=0)).to(dev)
torch.manual_seed(0)
G = AdaptPointAugmentor().to(dev)
with torch.no_grad():
    _, gen = G(pos)
print("generated clouds: finite", bool(torch.isfinite(gen).all()), "share of points at the origin",
      float((gen.abs().sum(-1) == 0).float().mean()), "max |coordinate|", float(gen.abs().max()))
both = torch.cat([gen, pos], 0).contiguous()
print(f"  b=64, [generated; real]: {t_xyz(both, 512):7.1f} us;  generated alone {t_xyz(gen.contiguous(), 512):7.1f} us;  real alone {t_xyz(pos, 512):7.1f} us")
