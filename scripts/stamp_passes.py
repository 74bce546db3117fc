"""Where the time of the two tile passes goes: per-wave wall-clock stamps (apn_sa_debug_stamps) of one launch.

    python scripts/stamp_passes.py [--tmap-bwd 0|1]

Stamps (100 MHz): 0 entry, 1 tile count known, 2 prologue done, 3..5 tiles 1..3 done, 6 loop done, 7 kernel end.
Printed: for every stamp the min / median / max over the waves, in us from the first wave's entry."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN  # noqa: E402
from adaptpoint_amd import _lib, fused  # noqa: E402


def report(name, st):
    st = st.reshape(-1, 16).astype(np.int64)
    live = st[:, 0] > 0
    st = st[live]
    t0 = st[:, 0].min()
    print(f"{name}: {live.sum()} waves")
    labels = ["entry", "tiles known", "prologue done", "tile 1", "tile 2", "tile 3+", "loop done", "end",
              "t2: body start", "t2: conv1+bn", "t2: sparse drops", "t2: dL/da1 mfma", "t2: stats", "t2: per-query sums",
              "t2: gram", "-"]
    for k, label in enumerate(labels):
        v = st[:, k]
        v = v[v > 0]
        if len(v):
            u = (v - t0) / 100.0
            print(f"  {label:14s} n={len(v):5d}  min {u.min():7.2f}  med {np.median(u):7.2f}  p90 {np.percentile(u, 90):7.2f}  max {u.max():7.2f} us")


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    torch.manual_seed(0)
    blk = BN.make_block(fused=True).to(dev).train()
    p, f = BN.make_inputs(32, 0)
    p, f = p.to(dev), f.to(dev).requires_grad_(True)
    smp = blk.sample(p)
    blk.index_for(smp, 1024, 32)
    ones = torch.ones(1, 1, 1, device=dev)
    for _ in range(3):
        _, out = blk([p, f], sampling=smp)
        torch.autograd.backward([out], [ones.expand_as(out)])
    torch.cuda.synchronize()
    buf = torch.zeros(16 * 4 * 1024, dtype=torch.int64, device=dev)
    fused.PER_KERNEL_LAUNCH = True
    orig = fused._call

    def hooked(name, d, *a, **k):
        if name in ("apn_sa_fwd_main", "apn_sa_bwd_main"):
            buf.zero_()
            torch.cuda.synchronize()
            lib.apn_sa_debug_stamps(buf.data_ptr())
            orig(name, d, *a, **k)
            torch.cuda.synchronize()
            lib.apn_sa_debug_stamps(None)
            report(name, buf.cpu().numpy())
        else:
            orig(name, d, *a, **k)
    fused._call = hooked
    _, out = blk([p, f], sampling=smp)
    torch.autograd.backward([out], [ones.expand_as(out)])
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
