"""Where the microseconds of the chain's glue kernels go: wall-clock stamps per workgroup phase.

    APN_EXTRA_CXXFLAGS=-DAPN_WG_STAMPS python -m adaptpoint_amd.build --force && python scripts/stamp_glue.py

(the stamps exist in builds with -DAPN_WG_STAMPS only).  Printed per kernel and stamp: min / median / max over the
workgroups, in us from the first workgroup's entry, and the median time since the workgroup's own entry."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN  # noqa: E402
from adaptpoint_amd import _lib, fused  # noqa: E402

LABELS = {
    "apn_sa_prep_stats": ["entry", "loads arrived", "LDS written", "barrier 2", "table rows stored", "statistics", "end"],
    "apn_sa_fwd_out": ["entry", "loads issued", "barrier 1", "fold + normalise", "barrier 3", "outputs stored", "zero fill issued"],
    "apn_sa_bwd_prep": ["entry", "loads issued", "barrier 1", "goa + sums", "barrier 2", "acc adds", "dWs share", "gip adds"],
    "apn_sa_bwd_point_grads": ["entry", "W1 + constants", "barrier 1", "tile loads", "barrier 2", "G formed", "barrier 3", "end"],
}


def report(name, st):
    st = st.reshape(-1, 16).astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    print(f"{name}: {len(st)} workgroups")
    for k, label in enumerate(LABELS[name]):
        v = st[:, k]
        ok = v > 0
        if not ok.any():
            continue
        u = (v[ok] - t0) / 100.0
        own = (v[ok] - st[ok, 0]) / 100.0
        print(f"  {label:18s} min {u.min():6.2f}  med {np.median(u):6.2f}  max {u.max():6.2f}   since own entry: med {np.median(own):5.2f} max {own.max():5.2f} us")


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    import ctypes
    try:
        attach = lib.apn_sa_debug_wg_stamps
    except AttributeError:
        raise SystemExit("libadaptpoint_amd.so was built without -DAPN_WG_STAMPS")
    attach.argtypes, attach.restype = [ctypes.c_void_p], ctypes.c_int
    attach_f = lib.apn_sa_debug_wg_stamps_fused
    attach_f.argtypes, attach_f.restype = [ctypes.c_void_p], ctypes.c_int
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
    buf = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
    fused.PER_KERNEL_LAUNCH = True
    orig = fused._call
    got = {}

    def hooked(name, d, *a, **k):
        if name in LABELS:
            buf.zero_()
            torch.cuda.synchronize()
            assert (attach_f if name == 'apn_sa_prep_stats' else attach)(buf.data_ptr()) == 0
            orig(name, d, *a, **k)
            torch.cuda.synchronize()
            assert (attach_f if name == 'apn_sa_prep_stats' else attach)(None) == 0
            got[name] = buf.cpu().numpy().copy()
        else:
            orig(name, d, *a, **k)

    fused._call = hooked
    _, out = blk([p, f], sampling=smp)
    torch.autograd.backward([out], [ones.expand_as(out)])
    torch.cuda.synchronize()
    for name in LABELS:
        if name in got:
            report(name, got[name])


if __name__ == "__main__":
    main()
