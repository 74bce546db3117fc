"""Per-operator timing on the GPU box (HIP events on torch's stream).

    python scripts/microbench_ops.py fps        # FPS by wave geometry and size
    python scripts/microbench_ops.py ops        # every extension op at the bench shape
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import golden_inputs as GI
from adaptpoint_amd import _lib, ops


def time_us(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    return t[len(t) // 2], t[0]


def bench_fps():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    for (b, n, m) in [(32, 1024, 512), (32, 512, 256), (32, 256, 128), (32, 128, 64), (32, 2048, 1200), (32, 2048, 512), (32, 4096, 1024), (32, 8192, 512), (1, 1024, 512), (256, 1024, 512)]:
        xyz = torch.from_numpy(GI.unit_sphere_cloud(b, n, seed=0)).to(dev)
        temp = torch.empty(b, n, device=dev)
        idx = torch.empty(b, m, dtype=torch.int32, device=dev)
        line = f"B={b:4d} N={n:5d} M={m:5d}: "
        for algo in (0, 1):
          line += f" [algo{algo}]"
          for w in (0, 2, 4, 8, 16):
            if w and (n + w * 64 - 1) // (w * 64) > 16:
                line += f" w{w}=  n/a "
                continue
            def run():
                temp.fill_(1e10)
                lib.apn_furthest_point_sampling_tuned(b, n, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr(),
                                                      w, algo, torch.cuda.current_stream().cuda_stream)
            med, mn = time_us(run)
            fill, _ = time_us(lambda: temp.fill_(1e10))
            line += f" w{w}={med - fill:6.1f}us({(med - fill) * 1e3 / max(m - 1, 1):4.0f})"
        print(line, flush=True)


def bench_ops():
    dev = torch.device("cuda:0")
    B, N, M, K, C = 32, 1024, 512, 32, 32
    xyz = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=0)).to(dev)
    temp = torch.full((B, N), 1e10, device=dev)
    fidx = torch.empty(B, M, dtype=torch.int32, device=dev)
    ops.furthest_point_sampling_wrapper(B, N, M, xyz, temp, fidx)
    q = torch.gather(xyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = torch.zeros(B, M, K, dtype=torch.int32, device=dev)
    print("ball_query     %8.1f us" % time_us(lambda: ops.ball_query_wrapper(B, N, M, 0.15, K, q, xyz, idx))[0])
    f = torch.randn(B, C, N, device=dev)
    out = torch.empty(B, C, M, K, device=dev)
    print("group C=32     %8.1f us" % time_us(lambda: ops.group_points_wrapper(B, C, N, M, K, f, idx, out))[0])
    g = torch.randn(B, C, M, K, device=dev)
    gp = torch.zeros(B, C, N, device=dev)
    print("group_grad C=32%8.1f us" % time_us(lambda: ops.group_points_grad_wrapper(B, C, N, M, K, g, idx, gp))[0])
    xt = xyz.transpose(1, 2).contiguous()
    o3 = torch.empty(B, 3, M, K, device=dev)
    print("group C=3      %8.1f us" % time_us(lambda: ops.group_points_wrapper(B, 3, N, M, K, xt, idx, o3))[0])
    d2 = torch.empty(B, N, 3, device=dev)
    i3 = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
    print("three_nn       %8.1f us" % time_us(lambda: ops.three_nn_wrapper(B, N, M, xyz, q, d2, i3))[0])
    w = torch.rand(B, N, 3, device=dev)
    fm = torch.randn(B, 64, M, device=dev)
    o = torch.empty(B, 64, N, device=dev)
    print("three_interp   %8.1f us" % time_us(lambda: ops.three_interpolate_wrapper(B, 64, M, N, fm, i3, w, o))[0])
    go = torch.randn(B, 64, N, device=dev)
    gm = torch.zeros(B, 64, M, device=dev)
    print("three_interp_g %8.1f us" % time_us(lambda: ops.three_interpolate_grad_wrapper(B, 64, N, M, go, i3, w, gm))[0])


def bench_seg():
    """the segmentation decoder's feature propagation (three_nn +
    three_interpolate fwd/bwd) at S3DIS level sizes, B=16 clouds of 15000 points."""
    dev = torch.device("cuda:0")
    B = 16
    gen = torch.Generator(dev).manual_seed(0)
    for n, m, c in [(15000, 3750, 64), (3750, 937, 128), (937, 234, 256), (234, 58, 512)]:
        u = torch.rand(B, n, 3, device=dev, generator=gen)
        kn = u[:, ::n // m][:, :m].contiguous()
        d2 = torch.empty(B, n, 3, device=dev)
        i3 = torch.empty(B, n, 3, dtype=torch.int32, device=dev)
        t_nn = time_us(lambda: ops.three_nn_wrapper(B, n, m, u, kn, d2, i3))[0]
        w = torch.rand(B, n, 3, device=dev)
        fm = torch.randn(B, c, m, device=dev)
        o = torch.empty(B, c, n, device=dev)
        t_i = time_us(lambda: ops.three_interpolate_wrapper(B, c, m, n, fm, i3, w, o))[0]
        go = torch.randn(B, c, n, device=dev)
        gm = torch.zeros(B, c, m, device=dev)
        t_g = time_us(lambda: ops.three_interpolate_grad_wrapper(B, c, n, m, go, i3, w, gm))[0]
        tests = B * n * m
        ib = B * (c * m * 4 + n * 24 + c * n * 4)
        print(f"n={n:6d} m={m:5d} c={c:4d}: three_nn {t_nn:8.1f} us ({tests / t_nn / 1e3:7.1f} G dist/s)  "
              f"interp {t_i:7.1f} us ({ib / t_i / 1e3:6.0f} GB/s)  interp_grad {t_g:7.1f} us")


if __name__ == "__main__" and (len(sys.argv) < 2 or sys.argv[1] in ("fps", "ops", "seg")):
    {"fps": bench_fps, "ops": bench_ops, "seg": bench_seg}[sys.argv[1] if len(sys.argv) > 1 else "fps"]()


def fps_stamps():
    """Where one FPS step spends its cycles (diagnostic kernel with s_memtime stamps)."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    b, n, m = 32, 1024, 512
    xyz = torch.from_numpy(GI.unit_sphere_cloud(b, n, seed=0)).to(dev)
    temp = torch.full((b, n), 1e10, device=dev)
    idx = torch.empty(b, m, dtype=torch.int32, device=dev)
    dbg = torch.zeros(8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        temp.fill_(1e10)
        rc = lib.apn_fps_debug_stamps(b, n, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr(), dbg.data_ptr(), st)
        assert rc == 0
    torch.cuda.synchronize()
    ref = torch.empty(b, m, dtype=torch.int32, device=dev)
    temp.fill_(1e10)
    ops.furthest_point_sampling_wrapper(b, n, m, xyz, temp, ref)
    assert torch.equal(ref, idx), "stamped kernel diverged from the product kernel"
    names = ["update", "wave max (6 DPP + readlane)", "ballot/ff1 + LDS write (+lgkm wait)", "s_barrier",
             "LDS read", "group max + ballot + 4 readlanes"]
    c = dbg.cpu().numpy()[:6] / (m - 1)
    print("memtime ticks per step (100 MHz? or shader clock -- see ratio), incl. ~40 per stamp:")
    for nme, v in zip(names, c):
        print(f"  {nme:40s} {v:8.1f}")
    print(f"  total {c.sum():.1f}")


if len(sys.argv) > 1 and sys.argv[1] == "stamps":
    fps_stamps()
