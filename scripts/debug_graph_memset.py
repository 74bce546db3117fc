"""Root cause of the stale PyTorch reductions under hipGraph replay (measured-and-rejected 10).

PyTorch's multi-block reductions (ATen/native/cuda/Reduce.cuh) zero a semaphore buffer with cudaMemsetAsync
before EVERY launch and never reset it in the kernel.  Captured, that is a memset node.  This script checks, in
isolation, (1) a multi-block reduction replayed on changing data, (2) a single-block one, (3) a raw captured
hipMemsetAsync followed by an increment kernel.
"""
import ctypes
import torch

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


for rows, cols, tag in ((768, 128, "multi-block (rows 768 x 128)"), (16384, 64, "multi-block (rows 16384 x 64)"),
                        (8, 128, "single-block (rows 8 x 128)")):
    x = torch.ones(rows, cols, device=dev)
    g, y = capture(lambda: x.double().sum(0))
    got = []
    for i in range(4):
        x.fill_(float(i + 1))
        g.replay()
        torch.cuda.synchronize()
        got.append((float(y[0]), float(y[-1])))
    print(f"reduction {tag}: replays give", got, "expected", [(rows * (i + 1.0),) * 2 for i in range(4)])

# a raw memset node
buf = torch.full((256,), 7, dtype=torch.int32, device=dev)


def memset_then_inc():
    st = torch.cuda.current_stream().cuda_stream
    rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(buf.numel() * 4), ctypes.c_void_p(st))
    assert rc == 0, rc
    buf.add_(1)
    return buf


try:
    g, _ = capture(memset_then_inc)
    vals = []
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        vals.append(int(buf[0]))
    print("captured hipMemsetAsync + add_(1): buf[0] after replays", vals, "(1, 1, 1, 1 if the memset node replays)")
except Exception as e:     # noqa: BLE001
    print("captured hipMemsetAsync failed:", type(e).__name__, e)
