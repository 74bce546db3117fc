import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adaptpoint_amd import _lib, fused
lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.zeros(4, 64, device=dev); y = torch.zeros(64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    el = time.perf_counter() - t0; torch.cuda.synchronize()
    return el / n * 1e6
print("raw ctypes launch (tiny kernel)      %.1f us" % t(lambda: lib.apn_sa_bn_fold(None, 0, None, 32, 1.0, None, None, 1e-5, 0.1, x.data_ptr(), x.data_ptr(), None, 0, x.data_ptr(), None, 0, None, st)))
call = fused._Launcher(dev)
print("via _Launcher                         %.1f us" % t(lambda: call("apn_sa_bn_fold", None, 0, None, 32, 1.0, None, None, 1e-5, 0.1, x.data_ptr(), x.data_ptr(), None, 0, x.data_ptr(), None, 0, None)))
print("torch.empty                           %.1f us" % t(lambda: torch.empty(32, 512, 64, device=dev)))
print("torch.zeros 4MB                       %.1f us" % t(lambda: torch.zeros(1 << 20, device=dev)))
print("torch add_ tiny                       %.1f us" % t(lambda: y.add_(1.0)))
print("data_ptr()                            %.2f us" % t(lambda: x.data_ptr(), 20000))
print("current_stream().cuda_stream          %.2f us" % t(lambda: torch.cuda.current_stream(dev).cuda_stream, 20000))
