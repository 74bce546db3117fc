"""Where a contraction kernel's microseconds go outside its main loop: wall-clock stamps (100 MHz) of every workgroup's
phases in a -DPW_STAMPS build of csrc/pointwise.hip (built here into adaptpoint_amd/variants/, selected by APN_LIB_PATH):

    python scripts/pw_stamps.py build        (here: hipcc cross-compiles)
    python scripts/pw_stamps.py [layer ...]  (on the GPU box)
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "adaptpoint_amd", "variants")
LIB = os.path.join(VAR, "libpw_stamps.so")

if sys.argv[1:2] == ["build"]:
    from adaptpoint_amd import build as B
    B.build()
    os.makedirs(VAR, exist_ok=True)
    flags = open(os.path.join(B.OBJ, "pointwise.o.flags")).read().split()
    others = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o") and f != "pointwise.o"]
    obj = os.path.join(VAR, "pw_stamps.o")
    subprocess.run([B.hipcc(), f"--offload-arch={B.ARCH}", *flags, "-DPW_STAMPS", "-c", os.path.join(B.CSRC, "pointwise.hip"), "-o", obj], check=True)
    subprocess.run([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", *others, obj, "-o", LIB], check=True)
    os.remove(obj)
    print(LIB)
    sys.exit(0)

os.environ["APN_LIB_PATH"] = LIB
os.environ["APN_ALLOW_UNSAFE_LIB"] = "1"
import numpy as np
import torch
from adaptpoint_amd import _lib
from adaptpoint_amd.fused import _call
from bench_pw_gemm import LAYERS

dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(LIB)
raw.apn_pw_debug_stamps.argtypes = [ctypes.c_void_p]
LABELS = ["entry", "first chunk staged", "main loop done", "tile stored", "end"]
B = 32
for name in (sys.argv[1:] or ["decode1", "extract1"]):
    C, O, N = LAYERS[name]
    x = torch.randn(B, C, N, device=dev); w = torch.randn(O, C, device=dev) * 0.05
    gy = torch.randn(B, O, N, device=dev)
    y = torch.empty(B, O, N, device=dev); gx = torch.empty_like(x); gw = torch.empty(O, C, device=dev)
    part = torch.empty(lib.apn_pw_conv_tiles(B, N), 2, O, device=dev)
    scratch = torch.empty(lib.apn_pw_conv_grad_weight_splits(B, C, O, N), O, C, device=dev)
    calls = {"forward": lambda: _call("apn_pw_conv_forward", dev, B, C, O, N, 3, x.data_ptr(), w.data_ptr(), y.data_ptr(), part.data_ptr()),
             "input gradient": lambda: _call("apn_pw_conv_grad_input", dev, B, C, O, N, 3, gy.data_ptr(), w.data_ptr(), gx.data_ptr()),
             "weight gradient": lambda: _call("apn_pw_conv_grad_weight", dev, B, C, O, N, 3, gy.data_ptr(), x.data_ptr(), scratch.data_ptr(), gw.data_ptr())}
    for what, fn in calls.items():
        st = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        assert raw.apn_pw_debug_stamps(st.data_ptr()) == 0
        fn()
        torch.cuda.synchronize()
        raw.apn_pw_debug_stamps(None)
        s = st.cpu().numpy()
        s = s[s[:, 0] > 0]
        t0 = s[:, 0].min()
        print(f"{name} {what}: {len(s)} workgroups")
        for k, label in enumerate(LABELS):
            v = s[:, k]
            ok = v > 0
            if not ok.any():
                continue
            u = (v[ok] - t0) / 100.0
            own = (v[ok] - s[ok, 0]) / 100.0
            print(f"  {label:20s} min {u.min():7.2f}  med {np.median(u):7.2f}  max {u.max():7.2f} us   since own entry: med {np.median(own):6.2f} max {own.max():6.2f}")
