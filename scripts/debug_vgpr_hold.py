"""Does plain register state survive a long kernel that runs BESIDE other kernels?  `apn_debug_vgpr_hold` keeps 24 values
per lane live for ~250 us (a barrier and an LDS atomic per turn, as the FPS step) and checks them; here it is replayed
from a graph on one stream while the classifier's blocks replay on another, as in tests/test_gpu_concurrency.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd.fused import _call
from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name

dev = torch.device("cuda:0")
B = 32
pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=900)).to(dev)
pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
bad = torch.zeros(32, dtype=torch.int64, device=dev)
turns = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
VPK = any(a in sys.argv for a in ("vpk", "vpk1", "vpk2"))    # the packed-FP32 probe instead of the register-hold kernel
# vpk1: the op_sel:[0,1] form (a pair's high register feeds both lanes); vpk2: the same behind `s_nop 7`
FORM = 2 if "vpk2" in sys.argv else 1 if "vpk1" in sys.argv else 0


def hold():
    for _ in range(6):
        if VPK:
            _call("apn_debug_vpk_probe", dev, 640, turns * 20, FORM, bad.data_ptr())
        else:
            _call("apn_debug_vgpr_hold", dev, 640, turns, bad.data_ptr())


def feature_work():
    keep = []
    with torch.no_grad():
        for _ in range(3):
            p0, f0 = pos, pts
            for stage in C.encoder.encoder:
                p0, f0 = stage[0]([p0, f0])
            keep.append(f0)
    return keep


def capture(fn):
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        fn()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


ga, _ = capture(hold)
gb, keep = capture(feature_work)
bad.zero_()
for _ in range(3):
    ga.replay()
torch.cuda.synchronize()
print("alone:  ", bad[:4].tolist() if VPK else bad[:24].tolist(), "(high halves wrong, of those with the other half, low halves wrong, results checked)" if VPK else "lanes with a changed value, per held value")
bad.zero_()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for it in range(12):
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        gb.replay()
    with torch.cuda.stream(sa):
        ga.replay()
torch.cuda.synchronize()
print("beside the classifier's blocks:", bad[:4].tolist() if VPK else bad[:24].tolist())
if VPK:
    import struct
    f32 = lambda u: struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]
    w = [int(v) & 0xFFFFFFFFFFFFFFFF for v in bad.tolist()]
    for k in range(min(8, w[7])):
        a, b, c = w[8 + 3 * k], w[9 + 3 * k], w[10 + 3 * k]
        lane = a & 255
        print(f"sample: turn {a >> 32} block {(a >> 8) & 0xFFFFFF} lane {lane}: found {f32(b >> 32)!r} wanted {f32(b)!r} "
              f"(a0 = {0.25 + 0.001 * lane:.6f}, operand {f32(c >> 32)!r}, live other half {1000.0 + lane}); other lane's result {f32(c)!r}")
else:
    print("samples (index, value found):", [(int(v) >> 32, hex(int(v) & 0xFFFFFFFF)) for v in bad[24:].tolist() if v])
