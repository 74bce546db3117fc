"""What exactly goes wrong in the LDS-atomic FPS step beside other kernels?  A debug build compares every pick with the
reference sequence (computed alone) and, at a cloud's first wrong step, dumps the state of the two lanes involved; the true
minimum distances of both points at that step are recomputed here in float32 numpy from the reference picks.

Needs the debug hooks in csrc/fps.hip (`apn_fps_debug_set`, the block behind `fps_dbg_ref` in fps_atomic_body), which are
NOT in the tree: the patch is in this file's history (round 3).  Recorded result (MI355X, ROCm 7.2): in every step of a
failing run all lanes of all waves hold the SAME centre, the slot's key is one lane's genuine key and the recorded pick is
that centre -- and still, at the first wrong step, either the winner's running minimum is LARGER than its true minimum
distance to the picks so far (cloud 24: 2.4e-2 for a point picked earlier, true 0: it missed an update) or the rightful
winner's is SMALLER (1.027e-2 against 1.192e-2: it saw a centre that was never picked).  State in registers that is
inconsistent with the instruction stream: below what this kernel's source can explain."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import golden_inputs as GI
from adaptpoint_amd import _lib
from adaptpoint_amd.layers import furthest_point_sample
from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name

dev = torch.device("cuda:0")
B, N, M = 32, 1024, 512
pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=900)).to(dev)
pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
enc = C.encoder
lib = ctypes.CDLL(_lib.LIB_PATH)
ref = furthest_point_sample(pos, M)
torch.cuda.synchronize()
dump = torch.zeros(B + 1, 16, dtype=torch.int32, device=dev)
lib.apn_fps_debug_set.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.apn_fps_debug_set(ref.data_ptr(), dump.data_ptr()) == 0


def index_work():
    return [furthest_point_sample(pos, M) for _ in range(6)]


def feature_work():
    keep = []
    with torch.no_grad():
        for _ in range(3):
            p0, f0 = pos, pts
            for stage in enc.encoder:
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


dump.zero_()
ga, got = capture(index_work)
gb, keep = capture(feature_work)
dump.zero_()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
xyz = pos.cpu().numpy()
refn = ref.cpu().numpy()
events = 0
for it in range(12):
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        gb.replay()
    with torch.cuda.stream(sa):
        ga.replay()
    torch.cuda.synchronize()
    d = dump.cpu().numpy().view(np.uint32)
    if d[B, 2]:
        print(f"round {it}: steps in which lanes of ONE wave held different centres: {d[B, 0]}; in which the WAVES of a workgroup held different centres: {d[B, 1]} (of {d[B, 2]} steps)")
    for c in range(B):
        if d[c, 0] == 0:
            continue
        events += 1
        j, gotp, khi, klo, want = int(d[c, 0]), int(d[c, 1]), d[c, 2], d[c, 3], int(d[c, 4])
        picks = refn[c, :j]
        def true_min(p):
            diff = xyz[c, p][None, :] - xyz[c, picks]
            d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
            return np.float32(d2.min())
        f = lambda u: np.array([u], np.uint32).view(np.float32)[0]
        print(f"round {it} cloud {c}: step {j}: picked {gotp}, reference {want}; slot key distance {f(khi):.6e}")
        print(f"    reference winner's lane: best {f(d[c,5]):.6e}, its slot's dmin {f(d[c,6]):.6e} (true {true_min(want):.6e}), wave max {f(d[c,7]):.6e}, bslot {d[c,8]} / slot {d[c,9]}")
        print(f"    actual winner's lane:    best {f(d[c,10]):.6e}, its slot's dmin {f(d[c,11]):.6e} (true {true_min(gotp):.6e}), wave max {f(d[c,12]):.6e}")
    dump.zero_()
    if events >= 6:
        break
print("events:", events)
