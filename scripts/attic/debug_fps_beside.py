"""The generator's second FPS (512 -> 256, 32 clouds) returned wrong picks when the classifier's stage 2 ran beside it
(scripts/debug_two_lane_forward.py).  Graph A: that FPS, eight times.  Graph B, replayed at the same time: candidates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import fused, graphs
from adaptpoint_amd.layers import ball_query, furthest_point_sample
from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name

dev = torch.device("cuda:0")
B = 32
pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=700)).to(dev)
C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
enc = C.encoder
pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with torch.no_grad():
    p0, f0 = enc.encoder[0][0]([pos, pts])
    p1, f1 = enc.encoder[1][0]([p0, f0])
    pyr = enc.index_pyramid(pos)
    xa = pos[:, :NA].contiguous()
sa2 = enc.encoder[2][0]


def work_a():
    return [furthest_point_sample(xa, NA // 2) for _ in range(8)]


from adaptpoint_amd import fused_wide


def whole2():
    return [sa2([p1, f1]) for _ in range(6)]


def nbr_only():
    out = []
    for _ in range(6):
        smp = fused.sample_and_query(p1, 256, sa2.grouper.radius, 32, geo=False)
        out.append(fused_wide.neighbour_index(smp.idx, smp.new_p, 512, fidx=smp.fidx))
        out.append(smp)
    return out


def sample_only():
    return [sa2.sample(p1) for _ in range(6)]


cands = {
    "stage-2 block whole (its own index stage) x6": whole2,
    "stage-2 sampler + neighbour_index x6": nbr_only,
    "stage-2 sa.sample x6": sample_only,
    "FPS 512 -> 256 (drop-in entry) x12": lambda: [furthest_point_sample(p1, 256) for _ in range(12)],
    "stage-2 sampler + ball query x12": lambda: [fused.sample_and_query(p1, 256, sa2.grouper.radius, 32, geo=False) for _ in range(12)],
    "stage-2 tile map + inverse map x12": lambda: [sa2.index_for(pyr[2], 512, 64) for _ in range(12)],
    "stage-2 wide block, index handed in x12": lambda: [sa2([p1, f1], sampling=pyr[2]) for _ in range(12)],
    "stage-1 fused block, index handed in x12": lambda: [enc.encoder[1][0]([p0, f0], sampling=pyr[1]) for _ in range(12)],
    "stage-3 wide block, index handed in x12": lambda: [enc.encoder[3][0]([pyr[2].new_p, sa2([p1, f1], sampling=pyr[2])[1]], sampling=pyr[3]) for _ in range(12)],
}
with torch.no_grad():
    ref = [t.clone() for t in work_a()]
    torch.cuda.synchronize()
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        work_a()
        for f in cands.values():
            f()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        cap = work_a()
    sa_, sb_ = torch.cuda.Stream(), torch.cuda.Stream()
    for name, f in cands.items():
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb):
            keep = f()
        bad = 0
        for it in range(8):
            torch.cuda.synchronize()
            with torch.cuda.stream(sb_):
                gb.replay()
            with torch.cuda.stream(sa_):
                ga.replay()
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(a, b)) for a, b in zip(cap, ref))
        print(f"FPS {NA} -> {NA // 2} beside [{name}]: wrong results {bad} of 64", flush=True)
