"""Which operator of the imitator's grouper goes wrong when another kernel stream runs beside it, and beside WHAT?
Graph A: the four PointsetGrouper index stages + group_max on fixed inputs.  Graph B (replayed at the same time on
another stream): one of several candidate workloads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import fused, graphs
from adaptpoint_amd.layers import ball_query, furthest_point_sample
from adaptpoint_amd.pointset import group_max

dev = torch.device("cuda:0")
B, N = 32, 1024
pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=700)).to(dev)
pos2 = torch.from_numpy(GI.unit_sphere_cloud(2 * B, N, seed=701)).to(dev)
feat = torch.randn(B, N, 128, device=dev)
alpha, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)


def work_a():
    out = {}
    xyz = pos
    for i, r in enumerate((0.1, 0.2, 0.4, 0.8)):
        fidx = furthest_point_sample(xyz, xyz.shape[1] // 2)
        new = torch.gather(xyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3))
        idx = ball_query(r, 24, xyz, new)
        out[f"fps{i}"], out[f"new{i}"], out[f"bq{i}"] = fidx, new, idx
        if i == 0:
            out["copy0"] = feat.permute(0, 2, 1).contiguous().permute(0, 2, 1).contiguous()
            out["gmax0"] = group_max(feat, idx, fidx, alpha, beta)
        if i == 1:
            out["copy1"] = feat[:, :512].permute(0, 2, 1).contiguous()
            out["gmax1"] = group_max(feat[:, :512].contiguous(), idx, fidx, alpha, beta)
        xyz = new
    return out


def b_fps():
    return [furthest_point_sample(pos2, 512) for _ in range(4)]


def b_fps_xyz():
    return [fused.sample_and_query(pos2, 512, 0.15, 32, geo=False) for _ in range(3)]


def b_bq():
    new = pos2[:, :512].contiguous()
    return [ball_query(0.15, 32, pos2, new) for _ in range(20)]


def b_fill():
    return [torch.full((64, 1024), 1e10, device=dev) for _ in range(50)]


def b_elementwise():
    x = torch.randn(64, 1024, 64, device=dev)
    return [x * 1.5 + i for i in range(60)]


from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
enc = C.encoder
pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
with torch.no_grad():
    p1, f1 = enc.encoder[0][0]([pos, pts])
    p1, f1 = enc.encoder[1][0]([p1, f1])
    smp2 = enc.index_pyramid(pos)[2]


def b_stage2():
    return [enc.encoder[2][0]([p1, f1], sampling=smp2) for _ in range(12)]


def b_stage2_with_index():
    return enc.encoder[2][0]([p1, f1])


with torch.no_grad():
    ref = {k: v.clone() for k, v in work_a().items()}
    torch.cuda.synchronize()
    warm = torch.cuda.Stream()
    cands = {"FPS (drop-in entry)": b_fps, "sampler + ball query (fused blocks' entry)": b_fps_xyz, "ball query": b_bq,
             "fills": b_fill, "elementwise": b_elementwise,
             "classifier stage 2 (wide block), index handed in": b_stage2,
             "classifier stage 2 (wide block) with its index stage": b_stage2_with_index}
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
        bad = {}
        for it in range(8):
            torch.cuda.synchronize()
            with torch.cuda.stream(sb_):
                gb.replay()
            with torch.cuda.stream(sa_):
                ga.replay()
            torch.cuda.synchronize()
            for k in ref:
                if not torch.equal(cap[k], ref[k]):
                    bad[k] = bad.get(k, 0) + 1
        print(f"beside {name}: rounds (of 8) in which an output differed:", bad or "none", flush=True)
