"""Hunt for the hazard of GanStep's 'real' part: the classifier's eval pass on the second lane BESIDE the generator's
forward on the first, replayed from one graph -- which of the generator's intermediate results differ from the eager
ones?  (No backward, no optimizer: forward tensors only, same weights every replay.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import graphs
from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise_on
from adaptpoint_amd.gan import real_loss_ahead
from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy, fill_parameters_by_name

dev = torch.device("cuda:0")
B, N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4, 1024
G = fill_parameters_by_name(AdaptPointAugmentor(fused=True)).to(dev).train()
C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
crit = SmoothCrossEntropy(0.3)
pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=700)).to(dev)
points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1)
label = (torch.arange(B) % 15).to(dev)
noise = draw_noise_on(dev, B, N, G.num_anchor)
real = {'pos': pos, 'x': points.transpose(1, 2).contiguous()}
taps = {}


def tap(name):
    def hook(mod, inp, out):
        t = out[1] if isinstance(out, tuple) else out
        taps[name] = t.detach().clone() if KEEP[0] else t.detach()
    return hook


KEEP = [True]
from adaptpoint_amd import pointset as _ps
_count = {"fps": 0, "bq": 0, "gm": 0}


def _wrap(name, key):
    orig = getattr(_ps, name)

    def f(*a, **k):
        out = orig(*a, **k)
        tag = f"grouper-internal {key}{_count[key] % 4 + 1}"
        _count[key] += 1
        taps[tag] = out.detach().clone() if KEEP[0] else out.detach()
        if key == "gm":
            pin = a[0]
            taps[tag + " input"] = pin.detach().clone() if KEEP[0] else pin.detach()
        return out
    setattr(_ps, name, f)


_wrap("furthest_point_sample", "fps")
_wrap("ball_query", "bq")
_wrap("group_max", "gm")
sa = G.predict_prob_layer
sa.embedding.register_forward_hook(tap("embedding"))
for i in range(4):
    sa.extract_feat_list[i].register_forward_hook(tap(f"extract{i + 1}"))
    sa.pointset_grouper_list[i].register_forward_hook(tap(f"grouper{i + 1}"))
    sa.decode_list[i].register_forward_hook(tap(f"decode{i + 1}"))
sa.head.register_forward_hook(tap("head"))
sa.localfeat_mask_selfattention.register_forward_hook(tap("mask attention"))


def forward(two_lanes):
    with torch.no_grad():
        lr = None
        if two_lanes:
            s = graphs.fork(graphs.LANE2, dev, pos, real['x'], label)
            with torch.cuda.stream(s):
                lr = real_loss_ahead(C, crit, real, label)
        _, gen = G(pos, noise)
        if two_lanes:
            graphs.join(s, lr)
        else:
            lr = real_loss_ahead(C, crit, real, label)
    return gen, lr


def momentum0():          # the same running statistics before every forward
    for m in G.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 0.0


momentum0()
ref_gen, ref_lr = forward(False)
ref = dict(taps)
torch.cuda.synchronize()
warm = torch.cuda.Stream()
warm.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(warm):
    forward(True)
torch.cuda.current_stream().wait_stream(warm)
torch.cuda.synchronize()
eager2 = dict(taps)
print("eager two lanes vs eager one lane:", {k: float((eager2[k] - ref[k]).abs().max()) for k in ref if float((eager2[k] - ref[k]).abs().max()) > 0})
KEEP[0] = False
if "--prime" in sys.argv:
    # does a trivial two-branch graph, replayed once, absorb whatever the first multi-branch replay sets up?
    from adaptpoint_amd.layers import furthest_point_sample
    xa = torch.rand(4, 1024, 3, device=dev)
    s2 = graphs.side_stream(graphs.LANE2, dev)
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        furthest_point_sample(xa, 64)
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    pg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(pg):
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            k1 = [furthest_point_sample(xa, 64) for _ in range(4)]
        k2 = [furthest_point_sample(xa, 64) for _ in range(4)]
        main.wait_stream(s2)
    pg.replay(); pg.replay()
    torch.cuda.synchronize()
    print("primed with a trivial two-branch graph")
g = graphs.new_graph()
with torch.cuda.graph(g):
    gen, lr = forward(True)
cap = dict(taps)
print("graph:", graphs.node_census(g))
for it in range(6):
    g.replay()
    torch.cuda.synchronize()
    diff = {k: float((cap[k] - ref[k]).abs().max()) for k in ref}
    diff["gen"] = float((gen - ref_gen).abs().max())
    diff["loss_real"] = float((lr - ref_lr).abs().max())
    print("replay", it, {k: f"{v:.2e}" for k, v in diff.items() if v > 0})

# ---- the same two pieces as two SINGLE-branch graphs replayed on two streams at once: is it the concurrency of the two
# kernel sets on the device, or the multi-branch graph?
if "--two-graphs" in sys.argv:
    def only_generator():
        with torch.no_grad():
            return G(pos, noise)[1]

    enc = C.encoder

    def b_pyramid():
        with torch.no_grad():
            return enc.index_pyramid(pos)

    def b_stages(upto):
        def f():
            with torch.no_grad():
                p0, f0 = pos, real['x']
                for i, stage in enumerate(enc.encoder):
                    if i > upto:
                        break
                    p0, f0 = stage[0]([p0, f0])
                return f0
        return f

    def b_full():
        with torch.no_grad():
            return real_loss_ahead(C, crit, real, label)

    KEEP[0] = False
    ga = graphs.new_graph()
    with torch.cuda.graph(ga):
        gen2 = only_generator()
    cap2 = dict(taps)
    sa_, sb_ = torch.cuda.Stream(), torch.cuda.Stream()
    from adaptpoint_amd import fused as _fused
    sa2 = enc.encoder[2][0]

    def b_partial(level):
        def f():
            with torch.no_grad():
                p0, f0 = enc.encoder[0][0]([pos, real['x']])
                p1, f1 = enc.encoder[1][0]([p0, f0])
                smp = _fused.sample_and_query(p1, 256, sa2.grouper.radius, 32, geo=False)
                if level >= 1:
                    sa2.index_for(smp, 512, 64)
                if level >= 2:
                    return sa2([p1, f1], sampling=smp)
                return smp
        return f

    def b_stage2_only():
        with torch.no_grad():
            p0, f0 = enc.encoder[0][0]([pos, real['x']])
            p1, f1 = enc.encoder[1][0]([p0, f0])
        torch.cuda.synchronize()

        def f():
            with torch.no_grad():
                return [sa2([p1, f1]) for _ in range(6)]
        return f

    cands = {"stem + stage 1 + stage-2 sampler / ball query": b_partial(0), "... + tile map / inverse map": b_partial(1),
             "... + wide block": b_partial(2), "stage 2 alone x6": b_stage2_only(), "... + stage 2": b_stages(2)}
    for name, f in cands.items():
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            f()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        gb = graphs.new_graph()
        with torch.cuda.graph(gb):
            keepb = f()
        first = {}
        for it in range(6):
            torch.cuda.synchronize()
            with torch.cuda.stream(sb_):
                gb.replay()
            with torch.cuda.stream(sa_):
                ga.replay()
            torch.cuda.synchronize()
            for k in ref:
                if float((cap2[k].float() - ref[k].float()).abs().max()) > 0:
                    first[k] = first.get(k, 0) + 1
        print(f"generator beside [{name}]: taps that differed, rounds of 6:", {k: v for k, v in first.items() if "grouper" in k or "extract" in k} or "none", flush=True)
    sys.exit(0)

    def only_real():
        with torch.no_grad():
            return real_loss_ahead(C, crit, real, label)

    KEEP[0] = False
    ga, gb = graphs.new_graph(), graphs.new_graph()
    with torch.cuda.graph(ga):
        gen2 = only_generator()
    cap2 = dict(taps)
    with torch.cuda.graph(gb):
        lr2 = only_real()
    sa_, sb_ = torch.cuda.Stream(), torch.cuda.Stream()
    for it in range(6):
        torch.cuda.synchronize()
        with torch.cuda.stream(sb_):
            gb.replay()
        with torch.cuda.stream(sa_):
            ga.replay()
        torch.cuda.synchronize()
        diff = {k: float((cap2[k] - ref[k]).abs().max()) for k in ref}
        diff["gen"] = float((gen2 - ref_gen).abs().max())
        diff["loss_real"] = float((lr2 - ref_lr).abs().max())
        print("two single-branch graphs at once, round", it, {k: f"{v:.2e}" for k, v in diff.items() if v > 0})
