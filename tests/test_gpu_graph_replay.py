"""GPU: whole training steps replayed from a hipGraph against the same steps run eagerly.

Round 2 measured the joint AdaptPoint step (10.6 ms) and the classifier step (3.1 ms) from captured graphs whose
PyTorch part -- losses, softmax, BatchNorm1d statistics, bias gradients, gradient-norm clipping, fused Adam -- had no
replay-vs-eager check, after PyTorch's cross-block reduction had been caught returning STALE values from the second
replay on inside another captured region (measured-and-rejected 10).  Here every step of a replayed
sequence sees a different batch, so a reduction that kept an earlier replay's value shows up as a wrong loss: the
losses of three consecutive replays and the weights after them must equal three eager steps from the same state.

Bars: the two runs execute the same kernels on the same data; what differs is the order of float atomics in
the fused blocks' backward passes (1e-6-level), which the discrete decisions downstream (arg-max, ReLU gates) and Adam's
first steps (update ~ lr * sign(gradient)) amplify on some weights -- measured between two EAGER runs as well.  Bars:
losses 1e-4 relative (a stale reduction shows up as an O(1) error there: every step has its own batch); BatchNorm
running statistics 2e-3 of their scale; the cosine between the two runs' total parameter updates >= 0.95 (or the eager-vs-eager
value minus 0.03).
"""
import copy
import gc
import os

import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu

STEPS = 3


def _height(pos):
    return pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]


def _no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def _snapshot(mods, opts):
    return ([copy.deepcopy(m.state_dict()) for m in mods],
            [{id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.state[p].items()}
              for g in o.param_groups for p in g['params'] if p in o.state} for o in opts])


def _restore(mods, opts, snap):
    """In place: a captured graph keeps pointing at the same parameter / buffer / optimizer-state tensors."""
    with torch.no_grad():
        for m, sd in zip(mods, snap[0]):
            own = m.state_dict()
            for k, v in sd.items():
                own[k].copy_(v)
        for o, st in zip(opts, snap[1]):
            for g in o.param_groups:
                for p in g['params']:
                    if p in o.state:
                        for k, v in st[id(p)].items():
                            if torch.is_tensor(v):
                                o.state[p][k].copy_(v)
            o.zero_grad(set_to_none=False)


def _update_cosine(after_a, after_b, before):
    """cosine between the two runs' total parameter updates (Adam moves EVERY element by ~lr per step, also those whose
    gradient is noise-level: element-wise comparisons of such weights are coin flips, their share of the whole update
    is small)"""
    ua = torch.cat([(after_a[k].double() - before[k].double()).flatten() for k in before if before[k].dtype.is_floating_point])
    ub = torch.cat([(after_b[k].double() - before[k].double()).flatten() for k in before if before[k].dtype.is_floating_point])
    return float((ua * ub).sum() / (ua.norm() * ub.norm()).clamp_min(1e-30))


def _compare(tag, eager_losses, replay_losses, mods, eager_weights, before, floor=None, floor_losses=None, buffer_bar=2e-3):
    """Losses of every step (each has its own batch: a reduction that kept an earlier replay's value is an O(1) error
    here); BatchNorm running statistics after the steps (reductions themselves); the parameters' total update.
    floor = the weights of a SECOND eager run from the same state: what two runs of the same kernels differ by."""
    for i, (a, b) in enumerate(zip(eager_losses, replay_losses)):
        for k in a:
            ea, rb = float(a[k]), float(b[k])
            tol = 1e-4 * max(1.0, abs(ea))
            if floor_losses is not None:        # later steps inherit the earlier updates' run-to-run differences
                tol = max(tol, 3.0 * abs(float(floor_losses[i][k]) - ea))
            assert np.isfinite(ea) and abs(ea - rb) <= tol and tol <= 2e-3 * max(1.0, abs(ea)), (tag, "step", i, k, ea, rb, tol)
    cos = []
    worst = [0.0]
    bad = []
    for j, (m, ref) in enumerate(zip(mods, eager_weights)):
        now = m.state_dict()
        params = {k for k, _ in m.named_parameters()}
        for k, q in now.items():
            if k in params:
                continue
            if not q.dtype.is_floating_point:
                assert torch.equal(q, ref[k]), (tag, k)
            else:
                # (2e-3 of the tensor's scale: after three updates at lr 2e-3 two eager runs differ by ~1e-4 of it)
                err, scale = float((q - ref[k]).abs().max()), float(ref[k].abs().max())
                worst[0] = max(worst[0], err / max(scale, 1e-30))
                if err > buffer_bar * scale + 1e-6:
                    bad.append((k, err / max(scale, 1e-30)))
        if not any(q.requires_grad for q in m.parameters()) or all(torch.equal(now[k], before[j][k]) for k in params):
            continue
        pb = {k: before[j][k] for k in params}
        c = _update_cosine(now, ref, pb)
        need = 0.95 if floor is None else min(0.95, _update_cosine(floor[j], ref, pb) - 0.03)
        cos.append(round(c, 4))
        assert c >= need, (tag, j, c, need)
    assert not bad, (tag, "buffers beyond the bar (name, deviation / scale):", bad)
    print(f"{tag}: replayed == eager over {STEPS} steps; losses",
          [{k: round(float(v), 6) for k, v in d.items()} for d in replay_losses], "update cosines", cos,
          "worst buffer deviation / scale %.2e" % worst[0])


@pytest.mark.parametrize("overlap", [False, True])
def test_joint_step_replayed_from_a_hipgraph_equals_eager(dev, overlap):
    """One `train_gan` iteration (adaptpoint_amd.gan.GanStep, device-side draws handed in, capturable fused Adam).
    overlap: the step with its two side branches on their own streams (GanStep(overlap=True): parallel branches of the
    captured graph) -- run eagerly AND replayed -- against the single-stream eager step."""
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, Noise, draw_noise_on
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.gan import GanStep
    from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy, fill_parameters_by_name
    B, N = int(os.environ.get("APN_REPLAY_BATCH", 4)), 1024          # (the bench's batch of 32 by hand: timing-dependent hazards)
    G = fill_parameters_by_name(AdaptPointAugmentor(fused=True)).to(dev)
    D = _no_dropout(fill_parameters_by_name(PointDiscriminator1(num_classes=15, fused=True))).to(dev)
    C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev)
    step = GanStep(G, D, C, SmoothCrossEntropy(0.3), capturable=True)
    torch.manual_seed(5)
    batches = []
    for i in range(STEPS + 1):
        pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=700 + i)).to(dev)
        batches.append((torch.cat([pos, _height(pos)], -1), torch.randint(0, 15, (B,), device=dev),
                        draw_noise_on(dev, B, N, G.num_anchor)))
    points, label = batches[0][0].clone(), batches[0][1].clone()
    noise = Noise(*[t.clone() for t in (batches[0][2].keep, batches[0][2].axes, batches[0][2].kernel_axes,
                                        batches[0][2].gumbel_expo)])

    def load(i):
        points.copy_(batches[i][0])
        label.copy_(batches[i][1])
        for dst, src in zip((noise.keep, noise.axes, noise.kernel_axes, noise.gumbel_expo),
                            (batches[i][2].keep, batches[i][2].axes, batches[i][2].kernel_axes, batches[i][2].gumbel_expo)):
            dst.copy_(src)

    keys = ("g_loss_raw", "feedback_loss", "g_loss", "d_loss")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up: allocator pools, optimizer state, lazy init
        for _ in range(2):
            load(STEPS)
            step(points, label, noise=noise)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    mods, opts = (G, D, C), (step.opt_g, step.opt_d)
    snap = _snapshot(mods, opts)
    eager = []
    def eager_run():
        out = []
        for i in range(STEPS):
            load(i)
            res = step(points, label, noise=noise)
            out.append({k: res[k].item() for k in keys})
        return out, [copy.deepcopy(m.state_dict()) for m in mods]
    eager, eager_w = eager_run()
    _restore(mods, opts, snap)
    eager2, eager_w2 = eager_run()          # the same three steps again: the noise floor of the weights after Adam
    _restore(mods, opts, snap)
    if overlap:
        step.overlap = frozenset(os.environ["APN_OVERLAP_PARTS"].split(",")) if os.environ.get("APN_OVERLAP_PARTS") else True
        for _ in range(2):                               # the side streams' allocator pools
            load(STEPS)
            step(points, label, noise=noise)
        torch.cuda.synchronize()
        _restore(mods, opts, snap)
        eager3, eager_w3 = eager_run()
        _compare("joint step, side streams, eager", eager, eager3, mods, eager_w, snap[0], floor=eager_w2, floor_losses=eager2)
        _restore(mods, opts, snap)
    gc.collect()
    from adaptpoint_amd import graphs
    load(STEPS)
    graph, captured, census = graphs.capture(lambda: step(points, label, noise=noise), what="the joint step's graph",
                                             leaves=[q for m in mods for q in m.parameters()])
    print("joint step graph:", census)
    _restore(mods, opts, snap)                           # (capture runs nothing; make the state explicit anyway)
    replayed = []
    for i in range(STEPS):
        load(i)
        graph.replay()
        torch.cuda.synchronize()
        replayed.append({k: captured[k].item() for k in keys})
    assert eager2 is not None
    _compare("joint step" + (", side streams" if overlap else ""), eager, replayed, mods, eager_w, snap[0],
             floor=eager_w2, floor_losses=eager2)


def test_classifier_step_replayed_from_a_hipgraph_equals_eager(dev, monkeypatch):
    """One `train_one_epoch` iteration (adaptpoint_amd.gan.ClassifierStep: resampler, fused PointNeXt-S, SmoothCE,
    gradient-norm clipping, capturable fused AdamW).  With fused.DETERMINISTIC the step has no order-dependent sum
    left (stage 1: fixed-point sums; stages 2-4: the width-generic family; the per-point layers: fixed-order folds), so
    two eager runs from the same state agree to the last bit -- and so must the replays (Adam turns a last-bit
    difference of a near-zero gradient into a full step: without this the comparison needs a run-to-run floor)."""
    from adaptpoint_amd import fused
    from adaptpoint_amd.gan import ClassifierStep
    monkeypatch.setattr(fused, "DETERMINISTIC", True)
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    B, N = 4, 2048
    C = _no_dropout(fill_parameters_by_name(PointNextSClassifier(fused=True))).to(dev)
    opt = torch.optim.AdamW(C.parameters(), lr=2e-3, weight_decay=0.05, capturable=True, fused=True)
    step = ClassifierStep(C, optimizer=opt)
    choice = torch.from_numpy(np.random.RandomState(3).choice(1200, 1024, False).astype(np.int32)).to(dev)
    torch.manual_seed(6)
    batches = []
    for i in range(STEPS + 1):
        pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=800 + i)).to(dev)
        batches.append((torch.cat([pos, _height(pos)], -1), torch.randint(0, 15, (B,), device=dev)))
    points, target = batches[0][0].clone(), batches[0][1].clone()

    def load(i):
        points.copy_(batches[i][0])
        target.copy_(batches[i][1])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            load(STEPS)
            step(points, target, choice=choice)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    snap = _snapshot((C,), (opt,))
    def eager_run():
        out = []
        for i in range(STEPS):
            load(i)
            out.append({"loss": step(points, target, choice=choice)[1].item()})
        return out, [copy.deepcopy(C.state_dict())]
    eager2, eager_w2 = eager_run()          # twice from the same state: the run-to-run floor
    _restore((C,), (opt,), snap)
    eager, _w = eager_run()
    # (no tensor of an eager step may stay referenced here: it would keep that step's autograd graph alive, whose
    # AccumulateGrad nodes belong to the default stream -- the backward under capture would then synchronise with
    # the default stream and the capture ends in a crash of the HIP runtime)
    eager_w = _w
    _restore((C,), (opt,), snap)
    gc.collect()
    from adaptpoint_amd import graphs
    load(STEPS)
    graph, (_, cap_loss), census = graphs.capture(lambda: step(points, target, choice=choice), leaves=list(C.parameters()),
                                                  what="the classifier step's graph")
    print("classifier step graph:", census)
    _restore((C,), (opt,), snap)
    replayed = []
    for i in range(STEPS):
        load(i)
        graph.replay()
        torch.cuda.synchronize()
        replayed.append({"loss": cap_loss.item()})
    exact = all(a["loss"] == b["loss"] for a, b in zip(eager, eager2)) and all(
        torch.equal(eager_w[0][k], eager_w2[0][k]) for k in eager_w[0])
    print("classifier step: two eager runs bit-identical:", exact)
    assert exact
    assert [d["loss"] for d in replayed] == [d["loss"] for d in eager]
    now = C.state_dict()
    assert all(torch.equal(now[k], eager_w[0][k]) for k in now)
    _compare("classifier step", eager, replayed, (C,), eager_w, snap[0], floor=eager_w2, floor_losses=eager2)


def test_memset_nodes_are_found_before_the_first_replay(dev):
    """The defect behind round 2's stale reductions, and its guard: a captured hipMemsetAsync is a MEMSET node, which
    this stack replays correctly once and with a garbage pattern afterwards (scripts/debug_graph_memset.py);
    `graphs.assert_replayable` finds it in the captured graph, `bench.py` / the step benches then run eagerly."""
    import ctypes
    from adaptpoint_amd import graphs
    hip = ctypes.CDLL("libamdhip64.so")
    buf = torch.full((256,), 7, dtype=torch.int32, device=dev)

    def with_memset():
        st = torch.cuda.current_stream().cuda_stream
        assert hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(1024), ctypes.c_void_p(st)) == 0
        buf.add_(1)

    def without():
        buf.zero_()
        buf.add_(1)
    for fn, has in ((with_memset, True), (without, False)):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = graphs.new_graph()
        with torch.cuda.graph(g):
            fn()
        census = graphs.node_census(g)
        print(fn.__name__, census)
        if has:
            assert census.get("memset", 0) == 1
            with pytest.raises(graphs.MemsetNodeInGraph):
                graphs.assert_replayable(g)
        else:
            assert census.get("memset", 0) == 0 and census.get("kernel", 0) >= 2
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            assert int(buf[0]) == 1


def test_capture_refuses_an_autograd_graph_kept_alive_instead_of_crashing(dev):
    """Round 3's first capture crash (a host segfault inside hipStreamEndCapture: gpurun_out/tnew.log, tc.log): a tensor
    with a grad_fn from an EARLIER eager step kept alive across the capture.  graphs.capture finds the live graph from
    the leaves' accumulator nodes and raises before capturing; with the tensor dropped the same capture goes through."""
    from adaptpoint_amd import graphs
    lin = torch.nn.Linear(64, 64).to(dev)
    x = torch.randn(32, 64, device=dev, requires_grad=True)

    def step():
        x.grad = None
        lin.zero_grad(set_to_none=True)
        loss = lin(x).square().sum()
        loss.backward()
        return loss
    kept = step()                                   # eager, on the default stream -- and its graph stays alive through `kept`
    torch.cuda.synchronize()
    leaves = [x] + list(lin.parameters())
    with pytest.raises(graphs.StaleAutogradGraph, match="earlier step"):
        graphs.capture(lambda: step().detach(), leaves=leaves, what="a step captured beside a kept loss")
    assert not torch.cuda.is_current_stream_capturing()
    kept = kept.detach()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g, out, census = graphs.capture(lambda: step().detach(), leaves=leaves, what="the same step")
    g.replay()
    torch.cuda.synchronize()
    assert census.get("memset", 0) == 0 and torch.isfinite(out) and abs(float(out) - float(kept)) <= 1e-3 * abs(float(kept))


def test_fork_refuses_the_lane_topology_that_crashed_capture(dev):
    """Round 3's second capture crash (gpurun_out/wg2.log): a lane forked again from the main stream after the main
    stream waited for an event recorded INSIDE that lane, in one capture.  graphs.fork raises LaneTopology at the second
    fork (a Python error before the runtime sees the topology); eagerly -- where the topology is harmless -- it runs;
    fork / join by wait_stream alone captures and replays."""
    from adaptpoint_amd import graphs
    a = torch.ones(1 << 16, device=dev)
    b = torch.zeros(1 << 16, device=dev)

    def lanes(mid_wait):
        s = graphs.fork("test-lane", dev, a)
        with torch.cuda.stream(s):
            b.add_(a)
            ev = graphs.ready_event()
            b.add_(a)
        if mid_wait:
            graphs.wait_ready(ev, b)
        graphs.join(s, b)                            # (everything joined: the capture below can end cleanly)
        s = graphs.fork("test-lane", dev, a)         # <- refused under capture after a mid-lane wait
        with torch.cuda.stream(s):
            b.add_(a)
        graphs.join(s, b)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        lanes(True)                                  # eager: allowed
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert float(b[0]) == 3.0
    with pytest.raises(graphs.LaneTopology, match="forked again"):
        graphs.capture(lambda: lanes(True), what="a lane re-forked after a mid-lane wait")
    assert not torch.cuda.is_current_stream_capturing()
    torch.cuda.synchronize()
    b.zero_()
    g, _, census = graphs.capture(lambda: lanes(False), what="fork / join by wait_stream only")
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert float(b[0]) == 6.0 and census["kernel"] == 3
