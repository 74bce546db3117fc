"""GPU: operators BESIDE each other.  Every isolated parity test launches one kernel stream on an idle device; the
benches and the two-lane training step do not.  Here two captured graphs are replayed at the same time on two streams
-- the index stages of the classifier and of the imitator on one, feature kernels (the classifier's blocks, the fused
set-abstraction step) on the other -- and the index results must equal the ones formed alone, bit for bit, round after
round.  (This is the situation in which the LDS-atomic FPS step returned wrong picks for ~2 % of the clouds while
passing every isolated test: DESIGN.md section 4c.)"""
import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu
ROUNDS = 8


def _replay_beside(ga, gb, rounds, check):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for it in range(rounds):
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            gb.replay()
        with torch.cuda.stream(sa):
            ga.replay()
        torch.cuda.synchronize()
        check(it)


def _capture(fn):
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


@pytest.mark.parametrize("partner", ["classifier blocks", "fused set-abstraction steps"])
def test_index_stages_beside_feature_kernels_are_bit_exact(dev, partner):
    from adaptpoint_amd.layers import ball_query, furthest_point_sample
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    B = 32
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=900)).to(dev)
    pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
    C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
    enc = C.encoder

    def index_work():
        out = []
        for smp in enc.index_pyramid(pos):                         # the classifier's four index stages (sampler entry)
            if smp is not None:
                out += [smp.fidx, smp.new_p, smp.idx]
        xyz = pos                                                  # the imitator's grouper chain (drop-in entries)
        for r in (0.1, 0.2, 0.4, 0.8):
            fidx = furthest_point_sample(xyz, xyz.shape[1] // 2)
            new = torch.gather(xyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3))
            out += [fidx, ball_query(r, 24, xyz, new)]
            xyz = new
        return out

    if partner == "classifier blocks":
        def feature_work():
            keep = []
            for _ in range(3):
                p0, f0 = pos, pts
                for stage in enc.encoder:
                    p0, f0 = stage[0]([p0, f0])
                keep.append(f0)
            return keep
    else:
        from adaptpoint_amd.set_abstraction import SetAbstraction
        torch.manual_seed(0)
        sa = SetAbstraction(32, 64, layers=2, stride=2, fused=True,
                            group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                            norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                            use_res=True).to(dev).train()
        f = torch.randn(B, 32, 1024, device=dev)
        smp = sa.sample(pos)

        def feature_work():
            keep = []
            for _ in range(12):
                fi = f.clone().requires_grad_(True)
                _, out = sa([pos, fi], sampling=smp)
                out.sum().backward()
                keep.append(fi.grad)
            return keep

    with torch.no_grad() if partner == "classifier blocks" else torch.enable_grad():
        with torch.no_grad():
            ref = [t.clone() for t in index_work()]
        torch.cuda.synchronize()
        with torch.no_grad():
            ga, got = _capture(index_work)
        gb, _keep = _capture(feature_work)

    # the partner's own results alone (one replay on the idle device): the eval-mode classifier blocks are deterministic
    # kernels (fixed-order sums), so beside the index stream they must give the same bits too
    gb.replay()
    torch.cuda.synchronize()
    feat_ref = [t.clone() for t in _keep] if partner == "classifier blocks" else None
    bad, bad_feat = [], []

    def check(it):
        for k, (a, b) in enumerate(zip(got, ref)):
            if not torch.equal(a, b):
                bad.append((it, k, int((a != b).flatten(1).any(1).sum())))
        if feat_ref is not None:
            for k, (a, b) in enumerate(zip(_keep, feat_ref)):
                if not torch.equal(a, b):
                    bad_feat.append((it, k, float((a - b).abs().max())))

    _replay_beside(ga, gb, ROUNDS, check)
    assert not bad, ("index results formed beside the feature kernels differ from the ones formed alone "
                     "(round, tensor, clouds):", bad[:10])
    assert not bad_feat, ("the classifier's features formed beside the index kernels differ from the ones formed alone "
                          "(round, tensor, max deviation):", bad_feat[:10])


def test_feature_kernels_beside_other_feature_kernels_are_bit_exact(dev):
    """MFMA kernels beside MFMA kernels: the eval-mode classifier blocks (deterministic kernels: fixed-order sums) replayed
    on one stream while the fused set-abstraction training step (forward + backward, MFMA-heavy) replays on another must
    give the bits they give alone -- the packed-FP32 arithmetic inside these kernels was never seen to fail, and this is
    where it would show."""
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    from adaptpoint_amd.set_abstraction import SetAbstraction
    B = 32
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=901)).to(dev)
    pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
    C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
    pyr = C.encoder.index_pyramid(pos)

    def blocks():
        keep = []
        with torch.no_grad():
            for _ in range(3):
                p0, f0 = pos, pts
                for i, stage in enumerate(C.encoder.encoder):
                    smp = pyr[i]
                    p0, f0 = stage[0]([p0, f0], sampling=smp) if smp is not None else stage[0]([p0, f0])
                    keep.append(f0)
        return keep

    torch.manual_seed(0)
    sa = SetAbstraction(32, 64, layers=2, stride=2, fused=True,
                        group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                        use_res=True).to(dev).train()
    f = torch.randn(B, 32, 1024, device=dev)
    smp1 = sa.sample(pos)

    def steps():
        keep = []
        for _ in range(12):
            fi = f.clone().requires_grad_(True)
            _, out = sa([pos, fi], sampling=smp1)
            out.sum().backward()
            keep.append(fi.grad)
        return keep

    ga, got = _capture(blocks)
    gb, _keep = _capture(steps)
    ga.replay()
    torch.cuda.synchronize()
    ref = [t.clone() for t in got]
    bad = []

    def check(it):
        for k, (a, b) in enumerate(zip(got, ref)):
            if not torch.equal(a, b):
                bad.append((it, k, float((a - b).abs().max())))

    _replay_beside(ga, gb, ROUNDS, check)
    assert not bad, ("classifier features formed beside the fused training step differ from the ones formed alone "
                     "(round, tensor, max deviation):", bad[:10])


def test_generator_forward_beside_the_fused_training_step_is_bit_exact(dev):
    """The AdaptPoint generator's forward pass (per-point contraction kernels with their explicitly two-wide arithmetic,
    grouper, attention, deformation kernels; training-mode BatchNorm from fixed-order folds: deterministic) replayed beside
    the fused set-abstraction training step on another stream: the augmented clouds and the imitator's outputs must be the
    bits they are alone."""
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise_on
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    from adaptpoint_amd.set_abstraction import SetAbstraction
    B = 32
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=902)).to(dev)
    G = fill_parameters_by_name(AdaptPointAugmentor(fused=True)).to(dev).train()
    noise = draw_noise_on(dev, B, 1024, G.num_anchor)
    taps = {}
    sac = G.predict_prob_layer
    hooks = [sac.embedding.register_forward_hook(lambda m, i, o: taps.__setitem__("embedding", o.detach())),
             sac.head.register_forward_hook(lambda m, i, o: taps.__setitem__("head", o.detach()))]
    for i in range(4):
        hooks.append(sac.decode_list[i].register_forward_hook(lambda m, i_, o, k=i: taps.__setitem__(f"decode{k}", o.detach())))

    def generator():
        with torch.no_grad():
            out = G(pos, noise)[1]
        return [out] + [taps[k] for k in sorted(taps)]

    torch.manual_seed(0)
    sa = SetAbstraction(32, 64, layers=2, stride=2, fused=True,
                        group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                        use_res=True).to(dev).train()
    f = torch.randn(B, 32, 1024, device=dev)
    smp1 = sa.sample(pos)

    def steps():
        keep = []
        for _ in range(16):
            fi = f.clone().requires_grad_(True)
            _, out = sa([pos, fi], sampling=smp1)
            out.sum().backward()
            keep.append(fi.grad)
        return keep

    ga, got = _capture(generator)
    gb, _keep = _capture(steps)
    for h in hooks:
        h.remove()
    ga.replay()
    torch.cuda.synchronize()
    ref = [t.clone() for t in got]
    bad = []

    def check(it):
        for k, (a, b) in enumerate(zip(got, ref)):
            if not torch.equal(a, b):
                bad.append((it, k, float((a - b).abs().max())))

    _replay_beside(ga, gb, ROUNDS, check)
    assert not bad, ("the generator's results formed beside the fused training step differ from the ones formed alone "
                     "(round, tensor, max deviation):", bad[:10])


def test_classifier_training_pass_beside_the_fused_training_step_is_bit_exact(dev, monkeypatch):
    """Forward AND backward of the PointNeXt-S classifier in training mode with `fused.DETERMINISTIC` (no order-dependent
    sum anywhere: stage 1 fixed-point, stages 2-4 the width-generic family, per-point layers fixed-order folds) beside
    the fused set-abstraction training step on another stream: logits and every parameter gradient equal the bits they
    have alone."""
    from adaptpoint_amd import fused
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    from adaptpoint_amd.set_abstraction import SetAbstraction
    monkeypatch.setattr(fused, "DETERMINISTIC", True)
    B = 16
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=903)).to(dev)
    pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
    gt = (torch.arange(B, device=dev) % 15)
    C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).train()
    for m in C.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    params = [q for q in C.parameters() if q.requires_grad]

    def train_pass():
        for q in params:
            q.grad = None
        logits, loss = C.get_logits_loss({'pos': pos, 'x': pts}, gt)
        loss.backward()
        return [logits.detach()] + [q.grad for q in params]

    torch.manual_seed(0)
    sa = SetAbstraction(32, 64, layers=2, stride=2, fused=True,
                        group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                        use_res=True).to(dev).train()
    pos32 = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=904)).to(dev)
    f = torch.randn(32, 32, 1024, device=dev)
    smp1 = sa.sample(pos32)

    def steps():
        keep = []
        for _ in range(24):
            fi = f.clone().requires_grad_(True)
            _, out = sa([pos32, fi], sampling=smp1)
            out.sum().backward()
            keep.append(fi.grad)
        return keep

    ga, got = _capture(train_pass)
    gb, _keep = _capture(steps)
    ga.replay()
    torch.cuda.synchronize()
    ref = [t.clone() for t in got]
    ga.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, ref)), "the pass is not bit-reproducible even alone"
    bad = []

    def check(it):
        for k, (a, b) in enumerate(zip(got, ref)):
            if not torch.equal(a, b):
                bad.append((it, k, float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))))

    _replay_beside(ga, gb, ROUNDS, check)
    assert not bad, ("the classifier's training pass formed beside the fused training step differs from the one formed "
                     "alone (round, tensor, relative deviation):", bad[:10])


def test_discriminator_training_pass_beside_the_fused_training_step_is_bit_exact(dev):
    """The discriminator's forward + backward (batched spectral normalisation, per-point contraction layers, the pooled
    last layer with its ballot-compaction gradient: all fixed-order) beside the fused set-abstraction training step."""
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    from adaptpoint_amd.set_abstraction import SetAbstraction
    B = 32
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=905)).to(dev)
    D = fill_parameters_by_name(PointDiscriminator1(num_classes=15, fused=True)).to(dev).eval()   # (eval: no power iteration, no dropout: the same state every replay)
    params = [q for q in D.parameters() if q.requires_grad]

    def d_pass():
        for q in params:
            q.grad = None
        x = pos.clone().requires_grad_(True)
        y = D(x)
        y.sum().backward()
        return [y.detach(), x.grad] + [q.grad for q in params]

    torch.manual_seed(0)
    sa = SetAbstraction(32, 64, layers=2, stride=2, fused=True,
                        group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                        use_res=True).to(dev).train()
    f = torch.randn(B, 32, 1024, device=dev)
    smp1 = sa.sample(pos)

    def steps():
        keep = []
        for _ in range(12):
            fi = f.clone().requires_grad_(True)
            _, out = sa([pos, fi], sampling=smp1)
            out.sum().backward()
            keep.append(fi.grad)
        return keep

    ga, got = _capture(d_pass)
    gb, _keep = _capture(steps)
    ga.replay()
    torch.cuda.synchronize()
    ref = [t.clone() for t in got]
    ga.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, ref)), "the pass is not bit-reproducible even alone"
    bad = []

    def check(it):
        for k, (a, b) in enumerate(zip(got, ref)):
            if not torch.equal(a, b):
                bad.append((it, k, float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))))

    _replay_beside(ga, gb, ROUNDS, check)
    assert not bad, ("the discriminator's pass formed beside the fused training step differs from the one formed alone "
                     "(round, tensor, relative deviation):", bad[:10])


def test_fused_training_step_beside_the_classifier_blocks_is_bit_exact(dev, monkeypatch):
    """The headline block itself (register-resident kernels, `fused.DETERMINISTIC`: every sum order-independent) as the
    one under test: its outputs and gradients beside the eval-mode classifier's blocks (width-generic MFMA kernels) on
    another stream equal the bits they have alone."""
    from adaptpoint_amd import fused
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    from adaptpoint_amd.set_abstraction import SetAbstraction
    monkeypatch.setattr(fused, "DETERMINISTIC", True)
    B = 32
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=906)).to(dev)
    pts = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
    C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
    torch.manual_seed(0)
    sa = SetAbstraction(32, 64, layers=2, stride=2, fused=True,
                        group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                        norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                        use_res=True).to(dev).train()
    params = [q for q in sa.parameters() if q.requires_grad]
    f = torch.randn(B, 32, 1024, device=dev)
    smp1 = sa.sample(pos)

    def step():
        res = []
        for _ in range(4):
            for q in params:
                q.grad = None
            fi = f.clone().requires_grad_(True)
            _, out = sa([pos, fi], sampling=smp1)
            out.sum().backward()
            res += [out.detach(), fi.grad] + [q.grad for q in params]
        return res

    def blocks():
        keep = []
        with torch.no_grad():
            for _ in range(3):
                p0, f0 = pos, pts
                for stage in C.encoder.encoder:
                    p0, f0 = stage[0]([p0, f0])
                keep.append(f0)
        return keep

    ga, got = _capture(step)
    gb, _keep = _capture(blocks)
    ga.replay()
    torch.cuda.synchronize()
    ref = [t.clone() for t in got]
    ga.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, ref)), "the step is not bit-reproducible even alone"
    bad = []

    def check(it):
        for k, (a, b) in enumerate(zip(got, ref)):
            if not torch.equal(a, b):
                bad.append((it, k, float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))))

    _replay_beside(ga, gb, ROUNDS, check)
    assert not bad, ("the fused step formed beside the classifier's blocks differs from the one formed alone "
                     "(round, tensor, relative deviation):", bad[:10])
