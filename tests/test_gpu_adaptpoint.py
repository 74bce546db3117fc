"""GPU: the AdaptPoint half of the hot path (BASELINE configs[3], and configs[2]'s training
step) through the extension and the fused operators, against the goldens the REFERENCE's modules
and trainer statements produced (tests/golden/make_golden.py, G9-G13).

Bars.  Index / copy operators: bit-exact.  Networks: the goldens are torch-CPU float32; on the GPU
the same float32 layers run in MIOpen / rocBLAS with other summation orders, through ~40 layers
with training-mode BatchNorm over 2 clouds -- the observed agreement is printed by every test and
the bars below sit a small factor above it.  The random draws of the generator are replayed
(`draw_noise` after the golden's seed), and every mask decision of the goldens clears a margin
(make_golden.MASK_MARGIN), so the discrete part of the output must agree exactly.
"""
import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu


def rel(a, ref):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(1e-12, np.abs(ref).max()))


def height_channel(pos):
    return pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


# ------------------------------------------------------------------ resampler (a22 / 8f row 3)
def test_resampler_matches_reference_statements(dev, golden_ap):
    from adaptpoint_amd.gan import resample
    pos = torch.from_numpy(GI.unit_sphere_cloud(2, 2048, seed=131))
    points = torch.cat([pos, height_channel(pos)], -1).to(dev)
    p1, x1 = resample(points, 1024, 4, golden_ap["g13_choice"])
    assert np.array_equal(p1.cpu().numpy(), golden_ap["g13_pos"])
    assert np.array_equal(x1.cpu().numpy(), golden_ap["g13_x"])


@pytest.mark.parametrize("B,N,C,cx", [(32, 2048, 4, 4), (3, 1500, 7, 5), (1, 1201, 3, 3), (2, 5000, 8, 0)])
def test_resampler_vs_oracle(dev, oracle, B, N, C, cx):
    from adaptpoint_amd import ops
    pts = GI.seeded_normal((B, N, C), seed=7 + B)
    fidx = oracle.furthest_point_sampling(np.ascontiguousarray(pts[:, :, :3]), min(1200, N))
    choice = np.random.RandomState(B).choice(fidx.shape[1], min(1024, fidx.shape[1]), False).astype(np.int32)
    s = choice.size
    pos = torch.full((B, s, 3), 7.0, device=dev)
    x = torch.full((B, max(cx, 1), s), 7.0, device=dev)
    ops.resample_points_wrapper(B, N, C, fidx.shape[1], s, cx, torch.from_numpy(pts).to(dev),
                                torch.from_numpy(fidx).to(dev), torch.from_numpy(choice).to(dev), pos, x)
    po, xo = oracle.resample_points(pts, fidx, choice, cx)
    assert np.array_equal(pos.cpu().numpy(), po)
    if cx:
        assert np.array_equal(x.cpu().numpy(), xo)
    else:
        assert (x == 7.0).all()


# ------------------------------------------------------------------ a12 / a14
def test_three_interpolation_and_knn_grouper(dev, golden_ap, oracle):
    from adaptpoint_amd.layers import KnnGrouper, three_interpolation
    xyz = GI.config1_xyz()
    known = GI.take_points(xyz, oracle.furthest_point_sampling(xyz, 256))
    feat = torch.from_numpy(GI.seeded_normal((2, 48, 256), seed=121)).to(dev).requires_grad_(True)
    up = three_interpolation(torch.from_numpy(xyz).to(dev), torch.from_numpy(known).to(dev), feat)
    (up * torch.from_numpy(GI.seeded_normal(tuple(up.shape), seed=122)).to(dev)).sum().backward()
    np.testing.assert_allclose(up.detach().cpu().numpy(), golden_ap["g12_interp"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(feat.grad.cpu().numpy(), golden_ap["g12_interp_grad"], rtol=1e-5, atol=1e-5)
    feats = torch.from_numpy(GI.seeded_normal((2, 16, 1024), seed=123)).to(dev)
    dp, fj = KnnGrouper(8, normalize_dp=True)(torch.from_numpy(known[:, :64]).to(dev),
                                              torch.from_numpy(xyz).to(dev), feats)
    # cdist on the GPU may order two near-equal distances differently: compare as neighbour SETS
    ref_fj = golden_ap["g12_knn_fj"]
    got = np.sort(fj.cpu().numpy(), axis=-1)
    same = (got == np.sort(ref_fj, axis=-1)).all(axis=(1, 3))          # per (cloud, query)
    assert same.mean() >= 0.98, same.mean()
    assert rel(dp.abs().amax(dim=(1, 2, 3)), np.abs(golden_ap["g12_knn_dp"]).max(axis=(1, 2, 3))) < 1e-5


# ------------------------------------------------------------------ a19: the generator
def _generator(dev, fused):
    from adaptpoint_amd.augmentor import AdaptPointAugmentor
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    return fill_parameters_by_name(AdaptPointAugmentor(fused=fused)).to(dev).train()


@pytest.mark.parametrize("fused", [True, False])
def test_generator_matches_reference(dev, golden_ap, fused):
    from adaptpoint_amd.augmentor import draw_noise
    g = _generator(dev, fused)
    x = torch.from_numpy(GI.unit_sphere_cloud(2, 512, seed=91)).to(dev)
    torch.manual_seed(int(golden_ap["g9_seed"]))
    noise = draw_noise(2, 512, 4, with_gumbel=True)           # the golden's draws, replayed from the CPU generator
    logits = {}
    g.predict_prob_layer.fuse_masking.register_forward_hook(
        lambda m, i, o: logits.__setitem__("v", o.detach().permute(0, 2, 1)))
    _, out = g(x, noise)
    (out * torch.from_numpy(GI.seeded_normal((2, 512, 3), seed=92)).to(dev)).sum().backward()
    ref = golden_ap["g9_gen_out"]
    e_log = float(np.abs(logits["v"].cpu().numpy() - golden_ap["g9_mask_logits"]).max())
    sac = g.predict_prob_layer
    errs = dict(out=rel(out, ref), logits_abs=e_log,
                embed=rel(sac.embedding.net[0].weight.grad, golden_ap["g9_grad_embed_w"]),
                head=rel(sac.head.prob_head[0].weight.grad, golden_ap["g9_grad_prob_head_w"]),
                mask=rel(sac.extract_local_feat_masking[0].weight.grad, golden_ap["g9_grad_mask_local_w"]))
    print("generator vs reference golden (fused=%s):" % fused, {k: "%.2e" % v for k, v in errs.items()})
    # measured on MI355X: out 8e-7, mask logits 2e-4 (of a 5e-3 margin), head 7e-6, mask branch 5e-4,
    # embedding (gradient through all ~40 layers, train-mode BatchNorm over 2 clouds) 4e-3
    assert e_log < 1e-3                                           # a fifth of the margin every decision clears
    assert np.array_equal(out.detach().abs().sum(-1).cpu().numpy() == 0, np.abs(ref).sum(-1) == 0)
    assert errs["out"] < 1e-5
    assert errs["head"] < 2e-4 and errs["mask"] < 5e-3 and errs["embed"] < 2e-2


def test_generator_own_draws_are_valid(dev):
    """Without a `noise` argument the module draws like the reference (Gumbel on the device
    generator, the rest on the CPU generator): output inside the unit sphere, masked points at
    the origin, both outcomes of the mask present."""
    g = _generator(dev, True)
    x = torch.from_numpy(GI.unit_sphere_cloud(4, 1024, seed=5)).to(dev)
    src, out = g(x)
    assert src.shape == out.shape == (4, 1024, 3) and torch.isfinite(out).all()
    norms = out.norm(dim=-1)
    assert float(norms.max()) < 1.0
    masked = (norms == 0).float().mean().item()
    assert 0.02 < masked < 0.98


# ------------------------------------------------------------------ a20: the discriminator
def test_discriminator_matches_reference(dev, golden_ap):
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    d = no_dropout(fill_parameters_by_name(PointDiscriminator1(num_classes=15))).to(dev)
    x = torch.from_numpy(GI.unit_sphere_cloud(2, 512, seed=101)).to(dev)
    d.eval()
    with torch.no_grad():
        e0 = rel(d(x), golden_ap["g10_dis_eval"])
    d.train()
    xg = x.clone().requires_grad_(True)
    out = d(xg)
    (out * torch.tensor([[1.0], [-2.0]], device=dev)).sum().backward()
    errs = dict(eval=e0, train=rel(out, golden_ap["g10_dis_train"]), grad_x=rel(xg.grad, golden_ap["g10_grad_x"]),
                conv0=rel(d.sa1.mlp_convs[0].parametrizations.weight.original.grad, golden_ap["g10_grad_conv0"]),
                u=rel(d.fc1.parametrizations.weight[0]._u, golden_ap["g10_u_fc1"]))
    print("discriminator vs reference golden:", {k: "%.2e" % v for k, v in errs.items()})
    assert errs["eval"] < 1e-5 and errs["train"] < 1e-5 and errs["u"] < 1e-5      # measured: 0, 0, 3e-7
    assert errs["grad_x"] < 1e-4 and errs["conv0"] < 1e-4                         # measured: 9e-7, 8e-7


# ------------------------------------------------------------------ a21: the joint step
@pytest.mark.parametrize("fused", [True, False])
def test_gan_step_matches_reference_trainer(dev, golden_ap, fused):
    """One `train_gan` iteration at B=2: the gradient path classifier -> points -> generator
    (group / interpolate gradients w.r.t. coordinates) runs end to end on the GPU."""
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.gan import GanStep
    from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy, fill_parameters_by_name
    G = fill_parameters_by_name(AdaptPointAugmentor(fused=fused)).to(dev)
    D = no_dropout(fill_parameters_by_name(PointDiscriminator1(num_classes=15, fused=fused))).to(dev)
    C = fill_parameters_by_name(PointNextSClassifier(fused=fused)).to(dev)
    pos = torch.from_numpy(GI.unit_sphere_cloud(2, 512, seed=int(golden_ap["g11_pos_seed"])))
    points = torch.cat([pos, height_channel(pos)], -1).to(dev)
    step = GanStep(G, D, C, SmoothCrossEntropy(0.3))
    grads = {}
    G.predict_prob_layer.embedding.net[0].weight.register_hook(lambda g: grads.__setitem__("embed", g.clone()))
    G.predict_prob_layer.head.prob_head[0].weight.register_hook(lambda g: grads.__setitem__("head", g.clone()))
    D.fc3.parametrizations.weight.original.register_hook(lambda g: grads.__setitem__("fc3", g.clone()))
    torch.manual_seed(int(golden_ap["g11_seed"]))
    res = step(points, torch.tensor([3, 11], device=dev), noise=draw_noise(2, 512, 4, with_gumbel=True))
    ref = golden_ap["g11_gen"]
    got = np.array([res[k].item() for k in ("g_loss_raw", "feedback_loss", "g_loss", "d_loss")])
    errs = dict(gen=rel(res["gen"], ref), losses=float(np.abs(got / golden_ap["g11_losses"] - 1).max()),
                embed=rel(grads["embed"], golden_ap["g11_grad_embed_w"]),
                head=rel(grads["head"], golden_ap["g11_grad_prob_head_w"]),
                fc3=rel(grads["fc3"], golden_ap["g11_grad_fc3"]))
    print("train_gan step vs reference golden (fused=%s):" % fused, {k: "%.2e" % v for k, v in errs.items()})
    assert np.array_equal(res["gen"].abs().sum(-1).cpu().numpy() == 0, np.abs(ref).sum(-1) == 0)
    # The golden's input is the one of 289 candidates whose anchor head is furthest from a discrete tie (top-24
    # selection gap 2e-2, smallest lead of a max over the 24 neighbours 1e-5 of the feature scale; round 2's input had
    # a near-tie there and the head gradient TWO outcomes, 1e-5 or 2.8e-3).  Measured: fused kernels head 2.2e-4,
    # embedding (the deepest gradient: ~40 layers, batch-norm over 2 clouds) 2.0e-3; composed from PyTorch / MIOpen
    # fp32 layers head 5.5e-6, embedding 4.1e-2 (MIOpen's convolution gradients are the less accurate ones).
    assert errs["gen"] < 1e-5 and errs["losses"] < 1e-5 and errs["fc3"] < 1e-4
    assert errs["head"] < (5e-4 if fused else 1e-4) and errs["embed"] < (5e-3 if fused else 8e-2)
    # (round 4) every parameter's gradient of both steps against the reference trainer's, in relative L2
    import classifier_b8_checks as K
    eg, _ = K.gradient_errors(G, golden_ap, "g11_gen")
    ed, _ = K.gradient_errors(D, golden_ap, "g11_dis")
    print("G11 every parameter (fused=%s): generator median %.2e worst %s; discriminator worst %s"
          % (fused, float(np.median(list(eg.values()))), [("%.2e" % v, n) for v, n in K.worst(eg, 3)], [("%.2e" % v, n) for v, n in K.worst(ed, 2)]))
    # measured: fused kernels generator median 6.1e-4, worst 1.5e-3 (69 parameters with a real gradient, 14 analytic zeros);
    # composed from PyTorch / MIOpen layers median 7.3e-4, worst 3.0e-2 (MIOpen's convolution gradients); discriminator 1.6e-6
    assert max(eg.values()) < (5e-3 if fused else 8e-2), K.worst(eg)
    assert max(ed.values()) < 1e-4, K.worst(ed)
    # Adam's first step moves every weight by lr * sign(grad): EVERY weight whose reference gradient exceeds ten times the
    # gradient bar of its tensor (x the tensor's rms) must land exactly where the reference's did (round 3: "97 % of the
    # entries", which allowed 3 % arbitrarily far off)
    for after, name_after, name_grad, bar in (
            (G.predict_prob_layer.embedding.net[0].weight, "g11_embed_w_after", "g11_grad_embed_w", 5e-3 if fused else 8e-2),
            (D.fc3.parametrizations.weight.original, "g11_fc3_after", "g11_grad_fc3", 1e-4)):
        gref = golden_ap[name_grad]
        sure = np.abs(gref) > 10 * bar * np.sqrt(np.mean(gref.astype(np.float64) ** 2))
        assert sure.sum() >= 0.3 * sure.size or bar > 1e-2, (name_grad, int(sure.sum()), sure.size)
        off = np.abs(after.detach().cpu().numpy() - golden_ap[name_after])[sure]
        assert (off < 1e-5).all(), (name_after, int((off >= 1e-5).sum()), int(sure.sum()))


def test_gan_step_full_size_runs(dev):
    """BASELINE configs[3] at its own size (B=32, N=1024): finite losses, a moving generator."""
    from adaptpoint_amd.augmentor import AdaptPointAugmentor
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.gan import GanStep
    from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy
    torch.manual_seed(0)
    G, D = AdaptPointAugmentor().to(dev), PointDiscriminator1(num_classes=15).to(dev)
    C = PointNextSClassifier(fused=True).to(dev)
    pos = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=3))
    points = torch.cat([pos, height_channel(pos)], -1).to(dev)
    label = torch.arange(32, device=dev) % 15
    step = GanStep(G, D, C, SmoothCrossEntropy(0.3))
    w0 = G.predict_prob_layer.embedding.net[0].weight.detach().clone()
    for _ in range(2):
        res = step(points, label)
    for k in ("g_loss_raw", "feedback_loss", "g_loss", "d_loss"):
        assert torch.isfinite(res[k]), k
    assert not torch.equal(w0, G.predict_prob_layer.embedding.net[0].weight)
    assert res["gen"].shape == (32, 1024, 3) and float(res["gen"].norm(dim=-1).max()) < 1.0


# ------------------------------------------------------------------ a22: the classifier step
@pytest.mark.parametrize("fused", [True, False])
def test_classifier_step_matches_reference_trainer(dev, golden_ap, fused):
    from adaptpoint_amd.gan import ClassifierStep
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    pos = torch.from_numpy(GI.unit_sphere_cloud(2, 2048, seed=131))
    points = torch.cat([pos, height_channel(pos)], -1).to(dev)
    m = no_dropout(fill_parameters_by_name(PointNextSClassifier(fused=fused))).to(dev)
    logits, loss = ClassifierStep(m)(points, torch.tensor([5, 14], device=dev), choice=golden_ap["g13_choice"])
    errs = dict(logits=rel(logits, golden_ap["g13_logits"]), loss=abs(loss.item() / float(golden_ap["g13_loss"]) - 1),
                bn=rel(m.encoder.encoder[1][0].convs[0][1].running_mean, golden_ap["g13_bn1_mean_after"]))
    after = m.prediction.head[-1][0].weight.detach().cpu().numpy()
    errs["head_sign_agree"] = float((np.abs(after - golden_ap["g13_head_w_after"]) < 1e-4).mean())
    print("train_one_epoch step vs reference golden (fused=%s):" % fused, {k: "%.2e" % v for k, v in errs.items()})
    # measured: unfused logits 1e-4 / loss 3e-5; fused 2e-3 / 5e-4 (four fused stages' discontinuities at B=2)
    assert errs["logits"] < (1e-2 if fused else 1e-3) and errs["loss"] < (2e-3 if fused else 2e-4) and errs["bn"] < 1e-5
    # (the updated weights are held to the reference entry by entry at B = 8, where per-parameter gradients are pinned:
    # test_classifier_iteration_at_b8_matches_reference_trainer; G13 at B = 2 stores no gradient to say which signs are sure)
    assert errs["head_sign_agree"] > 0.95


# ------------------------------------------------------------------ 8f row 4: the segmentation decoder path
def test_feature_propagation_and_decoder_match_reference(dev, golden_ap, oracle):
    from adaptpoint_amd.pointnext import FeaturePropagation, PointNextDecoder, PointNextEncoderS, fill_parameters_by_name
    fp = fill_parameters_by_name(FeaturePropagation([64 + 32, 32, 32])).to(dev).train()
    p1 = GI.unit_sphere_cloud(2, 512, seed=141)
    p2 = GI.take_points(p1, oracle.furthest_point_sampling(p1, 128))
    f1 = torch.from_numpy(GI.seeded_normal((2, 32, 512), seed=142)).to(dev).requires_grad_(True)
    f2 = torch.from_numpy(GI.seeded_normal((2, 64, 128), seed=143)).to(dev).requires_grad_(True)
    out = fp([torch.from_numpy(p1).to(dev), f1], [torch.from_numpy(p2).to(dev), f2])
    (out * torch.from_numpy(GI.seeded_normal(tuple(out.shape), seed=144)).to(dev)).sum().backward()
    errs = dict(out=rel(out, golden_ap["g14_fp_out"]), f1=rel(f1.grad, golden_ap["g14_fp_grad_f1"]),
                f2=rel(f2.grad, golden_ap["g14_fp_grad_f2"]), w0=rel(fp.convs[0][0].weight.grad, golden_ap["g14_fp_grad_w0"]))
    print("FeaturePropagation vs reference golden:", {k: "%.2e" % v for k, v in errs.items()})
    assert all(v < 1e-4 for v in errs.values()), errs
    dec = fill_parameters_by_name(PointNextDecoder([32, 64, 128, 256, 512])).to(dev).train()
    pl = [GI.unit_sphere_cloud(2, 256, seed=146)]
    for m in (128, 64, 32, 16):
        pl.append(GI.take_points(pl[-1], oracle.furthest_point_sampling(pl[-1], m)))
    fl = [torch.from_numpy(GI.seeded_normal((2, c, n), seed=147 + i)).to(dev)
          for i, (c, n) in enumerate(zip((32, 64, 128, 256, 512), (256, 128, 64, 32, 16)))]
    od = dec([torch.from_numpy(q).to(dev) for q in pl], fl)
    assert rel(od, golden_ap["g14_dec_out"]) < 1e-3
    # encoder levels + decoder end to end at a segmentation-like size (shapes and finiteness)
    enc = PointNextEncoderS(in_channels=4, strides=(1, 4, 4, 4, 4), fused=True).to(dev).train()
    pos = torch.from_numpy(GI.unit_sphere_cloud(2, 4096, seed=148)).to(dev)
    x = torch.cat([pos, pos[:, :, 1:2]], -1).transpose(1, 2).contiguous()
    p, f = enc.forward_seg_feat(pos, x)
    seg = PointNextDecoder([q.shape[1] for q in f[1:]], in_channels=4).to(dev).train()
    o = seg(p[1:], f[1:])
    assert o.shape == (2, 32, 4096) and torch.isfinite(o).all()


def test_anchor_transforms_kernel_matches_composed_form(dev):
    """csrc/augment.hip (one launch each way) against the composed PyTorch form in float64
    (generator_component4_15.py:236-297): matrices, offsets and the gradient w.r.t. the imitator's numbers, with
    every switch combination of `keep` and `axes` present (a scale switched off is exactly 1, gradient 0)."""
    from adaptpoint_amd.augmentor import Noise, anchor_transforms, anchor_transforms_composed
    B, M = 64, 4
    g = torch.Generator().manual_seed(7)
    prob = (2.0 * torch.randn(B, M, 9, generator=g)).to(dev).requires_grad_(True)
    noise = Noise(keep=torch.bernoulli(torch.full((B, M, 3), 0.5), generator=g),
                  axes=torch.randint(0, 2, (B, M, 3), generator=g).int(), kernel_axes=torch.ones(B, 1, 3).int()).to(dev)
    g_lin = torch.randn(B, M, 3, 3, generator=g).to(dev)
    g_off = torch.randn(B, M, 3, generator=g).to(dev)
    p64 = prob.detach().double().requires_grad_(True)
    lin64, off64 = anchor_transforms_composed(p64, noise, 10, 3, 0.25)
    ((lin64 * g_lin.double()).sum() + (off64 * g_off.double()).sum()).backward()
    lin, off = anchor_transforms(prob, noise, 10, 3, 0.25)
    ((lin * g_lin).sum() + (off * g_off).sum()).backward()
    host = lambda t: t.detach().cpu().numpy()
    assert rel(lin, host(lin64)) < 1e-6 and rel(off, host(off64)) < 1e-6
    assert rel(prob.grad, host(p64.grad)) < 1e-5


@pytest.mark.parametrize("shape", [(4, 1024, 4), (3, 777, 3), (2, 2048, 8)], ids=lambda s: "x".join(map(str, s)))
def test_deformation_kernel_matches_composed_form(dev, shape):
    """csrc/augment.hip's kernel regression + blend + unit sphere + mask (one launch each way) against the composed
    PyTorch form in float64 (generator_component4_15.py:156-232, 313-327, 180): the augmented cloud and the gradients
    w.r.t. the per-anchor matrices, offsets and the mask."""
    from adaptpoint_amd.augmentor import deform_normalise_mask
    B, N, M = shape
    g = torch.Generator().manual_seed(N + M)
    x = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=5)).to(dev)
    anchors = x[:, torch.randperm(N, generator=g)[:M]].contiguous()
    lin = (torch.eye(3) + 0.3 * torch.randn(B, M, 3, 3, generator=g)).to(dev).requires_grad_(True)
    off = (0.2 * torch.randn(B, M, 3, generator=g)).to(dev).requires_grad_(True)
    axes = torch.randint(0, 2, (B, 1, 3), generator=g).int()
    axes[:, :, 0] |= (axes.sum(-1) == 0).int()                       # at least one axis (codes 1..7)
    axes = axes.to(dev)
    mask0 = (torch.rand(B, N, generator=g) < 0.8).float().to(dev).requires_grad_(True)
    gout = torch.randn(B, N, 3, generator=g).to(dev)
    lin64, off64, m64 = (t.detach().double().requires_grad_(True) for t in (lin, off, mask0))
    ref = deform_normalise_mask(x.double(), anchors.double(), lin64, off64, axes, m64, 0.5, fused=False)
    (ref * gout.double()).sum().backward()
    out = deform_normalise_mask(x, anchors, lin, off, axes, mask0, 0.5)
    (out * gout).sum().backward()
    host = lambda t: t.detach().cpu().numpy()
    assert rel(out, host(ref)) < 2e-6
    assert rel(lin.grad, host(lin64.grad)) < 2e-5 and rel(off.grad, host(off64.grad)) < 2e-5
    assert rel(mask0.grad, host(m64.grad)) < 2e-5


def test_gan_step_at_the_size_the_reference_trains_at(dev):
    """The joint step at B=32, N=2048 -- what the reference actually feeds `train_gan` (scanobjectnn.py:40: 2048 points;
    train_autoaug.py:136-148) -- on the fused kernels against the same step composed from the unfused operators
    (same weights, same random draws): finite, the generator moves, the two agree.  Bars a small factor above the
    measured differences (printed), which come from the fused blocks' split-bf16 contractions (5e-5 per block)."""
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise_on
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.gan import GanStep
    from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy, fill_parameters_by_name
    B, N = 32, 2048
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=9))
    points = torch.cat([pos, height_channel(pos)], -1).to(dev)
    label = torch.arange(B, device=dev) % 15
    torch.manual_seed(3)
    noise = draw_noise_on(dev, B, N, 4)
    out = {}
    for fused in (True, False):
        G = fill_parameters_by_name(AdaptPointAugmentor(fused=fused)).to(dev)
        D = no_dropout(fill_parameters_by_name(PointDiscriminator1(num_classes=15, fused=fused))).to(dev)
        C = fill_parameters_by_name(PointNextSClassifier(fused=fused)).to(dev)
        grads = {}
        G.predict_prob_layer.embedding.net[0].weight.register_hook(lambda g, d=grads: d.__setitem__("embed", g.clone()))
        G.predict_prob_layer.head.prob_head[0].weight.register_hook(lambda g, d=grads: d.__setitem__("head", g.clone()))
        D.fc3.parametrizations.weight.original.register_hook(lambda g, d=grads: d.__setitem__("fc3", g.clone()))
        w0 = G.predict_prob_layer.embedding.net[0].weight.detach().clone()
        res = GanStep(G, D, C, SmoothCrossEntropy(0.3))(points, label, noise=noise)
        torch.cuda.synchronize()
        for k in ("g_loss_raw", "feedback_loss", "g_loss", "d_loss"):
            assert torch.isfinite(res[k]), (fused, k)
        assert not torch.equal(w0, G.predict_prob_layer.embedding.net[0].weight)
        assert res["gen"].shape == (B, N, 3) and float(res["gen"].norm(dim=-1).max()) < 1.0
        out[fused] = (res, grads)
    (rf, gf), (ru, gu) = out[True], out[False]
    errs = dict(gen=rel(rf["gen"], ru["gen"].cpu().numpy()),
                losses=max(abs(rf[k].item() / ru[k].item() - 1) for k in ("g_loss_raw", "feedback_loss", "g_loss", "d_loss")),
                **{k: rel(gf[k], gu[k].cpu().numpy()) for k in ("embed", "head", "fc3")})
    print("joint step at B=32, N=2048, fused vs unfused:", {k: "%.2e" % v for k, v in errs.items()})
    # measured: gen 9e-7, losses 2e-6, fc3 9e-7; the generator's gradients 1.8e-2 (head) and 6.8e-2 (embedding) -- the
    # distance between the fused kernels and MIOpen's fp32 convolution gradients through ~40 layers (against the
    # reference's golden the fused path is the closer of the two: test_gan_step_matches_reference_trainer)
    assert errs["gen"] < 1e-5 and errs["losses"] < 1e-4 and errs["fc3"] < 1e-4
    assert errs["head"] < 6e-2 and errs["embed"] < 0.2


# Measured (round 4): unfused logits 1.2e-5 / loss 4e-7 / worst parameter gradient 3.0e-3 on one box and 1.1e-2 on another
# (MIOpen picks its convolution-gradient solvers per box); fused 2.1e-4 / 1.3e-5 / 1.4e-2 on both
G18_BARS = {False: dict(logits=1e-4, loss=1e-5, grads=3e-2), True: dict(logits=1e-3, loss=1e-4, grads=4e-2)}


@pytest.mark.parametrize("fused", [True, False])
def test_classifier_iteration_at_b8_matches_reference_trainer(dev, golden_b8, fused):
    """G18: one `train_one_epoch` iteration at B = 8, N = 2048 (resampler, forward, SmoothCE, backward, clip, AdamW) against
    the reference trainer's golden: logits, loss, every parameter's pre-clip gradient in relative L2, BatchNorm running
    means, and EVERY sampled weight whose reference gradient exceeds ten times the gradient bar takes the reference's step."""
    import classifier_b8_checks as K
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    bar = G18_BARS[fused]
    r = K.run_g18(fill_parameters_by_name(PointNextSClassifier(fused=fused)), dev, golden_b8, grad_bar=bar["grads"])
    print("G18 on the GPU (fused=%s):" % fused,
          {k: ("%.2e" % r[k] if isinstance(r[k], float) else r[k]) for k in ("logits", "loss", "bn", "steps_checked", "step_mismatch")},
          "worst parameter gradients:", [("%.2e" % v, n) for v, n in K.worst(r["grads"], 4)])
    assert r["logits"] < bar["logits"] and r["loss"] < bar["loss"] and r["bn"] < 1e-4
    assert max(r["grads"].values()) < bar["grads"], K.worst(r["grads"])
    assert r["steps_checked"] > 20000 and r["step_mismatch"] == 0
