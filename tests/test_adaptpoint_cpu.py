"""CPU suite: the AdaptPoint half of the hot path -- generator, discriminator, feedback loss, the
joint G/D step, the classifier step with its resampler, three_interpolation, the kNN grouper --
as host-side mirrors over the oracle operators, against goldens made by the REFERENCE's own
modules and trainer statements (tests/golden/make_golden.py, G9-G13).  Both sides hold the same
name-seeded weights and consume the same CPU random draws, so the bars are float32 re-association
only (1e-5 .. 1e-4 of scale)."""
import numpy as np
import pytest
import torch

import golden_inputs as GI


def rel(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(1e-12, np.abs(ref).max()))


def height_channel(pos):
    return pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def test_draw_noise_consumes_the_generator_like_the_reference():
    """generator_component4_15.py:714 (Gumbel), :245-247 (uniform_, bernoulli, randint), :218/:308."""
    from adaptpoint_amd.augmentor import draw_noise
    torch.manual_seed(4)
    expo = torch.empty(3, 40, 2).exponential_()
    a = torch.Tensor(3, 4, 3).uniform_(0, 1)
    keep = torch.bernoulli(a)
    ax = torch.randint(1, 8, (3, 4))
    kax = torch.randint(1, 8, (3, 1))
    torch.manual_seed(4)
    nz = draw_noise(3, 40, 4, with_gumbel=True)
    assert torch.equal(nz.gumbel_expo, expo) and torch.equal(nz.keep, keep)
    bits = lambda c: ((c[:, :, None] & (1 << torch.arange(3))) > 0).int()
    assert torch.equal(nz.axes, bits(ax)) and torch.equal(nz.kernel_axes, bits(kax))


def test_generator_mirror_matches_reference(golden_ap, cpu_mirrors):
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    g = fill_parameters_by_name(AdaptPointAugmentor(fused=False))
    assert sum(q.numel() for q in g.parameters()) == 5998062
    g.train()
    x = torch.from_numpy(GI.unit_sphere_cloud(2, 512, seed=91))
    torch.manual_seed(int(golden_ap["g9_seed"]))
    noise = draw_noise(2, 512, 4, with_gumbel=True)
    src, out = g(x, noise)
    assert src is not None and out.shape == (2, 512, 3)
    (out * torch.from_numpy(GI.seeded_normal((2, 512, 3), seed=92))).sum().backward()
    ref = golden_ap["g9_gen_out"]
    assert np.array_equal(out.detach().abs().sum(-1).numpy() == 0, np.abs(ref).sum(-1) == 0)   # same mask
    assert rel(out.detach().numpy(), ref) < 2e-5
    sac = g.predict_prob_layer
    assert rel(sac.embedding.net[0].weight.grad.numpy(), golden_ap["g9_grad_embed_w"]) < 1e-3
    assert rel(sac.head.prob_head[0].weight.grad.numpy(), golden_ap["g9_grad_prob_head_w"]) < 1e-3
    assert rel(sac.extract_local_feat_masking[0].weight.grad.numpy(), golden_ap["g9_grad_mask_local_w"]) < 1e-3
    # the default path draws its own noise the same way (CPU logits -> CPU generator, Gumbel first)
    g2 = fill_parameters_by_name(AdaptPointAugmentor(fused=False)).train()
    torch.manual_seed(int(golden_ap["g9_seed"]))
    assert rel(g2(x)[1].detach().numpy(), ref) < 2e-5


def test_generator_geometry_pieces():
    """anchor_transforms / kernel_weights / deform / unit_sphere against a literal float64
    evaluation of generator_component4_15.py:204-327."""
    from adaptpoint_amd import augmentor as AU
    torch.manual_seed(1)
    B, N, M = 2, 50, 4
    x = torch.randn(B, N, 3).double()
    a = x[:, :M].clone()
    prob = torch.randn(B, M, 9).double()
    nz = AU.draw_noise(B, N, M)
    lin, off = AU.anchor_transforms(prob, nz, 10, 3, 0.25)
    keep, ax = nz.keep.double(), nz.axes.double()
    deg = np.pi * (torch.tanh(prob[..., :3]) * 10) / 180.0 * keep[..., 0:1]
    s = (torch.sigmoid(prob[..., 3:6]) * 2 + 1) * keep[..., 1:2] * ax
    s = s + (s == 0)
    t = torch.tanh(prob[..., 6:9]) * 0.25 * keep[..., 2:3] * ax
    sx, sy, sz = torch.sin(deg).unbind(-1)
    cx, cy, cz = torch.cos(deg).unbind(-1)
    R = torch.stack([cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                     sz * cy, sz * sy * sx + cz * cy, sz * sy * cx - cz * sx,
                     -sy, cy * sx, cy * cx], -1).reshape(B, M, 3, 3)
    moved = (x[:, None] - a[:, :, None]) @ R @ torch.diag_embed(s) + t[:, :, None] + a[:, :, None]
    sub = (a[:, :, None] - x[:, None]) * nz.kernel_axes.double()[:, :, None]
    w = torch.exp(-0.5 * sub.pow(2).sum(-1) / 0.25)
    want = (w[..., None] * moved).sum(1) / w.sum(1)[..., None]
    got = AU.deform(x, a, lin, off, AU.kernel_weights(x, a, nz.kernel_axes, 0.5))
    assert torch.allclose(got, want, atol=1e-12)
    z = want - want.mean(1, keepdim=True)
    z = z * (0.999999 / z.norm(dim=-1).amax(1)).view(-1, 1, 1)
    assert torch.allclose(AU.unit_sphere(want), z, atol=1e-12)
    assert float(AU.unit_sphere(want).norm(dim=-1).max()) < 1.0


def test_discriminator_mirror_matches_reference(golden_ap):
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.pointnext import fill_parameters_by_name
    d = no_dropout(fill_parameters_by_name(PointDiscriminator1(num_classes=15)))
    assert sum(q.numel() for q in d.parameters()) == 800671
    assert sorted(d.state_dict().keys()) == list(golden_ap["g10_state_keys"])     # the reference's names
    x = torch.from_numpy(GI.unit_sphere_cloud(2, 512, seed=101))
    d.eval()
    with torch.no_grad():
        assert rel(d(x).numpy(), golden_ap["g10_dis_eval"]) < 1e-6
    d.train()
    xg = x.clone().requires_grad_(True)
    out = d(xg)
    (out * torch.tensor([[1.0], [-2.0]])).sum().backward()
    assert rel(out.detach().numpy(), golden_ap["g10_dis_train"]) < 1e-6
    assert rel(xg.grad.numpy(), golden_ap["g10_grad_x"]) < 1e-5
    assert rel(d.sa1.mlp_convs[0].parametrizations.weight.original.grad.numpy(), golden_ap["g10_grad_conv0"]) < 1e-5
    assert rel(d.fc1.parametrizations.weight[0]._u.numpy(), golden_ap["g10_u_fc1"]) < 1e-6


@pytest.mark.parametrize("batched", [True, False])
def test_gan_step_matches_reference_trainer(golden_ap, cpu_mirrors, batched):
    """One `train_gan` iteration (train_autoaug.py:133-204): the four losses, a gradient of each
    network, and a parameter of each after its Adam step."""
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise
    from adaptpoint_amd.discriminator import PointDiscriminator1
    from adaptpoint_amd.gan import GanStep
    from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy, fill_parameters_by_name
    G = fill_parameters_by_name(AdaptPointAugmentor(fused=False))
    D = no_dropout(fill_parameters_by_name(PointDiscriminator1(num_classes=15)))
    C = fill_parameters_by_name(PointNextSClassifier())
    pos = torch.from_numpy(GI.unit_sphere_cloud(2, 512, seed=int(golden_ap["g11_pos_seed"])))
    points = torch.cat([pos, height_channel(pos)], -1)
    step = GanStep(G, D, C, SmoothCrossEntropy(0.3), batched_feedback=batched)
    grads = {}
    G.predict_prob_layer.embedding.net[0].weight.register_hook(lambda g: grads.__setitem__("embed", g.clone()))
    G.predict_prob_layer.head.prob_head[0].weight.register_hook(lambda g: grads.__setitem__("head", g.clone()))
    D.fc3.parametrizations.weight.original.register_hook(lambda g: grads.__setitem__("fc3", g.clone()))
    torch.manual_seed(int(golden_ap["g11_seed"]))
    res = step(points, torch.tensor([3, 11]), noise=draw_noise(2, 512, 4, with_gumbel=True))
    ref = golden_ap["g11_gen"]
    assert np.array_equal(res["gen"].abs().sum(-1).numpy() == 0, np.abs(ref).sum(-1) == 0)
    assert rel(res["gen"].numpy(), ref) < 2e-5
    got = np.array([res[k].item() for k in ("g_loss_raw", "feedback_loss", "g_loss", "d_loss")])
    np.testing.assert_allclose(got, golden_ap["g11_losses"], rtol=2e-5)
    assert rel(grads["embed"].numpy(), golden_ap["g11_grad_embed_w"]) < 2e-3      # (measured 1.1e-3: ~40 layers, batch-norm over 2 clouds)
    assert rel(grads["head"].numpy(), golden_ap["g11_grad_prob_head_w"]) < 1e-3
    assert rel(grads["fc3"].numpy(), golden_ap["g11_grad_fc3"]) < 1e-4      # the D-step gradient (hook fires last there)
    np.testing.assert_allclose(G.predict_prob_layer.embedding.net[0].weight.detach().numpy(),
                               golden_ap["g11_embed_w_after"], atol=2e-6)
    np.testing.assert_allclose(D.fc3.parametrizations.weight.original.detach().numpy(),
                               golden_ap["g11_fc3_after"], atol=2e-6)
    # the generator step leaves no gradient behind in the networks it only passes through
    assert all(q.grad is None for q in C.parameters())
    # (round 4) EVERY parameter's gradient of both steps against the reference trainer's, in relative L2
    import classifier_b8_checks as K
    eg, _ = K.gradient_errors(G, golden_ap, "g11_gen")
    ed, _ = K.gradient_errors(D, golden_ap, "g11_dis")
    print("G11 on CPU (batched=%s): generator gradients worst" % batched, K.worst(eg, 3), "discriminator worst", K.worst(ed, 2))
    assert max(eg.values()) < 3e-3, K.worst(eg)             # measured: 1.2e-3 (fuse_masking.1.weight), the embedding 1.1e-3
    assert max(ed.values()) < 1e-4, K.worst(ed)


def test_three_interpolation_and_knn_grouper(golden_ap, cpu_mirrors, oracle):
    from adaptpoint_amd.layers import KnnGrouper, three_interpolation
    xyz = GI.config1_xyz()
    known = GI.take_points(xyz, oracle.furthest_point_sampling(xyz, 256))
    feat = torch.from_numpy(GI.seeded_normal((2, 48, 256), seed=121)).requires_grad_(True)
    up = three_interpolation(torch.from_numpy(xyz), torch.from_numpy(known), feat)
    (up * torch.from_numpy(GI.seeded_normal(tuple(up.shape), seed=122))).sum().backward()
    np.testing.assert_allclose(up.detach().numpy(), golden_ap["g12_interp"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(feat.grad.numpy(), golden_ap["g12_interp_grad"], rtol=1e-5, atol=1e-5)
    feats = torch.from_numpy(GI.seeded_normal((2, 16, 1024), seed=123))
    dp, fj = KnnGrouper(8, normalize_dp=True)(torch.from_numpy(known[:, :64]), torch.from_numpy(xyz), feats)
    np.testing.assert_allclose(dp.numpy(), golden_ap["g12_knn_dp"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(fj.numpy(), golden_ap["g12_knn_fj"])


def test_classifier_step_with_resampler_matches_reference_trainer(golden_ap, cpu_mirrors):
    """One `train_one_epoch` iteration (train_autoaug.py:471-512): FPS 2048 -> 1200, the random
    1024 of them, gather; forward, SmoothCE, backward, clip 10, AdamW, zero_grad."""
    from adaptpoint_amd.gan import ClassifierStep, resample
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    pos = torch.from_numpy(GI.unit_sphere_cloud(2, 2048, seed=131))
    points = torch.cat([pos, height_channel(pos)], -1)
    np.random.seed(13)
    choice = np.random.choice(1200, 1024, False)
    assert np.array_equal(choice, golden_ap["g13_choice"])            # numpy's draw is part of the pin
    p1, x1 = resample(points, 1024, 4, choice)
    assert np.array_equal(p1.numpy(), golden_ap["g13_pos"]) and np.array_equal(x1.numpy(), golden_ap["g13_x"])
    np.random.seed(13)
    p2, _ = resample(points, 1024, 4)                                  # draws the same choice itself
    assert torch.equal(p1, p2)
    m = no_dropout(fill_parameters_by_name(PointNextSClassifier()))
    clip_norm = {}
    real_clip = torch.nn.utils.clip_grad_norm_
    step = ClassifierStep(m)
    try:
        torch.nn.utils.clip_grad_norm_ = lambda *a, **k: clip_norm.setdefault("v", real_clip(*a, **k))
        logits, loss = step(points, torch.tensor([5, 14]), choice=choice)
    finally:
        torch.nn.utils.clip_grad_norm_ = real_clip
    np.testing.assert_allclose(logits.detach().numpy(), golden_ap["g13_logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loss.item(), float(golden_ap["g13_loss"]), rtol=1e-5)
    np.testing.assert_allclose(clip_norm["v"].item(), float(golden_ap["g13_grad_norm"]), rtol=1e-4)
    np.testing.assert_allclose(m.prediction.head[-1][0].weight.detach().numpy(), golden_ap["g13_head_w_after"],
                               atol=1e-5)
    np.testing.assert_allclose(m.encoder.encoder[1][0].convs[0][1].running_mean.numpy(),
                               golden_ap["g13_bn1_mean_after"], rtol=1e-5, atol=1e-6)
    assert all(q.grad is None or not q.grad.any() for q in m.parameters())      # model.zero_grad() (:508)


def test_feature_propagation_and_decoder_match_reference(golden_ap, cpu_mirrors, oracle):
    """FeaturePropogation (pointnext.py:173-226), both variants, and the
    segmentation decoder (pointnext.py:461-500) over three_nn / three_interpolate."""
    from adaptpoint_amd.pointnext import FeaturePropagation, PointNextDecoder, fill_parameters_by_name
    fp = fill_parameters_by_name(FeaturePropagation([64 + 32, 32, 32])).train()
    p1 = GI.unit_sphere_cloud(2, 512, seed=141)
    p2 = GI.take_points(p1, oracle.furthest_point_sampling(p1, 128))
    f1 = torch.from_numpy(GI.seeded_normal((2, 32, 512), seed=142)).requires_grad_(True)
    f2 = torch.from_numpy(GI.seeded_normal((2, 64, 128), seed=143)).requires_grad_(True)
    out = fp([torch.from_numpy(p1), f1], [torch.from_numpy(p2), f2])
    (out * torch.from_numpy(GI.seeded_normal(tuple(out.shape), seed=144))).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), golden_ap["g14_fp_out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(f1.grad.numpy(), golden_ap["g14_fp_grad_f1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(f2.grad.numpy(), golden_ap["g14_fp_grad_f2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(fp.convs[0][0].weight.grad.numpy(), golden_ap["g14_fp_grad_w0"], rtol=1e-4, atol=1e-5)
    fpg = fill_parameters_by_name(FeaturePropagation([32, 32, 24], upsample=False)).train()
    og = fpg([None, torch.from_numpy(GI.seeded_normal((2, 32, 100), seed=145))])
    np.testing.assert_allclose(og.detach().numpy(), golden_ap["g14_fp_global_out"], rtol=1e-5, atol=1e-5)
    dec = fill_parameters_by_name(PointNextDecoder([32, 64, 128, 256, 512], decoder_layers=2, decoder_stages=4)).train()
    assert sorted(dec.state_dict().keys()) == list(golden_ap["g14_dec_keys"])
    pl = [GI.unit_sphere_cloud(2, 256, seed=146)]
    for m in (128, 64, 32, 16):
        pl.append(GI.take_points(pl[-1], oracle.furthest_point_sampling(pl[-1], m)))
    fl = [torch.from_numpy(GI.seeded_normal((2, c, n), seed=147 + i))
          for i, (c, n) in enumerate(zip((32, 64, 128, 256, 512), (256, 128, 64, 32, 16)))]
    od = dec([torch.from_numpy(q) for q in pl], fl)
    np.testing.assert_allclose(od.detach().numpy(), golden_ap["g14_dec_out"], rtol=1e-4, atol=1e-5)
