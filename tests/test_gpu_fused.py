"""GPU tests of the fused grouped-MLP kernels (csrc/sa_fused.hip) against plain PyTorch.

Tolerances (floating point, stated here as section 3 of the task asks):
  * against the reference that rounds where the bf16 MFMA path rounds
    (tests/fused_reference.py, float64 accumulation): max |err| <= 2e-2, mean |err| <= 5e-4
    on BN-normalised outputs of O(1) -- what is left is f32-vs-f64 accumulation order and the
    occasional one-ulp flip of a bf16 rounding;
  * against the plain fp32 chain: max |err| <= 1.5e-1, mean |err| <= 1e-2 (bf16 inputs have
    8 significant bits; BASELINE.json's north_star asks for the bf16 MFMA contraction).
"""
import numpy as np
import pytest
import torch

import golden_inputs as GI
from fused_reference import chain

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bf16x3", "bf16"])
def precision(request):
    from adaptpoint_amd import fused
    old = fused.PRECISION
    fused.PRECISION = request.param
    yield request.param
    fused.PRECISION = old


def _setup(dev, B=4, seed=0, neg_gamma=False):
    from adaptpoint_amd.layers import ball_query, furthest_point_sample
    p = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=seed)).to(dev)
    f = torch.from_numpy(GI.seeded_normal((B, 32, 1024), seed=seed + 1)).to(dev)
    fidx = furthest_point_sample(p, 512).long()
    new_p = torch.gather(p, 1, fidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = ball_query(0.15, 32, p, new_p)
    torch.manual_seed(seed)
    conv1 = torch.nn.Conv2d(35, 32, 1, bias=False).to(dev)
    conv2 = torch.nn.Conv2d(32, 64, 1, bias=False).to(dev)
    bn1 = torch.nn.BatchNorm2d(32).to(dev)
    bn2 = torch.nn.BatchNorm2d(64).to(dev)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        if neg_gamma:
            bn2.weight[::3] *= -1
            bn1.weight[::5] *= -1
    return p, new_p, f, idx, conv1, bn1, conv2, bn2


@pytest.mark.parametrize("neg_gamma", [False, True])
def test_fused_forward_matches_pytorch(dev, neg_gamma, precision):
    from adaptpoint_amd.fused import grouped_mlp_max, supported
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = _setup(dev, neg_gamma=neg_gamma)
    assert supported(p, f, idx, conv1, conv2)
    with torch.no_grad():
        out = grouped_mlp_max(p, new_p, f, idx, 0.15, conv1, bn1, conv2, bn2)
    torch.cuda.synchronize()
    args = (p, new_p, f, idx, 0.15, conv1.weight.view(32, 35), bn1.weight, bn1.bias,
            conv2.weight.view(64, 32), bn2.weight, bn2.bias)
    ref_bf, _ = chain(*args, emulate_bf16=True)
    ref_32, _ = chain(*args, emulate_bf16=False)
    e_bf = (out.double() - ref_bf).abs()
    e_32 = (out.double() - ref_32).abs()
    print("fused fwd err vs bf16-emulation max %.3e mean %.3e | vs fp32 max %.3e mean %.3e"
          % (e_bf.max(), e_bf.mean(), e_32.max(), e_32.mean()))
    if precision == "bf16":
        assert e_bf.max() <= 2e-2 and e_bf.mean() <= 5e-4
        assert e_32.max() <= 1.5e-1 and e_32.mean() <= 1e-2
    else:
        # split operands: fp32-grade.  max 2e-3 allows the rare arg-max flip between two
        # neighbours closer than 1e-5; the mean is the real measure.
        assert e_32.max() <= 2e-3 and e_32.mean() <= 2e-5


def test_fused_forward_updates_running_stats(dev):
    from adaptpoint_amd.fused import grouped_mlp_max
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = _setup(dev, B=2, seed=3)
    with torch.no_grad():
        grouped_mlp_max(p, new_p, f, idx, 0.15, conv1, bn1, conv2, bn2)
    _, mid = chain(p, new_p, f, idx, 0.15, conv1.weight.view(32, 35), bn1.weight, bn1.bias,
                   conv2.weight.view(64, 32), bn2.weight, bn2.bias, emulate_bf16=True)
    n = 2 * 512 * 32
    np.testing.assert_allclose(bn1.running_mean.cpu().numpy(), 0.1 * mid["m1"].flatten().cpu().numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(bn1.running_var.cpu().numpy(),
                               (0.9 + 0.1 * mid["v1"].flatten() * n / (n - 1)).cpu().numpy(), rtol=2e-3)
    np.testing.assert_allclose(bn2.running_mean.cpu().numpy(), 0.1 * mid["m2"].flatten().cpu().numpy(), rtol=5e-3, atol=5e-4)
    assert int(bn1.num_batches_tracked) == 1 and int(bn2.num_batches_tracked) == 1


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))


@pytest.mark.parametrize("neg_gamma", [False, True])
def test_fused_backward_matches_autograd(dev, neg_gamma, precision):
    """Gradients of the fused op vs torch autograd through the plain chain.
    Tolerance: max |err| / max |ref| <= 2e-2 against the bf16-rounding reference (the fused
    backward additionally rounds dL/dy2 and the one-hot operand to bf16 for its MFMAs).
    Against the fp32 chain the comparison is in relative L2 norm, <= 0.25: rounding the inputs
    to bf16 flips the arg-max of the K-pool wherever two neighbours are within 2^-8, which moves
    whole gradient entries from one point to another (an element-wise max norm is meaningless
    there); the aggregated parameter gradients stay within 0.15."""
    from adaptpoint_amd.fused import grouped_mlp_max
    from fused_reference import chain_grad
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = _setup(dev, neg_gamma=neg_gamma, seed=5)
    wts = torch.randn(4, 64, 512, device=dev, generator=torch.Generator(dev).manual_seed(9))
    p.requires_grad_(True); new_p.requires_grad_(True); f.requires_grad_(True)
    out = grouped_mlp_max(p, new_p, f, idx, 0.15, conv1, bn1, conv2, bn2)
    (out * wts).sum().backward()
    torch.cuda.synchronize()
    got = dict(f=f.grad.clone(), p=p.grad.clone(), newp=new_p.grad.clone(),
               w1=conv1.weight.grad.view(32, 35).clone(), w2=conv2.weight.grad.view(64, 32).clone(),
               g1=bn1.weight.grad.clone(), b1=bn1.bias.grad.clone(),
               g2=bn2.weight.grad.clone(), b2=bn2.bias.grad.clone())
    for emu, tol in (((True, 2e-2), (False, None)) if precision == "bf16" else ((False, 5e-3),)):
        leaves = [t.detach().clone().requires_grad_(True) for t in
                  (p, new_p, f, conv1.weight.view(32, 35), bn1.weight, bn1.bias,
                   conv2.weight.view(64, 32), bn2.weight, bn2.bias)]
        rp, rq, rf, rw1, rg1, rb1, rw2, rg2, rb2 = leaves
        ref, _ = chain_grad(rp, rq, rf, idx, 0.15, rw1, rg1, rb1, rw2, rg2, rb2, emulate_bf16=emu)
        (ref * wts.double()).sum().backward()
        want = dict(f=rf.grad, p=rp.grad, newp=rq.grad, w1=rw1.grad, w2=rw2.grad,
                    g1=rg1.grad, b1=rb1.grad, g2=rg2.grad, b2=rb2.grad)
        if precision == "bf16x3":
            # Split operands against the plain fp32 chain (float64 accumulation).  Per-term
            # precision is ~1e-5 (the dropped lo*lo product); BatchNorm's backward sums 524k
            # signed terms that cancel ~700-fold, and an arg-max that flips between two
            # neighbours closer than the forward error (7e-5) moves one whole gradient entry
            # (~1 in 2000 pooled values), so the bars are relative L2 <= 5e-3 and max-norm
            # <= 5e-2 (observed: L2 2e-3 through the pool, 8e-6 for dL/dW2 which bypasses it).
            errs = {k: _rel(got[k], want[k]) for k in got}
            l2 = {k: _rel_l2(got[k], want[k]) for k in got}
            print("bf16x3 vs fp32 max-norm", {k: "%.2e" % v for k, v in errs.items()})
            print("bf16x3 vs fp32 rel-L2  ", {k: "%.2e" % v for k, v in l2.items()})
            for k in got:
                assert errs[k] <= 5e-2 and l2[k] <= 5e-3, (k, errs[k], l2[k])
            continue
        errs = {k: (_rel if emu else _rel_l2)(got[k], want[k]) for k in got}
        print(("bf16-emu max-norm" if emu else "fp32 rel-L2      "), {k: "%.2e" % v for k, v in errs.items()})
        for k, v in errs.items():
            assert v <= (tol if emu else (0.25 if k in ("f", "p", "newp") else 0.15)), (k, v, emu)


def test_set_abstraction_fused_equals_unfused(dev, precision):
    """The block with fused=True against the same block over the nine unfused operators
    (the drop-in path): same weights, same input."""
    from adaptpoint_amd.set_abstraction import SetAbstraction
    kw = dict(layers=2, stride=2, group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32,
                                              'normalize_dp': True},
              norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
              use_res=True)
    torch.manual_seed(1)
    a = SetAbstraction(32, 64, **kw).to(dev)
    b = SetAbstraction(32, 64, fused=True, **kw).to(dev)
    b.load_state_dict(a.state_dict())
    p = torch.from_numpy(GI.unit_sphere_cloud(8, 1024, seed=2)).to(dev)
    f1 = torch.from_numpy(GI.seeded_normal((8, 32, 1024), seed=3)).to(dev).requires_grad_(True)
    f2 = f1.detach().clone().requires_grad_(True)
    pa, oa = a([p, f1])
    pb, ob = b([p, f2])
    assert torch.equal(pa, pb)
    oa.sum().backward(); ob.sum().backward()
    # the unfused side is MIOpen fp32 (Winograd-class kernels, ~1e-3); bf16x3 sits at that level
    # ten times the measured differences (bf16x3: out 4.9e-5 max / 5.7e-6 mean, gradients <= 1.9e-3; bf16: 2.6e-2 /
    # 3.0e-3, gradients <= 1.0e-1 through the arg-max flips at 2^-8)
    fo, fg, fw_ = (1.5e-1, 0.25, 0.3) if precision == "bf16" else (5e-4, 2e-2, 2e-2)
    werr = {k: _rel_l2(qb.grad, qa.grad) for (k, qa), (_, qb) in zip(a.named_parameters(), b.named_parameters())}
    print("fused vs unfused (%s): out max %.2e mean %.2e, grad f %.2e, weights %s" % (
        precision, (oa - ob).abs().max(), (oa - ob).abs().mean(), _rel_l2(f2.grad, f1.grad),
        {k: "%.1e" % v for k, v in werr.items()}))
    assert (oa - ob).abs().max() <= fo and (oa - ob).abs().mean() <= fo / 8
    assert _rel_l2(f2.grad, f1.grad) <= fg
    for k, v in werr.items():
        assert v <= fw_, k
    for (k, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=2e-2, atol=2e-3), k


def test_fused_block_new_p_and_idx_are_exact(dev, oracle):
    """The sampling / query stage inside the fused block is the index-exact one."""
    from adaptpoint_amd.fused import _call
    xyz = GI.unit_sphere_cloud(4, 1024, seed=11)
    p = torch.from_numpy(xyz).to(dev)
    fidx = torch.empty(4, 512, dtype=torch.int32, device=dev)
    new_p = torch.empty(4, 512, 3, device=dev)
    temp = torch.full((4, 1024), 1e10, device=dev)
    _call("apn_furthest_point_sampling_xyz", dev, 4, 1024, 512, p.data_ptr(), temp.data_ptr(),
          fidx.data_ptr(), new_p.data_ptr())
    o_idx = oracle.furthest_point_sampling(xyz, 512)
    assert np.array_equal(fidx.cpu().numpy(), o_idx)
    assert np.array_equal(new_p.cpu().numpy(), GI.take_points(xyz, o_idx))
    idx = torch.full((4, 512, 32), -5, dtype=torch.int32, device=dev)     # NOT pre-zeroed
    far = new_p.clone()
    far[:, :7] += 9.0                                                       # some empty balls
    _call("apn_ball_query_zero", dev, 4, 1024, 512, 0.15, 32, far.data_ptr(), p.data_ptr(), idx.data_ptr())
    assert np.array_equal(idx.cpu().numpy(), oracle.ball_query(0.15, 32, xyz, far.cpu().numpy()))


@pytest.mark.parametrize("B,N,M", [(4, 1024, 512), (3, 1200, 300), (2, 4096, 1024), (2, 512, 128),
                                   (2, 6000, 64)])
def test_overlapped_index_stages_are_exact(dev, oracle, B, N, M):
    """sample_and_query_many: FPS of batch i and ball query of batch i-1 share a launch (fused
    kernel for 512 < N <= 4096, the two launches back to back otherwise) -- same indices as the
    oracle, for every batch, including the first and last (single-role launches)."""
    from adaptpoint_amd.fused import sample_and_query_many
    clouds = [GI.unit_sphere_cloud(B, N, seed=20 + i) for i in range(3)]
    clouds[1][:, N // 2:] = clouds[1][:, :1]            # half the points duplicated: ties + empty-ish balls
    ps = [torch.from_numpy(c).to(dev) for c in clouds]
    outs = sample_and_query_many(ps, M, 0.15, 32)
    torch.cuda.synchronize()
    for c, o in zip(clouds, outs):
        o_idx = oracle.furthest_point_sampling(c, M)
        assert np.array_equal(o.fidx.cpu().numpy(), o_idx)
        q = GI.take_points(c, o_idx)
        assert np.array_equal(o.new_p.cpu().numpy(), q)
        assert np.array_equal(o.idx.cpu().numpy(), oracle.ball_query(0.15, 32, c, q))


def test_fused_block_eval_mode_and_xyz_grad(dev):
    """eval-mode BatchNorm (running statistics) forward+backward, with gradient flowing to the
    coordinates -- the AdaptPoint feedback path (function_adaptpoint/ganloss_cls.py:31-65)."""
    from adaptpoint_amd.set_abstraction import SetAbstraction
    kw = dict(layers=2, stride=2, group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32,
                                              'normalize_dp': True},
              norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
              use_res=True)
    torch.manual_seed(4)
    a = SetAbstraction(32, 64, **kw).to(dev)
    b = SetAbstraction(32, 64, fused=True, **kw).to(dev)
    with torch.no_grad():
        for bn in (a.convs[0][1], a.convs[1][1]):
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    b.load_state_dict(a.state_dict())
    a.eval(); b.eval()
    p1 = torch.from_numpy(GI.unit_sphere_cloud(4, 1024, seed=6)).to(dev).requires_grad_(True)
    p2 = p1.detach().clone().requires_grad_(True)
    f1 = torch.from_numpy(GI.seeded_normal((4, 32, 1024), seed=7)).to(dev).requires_grad_(True)
    f2 = f1.detach().clone().requires_grad_(True)
    wts = torch.randn(4, 64, 512, device=dev, generator=torch.Generator(dev).manual_seed(1))
    pa, oa = a([p1, f1]); ((oa * wts).sum() + pa.sum()).backward()
    pb, ob = b([p2, f2]); ((ob * wts).sum() + pb.sum()).backward()
    assert torch.equal(pa, pb)
    assert (oa - ob).abs().max() <= 1.5e-1 and (oa - ob).abs().mean() <= 1e-2
    assert _rel_l2(f2.grad, f1.grad) <= 0.25
    assert _rel_l2(p2.grad, p1.grad) <= 0.25
    sd_a, sd_b = a.state_dict(), b.state_dict()
    for k in sd_a:                     # eval mode: no buffer moved
        assert torch.equal(sd_a[k], sd_b[k]), k


def test_phased_syncbn_path_equals_single_call(dev):
    """The SyncBatchNorm code path (three phases per direction, float64 row sums handed to the
    consumers instead of partial rows) gives the same numbers as the single-call path."""
    from adaptpoint_amd import fused
    from adaptpoint_amd.set_abstraction import SetAbstraction
    kw = dict(layers=2, stride=2, group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32,
                                              'normalize_dp': True},
              norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
              use_res=True, fused=True)
    torch.manual_seed(2)
    a = SetAbstraction(32, 64, **kw).to(dev)
    b = SetAbstraction(32, 64, sync_bn=True, **kw).to(dev)
    b.load_state_dict(a.state_dict())
    p = torch.from_numpy(GI.unit_sphere_cloud(4, 1024, seed=8)).to(dev)
    f1 = torch.from_numpy(GI.seeded_normal((4, 32, 1024), seed=9)).to(dev).requires_grad_(True)
    f2 = f1.detach().clone().requires_grad_(True)
    _, oa = a([p, f1]); oa.sum().backward()
    try:
        fused.FORCE_PHASED = True
        _, ob = b([p, f2]); ob.sum().backward()
    finally:
        fused.FORCE_PHASED = False
    assert torch.allclose(oa, ob, rtol=1e-5, atol=1e-6)
    # G, dL/dW2 and the dL/dW1 replicas are summed by float atomics: run-to-run order effects
    assert _rel(f2.grad, f1.grad) <= 1e-4
    for (k, qa), (_, qb) in zip(a.named_parameters(), b.named_parameters()):
        assert _rel(qb.grad, qa.grad) <= 1e-3, k
    for (k, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-6, atol=1e-7), k


def test_per_kernel_diagnostic_path_equals_sequences(dev):
    from adaptpoint_amd import fused
    from adaptpoint_amd.set_abstraction import SetAbstraction
    kw = dict(layers=2, stride=2, group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32,
                                              'normalize_dp': True},
              norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
              use_res=True, fused=True)
    torch.manual_seed(3)
    a = SetAbstraction(32, 64, **kw).to(dev)
    b = SetAbstraction(32, 64, **kw).to(dev)
    b.load_state_dict(a.state_dict())
    p = torch.from_numpy(GI.unit_sphere_cloud(2, 1024, seed=8)).to(dev)
    f1 = torch.from_numpy(GI.seeded_normal((2, 32, 1024), seed=9)).to(dev).requires_grad_(True)
    f2 = f1.detach().clone().requires_grad_(True)
    _, oa = a([p, f1]); oa.sum().backward()
    try:
        fused.PER_KERNEL_LAUNCH = True
        _, ob = b([p, f2]); ob.sum().backward()
    finally:
        fused.PER_KERNEL_LAUNCH = False
    assert torch.equal(oa, ob)
    assert _rel(f2.grad, f1.grad) <= 1e-4          # float-atomic sums: order effects only


@pytest.mark.parametrize("B,N,stride", [(3, 777, 3), (1, 64, 2), (5, 2048, 2), (2, 4100, 4)])
def test_fused_block_odd_sizes(dev, B, N, stride):
    """Sizes that are not multiples of the kernels' tiles (64-point / 64-query tiles, 4-wave
    workgroups), both FPS step algorithms (n <= 4096 and above)."""
    from adaptpoint_amd.set_abstraction import SetAbstraction
    kw = dict(layers=2, stride=stride, group_args={'NAME': 'ballquery', 'radius': 0.2, 'nsample': 32,
                                                   'normalize_dp': True},
              norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
              use_res=True)
    torch.manual_seed(7)
    a = SetAbstraction(32, 64, **kw).to(dev)
    b = SetAbstraction(32, 64, fused=True, **kw).to(dev)
    b.load_state_dict(a.state_dict())
    p = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=N)).to(dev)
    f1 = torch.from_numpy(GI.seeded_normal((B, 32, N), seed=N + 1)).to(dev).requires_grad_(True)
    f2 = f1.detach().clone().requires_grad_(True)
    pa, oa = a([p, f1]); oa.sum().backward()
    pb, ob = b([p, f2]); ob.sum().backward()
    assert pb.shape == (B, N // stride, 3) and torch.equal(pa, pb)
    assert (oa - ob).abs().max() <= 2e-2
    assert _rel_l2(f2.grad, f1.grad) <= 3e-2
    for (k, qa), (_, qb) in zip(a.named_parameters(), b.named_parameters()):
        assert _rel_l2(qb.grad, qa.grad) <= 3e-2, k


def test_full_size_batch_is_cloud_independent_in_eval_mode(dev):
    """BASELINE configs[1] size (B=32, N=1024): with running BatchNorm statistics nothing couples
    the clouds of a batch, so the fused block on 32 clouds must equal, bit for bit, the same block
    on any single cloud of the batch -- the property that makes sharding by cloud
    exact.  Also checks the fused block against the unfused drop-in path at full size."""
    from adaptpoint_amd.set_abstraction import SetAbstraction
    kw = dict(layers=2, stride=2, group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32,
                                              'normalize_dp': True},
              norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
              use_res=True)
    torch.manual_seed(7)
    blk = SetAbstraction(32, 64, fused=True, **kw).to(dev)
    ref = SetAbstraction(32, 64, **kw).to(dev)
    ref.load_state_dict(blk.state_dict())
    p = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=40)).to(dev)
    f = torch.from_numpy(GI.seeded_normal((32, 32, 1024), seed=41)).to(dev)
    blk.train(); ref.train()
    with torch.no_grad():                       # one training-mode pass: non-trivial running statistics
        _, o_train = blk([p, f])
        _, r_train = ref([p, f])
    assert (o_train - r_train).abs().max() <= 1e-2 and (o_train - r_train).abs().mean() <= 1e-3
    blk.eval()
    with torch.no_grad():
        new_p, out = blk([p, f])
        for i in (0, 13, 31):
            np_i, out_i = blk([p[i:i + 1].contiguous(), f[i:i + 1].contiguous()])
            assert torch.equal(np_i[0], new_p[i])
            assert torch.equal(out_i[0], out[i])


SA_KW = dict(layers=2, stride=2, group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32,
                                             'normalize_dp': True},
             norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
             use_res=True)


@pytest.mark.parametrize("fused", [False, True])
def test_set_abstraction_matches_reference_golden_on_gpu(dev, golden, fused):
    """G4: the REFERENCE's SetAbstraction (openpoints/models/backbone/pointnext.py:82-170, run in the
    build container over the oracle operators) against this block on the GPU, unfused (nine
    extension operators + PyTorch conv/BN) and fused, with the reference's state_dict."""
    from adaptpoint_amd.set_abstraction import SetAbstraction
    blk = SetAbstraction(32, 64, fused=fused, **SA_KW).to(dev)
    blk.load_state_dict({k.split("/", 1)[1]: torch.from_numpy(golden[k]) for k in golden.files
                         if k.startswith("g4_sa_state/")})
    blk.train()
    p = torch.from_numpy(GI.unit_sphere_cloud(2, 1024, seed=3)).to(dev)
    f = torch.from_numpy(GI.seeded_normal((2, 32, 1024), seed=4)).to(dev).requires_grad_(True)
    new_p, out = blk([p, f])
    (out * torch.from_numpy(GI.seeded_normal(tuple(out.shape), seed=5)).to(dev)).sum().backward()
    assert np.array_equal(new_p.cpu().numpy(), golden["g4_sa_new_p"])
    ref_out, ref_gf = torch.from_numpy(golden["g4_sa_out"]), torch.from_numpy(golden["g4_sa_grad_f"])
    e_out = float((out.detach().cpu() - ref_out).abs().max())
    errs = {"out_max": e_out, "out_mean": float((out.detach().cpu() - ref_out).abs().mean()),
            "grad_f_l2": _rel_l2(f.grad.cpu(), ref_gf)}
    for k, prm in blk.named_parameters():
        errs[k] = _rel_l2(prm.grad.cpu(), torch.from_numpy(golden["g4_sa_grad/" + k]))
    print("SetAbstraction(fused=%s) vs reference golden:" % fused, {k: "%.2e" % v for k, v in errs.items()})
    # unfused: fp32 everywhere (MIOpen's summation order); fused: split-bf16 MFMA (1e-5-level terms,
    # a K-pool arg-max may flip between two neighbours closer than that)
    # measured: unfused 2e-6 / 3e-6 everywhere (fp32 end to end); fused 5e-5 max, 6e-6 mean in the output and
    # <= 2e-3 in the gradients upstream of the ReLU gates (test_discontinuities_explain_the_gradient_residual)
    if fused:
        assert errs["out_max"] <= 5e-4 and errs["out_mean"] <= 2e-5
        assert all(v <= 5e-3 for k, v in errs.items() if k not in ("out_max", "out_mean")), errs
    else:
        assert errs["out_max"] <= 2e-5 and errs["out_mean"] <= 1e-6
        assert all(v <= 2e-5 for k, v in errs.items() if k not in ("out_max", "out_mean")), errs


def _grads_vs_chain(dev, bn1_bias_shift, mask_pool, y1_noise_std=None):
    """Gradients of the fused op (bf16x3) and of the float64 chain for the same loss; optionally the
    pooled positions whose two best DISTINCT candidates lie within 1e-4 are taken out of the loss on
    both sides.  With y1_noise_std the comparison is chain-vs-chain: the second chain sees conv1's
    output perturbed by N(0, std) -- a model of a 1e-5-level forward error."""
    from adaptpoint_amd.fused import grouped_mlp_max
    from fused_reference import chain_grad
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = _setup(dev, seed=5)
    with torch.no_grad():
        bn1.bias += bn1_bias_shift
    wts = torch.randn(4, 64, 512, device=dev, generator=torch.Generator(dev).manual_seed(9))

    def reference(noise=None):
        leaves = [t.detach().clone().requires_grad_(True) for t in
                  (p, new_p, f, conv1.weight.view(32, 35), bn1.weight, bn1.bias,
                   conv2.weight.view(64, 32), bn2.weight, bn2.bias)]
        out, mid = chain_grad(leaves[0], leaves[1], leaves[2], idx, 0.15, *leaves[3:], y1_noise=noise)
        return out, mid, leaves
    ref, mid, leaves = reference()
    keep = torch.ones_like(wts)
    if mask_pool:
        with torch.no_grad():
            z = ((mid["y2"] - mid["m2"]) / torch.sqrt(mid["v2"] + 1e-5) * bn2.weight.double().view(1, -1, 1, 1)
                 + bn2.bias.double().view(1, -1, 1, 1))
            top2 = z.topk(2, dim=-1).values
            # copies of one neighbour (the ball-query fill) tie exactly and a flip between them moves
            # nothing: only gaps between distinct values count
            gap = torch.where(top2[..., 0] == top2[..., 1], torch.ones_like(top2[..., 0]),
                              top2[..., 0] - top2[..., 1])
            keep = (gap > 1e-4).to(wts.dtype)
    (ref * (wts * keep).double()).sum().backward()
    names = ("p", "newp", "f", "w1", "g1", "b1", "w2", "g2", "b2")
    want = {k: t.grad for k, t in zip(names, leaves)}
    if y1_noise_std is not None:
        noise = torch.randn(mid["y1"].shape, device=dev, dtype=torch.float64,
                            generator=torch.Generator(dev).manual_seed(77)) * y1_noise_std
        ref2, _, leaves2 = reference(noise)
        (ref2 * (wts * keep).double()).sum().backward()
        got = {k: t.grad for k, t in zip(names, leaves2)}
    else:
        p.requires_grad_(True); new_p.requires_grad_(True); f.requires_grad_(True)
        out = grouped_mlp_max(p, new_p, f, idx, 0.15, conv1, bn1, conv2, bn2)
        (out * wts * keep).sum().backward()
        got = dict(f=f.grad, p=p.grad, newp=new_p.grad, w1=conv1.weight.grad.view(32, 35),
                   w2=conv2.weight.grad.view(64, 32), g1=bn1.weight.grad, b1=bn1.bias.grad,
                   g2=bn2.weight.grad, b2=bn2.bias.grad)
    gate_near_zero = float((((mid["y1"] - mid["m1"]) / torch.sqrt(mid["v1"] + 1e-5) * bn1.weight.double().view(1, -1, 1, 1)
                             + bn1.bias.double().view(1, -1, 1, 1)).abs() < 2e-5).double().mean())
    return {k: _rel_l2(got[k], want[k]) for k in names}, 1.0 - keep.mean().item(), gate_near_zero


def test_discontinuities_explain_the_gradient_residual(dev):
    """VERDICT weak #3.  The split-operand (bf16x3) gradients upstream of BatchNorm-1 differ from
    the fp32 chain by ~2e-3 relative L2 although every product is good to ~1e-5.  Claim: the
    residual is the chain's own discontinuities -- mostly ReLU gates that a 1e-5-level forward error
    switches -- not arithmetic.  Two measurements on the same inputs:
      (b) fused vs float64 chain with the near-tie K-pool winners taken out of the loss on both
          sides: the residual stays (measured f 1.9e-3, w1 1.5e-3), so pool flips are not its
          main source; what bypasses the gates (dL/dW2, dL/dgamma2, dL/dbeta2) agrees to 6e-6;
      (c) the float64 chain compared WITH ITSELF under a N(0, 1e-5) perturbation of conv1's
          output shows a residual of the same size (f 3.3e-3, w1 3.1e-3): a fraction eps of
          switched gates moves the gradient by ~sqrt(eps) in relative L2 (1.4e-5 of the
          pre-activations lie within 2e-5 of zero here: sqrt = 3.8e-3).
    (Holding the gates open with a large BatchNorm-1 bias is no cleaner a probe: the offset costs the
    hi+lo split its dynamic range.)"""
    from adaptpoint_amd import fused
    assert fused.PRECISION == "bf16x3"
    b, frac_b, near0 = _grads_vs_chain(dev, bn1_bias_shift=0.0, mask_pool=True)
    print("(b) normal block, pool masked (%.3f%%), gates within 2e-5 of zero: %.2e:" % (100 * frac_b, near0),
          {k: "%.1e" % v for k, v in b.items()})
    c, _, _ = _grads_vs_chain(dev, bn1_bias_shift=0.0, mask_pool=True, y1_noise_std=1e-5)
    print("(c) float64 chain vs itself with conv1 output perturbed by 1e-5:", {k: "%.1e" % v for k, v in c.items()})
    for k in ("f", "p", "newp", "w1", "g1", "b1"):       # everything upstream of the ReLU gates
        assert b[k] <= 5e-3 and b[k] <= 4 * c[k] + 1e-4, (k, b[k], c[k])
    for k in ("g2", "b2"):                                # bypass gates and (masked) pool: arithmetic only
        assert b[k] <= 1e-4, (k, b[k])


def test_syncbn_two_ranks_equal_one_double_batch(dev):
    """ADVICE (high): SyncBatchNorm semantics of the fused block, checked numerically on one GPU.
    Two "ranks" hold uneven shards (3 and 5 clouds) of one batch.  The block's statistics exchange
    (`fused._allreduce_sum_`, four per step) is replaced by a replaying stub: pass k re-runs both
    ranks with exchanges 0..k-1 returning the true global sums found by the earlier passes, and
    records both ranks' local contribution to exchange k -- after five passes every exchange saw
    what a real all-reduce would have delivered.  Property: outputs and input gradients equal the
    single-process run on the concatenated batch with loss L0 + L1; parameter gradients, averaged
    over the ranks as DistributedDataParallel does, equal that run's divided by the world size --
    for the convolution weights AND the BatchNorm affine parameters (torch.nn.SyncBatchNorm keeps
    those rank-local; the fused block reports global / world)."""
    import copy
    from adaptpoint_amd import fused
    from adaptpoint_amd.set_abstraction import SetAbstraction
    torch.manual_seed(12)
    full = SetAbstraction(32, 64, fused=True, **SA_KW).to(dev)
    state0 = copy.deepcopy(full.state_dict())
    halves = [SetAbstraction(32, 64, fused=True, sync_bn=True, **SA_KW).to(dev) for _ in range(2)]
    sizes = (3, 5)                                   # uneven shards: the count rides with the sums
    p = torch.from_numpy(GI.unit_sphere_cloud(sum(sizes), 1024, seed=50)).to(dev)
    f = torch.from_numpy(GI.seeded_normal((sum(sizes), 32, 1024), seed=51)).to(dev)
    wts = torch.randn(sum(sizes), 64, 512, device=dev, generator=torch.Generator(dev).manual_seed(3))
    f_full = f.clone().requires_grad_(True)
    _, o_full = full([p, f_full])
    (o_full * wts).sum().backward()

    known, res = [], [None, None]

    def stub_for(log):
        def stub(t):
            k = len(log)
            log.append(t.clone())
            if k < len(known):
                t.copy_(known[k])
        return stub

    old = (fused._allreduce_sum_, fused.FORCE_PHASED)
    try:
        fused.FORCE_PHASED = True
        for it in range(5):
            logs = [[], []]
            for rank in range(2):
                lo = sum(sizes[:rank])
                sl = slice(lo, lo + sizes[rank])
                halves[rank].load_state_dict(copy.deepcopy(state0))
                halves[rank].zero_grad(set_to_none=True)
                fused._allreduce_sum_ = stub_for(logs[rank])
                fi = f[sl].clone().requires_grad_(True)
                _, o = halves[rank]([p[sl].contiguous(), fi])
                (o * wts[sl]).sum().backward()
                res[rank] = (o.detach(), fi.grad)
                assert len(logs[rank]) == 4           # two exchanges per direction
            if it < 4:
                known.append(logs[0][it] + logs[1][it])
    finally:
        fused._allreduce_sum_, fused.FORCE_PHASED = old
    # the reduced vectors carry {sums, global count, world size}
    assert known[0].numel() == 64 + 2 and float(known[0][-1]) == 2.0
    assert float(known[0][-2]) == sum(sizes) * 512 * 32
    o_cat = torch.cat([res[0][0], res[1][0]])
    g_cat = torch.cat([res[0][1], res[1][1]])
    assert torch.allclose(o_cat, o_full, rtol=1e-5, atol=1e-5)
    assert _rel(g_cat, f_full.grad) <= 1e-4
    for (k, q), (_, q0), (_, q1) in zip(full.named_parameters(), halves[0].named_parameters(),
                                        halves[1].named_parameters()):
        ddp = (q0.grad + q1.grad) / 2             # what DDP's averaging leaves on every rank
        assert _rel(ddp, q.grad / 2) <= 1e-3, k
    for (k, b), (_, b0), (_, b1) in zip(full.named_buffers(), halves[0].named_buffers(),
                                        halves[1].named_buffers()):
        assert torch.allclose(b.float(), b0.float(), rtol=1e-5, atol=1e-6), k   # running stats: global
        assert torch.equal(b0, b1), k


@pytest.mark.gpu
def test_register_resident_kernels_over_the_tile_map_equal_one_tile_per_query():
    """The distinct-hit tile map (index-stage work, fused_wide.tile_map) only regroups the positions: folded fill
    copies are weighted by their multiplicity, the pool runs per query over the distinct rows.  Same block, same
    inputs: the forward with and without the map (the backward runs over a map always since round 5 -- the one handed in
    with its row map, or one it builds): forward equal to summation-order rounding, gradients likewise."""
    import copy
    from adaptpoint_amd import fused, fused_wide
    from adaptpoint_amd.set_abstraction import SetAbstraction
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blk = SetAbstraction(32, 64, layers=2, stride=2, fused=True, use_res=True,
                         group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                         norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'}).to(dev)
    blk2 = copy.deepcopy(blk)
    p = torch.from_numpy(GI.unit_sphere_cloud(8, 1024, seed=4)).to(dev)
    f = torch.from_numpy(GI.seeded_normal((8, 32, 1024), seed=5)).to(dev)
    fa, fb = f.clone().requires_grad_(True), f.clone().requires_grad_(True)
    smp_a = blk.sample(p)
    smp_b = blk2.sample(p)
    smp_b.tmap = fused_wide.tile_map(smp_b.idx)
    smp_b.rowmap = fused_wide.row_map(smp_b.tmap, 8, 1024, 512)
    assert int(smp_b.tmap[0]) < smp_b.idx.shape[0] * smp_b.idx.shape[1] // 2          # the map does fold
    qa, oa = blk([p, fa], sampling=smp_a)
    qb, ob = blk2([p, fb], sampling=smp_b)
    assert torch.equal(qa, qb)
    err = (oa - ob).abs()
    assert err.max() <= 1e-4 and err.mean() <= 2e-6, (float(err.max()), float(err.mean()))
    w = torch.randn_like(oa)
    (oa * w).sum().backward()
    (ob * w).sum().backward()
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert rel(fb.grad, fa.grad) <= 2e-3, rel(fb.grad, fa.grad)          # a pooled near-tie may flip (see test_gpu_fused_wide)
    for (n1, q1), (_, q2) in zip(blk2.named_parameters(), blk.named_parameters()):
        assert rel(q1.grad, q2.grad) <= 2e-3, (n1, rel(q1.grad, q2.grad))
    for b1, b2 in zip(blk2.buffers(), blk.buffers()):
        assert torch.allclose(b1.float(), b2.float(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("with_map", [True, False])
@pytest.mark.parametrize("loss_scale", [1.0, 65536.0, 1e-6])
def test_gradients_are_bit_identical_run_to_run(dev, with_map, loss_scale):
    """Round 5: no float atomics on the chain -- the backward pass STORES every row's g_u at its place in the point-sorted
    order (fused_wide.row_map) and the per-point kernel sums a point's consecutive rows in ascending order; every other
    cross-workgroup sum is an integer accumulator set or a fixed-order fold.  Three runs of the B=32 block give torch.equal
    outputs, input gradients and parameter gradients, with the index stage's maps handed in or built in line; and the
    gradients scale with the upstream gradient (loss scaling): exactly for a power of two, to rounding otherwise."""
    import copy
    from adaptpoint_amd import fused_wide
    from adaptpoint_amd.set_abstraction import SetAbstraction
    torch.manual_seed(0)
    blk = SetAbstraction(32, 64, layers=2, stride=2, fused=True, use_res=True,
                         group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                         norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'}).to(dev)
    B = 32
    p = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=14)).to(dev)
    f = torch.from_numpy(GI.seeded_normal((B, 32, 1024), seed=15)).to(dev)
    w1 = torch.from_numpy(GI.seeded_normal((B, 64, 512), seed=16)).to(dev)

    def run(scale):
        b2 = copy.deepcopy(blk)
        fa = f.clone().requires_grad_(True)
        smp = b2.sample(p)
        if with_map:
            smp.tmap = fused_wide.tile_map(smp.idx)
            smp.rowmap = fused_wide.row_map(smp.tmap, B, 1024, 512)
        _, o = b2([p, fa], sampling=smp)
        (o * (w1 * scale)).sum().backward()
        return [o.detach(), fa.grad] + [q.grad for q in b2.parameters()] + [x.clone() for x in b2.buffers()]

    runs = [run(loss_scale) for _ in range(3)]
    for other in runs[1:]:
        for k, (a, b) in enumerate(zip(runs[0], other)):
            assert torch.equal(a, b), k
    ref = run(1.0)
    npar = 1 + len(list(blk.parameters()))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
    assert torch.equal(runs[0][0], ref[0])                       # the forward does not see the upstream gradient
    for k, (a, b) in enumerate(zip(runs[0][1:1 + npar], ref[1:1 + npar])):
        assert torch.isfinite(a).all(), k
        if loss_scale == 65536.0:
            assert torch.equal(a, b * loss_scale), k             # a power of two: the same bits, shifted
        else:
            # (the upstream gradient is an MFMA operand: its hi + lo split rounds at 2^-17 relative, differently after scaling)
            assert rel(a, b * loss_scale) <= 2e-5, (k, rel(a, b * loss_scale))


def test_non_finite_upstream_gradient_is_loud(dev):
    """A non-finite upstream gradient reaches the gradients as NaN (the accumulator sets flag it), never as a wrong number."""
    from adaptpoint_amd import fused
    p, new_p, f, idx, conv1, bn1, conv2, bn2 = _setup(dev, B=2)
    f = f.clone().requires_grad_(True)
    out = fused.grouped_mlp_max(p, new_p, f, idx, 0.15, conv1, bn1, conv2, bn2)
    g = torch.ones_like(out)
    g[1, 3, 7] = float("inf")
    out.backward(g)
    assert torch.isnan(f.grad).any()


@pytest.mark.parametrize("structure", ["ball", "random"])
def test_row_map_is_the_point_sorted_order_of_the_tile_map_rows(dev, structure):
    """fused_wide.row_map (csrc/sa_wide_glue.hip: apn_sa_rowmap_many), index-stage data of the backward pass: every live
    row of the tile map has a place; the places of a cloud's rows fill a prefix of the cloud's own range of row ids; a
    point's rows hold consecutive places, in ascending row order; pcnt / poff say how many and where.  Against numpy, for
    ball-query rows and for arbitrary ones; stacked maps (`row_maps`) equal the maps one by one."""
    from adaptpoint_amd import _lib, fused_wide
    B, N, M = 6, 1024, 512
    p, new_p, f, idx, *_ = _setup(dev, B=B, seed=21)
    if structure == "random":
        idx = torch.randint(0, N, (B, M, 32), generator=torch.Generator().manual_seed(3), dtype=torch.int32).to(dev)
    tmap = fused_wide.tile_map(idx)
    pcnt_poff, rowdst = fused_wide.row_map(tmap, B, N, M)
    torch.cuda.synchronize()
    t = tmap.cpu().numpy()
    BM = B * M
    tiles = int(t[0])
    off = 4 + ((BM + 3) & ~3)
    info = t[off:off + 32 * BM].view(np.uint32)[:32 * tiles]
    rownn = t[off + 32 * BM:off + 64 * BM][:32 * tiles]
    tq0 = t[4:4 + BM]
    live = ((info >> 16) & 0xff) != 0
    rows = np.nonzero(live)[0]
    q = tq0[rows >> 5] + (info[rows] & 0xff).astype(np.int64)
    pt = (q // M) * N + rownn[rows]
    pc, po = pcnt_poff[:B * N].cpu().numpy(), pcnt_poff[B * N:2 * B * N].cpu().numpy()
    assert pcnt_poff.numel() == _lib.load().apn_sa_rowmap_ints(B, N) == 2 * B * N + B
    assert (pcnt_poff[2 * B * N:] == 1).all()                       # picks not handed in: not known to be distinct
    rd = rowdst.cpu().numpy()
    assert np.array_equal(pc, np.bincount(pt, minlength=B * N))
    order = np.lexsort((rows, pt))                                  # by point, then by row
    want_place = np.empty(len(rows), dtype=np.int64)
    # a cloud's places start at the cloud's first row id
    first_row_of_cloud = {}
    for c in range(B):
        rc = rows[(q // M) == c]
        first_row_of_cloud[c] = int(rc.min()) & ~31 if len(rc) else 0
    pos_in_cloud = np.zeros(B, dtype=np.int64)
    for k in order:
        c = int(q[k] // M)
        want_place[k] = first_row_of_cloud[c] + pos_in_cloud[c]
        pos_in_cloud[c] += 1
    assert _lib.load().apn_sa_rowmap_places(B, N, M) == 32 * B * M
    assert np.array_equal(rd[rows], want_place)
    has = pc > 0
    first = np.full(B * N, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, pt, want_place)
    assert np.array_equal(po[has], first[has])
    # stacked: three maps in one launch == one by one
    idx3 = torch.cat([idx, idx.flip(0), idx.roll(1, 0)])
    maps = fused_wide.tile_maps(idx3, 3)
    pp, rr = fused_wide.row_maps(maps, 3, B, N, M)
    torch.cuda.synchronize()
    for z in range(3):
        a, b_ = fused_wide.row_map(fused_wide.tile_map(idx3[z * B:(z + 1) * B].contiguous()), B, N, M)
        t_z = int(maps[z][0])
        assert torch.equal(pp[z], a)
        tz = maps[z].cpu().numpy()
        lz = ((tz[off:off + 32 * BM].view(np.uint32)[:32 * t_z] >> 16) & 0xff) != 0
        assert np.array_equal(rr[z].cpu().numpy()[:32 * t_z][lz], b_.cpu().numpy()[:32 * t_z][lz])


def test_skip_branch_gradient_rows_are_stored_where_the_picks_are_distinct(dev):
    """The row map's per-cloud verdict on the FPS picks (csrc/sa_wide_glue.hip: 0 = m different points) lets the backward
    entry kernel STORE the residual branch's gradient rows instead of adding them with float atomics (csrc/sa_glue.hip):
    the verdict is right -- also for a cloud with fewer distinct points than picks, whose sampler returns a point twice --,
    stored rows are the added rows bit for bit where every row has one share, and a collapsed cloud keeps the atomics and
    its shares (against the same step with every cloud on the atomic path, to float rounding: the order of two adds)."""
    import copy
    from adaptpoint_amd import fused_wide
    from adaptpoint_amd.set_abstraction import SetAbstraction
    torch.manual_seed(0)
    blk = SetAbstraction(32, 64, layers=2, stride=2, fused=True, use_res=True,
                         group_args={'NAME': 'ballquery', 'radius': 0.15, 'nsample': 32, 'normalize_dp': True},
                         norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'}).to(dev)
    B, N, M = 6, 1024, 512
    pn = GI.unit_sphere_cloud(B, N, seed=31)
    pn[2, 300:] = pn[2, :N - 300][:N - 300][np.arange(N - 300) % 300]      # cloud 2: 300 distinct points, 512 picks
    p = torch.from_numpy(pn).to(dev)
    f = torch.from_numpy(GI.seeded_normal((B, 32, N), seed=32)).to(dev)
    w = torch.from_numpy(GI.seeded_normal((B, 64, M), seed=33)).to(dev)

    def run(clouds, verdict_from_picks):
        b2 = copy.deepcopy(blk)
        fa = f[clouds].clone().requires_grad_(True)
        pc = p[clouds].contiguous()
        smp = b2.sample(pc)
        smp.tmap = fused_wide.tile_map(smp.idx)
        smp.rowmap = fused_wide.row_map(smp.tmap, len(clouds), N, M, fidx=smp.fidx if verdict_from_picks else None)
        _, o = b2([pc, fa], sampling=smp)
        (o * w[clouds]).sum().backward()
        return smp, [o.detach(), fa.grad] + [q.grad for q in b2.parameters()]

    every = list(range(B))
    smp, stored = run(every, True)
    picks = smp.fidx.cpu().numpy()
    twice = np.array([len(np.unique(picks[c])) < M for c in range(B)])
    assert twice.tolist() == [False, False, True, False, False, False]
    assert smp.rowmap[0][2 * B * N:].cpu().numpy().tolist() == twice.astype(np.int32).tolist()
    _, added = run(every, False)
    for k, (a, b) in enumerate(zip(stored, added)):
        assert float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)) <= 1e-6, k
    sound = [0, 1, 3, 4, 5]                                           # without the collapsed cloud: the same bits
    _, stored = run(sound, True)
    _, added = run(sound, False)
    for k, (a, b) in enumerate(zip(stored, added)):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("kind", ["ball", "random", "partial_fill"])
def test_point_geo_equals_its_statement(dev, oracle, kind):
    """The index stage's occurrence statistics (csrc/sa_geo.hip): counts and fixed-point sums of the relative
    positions are INTEGERS -- bit-exact against the numpy statement for ball-query rows (fill run folded), random
    rows and rows with a partial fill structure; the second moments to float64 rounding."""
    from adaptpoint_amd import _lib, fused
    B, N, M = 3, 1024, 500
    p, new_p, f, idx, *_ = _setup(dev, B=B, seed=11)
    new_p, idx = new_p[:, :M].contiguous(), idx[:, :M].contiguous()
    g = torch.Generator().manual_seed(5)
    if kind == "random":
        idx = torch.randint(0, N, (B, M, 32), generator=g, dtype=torch.int32).to(dev)
    elif kind == "partial_fill":
        idx = idx.clone()
        rnd = torch.randint(0, N, (B, M, 32), generator=g, dtype=torch.int32).to(dev)
        idx[:, ::3, 5:9] = rnd[:, ::3, 5:9]            # breaks the fill structure of every third row
        idx[:, 1::3, 3] = idx[:, 1::3, 0]              # slot 0 repeated in the middle of a row
    geo, dd = fused.point_geo(p, new_p, idx, 0.15)
    torch.cuda.synchronize()
    sums = dd.view(B, -1, 6).sum((0, 1))
    o_geo, o_dd = oracle.point_geo(p.cpu().numpy(), new_p.cpu().numpy(), idx.cpu().numpy(), 0.15)
    assert np.array_equal(geo.cpu().numpy(), o_geo)
    # (float32 partial sums per workgroup: the off-diagonal moments cancel to ~1e-3 of the diagonal ones)
    np.testing.assert_allclose(sums[:6].cpu().numpy(), o_dd, rtol=1e-6, atol=1e-8 * float(np.abs(o_dd).max()))
    # a second run gives the same bits (integer accumulation)
    geo2, dd2 = fused.point_geo(p, new_p, idx, 0.15)
    assert torch.equal(geo, geo2) and torch.equal(dd, dd2)
    assert _lib.load().apn_sa_geo_dd_doubles(N) == dd.shape[1] == 6


@pytest.mark.parametrize("prec", [2, 1])
def test_per_point_statistics_equal_the_pass_over_the_positions(dev, prec):
    """apn_sa_prep_stats: BatchNorm-1's batch sums from the occurrence statistics (no pass over the positions)
    against sum / sum of squares of the materialised y1 = conv1(cat[dp, f[idx]]) in float64, operands rounded as
    the MFMA sees them; and the operand table(s) it writes."""
    from adaptpoint_amd import _lib, fused
    from fused_reference import bf16_round
    B, N, M = 4, 1024, 512
    p, new_p, f, idx, conv1, *_ = _setup(dev, B=B, seed=2)
    lib = _lib.load()
    geo, dd = fused.point_geo(p, new_p, idx, 0.15)
    rows = lib.apn_sa_prep_rows(B, N)
    ft = torch.empty(prec * B * N * 32, dtype=torch.bfloat16, device=dev)
    part = torch.zeros(rows, 64, device=dev)
    acc = torch.ones(lib.apn_sa_acc_words(128), dtype=torch.int64, device=dev)
    w1 = conv1.weight.detach().view(32, 35).contiguous()
    fused._call("apn_sa_prep_stats", dev, B, N, f.data_ptr(), geo.data_ptr(), dd.data_ptr(), w1.data_ptr(), prec, 1,
                ft.data_ptr(), part.data_ptr(), acc.data_ptr(), acc.numel())
    torch.cuda.synchronize()
    assert int(acc.abs().sum()) == 0                                   # the next launch's accumulator set is cleared

    def eff(t):
        hi = bf16_round(t)
        return hi if prec == 1 else hi + bf16_round(t - hi)
    ft_hi = ft[:B * N * 32].view(B, N, 32).float()
    assert torch.equal(ft_hi, bf16_round(f.transpose(1, 2)))
    if prec == 2:
        assert torch.equal(ft[B * N * 32:].view(B, N, 32).float(), bf16_round(f.transpose(1, 2) - ft_hi))
    li = idx.long()
    fj = torch.gather(eff(f).double().unsqueeze(2).expand(-1, -1, M, -1), 3, li.unsqueeze(1).expand(-1, 32, -1, -1))
    pj = torch.gather(p.unsqueeze(1).expand(-1, M, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, 3))
    d = ((pj - new_p.unsqueeze(2)) / 0.15).permute(0, 3, 1, 2).double()                  # (B,3,M,K), unrounded: see below
    y1 = torch.einsum("oc,bcmk->bomk", eff(w1).double(), torch.cat([d, fj], 1))
    got = part.double().sum(0)
    s1, s2 = y1.sum((0, 2, 3)), (y1 * y1).sum((0, 2, 3))
    # the kernel sums the exact relative positions (2^-36 fixed point) where the MFMA rounds each to the operand
    # precision: a 2^-17 (split) / 2^-9 (bf16) relative perturbation of three of the 35 inputs, random in sign
    tol = 2e-6 if prec == 2 else 2e-4
    e1 = float(((got[:32] - s1).abs() / s2.sqrt().clamp_min(1e-9) / (B * M * 32) ** 0.5).max())
    e2 = float(((got[32:] - s2).abs() / s2).max())
    print("per-point statistics: sum %.2e (of sigma sqrt(P)), sumsq %.2e relative" % (e1, e2))
    assert e1 <= tol and e2 <= tol


def test_headline_block_at_the_bench_configuration_matches_the_float64_chain(dev):
    """The kernels `bench.py`'s `value` is quoted on, at ITS grid: B = 32 clouds, the index stage handed in as a
    `Sampling` with tile map and occurrence statistics, forward + backward replayed from a hipGraph -- against the
    float64 chain (tests/fused_reference.py + the block's skip branch and ReLU) on every output and every gradient.
    Bars as in the small-batch tests (split operands vs the fp32 chain): outputs max 2e-3 / mean 2e-5; gradients
    relative L2 5e-3 where they pass the pool's arg-max or BatchNorm-1's ReLU gates, 1e-4 where they bypass them."""
    import bench as BN
    from fused_reference import chain_grad
    torch.manual_seed(0)
    blk = BN.make_block(fused=True).to(dev).train()
    with torch.no_grad():
        for bn in (blk.convs[0][1], blk.convs[1][1]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    p, f = BN.make_inputs(32, seed=0)
    p, f = p.to(dev), f.to(dev).requires_grad_(True)
    smp = blk.sample(p)
    blk.index_for(smp, 1024, 32)
    assert smp.tmap is not None and smp.geo is not None and int(smp.tmap[0]) < 32 * 512 // 2
    params = list(blk.parameters())
    # float64 reference of the whole block (forward first: the loss weights are zero where the final ReLU's input lies
    # within 1e-3 of its kink, so that the gradients that bypass the pool's arg-max and BatchNorm-1's ReLU -- the
    # skip branch's and BatchNorm-2's -- can be held to 1e-4 instead of the gates' own 1e-3)
    conv1, bn1, conv2, bn2 = blk.convs[0][0], blk.convs[0][1], blk.convs[1][0], blk.convs[1][1]
    skip = blk.skipconv[0]
    leaves = {k: q.detach().clone().requires_grad_(True) for k, q in blk.named_parameters()}
    rf = f.detach().clone().requires_grad_(True)
    pooled, _ = chain_grad(p, smp.new_p, rf, smp.idx, BN.RADIUS, leaves["convs.0.0.weight"].view(32, 35),
                           leaves["convs.0.1.weight"], leaves["convs.0.1.bias"], leaves["convs.1.0.weight"].view(64, 32),
                           leaves["convs.1.1.weight"], leaves["convs.1.1.bias"])
    fsel = torch.gather(rf.double(), 2, smp.fidx.long().unsqueeze(1).expand(-1, 32, -1))
    ident = torch.einsum("oc,bcm->bom", leaves["skipconv.0.weight"].double().view(64, 32), fsel) \
        + leaves["skipconv.0.bias"].double().view(1, -1, 1)
    pre = pooled + ident
    ref = torch.relu(pre)
    wts = torch.randn(32, 64, 512, device=dev, generator=torch.Generator(dev).manual_seed(1))
    wts = wts * (pre.detach().abs() > 1e-3).float()
    (ref * wts.double()).sum().backward()

    def step():
        for q in params:
            q.grad = None
        f.grad = None
        _, out = blk([p, f], sampling=smp)
        torch.autograd.backward([out], [wts])
        return out
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    for _ in range(3):                                      # a replay refreshes out and the .grad tensors in place
        graph.replay()
    torch.cuda.synchronize()
    got = {"f": f.grad.clone(), **{k: q.grad.clone() for k, q in blk.named_parameters()}}
    e = (out.double() - ref).abs()
    print("headline block at B=32 (graph replay, tile map): out max %.2e mean %.2e" % (e.max(), e.mean()))
    assert e.max() <= 2e-3 and e.mean() <= 2e-5
    for b in (0, 11, 20, 31):                               # per cloud too: no cloud is special
        assert (out[b].double() - ref[b]).abs().max() <= 2e-3
    want = {"f": rf.grad, **{k: q.grad for k, q in leaves.items()}}
    l2 = {k: _rel_l2(got[k], want[k].reshape(got[k].shape)) for k in got}
    print("gradients, relative L2:", {k: "%.1e" % v for k, v in l2.items()})
    for k, v in l2.items():
        bypass = k in ("convs.1.1.weight", "convs.1.1.bias", "skipconv.0.weight", "skipconv.0.bias")
        assert v <= (1e-4 if bypass else 5e-3), (k, v)
    assert conv1.weight.grad is not None and bn1.weight.grad is not None and conv2 is not None and bn2 is not None and skip is not None
