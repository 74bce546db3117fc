"""CPU suite: the oracle against the committed goldens, against an independent
pure-Python restatement on tiny inputs, and across its four distance roundings."""
import numpy as np
import pytest

import golden_inputs as GI


def _bitrev(x, bits):
    r = 0
    for i in range(bits):
        if (x >> i) & 1:
            r |= 1 << (bits - 1 - i)
    return r


def py_fps(xyz, m):
    """Independent restatement of the reference semantics (sampling_gpu.cu:101-215)
    through the closed-form tie rule: among equal maxima the winner has the smallest
    (bit-reversed thread id, k) -- SURVEY.md section 8a-3 -- instead of emulating the tree."""
    n = xyz.shape[0]
    bs = min(1 << (n.bit_length() - 1), 1024)
    bits = bs.bit_length() - 1
    temp = np.full(n, np.float32(1e10), np.float32)
    out = [0]
    for _ in range(1, m):
        c = xyz[out[-1]]
        d = xyz - c
        # pinned rounding fma(dz,dz, fma(dx,dx, dy*dy)) emulated in float64 then rounded per op
        t = (d[:, 1].astype(np.float64) * d[:, 1].astype(np.float64)).astype(np.float32)
        t = (d[:, 0].astype(np.float64) * d[:, 0].astype(np.float64) + t.astype(np.float64)).astype(np.float32)
        t = (d[:, 2].astype(np.float64) * d[:, 2].astype(np.float64) + t.astype(np.float64)).astype(np.float32)
        temp = np.minimum(temp, t)
        mx = temp.max()
        cand = np.nonzero(temp == mx)[0]
        out.append(int(min(cand, key=lambda p: (_bitrev(int(p) % bs, bits), int(p) // bs))))
    return np.array(out, np.int32)


def py_ball(r, k, xyz, q):
    r2 = np.float32(np.float32(r) * np.float32(r))
    out = np.zeros((q.shape[0], k), np.int32)
    for i, c in enumerate(q):
        d = c - xyz
        t = (d[:, 1].astype(np.float64) ** 2).astype(np.float32)
        t = (d[:, 0].astype(np.float64) ** 2 + t.astype(np.float64)).astype(np.float32)
        t = (d[:, 2].astype(np.float64) ** 2 + t.astype(np.float64)).astype(np.float32)
        hits = np.nonzero(t < r2)[0][:k]
        if len(hits):
            out[i, :] = hits[0]
            out[i, :len(hits)] = hits
    return out


def test_opt_n_threads(oracle):
    for n, want in [(1, 1), (2, 2), (3, 2), (63, 32), (64, 64), (1000, 512), (1024, 1024),
                    (1200, 1024), (2048, 1024), (40960, 1024)]:
        assert oracle.opt_n_threads(n) == want


def test_goldens_reproduce(oracle, golden):
    xyz = GI.config1_xyz()
    fps = oracle.furthest_point_sampling(xyz, 512)
    assert np.array_equal(fps, golden["g1_fps512"])
    q = GI.take_points(xyz, fps)
    assert np.array_equal(oracle.ball_query(0.15, 32, xyz, q), golden["g1_bq_r015"])
    fps2 = oracle.furthest_point_sampling(q, 256)
    assert np.array_equal(fps2, golden["g1_fps256"])
    assert np.array_equal(oracle.ball_query(0.15 * 1.5, 32, q, GI.take_points(q, fps2)), golden["g1_bq_r0225"])
    for name, cloud, m in GI.tie_cases():
        assert np.array_equal(oracle.furthest_point_sampling(cloud, m), golden[f"g3_fps_{name}"]), name
    d2, idx = oracle.three_nn(xyz, q)
    assert np.array_equal(idx, golden["g2_three_nn_idx"]) and np.array_equal(d2, golden["g2_three_nn_dist2"])


def test_index_goldens_hold_under_every_rounding(oracle, golden):
    """The CUDA compiler's contraction of dx*dx+dy*dy+dz*dz is not observable here;
    every committed index golden must be identical under all four roundings."""
    xyz = GI.config1_xyz()
    q = GI.take_points(xyz, golden["g1_fps512"])
    for v in oracle.ALL_DIST_VARIANTS:
        assert np.array_equal(oracle.furthest_point_sampling(xyz, 512, v), golden["g1_fps512"])
        assert np.array_equal(oracle.ball_query(0.15, 32, xyz, q, v), golden["g1_bq_r015"])
        assert np.array_equal(oracle.three_nn(xyz, q, v)[1], golden["g2_three_nn_idx"])
        for name, cloud, m in GI.tie_cases():
            assert np.array_equal(oracle.furthest_point_sampling(cloud, m, v), golden[f"g3_fps_{name}"])


@pytest.mark.parametrize("n,m", [(64, 20), (100, 100), (150, 60), (37, 37), (1030, 12), (2050, 9)])
def test_fps_tree_emulation_equals_closed_form_tie_rule(oracle, n, m):
    g = np.round(GI.seeded_uniform((1, n, 3), seed=n) * 2).astype(np.float32) / 2   # heavy ties
    assert np.array_equal(oracle.furthest_point_sampling(g, m)[0], py_fps(g[0], m))
    x = GI.seeded_uniform((1, n, 3), seed=n + 1).astype(np.float32)
    assert np.array_equal(oracle.furthest_point_sampling(x, m)[0], py_fps(x[0], m))


def test_ball_query_vs_python(oracle):
    xyz = GI.seeded_uniform((2, 300, 3), seed=3).astype(np.float32)
    q = GI.seeded_uniform((2, 40, 3), seed=4).astype(np.float32)
    for r, k in [(0.3, 8), (0.05, 4), (2.0, 16), (0.5, 400)]:
        got = oracle.ball_query(r, k, xyz, q)
        for b in range(2):
            assert np.array_equal(got[b], py_ball(r, k, xyz[b], q[b]))


def test_group_gather_and_adjoints(oracle):
    rng = np.random.default_rng(0)
    f = rng.standard_normal((2, 5, 50), dtype=np.float32)
    idx = rng.integers(0, 50, (2, 7, 4), dtype=np.int32)
    out = oracle.group_points(f, idx)
    for b in range(2):
        assert np.array_equal(out[b], f[b][:, idx[b]])
    g = rng.standard_normal(out.shape, dtype=np.float32)
    gp = oracle.group_points_grad(g, idx, 50)
    assert np.isclose((out.astype(np.float64) * g).sum(), (f.astype(np.float64) * gp).sum(), rtol=1e-5)
    i1 = rng.integers(0, 50, (2, 9), dtype=np.int32)
    ga = oracle.gather_points(f, i1)
    for b in range(2):
        assert np.array_equal(ga[b], f[b][:, i1[b]])
    g1 = rng.standard_normal(ga.shape, dtype=np.float32)
    gp1 = oracle.gather_points_grad(g1, i1, 50)
    assert np.isclose((ga.astype(np.float64) * g1).sum(), (f.astype(np.float64) * gp1).sum(), rtol=1e-5)


def test_three_nn_and_interpolate(oracle):
    rng = np.random.default_rng(1)
    u = rng.standard_normal((2, 30, 3), dtype=np.float32)
    kn = rng.standard_normal((2, 11, 3), dtype=np.float32)
    d2, idx = oracle.three_nn(u, kn)
    full = ((u[:, :, None].astype(np.float64) - kn[:, None].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(full, axis=-1, kind="stable")[..., :3]
    assert np.array_equal(idx, order.astype(np.int32))
    assert np.allclose(d2, np.take_along_axis(full, order, -1), rtol=1e-6)
    assert (np.diff(d2, axis=-1) >= 0).all()
    w = GI.three_nn_weights(d2)
    p = rng.standard_normal((2, 6, 11), dtype=np.float32)
    out = oracle.three_interpolate(p, idx, w)
    want = sum(w[:, None, :, j] * np.take_along_axis(p, np.broadcast_to(idx[:, None, :, j], (2, 6, 30)).astype(np.int64), 2) for j in range(3))
    assert np.allclose(out, want, rtol=1e-5, atol=1e-6)
    assert np.allclose(oracle.three_interpolate(p, idx, w, fused=False), out, rtol=1e-5, atol=1e-6)
    g = rng.standard_normal(out.shape, dtype=np.float32)
    gp = oracle.three_interpolate_grad(g, idx, w, 11)
    assert np.isclose((out.astype(np.float64) * g).sum(), (p.astype(np.float64) * gp).sum(), rtol=1e-4)
    # m < 3: unused slots are +inf / 0 (interpolate_gpu.cu:37-38)
    d2s, idxs = oracle.three_nn(u, kn[:, :1])
    assert np.isinf(d2s[..., 1:]).all() and (idxs[..., 1:] == 0).all()


def test_empty_and_degenerate(oracle):
    xyz = GI.seeded_uniform((1, 10, 3), seed=9).astype(np.float32)
    assert oracle.furthest_point_sampling(xyz, 0).shape == (1, 0)
    assert np.array_equal(oracle.furthest_point_sampling(xyz, 1), np.zeros((1, 1), np.int32))
    assert (oracle.ball_query(1e-6, 4, xyz, xyz + 5) == 0).all()


def test_pointset_group_oracle_reproduces_reference_module(golden, oracle):
    """oracle FPS + ball query + numpy group-max == the reference's
    PointsetGrouper (generator_component4_15.py:394-431) run in the build container (G6)."""
    import golden_inputs as GI
    xyz = GI.unit_sphere_cloud(2, 512, seed=63)
    pts = GI.seeded_normal((2, 512, 64), seed=64)
    alpha = GI.seeded_normal((1, 1, 1, 64), seed=61).reshape(-1)
    beta = GI.seeded_normal((1, 1, 1, 64), seed=62).reshape(-1)
    fidx = oracle.furthest_point_sampling(xyz, 256)
    new_xyz = GI.take_points(xyz, fidx)
    assert np.array_equal(new_xyz, golden["g6_pg_new_xyz"])
    idx = oracle.ball_query(0.2, 24, xyz, new_xyz)
    out, ksel = oracle.pointset_group_max(pts, idx, fidx, alpha, beta)
    assert np.array_equal(out, golden["g6_pg_out"])
    w = GI.seeded_normal(tuple(out.shape), seed=65)
    gp, ga, gb = oracle.pointset_group_max_grad(pts, idx, fidx, alpha, ksel, w)
    assert np.abs(gp - golden["g6_pg_grad_points"]).max() < 1e-5
    assert np.abs(ga - golden["g6_pg_grad_alpha"].reshape(-1)).max() < 1e-4
    assert np.abs(gb - golden["g6_pg_grad_beta"].reshape(-1)).max() < 1e-4


def _pins():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pins_golden.npz"))


def _fp_levels(seed=151):
    from oracle import oracle as O
    cloud = GI.unit_sphere_cloud(2, 1024, seed=seed)
    levels = [cloud]
    for m in (512, 256, 128, 64):
        levels.append(GI.take_points(levels[-1], O.furthest_point_sampling(levels[-1], m)))
    return [(levels[i], levels[i + 1]) for i in range(4)]


def three_nn_cases():
    """(name, unknown, known) of G15: BASELINE config 1 and the four feature-propagation levels at N = 1024."""
    from oracle import oracle as O
    xyz = GI.config1_xyz()
    cases = [("cfg1", xyz, GI.take_points(xyz, O.furthest_point_sampling(xyz, 512)))]
    return cases + [(f"fp{i}", u, k) for i, (u, k) in enumerate(_fp_levels())]


def test_interpolation_half_of_the_oracle_is_pinned_by_reference_held_code():
    """G15 (tests/golden/make_golden.py pins): indices of the three nearest from the reference's `square_distance` +
    sort (curvenet.py:213-222, 447-449), the interpolation and its gradient from the reference's forward
    (curvenet.py:451-455) and autograd through it; the scatter-add gradients from autograd through
    `torch_grouping_operation` (group.py:120-137) and `torch.gather`.  The oracle reproduces all of them."""
    from oracle import oracle as O
    O.build()
    g = _pins()
    for name, unk, kn in three_nn_cases():
        safe = g[f"g15_{name}_safe"]
        assert safe.mean() > 0.97
        for v in O.ALL_DIST_VARIANTS:
            assert np.array_equal(O.three_nn(unk, kn, v)[1][safe], g[f"g15_{name}_idx"][safe]), (name, v)
    for name, unk, kn in (c for c in three_nn_cases() if c[0] in ("cfg1", "fp3")):
        c = 32 if name != "fp3" else 96
        pts = GI.seeded_normal((2, c, kn.shape[1]), seed=152)
        idx, w = g[f"g15_{name}_idx"], g[f"g15_{name}_weight"]
        np.testing.assert_allclose(O.three_interpolate(pts, idx, w), g[f"g15_{name}_interp"], rtol=1e-6, atol=1e-6)
        gout = GI.seeded_normal(tuple(g[f"g15_{name}_interp"].shape), seed=153)
        np.testing.assert_allclose(O.three_interpolate_grad(gout, idx, w, kn.shape[1]), g[f"g15_{name}_interp_grad"],
                                   rtol=1e-6, atol=1e-6)
    xyz = GI.config1_xyz()
    fps512 = O.furthest_point_sampling(xyz, 512)
    bq = O.ball_query(0.15, 32, xyz, GI.take_points(xyz, fps512))
    np.testing.assert_allclose(O.group_points_grad(GI.seeded_normal((2, 32, 512, 32), seed=12), bq, 1024),
                               g["g15_group_grad"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(O.gather_points_grad(GI.seeded_normal((2, 32, 512), seed=13), fps512, 1024),
                               g["g15_gather_grad"], rtol=1e-6, atol=1e-6)
