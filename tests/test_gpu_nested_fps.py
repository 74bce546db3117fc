"""GPU: the nested sampler of the index pyramids (csrc/fps.hip, NEST; `layers.furthest_point_sample_nested`).  FPS is
progressive -- run on a sample's picks in pick order it returns picks 0, 1, 2, ... again whenever every arg-max was unique --
so a deeper level of a pyramid is a copy where the previous level's record allows and the full sampler elsewhere.  The bar
is the sampler's own: every level's picks equal the FULL sampler's (`furthest_point_sample`, the reference's semantics:
subsample.py:78-98 run per block / stage) bit for bit -- on generic clouds (where the copy path must actually be taken)
and on tie-heavy ones (duplicates, the AdaptPoint half-at-origin mask, lattices, dyadic grids: where it must not)."""
import numpy as np
import pytest
import torch

import golden_inputs as GI

pytestmark = pytest.mark.gpu
INT_MAX = 0x7fffffff


def _chain(xyz, levels):
    """levels = [m1, m2, ...]: nested chain and full-sampler chain over the same inputs -> per level (nested picks,
    full picks, record, input size)."""
    from adaptpoint_amd.layers import furthest_point_sample, furthest_point_sample_nested
    out, ties, cur = [], None, xyz
    for m in levels:
        picks, new_xyz, rec = furthest_point_sample_nested(cur, m, ties)
        full = furthest_point_sample(cur, m)
        assert torch.equal(new_xyz, torch.gather(cur, 1, full.long().unsqueeze(-1).expand(-1, -1, 3)))
        out.append((picks, full, rec, cur.shape[1]))
        ties, cur = rec, new_xyz
    return out


def test_generic_clouds_take_the_copy_path_and_equal_the_full_sampler(dev, oracle):
    xyz = torch.from_numpy(GI.unit_sphere_cloud(32, 1024, seed=41)).to(dev)
    res = _chain(xyz, [512, 256, 128, 64])
    assert np.array_equal(res[0][0].cpu().numpy(), oracle.furthest_point_sampling(xyz.cpu().numpy(), 512))
    for lvl, (picks, full, rec, n) in enumerate(res):
        assert torch.equal(picks, full), lvl
    # no exact ties in float-random clouds: the record says so, and levels 2 and 3 ARE the prefix (the copy path ran)
    assert int(res[0][2].min()) >= 512 and int(res[1][2].min()) >= 256
    ar = torch.arange(256, device=dev, dtype=torch.int32)
    assert torch.equal(res[1][0], ar.expand(32, -1)) and torch.equal(res[2][0], ar[:128].expand(32, -1))
    assert int(res[3][2].max()) == 0           # n = 128: outside the LDS-atomic step's range: full sampler, nothing recorded


@pytest.mark.parametrize("name", ["dup", "half_origin", "all_same", "lattice512", "n2048_grid", "n1000_grid", "n3000_grid", "n1200"])
def test_tie_heavy_clouds_equal_the_full_sampler_at_every_level(dev, name):
    cloud, m = next((c, m) for n_, c, m in GI.tie_cases() if n_ == name)
    xyz = torch.from_numpy(cloud).to(dev)
    n = xyz.shape[1]
    levels, cur = [], min(m, n)
    while cur >= 8 and len(levels) < 4:
        levels.append(cur)
        cur //= 2
    res = _chain(xyz, levels)
    for lvl, (picks, full, rec, n_in) in enumerate(res):
        assert torch.equal(picks, full), (name, lvl, n_in)


def test_record_is_never_later_than_the_first_repeated_maximum(dev):
    """Two mirror points at the same distance from the start tie at step 1; with the pair removed the cloud is generic.
    A level that needs more picks than the record allows must run the full sampler (its picks differ from the prefix
    whenever the positional tie rule picks the other mirror point)."""
    from adaptpoint_amd.layers import furthest_point_sample, furthest_point_sample_nested
    g = torch.Generator().manual_seed(3)
    base = (torch.rand(4, 1022, 3, generator=g) - 0.5) * 0.5
    start = torch.zeros(4, 1, 3)
    a = torch.tensor([2.0, 0.0, 0.0]).expand(4, 1, 3)
    xyz = torch.cat([start, base[:, :500], a, base[:, 500:], -a], 1).contiguous().to(dev)    # 1 + 500 + 1 + 522 + 1 = 1025
    xyz = xyz[:, :1024].contiguous() if xyz.shape[1] > 1024 else xyz
    xyz[:, 1023] = torch.tensor([-2.0, 0.0, 0.0], device=dev)
    picks, new_xyz, rec = furthest_point_sample_nested(xyz, 512, None)
    assert torch.equal(picks, furthest_point_sample(xyz, 512))
    assert int(rec.max()) == 1                                            # the very first arg-max is held twice
    p2, _, rec2 = furthest_point_sample_nested(new_xyz, 256, rec)
    assert torch.equal(p2, furthest_point_sample(new_xyz, 256))          # full sampler ran (record 1 < 256): still exact


def test_classifier_pyramid_and_imitator_chain_are_unchanged_by_the_nested_sampler(dev, monkeypatch):
    from adaptpoint_amd import imitator as IM, pointnext as PN
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    pos = torch.from_numpy(GI.unit_sphere_cloud(8, 1024, seed=43)).to(dev)
    half = pos.clone()
    half[:, ::2] = 0.0                                                    # the generator's masked clouds
    C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev).eval()
    for cloud in (pos, half):
        nested = [s for s in C.encoder.index_pyramid(cloud) if s is not None]
        from adaptpoint_amd import fused
        flat, p = [], cloud
        for stage in C.encoder.encoder:
            sa = stage[0]
            if sa.is_head or sa.all_aggr:
                continue
            smp = fused.sample_and_query(p, p.shape[1] // sa.stride, sa.grouper.radius, sa.grouper.nsample)
            flat.append(smp)
            p = smp.new_p
        for a, b in zip(nested, flat):
            assert torch.equal(a.fidx, b.fidx) and torch.equal(a.new_p, b.new_p) and torch.equal(a.idx, b.idx)
        x = torch.cat([cloud, cloud[:, :, 1:2]], -1).transpose(1, 2).contiguous()
        with torch.no_grad():
            monkeypatch.setattr(PN, "NESTED_PYRAMID", True)
            la = C({'pos': cloud, 'x': x})
            monkeypatch.setattr(PN, "NESTED_PYRAMID", False)
            lb = C({'pos': cloud, 'x': x})
        assert torch.equal(la, lb)
    from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise_on
    G = fill_parameters_by_name(AdaptPointAugmentor()).to(dev).eval()
    torch.manual_seed(5)
    noise = draw_noise_on(dev, 8, 1024, 4)
    with torch.no_grad():
        monkeypatch.setattr(IM, "NESTED_SAMPLING", True)
        _, ga = G(pos, noise)
        monkeypatch.setattr(IM, "NESTED_SAMPLING", False)
        _, gb = G(pos, noise)
    assert torch.equal(ga, gb)
