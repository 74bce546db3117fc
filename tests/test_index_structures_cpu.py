"""CPU checks of the oracle's statements of the index-stage structures (oracle.tile_map / inverse_map): the
GPU builders are compared with them bit for bit in tests/test_gpu_fused_wide.py, so the statements themselves
are checked here against the property they define -- rows x multiplicity give back every query's 32 slots."""
import numpy as np
import pytest

import golden_inputs as GI
from oracle import oracle as O


def _ball_idx(B, N, M, radius, seed):
    xyz = GI.unit_sphere_cloud(B, N, seed=seed)
    fidx = O.furthest_point_sampling(xyz, M)
    new_xyz = np.take_along_axis(xyz, fidx[..., None].astype(np.int64), axis=1)
    return O.ball_query(radius, 32, xyz, new_xyz), new_xyz


@pytest.mark.parametrize("kind", ["ball", "random", "nofold"])
def test_tile_map_statement_reconstructs_every_slot(kind):
    B, N, M = 2, 256, 100
    idx, new_xyz = _ball_idx(B, N, M, 0.25, seed=5)
    if kind == "random":
        rng = np.random.default_rng(1)
        idx = rng.integers(0, N, (B, M, 32)).astype(np.int32)
        idx[:, ::3, 5:] = idx[:, ::3, :1]
    tm = O.tile_map(idx, fold=kind != "nofold")
    flat = idx.reshape(B * M, 32)
    seen = [[] for _ in range(B * M)]
    prev_last = -1
    for t in range(tm["nt"]):
        qs = []
        for r in range(32):
            info = int(tm["rowinfo"][t, r])
            mult = (info >> 16) & 0xff
            if mult == 0:
                assert (info & 0xff) == 0xff                       # padding rows belong to no query
                continue
            q = int(tm["tq0"][t]) + (info & 0xff)
            assert tm["rownn"][t, r] == flat[q, (info >> 8) & 0xff]
            seen[q] += [int(tm["rownn"][t, r])] * mult
            qs.append(q)
        assert qs == sorted(qs) and qs[0] == prev_last + 1             # whole queries, in order, no gaps
        assert len(set(q // M for q in qs)) == 1                       # a tile never spans two clouds
        assert (int(tm["rowinfo"][t, 0]) >> 24) == len(set(qs))
        prev_last = qs[-1]
    assert prev_last == B * M - 1
    for q in range(B * M):
        assert sorted(seen[q]) == sorted(flat[q].tolist())
    if kind == "ball":
        assert tm["nt"] < B * M                                        # the fill copies do fold
    if kind == "nofold":
        assert tm["nt"] == B * M


def test_inverse_map_statement_counts_every_position():
    B, N, M = 2, 256, 100
    idx, new_xyz = _ball_idx(B, N, M, 0.25, seed=6)
    tm = O.tile_map(idx)
    inv = O.inverse_map(tm, new_xyz, N, M)
    occ = np.zeros(B * N)
    sp = np.zeros((B * N, 3))
    for b in range(B):
        for q in range(M):
            for k in range(32):
                occ[b * N + idx[b, q, k]] += 1
                sp[b * N + idx[b, q, k]] += new_xyz[b, q]
    for gn in range(B * N):
        assert inv["occ"].get(gn, 0) == occ[gn]
        np.testing.assert_allclose(inv["sp"].get(gn, np.zeros(3)), sp[gn], rtol=1e-9, atol=1e-9)
        rows = inv["lists"].get(gn, [])
        assert rows == sorted(rows)
