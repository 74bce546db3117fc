"""Debug aid: every output of the csrc/sa_wide.hip kernels against a float64 torch evaluation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as GI
from adaptpoint_amd import _lib
from adaptpoint_amd.fused import _call
from adaptpoint_amd.fused_wide import mfma_b_image, neighbour_index
from adaptpoint_amd.layers import ball_query, furthest_point_sample

dev = torch.device("cuda:0")
lib = _lib.load()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def check_map(idx, tmap, fold):
    """rows of the map, expanded by their multiplicity, must be each query's 32 slots (as a multiset)."""
    B, M, _ = idx.shape
    t = tmap.cpu().numpy().astype('int64')
    nt, bm = int(t[0]), B * M
    tq0 = t[4:4 + bm]
    rows = (t[4 + ((bm + 3) & ~3):][:32 * bm] & 0xffffffff).reshape(bm, 32)
    rnn = t[4 + ((bm + 3) & ~3) + 32 * bm:][:32 * bm].reshape(bm, 32)
    flat = idx.cpu().numpy().reshape(bm, 32)
    seen = [[] for _ in range(bm)]
    used = 0
    for tl in range(nt):
        nq = rows[tl, 0] >> 24
        qs = set()
        for r in range(32):
            info = int(rows[tl, r]); mult = (info >> 16) & 0xff
            if mult == 0:
                continue
            q = int(tq0[tl]) + (info & 0xff); slot = (info >> 8) & 0xff
            qs.add(q); used += 1
            assert rnn[tl, r] == flat[q, slot]
            seen[q] += [int(flat[q, slot])] * mult
        assert len(qs) == nq and max(qs) - min(qs) + 1 == nq, (tl, nq, qs)
        assert len({q // M for q in qs}) == 1
    for q in range(bm):
        assert sorted(seen[q]) == sorted(flat[q].tolist()), q
    return nt, used


CASES = [(H, N, M, radius, kind) for (H, N, M, radius) in ((32, 1024, 512, 0.15), (64, 512, 256, 0.225), (128, 256, 128, 0.34), (256, 128, 64, 0.5))
         for kind in ("ball", "ball-nofold", "random")]
for H, N, M, radius, kind in CASES:
    B, O = 2, 2 * H
    p = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=1)).to(dev)
    fidx = furthest_point_sample(p, M).long()
    new_p = torch.gather(p, 1, fidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = ball_query(radius, 32, p, new_p)
    if kind == "random":
        idx = torch.randint(0, N, (B, M, 32), device=dev, dtype=torch.int32, generator=torch.Generator(dev).manual_seed(3))
        idx[:, ::3, 5:] = idx[:, ::3, :1]            # some rows with the fill structure, some with repeats that are not
    nbr = neighbour_index(idx, new_p, N, fold=kind != "ball-nofold")
    tmap = nbr.tmap
    nt, used = check_map(idx, tmap, kind != "ball-nofold")
    print(f"[{kind}] tiles {nt} of {B * M}, rows {used}", end=" | ")
    g = torch.Generator(dev).manual_seed(0)
    U = torch.randn(B, N, H, device=dev, generator=g)
    V = 0.3 * torch.randn(B, M, H, device=dev, generator=g)
    scale1 = 0.5 + torch.rand(H, device=dev, generator=g)
    shift1 = 0.2 * torch.randn(H, device=dev, generator=g)
    mean1 = 0.1 * torch.randn(H, device=dev, generator=g)
    inv1 = 0.5 + torch.rand(H, device=dev, generator=g)
    pack1 = torch.cat([scale1, shift1, mean1, inv1]).contiguous()
    W2 = torch.randn(O, H, device=dev, generator=g) / H ** 0.5
    sgn2 = torch.where(torch.randn(O, device=dev, generator=g) > -0.5, 1.0, -1.0)
    grid = lib.apn_sa_wide_grid(B, M)
    # references (float64, materialised)
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    y1 = U.double()[bi, idx.long()] - V.double().unsqueeze(2)                      # (B,M,K,H)
    part1 = torch.empty(grid, 2 * H, device=dev)
    _call("apn_sa_wide_stats1", dev, B, N, M, H, U.data_ptr(), V.data_ptr(), idx.data_ptr(), tmap.data_ptr(), part1.data_ptr())
    s = part1.double().sum(0)
    print(f"H={H}: stats1 sum {rel(s[:H], y1.sum((0,1,2))):.1e} sumsq {rel(s[H:], (y1*y1).sum((0,1,2))):.1e}", end=" | ")
    a1 = torch.relu(y1 * scale1.double() + shift1.double())
    y2 = a1 @ W2.double().t()                                                       # (B,M,K,O)
    ysel = torch.empty(B, M, O, device=dev); ksel = torch.empty(B, M, O, dtype=torch.uint8, device=dev)
    part2 = torch.empty(grid, 2 * O, device=dev)
    w2img = mfma_b_image(W2.t().contiguous(), min(4, O // 32))
    _call("apn_sa_wide_fwd_main", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(), tmap.data_ptr(), w2img.data_ptr(),
          pack1.data_ptr(), sgn2.data_ptr(), ysel.data_ptr(), ksel.data_ptr(), part2.data_ptr())
    s2 = part2.double().sum(0)
    ext = (y2 * sgn2.double()).max(2)[0] * sgn2.double()
    at_k = torch.gather(y2, 2, ksel.long().unsqueeze(2)).squeeze(2)
    print(f"y2[ksel] {rel(at_k, ext):.1e}", end=" ")
    print(f"fwd ysel {rel(ysel, ext):.1e} sum {rel(s2[:O], y2.sum((0,1,2))):.1e} sumsq {rel(s2[O:], (y2*y2).sum((0,1,2))):.1e}", end=" | ")
    # backward
    goa = torch.randn(B, M, O, device=dev, generator=g)
    Qm = torch.randn(H, H, device=dev, generator=g) / H
    evec = 0.1 * torch.randn(H, device=dev, generator=g)
    S = torch.zeros(B, M, 32, O, dtype=torch.float64, device=dev)
    S.scatter_(2, ksel.long().unsqueeze(2), goa.double().unsqueeze(2))
    dA = S @ W2.double() + a1 @ Qm.double() + evec.double()
    gu = dA * (a1 > 0)
    yh = (y1 - mean1.double()) * inv1.double()
    A_ref = torch.zeros(B, N, H, dtype=torch.float64, device=dev)
    A_ref.scatter_add_(1, idx.long().view(B, M * 32, 1).expand(-1, -1, H), gu.view(B, M * 32, H))
    zimg = mfma_b_image(torch.cat([W2, Qm], 0).contiguous(), min(4, H // 32))
    GU = torch.full((B * M * 32, H), float("nan"), device=dev); HA = torch.empty(B, M, H, device=dev); HB = torch.empty(B, M, H, device=dev)
    partT = torch.empty(grid, 2 * H, device=dev)
    Rfused = torch.zeros(grid, (O + H) * H + H, device=dev)
    _call("apn_sa_wide_bwd_main", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(), tmap.data_ptr(), zimg.data_ptr(),
          pack1.data_ptr(), evec.data_ptr(), goa.data_ptr(), ksel.data_ptr(), GU.data_ptr(), HA.data_ptr(),
          HB.data_ptr(), partT.data_ptr(), Rfused.data_ptr())
    # the inverse map: every point's list is ascending, and the rows of a point sum to its scatter
    cnt, off = nbr.pcnt_poff[:B * N].long(), nbr.pcnt_poff[B * N:].long()
    owner = torch.repeat_interleave(torch.arange(B * N, device=dev), cnt)
    within = torch.arange(owner.numel(), device=dev) - torch.repeat_interleave(cnt.cumsum(0) - cnt, cnt)
    lrows = nbr.plist[torch.repeat_interleave(off, cnt) + within].long()
    same = owner[1:] == owner[:-1]
    assert bool((lrows[1:][same] > lrows[:-1][same]).all()), "lists not ascending"
    A = torch.zeros(B * N, H, dtype=torch.float64, device=dev).index_add_(0, owner, GU[lrows].double()).view(B, N, H)
    occ = torch.zeros(B, N, dtype=torch.float64, device=dev).scatter_add_(1, idx.long().view(B, -1), torch.ones(B, M * 32, dtype=torch.float64, device=dev))
    sp = torch.zeros(B, N, 3, dtype=torch.float64, device=dev).scatter_add_(
        1, idx.long().view(B, -1, 1).expand(-1, -1, 3), new_p.double().unsqueeze(2).expand(-1, -1, 32, -1).reshape(B, -1, 3))
    print(f"geo occ {rel(nbr.geo[..., 0], occ):.1e} sp {rel(nbr.geo[..., 1:], sp):.1e}", end=" ")
    T = partT.double().sum(0)
    print(f"bwd A {rel(A, A_ref):.1e} HA {rel(HA, gu.sum(2)):.1e} HB {rel(HB, yh.sum(2)):.1e} T1 {rel(T[:H], gu.sum((0,1,2))):.1e} "
          f"T2 {rel(T[H:], (gu*yh).sum((0,1,2))):.1e}", end=" | ")
    rows = O + H
    groups = (rows // 32 + 7) // 8
    splits = max(1, min(512 // groups, (B * M) // 4, (16 << 20) // (rows * H * 4)))
    Rpart = torch.empty(splits, rows * H + H, device=dev)
    _call("apn_sa_wide_wgrad", dev, B, N, M, H, O, U.data_ptr(), V.data_ptr(), idx.data_ptr(), tmap.data_ptr(), pack1.data_ptr(),
          goa.data_ptr(), ksel.data_ptr(), splits, Rpart.data_ptr())
    R = Rpart.double().sum(0)[:rows * H].view(rows, H)
    suma = Rpart.double().sum(0)[rows * H:]
    a1f = a1.view(-1, H)
    if lib.apn_sa_wide_wgrad_fused(H):
        Rf = Rfused.double().sum(0)
        print(f"fused-in-bwd sparse {rel(Rf[:O * H].view(O, H), S.view(-1, O).t() @ a1.view(-1, H)):.1e} gram "
              f"{rel(Rf[O * H:rows * H].view(H, H), R[O:]):.1e} suma {rel(Rf[rows * H:], suma):.1e}", end=" | ")
    print(f"wgrad sparse {rel(R[:O], S.view(-1, O).t() @ a1f):.1e} gram {rel(R[O:], a1f.t() @ a1f):.1e} suma {rel(suma, a1f.sum(0)):.1e}")
