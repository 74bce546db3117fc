"""Fused set-abstraction block (csrc/sa_fused.hip, sa_glue.hip, fps.hip, ball_query.hip).

Two entry points over the same kernels:

`fused_set_abstraction(p, f, npoint, radius, conv1, bn1, conv2, bn2, skip_conv, relu)`
    the whole block of openpoints/models/backbone/pointnext.py:140-170 for the shapes the
    kernels support -- FPS (+ the gather of the sampled points), ball query, then

        dp  = (p[idx] - new_p) / radius ; fj = f[idx]                 group.py:248-254
        y1  = conv1(cat[dp, fj]) ; a1 = relu(bn1(y1))                 pointnext.py:119-128,166
        y2  = conv2(a1) ; o = max_K bn2(y2)                           pointnext.py:166
        out = relu(o + skip_conv(f[:, fps_idx]))                      pointnext.py:157-168

    returning (new_p, out);

`grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2)`
    only the grouped MLP + pool `o`, for callers that sample / query themselves.

No (B, C, M, K) tensor is materialised, forward or backward.  BatchNorm follows the
modules' training flag (batch statistics + running-buffer update, or the running
buffers); the MFMA contractions are bf16 x bf16 -> f32 accumulate on split (hi + lo) operands
by default (`PRECISION`), every statistic is an f32 partial summed exactly (integer accumulator sets) or in f64.
Forward is 3 kernel launches, backward 4 (round 2: 6 + 6: every BatchNorm fold / constants kernel is now a
prologue of its consumer, and BatchNorm-1's statistics come per POINT from the index stage's occurrence
statistics `Sampling.geo`), all on the current stream with no host reads, so a step can be captured in a HIP graph.
With sync_bn=True the per-channel float64 sums are all-reduced across ranks
(SyncBatchNorm semantics) at the four points where statistics leave the kernels.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from . import ops as _ops

C_IN, C_MID, C_OUT, K_NS = 32, 32, 64, 32
# The register-resident passes run over the distinct-hit tile map of the index stage (fused_wide.tile_map): 3.7x fewer
# tiles at stage 1.  The BACKWARD pass runs over it always (round 5): it stores the rows of g_u through the map's row map
# (fused_wide.row_maps: a row's place in the point-sorted order) and the per-point kernel sums a point's consecutive rows
# in ascending order -- no float atomics anywhere in the chain.  A caller that hands no index stage in gets both maps
# built in line (two + one launches).
# build the tile map in line for the FORWARD too when the caller hands no index stage in (training: forward + backward)
TILE_MAP_INLINE = os.environ.get("APN_TMAP_INLINE", "1") == "1"
# Bit-reproducible gradients are the ONLY mode since round 5 (every cross-workgroup sum is an integer accumulator set, a
# fixed-order fold of partial rows, or the ordered sum of a point's rows); rounds 3-4 had a separate, slower mode
# (64-bit fixed-point integer atomics for the per-point sums) behind this flag, which is kept as an accepted no-op.
DETERMINISTIC = True

# Operand precision of the MFMA contractions:
#   "bf16x3" (default) every f32 operand is split into hi + lo bf16 parts and each product is
#            three MFMAs (hi*hi + hi*lo + lo*hi): agrees with an fp32 chain to ~1e-5;
#   "bf16"   operands rounded to bf16 (8 significant bits): fastest, ~3e-3 mean deviation.
PRECISION = "bf16x3"
_PREC = {"bf16": 1, "bf16x3": 2}

# Diagnostics: True issues every kernel as its own foreign call (the Python mirror of
# csrc/sa_seq.hip below) so that per-kernel HIP events can be placed around them
# (bench.py's per-kernel table).  The product default is one C call per direction.
PER_KERNEL_LAUNCH = False


def supported(p, f, idx_or_k, conv1, conv2, bns=(), npoint=None):
    """Whether the fused kernels cover this block: the 32 -> 32 -> 64, K = 32 shape, float32 CUDA
    tensors, no more queries than support points (the backward walks query tiles alongside point
    tiles), a batch that fits one grid dimension, and BatchNorms with a fixed momentum (the
    cumulative-average mode `momentum=None` stays on the unfused path)."""
    k = idx_or_k.shape[2] if torch.is_tensor(idx_or_k) else int(idx_or_k)
    m = idx_or_k.shape[1] if torch.is_tensor(idx_or_k) else npoint
    return (f.is_cuda and f.dtype == torch.float32 and p.dtype == torch.float32
            and f.shape[1] == C_IN and k == K_NS
            and tuple(conv1.weight.shape[:2]) == (C_MID, C_IN + 3)
            and tuple(conv2.weight.shape[:2]) == (C_OUT, C_MID)
            and conv1.bias is None and conv2.bias is None
            and (m is None or m <= p.shape[1]) and p.shape[0] <= 65535
            and all(bn.momentum is not None for bn in bns))


_FN = {}


def _call(name, dev, *args, stream=None):
    """One launch: the C entry point `name` on PyTorch's current stream of `dev` (or on the
    given raw stream handle).  Assumes `dev` is the current device (callers of this module
    run under the tensors' device, as torch.nn modules do); other devices take the guard."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib.load(), name)
    if stream is None:
        if dev.index is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                code = fn(*args, torch.cuda.current_stream(dev).cuda_stream)
            if code:
                _lib.check(code, name)
            return
        stream = torch.cuda.current_stream(dev).cuda_stream
    code = fn(*args, stream)
    if code:
        _lib.check(code, name)


class _Launcher:
    """A run of launches on one stream: resolves the stream handle once (host time matters:
    an eager step is ~25 launches and must not become host-bound).  It calls through the
    module-level name `_call`, so instrumentation that wraps `fused._call` sees every launch."""

    def __init__(self, dev):
        self.dev = dev
        if dev.index is not None and dev.index != torch.cuda.current_device():
            self.stream = None               # let _call take the device guard
        else:
            self.stream = torch.cuda.current_stream(dev).cuda_stream

    def __call__(self, name, *args):
        _call(name, self.dev, *args, stream=self.stream)


def _ptr(t):
    return None if t is None else t.data_ptr()


# Testing hook: run the phased SyncBatchNorm path (reduce rows -> all_reduce -> consumer with
# sums) even when there is a single rank, so that it can be validated on one GPU.
FORCE_PHASED = False


def _world(sync):
    if sync and dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def _allreduce_sum_(t):
    """In-place sum over ranks (the one collective of the SyncBatchNorm exchange; a seam for tests)."""
    # (FORCE_PHASED: the collective is issued at world size 1 too, so that a one-GPU box runs the RCCL call itself)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_PHASED):
        from . import dp
        dp.all_reduce_sum_(t)


def _phased(sync):
    return _world(sync) > 1 or (FORCE_PHASED and sync)


def _carve(dev, sizes, zero=False):
    """One allocation, many float32 views: sizes = [(name, n_floats)] -> dict of 1-D views.
    Offsets are rounded to 64 floats (256 B) so every view is safely aligned."""
    offs, total = [], 0
    for _, nfl in sizes:
        offs.append(total)
        total += (int(nfl) + 63) // 64 * 64
    buf = (torch.zeros if zero else torch.empty)(max(total, 64), dtype=torch.float32, device=dev)
    return {name: buf[o:o + int(nfl)] for (name, nfl), o in zip(sizes, offs)}, buf


def _all_reduce_rows(call, part, rows, ncol, count, dev):
    """SyncBatchNorm exchange: float64 column sums of the partial rows + {this rank's position
    count, 1}, all-reduced -> {global sums, global count, world size}: the consumers read the
    count and the world size from the vector (ranks may hold different batch sizes)."""
    sums = torch.empty(ncol + 2, dtype=torch.float64, device=dev)
    call("apn_sa_reduce_rows", part.data_ptr(), rows, ncol, float(count), sums.data_ptr())
    _allreduce_sum_(sums)
    return sums


def _all_reduce_acc(call, acc, ncol, count, dev):
    """The same exchange for an accumulator set (csrc/apn_common.h) instead of partial rows."""
    sums = torch.empty(ncol + 2, dtype=torch.float64, device=dev)
    call("apn_sa_reduce_acc", acc.data_ptr(), ncol, float(count), sums.data_ptr())
    _allreduce_sum_(sums)
    return sums


def _bn_args(bn):
    """(gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, training)."""
    training = bn.training or not bn.track_running_stats
    track = bn.track_running_stats
    if bn.momentum is None:
        raise RuntimeError("fused set abstraction: BatchNorm(momentum=None) is not covered "
                           "(fused.supported() routes it to the unfused path)")
    mom = bn.momentum
    return (_ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean) if track else None,
            _ptr(bn.running_var) if track else None,
            _ptr(bn.num_batches_tracked) if (track and bn.training) else None,
            float(bn.eps), float(mom), 1 if training else 0)


def _mat(w, rows, cols):
    w = w.detach().reshape(rows, cols)
    return w if w.is_contiguous() else w.contiguous()


def _acc_floats(ncol):
    """float32 slots of an accumulator set of ncol columns (it holds 64-bit words)."""
    return 2 * _lib.load().apn_sa_acc_words(ncol)


class _Forward:
    """Runs the forward launches and keeps what the backward needs."""

    def __init__(self, p, f, new_p, idx, fidx, radius, conv1, bn1, conv2, bn2, skip_conv, relu,
                 sync_bn, tmap=None, geo=None, dd=None, want_backward=False, rowmap=None):
        """want_backward: the backward's atomically accumulated region (A | gip | accS | accT) is allocated now and
        cleared by the forward's last launch, so that the backward runs without a fill launch of its own.
        tmap: the distinct-hit tile map of idx (adaptpoint_amd.fused_wide.tile_map; index-stage work) -- the
        passes over the positions then run over ~1/4 of the tiles at stage 1; None: one tile per query in the forward
        (the backward builds one: it runs over a tile map always).  rowmap: (pcnt_poff, rowdst) of tmap
        (adaptpoint_amd.fused_wide.row_map; index-stage work), built here when a backward will follow and the caller has none.
        geo, dd: the neighbourhoods' occurrence statistics (index-stage work too: `point_geo`; computed here when
        the caller has none)."""
        dev = f.device
        call = _Launcher(dev)
        lib = _lib.load()
        B, C, N = f.shape
        M = new_p.shape[1]
        self.dims = (B, N, M)
        self.radius = float(radius)
        self.sync = sync_bn
        self.relu = 1 if relu else 0
        self.prec = prec = _PREC[PRECISION]
        w1 = _mat(conv1.weight, C_MID, C_IN + 3)
        w2 = _mat(conv2.weight, C_OUT, C_MID)
        ws = bs = None
        if skip_conv is not None:
            ws = _mat(skip_conv.weight, C_OUT, C_IN)
            bs = skip_conv.bias.detach() if skip_conv.bias is not None else None
        count = float(B * M * K_NS)       # this rank's positions; SyncBatchNorm all-reduces it with the sums
        if geo is None:
            geo, dd = point_geo(p, new_p, idx, radius)
        if want_backward:
            from . import fused_wide
            if tmap is None:
                tmap = fused_wide.tile_map(idx)
                rowmap = None
            if rowmap is None:
                rowmap = fused_wide.row_map(tmap, B, N, M, fidx=fidx)
        rows1 = lib.apn_sa_prep_rows(B, N)
        v, _buf = _carve(dev, [("ft", prec * B * N * C // 2), ("pack1", 4 * C_MID), ("pack2", 4 * C_OUT),
                               ("ysel", B * M * C_OUT), ("ksel", B * M * C_OUT // 4),
                               ("part1", rows1 * 64), ("acc2", _acc_floats(128))])
        out = torch.empty(B, C_OUT, M, dtype=torch.float32, device=dev)
        # what the backward accumulates into with atomics (the skip branch's gradient rows at the sampled points; the two
        # integer accumulator sets): cleared by the forward's last launch
        self.zsizes = (([("gip", B * N * C_MID)] if ws is not None else [])
                       + [("accS", _acc_floats(128)), ("accT", _acc_floats(64))])
        self.zviews = self.zbuf = None
        if want_backward:
            self.zviews, self.zbuf = _carve(dev, self.zsizes)
        zptr, zfl = _ptr(self.zbuf), (self.zbuf.numel() if self.zbuf is not None else 0)
        bn1a, bn2a = _bn_args(bn1), _bn_args(bn2)
        self.train1, self.train2 = bool(bn1a[7]), bool(bn2a[7])

        def run(phases, sums1=None, sums2=None):
            if PER_KERNEL_LAUNCH:
                return _forward_per_kernel(call, phases, prec, B, N, M, self.radius, p, new_p, f, idx, fidx, geo, dd,
                                           w1, w2, ws, bs, bn1a, bn2a, count, self.relu, v, sums1,
                                           sums2, out, rows1, tmap, zptr, zfl)
            call("apn_sa_forward_seq", phases, prec, B, N, M, self.radius, p.data_ptr(), new_p.data_ptr(),
                 f.data_ptr(), idx.data_ptr(), _ptr(tmap), _ptr(fidx), geo.data_ptr(), dd.data_ptr(),
                 w1.data_ptr(), w2.data_ptr(), _ptr(ws), _ptr(bs), *bn1a, *bn2a, count, self.relu,
                 v["ft"].data_ptr(), v["part1"].data_ptr(), _ptr(sums1), _ptr(sums2),
                 v["pack1"].data_ptr(), v["pack2"].data_ptr(), v["acc2"].data_ptr(),
                 v["ysel"].data_ptr(), v["ksel"].data_ptr(), out.data_ptr(), zptr, zfl)

        if not _phased(sync_bn):
            run(7)
        else:                                   # SyncBatchNorm: all-reduce between the phases
            run(1)
            s1 = _all_reduce_rows(call, v["part1"], rows1, 64, count, dev) if self.train1 else None
            run(2, sums1=s1)
            s2 = _all_reduce_acc(call, v["acc2"], 128, count, dev) if self.train2 else None
            run(4, sums2=s2)
        self.out = out
        self.saved = dict(p=p, f=f, new_p=new_p, idx=idx, tmap=tmap, rowmap=rowmap, fidx=fidx,
                          geo=geo, ft=v["ft"], w1=w1, w2=w2, ws=ws,
                          has_bs=bs is not None, pack1=v["pack1"], pack2=v["pack2"], ysel=v["ysel"],
                          ksel=v["ksel"], count=count)


def _backward(fw, g_out, need_p, need_newp):
    """All gradients of the fused chain from g_out (B,64,M)."""
    if fw is None:
        raise RuntimeError("fused set abstraction: backward called a second time (its saved state is "
                           "released after the first backward; retain_graph is not supported)")
    sv = fw.saved
    B, N, M = fw.dims
    dev = fw.out.device
    P, sync = sv["count"], fw.sync
    f32 = dict(dtype=torch.float32, device=dev)
    if g_out is None:                 # only the other output was used downstream
        g_out = torch.zeros_like(fw.out)
    if g_out.dtype != torch.float32:
        g_out = g_out.float()
    gs = g_out.stride()               # read with strides: a broadcast gradient is never materialised
    w1, w2, ws = sv["w1"], sv["w2"], sv["ws"]
    has_skip = ws is not None
    lib = _lib.load()
    call = _Launcher(dev)
    rows = lib.apn_sa_bwd_main_rows(B, M)
    prow = lib.apn_sa_bwd_prep_rows(B, M)
    wrows = lib.apn_sa_bwd_weight_rows(B, N)

    # scratch: the zero-filled (atomically accumulated) region first, contiguous -- unless the forward's last
    # launch already allocated and cleared it (fw.zbuf)
    prezeroed = fw.zbuf is not None
    zsizes = [] if prezeroed else fw.zsizes
    if sv["rowmap"] is None:
        raise RuntimeError("fused set abstraction: backward without a row map (the forward was run with want_backward=False)")
    pcnt_poff, rowdst = sv["rowmap"]
    # GU: one 128-byte row per tile-map row, at the row's place in the point-sorted order (capacity 32 B M rows; only the
    # map's live rows are written and read)
    sizes = zsizes + [("goa", B * M * C_OUT), ("partWs", prow * C_OUT * C_IN if has_skip else 0),
                      ("partW2", rows * C_OUT * C_MID), ("partW", wrows * 32 * 38),
                      ("HA", B * M * C_MID), ("HB", B * M * C_MID), ("GU", lib.apn_sa_rowmap_places(B, N, M) * C_MID)]
    v, buf = _carve(dev, sizes)
    zero_floats = sum((nfl + 63) // 64 * 64 for _, nfl in zsizes)
    if prezeroed:
        v.update(fw.zviews)
        fw.zbuf = fw.zviews = None                        # consumed: a second backward must not trust it
    # small gradients in one buffer (kept alive by the parameters' .grad)
    gsz = [("w2", C_OUT * C_MID), ("w1", C_MID * (C_IN + 3)), ("g1", C_MID), ("b1", C_MID),
           ("g2", C_OUT), ("b2", C_OUT), ("ws", C_OUT * C_IN if has_skip else 0),
           ("bs", C_OUT if (has_skip and sv["has_bs"]) else 0)]
    g, _gbuf = _carve(dev, gsz)           # every view is fully written
    g_f = torch.empty(B, C_IN, N, **f32)
    g_p = _ops.zeros(B, N, 3, **f32) if need_p else None
    g_newp = torch.empty(B, M, 3, **f32) if need_newp else None
    gws = g["ws"].data_ptr() if has_skip else None
    gbs = g["bs"].data_ptr() if (has_skip and sv["has_bs"]) else None

    def run(phases, sumsS=None, sumsT=None):
        if PER_KERNEL_LAUNCH:
            return _backward_per_kernel(call, phases, fw, sv, g_out, buf, zero_floats, v, g, sumsS,
                                        sumsT, g_f, g_p, g_newp, rows, prow, wrows, has_skip, pcnt_poff, rowdst)
        call("apn_sa_backward_seq", phases, fw.prec, B, N, M, fw.radius, sv["p"].data_ptr(),
             sv["new_p"].data_ptr(), sv["idx"].data_ptr(), _ptr(sv["tmap"]), _ptr(sv["fidx"]), sv["geo"].data_ptr(),
             w1.data_ptr(), w2.data_ptr(), _ptr(ws), sv["ft"].data_ptr(), sv["pack1"].data_ptr(),
             sv["pack2"].data_ptr(), sv["ysel"].data_ptr(), sv["ksel"].data_ptr(),
             fw.out.data_ptr(), fw.relu, 1 if fw.train1 else 0, 1 if fw.train2 else 0, float(P),
             g_out.data_ptr(), gs[0], gs[1], gs[2], buf.data_ptr(), zero_floats * 4,
             v["gip"].data_ptr() if has_skip else None, v["accS"].data_ptr(),
             v["accT"].data_ptr(), pcnt_poff.data_ptr(), rowdst.data_ptr(), v["GU"].data_ptr(), v["goa"].data_ptr(),
             v["partWs"].data_ptr() if has_skip else None, v["partW2"].data_ptr(), v["partW"].data_ptr(),
             _ptr(sumsS), _ptr(sumsT), v["HA"].data_ptr(), v["HB"].data_ptr(),
             g_f.data_ptr(), _ptr(g_p), _ptr(g_newp), g["w1"].data_ptr(), g["w2"].data_ptr(),
             g["g1"].data_ptr(), g["b1"].data_ptr(), g["g2"].data_ptr(), g["b2"].data_ptr(), gws, gbs)

    if not _phased(sync):
        run(7)
    else:
        run(1)
        sS = _all_reduce_acc(call, v["accS"], 128, P, dev)
        run(2, sumsS=sS)
        sT = _all_reduce_acc(call, v["accT"], 64, P, dev)
        run(4, sumsS=sS, sumsT=sT)
    # conv-weight gradients are per-rank sums here and DistributedDataParallel (or
    # dp.allreduce_mean_) averages them; under SyncBatchNorm dL/dgamma, dL/dbeta already are
    # global / world on every rank (sa_glue.hip: bwd_finalize), which that averaging leaves as is
    return dict(f=g_f, p=g_p, new_p=g_newp, w1=g["w1"].view(C_MID, C_IN + 3, 1, 1),
                w2=g["w2"].view(C_OUT, C_MID, 1, 1), g1=g["g1"], b1=g["b1"], g2=g["g2"], b2=g["b2"],
                ws=g["ws"].view(C_OUT, C_IN, 1) if has_skip else None,
                bs=g["bs"] if (has_skip and sv["has_bs"]) else None)


def _forward_per_kernel(call, phases, prec, B, N, M, radius, p, new_p, f, idx, fidx, geo, dd, w1, w2, ws, bs,
                        bn1a, bn2a, count, relu, v, sums1, sums2, out, rows1, tmap=None, zptr=None, zfl=0):
    """Python mirror of apn_sa_forward_seq (csrc/sa_seq.hip), one foreign call per kernel."""
    lib = _lib.load()
    if phases & 1:
        call("apn_sa_prep_stats", B, N, f.data_ptr(), geo.data_ptr(), dd.data_ptr(), w1.data_ptr(), prec,
             bn1a[7], v["ft"].data_ptr(), v["part1"].data_ptr(), v["acc2"].data_ptr(), lib.apn_sa_acc_words(128))
    if phases & 2:
        call("apn_sa_fwd_main", B, N, M, prec, radius, p.data_ptr(), new_p.data_ptr(), v["ft"].data_ptr(),
             idx.data_ptr(), _ptr(tmap), w1.data_ptr(), w2.data_ptr(), *bn1a, count, v["part1"].data_ptr(), rows1,
             _ptr(sums1), v["pack1"].data_ptr(), bn2a[0], v["ysel"].data_ptr(), v["ksel"].data_ptr(),
             v["acc2"].data_ptr())
    if phases & 4:
        call("apn_sa_fwd_out", B, N, M, v["ysel"].data_ptr(), v["acc2"].data_ptr(), _ptr(sums2), *bn2a, count,
             v["pack2"].data_ptr(), v["ft"].data_ptr() if ws is not None else None, prec,
             _ptr(fidx) if ws is not None else None, _ptr(ws), _ptr(bs), relu, out.data_ptr(), zptr, zfl)


def _backward_per_kernel(call, phases, fw, sv, g_out, buf, zero_floats, v, g, sumsS, sumsT, g_f, g_p,
                         g_newp, rows, prow, wrows, has_skip, pcnt_poff, rowdst):
    """Python mirror of apn_sa_backward_seq (csrc/sa_seq.hip), one foreign call per kernel."""
    B, N, M = fw.dims
    w1, w2, ws, P = sv["w1"], sv["w2"], sv["ws"], float(sv["count"])
    gip = v["gip"].data_ptr() if has_skip else None
    if phases & 1:
        if zero_floats:
            call("apn_zero_fill", buf.data_ptr(), zero_floats * 4)
        call("apn_sa_bwd_prep", B, N, M, g_out.data_ptr(), *g_out.stride(), fw.out.data_ptr(), fw.relu,
             sv["ysel"].data_ptr(), sv["pack2"].data_ptr(), sv["ft"].data_ptr() if has_skip else None,
             fw.prec, _ptr(sv["fidx"]) if has_skip else None, _ptr(ws), v["goa"].data_ptr(),
             v["accS"].data_ptr(), v["partWs"].data_ptr() if has_skip else None, gip,
             pcnt_poff.data_ptr() + 8 * B * N)            # (the row map's per-cloud verdict on the picks, behind pcnt | poff)
    if phases & 2:
        call("apn_sa_bwd_main", B, N, M, fw.prec, fw.radius, sv["p"].data_ptr(), sv["new_p"].data_ptr(),
             sv["ft"].data_ptr(), sv["idx"].data_ptr(), _ptr(sv["tmap"]), w1.data_ptr(), w2.data_ptr(),
             sv["pack1"].data_ptr(), sv["pack2"].data_ptr(), v["accS"].data_ptr(), _ptr(sumsS), P,
             1 if fw.train2 else 0, v["goa"].data_ptr(), sv["ksel"].data_ptr(), v["accT"].data_ptr(),
             v["partW2"].data_ptr(), rowdst.data_ptr(), v["GU"].data_ptr(), v["HA"].data_ptr(), v["HB"].data_ptr())
    if phases & 4:
        call("apn_sa_bwd_point_grads", B, N, M, v["GU"].data_ptr(), pcnt_poff.data_ptr(), sv["geo"].data_ptr(),
             v["HA"].data_ptr(), v["HB"].data_ptr(), v["accT"].data_ptr(), _ptr(sumsT), P, 1 if fw.train1 else 0,
             sv["pack1"].data_ptr(), sv["ft"].data_ptr(), fw.prec, sv["p"].data_ptr(), sv["new_p"].data_ptr(),
             w1.data_ptr(), gip, fw.radius, v["partW"].data_ptr(), g_f.data_ptr(), _ptr(g_p), _ptr(g_newp))
        call("apn_sa_bwd_finalize", v["partW"].data_ptr(), wrows, fw.radius, g["w1"].data_ptr(),
             v["partWs"].data_ptr() if has_skip else None, prow, g["ws"].data_ptr() if has_skip else None,
             v["partW2"].data_ptr(), rows, g["w2"].data_ptr(), v["accS"].data_ptr(), _ptr(sumsS),
             v["accT"].data_ptr(), _ptr(sumsT), g["bs"].data_ptr() if (has_skip and sv["has_bs"]) else None,
             g["g2"].data_ptr(), g["b2"].data_ptr(), g["g1"].data_ptr(), g["b1"].data_ptr())


class _GroupedMlpMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, new_p, f, idx, w1, g1, b1, w2, g2, b2, mods):
        radius, conv1, bn1, conv2, bn2, sync_bn = mods
        fw = _Forward(p.contiguous(), f.contiguous(), new_p.contiguous(), idx.contiguous(), None,
                      radius, conv1, bn1, conv2, bn2, None, False, sync_bn, want_backward=any(ctx.needs_input_grad))
        ctx.fw = fw
        ctx.save_for_backward(p, new_p, f, idx)     # autograd's in-place version checks cover the inputs
        ctx.set_materialize_grads(False)
        ctx.flags = (p.requires_grad, new_p.requires_grad,
                     g1 is not None, b1 is not None, g2 is not None, b2 is not None)
        return fw.out

    @staticmethod
    def backward(ctx, g_out):
        need_p, need_q, a1, a2, a3, a4 = ctx.flags
        ctx.saved_tensors                           # raises if an input was modified in place since
        g = _backward(ctx.fw, g_out, need_p, need_q)
        ctx.fw = None
        return (g["p"], g["new_p"], g["f"], None, g["w1"], g["g1"] if a1 else None,
                g["b1"] if a2 else None, g["w2"], g["g2"] if a3 else None,
                g["b2"] if a4 else None, None)


def grouped_mlp_max(p, new_p, f, idx, radius, conv1, bn1, conv2, bn2, sync_bn=False):
    """out (B,64,M) = max_K bn2(conv2(relu(bn1(conv1(cat[(p[idx]-new_p)/r, f[idx]])))))."""
    return _GroupedMlpMax.apply(p, new_p, f, idx, conv1.weight, bn1.weight, bn1.bias,
                                conv2.weight, bn2.weight, bn2.bias,
                                (radius, conv1, bn1, conv2, bn2, sync_bn))


class Sampling:
    """The index stage of a block: fidx (B,M) i32 = FPS picks, new_p (B,M,3) = p[fidx],
    idx (B,M,K) i32 = ball-query neighbours.  It depends on the coordinates only -- never on
    features or weights -- so it can be computed ahead of the feature path (another stream,
    the next batch) and handed to `fused_set_abstraction(..., sampling=...)`.
    The three tensors are views of ONE buffer, so a double-buffered pipeline rotates them
    with a single copy.  A buffer may stack several batches (their index stages then run as single
    launches); `clouds(lo, hi)` hands out one batch."""

    def __init__(self, B, M, K, device):
        self.shape = (B, M, K)
        self.buf = torch.empty(B * M * (1 + 3 + K), dtype=torch.int32, device=device)
        o = 0
        self.fidx = self.buf[o:o + B * M].view(B, M)
        o += B * M
        self.new_p = self.buf[o:o + B * M * 3].view(torch.float32).view(B, M, 3)
        o += B * M * 3
        self.idx = self.buf[o:o + B * M * K].view(B, M, K)
        self.index = None        # adaptpoint_amd.fused_wide.NeighbourIndex of idx, when the width-generic kernels run
        self.tmap = None         # adaptpoint_amd.fused_wide.tile_map of idx, for the register-resident kernels
        self.rowmap = None       # (pcnt_poff, rowdst) of tmap (adaptpoint_amd.fused_wide.row_map), for their backward pass
        self.geo = None          # (B,N,4) int64 occurrence statistics of the neighbourhoods (csrc/sa_geo.hip) and
        self.dd = None           # (B, 6 * slabs) float64 their second moments, for the register-resident kernels
        self.ready = None        # event recorded behind the index stage when it ran on another stream (graphs.wait_ready)
        self.ties = None         # (B,) int32: the nested sampler's record (first step whose arg-max was not unique)

    def alloc_geo(self, n_points):
        if self.geo is None or self.geo.shape[1] != n_points:
            B = self.shape[0]
            dev = self.buf.device
            self.geo = torch.empty(B, n_points, 4, dtype=torch.int64, device=dev)
            self.dd = torch.empty(B, _lib.load().apn_sa_geo_dd_doubles(n_points), dtype=torch.float64, device=dev)
        return self.geo, self.dd

    def clouds(self, lo, hi):
        """The index stage of clouds lo..hi-1 as a `Sampling`-like view (no copy): index stages of
        several batches computed in ONE launch over the stacked clouds are handed out per batch."""
        v = object.__new__(Sampling)
        v.shape = (hi - lo,) + self.shape[1:]
        v.buf = None
        v.fidx, v.new_p, v.idx = self.fidx[lo:hi], self.new_p[lo:hi], self.idx[lo:hi]
        v.index = None
        v.tmap = None
        v.rowmap = None
        v.geo = v.dd = None
        v.ready = None
        v.ties = None if self.ties is None else self.ties[lo:hi]
        if self.geo is not None:
            v.geo, v.dd = self.geo[lo:hi], self.dd[lo:hi]
        return v


@torch.no_grad()
def point_geo(p, new_p, idx, radius, out=None):
    """The neighbourhoods' occurrence statistics (csrc/sa_geo.hip; index-stage work: coordinates and indices
    only): geo (B,N,4) int64 = per support point {occurrences, sum of (p - query)/radius in units of 2^-36},
    dd (B, 6 * slabs) float64 = each cloud's shares of sum d d^T over its positions."""
    B, N, _ = p.shape
    M, K = idx.shape[1], idx.shape[2]
    dev = p.device
    if out is None:
        geo = torch.empty(B, N, 4, dtype=torch.int64, device=dev)
        dd = torch.empty(B, _lib.load().apn_sa_geo_dd_doubles(N), dtype=torch.float64, device=dev)
    else:
        geo, dd = out
    _Launcher(dev)("apn_sa_point_geo", B, N, M, K, float(radius), p.contiguous().data_ptr(),
                   new_p.contiguous().data_ptr(), idx.contiguous().data_ptr(), geo.data_ptr(), dd.data_ptr())
    return geo, dd


@torch.no_grad()
def sample_and_query(p, npoint, radius, nsample=K_NS, out=None, geo=False, nested=False, ties=None):
    """FPS (+ gather of the sampled coordinates, one launch; pointnext.py:146-147) and ball
    query (group.py:245) on the current stream; geo=True: also the occurrence statistics the
    register-resident fused block takes (`point_geo`).  nested=True (the levels of an index pyramid): the sampler
    records its first non-unique arg-max in `Sampling.ties`, and with `ties` = the PREVIOUS level's record (p = that
    level's new_p) the level is a copy of the first npoint picks wherever the record allows (csrc/fps.hip, NEST)."""
    p = p.contiguous()
    dev = p.device
    B, N, _ = p.shape
    smp = out if out is not None else Sampling(B, npoint, nsample, dev)
    assert smp.shape == (B, npoint, nsample)
    gptr = dptr = None
    if geo and nsample == K_NS and smp.buf is not None:
        g_, d_ = smp.alloc_geo(N)
        gptr, dptr = g_.data_ptr(), d_.data_ptr()
    elif geo and nsample == K_NS and smp.geo is not None:
        gptr, dptr = smp.geo.data_ptr(), smp.dd.data_ptr()
    call = _Launcher(dev)
    if nested and N <= 4096:
        if smp.ties is None:
            smp.ties = torch.empty(B, dtype=torch.int32, device=dev)
        if PER_KERNEL_LAUNCH:
            call("apn_furthest_point_sampling_nested", B, N, npoint, p.data_ptr(), _ptr(ties), smp.fidx.data_ptr(),
                 smp.new_p.data_ptr(), smp.ties.data_ptr())
            call("apn_ball_query_zero", B, N, npoint, float(radius), nsample, smp.new_p.data_ptr(), p.data_ptr(),
                 smp.idx.data_ptr())
            if gptr is not None:
                call("apn_sa_point_geo", B, N, npoint, nsample, float(radius), p.data_ptr(), smp.new_p.data_ptr(),
                     smp.idx.data_ptr(), gptr, dptr)
            return smp
        call("apn_sa_sample_seq_nested", B, N, npoint, float(radius), nsample, p.data_ptr(), _ptr(ties), smp.ties.data_ptr(),
             smp.fidx.data_ptr(), smp.new_p.data_ptr(), smp.idx.data_ptr(), gptr, dptr)
        return smp
    if PER_KERNEL_LAUNCH:
        call("apn_furthest_point_sampling_xyz", B, N, npoint, p.data_ptr(), None,
             smp.fidx.data_ptr(), smp.new_p.data_ptr())
        call("apn_ball_query_zero", B, N, npoint, float(radius), nsample, smp.new_p.data_ptr(),
             p.data_ptr(), smp.idx.data_ptr())
        if gptr is not None:
            call("apn_sa_point_geo", B, N, npoint, nsample, float(radius), p.data_ptr(),
                 smp.new_p.data_ptr(), smp.idx.data_ptr(), gptr, dptr)
        return smp
    # temp = None: the sampler starts from 1e10 in registers (no fill launch, no min-distances kept)
    call("apn_sa_sample_seq", B, N, npoint, float(radius), nsample, p.data_ptr(), None,
         smp.fidx.data_ptr(), smp.new_p.data_ptr(), smp.idx.data_ptr(), gptr, dptr)
    return smp


@torch.no_grad()
def sample_and_query_many(ps, npoint, radius, nsample=K_NS, outs=None):
    """Index stages of consecutive batches `ps` (each (B,N,3)) on the current stream, overlapped:
    the FPS of batch i shares its launch with the ball query of batch i-1 (which needs ~1/8 of
    the sampler's time on the CUs the sampler leaves idle), so a batch costs one FPS, not
    FPS + ball query.  Returns the list of `Sampling`."""
    ps = [p.contiguous() for p in ps]
    B, N, _ = ps[0].shape
    dev = ps[0].device
    if outs is None:
        outs = [Sampling(B, npoint, nsample, dev) for _ in ps]
    assert len(outs) == len(ps) and all(o.shape == (B, npoint, nsample) for o in outs)
    call = _Launcher(dev)
    for i in range(len(ps) + 1):
        a = i if i < len(ps) else None               # sampler role
        b = i - 1 if i >= 1 else None                # search role
        call("apn_sa_sample_overlap", B, N, npoint, float(radius), nsample,
             ps[a].data_ptr() if a is not None else None,
             outs[a].fidx.data_ptr() if a is not None else None,
             outs[a].new_p.data_ptr() if a is not None else None,
             ps[b].data_ptr() if b is not None else None,
             outs[b].new_p.data_ptr() if b is not None else None,
             outs[b].idx.data_ptr() if b is not None else None)
        if b is not None and outs[b].geo is not None:           # buffers that carry occurrence statistics get them
            point_geo(ps[b], outs[b].new_p, outs[b].idx, radius, out=(outs[b].geo, outs[b].dd))
    return outs


class _SetAbstraction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, f, w1, g1, b1, w2, g2, b2, ws, bs, mods):
        npoint, radius, conv1, bn1, conv2, bn2, skip_conv, relu, sync_bn, sampling = mods
        p = p.contiguous()
        f = f.contiguous()
        smp = sampling if sampling is not None else sample_and_query(p, npoint, radius, geo=True)
        if sampling is None and TILE_MAP_INLINE and any(ctx.needs_input_grad):
            # the distinct-hit tile map, built in line (two launches): both passes then walk ~1/4 of the tiles -- forward
            # 29 -> 20 us, backward 66 -> 34 us at B = 32, N = 1024
            from . import fused_wide
            smp.tmap = fused_wide.tile_map(smp.idx)
            smp.rowmap = fused_wide.row_map(smp.tmap, p.shape[0], p.shape[1], npoint, fidx=smp.fidx)
        fidx, new_p, idx = smp.fidx, smp.new_p, smp.idx
        if p.requires_grad:
            new_p = new_p.clone()          # returned as a differentiable output
        fw = _Forward(p, f, new_p, idx, fidx, radius, conv1, bn1, conv2, bn2, skip_conv, relu,
                      sync_bn, tmap=getattr(smp, "tmap", None), geo=getattr(smp, "geo", None),
                      dd=getattr(smp, "dd", None), want_backward=any(ctx.needs_input_grad),
                      rowmap=getattr(smp, "rowmap", None))
        ctx.fw = fw
        ctx.save_for_backward(p, f)        # autograd's in-place version checks cover the inputs
        ctx.set_materialize_grads(False)   # an unused output's gradient arrives as None, not as zeros
        ctx.flags = (p.requires_grad, g1 is not None, b1 is not None, g2 is not None,
                     b2 is not None, ws is not None, bs is not None)
        ctx.mark_non_differentiable(new_p) if not p.requires_grad else None
        return new_p, fw.out

    @staticmethod
    def backward(ctx, g_newp_in, g_out):
        need_p, a1, a2, a3, a4, a5, a6 = ctx.flags
        fw = ctx.fw
        ctx.saved_tensors                  # raises if an input was modified in place since
        g = _backward(fw, g_out, need_p, need_p)
        g_p = g["p"]
        if need_p:
            # new_p = p[fidx]: its gradient (from the chain and from downstream users) returns to p
            gq = g["new_p"] if g_newp_in is None else g["new_p"] + g_newp_in
            g_p = g_p.scatter_add(1, fw.saved["fidx"].long().unsqueeze(-1).expand(-1, -1, 3), gq)
        ctx.fw = None
        return (g_p, g["f"], g["w1"], g["g1"] if a1 else None, g["b1"] if a2 else None, g["w2"],
                g["g2"] if a3 else None, g["b2"] if a4 else None, g["ws"] if a5 else None,
                g["bs"] if a6 else None, None)


def fused_set_abstraction(p, f, npoint, radius, conv1, bn1, conv2, bn2, skip_conv=None, relu=True,
                          sync_bn=False, sampling=None):
    """(new_p (B,npoint,3), out (B,64,npoint)) of a PointNeXt set-abstraction block.
    `sampling`: a precomputed `Sampling` of p (same npoint / radius), else computed here."""
    assert skip_conv is None or isinstance(skip_conv, nn.Conv1d)
    ws = skip_conv.weight if skip_conv is not None else None
    bs = skip_conv.bias if skip_conv is not None else None
    return _SetAbstraction.apply(p, f, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight,
                                 bn2.bias, ws, bs,
                                 (npoint, radius, conv1, bn1, conv2, bn2, skip_conv, relu, sync_bn,
                                  sampling))
