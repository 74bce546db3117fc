"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy front-end of `libpointnet2_oracle.so` (built from pointnet2_oracle.c by
oracle/Makefile).  Signatures mirror the Python-visible wrappers of the
reference extension (openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:11-23)
but allocate and return numpy arrays.  Parity pinning: see the header of
pointnet2_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpointnet2_oracle.so")

DIST_PLAIN, DIST_FMA_YX, DIST_FMA_XY, DIST_HIPCC = 0, 1, 2, 3
DIST_PINNED = DIST_FMA_XY
ALL_DIST_VARIANTS = (DIST_PLAIN, DIST_FMA_YX, DIST_FMA_XY, DIST_HIPCC)

_lib = None


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "libpointnet2_oracle.so"], check=True,
                   stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.apo_opt_n_threads.restype = ctypes.c_int
        _lib.apo_max_threads.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def set_threads(t):
    lib().apo_set_threads(ctypes.c_int(int(t)))


def max_threads():
    return int(lib().apo_max_threads())


def opt_n_threads(n):
    return int(lib().apo_opt_n_threads(ctypes.c_int(int(n))))


def furthest_point_sampling(xyz, m, variant=DIST_PINNED, return_temp=False):
    """xyz (B,N,3) f32 -> idx (B,m) i32 [, temp (B,N) f32]  (sampling_gpu.cu:101-215)."""
    xyz, pxyz = _f(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, dtype=np.float32)      # subsample.py:94
    idx = np.zeros((b, max(m, 0)), dtype=np.int32)
    lib().apo_furthest_point_sampling(b, n, int(m), pxyz, temp.ctypes.data_as(ctypes.c_void_p),
                                      idx.ctypes.data_as(ctypes.c_void_p), int(variant))
    return (idx, temp) if return_temp else idx


def ball_query(radius, nsample, xyz, new_xyz, variant=DIST_PINNED):
    """xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) i32  (ball_query_gpu.cu:15-51).
    `radius` is narrowed to float32 as at the pybind boundary (ball_query.cpp:29)."""
    xyz, pxyz = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)      # group.py:194
    lib().apo_ball_query(b, n, m, ctypes.c_float(float(np.float32(radius))), int(nsample),
                         pq, pxyz, idx.ctypes.data_as(ctypes.c_void_p), int(variant))
    return idx


def group_points(points, idx):
    """points (B,C,N), idx (B,M,K) -> (B,C,M,K)  (group_points_gpu.cu:53-72)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    _, m, k = idx.shape
    out = np.empty((b, c, m, k), dtype=np.float32)
    lib().apo_group_points(b, c, n, m, k, pp, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def group_points_grad(grad_out, idx, n):
    """grad_out (B,C,M,K), idx (B,M,K) -> grad_points (B,C,n)  (group_points_gpu.cu:14-31)."""
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m, k = grad_out.shape
    gp = np.zeros((b, c, n), dtype=np.float32)           # group.py:111
    lib().apo_group_points_grad(b, c, int(n), m, k, pg, pi, gp.ctypes.data_as(ctypes.c_void_p))
    return gp


def gather_points(points, idx):
    """points (B,C,N), idx (B,M) -> (B,C,M)  (sampling_gpu.cu:15-31)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), dtype=np.float32)
    lib().apo_gather_points(b, c, n, m, pp, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def gather_points_grad(grad_out, idx, n):
    """grad_out (B,C,M), idx (B,M) -> grad_points (B,C,n)  (sampling_gpu.cu:53-70)."""
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m = grad_out.shape
    gp = np.zeros((b, c, n), dtype=np.float32)           # subsample.py:136
    lib().apo_gather_points_grad(b, c, int(n), m, pg, pi, gp.ctypes.data_as(ctypes.c_void_p))
    return gp


def three_nn(unknown, known, variant=DIST_PINNED):
    """unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32, idx (B,n,3) i32
    (interpolate_gpu.cu:16-59; the Python caller takes sqrt, upsampling.py:33)."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.empty((b, n, 3), dtype=np.float32)
    idx = np.empty((b, n, 3), dtype=np.int32)
    lib().apo_three_nn(b, n, m, pu, pk, dist2.ctypes.data_as(ctypes.c_void_p),
                       idx.ctypes.data_as(ctypes.c_void_p), int(variant))
    return dist2, idx


def three_interpolate(points, idx, weight, fused=True):
    """points (B,C,M), idx/weight (B,n,3) -> (B,C,n)  (interpolate_gpu.cu:84-104)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), dtype=np.float32)
    lib().apo_three_interpolate(b, c, m, n, pp, pi, pw, out.ctypes.data_as(ctypes.c_void_p),
                                int(bool(fused)))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """grad_out (B,C,n), idx/weight (B,n,3) -> grad_points (B,C,m)  (interpolate_gpu.cu:127-149)."""
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, n = grad_out.shape
    gp = np.zeros((b, c, int(m)), dtype=np.float32)      # upsampling.py:82
    lib().apo_three_interpolate_grad(b, c, n, int(m), pg, pi, pw,
                                     gp.ctypes.data_as(ctypes.c_void_p))
    return gp


def resample_points(points, fidx, choice, cx):
    """examples/classification/train_autoaug.py:493-501: keep the FPS picks `choice` selects
    (one draw for the whole batch), gather the rows, split into pos (B,S,3) and x (B,cx,S)."""
    points = np.asarray(points, dtype=np.float32)
    sel = np.asarray(fidx)[:, np.asarray(choice)].astype(np.int64)              # :495-496
    rows = np.take_along_axis(points, sel[..., None].repeat(points.shape[-1], -1), 1)   # :497-498
    return (np.ascontiguousarray(rows[:, :, :3]),                               # :500
            np.ascontiguousarray(rows[:, :, :cx].transpose(0, 2, 1)))           # :501


# --- PointsetGrouper's grouping stage (numpy, float32) ---------------

def pointset_group_max(points, idx, fidx, alpha, beta):
    """points (B,N,C), idx (B,M,K), fidx (B,M), alpha/beta [C] -> (out (B,C,M), ksel (B,M,C) u8).
    generator_component4_15.py:413 (index_points), :422-427 (anchor normalisation + affine, three
    separately rounded float32 operations), :429 (max over K; first maximal position)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    idx = np.asarray(idx).astype(np.int64)
    fidx = np.asarray(fidx).astype(np.int64)
    al = np.asarray(alpha, dtype=np.float32).reshape(1, 1, 1, -1)
    be = np.asarray(beta, dtype=np.float32).reshape(1, 1, 1, -1)
    bi = np.arange(points.shape[0])[:, None, None]
    grouped = points[bi, idx, :]                                         # (B,M,K,C)
    mean = points[np.arange(points.shape[0])[:, None], fidx, :][:, :, None, :]
    grouped = (al * (grouped - mean)).astype(np.float32) + be
    ksel = grouped.argmax(axis=2).astype(np.uint8)                       # first occurrence
    return np.ascontiguousarray(grouped.max(axis=2).transpose(0, 2, 1)), ksel


def pointset_group_max_grad(points, idx, fidx, alpha, ksel, grad_out):
    """-> (g_points (B,N,C), g_alpha [C], g_beta [C]) in float64 accumulation."""
    points = np.asarray(points, dtype=np.float64)
    B, N, C = points.shape
    idx = np.asarray(idx).astype(np.int64)
    fidx = np.asarray(fidx).astype(np.int64)
    al = np.asarray(alpha, dtype=np.float64).reshape(-1)
    g = np.asarray(grad_out, dtype=np.float64).transpose(0, 2, 1)        # (B,M,C)
    sel = np.take_along_axis(idx, np.asarray(ksel).astype(np.int64), axis=2)   # (B,M,C) source point
    gp = np.zeros((B, N, C))
    bi = np.arange(B)[:, None, None]
    ci = np.arange(C)[None, None, :]
    np.add.at(gp, (bi, sel, ci), g * al)
    np.add.at(gp, (bi, fidx[:, :, None], ci), -g * al)
    x_sel = points[bi, sel, ci]
    anchor = points[np.arange(B)[:, None], fidx, :]
    return gp, (g * (x_sel - anchor)).sum((0, 1)), g.sum((0, 1))


# --- Anchor_selfattention's core (numpy, float64) ---------------------

def attention(q, k, v, heads):
    """q, k, v (B,M,heads*d) -> softmax(q k^T / sqrt(d)) v, (B,M,heads*d), in float64.
    generator_component4_15.py:460-474."""
    q, k, v = (np.asarray(t, dtype=np.float64) for t in (q, k, v))
    B, M, C = q.shape
    d = C // heads
    qh, kh, vh = (t.reshape(B, M, heads, d).transpose(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(0, 1, 3, 2) / np.sqrt(d)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p @ vh).transpose(0, 2, 1, 3).reshape(B, M, C)


def attention_grad(q, k, v, heads, grad_out):
    """-> (dq, dk, dv), float64."""
    q, k, v, g = (np.asarray(t, dtype=np.float64) for t in (q, k, v, grad_out))
    B, M, C = q.shape
    d = C // heads
    qh, kh, vh, gh = (t.reshape(B, M, heads, d).transpose(0, 2, 1, 3) for t in (q, k, v, g))
    s = qh @ kh.transpose(0, 1, 3, 2) / np.sqrt(d)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    dv = p.transpose(0, 1, 3, 2) @ gh
    dp = gh @ vh.transpose(0, 1, 3, 2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / np.sqrt(d)
    dq = ds @ kh
    dk = ds.transpose(0, 1, 3, 2) @ qh
    back = lambda t: t.transpose(0, 2, 1, 3).reshape(B, M, C)
    return back(dq), back(dk), back(dv)


# ------------------------------------------------------------------------------------------
# Index-stage structures of the fused passes (include/adaptpoint_amd.h: apn_sa_wide_tilemap,
# apn_sa_wide_csr).  They have no counterpart in the reference: these are plain-Python
# statements of their DEFINITION (written independently of the kernels: a serial next-fit
# loop, a dictionary of lists), against which the GPU builders are compared bit for bit.
# ------------------------------------------------------------------------------------------
def tile_map(idx, fold=True):
    """idx (B,M,32) int32 -> dict(nt, tq0 (nt,), rowinfo (nt,32) uint32, rownn (nt,32) int32).
    A query's rows: with the ball-query structure (slots 1..c-1 differ from slot 0, slots c..31 repeat it)
    its c leading slots, slot 0 with multiplicity 33 - c; otherwise all 32 slots with multiplicity 1.
    Whole queries are packed in order into 32-row tiles (next fit), a cloud's tiles first to last;
    rowinfo = qlocal | slot << 8 | mult << 16 (| queries of the tile << 24 in row 0); padding rows:
    rowinfo 0xff, rownn = slot 0 of the tile's LAST query."""
    idx = np.asarray(idx)
    B, M, K = idx.shape
    assert K == 32
    tq0, info, nn = [], [], []
    for b in range(B):
        rows_i, rows_n, first_q, nq = [], [], None, 0

        def close(last_q):
            pad = 32 - len(rows_i)
            ri = rows_i + [0xff] * pad
            ri[0] |= nq << 24
            info.append(ri)
            nn.append(rows_n + [int(idx[b, last_q, 0])] * pad)
            tq0.append(b * M + first_q)
        for q in range(M):
            row = idx[b, q]
            c = 32
            if fold:
                differ = row != row[0]
                c0 = int(differ.sum()) + 1
                if differ[1:c0].all() and not differ[c0:].any():
                    c = c0
            if first_q is not None and len(rows_i) + c > 32:
                close(q - 1)
                rows_i, rows_n, first_q, nq = [], [], None, 0
            if first_q is None:
                first_q = q
            ql = q - first_q
            for s in range(c):
                mult = (33 - c) if s == 0 else 1
                rows_i.append(ql | (s << 8) | (mult << 16))
                rows_n.append(int(row[s]))
            nq += 1
        close(M - 1)
    return dict(nt=len(tq0), tq0=np.array(tq0, np.int32), rowinfo=np.array(info, np.uint32),
                rownn=np.array(nn, np.int32))


def inverse_map(tm, new_xyz, n_points, m):
    """Tile map `tm` (of `tile_map`), new_xyz (B,M,3) -> dict(pcnt (B*N,), lists: per point the ascending row ids
    tile * 32 + row of the rows that gather it, occ (B*N,) = sum of their multiplicities, sp (B*N,3) float64 =
    sum of multiplicity * the gathering query's coordinates)."""
    new_xyz = np.asarray(new_xyz, np.float64).reshape(-1, 3)
    lists, occ, sp = {}, {}, {}
    for t in range(tm["nt"]):
        for r in range(32):
            info = int(tm["rowinfo"][t, r])
            mult = (info >> 16) & 0xff
            if mult == 0:
                continue
            q = int(tm["tq0"][t]) + (info & 0xff)
            gn = (q // m) * n_points + int(tm["rownn"][t, r])
            lists.setdefault(gn, []).append(t * 32 + r)
            occ[gn] = occ.get(gn, 0) + mult
            sp[gn] = sp.get(gn, 0.0) + mult * new_xyz[q]
    return dict(lists=lists, occ=occ, sp=sp)


def point_geo(xyz, new_xyz, idx, radius):
    """Statement of the index stage's occurrence statistics (adaptpoint_amd/csrc/sa_geo.hip): for every support
    point n, over the positions (query q, slot k) with idx[q][k] == n,
        geo[b][n] = {count, sum of rint(d * 2^36)} with d = (xyz[n] - new_xyz[q]) / radius in float32
    (the relative position of openpoints/models/layers/group.py:250-253) as four int64, and the second moments
    dd[6] = sum over all positions of the batch of {dx dx, dx dy, dx dz, dy dy, dy dz, dz dz} in float64.
    Integer sums: exact, independent of the order the positions are visited in."""
    xyz = np.asarray(xyz, np.float32)
    new_xyz = np.asarray(new_xyz, np.float32)
    idx = np.asarray(idx, np.int64)
    B, N, _ = xyz.shape
    r = np.float32(radius)
    geo = np.zeros((B, N, 4), np.int64)
    dd = np.zeros(6, np.float64)
    for b in range(B):
        pts = xyz[b][idx[b]]                                  # (M, K, 3)
        d = ((pts - new_xyz[b][:, None, :]) / r).astype(np.float32)
        fixed = np.rint(d.astype(np.float64) * 68719476736.0).astype(np.int64)
        flat = idx[b].reshape(-1)
        np.add.at(geo[b, :, 0], flat, 1)
        for j in range(3):
            np.add.at(geo[b, :, 1 + j], flat, fixed[..., j].reshape(-1))
        d64 = d.astype(np.float64).reshape(-1, 3)
        dd += np.array([(d64[:, 0] * d64[:, 0]).sum(), (d64[:, 0] * d64[:, 1]).sum(), (d64[:, 0] * d64[:, 2]).sum(),
                        (d64[:, 1] * d64[:, 1]).sum(), (d64[:, 1] * d64[:, 2]).sum(), (d64[:, 2] * d64[:, 2]).sum()])
    return geo, dd
