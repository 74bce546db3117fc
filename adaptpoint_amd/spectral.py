"""Spectral normalisation on the gfx950 kernels of csrc/spectral.hip.

`spectral_norm(module)` registers the same parametrisation as `torch.nn.utils.parametrizations.spectral_norm`
(what every layer of the reference's `PointDiscriminator1` carries, point_discriminator.py:17-73, 149-191) -- same
buffers `_u` / `_v`, same state_dict keys, same construction (15 start-up power iterations) -- whose forward runs as
three launches and whose backward as two where PyTorch composes ~13 and ~7 small ones; the joint GAN step evaluates
21 of these forwards and 14 backwards.  Anything the kernels do not serve (CPU tensors, other dtypes, more than one
power iteration, a permuted `dim`) takes PyTorch's own path of the parent class.
"""
import ctypes

import torch
from torch.nn.utils import parametrize
from torch.nn.utils.parametrizations import _SpectralNorm

from . import _lib
from .fused import _call


class _Normalise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, owner):
        dev = weight.device
        w = weight.detach().contiguous()
        rows, cols = w.shape[0], w[0].numel()
        wn = torch.empty_like(w)
        buf = torch.empty(2 * (rows + cols) + 1, device=dev)            # scratch | u_used | v_used | sigma
        scratch, used_u, used_v = buf[:rows + cols], buf[rows + cols:2 * rows + cols], buf[2 * rows + cols:-1]
        sigma = buf[-1:]
        _call("apn_spectral_norm", dev, rows, cols, w.data_ptr(), int(owner.training), float(owner.eps),
              owner._u.data_ptr(), owner._v.data_ptr(), scratch.data_ptr(), used_u.data_ptr(), used_v.data_ptr(),
              sigma.data_ptr(), wn.data_ptr())
        ctx.save_for_backward(wn, buf)
        ctx.shape = (rows, cols)
        return wn

    @staticmethod
    def backward(ctx, g):
        wn, buf = ctx.saved_tensors
        rows, cols = ctx.shape
        dev = wn.device
        g = g.contiguous()
        part = torch.empty(_lib.load().apn_spectral_norm_blocks(rows, cols), dtype=torch.float64, device=dev)
        gw = torch.empty_like(wn)
        used_u, used_v, sigma = buf[rows + cols:2 * rows + cols], buf[2 * rows + cols:-1], buf[-1:]
        _call("apn_spectral_norm_grad", dev, rows, cols, g.data_ptr(), wn.data_ptr(), sigma.data_ptr(),
              used_u.data_ptr(), used_v.data_ptr(), part.data_ptr(), gw.data_ptr())
        return gw, None


class SpectralNorm(_SpectralNorm):
    """torch's parametrisation module with its forward on the extension's kernels."""

    def forward(self, weight):
        if (weight.ndim > 1 and weight.is_cuda and weight.dtype == torch.float32 and self.dim == 0
                and self.n_power_iterations == 1 and self._u.is_contiguous() and self._v.is_contiguous()):
            return _Normalise.apply(weight, self)
        return super().forward(weight)


def spectral_norm(module, name="weight", n_power_iterations=1, eps=1e-12):
    weight = getattr(module, name)
    parametrize.register_parametrization(module, name, SpectralNorm(weight, n_power_iterations, 0, eps))
    return module


# ---------------------------------------------------------------- all layers of a network at once
def _ptrs(tensors, ctype=ctypes.c_void_p):
    return (ctype * len(tensors))(*[t.data_ptr() for t in tensors])


class _NormaliseMany(torch.autograd.Function):
    """`_Normalise` for a list of layers in 3 + 2 launches in all (apn_spectral_norm_many / _grad_many)."""

    @staticmethod
    def forward(ctx, owners, *weights):
        dev = weights[0].device
        ws = [w.detach().contiguous() for w in weights]
        rows = [w.shape[0] for w in ws]
        cols = [w[0].numel() for w in ws]
        n = len(ws)
        # one buffer for every layer's scratch | u_used | v_used | sigma (as _Normalise lays them out), one for the outputs
        sizes = [2 * (r + c) + 1 for r, c in zip(rows, cols)]
        buf = torch.empty(sum(sizes), device=dev)
        bufs, o = [], 0
        for sz in sizes:
            bufs.append(buf[o:o + sz])
            o += sz
        wn_flat = torch.empty(sum(w.numel() for w in ws), device=dev)
        wns, o = [], 0
        for w in ws:
            wns.append(wn_flat[o:o + w.numel()].view_as(w))
            o += w.numel()
        scratch = [b[:r + c] for b, r, c in zip(bufs, rows, cols)]
        used_u = [b[r + c:2 * r + c] for b, r, c in zip(bufs, rows, cols)]
        used_v = [b[2 * r + c:-1] for b, r, c in zip(bufs, rows, cols)]
        sigma = [b[-1:] for b in bufs]
        training = owners[0].training
        eps = float(owners[0].eps)
        I = ctypes.c_int * n
        _call("apn_spectral_norm_many", dev, n, I(*rows), I(*cols), _ptrs(ws), int(training), eps,
              _ptrs([o_._u for o_ in owners]), _ptrs([o_._v for o_ in owners]), _ptrs(scratch), _ptrs(used_u), _ptrs(used_v),
              _ptrs(sigma), _ptrs(wns))
        ctx.save_for_backward(wn_flat, buf)
        ctx.meta = (rows, cols, sizes, [w.shape for w in ws])
        ctx.set_materialize_grads(False)
        return tuple(wns)

    @staticmethod
    def backward(ctx, *grads):
        wn_flat, buf = ctx.saved_tensors
        rows, cols, sizes, shapes = ctx.meta
        dev = wn_flat.device
        lib = _lib.load()
        sel = [i for i, g in enumerate(grads) if g is not None]
        out = [None] * len(grads)
        if sel:
            offs_b, offs_w, ob, ow = [], [], 0, 0
            for r, c, sz in zip(rows, cols, sizes):
                offs_b.append(ob)
                offs_w.append(ow)
                ob += sz
                ow += r * c
            gs = [grads[i].contiguous() for i in sel]
            wns = [wn_flat[offs_w[i]:offs_w[i] + rows[i] * cols[i]] for i in sel]
            bs = [buf[offs_b[i]:offs_b[i] + sizes[i]] for i in sel]
            rs, cs = [rows[i] for i in sel], [cols[i] for i in sel]
            used_u = [b[r + c:2 * r + c] for b, r, c in zip(bs, rs, cs)]
            used_v = [b[2 * r + c:-1] for b, r, c in zip(bs, rs, cs)]
            sigma = [b[-1:] for b in bs]
            nbs = [lib.apn_spectral_norm_blocks(r, c) for r, c in zip(rs, cs)]
            part_flat = torch.empty(sum(nbs), dtype=torch.float64, device=dev)
            parts, o = [], 0
            for nb in nbs:
                parts.append(part_flat[o:o + nb])
                o += nb
            gws = [torch.empty(shapes[i], device=dev) for i in sel]
            I = ctypes.c_int * len(sel)
            _call("apn_spectral_norm_grad_many", dev, len(sel), I(*rs), I(*cs), _ptrs(gs), _ptrs(wns), _ptrs(sigma),
                  _ptrs(used_u), _ptrs(used_v), _ptrs(parts), _ptrs(gws))
            for i, gw in zip(sel, gws):
                out[i] = gw
        return (None, *out)


def normalise_many(modules, name="weight"):
    """The spectral-normalised `name` of every module of the list, all layers in ONE set of launches -- or None when any
    of them is not a `SpectralNorm` of this module in a state its kernels serve (the caller then reads `module.weight`
    layer by layer, as ever).  Reading a spectral-normalised weight in training mode IS a power iteration: a caller of
    this function must not read `module.weight` as well."""
    owners, weights = [], []
    for m in modules:
        plist = getattr(getattr(m, "parametrizations", None), name, None)
        if plist is None or len(plist) != 1 or not isinstance(plist[0], SpectralNorm):
            return None
        p, w = plist[0], plist.original
        if not (w.ndim > 1 and w.is_cuda and w.dtype == torch.float32 and p.dim == 0 and p.n_power_iterations == 1
                and p._u.is_contiguous() and p._v.is_contiguous()):
            return None
        owners.append(p)
        weights.append(w)
    if not owners or len(owners) > 8 or any(o.training != owners[0].training or o.eps != owners[0].eps for o in owners):
        return None
    return list(_NormaliseMany.apply(owners, *weights))
