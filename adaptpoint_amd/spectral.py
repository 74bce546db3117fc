"""Spectral normalisation on the gfx950 kernels of csrc/spectral.hip.

`spectral_norm(module)` registers the same parametrisation as `torch.nn.utils.parametrizations.spectral_norm`
(what every layer of the reference's `PointDiscriminator1` carries, point_discriminator.py:17-73, 149-191) -- same
buffers `_u` / `_v`, same state_dict keys, same construction (15 start-up power iterations) -- whose forward runs as
three launches and whose backward as two where PyTorch composes ~13 and ~7 small ones; the joint GAN step evaluates
21 of these forwards and 14 backwards.  Anything the kernels do not serve (CPU tensors, other dtypes, more than one
power iteration, a permuted `dim`) takes PyTorch's own path of the parent class.
"""
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
