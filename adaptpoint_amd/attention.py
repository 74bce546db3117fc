"""Multi-head self-attention over the points of a cloud, head_dim 16.

`attention(q, k, v, heads)` computes `softmax(q k^T / sqrt(16)) v` per head on (B, M, heads*16)
tensors -- lines 460-474 of the imitator's `Anchor_selfattention`
(openpoints/models_adaptpoint/generator_component4_15.py:434-481) -- without the (B,H,M,M) score
tensor (csrc/attention.hip).  `AnchorSelfAttention` mirrors the module (same sub-module names, so
a reference state_dict loads unchanged).
"""
import torch
import torch.nn as nn

from . import _lib
from .fused import _call

HEAD_DIM = 16


# calls that ran the composition instead of the kernel, by reason (the imitator's 4-anchor attention, M = 4, is the one
# caller: 128 rows of 16 numbers -- a launch of its own would cost what the five library launches do)
COMPOSED_CALLS = {}


def supported(q, heads):
    return (q.is_cuda and q.dim() == 3 and q.shape[-1] == heads * HEAD_DIM and q.shape[1] % 32 == 0
            and q.shape[1] > 0)


def _reference(q, k, v, heads):
    """The reference's composition (:460-474), used for shapes the kernel does not cover."""
    B, M, C = q.shape
    d = C // heads
    qh, kh, vh = (t.reshape(B, M, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    attn = qh @ kh.transpose(-2, -1)
    attn = attn / d ** 0.5
    attn = attn.softmax(dim=-1)
    return (attn @ vh).permute(0, 2, 1, 3).reshape(B, M, C)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = (t.contiguous().float() for t in (q, k, v))
        B, M, C = q.shape
        dev = q.device
        need_bwd = any(ctx.needs_input_grad[:3])
        n = B * heads * M * 32
        img = torch.empty((6 if need_bwd else 3) * n, dtype=torch.bfloat16, device=dev)
        out = torch.empty(B, M, C, dtype=torch.float32, device=dev)
        lse = torch.empty(B, heads, M, dtype=torch.float32, device=dev)
        _call("apn_attention_prep", dev, B, M, heads, q.data_ptr(), k.data_ptr(), v.data_ptr(),
              img.data_ptr(), 1 if need_bwd else 0)
        _call("apn_attention_fwd", dev, B, M, heads, img.data_ptr(), out.data_ptr(), lse.data_ptr())
        if need_bwd:
            ctx.save_for_backward(out, lse, img)
        ctx.dims = (B, M, C, heads)
        return out

    @staticmethod
    def backward(ctx, g):
        out, lse, img = ctx.saved_tensors
        B, M, C, heads = ctx.dims
        dev = out.device
        g = g.contiguous().float()
        n = B * heads * M * 32
        scratch = torch.empty(2 * n * 2 + B * heads * M * 4, dtype=torch.uint8, device=dev)
        dq, dk, dv = (torch.empty(B, M, C, dtype=torch.float32, device=dev) for _ in range(3))
        _call("apn_attention_bwd", dev, B, M, heads, img.data_ptr(), out.data_ptr(), lse.data_ptr(),
              g.data_ptr(), scratch.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr())
        return dq, dk, dv, None


SMALL_MAX = 32       # apn_attention_small_max(): few points go to the one-wave float32 kernel


class _AttentionSmall(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = (t.contiguous().float() for t in (q, k, v))
        B, M, C = q.shape
        out = torch.empty(B, M, C, dtype=torch.float32, device=q.device)
        _call("apn_attention_small_fwd", q.device, B, M, heads, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr())
        ctx.save_for_backward(q, k, v)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v = ctx.saved_tensors
        B, M, C = q.shape
        g = g.contiguous().float()
        dq, dk, dv = (torch.empty_like(q) for _ in range(3))
        _call("apn_attention_small_bwd", q.device, B, M, ctx.heads, q.data_ptr(), k.data_ptr(), v.data_ptr(), g.data_ptr(),
              dq.data_ptr(), dk.data_ptr(), dv.data_ptr())
        return dq, dk, dv, None


def attention(q, k, v, heads):
    """softmax(q k^T / sqrt(16)) v per head; q, k, v (B, M, heads*16) -> (B, M, heads*16)."""
    if not q.is_cuda:
        raise RuntimeError("adaptpoint_amd.attention needs CUDA/HIP tensors: the product path has no "
                           "CPU fallback")
    _lib.load()
    if (q.dim() == 3 and q.shape[-1] == heads * HEAD_DIM and 0 < q.shape[1] <= SMALL_MAX and q.shape[1] % 32
            and 0 < q.shape[0] <= 65535):
        return _AttentionSmall.apply(q, k, v, heads)   # few points (the 4-anchor head): one wave per (cloud, head)
    if not supported(q, heads):                      # other head dims / ragged M > 32: composed on the GPU, and counted
        why = "M=%d not a multiple of 32" % q.shape[1] if q.shape[-1] == heads * HEAD_DIM else "head dim %d" % (q.shape[-1] // heads)
        COMPOSED_CALLS[why] = COMPOSED_CALLS.get(why, 0) + 1
        return _reference(q, k, v, heads)
    return _Attention.apply(q, k, v, heads)


def _mean_over_points(xyz):
    """xyz (B,M,3) -> (B,1,3): the mean over the M points (generator_component4_15.py:449: torch.mean(xyz, dim=1)).
    On the GPU in two stages of 64: at M = 2048 PyTorch's strided reduction over dim 1 becomes a multi-block one, whose
    semaphore buffer is zeroed by a MEMSET before every launch -- captured into a hipGraph that node replays correctly
    once and with garbage afterwards (adaptpoint_amd/graphs.py): the joint step at N = 2048 then trained on stale
    centres from its second replay on.  (Contiguous rows of 64, then of M / 64: one block per output, no semaphores.)"""
    B, M, C = xyz.shape
    if not xyz.is_cuda or M % 64:
        return torch.mean(xyz, dim=1, keepdim=True)
    t = xyz.transpose(1, 2).reshape(B, C, M // 64, 64).sum(-1).sum(-1)
    return (t / M).unsqueeze(1)


class AnchorSelfAttention(nn.Module):
    """generator_component4_15.py:434-481 (`Anchor_selfattention`)."""

    def __init__(self, dim, head_num, fused=True):
        super().__init__()
        self.fused = fused            # False: the reference's composition (materialised scores), on the GPU
        self.dim = dim
        self.head_num = head_num
        self.head_dim = int(self.dim // self.head_num)
        self.to_qkv = nn.Linear(self.dim, self.dim * 3, bias=False)
        self.pos_embedding = nn.Sequential(nn.Conv1d(3, self.dim, 1), nn.BatchNorm1d(self.dim))
        self.res = nn.Sequential(nn.Conv1d(self.dim, self.dim, 1), nn.BatchNorm1d(self.dim))

    def forward(self, x, xyz=None):
        """x (B,M,C), xyz (B,M,3) -> (B,M,C)."""
        gravity_center = _mean_over_points(xyz)
        relative_xyz = xyz - gravity_center
        from . import pointwise
        on = self.fused and x.is_cuda
        emb = pointwise.conv_then_bn(relative_xyz.permute(0, 2, 1).contiguous(), self.pos_embedding, allow=on).permute(0, 2, 1)
        q, k, v = (pointwise.linear_nobias(x, self.to_qkv) if on else self.to_qkv(x)).chunk(3, dim=-1)
        q, k, v = q + emb, k + emb, v + emb
        if self.fused or not x.is_cuda:
            o = attention(q, k, v, self.head_num)      # raises on CPU tensors: no CPU fallback
        else:
            o = _reference(q, k, v, self.head_num)
        return pointwise.transpose12(pointwise.conv_then_bn(pointwise.transpose12(o), self.res, allow=on))
