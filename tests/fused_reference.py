"""Plain PyTorch references of the grouped shared-MLP chain for the fused-kernel tests:
an fp32 one (what the reference computes, AMP off) and one that rounds exactly where the
bf16 MFMA path rounds (inputs, weights and the conv2 input), accumulating in float64."""
import torch


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def bf16_round(x):
    """x rounded to bfloat16, returned in float32 (exactly representable)."""
    return x.to(torch.bfloat16).to(torch.float32)


def group(t, idx):
    """t (B,C,N), idx (B,M,K) -> (B,C,M,K)  (== group_points)."""
    B, C, N = t.shape
    _, M, K = idx.shape
    return torch.gather(t, 2, idx.long().reshape(B, 1, M * K).expand(-1, C, -1)).reshape(B, C, M, K)


def chain_grad(p, new_p, f, idx, radius, w1, g1, b1, w2, g2, b2, eps=1e-5, emulate_bf16=False,
               y1_noise=None):
    """Returns out (B,C2,M) and intermediates, in float64 (training-mode BatchNorm).
    y1_noise: optional tensor added to conv1's output (a forward-error model for sensitivity tests)."""
    rnd = _bf if emulate_bf16 else (lambda x: x.double())
    dp = (group(p.transpose(1, 2).contiguous(), idx) - new_p.transpose(1, 2).unsqueeze(-1)) / radius
    x = torch.cat([rnd(dp.float()), group(rnd(f).float() if emulate_bf16 else f, idx).double()], 1)  # (B,35,M,K)
    y1 = torch.einsum('oc,bcmk->bomk', rnd(w1), x)
    if y1_noise is not None:
        y1 = y1 + y1_noise
    # BatchNorm-1's batch statistics as the kernels form them (round 3): per point from the index stage's
    # occurrence statistics, i.e. over y1 with the relative positions NOT rounded to the operand precision
    # (features and weights rounded as the MFMA sees them) -- identical to y1's own statistics in fp32 mode
    ys = y1
    if emulate_bf16:
        ys = torch.einsum('oc,bcmk->bomk', rnd(w1), torch.cat([dp.double(), x[:, 3:]], 1))
        if y1_noise is not None:
            ys = ys + y1_noise
    m1 = ys.mean((0, 2, 3), keepdim=True)
    v1 = ys.var((0, 2, 3), unbiased=False, keepdim=True)
    a1 = torch.relu((y1 - m1) / torch.sqrt(v1 + eps) * g1.double().view(1, -1, 1, 1) + b1.double().view(1, -1, 1, 1))
    if emulate_bf16:
        a1 = _bf(a1.float())
    y2 = torch.einsum('oc,bcmk->bomk', rnd(w2), a1)
    m2 = y2.mean((0, 2, 3), keepdim=True)
    v2 = y2.var((0, 2, 3), unbiased=False, keepdim=True)
    z = (y2 - m2) / torch.sqrt(v2 + eps) * g2.double().view(1, -1, 1, 1) + b2.double().view(1, -1, 1, 1)
    return z.max(-1)[0], dict(y1=y1, a1=a1, y2=y2, m1=m1, v1=v1, m2=m2, v2=v2)


@torch.no_grad()
def chain(*a, **k):
    return chain_grad(*a, **k)
