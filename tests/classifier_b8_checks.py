"""Shared by the CPU and GPU suites: the classifier mirror against the reference's B = 8 training-mode goldens (G17: forward
+ SmoothCE + backward; G18: one whole train_one_epoch iteration), parameter by parameter in relative L2."""
import numpy as np
import torch

import golden_inputs as GI


def height_channel(pos):
    return pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]


def no_dropout(model):
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return model


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def norm_floor(golden, prefix):
    """Gradients below 1e-3 of the median parameter-gradient norm are analytic zeros -- a per-channel shift in front of a
    training-mode BatchNorm (the last stage's BatchNorm shift ahead of the batch-normalised head: reference norm 2e-7 against
    0.14 .. 25 for all others; in the generator every convolution bias and grouper `affine_beta` ahead of a BatchNorm:
    1e-11 .. 1.3e-7 against 1.7e-5 .. 3e-2).  Both sides hold rounding noise there: such a parameter is not held to a
    relative bar; it must stay what it is -- below 1e-2 of the median norm (`gradient_errors` reports it under "zero: ").
    (The gap between the two groups is two decades or more in every golden.)"""
    return 1e-3 * float(np.median([float(golden[k]) for k in golden.files if k.startswith(f"{prefix}_gnorm/")]))


def _rel_sampled(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300))


def gradient_errors(model, golden, prefix):
    """({parameter name: relative L2 of its gradient against the golden's sampled entries}, {name: relative error of its
    norm}) over the parameters with a real gradient; an analytic zero (see norm_floor) appears as "zero: <name>" with its
    norm over 1e-2 of the median norm (so that the same bars apply: it must stay below 1)."""
    errs, norms = {}, {}
    floor = norm_floor(golden, prefix)
    for name, q in model.named_parameters():
        if q.grad is None:
            continue
        g = q.grad.detach().cpu().numpy().reshape(-1)
        want = golden[f"{prefix}_grad/{name}"]
        ref_norm = float(golden[f"{prefix}_gnorm/{name}"])
        got_norm = float(np.linalg.norm(g.astype(np.float64)))
        if ref_norm < floor:
            errs["zero: " + name] = 1e-3 * got_norm / (10.0 * floor)        # < 1e-3 <=> the norm is below 1e-2 of the median
            continue
        errs[name] = _rel_sampled(g[GI.gradient_sample_index(name, g.size)], want)
        norms[name] = abs(got_norm - ref_norm) / ref_norm
    return errs, norms


def run_g17(model, dev, golden):
    """-> dict(logits, loss, grad_x: relative errors; grads: {name: rel L2}; norms; bn: worst running-mean error)."""
    model = no_dropout(model).to(dev).train()
    pos = torch.from_numpy(GI.unit_sphere_cloud(8, 1024, seed=171)).to(dev)
    x = torch.cat([pos, height_channel(pos)], -1).transpose(1, 2).contiguous().requires_grad_(True)
    target = torch.from_numpy(golden["g17_target"]).to(dev)
    logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, target)
    loss.backward()
    grads, norms = gradient_errors(model, golden, "g17")
    bn = max(_rel(b.detach().cpu().numpy(), golden[f"g17_bn/{n}"]) for n, b in model.named_buffers() if n.endswith("running_mean"))
    return dict(logits=_rel(logits.detach().cpu().numpy(), golden["g17_logits"]), loss=abs(loss.item() / float(golden["g17_loss"]) - 1),
                grad_x=_rel(x.grad.cpu().numpy(), golden["g17_grad_x"]), grads=grads, norms=norms, bn=bn)


def run_g18(model, dev, golden, grad_bar):
    """One ClassifierStep on the golden's batch.  Besides the relative errors: `step_mismatch` = the number of sampled
    weights whose reference gradient exceeds ten times the gradient tolerance (`grad_bar` x the tensor's rms) and that did
    NOT take the reference's AdamW step (first step: lr * sign(gradient) + decay -- so this is exact or a flipped sign)."""
    from adaptpoint_amd.gan import ClassifierStep
    model = no_dropout(model).to(dev)
    pos = torch.from_numpy(GI.unit_sphere_cloud(8, 2048, seed=181))
    points = torch.cat([pos, height_channel(pos)], -1).to(dev)
    target = torch.from_numpy(golden["g18_target"]).to(dev)
    taps = {}
    step = ClassifierStep(model, grad_sync=lambda grads: taps.update(
        {n: q.grad.detach().clone() for n, q in model.named_parameters()}))     # (called before clipping)
    before = {n: q.detach().clone() for n, q in model.named_parameters()}
    logits, loss = step(points, target, choice=golden["g18_choice"])
    grads, checked, mismatch = {}, 0, 0
    floor = norm_floor(golden, "g18")
    for name, q in model.named_parameters():
        g = taps[name].cpu().numpy().reshape(-1)
        idx = GI.gradient_sample_index(name, g.size)
        want = golden[f"g18_grad/{name}"]
        if float(golden[f"g18_gnorm/{name}"]) < floor:                      # an analytic zero: not held to a relative bar
            grads["zero: " + name] = 1e-3 * float(np.linalg.norm(g.astype(np.float64))) / (10.0 * floor)
        else:
            grads[name] = _rel_sampled(g[idx], want)
        moved = (q.detach() - before[name]).cpu().numpy().reshape(-1)[idx]
        # (an analytic-zero gradient -- see norm_floor -- is rounding noise on both sides: its signs are not "sure")
        rms = max(float(np.sqrt(np.mean(want.astype(np.float64) ** 2))), floor / g.size ** 0.5)
        sure = np.abs(want) > 10 * grad_bar * rms
        checked += int(sure.sum())
        mismatch += int((np.abs(moved[sure] - golden[f"g18_step/{name}"][sure]) > 2e-6).sum())
    bn = max(_rel(b.detach().cpu().numpy(), golden[f"g18_bn/{n}"]) for n, b in model.named_buffers() if n.endswith("running_mean"))
    return dict(logits=_rel(logits.cpu().numpy(), golden["g18_logits"]), loss=abs(loss.item() / float(golden["g18_loss"]) - 1),
                grads=grads, bn=bn, steps_checked=checked, step_mismatch=mismatch)


def worst(d, k=3):
    return sorted(((v, n) for n, v in d.items()), reverse=True)[:k]
