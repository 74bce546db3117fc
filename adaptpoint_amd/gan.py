"""The two training steps of AdaptPoint over the gfx950 operators (BASELINE configs[2]-[4]).

Host-side restatement of what the reference's trainer does per batch
(examples/classification/train_autoaug.py) -- only the per-iteration arithmetic, none of its data
loading, logging, checkpointing or epoch control:

  `resample`            the classifier loop's FPS(N -> 1200) + random 1024 + gather   (:481-501)
  `ClassifierStep`      forward, SmoothCE, backward, grad-clip 10, AdamW, zero_grad   (:502-512)
  `feedback_loss`       |1 - exp(L_fake - rho * L_real)| through the eval-mode classifier
                        (openpoints/function_adaptpoint/ganloss_cls.py:31-65)
  `GanStep`             generator step (BCE vs 0.9 through D + feedback), then discriminator step
                        (BCE real vs 0.9, fake vs 0.1)                               (:133-204)

Hyper-parameters default to cfgs/scanobjectnn/pointnext-s_adaptpoint_1.yaml:63-71 and
cfgs/scanobjectnn/default.yaml:36-56.
"""
import contextlib

import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import graphs
from .graphs import mark, mark_grad
from .layers import furthest_point_sample


def resample(points, npoints, in_channels, choice=None):
    """train_autoaug.py:481-501.  points (B,N,C): when N > npoints, FPS down to `point_all`
    (1200 for 1024, 4800 for 4096, 8192 for 8192), keep a random `npoints` of those -- ONE draw
    of `np.random.choice(point_all, npoints, False)` for the whole batch -- and return
    pos (B,npoints,3), x (B,in_channels,npoints).  `choice` overrides the draw (tests)."""
    B, N, C = points.shape
    if N <= npoints:
        return points[:, :, :3].contiguous(), points[:, :, :in_channels].transpose(1, 2).contiguous()
    point_all = {1024: 1200, 4096: 4800, 8192: 8192}.get(npoints)
    if point_all is None:
        raise NotImplementedError(f"no resampling rule for npoints={npoints}")
    point_all = min(point_all, N)
    points = points.contiguous()
    fidx = furthest_point_sample(points[:, :, :3].contiguous(), point_all)
    if torch.is_tensor(choice):
        # already on the device (int32, npoints indices in [0, point_all)): no host-to-device copy, so the step can be
        # captured in a hipGraph
        if choice.dtype != torch.int32 or choice.numel() != npoints or choice.device != points.device:
            raise RuntimeError("resample: a tensor `choice` must be int32, on the points' device, npoints long")
    else:
        if choice is None:
            choice = np.random.choice(point_all, npoints, False)
        choice = np.asarray(choice)
        if choice.size != npoints or choice.min() < 0 or choice.max() >= point_all:
            raise RuntimeError("resample: choice must hold npoints indices in [0, point_all)")
        choice = torch.from_numpy(choice.astype(np.int32)).to(points.device)
    pos = torch.empty(B, npoints, 3, dtype=torch.float32, device=points.device)
    x = torch.empty(B, in_channels, npoints, dtype=torch.float32, device=points.device)
    ops.resample_points_wrapper(B, N, C, point_all, npoints, in_channels, points, fidx, choice, pos, x)
    return pos, x


class ClassifierStep:
    """One iteration of `train_one_epoch` (train_autoaug.py:471-512) for step_per_update = 1."""

    def __init__(self, model, lr=2e-3, weight_decay=0.05, grad_norm_clip=10.0, npoints=1024,
                 in_channels=4, optimizer=None, grad_sync=None):
        self.model = model
        # grad_sync(list of .grad tensors): called before clipping and the optimizer step -- under data parallelism
        # `adaptpoint_amd.dp.allreduce_mean_`, what the reference's DistributedDataParallel wrapper of the classifier
        # does (train_autoaug.py:275-282, with BatchNorm converted to SyncBatchNorm there)
        self.grad_sync = grad_sync
        self.npoints, self.in_channels, self.clip = npoints, in_channels, grad_norm_clip
        self.opt = optimizer or torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)

    def __call__(self, points, target, choice=None):
        """points (B,N,C>=in_channels), target (B,) -> (logits, loss), both detached: a caller that keeps them must not
        keep this step's autograd graph alive with them (its AccumulateGrad nodes would belong to THIS step's stream; a
        later capture of the step then synchronises with that stream and dies inside hipStreamEndCapture -- graphs.py)."""
        self.model.train()
        pos, x = resample(points, self.npoints, self.in_channels, choice)
        logits, loss = self.model.get_logits_loss({'pos': pos, 'x': x}, target)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync([q.grad for q in self.model.parameters() if q.grad is not None])
        if self.clip is not None and self.clip > 0:
            nn.utils.clip_grad_norm_(self.model.parameters(), self.clip, norm_type=2)
        self.opt.step()
        self.model.zero_grad()
        return logits.detach(), loss.detach()


class _frozen:
    """Parameters of `module` without requires_grad inside the block: a graph recorded there holds no gradient
    path to them, so the backward passes of the fused blocks skip their weight-gradient kernels (the reference's
    `g_loss.backward()` computes the classifier's weight gradients and discards them, train_autoaug.py:165-170)."""

    def __init__(self, module):
        self.params = [q for q in module.parameters() if q.requires_grad]

    def __enter__(self):
        for q in self.params:
            q.requires_grad_(False)

    def __exit__(self, *exc):
        for q in self.params:
            q.requires_grad_(True)


def feedback_loss(classifier, criterion, real, fake, label, hard_ratio, batched=True, frozen=False, loss_real=None):
    """ganloss_cls.py:31-65: how much harder the augmented clouds are than the real ones for the
    CURRENT classifier, pulled towards `hard_ratio`: |1 - exp(L(fake) - hard_ratio * L(real))|.
    The classifier runs in eval mode (running BatchNorm statistics, no dropout), so every cloud
    is processed independently of its batch: `batched` stacks fake and real into ONE 2B pass
    (their FPS chains then run side by side) with the same result per cloud.
    `real` / `fake`: dicts with 'pos' (B,N,3) and 'x' (B,C,N).  frozen: only the inputs receive gradients (what
    the generator step needs).  loss_real: the real clouds' loss computed ahead (`real_loss_ahead`); only the fake
    clouds then pass through the classifier here."""
    classifier.eval()
    if loss_real is not None:
        with (_frozen(classifier) if frozen else contextlib.nullcontext()):
            loss_fake = criterion(classifier(fake), label.long())
        return torch.abs(1 - torch.exp(loss_fake - hard_ratio * loss_real)), loss_fake, loss_real
    with (_frozen(classifier) if frozen else contextlib.nullcontext()):
        if batched:
            both = classifier({'pos': torch.cat([fake['pos'], real['pos']], 0),
                               'x': torch.cat([fake['x'], real['x']], 0)})
            pred_fake, pred_real = both.chunk(2, 0)
        else:
            pred_fake = classifier(fake)
            pred_real = classifier(real)
    loss_fake = criterion(pred_fake, label.long())
    loss_real = criterion(pred_real, label.long())
    return torch.abs(1 - torch.exp(loss_fake - hard_ratio * loss_real)), loss_fake, loss_real


@torch.no_grad()
def real_loss_ahead(classifier, criterion, real, label):
    """The real clouds' half of the feedback loss on its own: in the generator step it is a constant (the real clouds
    need no gradient and the classifier's weights get none), so it needs no backward pass -- the 2B batched pass runs
    its backward over both halves -- and, a function of the batch alone, it can run before / beside the generator."""
    classifier.eval()
    return criterion(classifier(real), label.long())


def hard_ratio_at(epoch, epochs, start=3.0, end=3.0):
    """ganloss_cls.py:35-36."""
    return start + (end - start) * epoch / epochs


class GanStep:
    """One iteration of `train_gan` (train_autoaug.py:133-204).

    Differences from the reference, none of them numerical for the three networks' updates:
    the generator step differentiates with respect to the generator's parameters only (the
    reference's `g_loss.backward()` also fills the discriminator's and the classifier's .grad,
    which it then overwrites / zeroes without using); the 4th input channel keeps the real cloud's
    height, as the in-place `points[:, :, :3] = gen_imgs` leaves it (:155)."""

    def __init__(self, generator, discriminator, classifier, criterion, lr_generator=1e-4,
                 lr_discriminator=4e-4, betas=(0.5, 0.999), hard_ratio=3.0, feedback_ratio=1.0,
                 in_channels=4, batched_feedback=True, capturable=False, grad_sync=None, overlap=False):
        self.G, self.D, self.C = generator, discriminator, classifier
        # grad_sync(list of .grad tensors): called before each optimizer step -- under data parallelism
        # `adaptpoint_amd.dp.allreduce_mean_`, what the reference's DistributedDataParallel wrappers of the
        # generator and the discriminator do (train_autoaug.py:98-102; their BatchNorm stays per rank)
        self.grad_sync = grad_sync
        self.criterion = criterion
        # capturable: optimizer state on the device, so that the whole step can be a hipGraph -- and PyTorch's
        # fused multi-tensor Adam (one launch per ~parameter group instead of ~15 foreach launches: 1.4 -> 0.2 ms
        # of the step; the same update rule, torch/optim/adam.py)
        kw = dict(capturable=True, fused=True) if capturable else {}
        self.opt_g = torch.optim.Adam(generator.parameters(), lr=lr_generator, betas=betas, **kw)
        self.opt_d = torch.optim.Adam(discriminator.parameters(), lr=lr_discriminator, betas=betas, **kw)
        self.bce = nn.BCELoss()
        self.hard_ratio, self.feedback_ratio = hard_ratio, feedback_ratio
        self.in_channels, self.batched_feedback = in_channels, batched_feedback
        # overlap: the step runs as TWO lanes -- the caller's stream and one side stream forked from and joined to it, so
        # that a capture of the step holds them as parallel branches (a replayed hipGraph runs two branches
        # concurrently on this stack; a third one waits: scripts/experiment_graph_branches.py):
        #   generator forward    lane 1: the imitator's feature path            lane 2: the real clouds' classifier pass,
        #                        (stages, decoders, masking branch)                      then the anchor head
        #   after it             lane 1: the feedback pass (classifier          lane 2: D(gen), then the discriminator's
        #                        forward on the generated clouds)                        own step (two forwards, backward)
        #   backward             autograd runs a node on its forward's stream: the same split, mirrored
        # Same arithmetic in the same order per tensor: the discriminator's power-iteration state is advanced by
        # D(gen), D(real), D(gen.detach()) in that order on either schedule, its weights change after the generator's
        # backward has read them, and the draws (dropout, generator switches) are requested in the same sequence.
        self.overlap = overlap

    # + the discriminator chain on the second lane, always.  ("real" was held back for a while: replayed, the generator's
    # updates came out wrong -- the cause was the LDS-atomic FPS step returning wrong picks beside other kernels, not the
    # schedule; csrc/fps.hip, fps_default_algo.)
    OVERLAP_PARTS = frozenset(("imitator", "real"))

    def _discriminator_losses(self, xyz, gen, real_t, fake_t):
        """train_autoaug.py:181-196 up to the optimizer step: two forwards (each one spectral-norm power iteration),
        the mean of the two BCE terms, backward into the discriminator's .grad."""
        real_loss = self.bce(self.D(xyz), real_t)
        fake_loss = self.bce(self.D(gen.detach()), fake_t)
        d_loss = (real_loss + fake_loss) / 2
        self.opt_d.zero_grad()
        d_loss.backward()
        return d_loss

    def __call__(self, points, label, noise=None, device_noise=False):
        """points (B,N,C>=in_channels) with xyz first, label (B,) -> dict of the step's scalars
        (0-dim tensors, no host sync) and the generated clouds.  device_noise: draw the generator's
        random switches from the device generator instead of the CPU one (same distributions; no
        host-to-device copy, so the step can be captured in a hipGraph)."""
        G, D = self.G, self.D
        G.train()
        D.train()
        self.C.eval()
        B = points.shape[0]
        xyz = points[:, :, :3].contiguous()
        real_t = torch.full((B, 1), 0.9, device=points.device)
        fake_t = torch.full((B, 1), 0.1, device=points.device)

        # ---- generator
        if noise is None and device_noise:
            from .augmentor import draw_noise_on
            noise = draw_noise_on(xyz.device, B, xyz.shape[1], G.num_anchor)
        mark("step: start")
        overlap = bool(self.overlap) and points.is_cuda
        parts = self.overlap if isinstance(self.overlap, (set, frozenset)) else self.OVERLAP_PARTS
        if not set(parts) <= set(self.OVERLAP_PARTS):
            raise ValueError(f"GanStep(overlap=...): parts must be among {sorted(self.OVERLAP_PARTS)} (the experimental "
                             "'plan' / 'pyramid' / 'real_sync' schedules live in scripts/, not in the library)")
        loss_real = real = s_real = None
        if self.feedback_ratio > 0:
            real = {'pos': xyz, 'x': points[:, :, :self.in_channels].transpose(1, 2).contiguous()}
        if overlap and real is not None and "real" in parts:
            # the real clouds' classifier pass (no gradient, a function of the batch alone) opens the second lane, beside
            # the generator's forward; the feedback pass proper then carries the B generated clouds only, forward and backward
            s_real = graphs.fork(graphs.LANE2, points.device, xyz, real['x'], label)
            with torch.cuda.stream(s_real):
                loss_real = real_loss_ahead(self.C, self.criterion, real, label)
                mark("real clouds' classifier pass done (second lane)")
            joins_before = graphs.joins(s_real)
        with graphs.overlapping(overlap and "imitator" in parts):
            _, gen = G(xyz) if noise is None else G(xyz, noise)
        if s_real is not None and graphs.joins(s_real) == joins_before:
            # nobody joined the lane meanwhile (the imitator forks and joins it only when its coordinates need no gradient
            # and "imitator" is among the parts): the main stream reads loss_real below, so it joins the lane HERE, while
            # the real clouds' pass is still the lane's tail -- before the discriminator chain is queued behind it
            graphs.join(s_real, loss_real)
        mark("generator forward done")
        mark_grad(gen, "backward: dL/d(generated clouds) formed (feedback + D backward done)")
        if overlap:
            dev = points.device
            # the discriminator's three forwards, in the reference's order, on the second lane: D(gen) for the generator's
            # loss (its backward then runs there too, beside the feedback pass's), then the discriminator's own step;
            # the first lane meanwhile runs the feedback pass (index pyramid, 2B classifier forward)
            s_dis = graphs.fork(graphs.LANE2, dev, gen, xyz, real_t, fake_t)
            with torch.cuda.stream(s_dis):
                with _frozen(D):
                    g_raw = self.bce(D(gen), real_t)
                mark("D(gen) forward done (second lane)")
                d_loss = self._discriminator_losses(xyz, gen, real_t, fake_t)
                mark("discriminator losses + backward done (second lane)")
        else:
            # (the discriminator's weights are frozen inside this forward: the generator step needs dL/d(gen) only, so its
            # backward launches none of D's weight-gradient kernels -- and no gradient accumulator of D lives on this stream)
            with _frozen(D):
                g_raw = self.bce(D(gen), real_t)
            mark("D(gen) forward done")
        g_loss, fb = g_raw, None
        if self.feedback_ratio > 0:
            tail = points[:, :, 3:self.in_channels]
            fake = {'pos': gen, 'x': torch.cat([gen, tail], -1).transpose(1, 2).contiguous()}
            fb, _, _ = feedback_loss(self.C, self.criterion, real, fake, label, self.hard_ratio,
                                     self.batched_feedback, frozen=True, loss_real=loss_real)
            mark("feedback forward done")
            if overlap:
                graphs.join(s_dis, g_raw, d_loss)
            g_loss = g_raw + fb * self.feedback_ratio
        elif overlap:
            graphs.join(s_dis, g_raw, d_loss)
        self.opt_g.zero_grad()
        torch.autograd.backward(g_loss, inputs=[q for q in G.parameters() if q.requires_grad])
        mark("generator-step backward done")
        if self.grad_sync is not None:
            self.grad_sync([q.grad for q in G.parameters() if q.grad is not None])
        self.opt_g.step()
        mark("generator optimizer done")

        # ---- discriminator (two forwards, as the reference: each is one spectral-norm power iteration)
        if not overlap:
            d_loss = self._discriminator_losses(xyz, gen, real_t, fake_t)
            mark("discriminator losses + backward done")
        if self.grad_sync is not None:
            self.grad_sync([q.grad for q in D.parameters() if q.grad is not None])
        self.opt_d.step()
        mark("step: end")
        return {'g_loss_raw': g_raw.detach(), 'feedback_loss': None if fb is None else fb.detach(),
                'g_loss': g_loss.detach(), 'd_loss': d_loss.detach(), 'gen': gen.detach()}
