"""Launchable forms of BASELINE configs[2]-[4]: the classifier step, the AdaptPoint joint step and the two together
(one `train_gan` iteration + one `train_one_epoch` iteration on the clouds it generated: what the reference's epoch loop
does per batch, examples/classification/train_autoaug.py:368-394), single process or one process per GPU.

Under `torch.distributed` the networks are set up the way the reference sets them up:

  classifier   `torch.nn.SyncBatchNorm.convert_sync_batchnorm` + DistributedDataParallel (train_autoaug.py:275-282;
               main.py:27 forces SyncBatchNorm at world_size > 1)  ->  `sync_batchnorm_(model)`: the fused set-abstraction
               blocks all-reduce their BatchNorm sums themselves (`SetAbstraction.sync_bn`), every other BatchNorm becomes
               `dp.SyncBatchNormAllReduce`; gradients averaged by ONE flat all-reduce before clipping (`grad_sync`)
  generator,   DistributedDataParallel WITHOUT converted BatchNorm (train_autoaug.py:98-102): per-rank statistics, one
  discriminator   flat gradient all-reduce each before their Adam steps

Every rank builds identical weights (name-seeded or `torch.manual_seed`), draws its own shard of clouds
(`dp.shard_seed`), and the step functions are the ones the single-process benches and tests use
(`gan.ClassifierStep`, `gan.GanStep`) -- the only difference is `grad_sync` and the BatchNorm exchange.
`bench.py --workload classifier|gan|adaptpoint` and the world-2 gloo tests drive `build()`.
"""
import contextlib

import torch
import torch.distributed as dist

from . import dp

WORKLOADS = ("classifier", "gan", "adaptpoint")


def sync_batchnorm_(model):
    """The classifier under data parallelism: BatchNorm statistics span all ranks.  Fused set-abstraction blocks
    exchange their sums inside their own launches (sync_bn=True: four small all-reduces per block and step); all
    other BatchNorm modules are replaced by `dp.SyncBatchNormAllReduce`.  Returns (model, fused blocks, converted)."""
    from .set_abstraction import SetAbstraction
    own = []
    for m in model.modules():
        if isinstance(m, SetAbstraction) and m.fused and not m.is_head and not m.all_aggr and m._fused_parts() is not None:
            m.sync_bn = True
            own.append(m)
    keep = {id(b) for m in own for b in m.convs.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm)}

    def convert(mod):
        out = mod
        if (isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(mod, dp.SyncBatchNormAllReduce)
                and id(mod) not in keep):
            out = dp.convert_sync_batchnorm(mod)
        for name, child in list(mod.named_children()):
            new = convert(child)
            if new is not child:
                out.add_module(name, new)
        return out
    model = convert(model)
    converted = sum(isinstance(m, dp.SyncBatchNormAllReduce) for m in model.modules())
    return model, len(own), converted


class count_collectives(contextlib.AbstractContextManager):
    """Counts the collectives issued inside the block, by kind (`torch.distributed.all_reduce` is what this path uses)."""

    def __init__(self):
        self.calls = {}

    def __enter__(self):
        self._orig = dist.all_reduce

        def counted(t, *a, **k):
            key = "all_reduce"
            self.calls[key] = self.calls.get(key, 0) + 1
            self.calls["bytes"] = self.calls.get("bytes", 0) + t.numel() * t.element_size()
            return self._orig(t, *a, **k)
        dist.all_reduce = counted
        return self

    def __exit__(self, *exc):
        dist.all_reduce = self._orig
        return False


class Job:
    """One rank's share of a workload: networks, this rank's clouds, and `step()` = one iteration."""

    def __init__(self):
        self.nets = {}
        self.taps = None          # when a list: every grad_sync call appends (name, [gradient clones]) AFTER the exchange

    def parameters(self):
        return [q for n in self.nets.values() for q in n.parameters()]


def _clouds(batch, npoints, seed):
    from . import synthetic as GI
    pos = torch.from_numpy(GI.unit_sphere_cloud(batch, npoints, seed=seed))
    # (x, y, z, height): the ScanObjectNN input of cfgs/scanobjectnn/pointnext-s.yaml (in_channels 4)
    return torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1)


def build(workload, dev, batch=32, npoints=1024, fused=True, distributed=None, capturable=False, overlap=False,
          seed=0, name_seeded=False, dropout=True, noise_on_device=None, classes=15, record_grads=False):
    """-> Job.  distributed: None = follow torch.distributed's state.  capturable: optimizer state on the device and
    the generator's draws from the device generator, so that `step()` can be captured into a hipGraph.  name_seeded:
    `fill_parameters_by_name` weights (tests); else `torch.manual_seed(seed)` initialisation, identical on every rank."""
    from .augmentor import AdaptPointAugmentor
    from .discriminator import PointDiscriminator1
    from .gan import ClassifierStep, GanStep
    from .pointnext import PointNextSClassifier, SmoothCrossEntropy, fill_parameters_by_name
    if workload not in WORKLOADS:
        raise ValueError(f"workload must be one of {WORKLOADS}")
    if distributed is None:
        distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if (distributed and dist.is_initialized()) else 0
    world = dist.get_world_size() if (distributed and dist.is_initialized()) else 1
    job = Job()
    if record_grads:
        job.taps = []
    job.workload, job.distributed, job.world, job.rank, job.batch, job.npoints = workload, distributed, world, rank, batch, npoints
    torch.manual_seed(seed)                                   # identical initial weights on every rank
    mk = (lambda m: fill_parameters_by_name(m)) if name_seeded else (lambda m: m)

    def sync(name):
        if not distributed and job.taps is None:
            return None

        def hook(grads, _name=name):
            if distributed:
                dp.allreduce_mean_(grads)
            if job.taps is not None:
                job.taps.append((_name, [g.detach().clone() for g in grads]))
        return hook

    C = mk(PointNextSClassifier(num_classes=classes, fused=fused))
    if not dropout:
        for m in C.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    job.syncbn = None
    if distributed and workload in ("classifier", "adaptpoint"):
        C, n_fused, n_conv = sync_batchnorm_(C)
        job.syncbn = {"fused_blocks_exchanging_their_own_sums": n_fused, "modules_converted": n_conv}
    C = C.to(dev)
    job.nets["classifier"] = C
    opt_kw = dict(capturable=True, fused=True) if (capturable and dev.type == "cuda") else {}
    if workload in ("classifier", "adaptpoint"):
        opt_c = torch.optim.AdamW(C.parameters(), lr=2e-3, weight_decay=0.05, **opt_kw)
        # (more than 1024 points: the trainer's resampler, FPS to 1200 and a random 1024 of them, train_autoaug.py:481-501)
        job.cls_step = ClassifierStep(C, npoints=1024 if npoints > 1024 else npoints, optimizer=opt_c,
                                      grad_sync=sync("classifier"))
    if workload in ("gan", "adaptpoint"):
        G = mk(AdaptPointAugmentor(fused=fused)).to(dev)
        D = mk(PointDiscriminator1(num_classes=classes, fused=fused)).to(dev)
        if not dropout:
            D.drop1.p = D.drop2.p = 0.0
        job.nets["generator"], job.nets["discriminator"] = G, D
        job.gan_step = GanStep(G, D, C, SmoothCrossEntropy(0.3), batched_feedback=fused, capturable=bool(opt_kw),
                               overlap=overlap if fused else False, grad_sync=None)
        # (two hooks: the generator's and the discriminator's exchange are separate collectives, as two DDP wrappers are)
        hooks = {"g": sync("generator"), "d": sync("discriminator")}
        if hooks["g"] is not None:
            turn = [0]

            def both(grads):
                (hooks["g"] if turn[0] % 2 == 0 else hooks["d"])(grads)
                turn[0] += 1
            job.gan_step.grad_sync = both
    job.points = _clouds(batch, npoints, dp.shard_seed(seed, rank)).to(dev)
    # More than 1024 points: the classifier iteration resamples (FPS to 1200, a random 1024 of them; train_autoaug.py:481-501).
    # The reference draws the subset on the host per batch (np.random.choice); here ONE draw per job, kept on the device, so
    # that the step stays capturable (no host-to-device copy inside it) -- the bench line says so.
    job.choice = None
    if npoints > 1024 and workload in ("classifier", "adaptpoint"):
        import numpy as np
        job.choice = torch.from_numpy(np.random.RandomState(1000 + seed).choice(min(1200, npoints), 1024, False).astype(np.int32)).to(dev)
    job.label = ((torch.arange(batch) + 3 * rank) % classes).to(dev)
    device_noise = (dev.type == "cuda") if noise_on_device is None else noise_on_device

    def step(noise=None, choice=None):
        out = {}
        pts = job.points
        if workload in ("gan", "adaptpoint"):
            out = job.gan_step(job.points, job.label, noise=noise, device_noise=device_noise and noise is None)
            if workload == "adaptpoint":
                # the classifier trains on the generated clouds (train_autoaug.py:393: `fake_train_loader`); the 4th
                # channel keeps the real cloud's height (Form_dataset_cls stores `points` with xyz replaced, :155)
                pts = torch.cat([out['gen'], job.points[:, :, 3:]], -1)
        if workload in ("classifier", "adaptpoint"):
            logits, loss = job.cls_step(pts, job.label, choice if choice is not None else job.choice)
            out['cls_loss'] = loss.detach()
        return out
    job.step = step
    return job


def collectives_per_step(job):
    """{all_reduce: count, bytes: payload} of ONE eager step of this rank (state advances by one step)."""
    with count_collectives() as c:
        job.step()
    return dict(c.calls)
