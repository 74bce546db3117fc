"""CPU suite, world_size 2 over gloo: the PointNeXt-S classifier training step under data
parallelism with SyncBatchNorm semantics (BASELINE configs[4]; the reference forces SyncBN at
world_size > 1, examples/classification/main.py:27).  Property: two ranks, each with half of a
batch, SyncBatchNorm statistics exchanged by all-reduce (adaptpoint_amd.dp.SyncBatchNormAllReduce),
gradients averaged over ranks -- equals ONE process on the whole batch with plain BatchNorm:
same logits per cloud, same gradients, same running statistics."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NPTS = 256


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    import golden_inputs as GI
    pos = torch.from_numpy(GI.unit_sphere_cloud(4, NPTS, seed=17))
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    return pos, x, torch.tensor([1, 7, 3, 12])


def _model():
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    m = fill_parameters_by_name(PointNextSClassifier())
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m.train()


def _step(model, pos, x, y):
    from oracle import cpu_block as CB
    with CB.CpuOps():
        logits, loss = model.get_logits_loss({'pos': pos, 'x': x}, y)
        loss.backward()
    return logits.detach()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from adaptpoint_amd import dp
    dp.init("gloo")
    model = dp.convert_sync_batchnorm(_model())
    assert sum(isinstance(m, dp.SyncBatchNormAllReduce) for m in model.modules()) == 12
    pos, x, y = _data()
    sl = slice(2 * rank, 2 * rank + 2)
    logits = _step(model, pos[sl].contiguous(), x[sl].contiguous(), y[sl])
    grads = [q.grad for q in model.parameters()]
    dp.allreduce_mean_(grads)                               # what DistributedDataParallel leaves in .grad
    torch.save({"logits": logits, "grads": [g.clone() for g in grads],
                "buffers": {k: v.clone() for k, v in model.named_buffers()}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world2_syncbn_classifier_step_equals_one_process_on_the_whole_batch(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    torch.set_num_threads(2)
    ref = _model()
    pos, x, y = _data()
    ref_logits = _step(ref, pos, x, y)
    got = torch.cat([res[0]["logits"], res[1]["logits"]])
    np.testing.assert_allclose(got.numpy(), ref_logits.numpy(), rtol=2e-4, atol=2e-5)
    for (name, q), g0, g1 in zip(ref.named_parameters(), res[0]["grads"], res[1]["grads"]):
        assert torch.equal(g0, g1), name                    # the averaged gradient is the same on every rank
        scale = float(q.grad.abs().max())                   # (some gradients are identically zero up to rounding)
        assert float((g0 - q.grad).abs().max()) <= 2e-3 * scale + 1e-6, (name, float((g0 - q.grad).abs().max()), scale)
    for name, b in ref.named_buffers():                     # running statistics: global, identical on both ranks
        assert torch.equal(res[0]["buffers"][name], res[1]["buffers"][name]), name
        np.testing.assert_allclose(res[0]["buffers"][name].double().numpy(), b.double().numpy(),
                                   rtol=1e-4, atol=1e-6, err_msg=name)


def test_syncbn_allreduce_module_equals_batchnorm_at_world_1():
    from adaptpoint_amd import dp
    torch.manual_seed(0)
    for shape in ((6, 5, 7), (3, 4, 5, 2)):
        bn = (torch.nn.BatchNorm1d if len(shape) == 3 else torch.nn.BatchNorm2d)(shape[1])
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
        import copy
        sb = dp.convert_sync_batchnorm(copy.deepcopy(bn))
        x1 = torch.randn(*shape, requires_grad=True)
        x2 = x1.detach().clone().requires_grad_(True)
        w = torch.randn(*shape)
        (bn(x1) * w).sum().backward()
        (sb(x2) * w).sum().backward()
        assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-6)
        assert torch.allclose(bn.weight.grad, sb.weight.grad, rtol=1e-4, atol=1e-6)
        assert torch.allclose(bn.bias.grad, sb.bias.grad, rtol=1e-4, atol=1e-6)
        assert torch.allclose(bn.running_var, sb.running_var, rtol=1e-5) and torch.allclose(bn.running_mean, sb.running_mean, atol=1e-6)
        sb.eval(); bn.eval()
        assert torch.allclose(bn(x1), sb(x2), rtol=1e-5, atol=1e-6)


def _cls_worker(rank, world, port, out_dir):
    """One `ClassifierStep` (resampler + SyncBatchNorm classifier + clip + AdamW) per rank on its own clouds with
    grad_sync = the flat all-reduce: what train_autoaug.py:275-282 wraps the classifier in."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import golden_inputs as GI
    from oracle import cpu_block as CB
    from adaptpoint_amd import dp
    from adaptpoint_amd.gan import ClassifierStep
    dp.init("gloo")
    model = dp.convert_sync_batchnorm(_model())
    pos = torch.from_numpy(GI.unit_sphere_cloud(4, NPTS, seed=23))       # (N == npoints: the resampler passes through)
    points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1)
    target = torch.tensor([1, 7, 3, 12])
    sl = slice(2 * rank, 2 * rank + 2)
    step = ClassifierStep(model, npoints=NPTS, grad_sync=dp.allreduce_mean_)
    with CB.CpuOps():
        step(points[sl].contiguous(), target[sl])
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, f"cls{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world2_classifier_step_keeps_the_ranks_weights_identical(tmp_path):
    """ClassifierStep(grad_sync=...): after one step on different shards both ranks hold bit-identical weights and
    running statistics (without the hook every rank would step on its local gradients)."""
    mp.spawn(_cls_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f"cls{r}.pt") for r in range(2))
    ref = _model().state_dict()
    moved = 0
    for k in a:
        assert torch.equal(a[k], b[k]), k
        if a[k].dtype.is_floating_point and not torch.equal(a[k], ref[k]):
            moved += 1
    assert moved > 50                                            # the step did update the model
