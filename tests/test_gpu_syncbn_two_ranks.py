"""GPU: the fused blocks' SyncBatchNorm exchange with TWO REAL RANKS on the one GPU of the box.  Two fresh child processes,
both on cuda:0, form a gloo process group (device tensors staged through the host -- RCCL refuses two ranks on one
device) and run one classifier training iteration eagerly through `adaptpoint_amd.workloads.build("classifier",
fused=True)`: every stage's set-abstraction block on the fused kernels, their four statistics exchanges per block
issued between the launch phases, the remaining BatchNorm modules converted, the gradients averaged by one flat
all-reduce.  Property (the one tests/test_syncbn_gloo_cpu.py checks for the unfused CPU path): rank-local shards of 16
clouds each give the logits, every parameter gradient, the running statistics and the updated weights of ONE process on
the 32 clouds with plain BatchNorm (reference: examples/classification/main.py:27 forces SyncBatchNorm at world size > 1;
train_autoaug.py:275-282 wraps the classifier in DistributedDataParallel).  This is the only multi-rank evidence obtainable
without a multi-GPU node; RCCL with N > 1 ranks stays unmeasured."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "syncbn_two_ranks_helper.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_equal_one_process_on_the_whole_batch(dev, tmp_path):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    whole = subprocess.run([sys.executable, HELPER, "whole", str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert whole.returncode == 0, whole.stderr[-3000:]
    procs = [subprocess.Popen([sys.executable, HELPER, "rank", str(r), "2", port, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:                      # (exactly the processes started here)
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    ref = torch.load(tmp_path / "whole.pt")
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]

    # the code path: four fused blocks exchanging their own sums (4 all-reduces each), the other BatchNorm modules one per
    # direction, one flat gradient all-reduce -- and no fused block fell back to the composed operators
    for r in res:
        assert r["syncbn"]["fused_blocks_exchanging_their_own_sums"] == 4 and r["syncbn"]["modules_converted"] >= 3, r["syncbn"]
        assert r["collectives"]["all_reduce"] == 4 * 4 + 2 * r["syncbn"]["modules_converted"] + 1, r["collectives"]
        assert not r["fused_fallbacks"], r["fused_fallbacks"]
    assert not ref["fused_fallbacks"] and ref["syncbn"] is None and not ref["collectives"]

    # the property: shards of 16 + 16 clouds == one process on the 32
    logits = torch.cat([res[0]["logits"], res[1]["logits"]])
    e_logits = _rel(logits, ref["logits"])
    e_loss = abs(0.5 * (res[0]["loss"] + res[1]["loss"]) - ref["loss"]) / abs(ref["loss"])
    # (a per-channel shift in front of a training-mode BatchNorm has an analytically zero gradient -- the last stage's
    # BatchNorm shift ahead of the batch-normalised head: both sides hold rounding noise there; tests/classifier_b8_checks.py
    # `norm_floor` treats the same parameter the same way.  Such a tensor must stay small, not agree in relative terms)
    norms = [float(b.double().norm()) for b in ref["grads"]]
    floor = 1e-3 * sorted(norms)[len(norms) // 2]
    zeros = [i for i, n in enumerate(norms) if n < floor]
    assert len(zeros) <= 1 and all(float(res[0]["grads"][i].double().norm()) < 10 * floor for i in zeros), zeros
    e_grads = [_rel(a, b) for i, (a, b) in enumerate(zip(res[0]["grads"], ref["grads"])) if i not in zeros]
    worst = max(e_grads)
    # both ranks hold the same averaged gradients, buffers and updated weights
    for a, b in zip(res[0]["grads"], res[1]["grads"]):
        assert torch.equal(a, b)
    e_buf = {}
    for k, v in ref["buffers"].items():
        if v.dtype.is_floating_point:
            e_buf[k] = _rel(res[0]["buffers"][k], v)
            assert torch.equal(res[0]["buffers"][k], res[1]["buffers"][k]), k
        else:
            assert torch.equal(res[0]["buffers"][k], v), k           # num_batches_tracked
    e_par = max(_rel(a, b) for a, b in zip(res[0]["params"], ref["params"]))
    print("two ranks on one GPU vs one process: logits %.2e, loss %.2e, gradients median %.2e worst %.2e, "
          "running statistics worst %.2e, updated weights worst %.2e"
          % (e_logits, e_loss, sorted(e_grads)[len(e_grads) // 2], worst, max(e_buf.values()), e_par))
    # float32 sums in another order (per-rank partial sums of the BatchNorm statistics, float-atomic scatter order): not
    # bit-equal.  Measured (round 5): logits 1.9e-5, loss 2.4e-7, running statistics 7.9e-7, gradients median 2.2e-3, worst
    # 3.3e-3 (BatchNorm-1 scale / shift of the fused blocks: the handful of ReLU gates that a 1e-7 change of the batch
    # statistics switches, tests/test_gpu_fused.py::test_discontinuities_explain_the_gradient_residual); a block normalising
    # with RANK-LOCAL statistics is off by > 1e-1 everywhere.  Bars: 2 x measured.
    assert e_logits < 1e-4 and e_loss < 1e-5
    assert worst < 7e-3, sorted(zip(e_grads, range(len(e_grads))))[-5:]
    assert max(e_buf.values()) < 1e-5
    # (the updated weights are printed, not held: AdamW's first step is lr * g / (|g| + eps) -- a gradient entry near zero
    # moves its weight by up to 2 lr whichever way its last bits fall; G18 holds the optimizer step where it is meaningful)
    assert e_par < 2e-2
