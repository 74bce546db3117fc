"""GPU: BASELINE configs[2]-[4] in their launchable data-parallel form on ONE GPU -- a world-size-1 RCCL process group,
every SyncBatchNorm exchange and gradient all-reduce actually issued through RCCL -- against the same steps without
torch.distributed; and `bench.py --workload ...` taking that path end to end (eager collectives first, then captured
into the hipGraph).  No multi-rank RCCL run exists (one-GPU boxes); world-size-2 semantics are covered over gloo in
tests/test_workloads_gloo_cpu.py through the same `adaptpoint_amd.workloads.build`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    return dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))


def test_data_parallel_training_steps_equal_the_single_process_steps_world1(dev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "workloads_dp_helper.py"), "classifier", "gan"],
                         env=_env(29741), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print("data-parallel vs single-process training steps (world 1, RCCL):", d)
    c = d["classifier"]
    # four fused blocks exchange their own sums (4 all-reduces each), the other BatchNorm modules (group-all stage, head)
    # one per direction, plus the flat gradient all-reduce
    assert c["syncbn"]["fused_blocks_exchanging_their_own_sums"] == 4 and c["syncbn"]["modules_converted"] >= 3
    assert c["collectives"]["all_reduce"] == 4 * 4 + 2 * c["syncbn"]["modules_converted"] + 1, c
    assert c["relative_l2_of_gradient_difference"]["classifier"] < 2e-3, c      # (float-atomic order; the phased launches)
    g = d["gan"]
    assert g["collectives"]["all_reduce"] == 2 and g["syncbn"] is None, g        # G and D: one flat all-reduce each
    assert g["relative_l2_of_gradient_difference"]["generator"] < 2e-3, g
    assert g["relative_l2_of_gradient_difference"]["discriminator"] < 1e-4, g


@pytest.mark.parametrize("workload,port", [("classifier", 29742), ("adaptpoint", 29743)])
def test_bench_workloads_take_the_distributed_path_world1(dev, workload, port):
    env = dict(_env(port), APN_BENCH_FORCE_DISTRIBUTED="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", workload,
                          "--steps", "6", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print(workload, "world 1 over RCCL:", {k: d[k] for k in ("value", "value_eager_collectives", "ms_per_step")}, d["config"]["parallelism"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["value_eager_collectives"] > 0
    assert d["config"]["launch"].startswith("hipGraph replay, collectives captured"), (d["config"]["launch"], out.stderr[-1500:])
    assert d["config"]["graph_nodes"].get("memset", 0) == 0
    assert d["config"]["collectives_per_step"]["all_reduce"] >= (3 if workload == "adaptpoint" else 1)
    assert "syncbn(classifier)" in d["config"]["parallelism"] and d["config"]["fused_fallbacks"] == 0


def test_bench_gan_workload_single_process(dev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "gan", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > d["value_eager"] > 0 and d["config"]["launch"] == "hipGraph replay"
    assert d["config"]["parallelism"] == "dp1" and set(d["losses"]) >= {"g_loss", "d_loss"}
