"""GPU test of the N>1 code path of bench.py on ONE GPU: a world_size-1 NCCL (= RCCL) process
group, DistributedDataParallel around the fused block, SyncBatchNorm semantics switched on.
(Real multi-rank RCCL cannot be exercised on a one-GPU box; the sharding / averaging logic is
covered at world_size 2 over gloo in test_dp_gloo_cpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), APN_BENCH_FORCE_DISTRIBUTED="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10",
                          "--warmup", "3", "--no-cpu-baseline"] + extra, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_distributed_default_is_syncbn_world1(dev):
    """default at N>1 = what the reference forces (main.py:27): SyncBatchNorm -- eager, the phased
    statistics all-reduces inside the fused block, one flat gradient all-reduce -- with the
    per-rank-BatchNorm figure beside it"""
    d = _run([], 29731)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert "syncbn" in d["config"]["parallelism"] and "flat-allreduce" in d["config"]["parallelism"]
    # the collectives are captured into the hipGraph (thread-local capture mode)
    assert "collectives-in-graph" in d["config"]["parallelism"] and d["config"]["launch"].startswith("hipGraph replay")
    assert d["value_no_syncbn"] > 0
    e = _run(["--graph-collectives", "off", "--no-secondary"], 29734)       # round 1's structure: eager around them
    assert e["config"]["launch"] == "eager" and "collectives-in-graph" not in e["config"]["parallelism"]
    assert d["value"] > e["value"]


def test_bench_distributed_other_paths_world1(dev):
    """--sync-bn off: hipGraph step + one flat-bucket gradient all-reduce over RCCL; the unfused
    path (--mlp torch-f32) with the all-reduce SyncBatchNorm modules"""
    d = _run(["--sync-bn", "off"], 29732)
    assert "flat-allreduce" in d["config"]["parallelism"] and "syncbn" not in d["config"]["parallelism"]
    assert d["config"]["launch"].startswith("hipGraph replay") and "value_no_syncbn" not in d
    d = _run(["--mlp", "torch-f32", "--steps", "10", "--warmup", "2", "--no-secondary", "--graph-collectives", "off"], 29733)
    assert "syncbn" in d["config"]["parallelism"] and d["value"] > 0


def test_captured_and_eager_collectives_give_the_same_gradients_world1(dev):
    """The two launch structures of the N>1 path -- collectives captured into the hipGraph, or the step run eagerly
    around them -- are the same computation: identical gradients (up to the order of the backward's float atomics)
    at world_size 1, and no memset node in the captured graph (tests/dp_structures_helper.py)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29735")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_structures_helper.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print("captured vs eager collectives:", d)
    assert d["max_rel_diff"] < 1e-4 and d["graph_nodes"].get("memset", 0) == 0 and d["graph_nodes"]["kernel"] >= 7
