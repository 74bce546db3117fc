"""bench.py's launch variants on one GPU: every flag combination the design documents must keep
producing the contract's JSON line (tests/test_gpu_ddp_single.py covers the N>1 code paths)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary"] + extra, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,launch", [
    (["--steps", "8", "--warmup", "3"], "hipGraph replay, 4 step(s) per graph"),
    (["--steps", "7", "--warmup", "2"], "hipGraph replay, 1 step(s) per graph"),
    (["--steps", "8", "--warmup", "2", "--index-overlap", "on"], "hipGraph replay, 4 step(s) per graph"),
    (["--steps", "6", "--warmup", "2", "--pipeline", "off"], "hipGraph replay, 2 step(s) per graph"),
    (["--steps", "8", "--warmup", "2", "--index-batch", "1", "--distribution", "D2", "--seed", "3"], "hipGraph replay, 4 step(s) per graph"),
    (["--steps", "6", "--warmup", "2", "--graph", "off"], "eager"),
    (["--steps", "6", "--warmup", "2", "--mlp", "fused-bf16"], "hipGraph replay, 2 step(s) per graph"),
    (["--steps", "20", "--warmup", "2", "--deterministic"], "hipGraph replay, 20 step(s) per graph"),
    (["--steps", "4", "--warmup", "1", "--mlp", "torch-f32"], None),
])
def test_bench_variants_emit_the_contract_line(dev, extra, launch):
    d = _bench(extra)
    assert KEYS <= set(d) and d["value"] > 0 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["steps"] == int(extra[1]) and d["warmup"] == int(extra[3])
    assert abs(d["ms_per_step"] - 1e3 * 32 / d["value"]) <= 1e-3 * d["ms_per_step"] + 1e-4
    assert d["roofline"]["bound"] in ("hbm", "mfma") and d["roofline"]["frac"] >= 0
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] == "weak"
    st = d["roofline"]["step"]
    assert 0 < st["frac_mfma"] < 1 and 0 < st["frac_hbm"] < 1 and d["roofline"]["index_stream"]["fps_step_ns"] > 0
    # a short timed region is repeated: the median block is reported, min / max beside it
    assert d["timed_blocks"] == 25 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    if d["dtype"] == "bf16" and "--pipeline" not in extra:
        # the dominant kernel is the one with the largest total time per step on the stream that bounds `value`
        assert d["roofline"]["kernel"] == "sa_bwd_main" and d["roofline"]["bound"] == "mfma"
        assert d["roofline"]["avg_launch_us"] == d["roofline"]["kernels"]["sa_bwd_main"]["avg_us"]
    if launch is not None:
        assert d["config"]["launch"] == launch
    # pipelined runs: both index sets, filled BESIDE the MLP kernels all through the timed region, equal the index stage
    # run alone, bit for bit (the LDS-atomic FPS step failed exactly this for ~2 % of the clouds: csrc/fps.hip)
    ver = d["roofline"]["index_stream"].get("verified_after_timed_region")
    assert ver is True or (ver is None and ("--pipeline" in extra or d["dtype"] != "bf16")), ver


def test_index_stage_beside_the_mlp_stream_is_bit_exact_at_full_length(dev):
    """The default run (2000 steps, 100 replays of the two graphs side by side on two streams): the index sets after the
    timed region == the index stage run alone, and the MLP stream's last gradients == the step launched alone up to the
    order of its one float-atomic sum.  The same with the per-wave-record FPS step (APN_FPS_RECORDS=1)."""
    d = _bench([])
    ix = d["roofline"]["index_stream"]
    assert ix["verified_after_timed_region"] is True
    assert ix["mlp_stream_verified_after_timed_region"]["max_gradient_deviation_rel"] < 1e-5
    env = dict(os.environ, APN_FPS_RECORDS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "200",
                          "--warmup", "20"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["roofline"]["index_stream"]["verified_after_timed_region"] is True
