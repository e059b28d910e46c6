"""bench.py on one GPU, as the round driver runs it: one JSON line carrying every field of the contract."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_is_the_metric_configuration_with_roofline_and_cpu_baseline():
    d = _line(["--steps", "8", "--warmup", "2"])
    assert d["metric"] == "decode tokens/s" and d["unit"] == "tokens/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"].startswith("synthetic")
    cfg = d["config"]
    assert cfg["workload"] == "qwen2-0.5b-w4a16-b128-s1024" and cfg["weight_format"] == "w4a16"
    assert cfg["batch_per_replica"] == 128 and cfg["context"] == 1024 and cfg["graph"] is True and "model" not in cfg
    assert cfg["run_ahead"] is True          # the line says that steps were launched ahead of the driver's calls
    # value = whole-job tokens per second of the timed steps
    assert abs(d["value"] - 128 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3
    assert 0.5 < d["ms_per_step"] < 5.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_us"] / 1e3) / r["achieved"] < 1e-2
    # the PMC record under profiles/ was taken on this library's kernel revision: traffic is a number, just above the
    # algorithmic bytes (no re-reads)
    assert r["traffic"] is not None and r["bytes_per_launch"] <= r["traffic"] < 1.1 * r["bytes_per_launch"], r["traffic_source"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "tokens/s" and c["value"] > 0 and c["cores"] >= 1 and "batch 128" in c["sample"]
    assert 0.0 < d["step_hbm_fraction"] < 1.0 and d["ranks"] == 1 and d["rccl_ranks"] == 0


def test_other_workloads_run_and_say_what_ran():
    d = _line(["--steps", "4", "--warmup", "1", "--workload", "qwen2-0.5b-fp16-b1-s128"])
    assert d["config"]["workload"] == "qwen2-0.5b-fp16-b1-s128" and d["dtype"] == "fp16" and d["value"] > 0
    assert d["cpu_baseline"]["value"] > 0 and "batch 1," in d["cpu_baseline"]["sample"]
