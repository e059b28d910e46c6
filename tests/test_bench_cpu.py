"""bench.py pieces that need no GPU: the CPU-baseline leg (the oracle timed on the host cores) and the workload table."""

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_cpu_baseline_runs_the_workloads_own_batch():
    """A small batch and context keep this to seconds; the fields are the ones the bench line promises."""
    r = bench.cpu_baseline("qwen2-0.5b", "w4a16", torch.bfloat16, batch=4, context=64, budget_s=2.0)
    assert r["kind"] == "port" and r["unit"] == "tokens/s" and r["value"] > 0
    assert r["cores"] == torch.get_num_threads()
    assert "batch 4," in r["sample"] and "context 64" in r["sample"] and "whole model" in r["sample"]


def test_cpu_baseline_samples_layers_of_the_7b_geometry():
    """Qwen2-7B: two decoder layers are timed and scaled to 28 (a small vocabulary keeps this test to seconds); the
    sample text says so."""
    r = bench.cpu_baseline("qwen2-7b", "w4a16", torch.bfloat16, batch=2, context=64, budget_s=1.0, vocab_size=512)
    assert r["value"] > 0 and "2 of 28 decoder layers" in r["sample"] and "scaled by 28/2" in r["sample"]


def test_default_workload_is_the_metric_configuration():
    model, quant, dtype, batch, context = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
    assert (model, quant, dtype, batch, context) == ("qwen2-0.5b", "w4a16", "bf16", 128, 1024)


def _keeper_parent(body, deadline="30"):
    """A stand-in for rank 0: a process that starts the keeper, hands it lines and then runs ``body``."""
    import subprocess
    code = ("import sys, time, os; sys.path.insert(0, %r); import bench; k = bench.LineKeeper(); "
            "k.hold({'value': 1}); k.hold({'value': 2}); " % ROOT) + body
    return subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NV_BENCH_LEG_DEADLINE=deadline),
                          capture_output=True, text=True, timeout=120)


def test_keeper_stays_silent_when_the_job_finishes():
    out = _keeper_parent("print('final', flush=True); k.done()")
    assert out.returncode == 0 and out.stdout.strip() == "final"


def test_keeper_prints_the_last_held_line_when_the_job_dies():
    """os.abort() is what a GPU fault does to the process: no exception, no atexit."""
    out = _keeper_parent("os.abort()")
    assert out.returncode != 0 and out.stdout.strip() == '{"value": 2}'


def test_keeper_ends_a_stalled_job_and_prints_the_line():
    out = _keeper_parent("time.sleep(100)", deadline="2")
    assert out.returncode == -9 and out.stdout.strip() == '{"value": 2}'


def test_keeper_outlives_the_launchers_termination_signal():
    """A launcher that ends a failed job signals whole process groups: the keeper may get SIGTERM while the rank that
    died has not closed the pipe yet (it is still writing its core file).  It stays, and prints once the pipe closes."""
    out = _keeper_parent("import signal; time.sleep(1.0); os.kill(k.proc.pid, signal.SIGTERM); time.sleep(1.0); "
                         "assert k.proc.poll() is None, 'the keeper left on SIGTERM'; os.abort()")
    assert out.returncode != 0 and out.stdout.strip() == '{"value": 2}', (out.stdout, out.stderr[-500:])
