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
