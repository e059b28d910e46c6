"""bench.py with two ranks on one GPU: the multi-rank launch path (torch.distributed.run, barrier, max over ranks,
rank-0 JSON line) in both modes.  RCCL wants one device per rank, so the ranks talk over gloo (NV_BENCH_BACKEND)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["dp", "default", "xgmi"])
def test_bench_two_ranks(mode):
    """`default` passes no --parallelism, as the round driver does: with more than one rank that is the tensor-parallel
    split (one model over both ranks, all-reduce on the data path) plus the replicas leg."""
    env = dict(os.environ, NV_BENCH_BACKEND="gloo", NV_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533" if mode == "dp" else "29534",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--workload", "qwen2-0.5b-fp16-b1-s128"] + (["--parallelism", "dp"] if mode == "dp" else []) + (
               ["--allreduce", "xgmi"] if mode == "xgmi" else [])
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]          # exactly one line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0
    assert d["scaling"] == ("weak" if mode == "dp" else "strong")
    assert d["config"]["parallelism"] == ("dp2" if mode == "dp" else "tp2")
    assert d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert d["graph_captured"] == (mode != "default")      # gloo collectives cannot be captured, the library's own can
    assert ("replicas_tok_s" in d) == (mode != "dp") and d.get("replicas_tok_s", 1) > 0
    assert "cpu_baseline" not in d                      # rank 0 times the CPU baseline at N = 1 only


def test_bench_falls_back_to_replicas_when_the_tensor_parallel_leg_fails():
    """A tensor-parallel step that raises on every rank must not cost the measurement: the line then reports one
    replica per rank (weak scaling, dp2) and carries the error text."""
    env = dict(os.environ, NV_BENCH_BACKEND="gloo", NV_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", NV_BENCH_FAIL_TP="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29535",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--workload", "qwen2-0.5b-fp16-b1-s128"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert "NV_BENCH_FAIL_TP" in d["tp_error"] and "replicas_tok_s" not in d
