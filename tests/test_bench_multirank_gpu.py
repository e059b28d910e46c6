"""bench.py with two ranks on one GPU, started the way a user (or a driver that does not know about launchers) would:
plain `python bench.py --gpus 2`.  The script starts torch.distributed.run itself as a child process; the ranks then
go through barrier, max over ranks and the rank-0 JSON line in both parallel modes.  RCCL wants one device per rank,
so the ranks talk over gloo (NV_BENCH_BACKEND) and the line has to say so."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env_extra=None, launcher=False, expect_failure=False):
    env = dict(os.environ, NV_BENCH_BACKEND="gloo", NV_DIST_BACKEND="gloo", **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher:      # the round driver's form: an external launcher, one process per rank
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
    if "--workload" not in extra:
        cmd += ["--workload", "qwen2-0.5b-fp16-b1-s128"]
    cmd += extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert (out.returncode != 0) == expect_failure, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:] + "\n--- stderr ---\n" + out.stderr[-6000:]   # exactly one line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["dp", "default", "xgmi", "launcher"])
def test_bench_two_ranks(mode):
    """`default` passes no --parallelism, as the round driver does: with more than one rank that is the tensor-parallel
    split (one model over both ranks, all-reduce on the data path), then the same job with the other all-reduce, then
    the replicas leg.  `launcher`: the same under an external torch.distributed.run, the driver's documented form."""
    extra = {"dp": ["--parallelism", "dp"], "default": [], "xgmi": ["--allreduce", "xgmi"], "launcher": ["--single-leg"]}[mode]
    d = _bench(extra, launcher=mode == "launcher")
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["steps"] == 4 and d["value"] > 0
    assert d["scaling"] == ("weak" if mode == "dp" else "strong")
    assert d["config"]["parallelism"] == ("dp2" if mode == "dp" else "tp2")
    # two ranks, none of them talking through RCCL: the label must not claim otherwise
    assert d["rccl_ranks"] == 0 and d["backend"] == "gloo"
    # gloo collectives cannot be captured, the library's own all-reduce can; the config block says what ran
    captured = mode in ("dp", "xgmi")
    assert d["graph_captured"] == captured and d["config"]["graph"] == captured
    assert ("replicas_tok_s" in d) == (mode != "dp") and d.get("replicas_tok_s", 1) > 0
    assert d["tp_error"] is None
    if mode == "default":
        assert d["allreduce"].startswith("torch.distributed")
        assert d["xgmi_allreduce_tok_s"] > 0 and d["xgmi_allreduce_graph_captured"] is True
    if mode == "xgmi":
        assert d["allreduce"].startswith("nv_allreduce_add_rmsnorm")
        assert d["rccl_allreduce_tok_s"] > 0 and d["rccl_allreduce_graph_captured"] is False
    if mode == "launcher":
        assert "xgmi_allreduce_tok_s" not in d
    assert "cpu_baseline" not in d                      # rank 0 times the CPU baseline at N = 1 only
    assert d["roofline"]["traffic_source"]


def test_bench_falls_back_to_replicas_when_the_tensor_parallel_leg_fails():
    """A tensor-parallel step that raises must not cost the measurement: the ranks agree on the failure, the line
    reports one replica per rank (weak scaling, dp2) and carries the error text."""
    d = _bench([], env_extra={"NV_BENCH_FAIL_TP": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert "NV_BENCH_FAIL_TP" in d["tp_error"] and "replicas_tok_s" not in d


@pytest.mark.parametrize("hook", ["NV_BENCH_ABORT_TP", "NV_BENCH_STALL_TP"])
def test_bench_line_survives_a_tensor_parallel_leg_that_takes_the_job_down(hook):
    """A GPU fault aborts the process from a runtime thread and a lost peer can stall a collective for good: neither
    is an exception.  The replicas leg has been timed by then and the keeper process started by rank 0 prints its
    line — exactly one line, marked in `tp_error` — while the job itself ends with a failure status."""
    d = _bench([], env_extra={hook: "1", "NV_BENCH_LEG_DEADLINE": "20"}, expect_failure=True)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert "keeper process" in d["tp_error"] and d["roofline"]["frac"] > 0


def test_bench_two_ranks_on_the_metric_workload():
    """The configuration a multi-GPU scaling run takes by default, at its real size: Qwen2-0.5B int4, 128 rows, one
    model over two ranks, the decode graph captured with the fused all-reduce inside and the hand-over results written
    through (the result-store policy applies at 128 rows).  Two ranks share this GPU, so the process group is gloo."""
    d = _bench(["--workload", "qwen2-0.5b-w4a16-b128-s1024", "--allreduce", "xgmi", "--single-leg", "--no-replicas"])
    assert d["config"]["workload"] == "qwen2-0.5b-w4a16-b128-s1024" and d["config"]["parallelism"] == "tp2"
    assert d["graph_captured"] is True and d["allreduce"].startswith("nv_allreduce_add_rmsnorm")
    assert d["value"] > 0 and d["tp_error"] is None and d["scaling"] == "strong"
    # each rank holds one of the two KV heads: a launch reads K and V rows of 64 elements for every cached token
    tokens = d["roofline"]["bytes_per_launch"] / (2 * 64 * 2)
    assert 128 * 1024 <= tokens <= 128 * 1100 and d["roofline"]["frac"] > 0
