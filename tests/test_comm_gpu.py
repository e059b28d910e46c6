"""All-reduce over peer-mapped device memory fused with residual add + RMSNorm (nv_allreduce_add_rmsnorm, nv_allreduce).

Oracle: what the reference computes at this point — the sum over ranks of each rank's row-parallel output
(dist.all_reduce, nanovllm/layers/linear.py:145-146; restated as an fp32 sum in rank order with one rounding) followed
by RMSNorm.add_rms_forward (oracle.layers.add_rms_norm).  One MI355X is enough to run the protocol: the ranks are
either streams of one process (peers wired by pointer) or two processes sharing the device (peers wired through
hipIpc handles); what a single GPU cannot show is the traffic crossing xGMI.
"""

import os
import sys

import pytest
import torch

from oracle import layers as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ulp_ok(got, want, ulps=1.0):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs() <= ulps * 2.0 ** -7 * want.abs().clamp_min(1e-3)).all().item()


def oracle(xs, residual, weight, eps):
    total = xs[0].float()
    for x in xs[1:]:
        total = total + x.float()
    return L.add_rms_norm(total.to(xs[0].dtype), residual, weight, eps)


_STREAMS = []


def rank_streams(world):
    """One stream per in-process rank, created once and used once before any rank waits on another: the first launch
    on a fresh stream may have to set its hardware queue up, which must not happen behind a kernel that is waiting."""
    while len(_STREAMS) < world:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(8, device="cuda").add_(1)
        _STREAMS.append(s)
    torch.cuda.synchronize()
    return _STREAMS[:world]


def wait_for_ranks(streams):
    """Block until every in-process rank's stream has drained, and say so if the device-wide synchronize alone would
    not have: each stream is queried first (a query submits whatever the runtime still holds back for that stream: a
    rank whose launch is not submitted while the host blocks on its peer would leave that peer spinning), then
    synchronised on its own."""
    import time
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    pending = [not s.query() for s in streams]
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    if any(pending):
        print(f"torch.cuda.synchronize() returned after {1e3 * (time.perf_counter() - t0):.2f} ms with work pending on the "
              f"rank streams {pending}: waited for the streams one by one")


_WORLDS = {}
_REUSE_WORLDS = os.environ.get("NV_TEST_REUSE_WORLDS", "1") != "0"


def make_world(world, max_rows, hidden):
    """In-process ranks wired by pointer.  A world is built once per geometry and kept until the module's tests are
    over, as an engine keeps its communicator for its lifetime: a rank's call counter carries on from test to test
    (the protocol is indifferent to where it starts), and no buffer is freed while another test's may be mapped at
    the same address."""
    from byox_vllm_amd.layers.comm import XgmiComm
    key = (world, max_rows, hidden)
    if _REUSE_WORLDS and key in _WORLDS:
        return _WORLDS[key]
    comms = [XgmiComm(r, world, max_rows, hidden) for r in range(world)]
    for a in comms:
        for b in comms:
            if a is not b:
                a.set_peer(b.rank, b)
    if _REUSE_WORLDS:
        _WORLDS[key] = comms
    return comms


def release(comms):
    if not _REUSE_WORLDS:
        for c in comms:
            c.close()


@pytest.fixture(scope="module", autouse=True)
def _close_worlds():
    yield
    torch.cuda.synchronize()
    for comms in _WORLDS.values():
        for c in comms:
            c.close()
    _WORLDS.clear()


def _describe_difference(mine, theirs, want, comms, rows):
    """Which rows differ between two ranks' outputs, who owns them, and which of the two matches the oracle."""
    bad = (mine != theirs).any(dim=1).nonzero().flatten().tolist()
    owners = {r: comms[r].owned_rows(rows) for r in range(len(comms))}
    near = lambda t: [bool(ulp_ok(t[i], want[i], ulps=2.0)) for i in bad[:8]]  # noqa: E731
    zero = lambda t: [bool((t[i] == 0).all().item()) for i in bad[:8]]  # noqa: E731
    return (f"memory {comms[0].memory_kind}; {len(bad)} rows differ, first {bad[:8]}; owned rows {owners}; this rank "
            f"matches the oracle there: {near(mine)}, rank 0 does: {near(theirs)}; all-zero rows here {zero(mine)}, "
            f"on rank 0 {zero(theirs)}")


def test_single_rank_equals_add_rmsnorm():
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.comm import XgmiComm
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 896, generator=g).to(torch.bfloat16).cuda()
    res = torch.randn(37, 896, generator=g).to(torch.bfloat16).cuda()
    w = (1 + 0.1 * torch.randn(896, generator=g)).to(torch.bfloat16).cuda()
    comm = XgmiComm(0, 1, 64, 896)
    want_out, want_res = kernels.add_rms_norm(x.clone(), res.clone(), w, 1e-6)
    got_res = res.clone()
    got_out, _ = comm.all_reduce_add_rms_norm(x, got_res, w, 1e-6)
    assert torch.equal(got_res, want_res) and ulp_ok(got_out, want_out)
    assert torch.equal(comm.all_reduce(x), x)
    assert comm.error() == 0
    comm.close()


# Two ranks as two streams of one process.  More ranks need more processes (the test at the end): a process has 4
# hardware queues, kernels of streams that happen to share one run one after the other, and a rank that waits for a
# peer queued behind it can only run into its time limit.
@pytest.mark.parametrize("world,rows,hidden", [(2, 128, 896), (2, 5, 896), (2, 128, 3584), (2, 1, 128), (2, 67, 896)])
@pytest.mark.parametrize("captured", [False, True])
def test_ranks_on_streams_match_the_oracle(world, rows, hidden, captured):
    """`world` ranks as concurrent streams of one process, several calls in a row (both buffer generations), ragged row
    ownership (rows not a multiple of world, ranks that own nothing), eagerly and replayed from per-rank hipGraphs."""
    comms = make_world(world, 128, hidden)
    assert comms[0].memory_kind in ("uncached", "fine-grained", "plain")
    streams = rank_streams(world)
    g = torch.Generator().manual_seed(world * 1000 + rows)
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(torch.bfloat16)
    wd = w.cuda()
    xs_dev = [torch.empty(rows, hidden, dtype=torch.bfloat16, device="cuda") for _ in range(world)]
    res_dev = [torch.empty(rows, hidden, dtype=torch.bfloat16, device="cuda") for _ in range(world)]
    outs = [torch.empty(rows, hidden, dtype=torch.bfloat16, device="cuda") for _ in range(world)]

    def launch_all():
        for r in range(world):
            with torch.cuda.stream(streams[r]):
                comms[r].all_reduce_add_rms_norm(xs_dev[r], res_dev[r], wd, 1e-6, out=outs[r])

    graphs = None
    if captured:
        graphs = []
        for r in range(world):   # warm-up call outside capture would desynchronise the ranks' call counters: capture cold
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[r]):
                comms[r].all_reduce_add_rms_norm(xs_dev[r], res_dev[r], wd, 1e-6, out=outs[r])
            graphs.append(gr)
    for it in range(3):
        xs = [torch.randn(rows, hidden, generator=g).to(torch.bfloat16) for _ in range(world)]
        res = torch.randn(rows, hidden, generator=g).to(torch.bfloat16)
        for r in range(world):
            xs_dev[r].copy_(xs[r])
            res_dev[r].copy_(res)
        torch.cuda.synchronize()
        if captured:
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    graphs[r].replay()
        else:
            launch_all()
        wait_for_ranks(streams)
        assert all(c.error() == 0 for c in comms), "a rank waited for a peer in vain"
        want_out, want_res = oracle(xs, res, w, 1e-6)
        for r in range(world):
            assert torch.equal(outs[r], outs[0]), (f"rank {r} holds other bits than rank 0 (iteration {it}): "
                                                   + _describe_difference(outs[r], outs[0], want_out, comms, rows))
            lo, hi = comms[r].owned_rows(rows)
            assert torch.equal(res_dev[r][lo:hi].cpu(), want_res[lo:hi]), "owned residual rows: one add, one rounding"
            mask = torch.ones(rows, dtype=torch.bool)
            mask[lo:hi] = False
            assert torch.equal(res_dev[r].cpu()[mask], res[mask]), "rows owned by others must stay untouched"
        # the residual rows above pin the reduction bit for bit; the normalised rows go through rsqrt and two
        # roundings, where the device's 1/sqrt and the oracle's rsqrt may differ in the last place
        assert ulp_ok(outs[0], want_out, ulps=2.0), (outs[0].float().cpu() - want_out.float()).abs().max()
    release(comms)


def test_plain_all_reduce_and_split_k_slabs_as_input():
    """nv_allreduce (the vocab-parallel embedding's sum) and a rank contribution handed over as split-K slabs."""
    from byox_vllm_amd import kernels
    from byox_vllm_amd.layers.quant import LinearWeight
    world, rows, hidden, K = 2, 40, 256, 512
    comms = make_world(world, 64, hidden)
    streams = rank_streams(world)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(rows, hidden, generator=g).to(torch.bfloat16) for _ in range(world)]
    outs = [None] * world
    for r in range(world):
        with torch.cuda.stream(streams[r]):
            outs[r] = comms[r].all_reduce(xs[r].cuda())
    wait_for_ranks(streams)
    want = (xs[0].float() + xs[1].float()).to(torch.bfloat16)
    assert torch.equal(outs[0].cpu(), want) and torch.equal(outs[1].cpu(), want)
    # slabs: each rank's GEMM leaves 4 fp32 partial sums; the all-reduce rounds their sum exactly as the GEMM would
    a = [torch.randn(rows, K, generator=g).to(torch.bfloat16).cuda() for _ in range(world)]
    wt = [(torch.randn(hidden, K, generator=g) * 0.05).to(torch.bfloat16).cuda() for _ in range(world)]
    ys = [kernels.gemm(a[r], LinearWeight.dense(wt[r]), split_k=4).clone() for r in range(world)]
    res = torch.randn(rows, hidden, generator=g).to(torch.bfloat16)
    nw = (1 + 0.1 * torch.randn(hidden, generator=g)).to(torch.bfloat16)
    got = [None] * world
    for r in range(world):
        with torch.cuda.stream(streams[r]):   # the workspace is per stream, so the two ranks' slabs do not collide
            parts = kernels.gemm(a[r], LinearWeight.dense(wt[r]), split_k=4, partial_out=True)
            got[r], _ = comms[r].all_reduce_add_rms_norm(parts, res.cuda(), nw.cuda(), 1e-6)
    wait_for_ranks(streams)
    want_out, _ = oracle([y.cpu() for y in ys], res, nw, 1e-6)
    assert torch.equal(got[0], got[1]) and ulp_ok(got[0], want_out)
    assert all(c.error() == 0 for c in comms)
    release(comms)


def test_bad_arguments_are_rejected():
    from byox_vllm_amd.layers.comm import XgmiComm
    comm = XgmiComm(0, 2, 16, 128)
    x = torch.zeros(8, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="not been connected"):
        comm.all_reduce(x)
    comm.close()
    comm = XgmiComm(0, 1, 16, 128)
    with pytest.raises(RuntimeError, match="exceeds"):
        comm.all_reduce(torch.zeros(17, 128, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(RuntimeError, match="hidden"):
        comm.all_reduce(torch.zeros(4, 256, dtype=torch.bfloat16, device="cuda"))
    comm.close()


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from byox_vllm_amd.layers.comm import XgmiComm
from oracle import layers as L
rank, world = int(sys.argv[1]), int(sys.argv[2])
rows = int(sys.argv[4])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[3], rank=rank, world_size=world)
torch.cuda.set_device(0)
hidden = 896
comm = XgmiComm(rank, world, 128, hidden)
comm.connect()
g = torch.Generator().manual_seed(11)
w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(torch.bfloat16)
x_dev = torch.empty(rows, hidden, dtype=torch.bfloat16, device="cuda")
res_dev = torch.empty(rows, hidden, dtype=torch.bfloat16, device="cuda")
out = torch.empty_like(x_dev)
graph = None
for it in range(6):
    xs = [torch.randn(rows, hidden, generator=g).to(torch.bfloat16) for _ in range(world)]
    res = torch.randn(rows, hidden, generator=g).to(torch.bfloat16)
    x_dev.copy_(xs[rank]); res_dev.copy_(res)
    torch.cuda.synchronize(); dist.barrier()
    if it < 3:
        comm.all_reduce_add_rms_norm(x_dev, res_dev, w.cuda(), 1e-6, out=out)
    else:
        if graph is None:
            wd = w.cuda()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                comm.all_reduce_add_rms_norm(x_dev, res_dev, wd, 1e-6, out=out)
        graph.replay()
    torch.cuda.synchronize()
    assert comm.error() == 0, "waited for the peer in vain"
    total = xs[0].float()
    for t in xs[1:]:
        total = total + t.float()
    want_out, want_res = L.add_rms_norm(total.to(torch.bfloat16), res, w, 1e-6)
    lo, hi = comm.owned_rows(rows)
    assert torch.equal(res_dev[lo:hi].cpu(), want_res[lo:hi]), it
    err = (out.float().cpu() - want_out.float()).abs()
    assert (err <= 2.0 ** -6 * want_out.float().abs().clamp_min(1e-3)).all(), (it, err.max())   # 2 ulp, see above
dist.barrier()
comm.close()
print("rank", rank, "ok", comm.memory_kind if False else "")
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,rows", [(2, 96), (3, 5), (4, 97), (4, 2), (8, 128)])
def test_processes_sharing_the_gpu_through_ipc_handles(tmp_path, world, rows):
    """2 to 8 OS processes on one GPU (ragged ownership: 5 rows on 3 ranks, 2 rows on 4 ranks of which two own nothing): buffers exchanged as hipIpc handles (XgmiComm.connect over a gloo group),
    three eager calls and three replays of a captured call, each checked against the oracle in every process."""
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    port = str(29540 + world)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), port, str(rows)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate()[0])
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]


def test_lost_peer_fails_the_step():
    """A peer that never arrives: the bounded wait runs out (about two seconds), the kernel finishes, the error word is
    set, and the runner's per-step check turns it into an exception instead of returning tokens computed from stale
    staging buffers."""
    from byox_vllm_amd.engine.model_runner import ModelRunner
    from byox_vllm_amd.layers.comm import XgmiComm
    comms = [XgmiComm(r, 2, 16, 128) for r in range(2)]      # a world of its own: it ends with the error word set
    comms[0].set_peer(1, comms[1])
    comms[1].set_peer(0, comms[0])
    x = torch.ones(4, 128, dtype=torch.bfloat16, device="cuda")
    comms[0].all_reduce(x)                # rank 1 never calls
    torch.cuda.synchronize()
    assert comms[0].error() != 0
    runner = ModelRunner.__new__(ModelRunner)
    runner.comm, runner.rank = comms[0], 0
    with pytest.raises(RuntimeError, match="timed out waiting for a peer"):
        runner._check_comm()
    runner.comm = None
    runner._check_comm()                  # no communicator: nothing to check
    for c in comms:
        c.close()
