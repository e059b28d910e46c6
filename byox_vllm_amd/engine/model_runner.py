"""Per-rank executor of the decode hot path (interface of nanovllm/engine/model_runner.py:17-297).

``run(seqs, is_prefill)`` = prepare integer metadata on the host -> one forward
pass on the HIP kernels -> sample.  Decode steps replay a hipGraph captured per
padded batch size; prefill runs eagerly.  With tensor parallelism rank 0 ships
the PREPARED integer arrays of the step (not pickled Sequence objects) to the
other ranks through shared memory, so the control message stays a few KB at
batch 512.
"""

import pickle
from multiprocessing.shared_memory import SharedMemory

import os

import numpy as np
import torch
import torch.distributed as dist

from byox_vllm_amd import kernels
from byox_vllm_amd.engine.sequence import Sequence
from byox_vllm_amd.layers.sampler import Sampler
from byox_vllm_amd.models.qwen3 import Qwen3ForCausalLM
from byox_vllm_amd.utils.context import get_context, reset_context, set_context
from byox_vllm_amd.utils.loader import init_random_weights, load_model
from byox_vllm_amd.utils.logging import logger

SHM_NAME = "byox_vllm_amd"      # defaults of a job that did not choose its own (Config.shm_name / Config.tp_port)
TP_PORT = 2333
SHM_BYTES = 1 << 22
MAX_GRAPH_BS = 512


def graph_batch_sizes(max_bs: int) -> list[int]:
    """Padded batch sizes a decode graph is captured for (model_runner.py:269)."""
    return [b for b in (1, 2, 4, 8) if b <= max_bs] + list(range(16, max_bs + 1, 16))


class ModelRunner:
    def __init__(self, config, rank: int, event):
        self.config = config
        hf_config = config.hf_config
        self.block_size = config.kvcache_block_size
        self.use_xgmi = config.tensor_parallel_size > 1 and (
            config.xgmi_allreduce or os.environ.get("NV_XGMI_ALLREDUCE", "0") == "1")
        # gloo collectives cannot be captured; with the library's own all-reduce nothing of torch.distributed is
        # left inside the forward pass, so the graph path works under any host-side backend
        self.enforce_eager = config.enforce_eager or (
            config.tensor_parallel_size > 1 and os.environ.get("NV_DIST_BACKEND", "nccl") != "nccl"
            and not self.use_xgmi)
        self.world_size = config.tensor_parallel_size
        self.device_bookkeeping = os.environ.get("NV_DEVICE_BOOKKEEPING", "1") != "0"
        # tensor-parallel decode steps travel as the packed staging buffer (run_packed); 0: as seven arrays (run_step)
        self.tp_packed_steps = os.environ.get("NV_TP_PACKED_STEPS", "1") != "0"
        # launch the next decode step before the caller asks for it, whenever no other next step is possible
        self.run_ahead = self.device_bookkeeping and os.environ.get("NV_RUN_AHEAD", "1") != "0"
        self.spin_wait = os.environ.get("NV_SPIN_WAIT", "1") != "0"
        self.rank = rank
        self.event = event
        if not torch.cuda.is_available():
            raise RuntimeError("ModelRunner needs an MI355X: byox_vllm_amd has no CPU compute path")

        # NV_TP_SINGLE_RANK_RCCL=1: a one-GPU machine runs the tensor-parallel code path for real: RCCL process group
        # of one rank, every collective issued (and captured into the decode graphs), LM head and sampler outside
        self.single_rank_rccl = self.world_size == 1 and os.environ.get("NV_TP_SINGLE_RANK_RCCL", "0") == "1"
        self._own_pg = False
        if self.single_rank_rccl and not dist.is_initialized():
            dist.init_process_group("nccl", f"tcp://127.0.0.1:{(getattr(config, 'tp_port', 0) or TP_PORT) + 1}",
                                    world_size=1, rank=0)
            self._own_pg = True
        if self.world_size > 1 and not dist.is_initialized():
            # NV_DIST_BACKEND=gloo: rehearsal of the tensor-parallel runtime with the ranks sharing GPUs (RCCL wants one
            # device per rank); gloo collectives cannot be captured, so that mode runs eagerly
            # the engine that spawned the ranks chose a free port and a shared-memory name of its own
            # (LLMEngine.__init__), so several engines can live on one node; the reference's fixed
            # tcp://localhost:2333 and "nanovllm" (model_runner.py:27,42) admit one
            dist.init_process_group(os.environ.get("NV_DIST_BACKEND", "nccl"),
                                    f"tcp://127.0.0.1:{getattr(config, 'tp_port', 0) or TP_PORT}",
                                    world_size=self.world_size, rank=rank)
            self._own_pg = True
        from byox_vllm_amd.layers.linear import set_comm, set_tp
        set_tp(rank if self.world_size > 1 else 0, self.world_size, getattr(config, "tp_group", None),
               single_rank_collectives=self.single_rank_rccl)
        set_comm(None)
        self.comm = None
        local = int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        # ranks that share a GPU (a rehearsal): no vendor-library products beside the all-reduce's waiting workgroups
        self._shares_gpu = (self.world_size > max(torch.cuda.device_count(), 1)
                            and os.environ.get("NV_GEMM_LT_SHARED_GPU", "0") != "1")     # 1: measurement aid
        if self._shares_gpu:
            kernels.block_many_rows(True)
        self.device = torch.device("cuda", local)
        default_dtype = torch.get_default_dtype()
        default_device = torch.get_default_device()
        torch.set_default_dtype(hf_config.dtype)
        torch.set_default_device(self.device)
        try:
            if self.use_xgmi:
                from byox_vllm_amd.layers.comm import XgmiComm
                self.comm = XgmiComm(rank, self.world_size, config.max_num_batched_tokens, hf_config.hidden_size)
                self.comm.connect(getattr(config, "tp_group", None))
                set_comm(self.comm)
                logger.info("tensor-parallel all-reduce over peer-mapped %s device memory", self.comm.memory_kind)
            self.model = Qwen3ForCausalLM(hf_config)
            self._load_weights()
            self.model.finalize(config.quantization)
            self.decode_result_policy = self._pick_decode_result_policy()
            self.prefill_result_policy = int(os.environ.get("NV_RESULT_POLICY_PREFILL", "0"))
            self.sampler = Sampler(seed=config.seed)
            self._warmup_model()
            self._allocate_kv_cache()
            if not self.enforce_eager:
                self._capture_graphs()
        except BaseException:
            self._release_after_failed_init()
            raise
        finally:
            # Back to the caller's default.  Where that is the CPU the device context is REMOVED rather than set to
            # "cpu" (the reference sets "cpu", model_runner.py:47): an installed context routes every later torch call
            # - each tensor.data_ptr() and .stride() of the launch wrappers included - through a Python
            # __torch_function__ hook, about 3,400 calls and several milliseconds per eager prefill step.
            torch.set_default_device(None if default_device.type == "cpu" else default_device)
            torch.set_default_dtype(default_dtype)

        if self.world_size > 1 and event is not None:
            shm_name = getattr(config, "shm_name", "") or SHM_NAME
            if rank == 0:
                self.shm = SharedMemory(name=shm_name, create=True, size=SHM_BYTES)
                dist.barrier()
            else:
                dist.barrier()
                self.shm = SharedMemory(name=shm_name)
                self._loop()

    def _pick_decode_result_policy(self) -> int:
        """Write-through result stores (nv_set_result_policy) for DECODE steps: what a launch hands to the next one
        (norm rows, residual rows, SiLU*up rows, split-K slabs, q/k/v and KV-cache rows) goes to the memory side while
        the kernel still runs, which is where a consumer on another XCD finds it.  Measured per decode step:
        Qwen2-0.5B int4 -2.4 %, Qwen2-7B int4 -2.4 %, 0.5B int8 and bf16 -0.7 %, Qwen3-0.6B bf16 -0.9 %.  Prefill steps
        keep plain stores (their results are large: 8 x 1024 tokens -1 %, 16 x 2048 -5 % with the policy).
        Decode steps outside RESULT_WT_ROWS keep plain stores too: with few rows the results are a few KB and waiting
        for their write-through costs more than the L2 write-back it saves (Qwen2-0.5B int4, policy on against off:
        batch 1 +3.4 %, 4 +1.5 %, 16 +2.1 %, 32 +0.2 %, 64 +0.6 % / int8 -0.7 %, 128 -1.9 %), with 512 rows they are
        megabytes (Qwen2-7B: +0.5 %).
        NV_RESULT_POLICY=0/1 overrides the choice for decode, NV_RESULT_POLICY_PREFILL=1 switches it on for prefill."""
        return int(os.environ.get("NV_RESULT_POLICY", "1"))

    RESULT_WT_ROWS = (64, 256)

    def _decode_policy_for(self, rows: int) -> int:
        lo, hi = self.RESULT_WT_ROWS
        if os.environ.get("NV_RESULT_POLICY_ROWS"):        # test aid: "lo,hi" (worker processes read it too)
            lo, hi = (int(v) for v in os.environ["NV_RESULT_POLICY_ROWS"].split(","))
        return self.decode_result_policy if lo <= rows <= hi else 0

    def _release_after_failed_init(self) -> None:
        """A constructor that raised half-way (an uncapturable collective, out of memory) gives back what it had
        taken - decode graphs, KV cache, model, the peer-memory buffer and its mappings, its own process group - so
        that the caller can build another runner in this process (bench.py retries eagerly, then falls back to
        replicas).  No collective is issued here: the peers may be anywhere."""
        from byox_vllm_amd.layers.linear import set_comm
        if getattr(self, "_shares_gpu", False):
            self._shares_gpu = False
            kernels.block_many_rows(False)
        self.graphs, self.graph_pool, self.graph_vars = {}, None, None
        self.kv_cache = None
        for module in (self.model.modules() if getattr(self, "model", None) is not None else ()):
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = module.v_cache = torch.tensor([])
        self.model = None
        set_comm(None)
        if self.comm is not None:
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            self.comm.close()
            self.comm = None
        if self._own_pg and dist.is_initialized():
            dist.destroy_process_group()
            self._own_pg = False
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ weights
    def _load_weights(self):
        """Checkpoint -> parameters (nanovllm model_runner.py:33).  Synthetic weights are drawn only when the caller
        asked for them (model == "synthetic"); a directory without safetensors files is an error, as in the reference."""
        import glob
        path = self.config.model
        if self.config.state_dict is not None:
            from byox_vllm_amd.utils.loader import load_state_dict_packed
            load_state_dict_packed(self.model, self.config.state_dict, self.config.hf_config)
        elif path == "synthetic":
            logger.info("drawing synthetic weights (seed %d)", self.config.seed)
            init_random_weights(self.model, self.config.seed)
        else:
            if not (isinstance(path, str) and os.path.isdir(path) and glob.glob(os.path.join(path, "*.safetensors"))):
                raise FileNotFoundError(f"no *.safetensors under {path!r} (pass model='synthetic' for random weights)")
            load_model(self.model, path, self.config.quantization)

    # ------------------------------------------------------------------ RPC (tensor parallel)
    def call(self, method_name, *args):
        if method_name == "run" and self.world_size > 1:
            seqs, is_prefill = args
            if not is_prefill and not self.enforce_eager and 0 < len(seqs) <= self.max_graph_bs and self.tp_packed_steps:
                # a decode step for the captured graphs: the control message is the packed buffer the single-rank
                # path stages from (one array, built with the same caches), not seven tensors
                host = self._host_stage_np
                n, cols = self._pack_decode(seqs, host[2:])
                host[0], host[1] = len(seqs), cols
                if self.rank == 0 and getattr(self, "shm", None) is not None:
                    self._write_shm("run_packed", host[:2 + n])
                return self.run_packed(None)
            step = self._prepare_step(seqs, is_prefill)
            if self.rank == 0 and getattr(self, "shm", None) is not None:
                self._write_shm("run_step", step)
            return self.run_step(step)
        if self.world_size > 1 and self.rank == 0 and getattr(self, "shm", None) is not None:
            self._write_shm(method_name, *args)
        return getattr(self, method_name)(*args)

    def _loop(self):
        while True:
            method_name, args = self._read_shm()
            getattr(self, method_name)(*args)
            if method_name == "exit":
                break

    def _read_shm(self):
        assert self.world_size > 1 and self.rank > 0
        self.event.wait()
        n = int.from_bytes(self.shm.buf[0:4], "little")
        method_name, *args = pickle.loads(self.shm.buf[4: n + 4])
        self.event.clear()
        return method_name, args

    def _write_shm(self, method_name, *args):
        assert self.world_size > 1 and self.rank == 0
        data = pickle.dumps([method_name, *args])
        n = len(data)
        assert n + 4 <= self.shm.size, f"control message of {n} bytes exceeds the {self.shm.size}-byte channel"
        self.shm.buf[0:4] = n.to_bytes(4, "little")
        self.shm.buf[4: n + 4] = data
        for event in self.event:
            event.set()

    def exit(self):
        if getattr(self, "_shares_gpu", False):
            self._shares_gpu = False
            kernels.block_many_rows(False)
        if self.world_size > 1 and getattr(self, "shm", None) is not None:
            self.shm.close()
            dist.barrier()
            if self.rank == 0:
                self.shm.unlink()
        torch.cuda.synchronize()              # a step launched ahead may still be running: no graph is dropped under it
        if not self.enforce_eager:
            self.graphs, self.graph_pool = {}, None
        if self.comm is not None:
            from byox_vllm_amd.layers.linear import set_comm
            if self.world_size > 1 and dist.is_initialized():
                dist.barrier()        # no peer may still be reading this rank's buffer
            set_comm(None)
            self.comm.close()
            self.comm = None
        if self._own_pg:
            dist.destroy_process_group()

    # ------------------------------------------------------------------ the step
    def run(self, seqs: list[Sequence], is_prefill: bool) -> list[int] | None:
        if (not is_prefill and not self.enforce_eager and self.world_size == 1
                and 0 < len(seqs) <= self.max_graph_bs and self.graph_has_head):
            return self._run_decode_graph(seqs)
        self._drop_step_ahead()
        input_ids, positions = self._prepare_prefill(seqs) if is_prefill else self._prepare_decode(seqs)
        temperatures = self._prepare_sample(seqs) if self.rank == 0 else None
        logits = self._run_model(input_ids, positions, is_prefill)
        token_ids = self.sampler(logits, temperatures).tolist() if self.rank == 0 else None
        reset_context()
        return token_ids

    @torch.inference_mode()
    def _run_decode_graph(self, seqs: list[Sequence]) -> list[int]:
        """Single-rank decode fast path: the step's integers (the values _prepare_decode, _prepare_block_tables and
        _prepare_sample would produce) go to the device as ONE packed buffer; the replayed graph runs staging ->
        forward -> LM head -> sampler; the only host sync is reading the token ids back.  When the step that must
        follow is already known (_next_step_is_certain) it is launched before this one's tokens are returned, and the
        next call only checks that it asks for exactly that step."""
        bs = len(seqs)
        g = self.graph_vars
        host = self._host_stage_np
        padded = next(x for x in self.graph_bs if x >= bs)
        ntok = self._continues_previous_step(seqs, padded)
        cache = getattr(self, "_pack_cache", None)
        continued = ntok is not None
        if self._step_ahead and not continued:
            self._wait_for_step()            # before the pinned buffer it reads is rewritten
            self._drop_step_ahead()
        if self._step_ahead:                 # the step asked for is the one already running
            self._step_ahead = False
            self._ahead_backoff = 0
        else:
            if continued:                    # the device advances its own copy of the step: only the header travels
                host[0], host[1] = bs, cache["key"][1] | kernels.STAGE_ADVANCE
            else:
                n, cols = self._pack_decode(seqs, host[2:])
                host[0], host[1] = bs, cols  # header read on the device by the graph's own staging node
                cache = self._pack_cache
                ntok = cache["ntok"].tolist()
            host[1] |= self._step_back
            self._step_back = 0
            g["padded_bs"] = padded
            self._replay(padded)
        self._wait_for_step()
        tokens = self._host_tokens_np[:bs].tolist()
        cache["padded"], cache["tokens"], cache["next_ntok"] = padded, tokens, [x + 1 for x in ntok]
        if self.run_ahead and self._next_step_is_certain(seqs, cache, ntok, tokens):
            if self._ahead_hold == 0:
                host[0], host[1] = bs, cache["key"][1] | kernels.STAGE_ADVANCE
                self._replay(padded)
                self._step_ahead = True
            else:
                self._ahead_hold -= 1
        return tokens

    def _next_step_is_certain(self, seqs: list[Sequence], cache: dict, ntok: list[int], tokens: list[int]) -> bool:
        """May the decode step after this one be launched before the caller has asked for it?  Only if the caller's
        own rules leave it no other next step for these sequences: none of them ends on the token just sampled
        (its completion stays short of max_tokens; the token is not the end-of-sequence id unless the sequence ignores
        it: Scheduler.postprocess, nanovllm/engine/scheduler.py:65-71) and none needs a new block for the next token
        (BlockManager.may_append allocates when len % block_size == 1, block_manager.py:96-99 - the only point at which
        a running sequence can be preempted, scheduler.py:45-52).  The count of such steps ahead is taken once per
        full packing and counted down.  A caller that still presents another step (new prompts arrived, sequences
        aborted) is served correctly - the launched step is dropped (_drop_step_ahead) - but has waited for it.
        The block rule is what correctness needs (a step launched without the block it would write to has no slot);
        the two ending rules only spare a launch that is certain to be dropped: the row such a step writes for an
        ended sequence lies in a freed block that carries no hash, and whoever gets the block writes before reading."""
        sure = cache.get("sure")
        if sure is None:
            n = np.asarray(ntok, dtype=np.int64)
            if "limit" not in cache:
                cache["limit"] = np.array([s.num_prompt_tokens + s.max_tokens for s in seqs], dtype=np.int64)
                cache["eos_ends"] = any(not s.ignore_eos for s in seqs)
            # steps from here in which nobody enters a new block / nobody reaches max_tokens
            sure = int(min(((-n) % self.block_size).min(), (cache["limit"] - n - 1).min()))
        cache["sure"] = sure - 1
        if sure < 1:
            return False
        if cache["eos_ends"]:
            eos = self.config.eos
            if eos in tokens and any(t == eos and not s.ignore_eos for s, t in zip(seqs, tokens)):
                return False
        return True

    def _drop_step_ahead(self) -> None:
        """The caller asked for something else than the step launched ahead of it.  That step's results are not used:
        its tokens are never returned, the next decode step is packed on the host in full (ids, positions, slots,
        context lengths, tables), and the K/V rows it wrote are the rows of the token every one of its sequences has
        just been given - the step that does decode that token writes the same values there again (rows of a sequence
        that ended cannot be among them: see _next_step_is_certain).  Its sampler draw is given back
        (STAGE_STEP_BACK) so that later draws are those of a run that never launched it.  Work queued after it on the
        stream runs after it; only the pinned staging buffer needs the host to wait, which the graph path does."""
        if not self._step_ahead:
            return
        self._step_ahead = False
        self._step_back = kernels.STAGE_STEP_BACK
        # a caller that keeps asking for other steps than the certain one (it repeats a step, say) soon stops paying
        # for launches it does not use: 1, 2, 4 ... 64 certain steps are let pass before the next is launched ahead
        self._ahead_backoff = min(64, max(1, 2 * self._ahead_backoff))
        self._ahead_hold = self._ahead_backoff
        cache = getattr(self, "_pack_cache", None)
        if cache is not None:
            cache["tokens"] = None           # the device's copy of the step is one ahead: no continuing from it

    def _wait_for_step(self) -> None:
        """Block until the replayed graph (which ends by delivering the tokens to pinned host memory) has finished.
        The step is synchronous by contract - the scheduler needs this step's tokens before it can form the next -
        so the GPU idles from the graph's end until the host has noticed and launched again.  Polling the stream
        (hipStreamQuery, about a microsecond per call) notices within microseconds; a blocking synchronize sleeps on an
        interrupt and wakes up tens of microseconds later.  NV_SPIN_WAIT=0 restores the blocking wait."""
        stream = torch.cuda.current_stream()
        if self.spin_wait:
            query = stream.query
            while not query():
                pass
        else:
            stream.synchronize()

    def _continues_previous_step(self, seqs: list[Sequence], padded: int):
        """The sequences' lengths when this decode step is the previous graph step continued: the same Sequence objects
        in the same order, every one longer by exactly the token this runner returned for it, no block table touched
        (Sequence.table_epoch, total block count) and the same captured graph; None otherwise.  The device then holds
        everything the step needs (nv_stage_decode_packed_advance) and the host builds no arrays at all."""
        cache = getattr(self, "_pack_cache", None)
        if (not self.device_bookkeeping or cache is None or cache.get("tokens") is None or cache["padded"] != padded
                or cache["seqs"] != seqs or cache["key"] is None or cache["key"][2] != Sequence.table_epoch):
            return None
        total = 0
        for seq in seqs:
            total += len(seq.block_table)
        ntok = [seq.num_tokens for seq in seqs]
        if total != cache["key"][0] or ntok != cache["next_ntok"] or [seq.last_token for seq in seqs] != cache["tokens"]:
            return None
        return ntok

    def _pack_decode(self, seqs: list[Sequence], host: np.ndarray) -> tuple[int, int]:
        """Fill ``host`` (int32) with one decode step's integers, laid out as nv_stage_decode_packed reads them:
        [last tokens | positions | slots | context lens | temperatures (f32 bits) | block tables bs x cols, -1 padded].
        Same values as _prepare_decode / _prepare_block_tables / _prepare_sample (nanovllm model_runner.py:152-158,
        206-235).  Returns (words used, cols).  The padded tables are rebuilt only when a table changed.  Invariant
        the cache rests on: for an unchanged list of Sequence objects, every change of a block table bumps
        Sequence.table_epoch (BlockManager.allocate / deallocate / may_append do) or alters the total block count
        (a driver that appends block ids by hand)."""
        bs, block = len(seqs), self.block_size
        tables_in = [seq.block_table for seq in seqs]
        lens = list(map(len, tables_in))
        cols, total = max(lens), sum(lens)
        n = 5 * bs + bs * cols
        head = host[:4 * bs].reshape(4, bs)
        head[0] = [seq.last_token for seq in seqs]
        ntok = head[3]
        ntok[:] = [seq.num_tokens for seq in seqs]
        np.subtract(ntok, 1, out=head[1])
        ntok_copy = ntok.copy()
        # Steady-state decode presents the same Sequence objects step after step: what cannot change for a given
        # object (its temperature) and what rarely changes (the padded tables, the last block of every row) is kept
        # from the previous step.  List equality short-cuts on identity, so the checks cost well under a microsecond.
        cached = getattr(self, "_pack_cache", None)
        if cached is None or cached["seqs"] != seqs:
            cached = self._pack_cache = {"seqs": list(seqs), "key": None, "tokens": None,
                                         "temps": np.array([seq.temperature for seq in seqs], dtype=np.float32)}
        cached["ntok"] = ntok_copy
        cached["sure"] = None                # steps ahead known to be plain continuations: counted again from here
        host[4 * bs:5 * bs].view(np.float32)[:] = cached["temps"]
        tables = host[5 * bs:n].reshape(bs, cols)
        key = (total, cols, Sequence.table_epoch)
        if cached["key"] == key:
            tables[:] = cached["tables"]
        else:
            if total == bs * cols:
                tables[:] = tables_in
            else:
                tables.fill(-1)
                for i, t in enumerate(tables_in):
                    tables[i, : len(t)] = t
            cached["key"], cached["tables"] = key, tables.copy()
            cached["last_block_row"] = np.array([t[-1] for t in tables_in], dtype=np.int32) * block
        np.add(cached["last_block_row"], head[1] % block, out=head[2])
        return n, cols

    def _prepare_step(self, seqs, is_prefill) -> dict:
        """Host integers of one step as numpy arrays: the whole tensor-parallel control message."""
        input_ids, positions = self._prepare_prefill(seqs, device="cpu") if is_prefill else self._prepare_decode(seqs, device="cpu")
        ctx = get_context()
        step = {"is_prefill": is_prefill, "input_ids": input_ids.numpy(), "positions": positions.numpy(),
                "max_seqlen_q": ctx.max_seqlen_q, "max_seqlen_k": ctx.max_seqlen_k,
                "temperatures": np.array([s.temperature for s in seqs], dtype=np.float32)}
        for key in ("cu_seqlens_q", "cu_seqlens_k", "slot_mapping", "context_lens", "block_tables"):
            t = getattr(ctx, key)
            step[key] = None if t is None else t.numpy()
        reset_context()
        return step

    def run_step(self, step: dict):
        dev = lambda a: None if a is None else self._to_device(torch.from_numpy(a))  # noqa: E731
        set_context(step["is_prefill"], dev(step["cu_seqlens_q"]), dev(step["cu_seqlens_k"]), step["max_seqlen_q"],
                    step["max_seqlen_k"], dev(step["slot_mapping"]), dev(step["context_lens"]), dev(step["block_tables"]))
        logits = self._run_model(dev(step["input_ids"]), dev(step["positions"]), step["is_prefill"])
        token_ids = None
        if self.rank == 0:
            token_ids = self.sampler(logits, dev(step["temperatures"])).tolist()
        reset_context()
        self._check_comm()
        return token_ids

    @torch.inference_mode()
    def run_packed(self, packed):
        """One decode step of a tensor-parallel job from the packed control message (layout of _pack_decode behind the
        two header words {rows, table width}): one host-to-device copy and one staging launch fill the captured
        graph's inputs on every rank - the same integers run_step receives as seven arrays (reference:
        model_runner.py:206-235 on every rank after read_shm) - then the graph, the LM-head shard, the gather and, on
        rank 0, the sampler.  ``packed`` is None on the rank that packed the step into its own staging buffer."""
        host = self._host_stage_np
        if packed is not None:
            host[:packed.size] = packed
        bs, cols = int(host[0]), int(host[1])
        words = 2 + bs * (5 + cols)
        g = self.graph_vars
        padded = next(x for x in self.graph_bs if x >= bs)
        g["padded_bs"] = padded
        self._dev_stage[:words].copy_(self._host_stage[:words], non_blocking=True)
        kernels.stage_decode_packed(self._dev_stage, -1, 0, g)
        self._replay(padded)
        logits = self.model.compute_logits(g["outputs"][:bs])
        token_ids = None
        if self.rank == 0:
            token_ids = self.sampler(logits, g["temperatures"][:bs]).tolist()
        else:
            torch.cuda.current_stream().synchronize()     # the staging buffer is rewritten by the next message
        self._check_comm()
        return token_ids

    def _check_comm(self) -> None:
        """Once per step, after the step's own synchronisation: a bounded wait of the peer-memory all-reduce that ran
        out (a peer lost or late by seconds) leaves an error word in this rank's buffer and the kernels carry on with
        whatever the staging buffers held.  Such a step must fail, not return tokens."""
        if self.comm is not None:
            code = self.comm.error()
            if code:
                raise RuntimeError(f"rank {self.rank}: the peer-memory all-reduce timed out waiting for a peer "
                                   f"(nv_comm_error = {code}: phase {code - 1}); this step's hidden states are invalid")

    # ------------------------------------------------------------------ host -> device metadata
    def _to_device(self, t: torch.Tensor, device=None) -> torch.Tensor:
        device = device if device is not None else getattr(self, "device", None)
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        if torch.device(device).type == "cpu":
            return t
        return t.pin_memory().to(device, non_blocking=True)

    def _prepare_block_tables(self, seqs: list[Sequence], device=None):
        width = max(len(seq.block_table) for seq in seqs)
        table = np.full((len(seqs), width), -1, dtype=np.int32)
        for i, seq in enumerate(seqs):
            table[i, : len(seq.block_table)] = seq.block_table
        return self._to_device(torch.from_numpy(table), device)

    def _prepare_prefill(self, seqs: list[Sequence], device=None):
        bs = self.block_size
        input_ids, positions, slot_mapping = [], [], []
        cu_q, cu_k = [0], [0]
        max_q = max_k = 0
        for seq in seqs:
            n, cached = len(seq), seq.num_cached_tokens
            input_ids.extend(seq[cached:])
            positions.extend(range(cached, n))
            cu_q.append(cu_q[-1] + n - cached)
            cu_k.append(cu_k[-1] + n)
            max_q, max_k = max(max_q, n - cached), max(max_k, n)
            if not seq.block_table:  # warmup sequences own no blocks
                continue
            last = seq.num_blocks - 1
            for i in range(seq.num_cached_blocks, seq.num_blocks):
                start = seq.block_table[i] * bs
                slot_mapping.extend(range(start, start + (bs if i != last else seq.last_block_num_tokens)))
        block_tables = self._prepare_block_tables(seqs, device) if cu_k[-1] > cu_q[-1] else None  # prefix hit
        to = lambda data, dt: self._to_device(torch.tensor(data, dtype=dt, device="cpu"), device)  # noqa: E731
        set_context(True, to(cu_q, torch.int32), to(cu_k, torch.int32), max_q, max_k,
                    to(slot_mapping, torch.int32), None, block_tables)
        return to(input_ids, torch.int64), to(positions, torch.int64)

    def _prepare_decode(self, seqs: list[Sequence], device=None):
        bs = self.block_size
        input_ids = [seq.last_token for seq in seqs]
        positions = [len(seq) - 1 for seq in seqs]
        context_lens = [len(seq) for seq in seqs]
        slot_mapping = [seq.block_table[-1] * bs + seq.last_block_num_tokens - 1 for seq in seqs]
        to = lambda data, dt: self._to_device(torch.tensor(data, dtype=dt, device="cpu"), device)  # noqa: E731
        slot_t, ctx_t = to(slot_mapping, torch.int32), to(context_lens, torch.int32)
        block_tables = self._prepare_block_tables(seqs) if device is None else self._prepare_block_tables(seqs, device)
        set_context(False, slot_mapping=slot_t, context_lens=ctx_t, block_tables=block_tables)
        return to(input_ids, torch.int64), to(positions, torch.int64)

    def _prepare_sample(self, seqs: list[Sequence]):
        return self._to_device(torch.tensor([seq.temperature for seq in seqs], dtype=torch.float32, device="cpu"))

    # ------------------------------------------------------------------ forward
    @torch.inference_mode()
    def _run_model(self, input_ids: torch.Tensor, positions: torch.Tensor, is_prefill: bool):
        bs = input_ids.size(0)
        # single-rank graphs end in the sampler and stage their own inputs (_run_decode_graph): a caller that wants
        # logits from here gets the eager forward
        if is_prefill or self.enforce_eager or bs > self.max_graph_bs or self.graph_has_head:
            if not is_prefill:
                self._set_decode_splits(bs)
                with kernels.result_policy(self._decode_policy_for(bs)):
                    return self.model.compute_logits(self.model(input_ids, positions))
            with kernels.result_policy(self.prefill_result_policy):
                return self.model.compute_logits(self.model(input_ids, positions))
        context = get_context()
        padded = next(x for x in self.graph_bs if x >= bs)
        g = self.graph_vars
        g["padded_bs"] = padded
        kernels.stage_decode_inputs(input_ids, positions, context.slot_mapping, context.context_lens,
                                    context.block_tables.contiguous(), g)
        self._replay(padded)
        return self.model.compute_logits(g["outputs"][:bs])

    def _replay(self, padded_bs: int) -> None:
        self.graphs[padded_bs].replay()

    def _set_decode_splits(self, bs: int) -> None:
        hkv = self.model.model.layers[0].self_attn.num_kv_heads
        splits = kernels.pick_decode_splits(bs, hkv, self.config.max_model_len)
        for layer in self.model.model.layers:
            layer.self_attn.attn.decode_splits = splits

    def _warm_vendor_library(self) -> None:
        """The first hipBLASLt call of a process loads the library and its kernels (0.2 s with a warm page cache,
        longer from a cold disk).  Pay that here, outside any step, and let the ranks of a tensor-parallel job meet
        again afterwards: inside the warm-up prefill it would make ranks arrive at their first all-reduce seconds
        apart, and the peer-memory all-reduce waits for a peer for two seconds only."""
        rows = kernels.MANY_ROWS
        if rows and kernels.use_many_rows_path(rows, 128, 128):
            from byox_vllm_amd.layers.quant import LinearWeight
            dtype = self.config.hf_config.dtype
            x = torch.zeros(rows, 128, dtype=dtype, device=self.device)
            kernels.gemm_many_rows(x, LinearWeight.dense(torch.zeros(128, 128, dtype=dtype, device=self.device)))
            torch.cuda.synchronize()
        if self.world_size > 1 and dist.is_initialized():
            dist.barrier()

    def _warmup_model(self):
        self._warm_vendor_library()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        max_tokens, max_len = self.config.max_num_batched_tokens, self.config.max_model_len
        num_seqs = max(1, min(max_tokens // max_len, self.config.max_num_seqs))
        seqs = [Sequence([0] * max_len, self.block_size) for _ in range(num_seqs)]
        self.run(seqs, True)
        torch.cuda.empty_cache()

    def _allocate_kv_cache(self):
        config, hf = self.config, self.config.hf_config
        num_kv_heads = self.model.model.layers[0].self_attn.num_kv_heads   # local; replicated when ranks > KV heads
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        block_bytes = 2 * hf.num_hidden_layers * self.block_size * num_kv_heads * head_dim * hf.dtype.itemsize
        if config.num_kvcache_blocks <= 0:
            free, total = torch.cuda.mem_get_info()
            stats = torch.cuda.memory_stats()
            peak, current = stats["allocated_bytes.all.peak"], stats["allocated_bytes.all.current"]
            config.num_kvcache_blocks = int(total * config.gpu_memory_utilization - (total - free) - peak + current) // block_bytes
        assert config.num_kvcache_blocks > 0
        # [2, L, NB, block, Hkv, D]: layer l reads kv_cache[0, l] / kv_cache[1, l], as the reference binds them
        self.kv_cache = torch.zeros(2, hf.num_hidden_layers, config.num_kvcache_blocks, self.block_size,
                                    num_kv_heads, head_dim, dtype=hf.dtype, device=self.device)
        layer_id = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = self.kv_cache[0, layer_id]
                module.v_cache = self.kv_cache[1, layer_id]
                layer_id += 1

    @torch.inference_mode()
    def _capture_graphs(self):
        """Capture one hipGraph of the decode step per padded batch size, largest first, sharing one pool
        (nanovllm model_runner.py:256-297)."""
        config, hf = self.config, self.config.hf_config
        max_bs = min(config.max_num_seqs, MAX_GRAPH_BS)
        max_blocks = (config.max_model_len + self.block_size - 1) // self.block_size
        dev = self.device
        g = dict(
            input_ids=torch.zeros(max_bs, dtype=torch.int64, device=dev),
            positions=torch.zeros(max_bs, dtype=torch.int64, device=dev),
            slot_mapping=torch.full((max_bs,), -1, dtype=torch.int32, device=dev),
            context_lens=torch.zeros(max_bs, dtype=torch.int32, device=dev),
            block_tables=torch.zeros(max_bs, max_blocks, dtype=torch.int32, device=dev),
            outputs=torch.zeros(max_bs, hf.hidden_size, dtype=hf.dtype, device=dev),
            temperatures=torch.ones(max_bs, dtype=torch.float32, device=dev),
            tokens=torch.zeros(max_bs, dtype=torch.int64, device=dev),
            step=torch.zeros(1, dtype=torch.int64, device=dev),     # bumped by the staging kernel on every replay
            padded_bs=max_bs,
        )
        stage_words = 2 + max_bs * (5 + max_blocks)
        self._host_stage = torch.zeros(stage_words, dtype=torch.int32, device="cpu").pin_memory()
        self._host_stage_np = self._host_stage.numpy()
        self._dev_stage = torch.zeros(stage_words, dtype=torch.int32, device=dev)
        self._host_tokens = torch.zeros(max_bs, dtype=torch.int64, device="cpu").pin_memory()
        self._host_tokens_np = self._host_tokens.numpy()
        # Single rank: the graph also stages its inputs and ends in LM head + sampler.  With tensor parallelism (or
        # Config.graph_head = False) it holds the forward only; logits gather and sampling follow outside, as in the
        # reference (model_runner.py:253-254).
        self.graph_has_head = self.world_size == 1 and self.config.graph_head and not self.single_rank_rccl
        self.fused_head_sampler = os.environ.get("NV_FUSED_HEAD_SAMPLER", "1") != "0"   # 0: measurement aid
        self.graph_bs = sorted(config.graph_batch_sizes) if config.graph_batch_sizes else graph_batch_sizes(max_bs)
        max_bs = self.graph_bs[-1]
        self.max_graph_bs = max_bs
        self.graphs, self.graph_pool = {}, None

        def forward_rows(bs, stage_words=0):
            """Decode forward (+ LM head + sampler on a single rank) for rows [0, bs) of the static buffers."""
            if stage_words:
                self._dev_stage[:stage_words].copy_(self._host_stage[:stage_words], non_blocking=True)
                kernels.stage_decode_packed(self._dev_stage, -1, 0, g, prev_tokens=g["tokens"],
                                            block_size=self.block_size)
            self._set_decode_splits(bs)
            set_context(False, slot_mapping=g["slot_mapping"][:bs], context_lens=g["context_lens"][:bs],
                        block_tables=g["block_tables"][:bs])
            hidden = self.model(g["input_ids"][:bs], g["positions"][:bs])
            if not self.graph_has_head:
                g["outputs"][:bs] = hidden        # the caller computes logits outside the graph (TP gather)
            else:                                 # head and sampler inside the graph: no copy of the hidden states
                if self.fused_head_sampler:       # the LM head's epilogue feeds the sampler: one pass over the logits less
                    self.model.sample_tokens(hidden, g["temperatures"][:bs], self.config.seed, seed_offset=g["step"],
                                             out=g["tokens"][:bs])
                else:
                    logits = self.model.compute_logits(hidden)
                    kernels.sample(logits, g["temperatures"][:bs], seed=self.config.seed, seed_offset=g["step"],
                                   out=g["tokens"][:bs])
                # the graph ends by delivering the tokens to pinned host memory: the host then only waits for the stream
                self._host_tokens[:bs].copy_(g["tokens"][:bs], non_blocking=True)

        for bs in reversed(self.graph_bs):
            # a single-rank step carries its staging inside the graph: the host then only packs the pinned buffer and
            # launches one graph
            words = 2 + bs * (5 + max_blocks) if self.graph_has_head else 0
            g["padded_bs"] = bs
            self._host_stage_np[:2] = 0          # warm-up and capture run on an empty step (all rows padding)
            with kernels.result_policy(self._decode_policy_for(bs)):
                forward_rows(bs, words)          # warm-up: sizes every workspace before anything is captured
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, self.graph_pool):
                    forward_rows(bs, words)
            if self.graph_pool is None:
                self.graph_pool = graph.pool()
            torch.cuda.synchronize()
            self.graphs[bs] = graph
            reset_context()
        self.graph_vars = g

    max_graph_bs = 0
    graph_has_head = False
    run_ahead = False
    _step_ahead = False
    _step_back = 0
    _ahead_backoff = 0
    _ahead_hold = 0
    decode_result_policy = 0
    prefill_result_policy = 0
