"""Public engine: request intake, the generate loop, tensor-parallel worker spawn
(interface of nanovllm/engine/llm_engine.py:17-104)."""

import atexit
import os
from dataclasses import fields
from time import perf_counter

import torch.multiprocessing as mp

from byox_vllm_amd.config import Config
from byox_vllm_amd.engine.model_runner import ModelRunner
from byox_vllm_amd.engine.scheduler import Scheduler
from byox_vllm_amd.engine.sequence import Sequence
from byox_vllm_amd.sampling_params import SamplingParams
from byox_vllm_amd.utils.logging import logger


def _load_tokenizer(path):
    try:
        if isinstance(path, str) and os.path.isdir(path):
            from transformers import AutoTokenizer
            return AutoTokenizer.from_pretrained(path, use_fast=True)
    except Exception as e:  # a synthetic-weight directory has no tokenizer files
        logger.warning("no tokenizer under %r (%s): prompts must be token-id lists", path, e)
    return None


class LLMEngine:
    def __init__(self, model, **kwargs):
        names = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in names})
        self.ps, self.events = [], []
        if config.tensor_parallel_size > 1:
            # this engine's own rendezvous port and control-channel name: a second engine on the node (replicas beside
            # a tensor-parallel job, two tests) must not meet this one's ranks
            import socket
            if not config.tp_port:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    config.tp_port = sk.getsockname()[1]
            if not config.shm_name:
                config.shm_name = f"byox_vllm_amd_{os.getpid()}_{config.tp_port}"
        ctx = mp.get_context("spawn")
        for i in range(1, config.tensor_parallel_size):
            event = ctx.Event()
            process = ctx.Process(target=ModelRunner, args=(config, i, event), name=f"worker-{i}")
            process.start()
            self.ps.append(process)
            self.events.append(event)
        self.model_runner = ModelRunner(config, 0, self.events)
        self.tokenizer = _load_tokenizer(config.model)
        config.eos = self.tokenizer.eos_token_id if self.tokenizer is not None else -1
        self.scheduler = Scheduler(config)
        self.config = config
        atexit.register(self._exit)

    def _exit(self):
        if getattr(self, "model_runner", None) is None:
            return
        self.model_runner.call("exit")
        self.model_runner = None
        for p in self.ps:
            p.join()

    def _add_request(self, prompt, sampling_params: SamplingParams):
        if isinstance(prompt, str):
            assert self.tokenizer is not None, "string prompts need a tokenizer in the model directory"
            prompt = self.tokenizer.encode(prompt)
        self.scheduler.add(Sequence(prompt, self.scheduler.block_manager.block_size, sampling_params))

    def _step(self):
        seqs, is_prefill = self.scheduler.schedule()
        if not seqs:
            return [], 0
        token_ids = self.model_runner.call("run", seqs, is_prefill)
        finished = self.scheduler.postprocess(seqs, token_ids)
        outputs = [(seq.seq_id, seq.completion_token_ids) for seq in finished]
        return outputs, (sum(len(seq) for seq in seqs) if is_prefill else -len(seqs))

    def _is_finished(self):
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params, use_tqdm: bool = True):
        pbar = None
        if use_tqdm:
            from tqdm.auto import tqdm
            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for prompt, sp in zip(prompts, sampling_params):
            self._add_request(prompt, sp)
        results = {}
        prefill_tps = decode_tps = 0.0
        # totals of this call, kept for callers that report prefill and decode throughput separately
        stats = self.last_stats = {"prefill_tokens": 0, "prefill_s": 0.0, "prefill_steps": 0,
                                   "decode_tokens": 0, "decode_s": 0.0, "decode_steps": 0}
        while not self._is_finished():
            t0 = perf_counter()
            finished, num_tokens = self._step()
            dt = perf_counter() - t0
            if num_tokens > 0:
                prefill_tps = num_tokens / dt
                stats["prefill_tokens"] += num_tokens
                stats["prefill_s"] += dt
                stats["prefill_steps"] += 1
            elif num_tokens < 0:
                decode_tps = -num_tokens / dt
                stats["decode_tokens"] -= num_tokens
                stats["decode_s"] += dt
                stats["decode_steps"] += 1
            if pbar is not None:
                pbar.set_postfix({"Prefill": f"{int(prefill_tps)}tok/s", "Decode": f"{int(decode_tps)}tok/s"})
            for seq_id, token_ids in finished:
                results[seq_id] = token_ids
                if pbar is not None:
                    pbar.update(1)
        if pbar is not None:
            pbar.close()
        decode = self.tokenizer.decode if self.tokenizer is not None else (lambda ids: "")
        return [{"text": decode(results[i]), "token_ids": results[i]} for i in sorted(results)]
