"""Where a decode step of LLM.generate goes on the reference's own workload (128 requests, U[100,128] tokens in and out,
Qwen3-0.6B bf16): scheduler.schedule, ModelRunner.call("run"), scheduler.postprocess, the loop's own bookkeeping."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from byox_vllm_amd import LLM, SamplingParams
from byox_vllm_amd.config import qwen2_config

hf = qwen2_config("qwen3-0.6b", dtype=torch.bfloat16)
random.seed(0)
prompts = [[random.randint(0, 10000) for _ in range(random.randint(100, 128))] for _ in range(128)]
params = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=random.randint(100, 128)) for _ in range(128)]
llm = LLM("synthetic", hf_config=hf, max_model_len=256, max_num_seqs=128, max_num_batched_tokens=16384,
          kvcache_block_size=256, num_kvcache_blocks=2 * 128 + 8)
llm.generate([[1, 2, 3]], SamplingParams(max_tokens=4), use_tqdm=False)
T = {"schedule": 0.0, "run": 0.0, "postprocess": 0.0}
N = {"decode_steps": 0}
pc = time.perf_counter
sched, runner = llm.scheduler, llm.model_runner
r_schedule, r_post, r_call = sched.schedule, sched.postprocess, runner.call
def schedule():
    t = pc(); out = r_schedule(); T["schedule"] += pc() - t; return out
def postprocess(seqs, toks):
    t = pc(); out = r_post(seqs, toks); T["postprocess"] += pc() - t; return out
def call(name, *a):
    t = pc(); out = r_call(name, *a); dt = pc() - t
    if name == "run" and not a[1]:
        T["run"] += dt; N["decode_steps"] += 1
    return out
sched.schedule, sched.postprocess, runner.call = schedule, postprocess, call
for rep in range(2):
    for k in T: T[k] = 0.0
    N["decode_steps"] = 0
    torch.cuda.synchronize(); t0 = pc()
    llm.generate(prompts, params, use_tqdm=False)
    torch.cuda.synchronize(); wall = pc() - t0
n = N["decode_steps"]
print(f"generate {wall * 1e3:.1f} ms, {n} decode steps: per decode step run {T['run'] / n * 1e6:.0f} us | schedule (all steps) "
      f"{T['schedule'] / (n + 1) * 1e6:.0f} us | postprocess {T['postprocess'] / (n + 1) * 1e6:.0f} us | "
      f"rest of the loop {(wall - T['run'] - T['schedule'] - T['postprocess']) / (n + 1) * 1e6:.0f} us (incl. the prefill step)")
llm._exit()
