"""Qwen2 / Qwen3 decoder on the HIP kernels (wiring of nanovllm/models/qwen3.py:14-208).

Same modules, parameter names and tensor-parallel split as the reference, so
its checkpoints and packed-module mapping load unchanged.  The forward pass is
arranged around the fused kernels:

  add_rmsnorm -> qkv GEMM (+bias) -> [q/k head norm] RoPE + KV-cache store (one
  launch, in place on the qkv buffer) -> paged attention -> o GEMM [-> all-reduce]
  -> add_rmsnorm -> gate_up GEMM -> SiLU*up -> down GEMM [-> all-reduce]

``attention_bias`` selects the Qwen2 block (bias on QKV, no head norms); without
it the Qwen3 block (q_norm / k_norm) is built (qwen3.py:66-68,80-82).
"""

import os

import torch
from torch import nn

from byox_vllm_amd import kernels
from byox_vllm_amd.layers.activation import SiluAndMul
from byox_vllm_amd.layers.attention import Attention
from byox_vllm_amd.layers.embed_head import ParallelLMHead, VocabParallelEmbedding
from byox_vllm_amd.layers.layernorm import RMSNorm
from byox_vllm_amd.layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from byox_vllm_amd.layers.rotary_embedding import get_rope
from byox_vllm_amd.utils.context import get_context


class Qwen3Attention(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, num_kv_heads: int, max_position: int = 4096 * 32,
                 head_dim: int | None = None, rms_norm_eps: float = 1e-06, qkv_bias: bool = False,
                 rope_theta: float = 10000, rope_scaling: tuple | None = None) -> None:
        super().__init__()
        self.total_num_heads = num_heads
        self.total_num_kv_heads = num_kv_heads
        self.head_dim = head_dim or hidden_size // self.total_num_heads
        self.scaling = self.head_dim ** -0.5
        self.qkv_bias = qkv_bias
        self.rms_norm_eps = rms_norm_eps

        # The rank's heads come from the QKV layer's HeadLayout: the reference's split where the KV heads divide
        # by the rank count (qwen3.py:30-34), KV-head replication with zero-padded query slots otherwise.
        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, self.total_num_heads,
                                          self.total_num_kv_heads, bias=qkv_bias)
        layout = self.qkv_proj.layout
        self.num_heads, self.num_kv_heads = layout.num_heads, layout.num_kv_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.o_proj = RowParallelLinear(self.total_num_heads * self.head_dim, hidden_size, bias=False,
                                        input_segments=layout.segments(layout.q_heads), local_input_size=self.q_size)
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim, max_position=max_position,
                                   base=rope_theta, rope_scaling=rope_scaling)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, self.num_kv_heads)
        self.fuse_qkv_rope = os.environ.get("NV_FUSE_QKV_ROPE", "1") != "0"
        self.force_fuse_qkv_rope = os.environ.get("NV_FUSE_QKV_ROPE") == "2"    # measurement aid: skip the planner
        if not self.qkv_bias:
            self.q_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)
            self.k_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor) -> torch.Tensor:
        context = get_context()
        attn = self.attn
        has_cache = attn.k_cache.numel() > 0
        rope = dict(slot_mapping=context.slot_mapping if has_cache else None,
                    k_cache=attn.k_cache if has_cache else None, v_cache=attn.v_cache if has_cache else None,
                    q_norm_w=None if self.qkv_bias else self.q_norm.weight.data,
                    k_norm_w=None if self.qkv_bias else self.k_norm.weight.data, eps=self.rms_norm_eps)
        proj = self.qkv_proj
        if proj.packed is None:
            proj.finalize(None)
        rows = hidden_states.size(0)
        # Decode rows run inside a captured graph, where the library's estimate of GPU time decides.  A prefill step
        # is launched eagerly and is bound by the host up to about 2 k tokens (5.4 ms for 24 layers whatever the
        # size): there one launch less per layer wins (0.5B: 5.85 -> 5.49 ms at 2 k tokens; equal at 4 k).
        many_rows = kernels.use_many_rows_path(rows, proj.packed.in_features, proj.packed.out_features)
        fused = self.fuse_qkv_rope and (
            self.force_fuse_qkv_rope or (context.is_prefill and 16 <= rows <= 4096 and not many_rows) or
            kernels.lib().nv_gemm_rope_kv_store_is_faster(rows, proj.packed.in_features, proj.packed.fmt,
                                                          self.num_heads, self.num_kv_heads, self.head_dim))
        if fused:
            # projection, bias, head norms, RoPE and the KV-cache store in one launch (one workgroup per head)
            qkv = kernels.gemm_rope_kv_store(hidden_states, proj.packed, None if proj.bias is None else proj.bias.data,
                                             positions, self.rotary_emb.cos_sin_cache, self.num_heads,
                                             self.num_kv_heads, self.head_dim, **rope)
        else:
            qkv = self.qkv_proj(hidden_states, partial_out=True)   # split-K slabs + bias, summed inside the RoPE kernel
            qkv = kernels.rope_kv_store(qkv, positions, self.rotary_emb.cos_sin_cache, self.num_heads,
                                        self.num_kv_heads, self.head_dim, **rope)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        q = q.view(-1, self.num_heads, self.head_dim)
        k = k.view(-1, self.num_kv_heads, self.head_dim)
        v = v.view(-1, self.num_kv_heads, self.head_dim)
        o = attn.forward_fused(q, k, v)
        return self.o_proj(o.view(o.size(0), -1), partial_out=True)


class Qwen3MLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str) -> None:
        super().__init__()
        # shards of the intermediate dimension are whole 128-column K-blocks of down_proj (uneven shares are padded
        # with zero rows / columns: silu(0) * 0 = 0 meets zero weights)
        align = 128 if intermediate_size % 128 == 0 else 1
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=False, align=align)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=False, align=align)
        assert hidden_act == "silu"
        self.act_fn = SiluAndMul()

    def forward(self, x):
        if self.gate_up_proj.fused_silu:
            act = self.gate_up_proj.forward_silu_mul(x)
        else:
            act = self.act_fn(self.gate_up_proj(x))
        return self.down_proj(act, partial_out=True)


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.self_attn = Qwen3Attention(
            hidden_size=config.hidden_size,
            num_heads=config.num_attention_heads,
            num_kv_heads=config.num_key_value_heads,
            max_position=config.max_position_embeddings,
            rms_norm_eps=config.rms_norm_eps,
            qkv_bias=getattr(config, "attention_bias", True),
            head_dim=getattr(config, "head_dim", None),
            rope_theta=getattr(config, "rope_theta", 1000000),
            rope_scaling=getattr(config, "rope_scaling", None),
        )
        self.mlp = Qwen3MLP(config.hidden_size, config.intermediate_size, config.hidden_act)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            hidden_states, residual = self.input_layernorm(hidden_states), hidden_states
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class Qwen3Model(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Qwen3DecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class Qwen3ForCausalLM(nn.Module):
    packed_modules_mapping = {
        "q_proj": ("qkv_proj", "q"),
        "k_proj": ("qkv_proj", "k"),
        "v_proj": ("qkv_proj", "v"),
        "gate_proj": ("gate_up_proj", 0),
        "up_proj": ("gate_up_proj", 1),
    }

    def __init__(self, config) -> None:
        super().__init__()
        self.model = Qwen3Model(config)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if config.tie_word_embeddings:
            self.lm_head.weight.data = self.model.embed_tokens.weight.data

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.lm_head(hidden_states)

    def sample_tokens(self, hidden_states: torch.Tensor, temperatures: torch.Tensor, seed: int, seed_offset=None,
                      out=None) -> torch.Tensor:
        """compute_logits followed by the sampler, single rank (the step's tail, model_runner.py:253-254 and :62 of the
        reference): the tokens ``Sampler`` would draw from ``compute_logits(hidden_states)`` with this seed."""
        tokens, _ = self.lm_head.forward_sample(hidden_states, temperatures, seed, seed_offset=seed_offset, out=out)
        return tokens

    def finalize(self, quantization: str | None = None) -> None:
        """Convert every layer linear to its kernel-side weight (the LM head and embedding stay dense)."""
        from byox_vllm_amd.layers.linear import LinearBase
        for module in self.modules():
            if isinstance(module, MergedColumnParallelLinear):
                module.finalize(quantization, fuse_silu=True)
            elif isinstance(module, LinearBase):
                module.finalize(quantization)
        if self.lm_head.weight is not None and \
                self.lm_head.weight.data_ptr() != self.model.embed_tokens.weight.data_ptr():
            self.lm_head.release_row_major()
