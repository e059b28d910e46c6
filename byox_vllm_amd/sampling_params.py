"""Per-request sampling options (mirrors nanovllm/sampling_params.py:4-11).

Extension over the reference: ``temperature == 0`` selects greedy argmax.  The
reference rejects it (its sampler always draws Gumbel noise); negative values
are still an error here.
"""

from dataclasses import dataclass


@dataclass
class SamplingParams:
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False

    def __post_init__(self):
        assert self.temperature >= 0.0, "temperature must be >= 0 (0 selects greedy argmax)"
        assert self.temperature == 0.0 or self.temperature > 1e-10, "temperature too small; use 0 for greedy"
