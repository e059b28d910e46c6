from byox_vllm_amd.engine.llm_engine import LLMEngine


class LLM(LLMEngine):
    pass
