from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Optional


@dataclass
class ForwardContext:
    attn_metadata: Any
    virtual_engine: int = 0
    num_tokens: Optional[int] = None
    skip_cuda_graphs: bool = False
    no_compile_layers: dict = None        # layer name -> module (vLLM: compilation_config.static_forward_context)


_ctx: Optional[ForwardContext] = None
history = []          # (num_tokens, skip_cuda_graphs) of every context entered (tests look at it)


def get_forward_context() -> ForwardContext:
    assert _ctx is not None, "no forward context is set"
    return _ctx


@contextmanager
def set_forward_context(attn_metadata, vllm_config, virtual_engine: int = 0, num_tokens=None, num_tokens_across_dp=None,
                        skip_cuda_graphs: bool = False):
    global _ctx
    prev = _ctx
    _ctx = ForwardContext(attn_metadata, virtual_engine, num_tokens, skip_cuda_graphs,
                          vllm_config.compilation_config.static_forward_context)
    history.append((num_tokens, skip_cuda_graphs))
    try:
        yield
    finally:
        _ctx = prev
