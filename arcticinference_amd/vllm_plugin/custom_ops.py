"""The patched Attention.forward as ONE torch custom op, `arctic_inference::attention`.

vLLM V1 traces the model with Dynamo (fullgraph) under @support_torch_compile at its default compilation level,
Attention.forward included, up to vLLM's own `unified_attention` custom op; the reference keeps its patched forward to
traceable torch ops (view / cat / all_to_all_single, ulysses.py:493-517).  This build's forward is not traceable — it
calls libarctic_hip.so through ctypes and decides on the host (forward context, step geometry, cache layout) whether
the HIP route serves the step — so all of it sits behind an op boundary, the way vLLM hides its own backend:

    torch.ops.arctic_inference.attention(query, key, value, layer_name) -> [num_tokens, heads * head_size]

The layer is looked up by name in the forward context INSIDE the op (vLLM's own mechanism, ForwardContext.
no_compile_layers), a fake implementation gives Dynamo / Inductor the output's shape, and the op's name is added to
CompilationConfig.splitting_ops (CompilationConfigPatch below), so vLLM's piecewise graphs are cut at it exactly as
they are cut at `vllm.unified_attention`: the op always runs eagerly between the captured pieces, with the step's real
metadata.  Under full-graph capture the op itself is captured; the route then uses the device-geometry entry point
(vllm_plugin/ulysses.py::_arctic_verify).
"""
from __future__ import annotations

from typing import Optional

import torch

OP_NAMESPACE = "arctic_inference"
OP_QUALNAME = "arctic_inference::attention"
# vLLM 0.9.2's split_graph cuts a traced graph at the nodes with `str(node.target) in splitting_ops` (recalled).  What that
# string is depends on HOW the op was called: through the packet, torch.ops.arctic_inference.attention(...), Dynamo records
# the OpOverloadPacket, which prints as "arctic_inference.attention" (the spelling of vLLM's own "vllm.unified_attention");
# through the CustomOpDef object returned by torch.library.custom_op it records the OpOverload,
# "arctic_inference.attention.default" (checked under Dynamo in this image's torch).  The patched forward calls
# the PACKET (call_attention below) and the list carries BOTH spellings, so a cut cannot be missed either way — an op that
# is not cut would be captured inside a piecewise graph piece with attn_metadata = None and replay as a no-op.
SPLITTING_OP = "arctic_inference.attention"
SPLITTING_OPS = (SPLITTING_OP, SPLITTING_OP + ".default")

_op = None


def attention_op():
    """Registers the op on first use (once per process) and returns it."""
    global _op
    if _op is not None:
        return _op

    @torch.library.custom_op(OP_QUALNAME, mutates_args=())
    def attention(query: torch.Tensor, key: Optional[torch.Tensor], value: Optional[torch.Tensor],
                  layer_name: str) -> torch.Tensor:
        from vllm.forward_context import get_forward_context
        layer = get_forward_context().no_compile_layers[layer_name]
        out = layer._arctic_forward(query, key, value)
        # a custom op may not return one of its inputs (or a view of one)
        if out.data_ptr() == query.data_ptr():
            out = out.clone()
        return out

    @attention.register_fake
    def _(query, key, value, layer_name):
        width = query.shape[1] if query.dim() == 2 else query.shape[1] * query.shape[2]
        return query.new_empty((query.shape[0], width))

    _op = attention
    return _op


def call_attention(query, key, value, layer_name: str):
    """The patched forward's call: through the op PACKET, so that the traced node's target prints as SPLITTING_OP."""
    return torch.ops.arctic_inference.attention(query, key, value, layer_name)


def build_compilation_patches():
    """CompilationConfig.set_splitting_ops_for_v1 also lists this build's op (no effect under full_cuda_graph, where
    vLLM splits nowhere)."""
    from vllm.config import CompilationConfig

    from ..patching import ArcticPatch

    class CompilationConfigPatch(ArcticPatch[CompilationConfig]):
        _orig_set_splitting_ops_for_v1 = CompilationConfig.set_splitting_ops_for_v1

        def set_splitting_ops_for_v1(self):
            self._orig_set_splitting_ops_for_v1()
            if not getattr(self, "full_cuda_graph", False):
                self.splitting_ops = list(self.splitting_ops) + [s for s in SPLITTING_OPS if s not in self.splitting_ops]

    return [CompilationConfigPatch]
