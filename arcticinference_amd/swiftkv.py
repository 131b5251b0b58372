"""SwiftKV hot-path pieces: what LlamaSwiftKVModel does BETWEEN its two halves
(/root/reference/arctic_inference/vllm/swiftkv/llama_swiftkv.py).

A SwiftKV model runs its first `num_key_value_layers` layers on every token of the step and produces, in one
fused projection, the K/V of ALL remaining layers for those tokens ([T, Lkv * Hkv * D] each, :262-276).  Then,
before the decode half runs,

  1. under Ulysses SP the prefill half's outputs are all-gathered over the SP group (C7, :250-257): the decode half
     always runs in full TP (SP x TP), every rank needs every token               -> sp_all_gather()
  2. the K/V of all remaining layers are written into their paged caches at once (:599-628), the only in-repo
     caller of the bulk op A16 — straight from the strided projection output        -> SwiftKVSelector.write_kv()
  3. the attention metadata is rewritten so that only the sampled tokens continue (:418-431): query_start_loc by
     searchsorted over logits_indices, slot_mapping gathered                        -> fix_flash_attention_metadata()
  4. the five per-token tensors are gathered down to the sampled rows, into the decode runner's persistent graph
     buffers when the batch fits a captured size (index_fn, :665-685)                -> SwiftKVSelector.select()

The dense layers around these steps are vLLM's (LlamaDecoderLayer etc.) and are not part of this package; the
model class that strings them together registers with vLLM from the model's own repository.  MI355X notes: step
2 is one launch of kv_bulk_write_kernel for all layers (no per-layer loop, no chunk/view copies); step 4 is one
launch of row_gather_kernel for all five tensors instead of five index_select launches.
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import _native as N
from . import ops


def sp_all_gather(tensors: Sequence[torch.Tensor], sp_size: int, group) -> List[torch.Tensor]:
    """C7: every tensor [n, ...] -> [sp_size * n, ...], rank-major (parallel_state._SP.all_gather(x, dim=0))."""
    if sp_size == 1:
        return list(tensors)
    from .dist_utils import all_gather_into_tensor
    out = []
    for t in tensors:
        t = t.contiguous()
        full = torch.empty((sp_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        all_gather_into_tensor(full, t, group=group)
        out.append(full)
    return out


def fix_flash_attention_metadata(attn_metadata, logits_indices: torch.Tensor) -> None:
    """_fix_flash_attention_metadata (:418-431): after the selection, request i's queries are the sampled rows that
    fell inside its original query range."""
    attn_metadata.num_actual_tokens = int(logits_indices.numel())
    attn_metadata.query_start_loc = torch.searchsorted(logits_indices, attn_metadata.query_start_loc.to(logits_indices.dtype),
                                                       out_int32=True)
    attn_metadata.slot_mapping = attn_metadata.slot_mapping[logits_indices]
    if getattr(attn_metadata, "query_start_loc_cpu", None) is not None:
        attn_metadata.query_start_loc_cpu = None      # a host mirror some metadata classes keep: stale now
    # cascade attention is not combined with SwiftKV
    attn_metadata.use_cascade = False
    for name in ("cu_prefix_query_lens", "prefix_kv_lens", "suffix_kv_lens", "prefix_scheduler_metadata"):
        setattr(attn_metadata, name, None)


def row_gather(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor], index: torch.Tensor) -> None:
    """dst[t][i] = src[t][index[i]] for every tensor pair, one launch (aic_row_gather)."""
    n = len(srcs)
    assert n == len(dsts) and index.dtype == torch.int64 and index.is_cuda
    for s, d in zip(srcs, dsts):
        if not (s.is_cuda and d.is_cuda):
            raise RuntimeError("arcticinference_amd ops run on the GPU only; there is no CPU fallback")
        assert s.dtype == d.dtype and s.shape[1:] == d.shape[1:] and d.shape[0] >= index.numel()
        assert s.stride(-1) == 1 or s.dim() == 1
    VP, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
    row_bytes = [int(s[0].numel()) * s.element_size() if s.dim() > 1 else s.element_size() for s in srcs]
    N.check(N.lib().aic_row_gather(
        n, VP(*[s.data_ptr() for s in srcs]), VP(*[d.data_ptr() for d in dsts]),
        I64(*[int(s.stride(0)) * s.element_size() for s in srcs]), I64(*[int(d.stride(0)) * d.element_size() for d in dsts]),
        I32(*row_bytes), index.data_ptr(), int(index.numel()), int(srcs[0].shape[0]), N.current_stream_ptr()))


class SwiftKVSelector:
    """State of swiftkv_select for one model: the KV caches of the decode-half layers and the decode runner's graph
    input buffers (llama_swiftkv.py:392-413)."""

    NAMES = ("hidden_states", "residual", "positions", "k_states", "v_states")

    def __init__(self, hidden_size: int, num_kv_layers: int, num_kv_heads: int, head_size: int, dtype: torch.dtype,
                 device, cuda_graph_max_batch_size: int = 0, pad_for_cudagraph: Optional[Callable[[int], int]] = None):
        self.num_kv_layers, self.num_kv_heads, self.head_size = num_kv_layers, num_kv_heads, head_size
        self.kv_size = num_kv_layers * num_kv_heads * head_size
        self.max_graph = int(cuda_graph_max_batch_size)
        self.pad = pad_for_cudagraph or (lambda n: n)
        self.inputs = None
        if self.max_graph > 0:
            mk = lambda w, dt: torch.empty(self.max_graph, w, dtype=dt, device=device) if w else torch.empty(
                self.max_graph, dtype=dt, device=device)
            self.inputs = {"hidden_states": mk(hidden_size, dtype), "residual": mk(hidden_size, dtype),
                           "positions": mk(0, torch.long), "k_states": mk(self.kv_size, dtype),
                           "v_states": mk(self.kv_size, dtype)}
        self._writer = None
        self._writer_key = None

    # -- step 2 --------------------------------------------------------------------------------------------
    def write_kv(self, k_states: torch.Tensor, v_states: torch.Tensor, kv_caches: List[torch.Tensor], slot_mapping: torch.Tensor,
                 kv_cache_dtype: str, k_scales: List[torch.Tensor], v_scales: List[torch.Tensor]) -> None:
        """kv_caches: one [2, num_blocks, block_size, Hkv, D] tensor per decode-half layer (FlashAttention layout, :617);
        k_states / v_states [T, Lkv * Hkv * D], row stride taken as it is."""
        live = [c for c in kv_caches if c.numel()]
        if not live:
            return                                         # profile run: no cache bound yet
        # through the registered torch op, like the reference's caller (llama_swiftkv.py:599-628 -> py_custom_ops.py:52):
        # torch.ops.arctic_inference.reshape_and_cache_flash_bulk is visible to Dynamo, the ctypes call behind it is not.
        # The K / V views of the caches are made once per cache set (32 view objects per step otherwise).
        key = tuple(c.data_ptr() for c in live) + (kv_cache_dtype,)
        if self._writer_key != key:
            from . import py_custom_ops
            py_custom_ops.register_torch_ops()
            self._writer = ([c[0] for c in live], [c[1] for c in live])
            self._writer_key = key
        torch.ops.arctic_inference.reshape_and_cache_flash_bulk(k_states, v_states, self._writer[0], self._writer[1],
                                                                slot_mapping, kv_cache_dtype, list(k_scales), list(v_scales),
                                                                self.num_kv_heads, self.head_size)

    # -- step 4 --------------------------------------------------------------------------------------------
    def select(self, tensors: Sequence[torch.Tensor], logits_indices: torch.Tensor) -> Tuple[torch.Tensor, ...]:
        """(hidden_states, residual, positions, k_states, v_states) restricted to the sampled rows.  A batch that fits
        the captured sizes lands in the persistent buffers and comes back padded to the graph size (rows past the
        batch keep whatever the buffer held, as in the reference)."""
        n = int(logits_indices.numel())
        idx = logits_indices.to(torch.int64)
        if 0 < n <= self.max_graph:
            dsts = [self.inputs[name] for name in self.NAMES]
            row_gather(list(tensors), dsts, idx)
            padded = self.pad(n)
            return tuple(d[:padded] for d in dsts)
        outs = [torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in tensors]
        if n:
            row_gather(list(tensors), outs, idx)
        return tuple(outs)

    def graph_capture_inputs(self, batch_size: int) -> Optional[Tuple[torch.Tensor, ...]]:
        """What swiftkv_select returns while a graph is being captured / during the profile run (no metadata, :586-598)."""
        if self.inputs is None or batch_size > self.max_graph:
            return None
        padded = self.pad(batch_size)
        return tuple(self.inputs[name][:padded] for name in self.NAMES)


def swiftkv_select(selector: SwiftKVSelector, hidden_states, residual, positions, k_states, v_states, attn_metadata,
                   kv_caches: List[torch.Tensor], kv_cache_dtype: str, k_scales, v_scales):
    """LlamaSwiftKVModel.swiftkv_select (:573-685) for FlashAttention-layout metadata."""
    if attn_metadata is None:
        got = selector.graph_capture_inputs(hidden_states.shape[0])
        return got if got is not None else (hidden_states, residual, positions, k_states, v_states)
    selector.write_kv(k_states, v_states, kv_caches, attn_metadata.slot_mapping, kv_cache_dtype, k_scales, v_scales)
    logits_indices = attn_metadata.swiftkv_logits_indices
    fix_flash_attention_metadata(attn_metadata, logits_indices)
    return selector.select((hidden_states, residual, positions, k_states, v_states), logits_indices)
