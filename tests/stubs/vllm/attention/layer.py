"""`Attention` of the stand-in: vLLM's constructor / forward signature over a plain torch paged attention.

forward() does what vLLM's unified attention op does in V1: write the step's K/V into the paged cache at
`slot_mapping`, then causal variable-length attention of every request's query tokens over its cached context
(cache layout [2, num_blocks, block_size, num_kv_heads, head_size]).  It is the "vLLM backend" that the plugin's
HIP route is compared with in tests."""
from dataclasses import dataclass
from typing import Any, Optional

import torch

from vllm.forward_context import get_forward_context


@dataclass
class AttentionMetadata:
    num_actual_tokens: int
    max_query_len: int
    query_start_loc: torch.Tensor      # int32 [B + 1]
    max_seq_len: int
    seq_lens: torch.Tensor             # int32 [B]
    block_table: torch.Tensor          # int32 [B, max_blocks]
    slot_mapping: torch.Tensor         # int64 [T]
    # host copies (the stand-in's torch loop uses them; vLLM's kernels read the device tensors)
    query_start_loc_cpu: Any = None
    seq_lens_cpu: Any = None


class _Impl:
    def __init__(self, scale, sinks=None):
        self.scale = scale
        self.alibi_slopes = None
        self.logits_soft_cap = None
        self.sinks = sinks          # [num_heads] extra soft-max logits (gpt-oss), or None


class Attention(torch.nn.Module):
    calls = 0      # forwards that reached THIS implementation (tests: was the HIP route taken instead?)

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes=None, cache_config=None, quant_config=None, blocksparse_params=None, logits_soft_cap=None,
                 per_layer_sliding_window=None, use_mla: bool = False, prefix: str = "", attn_type: str = "decoder", **extra):
        super().__init__()
        self.num_heads, self.head_size = num_heads, head_size
        self.num_kv_heads = num_kv_heads if num_kv_heads is not None else num_heads
        self.impl = _Impl(scale, extra.get("sinks"))
        self.layer_name = prefix
        self.sliding_window = per_layer_sliding_window
        self.kv_cache = [torch.tensor([])]
        self._k_scale = torch.tensor(1.0, dtype=torch.float32)
        self._v_scale = torch.tensor(1.0, dtype=torch.float32)
        from vllm.config import get_current_vllm_config
        cfg = get_current_vllm_config()
        if cfg is not None:
            ctx = cfg.compilation_config.static_forward_context
            if prefix in ctx:
                raise ValueError(f"Duplicate layer name: {prefix}")
            ctx[prefix] = self

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, output_shape=None) -> torch.Tensor:
        Attention.calls += 1
        ctx = get_forward_context()
        meta = ctx.attn_metadata
        if isinstance(meta, dict):
            meta = meta[self.layer_name]
        kv = self.kv_cache[ctx.virtual_engine]
        Hq, Hkv, D = self.num_heads, self.num_kv_heads, self.head_size
        out = torch.zeros(query.shape[0], Hq * D, dtype=query.dtype, device=query.device)
        if meta is None or kv.numel() == 0:     # profile / warm-up run without metadata, or no cache bound yet
            return out
        n = meta.num_actual_tokens
        bs = kv.shape[2]
        kc, vc = kv[0].view(-1, Hkv, D), kv[1].view(-1, Hkv, D)
        slots = meta.slot_mapping[:n]
        kc[slots] = key[:n].view(n, Hkv, D).to(kc.dtype)
        vc[slots] = value[:n].view(n, Hkv, D).to(vc.dtype)
        qsl = meta.query_start_loc_cpu if meta.query_start_loc_cpu is not None else meta.query_start_loc.cpu()
        sls = meta.seq_lens_cpu if meta.seq_lens_cpu is not None else meta.seq_lens.cpu()
        G = Hq // Hkv
        q = query[:n].view(n, Hq, D).float()
        for i in range(len(sls)):
            q0, q1, ctx_len = int(qsl[i]), int(qsl[i + 1]), int(sls[i])
            ql = q1 - q0
            if ql == 0:
                continue
            nblk = (ctx_len + bs - 1) // bs
            idx = (meta.block_table[i, :nblk].long().unsqueeze(1) * bs + torch.arange(bs, device=query.device)).reshape(-1)[:ctx_len]
            K = kc[idx].float().repeat_interleave(G, dim=1)
            V = vc[idx].float().repeat_interleave(G, dim=1)
            s = torch.einsum("qhd,khd->hqk", q[q0:q1], K) * self.impl.scale
            pos = torch.arange(ql, device=query.device).unsqueeze(1) + (ctx_len - ql)
            mask = torch.arange(ctx_len, device=query.device).unsqueeze(0) <= pos
            if self.sliding_window:                 # the last `sliding_window` keys, the query's own included
                mask = mask & (torch.arange(ctx_len, device=query.device).unsqueeze(0) > pos - self.sliding_window)
            s = s.masked_fill(~mask.unsqueeze(0), float("-inf"))
            if self.impl.sinks is not None:         # one extra logit per head in the normalisation, no value
                col = self.impl.sinks.float().to(s.device).view(Hq, 1, 1).expand(Hq, ql, 1)
                p = torch.softmax(torch.cat([s, col], dim=-1), dim=-1)[..., :ctx_len]
            else:
                p = torch.softmax(s, dim=-1)
            out[q0:q1] = torch.einsum("hqk,khd->qhd", p, V).reshape(ql, Hq * D).to(out.dtype)
        return out
