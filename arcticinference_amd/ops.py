"""torch-facing wrappers over the C ABI (include/arctic_hip.h).  PyTorch is plumbing here: it owns
device memory and streams; every op below is a single call into libarctic_hip.so on the current
stream.  Nothing falls back to torch or the CPU — a missing library or device raises.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as N


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("arcticinference_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


# ------------------------------------------------------------------------------------------------
# A16 bulk KV write — same signature as arctic_inference.py_custom_ops.reshape_and_cache_flash_bulk
# (/root/reference/arctic_inference/py_custom_ops.py:40-54)
# ------------------------------------------------------------------------------------------------
_KV_DTYPES = {"auto": None, "fp8": N.DT_FP8_E4M3, "fp8_e4m3": N.DT_FP8_E4M3, "fp8_e5m2": N.DT_FP8_E5M2}


class KvBulkWriter:
    """reshape_and_cache_flash_bulk for a fixed set of caches: the checks and the per-layer pointer tables of the
    wrapper are done once (an engine calls it every step with the same caches)."""

    def __init__(self, key_caches: List[torch.Tensor], value_caches: List[torch.Tensor], kv_cache_dtype: str,
                 k_scales: List[torch.Tensor], v_scales: List[torch.Tensor], num_heads: int, head_size: int):
        L = len(key_caches)
        # same checks as the reference launcher (kernels.cu:103-106)
        if not (L == len(value_caches) == len(k_scales) == len(v_scales)):
            raise RuntimeError("key_caches, value_caches, k_scales and v_scales must have the same length")
        if kv_cache_dtype not in _KV_DTYPES:
            raise RuntimeError(f"unsupported kv cache dtype '{kv_cache_dtype}'")
        self.L = L
        if L == 0:
            return
        _need_cuda(*key_caches, *value_caches)
        self.kvd = _KV_DTYPES[kv_cache_dtype]
        self.cache_dtype = key_caches[0].dtype
        self.block_size = int(key_caches[0].size(1))
        self.block_stride = int(key_caches[0].stride(0))
        if self.block_stride != value_caches[0].stride(0):
            raise RuntimeError("key and value caches must share the block stride")
        VP = ctypes.c_void_p * L
        self._keep = (list(key_caches), list(value_caches), list(k_scales), list(v_scales))   # the tables hold raw pointers
        self.kc = VP(*[c.data_ptr() for c in key_caches])
        self.vc = VP(*[c.data_ptr() for c in value_caches])
        self.ks = VP(*[s.data_ptr() for s in k_scales])
        self.vs = VP(*[s.data_ptr() for s in v_scales])
        self.num_heads, self.head_size = int(num_heads), int(head_size)

    def __call__(self, keys: torch.Tensor, values: torch.Tensor, slot_mapping: torch.Tensor, stream: Optional[int] = None) -> None:
        if self.L == 0:
            return
        _need_cuda(keys, values, slot_mapping)
        if slot_mapping.dtype != torch.int64:
            raise RuntimeError("slot_mapping must be int64")
        src = N.torch_dtype_code(keys.dtype)
        kvd = self.kvd
        if kvd is None:
            kvd = src
            if self.cache_dtype != keys.dtype:
                raise RuntimeError("kv_cache_dtype 'auto' needs caches of the source dtype")
        N.check(N.lib().aic_reshape_and_cache_flash_bulk(
            keys.data_ptr(), values.data_ptr(), self.kc, self.vc, slot_mapping.data_ptr(), slot_mapping.size(0), self.L,
            self.num_heads, self.head_size, self.block_size, self.block_stride, int(keys.stride(0)),
            int(values.stride(0)), src, kvd, self.ks, self.vs, N.current_stream_ptr() if stream is None else stream))


def reshape_and_cache_flash_bulk(keys: torch.Tensor, values: torch.Tensor, key_caches: List[torch.Tensor],
                                 value_caches: List[torch.Tensor], slot_mapping: torch.Tensor, kv_cache_dtype: str,
                                 k_scales: List[torch.Tensor], v_scales: List[torch.Tensor], num_heads: int,
                                 head_size: int) -> None:
    if len(key_caches) == 0:
        return
    KvBulkWriter(key_caches, value_caches, kv_cache_dtype, k_scales, v_scales, num_heads, head_size)(keys, values, slot_mapping)


# ------------------------------------------------------------------------------------------------
# A6 rejection acceptance
# ------------------------------------------------------------------------------------------------
class RejectionResult:
    __slots__ = ("output_token_ids", "num_accepted", "last_token", "hidden_index")

    def __init__(self, output_token_ids, num_accepted, last_token, hidden_index):
        self.output_token_ids = output_token_ids
        self.num_accepted = num_accepted
        self.last_token = last_token
        self.hidden_index = hidden_index


_ws_cache = {}
_ws_retired = []     # outgrown workspaces stay alive: a kernel on a caller-supplied stream may still be reading one


def _workspace(nbytes: int, device, stream: Optional[int] = None, purpose: str = "attn") -> torch.Tensor:
    """Scratch memory of one (device, stream, purpose): calls enqueued on different streams never share partials, and
    attention (per-layer, possibly on a side stream) never shares with the acceptance kernel.  Sizes only grow (by
    doubling), so the retired list stays a handful of blocks."""
    if stream is None:
        stream = N.current_stream_ptr()
    key = (device.index, int(stream or 0), purpose)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _ws_retired.append(ws)
        ws = torch.empty(max(nbytes, 1 << 20, 0 if ws is None else 2 * ws.numel()), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def rejection_sample(target_logits: torch.Tensor, draft_token_ids: torch.Tensor, cu_num_draft_tokens: torch.Tensor,
                     bonus_token_ids: torch.Tensor, max_spec_len: int, temperature: Optional[torch.Tensor] = None,
                     uniform_probs: Optional[torch.Tensor] = None, exp_noise: Optional[torch.Tensor] = None,
                     target_row_index: Optional[torch.Tensor] = None,
                     bonus_row_index: Optional[torch.Tensor] = None) -> RejectionResult:
    """out[B, max_spec_len+1] int32 (-1 padded) as vLLM's RejectionSampler returns it for
    draft_probs=None (call site model_runner.py:405-411), plus the proposer inputs of the next step.
    `target_row_index` (int64 [num_draft_total]): pass the model's full [T, V] logits as `target_logits` and the
    rows to read (SpecDecodeMetadata.target_logits_indices) instead of a gathered copy.  `bonus_row_index` (int64 [B],
    greedy only): the bonus tokens are the arg-max of those rows of `target_logits`, computed in the same launch
    (`bonus_token_ids` may then be None)."""
    _need_cuda(target_logits, draft_token_ids, cu_num_draft_tokens, bonus_token_ids, target_row_index, bonus_row_index)
    if bonus_row_index is not None and (temperature is not None or bonus_row_index.dtype != torch.int64 or
                                        bonus_row_index.numel() != cu_num_draft_tokens.numel()):
        raise ValueError("bonus_row_index: int64, one row per request, greedy acceptance only")
    if bonus_token_ids is None and bonus_row_index is None:
        raise ValueError("bonus_token_ids or bonus_row_index is required")
    if target_row_index is not None and (target_row_index.dtype != torch.int64 or
                                         target_row_index.numel() != draft_token_ids.numel()):
        raise ValueError("target_row_index must be int64 with one entry per draft token")
    B = cu_num_draft_tokens.numel()
    rows = draft_token_ids.numel()
    V = target_logits.size(-1) if (rows or bonus_row_index is not None) else 1
    dev = cu_num_draft_tokens.device
    out = torch.empty((B, max_spec_len + 1), dtype=torch.int32, device=dev)
    nacc = torch.empty(B, dtype=torch.int32, device=dev)
    last = torch.empty(B, dtype=torch.int32, device=dev)
    hidx = torch.empty(B, dtype=torch.int32, device=dev)
    ws = _workspace(N.lib().aic_rejection_workspace_bytes(rows + (B if bonus_row_index is not None else 0), V), dev,
                    purpose="accept")
    draft = draft_token_ids.to(torch.int32)
    cu = cu_num_draft_tokens.to(torch.int32)
    bonus = None if bonus_token_ids is None else bonus_token_ids.reshape(-1).to(torch.int32)
    has_logits = rows > 0 or bonus_row_index is not None
    dt = N.torch_dtype_code(target_logits.dtype) if has_logits else N.DT_F32
    stride = target_logits.stride(0) if has_logits else 0
    if temperature is None:
        N.check(N.lib().aic_rejection_greedy(_ptr(target_logits) if has_logits else None, dt, stride, V, draft.data_ptr(),
                                             cu.data_ptr(), _ptr(bonus), B, rows, max_spec_len, out.data_ptr(),
                                             nacc.data_ptr(), last.data_ptr(), hidx.data_ptr(), _ptr(target_row_index),
                                             _ptr(bonus_row_index), ws.data_ptr(), N.current_stream_ptr()))
    else:
        N.check(N.lib().aic_rejection_random(_ptr(target_logits) if rows else None, dt, stride, V, draft.data_ptr(),
                                             cu.data_ptr(), bonus.data_ptr(), temperature.float().data_ptr(),
                                             uniform_probs.double().data_ptr(), exp_noise.float().data_ptr(), B, rows,
                                             max_spec_len, out.data_ptr(), nacc.data_ptr(), last.data_ptr(),
                                             hidx.data_ptr(), _ptr(target_row_index), ws.data_ptr(),
                                             N.current_stream_ptr()))
    return RejectionResult(out, nacc, last, hidx)


# ------------------------------------------------------------------------------------------------
# A5 verify attention
# ------------------------------------------------------------------------------------------------
def verify_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_table: torch.Tensor,
                     seq_lens: torch.Tensor, query_start_loc: torch.Tensor, max_q_len: int, max_seq_len: int,
                     sm_scale: float, out: Optional[torch.Tensor] = None, num_splits_max: int = 64,
                     q_lens_host: Optional[Sequence[int]] = None, req_split=None,
                     k_scale: Optional[torch.Tensor] = None, v_scale: Optional[torch.Tensor] = None,
                     stream: Optional[int] = None, sliding_window: int = 0,
                     sinks: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [T, Hq, D] (token stride may exceed Hq*D: a view into an all-to-all receive buffer works),
    caches [num_blocks, block_size, Hkv, D] in bf16, or float8_e4m3fn with per-tensor `k_scale` / `v_scale`
    (device scalars, the scales A16 divided by); returns [T, Hq, D].  `q_lens_host` (the per-request query
    lengths, which vLLM has on the host) lets long drafts take the shared-tile kernel.  `stream`: raw HIP stream
    handle (default: torch's current stream; a caller that issues one call per layer looks it up once).
    `sliding_window` W > 0 / `sinks` f32 [Hq]: the per-layer features of gpt-oss-class models (aic_verify_attention_win)."""
    _need_cuda(q, k_cache, v_cache, block_table, seq_lens, query_start_loc, sinks)
    if sinks is not None and (sinks.dtype != torch.float32 or sinks.numel() != q.shape[1] or not sinks.is_contiguous()):
        raise ValueError("sinks: contiguous float32, one value per query head")
    T, Hq, D = q.shape
    nb, bs, Hkv, D2 = k_cache.shape
    assert D == D2 and q.stride(2) == 1 and q.stride(1) == D
    if out is None:
        out = torch.empty((T, Hq, D), dtype=q.dtype, device=q.device)
    B = seq_lens.numel()
    wsb = N.lib().aic_verify_attention_workspace_bytes(T, Hq, D, num_splits_max)
    ws = _workspace(wsb, q.device, stream)
    kvd = N.torch_dtype_code(k_cache.dtype)
    if req_split is None and q_lens_host is not None:
        req_split = split_requests(q_lens_host, Hq // Hkv, q.device)
    short, n_short, long_, n_long = req_split if req_split is not None else (None, 0, None, 0)
    N.check(N.lib().aic_verify_attention_win(
        q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), k_cache.stride(0), kvd, _ptr(k_scale),
        _ptr(v_scale), block_table.data_ptr(), block_table.size(1), seq_lens.data_ptr(), query_start_loc.data_ptr(), B, T,
        int(max_q_len), Hq, Hkv, D, bs, float(sm_scale), out.data_ptr(), out.stride(0), ws.data_ptr(), ws.numel(),
        int(max_seq_len), _ptr(short) if n_short else None, n_short, _ptr(long_) if n_long else None, n_long,
        int(sliding_window or 0), _ptr(sinks), N.current_stream_ptr() if stream is None else stream))
    return out


class VerifyAttentionPlan:
    """verify_attention for a caller that issues one call per layer with only the cache pointers changing (every layer
    of a step shares the batch geometry): argument checks, workspace sizing and the ctypes argument list are built once
    per step, `run(k_cache, v_cache)` is one foreign call.  Same semantics as verify_attention()."""

    def __init__(self, q: torch.Tensor, out: torch.Tensor, kv_like: torch.Tensor, block_table: torch.Tensor,
                 seq_lens: torch.Tensor, query_start_loc: torch.Tensor, max_q_len: int, max_seq_len: int, sm_scale: float,
                 req_split=None, k_scale: Optional[torch.Tensor] = None, v_scale: Optional[torch.Tensor] = None,
                 num_splits_max: int = 64, stream: Optional[int] = None):
        _need_cuda(q, out, kv_like, block_table, seq_lens, query_start_loc)
        T, Hq, D = q.shape
        nb, bs, Hkv, D2 = kv_like.shape
        assert D == D2 and q.stride(2) == 1 and q.stride(1) == D and out.shape == q.shape
        ws = _workspace(N.lib().aic_verify_attention_workspace_bytes(T, Hq, D, num_splits_max), q.device, stream)
        short, n_short, long_, n_long = req_split if req_split is not None else (None, 0, None, 0)
        self._keep = (q, out, block_table, seq_lens, query_start_loc, ws, short, long_, k_scale, v_scale)
        self._fn = N.lib().aic_verify_attention_ex
        self._args = [q.data_ptr(), q.stride(0), 0, 0, kv_like.stride(0), N.torch_dtype_code(kv_like.dtype), _ptr(k_scale),
                      _ptr(v_scale), block_table.data_ptr(), block_table.size(1), seq_lens.data_ptr(),
                      query_start_loc.data_ptr(), seq_lens.numel(), T, int(max_q_len), Hq, Hkv, D, bs, float(sm_scale),
                      out.data_ptr(), out.stride(0), ws.data_ptr(), ws.numel(), int(max_seq_len),
                      _ptr(short) if n_short else None, n_short, _ptr(long_) if n_long else None, n_long,
                      N.current_stream_ptr() if stream is None else stream]
        self._kv_shape, self._kv_dtype, self._kv_stride = tuple(kv_like.shape), kv_like.dtype, kv_like.stride(0)

    def layer_tables(self, k_caches: Sequence[torch.Tensor], v_caches: Sequence[torch.Tensor]):
        """Pointer tables for run_layers() (build once for a fixed set of caches)."""
        for c in list(k_caches) + list(v_caches):
            if c.shape != self._kv_shape or c.dtype is not self._kv_dtype or c.stride(0) != self._kv_stride:
                raise ValueError("the plan was made for caches of another shape / dtype")
        VP = ctypes.c_void_p * len(k_caches)
        return (VP(*[c.data_ptr() for c in k_caches]), VP(*[c.data_ptr() for c in v_caches]), len(k_caches),
                (list(k_caches), list(v_caches)))

    def run_layers(self, tables) -> None:
        """Every layer of the step in ONE foreign call (aic_verify_attention_layers): same q / out for all layers, as
        the stand-alone engine has them."""
        a = self._args
        kt, vt, n, _keep = tables
        N.check(N.lib().aic_verify_attention_layers(a[0], a[1], 0, kt, vt, n, *a[4:20], a[20], a[21], 0, *a[22:]))

    def run(self, k_cache: torch.Tensor, v_cache: torch.Tensor) -> None:
        if k_cache.shape != self._kv_shape or k_cache.dtype is not self._kv_dtype or k_cache.stride(0) != self._kv_stride:
            raise ValueError("the plan was made for caches of another shape / dtype")
        a = self._args
        a[2] = k_cache.data_ptr()
        a[3] = v_cache.data_ptr()
        N.check(self._fn(*a))


def split_order(q_lens_host: Sequence[int], group_size: int):
    """Host half of split_requests: (request ids with the short ones first, as int32; number of short requests).
    Short = q_len * Hq/Hkv <= 32 query rows.  A batch of short requests only still gets its (identity) list: the
    partitioned call is what lets every workgroup take the one- or two-row-tile form by its own request's rows, where
    the call without lists has to size every request for the longest."""
    import numpy as np
    ql = np.asarray(q_lens_host)
    # same rule as aic_verify_attention_ex
    is_short = ql * group_size <= 32
    n_short = int(is_short.sum())
    if n_short == len(ql):
        return np.arange(len(ql), dtype=np.int32), n_short
    order = np.concatenate([np.nonzero(is_short)[0], np.nonzero(~is_short)[0]]).astype(np.int32)
    return order, n_short


def split_requests(q_lens_host: Sequence[int], group_size: int, device, order_dev: Optional[torch.Tensor] = None):
    """Partition a batch by query length for aic_verify_attention_ex: (short_ids, n_short, long_ids, n_long) with
    device int32 id lists.  Build it once per engine step and pass it to every layer's verify_attention call.
    `order_dev`: the id list already on the device (a caller that stages all of a step's index arrays in one copy
    passes split_order()'s array through that copy)."""
    order, n_short = split_order(q_lens_host, group_size)
    lists = order_dev if order_dev is not None else torch.from_numpy(order).to(device, non_blocking=True)
    return lists[:n_short], n_short, lists[n_short:], len(order) - n_short


# ------------------------------------------------------------------------------------------------
# A12 Ulysses repartition copies
# ------------------------------------------------------------------------------------------------
def ulysses_pack_qkv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, sp: int) -> torch.Tensor:
    """q [n, SP*qw], k/v [n, SP*kw] -> send [SP*n, qw+2kw] (rank-major), ulysses.py:493-499."""
    _need_cuda(q, k, v)
    n = q.size(0)
    qw, kw = q.size(1) // sp, k.size(1) // sp
    send = torch.empty((sp * n, qw + 2 * kw), dtype=q.dtype, device=q.device)
    N.check(N.lib().aic_ulysses_pack_qkv(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0),
                                         v.stride(0), send.data_ptr(), n, sp, qw, kw, N.current_stream_ptr()))
    return send


def ulysses_split_qkv(recv: torch.Tensor, qw: int, kw: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _need_cuda(recv)
    rows = recv.size(0)
    q = torch.empty((rows, qw), dtype=recv.dtype, device=recv.device)
    k = torch.empty((rows, kw), dtype=recv.dtype, device=recv.device)
    v = torch.empty((rows, kw), dtype=recv.dtype, device=recv.device)
    N.check(N.lib().aic_ulysses_split_qkv(recv.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), rows, qw, kw,
                                          N.current_stream_ptr()))
    return q, k, v


def ulysses_unpack_out(recv: torch.Tensor, sp: int) -> torch.Tensor:
    """recv [SP*n, w] (rank-major) -> out [n, SP*w], ulysses.py:515-517."""
    _need_cuda(recv)
    n = recv.size(0) // sp
    w = recv.size(1)
    out = torch.empty((n, sp * w), dtype=recv.dtype, device=recv.device)
    N.check(N.lib().aic_ulysses_unpack_out(recv.data_ptr(), out.data_ptr(), n, sp, w, N.current_stream_ptr()))
    return out


def ulysses_pack_pair(a: torch.Tensor, b: Optional[torch.Tensor], parts: int) -> torch.Tensor:
    """KV-replicated variant (ulysses.py:463-474): a [n, parts*aw] (and b [n, parts*bw]) -> send [parts*n, aw (+ bw)],
    part-major: the q pack for the all-to-all over SP (b None) and the K|V pack for the all-to-all inside SP_AA."""
    _need_cuda(a, b)
    n = a.size(0)
    aw = a.size(1) // parts
    bw = 0 if b is None else b.size(1) // parts
    send = torch.empty((parts * n, aw + bw), dtype=a.dtype, device=a.device)
    N.check(N.lib().aic_ulysses_pack_pair(a.data_ptr(), _ptr(b), a.stride(0), 0 if b is None else b.stride(0),
                                          send.data_ptr(), n, parts, aw, bw, N.current_stream_ptr()))
    return send


def ulysses_reorder_split_kv(gathered: torch.Tensor, sp: int, order: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """gathered [sp*n, 2*kw] in all-gather chunk order -> (k, v) [sp*n, kw] each with chunk c taken from chunk order[c]
    (ulysses.py:486-490: chunk / cat in `self.order` / split)."""
    _need_cuda(gathered)
    rows, w2 = gathered.shape
    n, kw = rows // sp, w2 // 2
    k = torch.empty((rows, kw), dtype=gathered.dtype, device=gathered.device)
    v = torch.empty((rows, kw), dtype=gathered.dtype, device=gathered.device)
    arr = (ctypes.c_int32 * sp)(*[int(x) for x in order])
    N.check(N.lib().aic_ulysses_reorder_split_kv(gathered.data_ptr(), k.data_ptr(), v.data_ptr(), n, sp, kw, arr,
                                                 N.current_stream_ptr()))
    return k, v


def quantize_fp8_per_tensor(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_cuda(x)
    x = x.contiguous()
    q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    N.check(N.lib().aic_quantize_fp8_per_tensor(x.data_ptr(), q.data_ptr(), scale.data_ptr(), x.numel(),
                                                N.current_stream_ptr()))
    return q, scale
