"""ORACLE — test infrastructure only (never imported by the product package).

CPU restatements (torch on the CPU / numpy) of the parts of the hot path that live in Python or in
third-party code on the reference side.  Each function cites what it follows.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY STATUS
  * lstm_*            : literal restatement of the reference's torch op sequence
                        (arctic_speculator.py:88-95, :648-751; fp8.py:207-223, :276-308), executed with
                        torch CPU bf16 tensors so every op rounds where the reference's does.  The
                        reference module cannot be imported (vLLM absent) and ships no fixture for it:
                        PARITY UNPINNED against the reference itself; pinned to this restatement.
  * rejection_*       : vLLM 0.9.2 RejectionSampler semantics for draft_probs=None, recalled
                        (SURVEY.md §8a A6); no source or fixture under /root/reference: PARITY UNPINNED.
  * verify_attention  : plain fp32 softmax(QK^T)V with the causal rule of SURVEY §8a A5: PARITY UNPINNED
                        (tolerance 1e-3 in bf16 per BASELINE.json).
  * kv_bulk_write     : restates csrc/custom_ops/kernels.cu:29-68; pinned only in interface/shape by the
                        reference's own (self-comparing) test tests/unit_tests/test_custom_ops.py:56-118.
  * ulysses_*         : the literal torch expression of ulysses.py:493-517, with a single-process
                        emulation of all_to_all_single.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

SQRT2 = 2 ** 0.5


# ----------------------------------------------------------------------------------------------
# LSTM speculator (arctic_speculator.py)
# ----------------------------------------------------------------------------------------------
def _l2norm(x: torch.Tensor, weight=None, bias=None, eps: float = 1e-6) -> torch.Tensor:
    """MLPSpeculatorLayerNorm.forward (arctic_speculator.py:88-95), dtype-preserving like the original."""
    xf = x
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    x = xf.type_as(x)
    if weight is not None:
        x = weight * x
        x = x + bias
    return x


def fp8_quant_per_tensor(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dynamic per-tensor e4m3fn quantisation (ops.scaled_fp8_quant(x, scale=None), fp8.py:209):
    scale = max(amax/448, 1/(448*512)); q = sat(x * (1/scale))."""
    amax = x.abs().max().to(torch.float32)
    scale = torch.maximum(amax / 448.0, torch.tensor(1.0 / (448.0 * 512.0)))
    inv = (torch.tensor(1.0, dtype=torch.float32) / scale)
    q = (x.to(torch.float32) * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, scale.reshape(1)


def lstm_generate_proposals(w: Dict[str, torch.Tensor], input_ids: torch.Tensor, hidden: torch.Tensor, k: int,
                            n_predict: int, scale_input: bool = True, fp8_head: bool = False,
                            return_logits: bool = False, forced_tokens: torch.Tensor = None):
    """`forced_tokens` [B, k]: feed THESE tokens to the next head instead of the oracle's own arg-max (teacher forcing:
    lets a test judge every head of a row after the kernel and the oracle parted at a near-tie).
    generate_states + generate_token_ids for method == "sum_lstm", tp_size == 1
    (arctic_speculator.py:648-751), on CPU bf16.  `w` uses the module's parameter names after the
    reference loader ran: projs.{0,1}.weight = cat(forget, input, output, cell) (:886-891)."""
    dt = torch.bfloat16
    Ds = w["cell_ln.0.weight"].numel()
    state_weight = 0.5 ** (0.5 / n_predict)
    emb_weight = math.sqrt((1 - state_weight ** 2) * (Ds / 2))
    prev = hidden.to(dt).unsqueeze(1)          # b 1 d
    last = input_ids.long().unsqueeze(1)       # b 1
    cell = torch.zeros(prev.shape[0], 1, Ds, dtype=dt)
    head_w = w["head.0.weight"].to(dt)
    if fp8_head:
        qw, w_scale = fp8_quant_per_tensor(head_w)      # post-load hook, fp8.py:207-223
        qw_f = qw.to(torch.float32)
    gelu = torch.nn.GELU()
    out, all_logits = [], []
    for head_index in range(k):
        if head_index == 0 and scale_input:
            prev = _l2norm(prev) / SQRT2
        proj = w["projs.0.weight"] if head_index == 0 else w["projs.1.weight"]
        z = torch.nn.functional.embedding(last, w["forget_emb.0.weight"].to(dt)).repeat(1, 1, 4)
        states = torch.nn.functional.linear(prev, proj.to(dt))
        added = torch.add(states, z, alpha=emb_weight / state_weight)
        fio, cand = added.split([Ds * 3, Ds], dim=-1)
        f, i, o = torch.sigmoid(fio).split([Ds, Ds, Ds], dim=-1)
        cand = gelu(_l2norm(cand, w["cell_ln.0.weight"].to(dt), w["cell_ln.0.bias"].to(dt)))
        cand = cand * i
        cell = cell * f
        cell = cell + cand
        st = gelu(_l2norm(cell, w["state_ln.0.weight"].to(dt), w["state_ln.0.bias"].to(dt)))
        state = st * o
        prev = state
        flat = state.flatten(0, 1)
        if fp8_head:
            qx, x_scale = fp8_quant_per_tensor(flat)    # Fp8LinearOp dynamic activation scale
            logits = ((qx.to(torch.float32) @ qw_f.t()) * (x_scale * w_scale)).to(dt)
        else:
            logits = torch.nn.functional.linear(flat, head_w)
        last = torch.argmax(logits, dim=-1).reshape(-1, 1)
        out.append(last)
        all_logits.append(logits)
        if forced_tokens is not None:
            last = forced_tokens[:, head_index].long().reshape(-1, 1)
    toks = torch.cat(out, dim=-1)
    return (toks, all_logits) if return_logits else toks


def mlp_generate_proposals(w: Dict[str, torch.Tensor], input_ids: torch.Tensor, hidden: torch.Tensor, k: int,
                           n_predict: int, inner_dim: int, tie_weights: bool, scale_input: bool = False,
                           fp8_head: bool = False, return_logits: bool = False, forced_tokens: torch.Tensor = None,
                           stacks: Tuple[int, int, int] = (0, 0, 0)):
    """ArcticMLPSpeculator.generate_states + generate_token_ids, tp_size == 1 (arctic_speculator.py:264-321), on CPU
    bf16 with the module's parameter names (emb.i / proj.i / ln.i / head.i; tied models keep stage 0, proj 0 and 1).
    `stacks` = (extra emb stages, extra proj stages, extra ln stages) of the LSTM class's "sum_rnn" form with multi-entry
    dimension lists (arctic_speculator.py:478-542): emb.i / proj.i are nn.Sequentials [base, (LayerNorm, GELU, Linear)*]
    with parameters `emb.i.{3j-2}.weight|bias`, `emb.i.{3j}.weight`; ln.i is [LayerNorm, (GELU, Linear, LayerNorm)*] with
    `ln.i.{3j-1}.weight`, `ln.i.{3j}.weight|bias`; the base modules keep the names `emb.i.weight`, `proj.i.weight`,
    `ln.i.weight|bias` here (the loader maps `*.0.*` to them)."""
    dt = torch.bfloat16
    state_weight = 0.5 ** (0.5 / n_predict)
    emb_weight = math.sqrt((1 - state_weight ** 2) * (inner_dim / 2))
    stage = (lambda i, kind: (min(i, 1) if kind == "proj" else 0)) if tie_weights else (lambda i, kind: i)
    prev = hidden.to(dt).unsqueeze(1)          # b 1 d
    last = input_ids.long().unsqueeze(1)       # b 1
    gelu = torch.nn.GELU()
    out, all_logits = [], []
    for i in range(k):
        if i == 0 and scale_input:
            prev = _l2norm(prev) / SQRT2
        F = torch.nn.functional
        se, sp, sl = stage(i, 'emb'), stage(i, 'proj'), stage(i, 'ln')
        z = F.embedding(last, w[f"emb.{se}.weight"].to(dt))
        for j in range(1, stacks[0] + 1):          # Sequential: LayerNorm, GELU, Linear
            z = F.linear(gelu(_l2norm(z, w[f"emb.{se}.{3 * j - 2}.weight"].to(dt), w[f"emb.{se}.{3 * j - 2}.bias"].to(dt))),
                         w[f"emb.{se}.{3 * j}.weight"].to(dt))
        states = F.linear(prev, w[f"proj.{sp}.weight"].to(dt))
        for j in range(1, stacks[1] + 1):
            states = F.linear(gelu(_l2norm(states, w[f"proj.{sp}.{3 * j - 2}.weight"].to(dt),
                                           w[f"proj.{sp}.{3 * j - 2}.bias"].to(dt))), w[f"proj.{sp}.{3 * j}.weight"].to(dt))
        states.add_(z, alpha=emb_weight / state_weight)
        y = _l2norm(states, w[f"ln.{sl}.weight"].to(dt), w[f"ln.{sl}.bias"].to(dt))
        for j in range(1, stacks[2] + 1):          # Sequential tail: GELU, Linear, LayerNorm
            y = _l2norm(F.linear(gelu(y), w[f"ln.{sl}.{3 * j - 1}.weight"].to(dt)), w[f"ln.{sl}.{3 * j}.weight"].to(dt),
                        w[f"ln.{sl}.{3 * j}.bias"].to(dt))
        states = gelu(y)
        prev = states
        flat = states.flatten(0, 1)
        head_w = w[f"head.{stage(i, 'head')}.weight"].to(dt)
        if fp8_head:
            qw, w_scale = fp8_quant_per_tensor(head_w)
            qx, x_scale = fp8_quant_per_tensor(flat)
            logits = ((qx.to(torch.float32) @ qw.to(torch.float32).t()) * (x_scale * w_scale)).to(dt)
        else:
            logits = torch.nn.functional.linear(flat, head_w)
        last = torch.argmax(logits, dim=-1).reshape(-1, 1)
        out.append(last)
        all_logits.append(logits)
        if forced_tokens is not None:
            last = forced_tokens[:, i].long().reshape(-1, 1)
    toks = torch.cat(out, dim=-1)
    return (toks, all_logits) if return_logits else toks


def merge_lstm_checkpoint(ckpt: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The reference loader's renaming (arctic_speculator.py:874-902) on a plain dict."""
    w = dict(ckpt)
    for drop in ("input_emb.0.weight", "cell_emb.0.weight", "output_emb.0.weight"):
        w.pop(drop, None)
    for i in (0, 1):
        w[f"projs.{i}.weight"] = torch.cat([w.pop(f"{g}_proj.{i}.weight") for g in ("forget", "input", "output", "cell")])
    return w


# ----------------------------------------------------------------------------------------------
# hidden-state pick (arctic_proposer.py:133-147)
# ----------------------------------------------------------------------------------------------
def hidden_state_index(sampled_token_ids: np.ndarray, num_draft_tokens: Sequence[int]) -> np.ndarray:
    valid = sampled_token_ids != -1
    gen_lens = valid.sum(axis=1)
    n = np.asarray(num_draft_tokens) + 1
    return (gen_lens - 1) + np.cumsum(n) - n


# ----------------------------------------------------------------------------------------------
# rejection acceptance (vLLM RejectionSampler, draft_probs=None) — recalled semantics
# ----------------------------------------------------------------------------------------------
def rejection_greedy(target_logits: torch.Tensor, draft_token_ids: Sequence[int], num_draft_tokens: Sequence[int],
                     bonus_token_ids: Sequence[int], max_spec_len: int) -> np.ndarray:
    B = len(num_draft_tokens)
    out = np.full((B, max_spec_len + 1), -1, dtype=np.int32)
    argmax = torch.argmax(target_logits.float(), dim=-1).numpy() if target_logits.numel() else np.zeros(0, np.int64)
    start = 0
    for i, n in enumerate(num_draft_tokens):
        rejected = False
        for pos in range(n):
            if not rejected:
                tgt = int(argmax[start + pos])
                out[i, pos] = tgt
                if int(draft_token_ids[start + pos]) != tgt:
                    rejected = True
        if not rejected:
            out[i, n] = int(bonus_token_ids[i])
        start += n
    return out


def rejection_random(target_logits: torch.Tensor, draft_token_ids: Sequence[int], num_draft_tokens: Sequence[int],
                     bonus_token_ids: Sequence[int], max_spec_len: int, temperature: Sequence[float],
                     uniform: np.ndarray, exp_noise: torch.Tensor) -> np.ndarray:
    """Rows with temperature <= 0 are greedy.  Random rows: x = logits.div_(T) in the logits dtype,
    p = softmax(x, fp32); accept the draft iff p[draft] >= u; otherwise emit
    argmax(p / q) with p[draft] := 0, q ~ Exp(1) per request."""
    B = len(num_draft_tokens)
    out = np.full((B, max_spec_len + 1), -1, dtype=np.int32)
    start = 0
    for i, n in enumerate(num_draft_tokens):
        rejected = False
        T = float(temperature[i])
        for pos in range(n):
            if rejected:
                break
            row = target_logits[start + pos]
            draft = int(draft_token_ids[start + pos])
            if T <= 0:
                tok = int(torch.argmax(row.float()))
                out[i, pos] = tok
                rejected = draft != tok
            else:
                x = (row / T).to(row.dtype)
                p = torch.softmax(x.float(), dim=-1)
                if float(p[draft]) >= float(uniform[start + pos]):
                    out[i, pos] = draft
                else:
                    p2 = p.clone()
                    p2[draft] = 0.0
                    out[i, pos] = int(torch.argmax(p2 / exp_noise[i].float()))
                    rejected = True
        if not rejected:
            out[i, n] = int(bonus_token_ids[i])
        start += n
    return out


# ----------------------------------------------------------------------------------------------
# verify attention — fp32 reference
# ----------------------------------------------------------------------------------------------
def verify_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_table: torch.Tensor,
                     seq_lens: Sequence[int], query_start_loc: Sequence[int], sm_scale: float, k_scale: float = 1.0,
                     v_scale: float = 1.0, sliding_window: int = 0, sinks: torch.Tensor = None) -> torch.Tensor:
    """q [T,Hq,D]; caches [num_blocks, block_size, Hkv, D]; returns fp32 [T,Hq,D].
    gpt-oss layers (BASELINE configs[4]; semantics of vLLM's attention backends, recalled — parity unpinned):
    `sliding_window` W > 0: query position p sees keys p - W + 1 .. p only (W keys, itself included);
    `sinks` f32 [Hq]: one extra logit per head that takes part in the soft-max normalisation and carries no value
    (softmax over [scores, sink], the sink column dropped): it is NOT multiplied by sm_scale."""
    T, Hq, D = q.shape
    _, bs, Hkv, _ = k_cache.shape
    G = Hq // Hkv
    out = torch.zeros(T, Hq, D, dtype=torch.float32)
    for i, ctx in enumerate(seq_lens):
        q0, q1 = int(query_start_loc[i]), int(query_start_loc[i + 1])
        qlen = q1 - q0
        if qlen == 0:
            continue
        nblk = (ctx + bs - 1) // bs
        blocks = block_table[i, :nblk].long()
        K = k_cache[blocks].reshape(-1, Hkv, D)[:ctx].float() * k_scale  # [ctx, Hkv, D]; fp8 caches hold x / scale
        V = v_cache[blocks].reshape(-1, Hkv, D)[:ctx].float() * v_scale
        Q = q[q0:q1].float()                                     # [qlen, Hq, D]
        Kh = K.repeat_interleave(G, dim=1)                       # [ctx, Hq, D]
        Vh = V.repeat_interleave(G, dim=1)
        s = torch.einsum("qhd,khd->hqk", Q, Kh) * sm_scale
        pos = torch.arange(qlen).unsqueeze(1) + (ctx - qlen)
        mask = torch.arange(ctx).unsqueeze(0) <= pos             # [qlen, ctx]
        if sliding_window and sliding_window > 0:
            mask = mask & (torch.arange(ctx).unsqueeze(0) > pos - sliding_window)
        s = s.masked_fill(~mask.unsqueeze(0), float("-inf"))
        if sinks is not None:
            col = sinks.float().view(Hq, 1, 1).expand(Hq, qlen, 1)
            p = torch.softmax(torch.cat([s, col], dim=-1), dim=-1)[..., :ctx]
        else:
            p = torch.softmax(s, dim=-1)
        out[q0:q1] = torch.einsum("hqk,khd->qhd", p, Vh)
    return out


# ----------------------------------------------------------------------------------------------
# bulk KV write (csrc/custom_ops/kernels.cu:29-68)
# ----------------------------------------------------------------------------------------------
def fp8_sat(x: torch.Tensor, kind: str) -> torch.Tensor:
    """x (fp32) -> OCP fp8 with saturation to the largest finite value (CUDA __NV_SATFINITE)."""
    if kind == "e4m3":
        return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return x.clamp(-57344.0, 57344.0).to(torch.float8_e5m2)


def kv_bulk_write(keys, values, key_caches, value_caches, slot_mapping, kv_cache_dtype, k_scales, v_scales,
                  num_heads, head_size) -> None:
    n = num_heads * head_size
    bs = key_caches[0].shape[1]
    for layer in range(len(key_caches)):
        for t in range(slot_mapping.numel()):
            slot = int(slot_mapping[t])
            if slot < 0:
                continue
            b, o = slot // bs, slot % bs
            ksrc = keys[t, layer * n:(layer + 1) * n].reshape(num_heads, head_size)
            vsrc = values[t, layer * n:(layer + 1) * n].reshape(num_heads, head_size)
            if kv_cache_dtype == "auto":
                key_caches[layer][b, o] = ksrc
                value_caches[layer][b, o] = vsrc
            else:
                kind = "e5m2" if kv_cache_dtype == "fp8_e5m2" else "e4m3"
                key_caches[layer][b, o] = fp8_sat(ksrc.float() / k_scales[layer].float(), kind)
                value_caches[layer][b, o] = fp8_sat(vsrc.float() / v_scales[layer].float(), kind)


# ----------------------------------------------------------------------------------------------
# Ulysses pack / unpack (ulysses.py:493-517) with a single-process all_to_all_single emulation
# ----------------------------------------------------------------------------------------------
def ulysses_pack(q, k, v, sp, hq, hkv, D):
    return (torch.cat((q.view(-1, sp, hq * D), k.view(-1, sp, hkv * D), v.view(-1, sp, hkv * D)), dim=-1)
            .transpose(0, 1).reshape(-1, (hq + 2 * hkv) * D))


def ulysses_unpack(c, sp, hq, D):
    return c.view(sp, -1, hq * D).transpose(0, 1).reshape(-1, hq * sp * D)


def ulysses_pack_pair(a, b, parts):
    """KV-replicated variant: the q pack (b None, ulysses.py:464-467) and the K|V pack (:471-474), part-major."""
    n = a.shape[0]
    if b is None:
        return a.view(n, parts, -1).transpose(0, 1).reshape(parts * n, -1)
    return torch.cat((a.view(n, parts, -1), b.view(n, parts, -1)), dim=-1).transpose(0, 1).reshape(parts * n, -1)


def ulysses_reorder_split(gathered, sp, order):
    """ulysses.py:486-490: chunk the all-gathered K|V, concatenate the chunks in `order`, split K from V."""
    chunks = gathered.chunk(sp)
    ordered = torch.cat([chunks[i] for i in order])
    kw = gathered.shape[1] // 2
    k, v = ordered.split([kw, kw], dim=-1)
    return k.contiguous(), v.contiguous()


def all_to_all_single_emulated(per_rank_inputs: List[torch.Tensor]) -> List[torch.Tensor]:
    """Equal-split all_to_all_single over len(per_rank_inputs) ranks: rank r receives chunk r of everyone."""
    sp = len(per_rank_inputs)
    chunks = [t.chunk(sp, dim=0) for t in per_rank_inputs]
    return [torch.cat([chunks[src][dst] for src in range(sp)], dim=0) for dst in range(sp)]


# ----------------------------------------------------------------------------------------------
# SwiftKV selection (vllm/swiftkv/llama_swiftkv.py:418-431, :573-685) — literal torch expressions
# ----------------------------------------------------------------------------------------------
def swiftkv_select(hidden_states, residual, positions, k_states, v_states, query_start_loc, slot_mapping, logits_indices,
                   key_caches, value_caches, kv_cache_dtype, k_scales, v_scales, num_heads, head_size):
    """Returns (selected five tensors, new query_start_loc, new slot_mapping); the caches are written in place."""
    kv_bulk_write(k_states, v_states, key_caches, value_caches, slot_mapping, kv_cache_dtype, k_scales, v_scales, num_heads,
                  head_size)
    new_qsl = torch.searchsorted(logits_indices, query_start_loc.to(logits_indices.dtype), out_int32=True)
    new_slots = slot_mapping[logits_indices]
    sel = tuple(t.index_select(0, logits_indices) for t in (hidden_states, residual, positions, k_states, v_states))
    return sel, new_qsl, new_slots


# ---------------------------------------------------------------------------------------------------------------------
# SwiftKV Llama, whole model (/root/reference/arctic_inference/vllm/swiftkv/llama_swiftkv.py:219-321, 690-712): logits of
# the LAST position of one sequence, recomputed from scratch in fp32.  Layers < n_kv are plain Llama layers over every
# token; the later layers' K / V of every token come from norm_swiftkv(residual stream after layer n_kv - 1); only the
# last token runs through the later layers (query from q_proj_swiftkv).  Rotary: neox style over the whole head.
# ---------------------------------------------------------------------------------------------------------------------
def swiftkv_llama_last_logits(w: dict, tokens, n_layers: int, n_kv: int, n_heads: int, n_kv_heads: int, head_dim: int,
                              rope_theta: float = 10000.0, eps: float = 1e-5) -> torch.Tensor:
    f = lambda name: w[name].float()
    T = len(tokens)
    pos = torch.arange(T, dtype=torch.float32)

    def rms(x, name):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * f(name)

    def rope(x, positions):                      # x [n, heads * D]
        n = x.shape[0]
        h = x.view(n, -1, head_dim)
        half = head_dim // 2
        inv = 1.0 / (rope_theta ** (torch.arange(0, half, dtype=torch.float32) / half))
        ang = positions[:, None] * inv[None, :]
        cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
        a, b = h[..., :half], h[..., half:]
        return torch.cat([a * cos - b * sin, b * cos + a * sin], dim=-1).reshape(n, -1)

    def attend(q, k, v, q_pos):                  # q [nq, Hq*D] at absolute positions q_pos; k, v [T, Hkv*D]
        G = n_heads // n_kv_heads
        qh = q.view(-1, n_heads, head_dim)
        kh = k.view(-1, n_kv_heads, head_dim).repeat_interleave(G, dim=1)
        vh = v.view(-1, n_kv_heads, head_dim).repeat_interleave(G, dim=1)
        s = torch.einsum("qhd,khd->hqk", qh, kh) * head_dim ** -0.5
        mask = torch.arange(k.shape[0])[None, :] > q_pos[:, None]
        s = s.masked_fill(mask[None], float("-inf"))
        return torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), vh).reshape(-1, n_heads * head_dim)

    def mlp(x, pre):
        return (torch.nn.functional.silu(x @ f(pre + "gate_proj.weight").T) * (x @ f(pre + "up_proj.weight").T)) @ \
            f(pre + "down_proj.weight").T

    x = f("model.embed_tokens.weight")[torch.as_tensor(tokens)]
    for i in range(n_kv):
        pre = f"model.layers.{i}."
        h = rms(x, pre + "input_layernorm.weight")
        q = rope(h @ f(pre + "self_attn.q_proj.weight").T, pos)
        k = rope(h @ f(pre + "self_attn.k_proj.weight").T, pos)
        v = h @ f(pre + "self_attn.v_proj.weight").T
        x = x + attend(q, k, v, pos.long()) @ f(pre + "self_attn.o_proj.weight").T
        x = x + mlp(rms(x, pre + "post_attention_layernorm.weight"), pre + "mlp.")
    swift = rms(x, "model.norm_swiftkv.weight")
    last = x[-1:]
    last_pos = pos[-1:]
    for i in range(n_kv, n_layers):
        pre = f"model.layers.{i}."
        k = rope(swift @ f(pre + "self_attn.k_proj_swiftkv.weight").T, pos)
        v = swift @ f(pre + "self_attn.v_proj_swiftkv.weight").T
        h = rms(last, pre + "input_layernorm.weight")
        q = rope(h @ f(pre + "self_attn.q_proj_swiftkv.weight").T, last_pos)
        last = last + attend(q, k, v, last_pos.long()) @ f(pre + "self_attn.o_proj.weight").T
        last = last + mlp(rms(last, pre + "post_attention_layernorm.weight"), pre + "mlp.")
    return (rms(last, "model.norm.weight") @ f("lm_head.weight").T)[0]
