"""ORACLE — test infrastructure only (never imported by the product package).

Plain-Python restatement of the proposal half of the reference's patched execute_model
(/root/reference/arctic_inference/vllm/model_runner.py), statement by statement, on a numpy `token_ids_cpu`
matrix and a `num_tokens_no_spec` vector:

  commit_sampled()               :469-486   sampled ids appended to the row, num_tokens_no_spec ADVANCED
  update_suffix_cache()          :657-678   prompt cached at first sight, update_response, eviction of unseen prompts
  propose_suffix()               :680-744   start_idx = num_tokens_no_spec[i] (already advanced), end_idx = start_idx +
                                            len(sampled_ids), row re-written, pattern / clamps from end_idx
  propose_arctic()               :603-655   same arithmetic; running minimum of max_spec_tokens; `return [[]] * n` when a
                                            request was emptied and suffix decoding is on
  propose_draft_token_ids()      :526-601   score >= min_score -> suffix ids, request emptied for the draft model; merge

`double_count=True` is the reference as written.  `double_count=False` changes ONE thing — end_idx = start_idx, no row
re-write — which is the build's "single_advance" mode (vllm_plugin/runner_logic.py).  The suffix cache and the drafter
are parameters (an oracle SuffixCache / a recording fake), so the same code pins patterns, keyword arguments, clamps,
row contents and merged drafts.

Parity status: vLLM is not importable here and the reference holds no fixture for these functions: this
restatement is what the reference's lines say, read as text — "parity unpinned" against a running reference.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

MAX_SPEC_LEN = 32   # vllm.v1.sample.rejection_sampler.MAX_SPEC_LEN (model_runner.py:42)


@dataclass
class SpecCfg:
    """The speculative_config fields the proposal code reads (config.py:55-62)."""
    method: str = "arctic"
    num_speculative_tokens: int = 3
    enable_suffix_decoding: bool = True
    suffix_cache_max_depth: int = 64
    suffix_max_spec_factor: float = 1.0
    suffix_max_spec_offset: float = 0.0
    suffix_min_token_prob: float = 0.1
    disable_by_batch_size: Optional[int] = 64


@dataclass
class EmptyResult:
    """SuffixSpecResult() (suffix_cache.py:24-54 defaults)."""
    token_ids: List[int] = field(default_factory=list)
    score: float = 0.0
    match_len: int = 0


def commit_sampled(token_ids_cpu: np.ndarray, num_tokens_no_spec: np.ndarray, valid_sampled: Sequence[Sequence[int]],
                   max_model_len: int) -> None:
    """model_runner.py:469-486."""
    for req_idx, sampled_ids in enumerate(valid_sampled):
        if not sampled_ids:
            continue
        start_idx = int(num_tokens_no_spec[req_idx])
        end_idx = start_idx + len(sampled_ids)
        assert end_idx <= max_model_len
        token_ids_cpu[req_idx, start_idx:end_idx] = sampled_ids
        num_tokens_no_spec[req_idx] = end_idx


def update_suffix_cache(cache, req_ids: Sequence, token_ids_cpu: np.ndarray, num_prompt_tokens: Sequence[int],
                        sampled_token_ids: Sequence[Sequence[int]]) -> None:
    """model_runner.py:657-678."""
    seen = set()
    for i, sampled_ids in enumerate(sampled_token_ids):
        req_id = req_ids[i]
        seen.add(req_id)
        if not sampled_ids:
            continue
        if not cache.has_cached_prompt(req_id):
            cache.cache_prompt(req_id, token_ids_cpu[i, :num_prompt_tokens[i]].tolist())
        cache.update_response(req_id, list(sampled_ids))
    for req_id in cache.cached_prompt_ids():
        if req_id not in seen:
            cache.evict_prompt(req_id)


def propose_suffix(cache, cfg: SpecCfg, req_ids: Sequence, token_ids_cpu: np.ndarray, num_tokens_no_spec: np.ndarray,
                   sampled_token_ids: Sequence[Sequence[int]], max_model_len: int,
                   spec_token_ids: Optional[Sequence[Sequence[int]]] = None, double_count: bool = True) -> list:
    """model_runner.py:680-744.  Returns one result per request (objects with token_ids / score)."""
    results = []
    for i, sampled_ids in enumerate(sampled_token_ids):
        spec_ids = list(spec_token_ids[i]) if spec_token_ids is not None else []
        num_sampled_ids = len(sampled_ids)
        if not num_sampled_ids:
            results.append(EmptyResult())
            continue
        req_id = req_ids[i]
        start_idx = int(num_tokens_no_spec[i])
        end_idx = start_idx + (len(sampled_ids) if double_count else 0)
        if end_idx >= max_model_len:
            results.append(EmptyResult())
            if double_count:
                token_ids_cpu[i, start_idx:max_model_len] = list(sampled_ids)[:max_model_len - start_idx]
            continue
        if double_count:
            token_ids_cpu[i, start_idx:end_idx] = sampled_ids
        size = min(end_idx, cfg.suffix_cache_max_depth)
        pattern = token_ids_cpu[i, end_idx - size:end_idx]
        pattern = pattern.tolist() + spec_ids
        if len(pattern) > cfg.suffix_cache_max_depth:
            pattern = pattern[-cfg.suffix_cache_max_depth:]
        max_spec_tokens = min(MAX_SPEC_LEN - len(spec_ids), cfg.suffix_cache_max_depth, max_model_len - end_idx - 1)
        max_spec_factor = cfg.suffix_max_spec_factor
        max_spec_offset = cfg.suffix_max_spec_offset - len(spec_ids) * (max_spec_factor + 1)
        results.append(cache.speculate(req_id, pattern, max_spec_tokens=max_spec_tokens, max_spec_factor=max_spec_factor,
                                       max_spec_offset=max_spec_offset, min_token_prob=cfg.suffix_min_token_prob))
    return results


def propose_arctic(drafter: Callable, cfg: SpecCfg, token_ids_cpu: np.ndarray, num_tokens_no_spec: np.ndarray,
                   sampled_token_ids: Sequence[Sequence[int]], max_model_len: int,
                   next_known_token: Optional[Callable[[int], int]] = None, double_count: bool = True) -> List[List[int]]:
    """model_runner.py:603-655.  `drafter(last_tokens, num_predict_tokens) -> [len(last_tokens)][k]` stands for
    self.drafter.propose (:643-647); `next_known_token(i)` for req_state.get_token_id(seq_len) (:619-621)."""
    last_tokens: List[int] = []
    max_spec_tokens = cfg.num_speculative_tokens
    for i, sampled_ids in enumerate(sampled_token_ids):
        num_sampled_ids = len(sampled_ids)
        if num_sampled_ids == 0:
            if cfg.enable_suffix_decoding:
                return [[]] * len(sampled_token_ids)
            sampled_ids = [next_known_token(i)]
        start_idx = int(num_tokens_no_spec[i])
        end_idx = start_idx + (num_sampled_ids if double_count else 0)
        max_spec_tokens = min(max_spec_tokens, max_model_len - end_idx - 1)
        if max_spec_tokens <= 0:
            continue
        if double_count:
            token_ids_cpu[i, start_idx:end_idx] = sampled_ids[-1]
            last_tokens.append(int(token_ids_cpu[i, end_idx - 1]))
        else:
            last_tokens.append(int(sampled_ids[-1]))
    if max_spec_tokens <= 0:
        return [[] for _ in sampled_token_ids]
    draft_token_ids = [list(map(int, row)) for row in drafter(last_tokens, max_spec_tokens)]
    for i, sampled_ids in enumerate(sampled_token_ids):
        if not sampled_ids:
            draft_token_ids[i] = []
    return draft_token_ids


def propose_draft_token_ids(cache, drafter: Optional[Callable], cfg: SpecCfg, req_ids: Sequence, token_ids_cpu: np.ndarray,
                            num_tokens_no_spec: np.ndarray, sampled_token_ids: Sequence[Sequence[int]], max_model_len: int,
                            next_known_token: Optional[Callable[[int], int]] = None, double_count: bool = True):
    """model_runner.py:526-601.  Returns (spec_token_ids, suffix results or None)."""
    if cfg.disable_by_batch_size and len(req_ids) > cfg.disable_by_batch_size:
        return [[] for _ in sampled_token_ids], None
    suffix_spec_token_ids = None
    results = None
    new_sampled_token_ids = list(sampled_token_ids)
    if cache is not None:
        results = propose_suffix(cache, cfg, req_ids, token_ids_cpu, num_tokens_no_spec, new_sampled_token_ids,
                                 max_model_len, double_count=double_count)
        suffix_spec_token_ids = []
        min_score = 0 if cfg.method == "suffix" else cfg.num_speculative_tokens
        for i, result in enumerate(results):
            if result.score >= min_score:
                new_sampled_token_ids[i] = []
                suffix_spec_token_ids.append(list(result.token_ids))
            else:
                suffix_spec_token_ids.append([])
    spec_token_ids = None
    if cfg.method == "suffix":
        pass
    elif cfg.method in ("arctic", "mlp_speculator"):
        spec_token_ids = propose_arctic(drafter, cfg, token_ids_cpu, num_tokens_no_spec, new_sampled_token_ids,
                                        max_model_len, next_known_token, double_count=double_count)
    if spec_token_ids is None:
        spec_token_ids = suffix_spec_token_ids
    elif suffix_spec_token_ids is not None:
        spec_token_ids = [suffix_spec_token_ids[i] or spec_token_ids[i] for i in range(len(suffix_spec_token_ids))]
    return spec_token_ids, results
