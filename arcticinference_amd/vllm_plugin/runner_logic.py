"""The selection / slicing / clamp rules of GPUModelRunnerPatch's proposal code
(/root/reference/arctic_inference/vllm/model_runner.py:526-744) as vLLM-free functions, shared by the
stand-alone engine and by the vLLM patch (model_runner.py in this package)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

MAX_SPEC_LEN = 32  # vllm.v1.sample.rejection_sampler.MAX_SPEC_LEN


def suffix_query(row: Sequence[int], end_idx: int, spec_ids: Sequence[int], max_model_len: int, max_depth: int,
                 factor: float, offset: float, min_token_prob: float) -> Optional[Tuple[List[int], dict]]:
    """(pattern, kwargs) of one request's SuffixCache.speculate call (model_runner.py:709-740), or None when
    the request is at max_model_len (:701-707)."""
    if end_idx >= max_model_len:
        return None
    size = min(end_idx, max_depth)
    pattern = list(row[end_idx - size:end_idx]) + list(spec_ids)
    if len(pattern) > max_depth:
        pattern = pattern[-max_depth:]
    max_spec_tokens = min(MAX_SPEC_LEN - len(spec_ids), max_depth, max_model_len - end_idx - 1)
    # the offset is rewritten as if the already-speculated tokens came from suffix decoding (:719-733)
    max_spec_offset = offset - len(spec_ids) * (factor + 1)
    return pattern, dict(max_spec_tokens=max_spec_tokens, max_spec_factor=factor, max_spec_offset=max_spec_offset,
                         min_token_prob=min_token_prob)


def min_suffix_score(method: str, num_speculative_tokens: int) -> int:
    """Suffix drafts replace the draft model's iff score >= this (model_runner.py:555-566)."""
    return 0 if method == "suffix" else num_speculative_tokens


def merge_proposals(suffix_ids: Optional[List[List[int]]], model_ids: Optional[List[List[int]]]) -> Optional[List[List[int]]]:
    """`suffix_spec_token_ids[i] or spec_token_ids[i]` (model_runner.py:595-601)."""
    if model_ids is None:
        return suffix_ids
    if suffix_ids is None:
        return model_ids
    return [suffix_ids[i] or model_ids[i] for i in range(len(suffix_ids))]


def arctic_max_spec_tokens(num_speculative_tokens: int, end_indices: Sequence[int], max_model_len: int) -> int:
    """The running clamp of propose_arctic_draft_token_ids (model_runner.py:629-641): one value for the batch."""
    m = num_speculative_tokens
    for end_idx in end_indices:
        m = min(m, max_model_len - end_idx - 1)
        if m <= 0:
            break
    return m
