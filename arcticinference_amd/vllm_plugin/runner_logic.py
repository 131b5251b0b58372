"""The selection / slicing / clamp rules of GPUModelRunnerPatch's proposal code
(/root/reference/arctic_inference/vllm/model_runner.py:526-744) as vLLM-free functions, shared by the
stand-alone engine and by the vLLM patch (model_runner.py in this package)."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

MAX_SPEC_LEN = 32  # vllm.v1.sample.rejection_sampler.MAX_SPEC_LEN

# ---------------------------------------------------------------------------------------------------
# Where a request's row ends when the proposers look at it.
#
# The reference's execute_model appends the step's sampled ids to token_ids_cpu and ADVANCES num_tokens_no_spec
# (model_runner.py:469-486); propose_suffix_draft_token_ids (:696-709) and propose_arctic_draft_token_ids (:623-636)
# then take `start_idx = num_tokens_no_spec[i]` and `end_idx = start_idx + len(sampled_ids)` once more and write the
# sampled ids again at [start_idx, end_idx).  Both readings are available:
#   "reference"       the literal arithmetic: the suffix pattern ends with the sampled ids repeated, every
#                     `max_model_len - end_idx - 1` clamp is len(sampled_ids) tighter, the row is re-written;
#   "single_advance"  the row is taken as execute_model left it (end_idx = num_tokens_no_spec[i]): the algorithm of the
#                     reference's own simulator (`(prompt + response)[-max_depth:]`, simulator.py:70-90) and of the golden
#                     fixtures.
# Chosen by the speculative config key `proposal_indexing` (ArcticSpeculativeConfig / engine.SpecConfig) or, above it,
# the environment variable ARCTIC_INFERENCE_PROPOSAL_INDEXING.
#
# DEFAULT (r04): "single_advance".  The literal arithmetic is a leftover of the vLLM 0.8-era runner, whose execute_model
# did not commit the sampled ids before the proposers ran; against vLLM 0.9.2's commit loop (kept verbatim in the
# reference's execute_model, :469-486) it counts every sampled id twice, a suffix pattern that ends in a repeated token
# almost never matches, and suffix decoding contributes nothing (measured: 1.002 tokens per request-step, BENCH_r03
# `other_indexing_mode`).  The emitted tokens are the target's in either mode — only the drafts differ — so the default
# is the algorithm the reference's simulator, its published acceptance numbers and the golden fixtures describe, and
# `proposal_indexing="reference"` is the opt-in switch that reproduces the plugin's drafts, clamps and row contents
# line for line (pinned per step in tests/test_proposal_indexing.py and on the GPU in both modes).
# ---------------------------------------------------------------------------------------------------
INDEXING_REFERENCE = "reference"
INDEXING_SINGLE_ADVANCE = "single_advance"
INDEXING_MODES = (INDEXING_REFERENCE, INDEXING_SINGLE_ADVANCE)
DEFAULT_INDEXING = INDEXING_SINGLE_ADVANCE
INDEXING_ENV = "ARCTIC_INFERENCE_PROPOSAL_INDEXING"


def proposal_indexing(config=None) -> str:
    """The mode in force for a speculative config (any object with an optional `proposal_indexing` attribute)."""
    mode = os.environ.get(INDEXING_ENV) or getattr(config, "proposal_indexing", None) or DEFAULT_INDEXING
    if mode not in INDEXING_MODES:
        raise ValueError(f"proposal_indexing must be one of {INDEXING_MODES}, got {mode!r}")
    return mode


def proposal_end_index(num_tokens_no_spec: int, num_sampled: int, mode: str) -> int:
    """`end_idx` of the proposers (model_runner.py:623-624, :698-699) for a row execute_model has already advanced."""
    return int(num_tokens_no_spec) + (int(num_sampled) if mode == INDEXING_REFERENCE else 0)


def rewrite_sampled(row, start_idx: int, sampled_ids: Sequence[int], max_model_len: int) -> None:
    """The row write of propose_suffix_draft_token_ids in "reference" mode (model_runner.py:701-709): the sampled ids
    once more at [start_idx, end_idx), cut at max_model_len when the request is at the limit."""
    end_idx = start_idx + len(sampled_ids)
    if end_idx >= max_model_len:
        room = max_model_len - start_idx          # numpy slicing rules, as in the reference (a negative bound wraps)
        row[start_idx:max_model_len] = list(sampled_ids)[:room]
    else:
        row[start_idx:end_idx] = sampled_ids


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


def hip_acceptance_kind(sm) -> Optional[str]:
    """Which acceptance kernel serves a verify step with this vLLM SamplingMetadata: "greedy" (aic_rejection_greedy),
    "random" (aic_rejection_random: temperature-only rows, greedy rows mixed in) or None (vLLM's sampler +
    RejectionSampler: top-k / top-p / min-p, penalties, logit bias, min_tokens, bad words, allowed-token masks and
    logprobs live there).  Read field by field the way vLLM 0.9.2's InputBatch fills it: `logit_bias` is a list with one
    entry per request (None where unused) and is therefore never empty; `min_tokens` / `bad_words_token_ids` are dicts;
    tensors are None when no request uses them."""
    if getattr(sm, "max_num_logprobs", None) is not None or not getattr(sm, "no_penalties", True):
        return None
    if getattr(sm, "allowed_token_ids_mask", None) is not None or getattr(sm, "bad_words_token_ids", None):
        return None
    if any(b is not None and len(b) > 0 for b in (getattr(sm, "logit_bias", None) or ())):
        return None
    if getattr(sm, "min_tokens", None) or getattr(sm, "min_p", None) is not None:
        return None
    if getattr(sm, "all_greedy", False):
        return "greedy"
    if getattr(sm, "top_k", None) is not None or getattr(sm, "top_p", None) is not None:
        return None
    return "random" if getattr(sm, "temperature", None) is not None else None
