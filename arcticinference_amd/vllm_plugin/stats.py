"""Speculative-decoding counters that tolerate drafts longer than num_speculative_tokens (suffix drafts reach
32 tokens while k = 3): /root/reference/arctic_inference/vllm/stats.py:25-69.  The growth / padding rules
are plain functions (tested without vLLM); build_stats_patches() wraps them as ArcticPatch classes."""
from __future__ import annotations

from typing import List


def grow_for_draft(num_spec_tokens: int, accepted_per_pos: List[int], num_draft_tokens: int) -> int:
    """observe_draft prologue (stats.py:42-47): returns the new num_spec_tokens; extends the list in place."""
    if num_draft_tokens > num_spec_tokens:
        num_spec_tokens = num_draft_tokens
        accepted_per_pos.extend([0] * (num_draft_tokens - len(accepted_per_pos)))
    return num_spec_tokens


def pad_accepted_lists(lists: List[List[int]]) -> None:
    """log prologue (stats.py:60-67): zero-pad every per-position list to the longest one."""
    if not lists:
        return
    longest = max(len(x) for x in lists)
    for x in lists:
        x.extend([0] * (longest - len(x)))


def mean_accepted_draft_length(num_accepted_tokens: int, num_drafts: int) -> float:
    """The reported "mean accepted draft len": accepted draft tokens per draft (vLLM logs 1 + this)."""
    return num_accepted_tokens / num_drafts if num_drafts else 0.0


def build_stats_patches():
    from vllm.v1.spec_decode.metrics import SpecDecodingLogging, SpecDecodingStats

    from ..patching import ArcticPatch

    class SpecDecodingStatsPatch(ArcticPatch[SpecDecodingStats]):
        _orig_observe_draft = SpecDecodingStats.observe_draft

        def observe_draft(self, num_draft_tokens: int, num_accepted_tokens: int):
            self.num_spec_tokens = grow_for_draft(self.num_spec_tokens, self.num_accepted_tokens_per_pos, num_draft_tokens)
            self._orig_observe_draft(num_draft_tokens, num_accepted_tokens)

    class SpecDecodingLoggingPatch(ArcticPatch[SpecDecodingLogging]):
        _orig_log = SpecDecodingLogging.log

        def log(self, *args, **kwargs):
            if not self.num_drafts:
                return
            pad_accepted_lists(self.accepted_tokens_per_pos_lists)
            self._orig_log(*args, **kwargs)

    return [SpecDecodingStatsPatch, SpecDecodingLoggingPatch]
