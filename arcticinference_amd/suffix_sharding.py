"""Rank-owned prompt trees: the suffix-decoding control plane under sequence parallelism, sharded.

The reference replicates spec-decode control on every rank: each of the SP ranks keeps every request's
prompt tree and the global tree, updates all of them with the step's accepted tokens and speculates for every request
(`_update_suffix_cache`, `propose_suffix_draft_token_ids`, /root/reference/arctic_inference/vllm/model_runner.py:657-744).
That host chain is serial with the GPU — the next step's query lengths depend on the drafts — and at SP = 8 it is 0.8 ms
of a 2.1 ms step, two thirds of the tree update being the per-request prompt trees.

Here a request's prompt tree has ONE owner (slot % world): the owner builds it, extends it with the request's accepted
tokens and asks it for drafts; every rank still feeds every request's tokens into its replica of the global tree, in the
same order (seq ids and tie order stay identical).  Each rank speculates for its own requests only — prompt tree AND
global tree, `SuffixCache.speculate`'s rule that the prompt tree wins ties is applied where both are — and the per-request
results (count, score bits, <= 32 tokens: 34 int32) are combined by ONE all-reduce (sum) of a [B, 34] int32 matrix in which
a rank's rows for requests it does not own are zero: every row has exactly one non-zero contributor, so the sum is that
contributor's bits — all ranks end the step with bit-identical drafts, the same ones the replicated form computes
(tests/test_suffix_sharding_gloo.py, world sizes 2 and 8, against a replicated cache step by step).

Per rank and step at SP = 8, B = 64: 8 prompt trees extended instead of 64 (and a new request's 4096-token prompt tree is
built once, not 8 times), 8 speculation queries instead of 64, plus one 8.7 KB all-reduce.  The collective's latency on
RCCL / a gloo control group over 8 GPUs is NOT measured (no multi-GPU box in this round): the form is opt-in
(`HotPathEngine(suffix_owner=...)`, `bench.py --rank-owned-trees`).
"""
from __future__ import annotations

from typing import Callable, Hashable, Optional, Sequence

import numpy as np

RESULT_WIDTH = 34      # n_tokens, score bits, 32 token ids


class RankOwnedSuffix:
    """`cache`: a SuffixCache (or anything with its cache_prompt / cache_prompt_async / update_response(s) / evict_prompt /
    has_cached_prompt surface).  `speculate_rows(req_ids, flat_patterns, pattern_lens, max_spec_tokens, factor, offset,
    min_prob) -> (tokens [n, cap], n_tokens [n], score [n])` for requests this rank owns (prompt tree used).
    `exchange(matrix int32 [B, RESULT_WIDTH]) -> the element-wise sum over all ranks` (an all-reduce)."""

    def __init__(self, cache, rank: int, world: int, exchange: Callable[[np.ndarray], np.ndarray],
                 speculate_rows: Optional[Callable] = None):
        assert 0 <= rank < world
        self.cache, self.rank, self.world, self.exchange = cache, rank, world, exchange
        self._speculate_rows = speculate_rows or self._speculate_with_cache
        self.stats = {"prompt_trees_built": 0, "prompt_trees_skipped": 0, "queries_owned": 0, "queries_total": 0,
                      "exchanges": 0}

    def owns(self, slot: int) -> bool:
        return int(slot) % self.world == self.rank

    # ---- admission -----------------------------------------------------------------------------------------------
    def admit(self, slot: int, req_id: Hashable, prompt: Sequence[int], generated: Sequence[int],
              old_req_id: Optional[Hashable] = None) -> None:
        """A request enters `slot` (its predecessor there, if any, leaves).  The owner builds the prompt tree (on a host
        thread where the cache can); every rank puts the tokens generated so far into the global tree, in call order."""
        c = self.cache
        if old_req_id is not None and c.has_cached_prompt(old_req_id):
            c.evict_prompt(old_req_id)                               # model_runner.py:675-678 (only the owner holds one)
        gen = [int(t) for t in generated]
        if self.owns(slot):
            self.stats["prompt_trees_built"] += 1
            if hasattr(c, "cache_prompt_async"):
                c.cache_prompt_async(req_id, prompt, gen)
                return
            c.cache_prompt(req_id, prompt)
        else:
            self.stats["prompt_trees_skipped"] += 1
        if gen:
            c.update_response(req_id, gen)

    def admit_many(self, slots: Sequence[int], req_ids: Sequence[Hashable], prompts, generated, n_threads: int = 8) -> None:
        c = self.cache
        mine = [i for i, s in enumerate(slots) if self.owns(s)]
        self.stats["prompt_trees_built"] += len(mine)
        self.stats["prompt_trees_skipped"] += len(slots) - len(mine)
        if mine:
            if hasattr(c, "cache_prompts"):
                c.cache_prompts([req_ids[i] for i in mine], [list(prompts[i]) for i in mine], n_threads=n_threads)
            else:
                for i in mine:
                    c.cache_prompt(req_ids[i], list(prompts[i]))
        for rid, g in zip(req_ids, generated):                       # every request, in list order: the global tree's order
            g = [int(t) for t in np.asarray(g).reshape(-1)]
            if g:
                c.update_response(rid, g)

    # ---- one step ----------------------------------------------------------------------------------------------------
    def update(self, req_ids: Sequence[Hashable], flat_tokens: np.ndarray, lens: np.ndarray) -> None:
        """Accepted tokens of the step: into the global tree for every request; into the prompt tree where this rank holds
        one (the cache extends a prompt tree only if it exists: the owner's)."""
        if hasattr(self.cache, "update_responses"):
            self.cache.update_responses(req_ids, flat_tokens, lens)
            return
        at = 0
        for rid, n in zip(req_ids, np.asarray(lens).tolist()):
            if n:
                self.cache.update_response(rid, [int(t) for t in flat_tokens[at:at + n]])
            at += n

    def propose(self, slots: Sequence[int], req_ids: Sequence[Hashable], flat_patterns: np.ndarray, pattern_lens: np.ndarray,
                max_spec_tokens: np.ndarray, factor: np.ndarray, offset: np.ndarray, min_prob: np.ndarray):
        """Drafts for the requests (slots[i], req_ids[i]): this rank speculates for the ones it owns, one all-reduce brings
        everyone's.  Returns (tokens int32 [n, 32], n_tokens int32 [n], score float32 [n]) — identical on every rank."""
        mat = self.local_matrix(slots, req_ids, flat_patterns, pattern_lens, max_spec_tokens, factor, offset, min_prob)
        self.stats["exchanges"] += 1
        return self.unpack(self.exchange(mat))

    @staticmethod
    def unpack(mat: np.ndarray):
        return mat[:, 2:].copy(), mat[:, 0].copy(), mat[:, 1].copy().view(np.float32)

    def local_matrix(self, slots, req_ids, flat_patterns, pattern_lens, max_spec_tokens, factor, offset, min_prob) -> np.ndarray:
        """This rank's contribution to the step's result matrix: [n, RESULT_WIDTH] int32, zero rows for requests of others."""
        n = len(req_ids)
        mat = np.zeros((n, RESULT_WIDTH), np.int32)
        mine = np.asarray([i for i, s in enumerate(slots) if self.owns(s)], dtype=np.int64)
        self.stats["queries_total"] += n
        self.stats["queries_owned"] += len(mine)
        if len(mine):
            lens = np.asarray(pattern_lens, np.int64)
            first = np.cumsum(lens) - lens
            flat = np.concatenate([flat_patterns[first[i]:first[i] + lens[i]] for i in mine]).astype(np.int32, copy=False)
            pick = lambda a: np.ascontiguousarray(np.asarray(a)[mine])
            toks, n_tok, score = self._speculate_rows([req_ids[i] for i in mine], flat, pick(pattern_lens).astype(np.int32),
                                                      pick(max_spec_tokens), pick(factor), pick(offset), pick(min_prob))
            w = min(toks.shape[1], RESULT_WIDTH - 2)
            mat[mine, 0] = n_tok
            mat[mine, 1] = np.asarray(score, np.float32).view(np.int32)
            mat[mine, 2:2 + w] = toks[:, :w]
        return mat

    def _speculate_with_cache(self, req_ids, flat, lens, mst, fac, off, mpr):
        o_tok, _, o_n, o_sc, _ = self.cache.speculate_batch_arrays(req_ids, flat, lens, mst, fac, off, mpr,
                                                                   np.ones(len(req_ids), np.int32))
        return o_tok, o_n, o_sc


def gloo_exchange(group) -> Callable[[np.ndarray], np.ndarray]:
    """All-reduce of a host int32 matrix over a gloo (CPU) process group — the control-plane collective, beside the device
    collectives of the data path (vLLM keeps such a group next to every device group: GroupCoordinator.cpu_group)."""
    import torch
    import torch.distributed as dist

    def exchange(mat: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(mat))
        dist.all_reduce(t, group=group)
        return t.numpy()
    return exchange


def device_exchange(group, device) -> Callable[[np.ndarray], np.ndarray]:
    """The same all-reduce through the device group (RCCL): staging copy in, all-reduce, copy out (synchronises)."""
    import torch
    import torch.distributed as dist

    def exchange(mat: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(mat)).to(device, non_blocking=True)
        dist.all_reduce(t, group=group)
        return t.cpu().numpy()
    return exchange

