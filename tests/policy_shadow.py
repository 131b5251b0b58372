"""Shadow of the reference's proposal policy for the GPU engine / plugin tests: its own token_ids_cpu rows and
num_tokens_no_spec, an oracle SuffixCache, and oracle/runner_policy_oracle.py (the restated lines of
model_runner.py:469-486, :526-744) run on them step by step — in either indexing mode."""
from typing import List, Sequence

import numpy as np

from oracle import runner_policy_oracle as RPO
from oracle.suffix_oracle import OracleSuffixCache

LSTM = -7      # placeholder id the shadow's draft model emits: "a draft-model token goes here"


class ShadowPolicy:
    def __init__(self, method: str, k: int, max_model_len: int, mode: str, enable_suffix: bool = True, depth: int = 64,
                 width: int = 0):
        self.cfg = RPO.SpecCfg(method=method, num_speculative_tokens=k if method != "suffix" else depth,
                               enable_suffix_decoding=enable_suffix or method == "suffix", suffix_cache_max_depth=depth)
        self.limit, self.double = max_model_len, mode == "reference"
        self.cache = OracleSuffixCache(depth) if self.cfg.enable_suffix_decoding else None
        self.rows = {}
        self.nts = {}
        self.width = width or max_model_len + 64

    def admit(self, req_id, prompt: Sequence[int], generated: Sequence[int] = ()) -> None:
        row = np.zeros(self.width, np.int32)
        toks = [int(x) for x in prompt] + [int(x) for x in generated]
        row[:len(toks)] = toks
        self.rows[req_id], self.nts[req_id] = row, len(toks)
        if self.cache is not None:
            if self.cache.has_cached_prompt(req_id):
                self.cache.evict_prompt(req_id)
            self.cache.cache_prompt(req_id, [int(x) for x in prompt])
            if len(generated):
                self.cache.update_response(req_id, [int(x) for x in generated])

    def step(self, req_ids: Sequence, emitted: Sequence[Sequence[int]], evict_unseen: bool = False):
        """One engine step over `req_ids` (batch order) that emitted `emitted`: commit, update the cache for the whole
        batch, propose.  Returns (drafts per request with LSTM placeholders, suffix results or None)."""
        rows = np.stack([self.rows[r] for r in req_ids])
        nts = np.asarray([self.nts[r] for r in req_ids], np.int64)
        emitted = [[int(t) for t in e] for e in emitted]
        RPO.commit_sampled(rows, nts, emitted, self.limit)
        if self.cache is not None:
            for r, e in zip(req_ids, emitted):                       # :657-673 (prompts are cached at admission here)
                if e:
                    self.cache.update_response(r, e)
            if evict_unseen:
                for r in self.cache.cached_prompt_ids():
                    if r not in req_ids:
                        self.cache.evict_prompt(r)
        drafts, results = RPO.propose_draft_token_ids(
            self.cache, lambda last, k: [[LSTM] * k for _ in last], self.cfg, list(req_ids), rows, nts, emitted,
            self.limit, double_count=self.double)
        for i, r in enumerate(req_ids):
            self.rows[r], self.nts[r] = rows[i], int(nts[i])
        return [list(d) for d in drafts], results


def check_drafts(got: List[int], want: List[int], where) -> str:
    """Compares one request's drafts with the shadow's; returns which proposer they came from."""
    if want and want[0] == LSTM:
        assert len(got) == len(want), (where, got, want)        # values are the draft model's (checked in test_gpu_kernels)
        return "lstm"
    assert got == want, (where, got, want)
    return "suffix" if want else "none"
