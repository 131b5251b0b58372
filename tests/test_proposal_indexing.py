"""CPU: the two readings of the reference's proposal indexing (vllm_plugin/runner_logic.py: "reference" = the literal
arithmetic of model_runner.py:623-636 / :696-718 after :469-486 advanced the row; "single_advance" = the row as the step
left it), each pinned against oracle/runner_policy_oracle.py (the statement-by-statement restatement of those lines).

The patched GPUModelRunner's proposal methods are driven directly on a constructed InputBatch, with a recording fake in
place of the suffix cache (whose matcher needs the GPU) and of the draft model: what is compared per step is every
speculate call's (request, pattern, keyword arguments), the draft model's (last tokens, k), the merged drafts, and the
contents of token_ids_cpu afterwards — under both modes, incl. requests at and near max_model_len."""
import hashlib

import numpy as np
import pytest
import torch

import vllm_harness as H
from oracle import runner_policy_oracle as RPO


class Result:
    def __init__(self, token_ids, score):
        self.token_ids, self.score, self.match_len = token_ids, score, len(token_ids)


def _fake_result(req_id, pattern, max_spec_tokens, scale=1.0):
    """A deterministic stand-in for the matcher: drafts and a score derived from the query itself."""
    h = hashlib.sha256(repr((req_id, tuple(int(x) for x in pattern), int(max_spec_tokens))).encode()).digest()
    n = min(h[0] % 7, max(int(max_spec_tokens), 0))
    toks = [int(x) + 1 for x in h[1:1 + n]]
    return Result(toks, float(n) * scale * (h[8] / 255.0 + 0.4))


class RecordingCache:
    """Both call forms: the reference's per-request speculate() (the oracle drives it) and this build's
    speculate_batch() (the patched runner drives it)."""

    def __init__(self, scale=1.0):
        self.calls = []
        self.prompts = {}
        self.scale = scale

    def has_cached_prompt(self, r):
        return r in self.prompts

    def cached_prompt_ids(self):
        return list(self.prompts)

    def speculate(self, req_id, pattern, max_spec_tokens=None, max_spec_factor=1.0, max_spec_offset=0.0,
                  min_token_prob=0.1):
        self.calls.append((req_id, [int(x) for x in pattern], int(max_spec_tokens), float(max_spec_factor),
                           float(max_spec_offset), float(min_token_prob)))
        return _fake_result(req_id, pattern, max_spec_tokens, self.scale)

    def speculate_batch(self, req_ids, patterns, mst, fac, off, mpr, use_prompt):
        assert all(use_prompt)
        return [self.speculate(r, p, m, f, o, q) for r, p, m, f, o, q in zip(req_ids, patterns, mst, fac, off, mpr)]


class RecordingDrafter:
    def __init__(self):
        self.calls = []

    def __call__(self, last_tokens, k):
        self.calls.append(([int(t) for t in last_tokens], int(k)))
        return [[(int(t) * 7 + j) % 1999 for j in range(k)] for t in last_tokens]


def _runner(method, enable_suffix, mode, max_model_len):
    from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig,
                             SpeculativeConfig, VllmConfig, set_current_vllm_config)
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    from arcticinference_amd.vllm_plugin.arctic_proposer import ArcticProposer
    spec = SpeculativeConfig(method=method, num_speculative_tokens=3 if method == "arctic" else None,
                             enable_suffix_decoding=enable_suffix, proposal_indexing=mode)
    cfg = VllmConfig(model_config=ModelConfig(max_model_len=max_model_len, dtype=torch.float32),
                     parallel_config=ParallelConfig(), scheduler_config=SchedulerConfig(max_num_seqs=8),
                     cache_config=CacheConfig(block_size=16), speculative_config=spec,
                     compilation_config=CompilationConfig(level=0), device_config=DeviceConfig("cpu"))
    H.init_single_process_groups(cfg)
    r = GPUModelRunner(cfg, torch.device("cpu"))
    set_current_vllm_config(cfg)
    drafter = RecordingDrafter()
    if method == "arctic":
        class FakeProposer(ArcticProposer):
            def prepare_hidden_states(self, sample_hidden_states, sampled_token_ids, spec_decode_metadata, fused=False):
                return ("hidden", None)

            def propose(self, context_token_ids, previous_hidden_states, num_predict_tokens, hidden_index=None):
                return np.asarray(drafter(context_token_ids, num_predict_tokens), dtype=np.int64)

        r.drafter = FakeProposer(r.vllm_config)
    # under "arctic" the scores mostly stay below k = 3, so that steps without a suffix winner (draft-model steps) occur
    cache = RecordingCache(0.3 if method == "arctic" else 1.0) if (enable_suffix or method == "suffix") else None
    r._suffix_cache = cache
    return r, cache, drafter


@pytest.mark.parametrize("mode", ["reference", "single_advance"])
@pytest.mark.parametrize("method,enable_suffix", [("suffix", True), ("arctic", True), ("arctic", False)])
def test_patched_runner_proposals_equal_the_restated_reference_lines(stub_vllm, mode, method, enable_suffix):
    from vllm.v1.worker.gpu_model_runner import CachedRequestState
    from arcticinference_amd.vllm_plugin import runner_logic as RL
    H.load_plugin()
    LIMIT, B = 96, 5
    runner, cache, drafter = _runner(method, enable_suffix, mode, LIMIT)
    assert RL.proposal_indexing(runner.speculative_config) == mode
    rng = np.random.default_rng(11)
    ib = runner.input_batch
    plens = [20, 60, 88, 93, 95]
    for i in range(B):
        st = CachedRequestState(f"r{i}", rng.integers(0, 1999, plens[i]).tolist(), [], num_computed_tokens=plens[i])
        runner.requests[st.req_id] = st
        ib.add(st)
    o_rows = ib.token_ids_cpu.copy()
    o_nts = ib.num_tokens_no_spec.copy()
    o_cache = RecordingCache(cache.scale) if cache is not None else None
    o_drafter = RecordingDrafter()
    cfg = RPO.SpecCfg(method=method, num_speculative_tokens=runner.speculative_config.num_speculative_tokens,
                      enable_suffix_decoding=enable_suffix or method == "suffix")

    class SO:
        num_scheduled_tokens = {f"r{i}": 1 for i in range(B)}

    saw_limit = saw_short_k = saw_model = 0
    for step in range(40):
        sampled = []
        for i in range(B):
            room = LIMIT - int(ib.num_tokens_no_spec[i])
            n = int(rng.integers(1, 5))
            sampled.append(rng.integers(0, 1999, min(n, room)).tolist())
        if step % 3 == 2:              # a request whose sampled token was discarded (partial prefill) / not sampled
            sampled[int(rng.integers(0, B))] = []
        # the step's commit (:469-486) on both sides; the runner's arrays are what execute_model would leave
        RPO.commit_sampled(ib.token_ids_cpu, ib.num_tokens_no_spec, sampled, LIMIT)
        for i, s in enumerate(sampled):
            runner.requests[f"r{i}"].output_token_ids.extend(s)
        RPO.commit_sampled(o_rows, o_nts, sampled, LIMIT)
        # empty sampled lists under "arctic" without suffix decoding read the request's next KNOWN token: make one exist
        for i in range(B):
            st = runner.requests[f"r{i}"]
            st.num_computed_tokens = st.num_tokens - 2
        want, _ = RPO.propose_draft_token_ids(
            o_cache, o_drafter, cfg, ib.req_ids, o_rows, o_nts, sampled, LIMIT,
            next_known_token=lambda i: runner.requests[f"r{i}"].get_token_id(
                runner.requests[f"r{i}"].num_computed_tokens + 1), double_count=(mode == "reference"))
        got = runner.propose_draft_token_ids(SO, [list(s) for s in sampled], None, None, None, "sample_hidden", None, None, None)
        assert got == want, (step, mode, got, want)
        if cache is not None:
            assert cache.calls == o_cache.calls, step
        assert drafter.calls == o_drafter.calls, step
        assert np.array_equal(ib.token_ids_cpu[:B, :LIMIT], o_rows[:B, :LIMIT]), step
        assert np.array_equal(ib.num_tokens_no_spec[:B], o_nts[:B])
        saw_limit += any(int(ib.num_tokens_no_spec[i]) + (len(sampled[i]) if mode == "reference" else 0) >= LIMIT
                         and sampled[i] for i in range(B))
        saw_short_k += bool(drafter.calls and drafter.calls[-1][1] < 3)
        saw_model += bool(drafter.calls)
        done = [i for i in range(B) if int(ib.num_tokens_no_spec[i]) >= LIMIT]      # the scheduler retires these
        for j in done + ([int(rng.integers(0, B))] if step % 7 == 6 else []):
            st = CachedRequestState(f"r{j}", rng.integers(0, 1999, int(rng.integers(10, 90))).tolist(), [], 0)
            runner.requests[f"r{j}"] = st
            toks = st.prompt_token_ids
            for rows, nts in ((ib.token_ids_cpu, ib.num_tokens_no_spec), (o_rows, o_nts)):
                rows[j, :len(toks)] = toks
                nts[j] = len(toks)
    assert saw_limit > 0, "no request reached the max_model_len branches"
    if method == "arctic":
        assert saw_model > 0
        assert len(drafter.calls) > 5 and any(k < 3 for _, k in drafter.calls), "the short-k clamp never ran"
    if cache is not None:
        assert len(cache.calls) > 20


def test_the_two_modes_differ_exactly_by_the_second_count():
    """runner_logic's pieces: "reference" = end_idx one len(sampled) further, row re-written behind itself."""
    from arcticinference_amd.vllm_plugin import runner_logic as RL
    row = np.arange(100, 140, dtype=np.int32)
    sampled = [7, 8, 9]
    row[20:23] = sampled                    # the step's commit: num_tokens_no_spec = 23
    assert RL.proposal_end_index(23, 3, "single_advance") == 23 and RL.proposal_end_index(23, 3, "reference") == 26
    q1 = RL.suffix_query(row, 23, [], 40, 8, 1.0, 0.0, 0.1)
    assert q1[0] == [115, 116, 117, 118, 119, 7, 8, 9] and q1[1]["max_spec_tokens"] == 8
    RL.rewrite_sampled(row, 23, sampled, 40)
    assert row[23:26].tolist() == sampled and row[26] == 126
    q2 = RL.suffix_query(row, 26, [], 40, 8, 1.0, 0.0, 0.1)
    assert q2[0] == [118, 119, 7, 8, 9, 7, 8, 9] and q2[1]["max_spec_tokens"] == 8
    # at the limit the write is cut (model_runner.py:701-707) and nothing is proposed
    row2 = np.zeros(40, np.int32)
    RL.rewrite_sampled(row2, 38, sampled, 40)
    assert row2[38:].tolist() == [7, 8] and RL.suffix_query(row2, 41, [], 40, 8, 1.0, 0.0, 0.1) is None
    # already-speculated tokens shift the offset and shrink the budget (:716-733)
    q3 = RL.suffix_query(row, 26, [1, 2], 40, 8, 2.0, -1.0, 0.1)
    assert q3[0][-2:] == [1, 2] and q3[1]["max_spec_tokens"] == 8 and q3[1]["max_spec_offset"] == -1.0 - 2 * 3.0


def test_mode_selection(monkeypatch):
    from arcticinference_amd.engine import SpecConfig
    from arcticinference_amd.vllm_plugin import runner_logic as RL
    monkeypatch.delenv(RL.INDEXING_ENV, raising=False)
    # r04: the shipped default is the single count (what bench.py headlines); "reference" is the opt-in parity switch
    assert RL.proposal_indexing(None) == RL.DEFAULT_INDEXING == "single_advance"
    assert RL.proposal_indexing(SpecConfig()) == "single_advance"
    assert RL.proposal_indexing(SpecConfig(proposal_indexing="reference")) == "reference"
    monkeypatch.setenv(RL.INDEXING_ENV, "reference")
    assert RL.proposal_indexing(SpecConfig(proposal_indexing="single_advance")) == "reference"   # the environment wins
    monkeypatch.setenv(RL.INDEXING_ENV, "twice")
    with pytest.raises(ValueError, match="proposal_indexing"):
        RL.proposal_indexing(None)
