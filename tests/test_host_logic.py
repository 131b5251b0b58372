"""CPU: host-side logic of the plugin surface (flags/config defaults, stats growth, proposal selection,
Ulysses group algebra) and the C-ABI library's export list."""
import ctypes
import re

import numpy as np
import pytest

import golden_utils as gu
from arcticinference_amd import _native
from arcticinference_amd.speculator import pad_vocab_size, padding_size
from arcticinference_amd.ulysses import (local_heads, pad_tokens_for_sp, rank_groups, sp_tp_head_slice,
                                         use_shift_model)
from arcticinference_amd.vllm_plugin import runner_logic as RL
from arcticinference_amd.vllm_plugin.config import ArcticArgs, ArcticParallelSettings, ArcticSpeculativeSettings
from arcticinference_amd.vllm_plugin.stats import grow_for_draft, mean_accepted_draft_length, pad_accepted_lists


def test_library_exports_every_declared_symbol():
    import os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "arctic_hip.h")).read()
    declared = set(re.findall(r"\b(aic_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    L = _native.lib()
    assert L.aic_version() >= 100 and L.aic_device_count() >= 0


def test_compute_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from arcticinference_amd.suffix_cache import SuffixCache
    c = SuffixCache(8)
    c.cache_prompt("a", [1, 2, 3])
    with pytest.raises(_native.NativeError, match="no HIP device"):
        c.speculate("a", [1, 2])
    from arcticinference_amd import ops
    with pytest.raises(RuntimeError):
        ops.reshape_and_cache_flash_bulk(torch.zeros(1, 32), torch.zeros(1, 32), [torch.zeros(1, 16, 2, 16)],
                                         [torch.zeros(1, 16, 2, 16)], torch.zeros(1, dtype=torch.int64), "auto",
                                         [torch.ones(1)], [torch.ones(1)], 2, 16)


def test_flag_and_config_defaults():
    a = ArcticArgs()
    assert (a.ulysses_sequence_parallel_size, a.enable_shift_parallel, a.shift_parallel_threshold) == (1, False, 512)
    with pytest.raises(ValueError):
        ArcticParallelSettings(enable_shift_parallel=True)
    p = ArcticParallelSettings(ulysses_sequence_parallel_size=4, tensor_parallel_size=2)
    assert p.world_size == 8 and p.distributed_executor_backend(None) == "mp" and p.distributed_executor_backend("ray") == "ray"
    s = ArcticSpeculativeSettings()
    assert (s.enable_suffix_decoding, s.suffix_cache_max_depth, s.suffix_max_spec_factor, s.suffix_max_spec_offset,
            s.suffix_min_token_prob) == (False, 64, 1.0, 0.0, 0.1)
    assert ArcticSpeculativeSettings(method="arctic", num_speculative_tokens=3).disable_by_batch_size == 64
    sfx = ArcticSpeculativeSettings(method="suffix", suffix_cache_max_depth=48)
    assert sfx.num_speculative_tokens == 48 and sfx.enable_suffix_decoding and sfx.disable_by_batch_size == 64
    assert ArcticSpeculativeSettings(enable_suffix_decoding=True).method == "suffix"


def test_stats_growth_and_padding():
    per_pos = [0, 0, 0]
    n = grow_for_draft(3, per_pos, 7)
    assert n == 7 and per_pos == [0] * 7
    assert grow_for_draft(7, per_pos, 2) == 7 and len(per_pos) == 7
    lists = [[1, 2], [3], [4, 5, 6]]
    pad_accepted_lists(lists)
    assert lists == [[1, 2, 0], [3, 0, 0], [4, 5, 6]]
    assert mean_accepted_draft_length(12, 8) == 1.5 and mean_accepted_draft_length(0, 0) == 0.0


def test_padding_size_matches_library():
    L = _native.lib()
    for n in list(range(1, 130)) + [255, 256, 257, 1000]:
        assert padding_size(n) == L.aic_lstm_padding_size(n), n
    assert [padding_size(n) for n in (1, 2, 3, 5, 9, 17, 33, 64)] == [1, 2, 3, 6, 12, 24, 48, 64]
    assert pad_vocab_size(128256) == 128256 and pad_vocab_size(128257) == 128320


def test_suffix_query_reproduces_runner_call_pattern():
    """The (pattern, kwargs) of every speculate call recorded in the golden SuffixCache fixture were produced
    by the reference's call pattern (gen_golden.runner_call); suffix_query must rebuild them from the row."""
    checked = 0
    for case in gu.load("suffix_cache.json"):
        cfg = case["cfg"]
        rows = {}
        for ev in case["events"]:
            if ev[0] == "cache_prompt":
                rows[ev[1]] = list(ev[2])
            elif ev[0] == "update":
                rows[ev[1]].extend(ev[2])
            elif ev[0] == "speculate":
                _, rid, pattern, kw, _ = ev
                row = rows[rid]
                # the fixture appended 0-3 already-speculated ids after the row's tail
                for n_spec in range(0, 4):
                    q = RL.suffix_query(row, len(row), pattern[len(pattern) - n_spec:] if n_spec else [], 400,
                                        cfg["suffix_cache_max_depth"], cfg["suffix_max_spec_factor"],
                                        cfg["suffix_max_spec_offset"], cfg["suffix_min_token_prob"])
                    if q is not None and q[0] == pattern and q[1] == kw:
                        checked += 1
                        break
                else:
                    raise AssertionError(f"could not rebuild query for {rid}")
    assert checked > 100
    assert RL.suffix_query([1, 2, 3], 3, [], 3, 64, 1.0, 0.0, 0.1) is None      # at max_model_len


def test_selection_rules():
    assert RL.min_suffix_score("suffix", 3) == 0 and RL.min_suffix_score("arctic", 3) == 3
    assert RL.merge_proposals([[1, 2], []], [[7, 8, 9], [4, 5, 6]]) == [[1, 2], [4, 5, 6]]
    assert RL.merge_proposals(None, [[1]]) == [[1]] and RL.merge_proposals([[2]], None) == [[2]]
    assert RL.arctic_max_spec_tokens(3, [10, 20], 100) == 3
    assert RL.arctic_max_spec_tokens(3, [10, 97], 100) == 2 and RL.arctic_max_spec_tokens(3, [99], 100) == 0


def test_rank_groups_match_the_reference_layout():
    """TP=2, SP=4 -> SP groups stride TP, TP groups contiguous, SP_TP lists ranks TP-major/SP-minor."""
    g = rank_groups(8, 1, 1, 4, 2)
    assert g["TP"] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert g["SP"] == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert g["SP_TP"] == [[0, 2, 4, 6, 1, 3, 5, 7]]
    # torch restatement of ulysses.py:160-234 for a bigger layout
    import torch
    dp, pp, sp, tp = 2, 2, 2, 2
    ws = 2 * dp * pp * sp * tp
    all_ranks = torch.arange(ws).reshape(-1, dp, pp, sp, tp)
    as_set = lambda groups: sorted(tuple(x) for x in groups)
    got = rank_groups(ws, dp, pp, sp, tp)
    assert as_set(got["TP"]) == as_set(all_ranks.view(-1, tp).tolist())
    assert as_set(got["PP"]) == as_set(all_ranks.transpose(2, 4).reshape(-1, pp).tolist())
    assert as_set(got["DP"]) == as_set(all_ranks.transpose(1, 4).reshape(-1, dp).tolist())
    assert as_set(got["EP"]) == as_set(all_ranks.transpose(1, 3).reshape(-1, dp * tp).tolist())
    assert as_set(got["SP"]) == as_set(all_ranks.transpose(3, 4).reshape(-1, sp).tolist())
    assert as_set(got["SP_TP"]) == as_set(all_ranks.transpose(3, 4).reshape(-1, sp * tp).tolist())
    # KV-replicated split of SP=8 with 2 kv heads (ulysses.py:251-281)
    g8 = rank_groups(8, 1, 1, 8, 1, num_kv_heads=2)
    ar = torch.arange(8).reshape(-1, 1, 1, 2, 4, 1)
    assert as_set(g8["SP_AA"]) == as_set(ar.transpose(3, 5).reshape(-1, 2).tolist())
    assert as_set(g8["SP_AG"]) == as_set(ar.transpose(4, 5).reshape(-1, 4).tolist())


def test_head_partition_and_shift_rules():
    assert local_heads(32, 8, 8) == local_heads(32, 8, 4, tp=2)
    lh = local_heads(32, 8, 8)
    assert (lh.num_q_heads, lh.num_kv_heads, lh.kv_replicated) == (4, 1, False)
    assert local_heads(64, 8, 16).kv_replicated
    assert use_shift_model(256, 8, True) and not use_shift_model(513, 8, True) and not use_shift_model(10, 1, True)
    assert not use_shift_model(10, 8, False)
    assert pad_tokens_for_sp(13, 8) == 16 and pad_tokens_for_sp(16, 8) == 16
    # KV-cache invariance: SP layout and shift (TP over SP_TP) layout own the same head slice
    owned = sorted(sp_tp_head_slice(32, 4, 2, s, t) for s in range(4) for t in range(2))
    assert owned == [(i * 4, i * 4 + 4) for i in range(8)]
    assert sp_tp_head_slice(32, 4, 2, sp_rank=1, tp_rank=1) == (20, 24)


def test_bench_starts_its_own_ranks(monkeypatch):
    """`bench.py --gpus N` without a launcher must start N ranks as a child torch.distributed.run (never an exec, and
    before anything touches the GPU) and exit with the child's code."""
    import subprocess
    import sys

    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before the spawn")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_lstm_family_speculator_by_method():
    """`method` of an ArcticLSTMSpeculator checkpoint (arctic_speculator.py:441): sum_lstm and sum_rnn build (stacked
    sum_rnn stages included), unequal widths and unknown methods are refused with a message.  (Construction only: no device is touched.)"""
    from arcticinference_amd.speculator import (ArcticLSTMSpeculator, ArcticSumRNNSpeculator, LSTMSpeculatorConfig,
                                                lstm_family_speculator)
    base = dict(vocab_size=1000, input_hidden_dim=768, n_predict=3, num_lookahead_tokens=3)
    m = lstm_family_speculator(LSTMSpeculatorConfig(inner_dim="512", emb_dim="512", proj_dim="512", method="sum_rnn",
                                                    tie_weights=False, **base))
    assert isinstance(m, ArcticSumRNNSpeculator) and m.inner_dim == 512 and m.input_hidden_dim == 768 and not m.tie_weights
    assert type(lstm_family_speculator(LSTMSpeculatorConfig(inner_dim="512", emb_dim="512", proj_dim="512", **base))) \
        is ArcticLSTMSpeculator
    st = lstm_family_speculator(LSTMSpeculatorConfig(inner_dim="512.512", emb_dim="512", proj_dim="512.512.512", method="sum_rnn",
                                                     **base))
    assert isinstance(st, ArcticSumRNNSpeculator) and st.stacks == (0, 2, 1)      # extra stages per emb / proj / ln stack
    with pytest.raises(NotImplementedError, match="three extra stages"):
        lstm_family_speculator(LSTMSpeculatorConfig(inner_dim="512.512.512.512.512", emb_dim="512", proj_dim="512",
                                                    method="sum_rnn", **base))
    with pytest.raises(ValueError, match="must be equal"):
        lstm_family_speculator(LSTMSpeculatorConfig(inner_dim="512.256", emb_dim="512", proj_dim="512", method="sum_rnn", **base))
    with pytest.raises(ValueError, match="must be equal"):
        lstm_family_speculator(LSTMSpeculatorConfig(inner_dim="512", emb_dim="256", proj_dim="512", method="sum_rnn", **base))
    with pytest.raises(ValueError, match="unknown speculator method"):
        lstm_family_speculator(LSTMSpeculatorConfig(inner_dim="512", emb_dim="512", proj_dim="512", method="sum_gru", **base))


def test_hip_acceptance_gate_reads_vllm_sampling_metadata_field_by_field():
    """vLLM 0.9.2's InputBatch always passes logit_bias = [None] * num_reqs (a non-empty list) — the gate
    must look inside it; min_tokens and min_p keep a batch on vLLM's sampler; temperature-only batches are "random"."""
    from types import SimpleNamespace as NS
    import torch
    from arcticinference_amd.vllm_plugin.runner_logic import hip_acceptance_kind

    def sm(**kw):
        base = dict(temperature=None, all_greedy=True, all_random=False, top_p=None, top_k=None, min_p=None, generators={},
                    max_num_logprobs=None, no_penalties=True, min_tokens={}, logit_bias=[None, None, None],
                    allowed_token_ids_mask=None, bad_words_token_ids={})
        base.update(kw)
        return NS(**base)

    assert hip_acceptance_kind(sm()) == "greedy"                                  # the list of Nones is "no bias"
    assert hip_acceptance_kind(sm(logit_bias=[None, {7: 2.0}, None])) is None
    assert hip_acceptance_kind(sm(logit_bias=[None, {}, None])) == "greedy"
    assert hip_acceptance_kind(sm(min_tokens={1: (5, {2})})) is None              # EOS must stay masked until min_tokens
    assert hip_acceptance_kind(sm(min_p=torch.tensor([0.0, 0.1, 0.0]))) is None
    assert hip_acceptance_kind(sm(max_num_logprobs=0)) is None
    assert hip_acceptance_kind(sm(no_penalties=False)) is None
    assert hip_acceptance_kind(sm(bad_words_token_ids={0: [[3]]})) is None
    assert hip_acceptance_kind(sm(allowed_token_ids_mask=torch.zeros(3, 8, dtype=torch.bool))) is None
    t = torch.tensor([-1.0, 0.8, 1.0])
    assert hip_acceptance_kind(sm(all_greedy=False, temperature=t)) == "random"   # greedy rows mixed in
    assert hip_acceptance_kind(sm(all_greedy=False, all_random=True, temperature=t.abs())) == "random"
    assert hip_acceptance_kind(sm(all_greedy=False, temperature=t, top_k=torch.tensor([0, 40, 0]))) is None
    assert hip_acceptance_kind(sm(all_greedy=False, temperature=t, top_p=torch.tensor([1.0, 0.9, 1.0]))) is None
    assert hip_acceptance_kind(sm(all_greedy=False, temperature=None)) is None
