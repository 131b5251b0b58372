"""GPU: the whole hot loop (engine.HotPathEngine = the execute_model slice of SURVEY §3.3) against an
oracle-driven replay of the same seeded workload: emitted tokens must be the ground-truth stream, and the
drafts chosen each step must be what the reference policy (oracle SuffixCache + selection rule of
model_runner.py:555-566, :595-601) would choose, with the LSTM tokens taken from the engine's own drafter."""
import numpy as np
import pytest
import torch

from oracle.suffix_oracle import OracleSuffixCache

pytestmark = pytest.mark.gpu


def _build(method, with_lstm, head_size=128, per_request=False, indexing="single_advance"):
    from arcticinference_amd.engine import HotPathEngine, ModelShape, SpecConfig
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    shape = ModelShape(num_layers=2, num_q_heads=8, num_kv_heads=4, head_size=head_size, hidden_size=512, vocab_size=2000,
                       block_size=16)
    spec = SpecConfig(method=method, num_speculative_tokens=3, enable_suffix_decoding=True,
                      draft_model_per_request=per_request, proposal_indexing=indexing)
    drafter = None
    if with_lstm:
        cfg = LSTMSpeculatorConfig(vocab_size=2000, input_hidden_dim=512, inner_dim="512", emb_dim="512", proj_dim="512")
        drafter = ArcticLSTMSpeculator(cfg, max_num_seqs=4, device="cuda", quantize_lm_head=False)
        drafter.load_weights(random_lstm_weights(cfg, seed=0, std=0.05).items())
    return HotPathEngine(shape, spec, 4, 400, drafter, device="cuda", seed=0), spec


@pytest.mark.parametrize("indexing", ["single_advance", "reference"])
@pytest.mark.parametrize("method,with_lstm,head_size,per_request", [
    ("suffix", False, 128, False), ("arctic", True, 128, False), ("arctic", True, 64, False), ("arctic", True, 128, True)])
def test_engine_steps_match_oracle_policy(method, with_lstm, head_size, per_request, indexing):
    """Per step, the engine's drafts (ids, lengths, which proposer) equal the restated reference lines
    (oracle/runner_policy_oracle.py over an oracle SuffixCache) under BOTH readings of the proposal indexing
    (runner_logic.py).  per_request=False is the reference's rule: a step in which suffix decoding takes ANY request gives
    no draft-model proposal to the others (model_runner.py:616-618); True is this build's extension."""
    from arcticinference_amd.workload import TokenSource
    from policy_shadow import LSTM, ShadowPolicy, check_drafts
    eng, spec = _build(method, with_lstm, head_size, per_request, indexing)
    assert eng._indexing == indexing
    # a repetitive source (so that suffix drafts are long) with doubled tokens here and there: only a pattern that ends
    # in a repeated token can match more than one token under the reference's indexing
    src = TokenSource(vocab_size=2000, seed=3, n_motifs=3, motif_min=8, motif_max=16, p_motif=0.9)
    B, PL = 4, 96
    streams = {}
    for r in range(B):
        s = np.asarray(src.stream(PL + 260, r)).copy()
        if indexing == "reference":
            s[1::2] = s[0::2][:len(s[1::2])]        # every token doubled: the literal pattern "...a a" finds matches
        streams[r] = s
    eng.add_requests(list(range(B)), list(range(B)), [streams[r][:PL] for r in range(B)],
                     [int(streams[r][PL]) for r in range(B)])
    shadow = ShadowPolicy(method, 3, 400, indexing)
    for r in range(B):
        shadow.admit(r, streams[r][:PL], [int(streams[r][PL])])

    def truth(req, n):
        s = streams[req.req_id]
        return s[len(req.tokens):len(req.tokens) + n]

    used = {"suffix": 0, "lstm": 0, "none": 0}
    used_long = 0
    for step in range(30):
        before = [len(r.tokens) for r in eng.requests]
        drafts_before = [list(r.drafts) for r in eng.requests]
        emitted = eng.step(truth)
        for i, r in enumerate(eng.requests):
            s = streams[r.req_id]
            toks = emitted[i]
            # greedy target follows the stream: accepted drafts + one more token, all equal to the ground truth
            assert toks == [int(x) for x in s[before[i]:before[i] + len(toks)]], (step, i)
            want_acc = 0
            for d in drafts_before[i]:
                if int(s[before[i] + want_acc]) == d:
                    want_acc += 1
                else:
                    break
            assert len(toks) == want_acc + 1
        wants, results = shadow.step([r.req_id for r in eng.requests], list(emitted))
        if per_request:
            # the extension: requests suffix decoding did not take still get the draft model's k tokens, clamped per request
            for i, r in enumerate(eng.requests):
                if not wants[i]:
                    end = shadow.nts[r.req_id] + (len(emitted[i]) if indexing == "reference" else 0)
                    wants[i] = [LSTM] * max(min(3, 400 - end - 1), 0)
        for i, r in enumerate(eng.requests):
            kind = check_drafts(r.drafts, wants[i], (step, i))
            used[kind] += 1
            used_long += kind == "suffix" and len(wants[i]) > 3
            assert np.array_equal(r.tokens, shadow.rows[r.req_id][:shadow.nts[r.req_id]])
    if indexing == "single_advance":
        assert used["suffix"] > 10 and used_long > 0, (used, used_long)
    else:
        assert used["suffix"] > 0 or method == "arctic", used
    assert used["lstm"] > 0 or not with_lstm
    st = eng.stats
    assert st.emitted == sum(len(r.tokens) - 97 for r in eng.requests)
    assert st.accepted <= st.drafted and st.num_drafts > 0
    assert eng.suffix_cache._global_tree().selfcheck() == 0


@pytest.mark.parametrize("indexing", ["single_advance", "reference"])
@pytest.mark.parametrize("method,with_lstm", [("suffix", False), ("arctic", True)])
def test_engine_runs_to_the_model_length_limit(method, with_lstm, indexing):
    """The reference's tests/unit_tests/test_arctic_spec_max_len.py asks that generation up to max_model_len (and 1, 2, 3
    tokens short of it) works with both speculative methods.  Here: requests run into the limit, drafts are clamped so
    that no request ever holds more than max_model_len tokens, finished requests leave the batch (it shrinks), and every
    emitted token is the target's."""
    from arcticinference_amd.workload import TokenSource
    from policy_shadow import ShadowPolicy, check_drafts
    eng, spec = _build(method, with_lstm, indexing=indexing)
    limit = eng.max_model_len                      # 400
    src = TokenSource(vocab_size=2000, seed=8, n_motifs=3, motif_min=8, motif_max=16, p_motif=0.9)
    plens = [limit - 3, limit - 40, limit - 12, limit - 90]
    streams = {r: src.stream(limit + 40, r) for r in range(4)}
    eng.add_requests(list(range(4)), list(range(4)), [streams[r][:plens[r]] for r in range(4)],
                     [int(streams[r][plens[r]]) for r in range(4)])
    shadow = ShadowPolicy(method, 3, limit, indexing)
    for r in range(4):
        shadow.admit(r, streams[r][:plens[r]], [int(streams[r][plens[r]])])

    def truth(req, n):
        s = streams[req.req_id]
        return s[len(req.tokens):len(req.tokens) + n]

    finished = 0
    clamped = 0
    for step in range(400):
        for slot, r in enumerate(eng.requests):      # the scheduler retires a request at the length limit
            if r is not None and len(r.tokens) >= limit:
                eng.requests[slot] = None
                finished += 1
        live = [r for r in eng.requests if r is not None]
        if not live:
            break
        before = {r.req_id: len(r.tokens) for r in live}
        emitted = eng.step(truth)
        for r, toks in zip(live, emitted):
            assert len(r.tokens) <= limit, (step, r.req_id, len(r.tokens))
            s = streams[r.req_id]
            assert toks == [int(x) for x in s[before[r.req_id]:before[r.req_id] + len(toks)]]
            assert len(r.tokens) + r.num_drafts <= limit, "a draft may not reach past the last position"
        # drafts and their length clamps near the limit are the restated reference lines' (both indexing modes)
        wants, _ = shadow.step([r.req_id for r in live], list(emitted))
        for r, w in zip(live, wants):
            kind = check_drafts(r.drafts, w, (step, r.req_id))
            clamped += kind == "lstm" and len(w) < 3
    assert finished == 4
    assert clamped > 0 or not with_lstm, "the draft model's length clamp (model_runner.py:629-641) never engaged"


@pytest.mark.parametrize("method,with_lstm,indexing,phase_streams", [
    (m, w, ix, ph) for m, w in (("suffix", False), ("arctic", True)) for ix in ("single_advance", "reference")
    for ph in (("2", "0", "1") if ix == "single_advance" else ("2",))])      # stream variants under one indexing mode
def test_two_interleaved_lanes_follow_the_reference_policy_per_lane_step(method, with_lstm, indexing, phase_streams, monkeypatch):
    """begin() / finish() with two lanes (bench.py --lanes 2): lane A's host half runs after lane B's device half has been
    enqueued.  Every lane step is an engine step over that lane's requests: emitted tokens are the target's, and the
    drafts are what the restated reference policy gives when it sees the same sequence of updates (lane by lane).
    `phase_streams`: the launches ahead of a lane's attention on their own stream (2, the default), everything on one
    stream (0), the acceptance on a third stream as well (1) — the same results whichever way the launches are spread."""
    from arcticinference_amd.workload import TokenSource
    from policy_shadow import ShadowPolicy, check_drafts
    monkeypatch.setenv("AIC_ENGINE_PHASE_STREAMS", phase_streams)
    eng, spec = _build(method, with_lstm, indexing=indexing)
    assert eng._phase_streams == int(phase_streams)
    src = TokenSource(vocab_size=2000, seed=5, n_motifs=3, motif_min=8, motif_max=16, p_motif=0.9)
    B, PL = 4, 96
    streams = {r: src.stream(PL + 200, r) for r in range(B)}
    eng.add_requests(list(range(B)), list(range(B)), [streams[r][:PL] for r in range(B)], [int(streams[r][PL]) for r in range(B)])
    shadow = ShadowPolicy(method, 3, 400, indexing)
    for r in range(B):
        shadow.admit(r, streams[r][:PL], [int(streams[r][PL])])
    truth = lambda req, n: streams[req.req_id][len(req.tokens):len(req.tokens) + n]
    lanes = [[0, 2], [1, 3]]
    pending = [None, None]
    used = {"suffix": 0, "lstm": 0, "none": 0}
    for rnd in range(24):
        for l in (0, 1):
            c = pending[l]
            if c is not None:
                before = [len(r.tokens) for r in c.reqs]
                emitted = eng.finish(c)
                for r, toks, b0 in zip(c.reqs, emitted, before):
                    assert toks == [int(x) for x in streams[r.req_id][b0:b0 + len(toks)]]
                wants, _ = shadow.step([r.req_id for r in c.reqs], list(emitted))
                for r, w in zip(c.reqs, wants):
                    used[check_drafts(r.drafts, w, (rnd, l, r.req_id))] += 1
            pending[l] = eng.begin(truth, lanes[l], lane=l)
            assert [eng.requests[i] for i in lanes[l]] == pending[l].reqs
    if indexing == "single_advance":
        assert used["suffix"] > 10
    assert used["lstm"] > 0 or not with_lstm
    for c in pending:
        eng.finish(c)
    assert eng.suffix_cache._global_tree().selfcheck() == 0


@pytest.mark.parametrize("method,with_lstm", [("suffix", False), ("arctic", True)])
def test_rank_owned_prompt_trees_in_the_engine_equal_the_replicated_control(method, with_lstm):
    """HotPathEngine(suffix_owner=(rank, 2, exchange)): two engines play the two ranks of an SP = 2 control plane on this one
    GPU (a thread each; the exchange is the element-wise sum of the two ranks' result matrices, met at a barrier: what the
    all-reduce gives), a third engine is the replicated form.  Same seeded workload, requests replaced along the way: per
    step the three emit the same tokens and schedule the same drafts, and each sharded engine holds only its own slots'
    prompt trees.  (World sizes 2 and 8 across processes: tests/test_suffix_sharding_gloo.py.)"""
    import threading
    from arcticinference_amd.workload import TokenSource
    B, PL, GL, steps = 4, 96, 40, 30
    src = TokenSource(vocab_size=2000, seed=5, n_motifs=3, motif_min=8, motif_max=16, p_motif=0.9)
    streams = {r: np.asarray(src.stream(PL + GL + 80, r)) for r in range(40)}
    barrier = threading.Barrier(2)
    box = [None, None]

    def exchange_for(rank):
        def exchange(mat):
            box[rank] = mat
            barrier.wait(timeout=60)
            total = box[0] + box[1]
            barrier.wait(timeout=60)
            return total
        return exchange

    def build(owner):
        from arcticinference_amd.engine import HotPathEngine, ModelShape, SpecConfig
        from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
        shape = ModelShape(num_layers=2, num_q_heads=8, num_kv_heads=4, head_size=128, hidden_size=512, vocab_size=2000, block_size=16)
        spec = SpecConfig(method=method, num_speculative_tokens=3, enable_suffix_decoding=True)
        drafter = None
        if with_lstm:
            cfg = LSTMSpeculatorConfig(vocab_size=2000, input_hidden_dim=512, inner_dim="512", emb_dim="512", proj_dim="512")
            drafter = ArcticLSTMSpeculator(cfg, max_num_seqs=4, device="cuda", quantize_lm_head=False)
            drafter.load_weights(random_lstm_weights(cfg, seed=0, std=0.05).items())
        return HotPathEngine(shape, spec, B, 400, drafter, device="cuda", seed=0, suffix_owner=owner)

    def drive(eng, log, errors):
        try:
            next_id = [B]
            eng.add_requests(list(range(B)), list(range(B)), [streams[r][:PL] for r in range(B)], [int(streams[r][PL]) for r in range(B)])

            def truth(req, n):
                s = streams[req.req_id]
                return s[len(req.tokens):len(req.tokens) + n]

            for _ in range(steps):
                emitted = eng.step(truth)
                log.append(([list(e) for e in emitted], [list(r.drafts) for r in eng.requests]))
                for slot, r in enumerate(eng.requests):
                    if r.num_tokens - r.num_prompt >= GL:
                        rid = next_id[0]
                        next_id[0] += 1
                        eng.add_request(slot, rid, streams[rid][:PL], int(streams[rid][PL]))
        except BaseException as e:     # noqa: BLE001 - a dead rank must not leave the other at the barrier
            errors.append(e)
            barrier.abort()
            raise

    rep_log, errs = [], []
    rep = build(None)
    drive(rep, rep_log, errs)
    engines = [build((rk, 2, exchange_for(rk))) for rk in range(2)]
    logs = [[], []]
    threads = [threading.Thread(target=drive, args=(engines[rk], logs[rk], errs)) for rk in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errs, errs
    assert len(rep_log) == steps and all(len(l) == steps for l in logs)
    took_suffix = 0
    for i in range(steps):
        for rk in range(2):
            assert logs[rk][i] == rep_log[i], (i, rk, logs[rk][i], rep_log[i])
        took_suffix += sum(len(d) > 3 for d in rep_log[i][1])
    assert took_suffix > 5                                  # suffix drafts (longer than the draft model's 3) were really scheduled
    for rk in range(2):
        held = set(engines[rk].suffix_cache.cached_prompt_ids())
        want = {engines[rk].requests[s].req_id for s in range(B) if s % 2 == rk}
        assert held == want, (rk, held, want)
        st = engines[rk]._sharded.stats
        assert st["queries_owned"] * 2 == st["queries_total"] and st["exchanges"] == steps
