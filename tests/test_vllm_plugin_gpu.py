"""GPU: the patched vLLM worker end to end on the MI355X, through the stand-in for vLLM 0.9.2 (tests/stubs).

With every patch applied, GPUModelRunner.execute_model is driven for a batch of requests and must
  * route decode / verify attention to aic_verify_attention_ex (the stand-in's torch attention only sees prefills),
  * accept with the HIP kernel (vLLM's RejectionSampler is never called for a greedy batch),
  * emit exactly the target model's greedy tokens,
  * propose, every step, the drafts of the reference policy: the oracle SuffixCache's result where its score reaches
    the bar, else (method "arctic", and only in steps where suffix decoding took nobody) the LSTM speculator's tokens
    for the hidden-state row arctic_proposer.py:133-147 selects — recomputed here from the captured hidden states.
The target is the toy model with its logits steered along a seeded ground-truth stream (so drafts get accepted)."""
import os
import socket

import numpy as np
import pytest
import torch

import vllm_harness as H

pytestmark = pytest.mark.gpu
DEV = "cuda"
HF = dict(num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=4, hidden_size=512, head_dim=128, vocab_size=2000)


def _vllm_config(spec=None, parallel=None, device=DEV, dtype=torch.bfloat16, level=0, hf=None):
    from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, HfConfig, ModelConfig, ParallelConfig,
                             SchedulerConfig, VllmConfig)
    return VllmConfig(model_config=ModelConfig(hf_config=HfConfig(**(hf or HF)), max_model_len=400, dtype=dtype),
                      parallel_config=parallel or ParallelConfig(), scheduler_config=SchedulerConfig(max_num_seqs=8),
                      cache_config=CacheConfig(block_size=16), speculative_config=spec,
                      compilation_config=CompilationConfig(level=level, cudagraph_capture_sizes=(64, 32, 16, 8, 4)),
                      device_config=DeviceConfig(device))


def _lstm_spec_config(enable_suffix=True):
    from vllm.config import ModelConfig, ParallelConfig, SpeculativeConfig

    class DraftHf:
        architectures = ["ArcticLSTMSpeculatorPreTrainedModel"]
        base_model_archs = ["ToyLlamaForCausalLM"]
        vocab_size, input_hidden_dim, inner_dim, emb_dim, proj_dim = 2000, 512, "512", "512", "512"
        n_predict = num_lookahead_tokens = 3
        tie_weights = tie_lstm_embs = True
        scale_input = True
        method = "sum_lstm"

    return SpeculativeConfig(method="arctic", num_speculative_tokens=3, enable_suffix_decoding=enable_suffix,
                             draft_model_config=ModelConfig(hf_config=DraftHf()), draft_parallel_config=ParallelConfig())


class Steered:
    """Makes the toy model's greedy token at sequence position p + 1 the ground truth's, for every sampled row."""

    def __init__(self, runner, sched, streams):
        self.runner, self.sched, self.streams = runner, sched, streams
        self.hidden = None
        runner.model.logit_hook = self

    def plan(self, so):
        rows = []
        for rid, r in self.sched.reqs.items():
            n, k = so.num_scheduled_tokens[rid], len(so.scheduled_spec_decode_tokens.get(rid, ()))
            last = r["computed"] + n - 1
            rows += [(rid, p) for p in range(last - k, last + 1)]
        self.rows = rows

    def __call__(self, hidden_states, logits):
        self.hidden = hidden_states
        assert logits.shape[0] == len(self.rows)
        tok = torch.tensor([int(self.streams[rid][p + 1]) for rid, p in self.rows], device=logits.device)
        logits[torch.arange(len(self.rows), device=logits.device), tok] = 60.0
        return logits.to(torch.bfloat16)


def _requests(n, plen, seed):
    from arcticinference_amd.workload import TokenSource
    src = TokenSource(vocab_size=2000, seed=seed, n_motifs=3, motif_min=8, motif_max=16, p_motif=0.9)
    return {f"r{i}": src.stream(plen + 260, i) for i in range(n)}


@pytest.mark.parametrize("indexing", ["single_advance", "reference"])
@pytest.mark.parametrize("method", ["suffix", "arctic"])
def test_patched_execute_model_runs_the_hip_path_and_the_reference_policy(stub_vllm, method, indexing):
    from policy_shadow import LSTM, ShadowPolicy
    H.load_plugin()
    from vllm.attention.layer import Attention
    from vllm.config import SpeculativeConfig, set_current_vllm_config
    from vllm.v1.sample.rejection_sampler import RejectionSampler
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    from arcticinference_amd.speculator import LSTMSpeculatorConfig, random_lstm_weights
    from arcticinference_amd.vllm_plugin import step_context
    spec = SpeculativeConfig(method="suffix") if method == "suffix" else _lstm_spec_config()
    spec.proposal_indexing = indexing
    cfg = _vllm_config(spec)
    H.init_single_process_groups(cfg)
    runner = GPUModelRunner(cfg, torch.device(DEV))
    set_current_vllm_config(cfg)
    runner.load_model()
    if method == "arctic":
        lcfg = LSTMSpeculatorConfig(vocab_size=2000, input_hidden_dim=512, inner_dim="512", emb_dim="512", proj_dim="512")
        runner.drafter.model.load_weights(random_lstm_weights(lcfg, seed=0, std=0.05).items())
        assert runner.drafter.input_hidden_dim == 512
    runner.initialize_kv_cache((200, torch.bfloat16))
    B, PL, limit = 4, 96, 400
    streams = _requests(B, PL, seed=3)
    if indexing == "reference":       # doubled tokens: only a pattern ending in a repeated token matches under this mode
        for rid in streams:
            s = np.asarray(streams[rid]).copy()
            s[1::2] = s[0::2][:len(s[1::2])]
            streams[rid] = s
    sched = H.MiniScheduler(16, limit)
    for rid, s in streams.items():
        sched.add(rid, [int(x) for x in s[:PL]])
    steer = Steered(runner, sched, streams)
    shadow = ShadowPolicy(method, 3, limit, indexing)
    step_context.calls.update(verify=0, fallback=0)
    Attention.calls = RejectionSampler.calls = 0
    used_suffix = used_lstm = long_drafts = 0
    for step in range(28):
        so = sched.schedule()
        steer.plan(so)
        out = runner.execute_model(so)
        n_spec = {rid: len(so.scheduled_spec_decode_tokens.get(rid, ())) for rid in sched.reqs}
        hidden = steer.hidden
        emitted = sched.update(out)
        at = 0
        rows_of = {}
        for rid in out.req_ids:
            rows_of[rid] = at
            at += n_spec[rid] + 1
        for i, rid in enumerate(out.req_ids):
            r = sched.reqs[rid]
            seq = r["prompt"] + r["out"]
            toks = emitted[rid]
            assert toks == [int(x) for x in streams[rid][len(seq) - len(toks):len(seq)]], (step, rid)   # the target's tokens
            if rid not in shadow.rows:
                shadow.admit(rid, r["prompt"])
        wants, _ = shadow.step(list(out.req_ids), [emitted[rid] for rid in out.req_ids])
        lstm_want = None
        if any(w and w[0] == LSTM for w in wants):
            k = max(len(w) for w in wants)
            idx = torch.tensor([rows_of[rid] + len(emitted[rid]) - 1 for rid in out.req_ids], device=DEV)
            last = torch.tensor([emitted[rid][-1] for rid in out.req_ids], device=DEV)
            lstm_want = runner.drafter.model.generate_proposals(last, hidden[idx], k).cpu().tolist()
        for i, rid in enumerate(out.req_ids):
            got = out.spec_token_ids[i]
            if wants[i] and wants[i][0] == LSTM:
                assert got == lstm_want[i], (step, rid)
                used_lstm += 1
            else:
                assert got == wants[i], (step, rid)
                used_suffix += bool(got)
                long_drafts += len(got) > 3
            # the runner's row equals the shadow's after the step, incl. the reference mode's re-write behind the row's end
            # (rejected drafts of the previous step may still sit further out in the runner's row)
            j = runner.input_batch.req_id_to_index[rid]
            upto = min(shadow.nts[rid] + (len(emitted[rid]) if indexing == "reference" else 0), limit)
            assert np.array_equal(runner.input_batch.token_ids_cpu[j, :upto], shadow.rows[rid][:upto]), (step, rid)
    if indexing == "single_advance":
        assert used_suffix > 10 and long_drafts > 0
    else:
        assert used_suffix > 0 or method == "arctic"
    assert method == "suffix" or used_lstm > 0
    assert sched.stats["accepted"] > 0
    # routing: 28 steps x 2 layers; only the first (prefill, 96 tokens per request) went to vLLM's attention
    assert Attention.calls == 2 and step_context.calls["fallback"] == 2 and step_context.calls["verify"] == 27 * 2
    assert RejectionSampler.calls == 0          # every verify step was accepted by aic_rejection_greedy
    assert runner._suffix_cache._global_tree().selfcheck() == 0


GPT_OSS_LIKE = dict(num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=2, hidden_size=512, head_dim=64,
                    vocab_size=2000, sliding_window=48, attention_sinks=True)


@pytest.mark.parametrize("hf", [None, GPT_OSS_LIKE], ids=["llama-like", "gpt-oss-like"])
def test_hip_attention_route_equals_the_stand_in_backend(stub_vllm, hf):
    """The same request batch through vLLM's (stand-in) attention and through the plugin's HIP route: decode-step
    outputs (last-layer hidden states of the sampled rows) agree within the kernel tolerance.  gpt-oss-like: head size 64,
    G = 8, a sliding window on the second layer and per-head sinks on both — those layers stay on the HIP route."""
    from vllm.config import SpeculativeConfig, set_current_vllm_config
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    streams = _requests(3, 80, seed=9)

    def run(patched):
        if patched:
            H.install()
            H.load_plugin()
        from vllm.config import SpeculativeConfig, set_current_vllm_config
        from vllm.v1.worker.gpu_model_runner import GPUModelRunner
        cfg = _vllm_config(SpeculativeConfig(method="ngram", num_speculative_tokens=2), hf=hf)
        H.init_single_process_groups(cfg)
        r = GPUModelRunner(cfg, torch.device(DEV))
        set_current_vllm_config(cfg)
        r.load_model()
        r.initialize_kv_cache((200, torch.bfloat16))
        sched = H.MiniScheduler(16, 400)
        for rid, s in streams.items():
            sched.add(rid, [int(x) for x in s[:80]])
        steer = Steered(r, sched, streams)
        hs = []
        for _ in range(6):
            so = sched.schedule()
            steer.plan(so)
            sched.update(r.execute_model(so))
            hs.append(steer.hidden.float().cpu())
        return hs

    a = run(False)
    from arcticinference_amd.vllm_plugin import step_context
    step_context.calls.update(verify=0, fallback=0)
    b = run(True)
    assert step_context.calls["verify"] == 10 and step_context.calls["fallback"] == 2     # only the prefill step falls back
    for x, y in zip(a, b):
        assert x.shape == y.shape
        assert torch.allclose(x, y, atol=3e-2, rtol=3e-2), (x - y).abs().max()     # two bf16 layers of residual stream


# ---------------------------------------------------------------------------------------------------------------
# SP = 2 with shift parallelism, two processes on the one GPU (collectives over gloo, staged through the host)
# ---------------------------------------------------------------------------------------------------------------
def _sp_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        H.install()
        if world > 1:
            H.load_plugin()
        from vllm.config import ParallelConfig, set_current_vllm_config
        from vllm.distributed import parallel_state as ps
        from vllm.v1.worker.gpu_model_runner import GPUModelRunner
        kw = dict(ulysses_sequence_parallel_size=world, enable_shift_parallel=True, shift_parallel_threshold=16) if world > 1 else {}
        cfg = _vllm_config(parallel=ParallelConfig(**kw))
        cfg.parallel_config.rank = rank
        ps.reset_for_tests()
        ps.init_world_group(rank)
        set_current_vllm_config(cfg)
        ps.initialize_model_parallel(1, 1)
        r = GPUModelRunner(cfg, torch.device(DEV))
        r.load_model()
        r.initialize_kv_cache((200, torch.bfloat16))
        streams = _requests(3, 64, seed=4)
        sched = H.MiniScheduler(16, 400)
        for rid, s in streams.items():
            sched.add(rid, [int(x) for x in s[:64]])
        steer = Steered(r, sched, streams)
        hs, toks = [], []
        for _ in range(6):
            so = sched.schedule()
            steer.plan(so)
            toks.append(sched.update(r.execute_model(so)))
            hs.append(steer.hidden.float().cpu().numpy())      # by value through the queue
        calls = None
        if world > 1:
            from arcticinference_amd.vllm_plugin import step_context
            calls = dict(step_context.calls)
        out_q.put((world, rank, toks, hs, calls))
    except BaseException:
        import traceback
        out_q.put((world, rank, "error", traceback.format_exc(), None))
        raise
    finally:
        dist.destroy_process_group()


def test_sp2_shift_on_the_gpu_matches_single_process():
    """Ulysses SP = 2 (HIP pack / unpack kernels, all-to-all) for the prefill step and the shift replica (TP = 2) for the
    decode steps, both attending through the HIP kernel over one KV cache: hidden states equal the single-process run."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()

    def launch(world):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ps = [ctx.Process(target=_sp_worker, args=(r, world, port, q)) for r in range(world)]
        for p in ps:
            p.start()
        res = [q.get(timeout=300) for _ in range(world)]
        for p in ps:
            p.join(60)
        for x in res:
            assert x[2] != "error", x[3]
        return res

    (_, _, ref_toks, ref_hs, _), = launch(1)
    for _, rank, toks, hs, calls in launch(2):
        assert toks == ref_toks, rank
        for a, b in zip(ref_hs, hs):
            assert np.allclose(a, b, atol=3e-2, rtol=3e-2), (rank, np.abs(a - b).max())
        # 5 decode steps x 2 layers on the HIP kernel in shift mode; the prefill (192 tokens) on vLLM's backend in SP mode
        assert calls["verify"] == 10 and calls["fallback"] == 2, calls


# ---------------------------------------------------------------------------------------------------------------
# random-sampling acceptance on the product path
# ---------------------------------------------------------------------------------------------------------------
class SoftSteered(Steered):
    """Flattens the toy model's logits and plants the stream's token (logit 9) and one distractor (logit 8) on every
    sampled row: a temperature-1 request then emits the stream's token with p ~ 0.6, so drafts are accepted AND rejected."""

    def __call__(self, hidden_states, logits):
        self.hidden = hidden_states
        lg = logits.float()
        lg = (lg - lg.mean(dim=-1, keepdim=True)) / lg.std(dim=-1, keepdim=True) * 0.1
        rows = torch.arange(len(self.rows), device=lg.device)
        tok = torch.tensor([int(self.streams[rid][p + 1]) for rid, p in self.rows], device=lg.device)
        lg[rows, (tok + 977) % lg.shape[1]] = 8.0
        lg[rows, tok] = 9.0
        self.logits = lg.to(torch.bfloat16)
        return self.logits


def test_mixed_greedy_and_random_batches_are_accepted_by_the_hip_kernel(stub_vllm):
    """Requests with a temperature (seeded) next to greedy ones, suffix drafts to verify: every verify step goes through
    aic_rejection_random (the stand-in's RejectionSampler is never reached; the bonus token is vLLM's sampler's, as in
    the reference, model_runner.py:394-411).  The emitted tokens must equal oracle.rejection_random fed the SAME draws:
    twin generators with the requests' seeds replay the calls of the step in order (sampler noise, uniforms, recovery
    noise — a request without draft tokens draws no uniforms and no recovery noise)."""
    from oracle import spec_oracle as O
    H.load_plugin()
    from vllm.config import SpeculativeConfig, set_current_vllm_config
    from vllm.v1.sample.rejection_sampler import RejectionSampler
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner, Sampler
    spec = SpeculativeConfig(method="suffix")
    spec.proposal_indexing = "single_advance"           # long drafts, so that several positions per request are verified
    cfg = _vllm_config(spec)
    H.init_single_process_groups(cfg)
    runner = GPUModelRunner(cfg, torch.device(DEV))
    set_current_vllm_config(cfg)
    runner.load_model()
    runner.initialize_kv_cache((200, torch.bfloat16))
    B, PL, V = 4, 96, 2000
    streams = _requests(B, PL, seed=6)
    temps = {"r0": None, "r1": (1.0, 101), "r2": (0.7, 102), "r3": None}
    sched = H.MiniScheduler(16, 400)
    for rid, s in streams.items():
        sched.add(rid, [int(x) for x in s[:PL]], *(temps[rid] or ()))
    steer = SoftSteered(runner, sched, streams)
    twins = {rid: torch.Generator(device=DEV).manual_seed(t[1]) for rid, t in temps.items() if t}
    RejectionSampler.calls = Sampler.calls = 0
    verify_steps = accepted_random = recovered_random = 0
    for step in range(40):
        so = sched.schedule()
        steer.plan(so)
        out = runner.execute_model(so)
        ids = list(out.req_ids)
        n = [len(so.scheduled_spec_decode_tokens.get(rid, ())) for rid in ids]
        lg = steer.logits                                   # [sum(n_i + 1), V] rows of the step, request by request
        temp = [temps[rid][0] if temps[rid] else -1.0 for rid in ids]
        # (1) the sampler on the bonus rows: greedy rows arg-max, random rows softmax(l / T) / Exp(1) noise
        bonus_rows = np.cumsum(np.asarray(n) + 1) - 1
        bonus = []
        for i, rid in enumerate(ids):
            row = lg[bonus_rows[i]].float()
            if temps[rid] is None:
                bonus.append(int(row.argmax()))
            else:
                q = torch.empty(V, dtype=torch.float32, device=DEV).exponential_(generator=twins[rid])
                bonus.append(int((torch.softmax(row / temp[i], dim=-1) / q).argmax()))
        if sum(n) == 0:
            want = [[b] for b in bonus]
        else:
            verify_steps += 1
            # (2) uniforms, (3) recovery noise — only for requests that have draft tokens
            u = np.zeros(sum(n))
            noise = torch.ones(len(ids), V)
            at = 0
            for i, rid in enumerate(ids):
                if n[i] and rid in twins:
                    u[at:at + n[i]] = torch.empty(n[i], dtype=torch.float64, device=DEV).uniform_(generator=twins[rid]).cpu().numpy()
                at += n[i]
            for i, rid in enumerate(ids):
                if n[i] and rid in twins:
                    noise[i] = torch.empty(V, dtype=torch.float32, device=DEV).exponential_(generator=twins[rid]).cpu()
            is_bonus = np.zeros(lg.shape[0], dtype=bool)
            is_bonus[bonus_rows] = True
            target = lg[torch.from_numpy(np.nonzero(~is_bonus)[0]).to(DEV)].cpu()
            drafts = [t for rid in ids for t in so.scheduled_spec_decode_tokens.get(rid, ())]
            mat = O.rejection_random(target, drafts, n, bonus, max(n), temp, u, noise)
            want = [[int(t) for t in row if t != -1] for row in mat]
            for i, rid in enumerate(ids):
                if rid in twins and n[i]:
                    acc = sum(1 for p in range(min(n[i], len(want[i]))) if want[i][p] == so.scheduled_spec_decode_tokens[rid][p])
                    accepted_random += acc
                    recovered_random += acc < n[i]
        emitted = sched.update(out)
        assert [emitted[rid] for rid in ids] == want, (step, emitted, want)
    assert verify_steps > 10 and accepted_random > 5 and recovered_random > 3, (verify_steps, accepted_random, recovered_random)
    assert RejectionSampler.calls == 0, "a temperature-only batch reached vLLM's RejectionSampler"
    assert Sampler.calls == 40                          # one sampler call per step: the bonus / only token
