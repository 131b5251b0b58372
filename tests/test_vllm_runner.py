"""CPU: the patched GPUModelRunner (arcticinference_amd/vllm_plugin/model_runner.py) driven through execute_model by a
minimal scheduler over the stand-in for vLLM 0.9.2 — in the reference's call order (_update_states, _prepare_inputs,
model, sample / accept, _update_suffix_cache, propose_draft_token_ids).  On CPU tensors the attention and acceptance
routes stay on the stand-in's implementations, so what is compared here is the control flow: the patched step must
produce exactly the stock step's tokens when the Arctic features are off, and must not break vLLM's own methods."""
import numpy as np
import pytest
import torch

import vllm_harness as H


def build_runner(device="cpu", spec=None, dtype=torch.float32, parallel=None, max_model_len=256, level=0):
    from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig,
                             VllmConfig, set_current_vllm_config)
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    cfg = VllmConfig(model_config=ModelConfig(max_model_len=max_model_len, dtype=dtype),
                     parallel_config=parallel or ParallelConfig(), scheduler_config=SchedulerConfig(max_num_seqs=8),
                     cache_config=CacheConfig(block_size=16), speculative_config=spec,
                     compilation_config=CompilationConfig(level=level), device_config=DeviceConfig(device))
    H.init_single_process_groups(cfg)
    runner = GPUModelRunner(cfg, torch.device(device))
    set_current_vllm_config(cfg)
    runner.load_model()
    runner.initialize_kv_cache((160, dtype))
    return runner


def drive(runner, prompts, steps, chunk=None, late=None, finish_at=None, sampling=None):
    """Runs `steps` engine steps; returns [(emitted tokens per request, spec tokens per request)] per step.
    `sampling`: {req_id: (temperature, seed)} for requests that sample randomly."""
    sched = H.MiniScheduler(16, runner.max_model_len, chunk=chunk)
    for rid, p in prompts.items():
        sched.add(rid, p, *(sampling or {}).get(rid, ()))
    trace = []
    for step in range(steps):
        if late and step == late[0]:
            sched.add(late[1], late[2])
        if finish_at and step == finish_at[0]:
            sched.finish(finish_at[1])
        out = runner.execute_model(sched.schedule())
        emitted = sched.update(out)
        trace.append((emitted, {r: list(sched.reqs[r]["spec"]) for r in sched.reqs}))
    return trace, sched


def prompts_for(seed, n, lo=20, hi=60, vocab=2000):
    rng = np.random.default_rng(seed)
    return {f"r{i}": rng.integers(0, vocab, size=int(rng.integers(lo, hi))).tolist() for i in range(n)}


@pytest.mark.parametrize("chunk", [None, 24])
def test_patched_step_equals_stock_step_without_arctic_features(stub_vllm, chunk):
    """No speculative config: the patched execute_model must be vLLM's.  Covers chunked prefill (sampled tokens of a
    partial prefill are discarded), a request joining and one finishing mid-run."""
    P = prompts_for(1, 3)
    late = (4, "late", prompts_for(2, 1)["r0"])
    want, _ = drive(build_runner(), P, 10, chunk=chunk, late=late, finish_at=(6, "r1"))
    H.install()
    H.load_plugin()
    got, _ = drive(build_runner(), P, 10, chunk=chunk, late=late, finish_at=(6, "r1"))
    assert got == want
    assert any(len(t) for e, _ in got for t in e.values())


def test_patched_runner_keeps_vllms_own_speculative_methods_working(stub_vllm):
    """with the plugin loaded, a stock method ("ngram") must still run — the patched 9-argument
    propose_draft_token_ids is what the patched execute_model calls, and it hands the 8 stock arguments to vLLM's."""
    from vllm.config import SpeculativeConfig
    P = prompts_for(3, 3)
    want, s0 = drive(build_runner(spec=SpeculativeConfig(method="ngram", num_speculative_tokens=2)), P, 8)
    assert s0.stats["drafts"] > 0
    H.install()
    H.load_plugin()
    from vllm.config import SpeculativeConfig
    got, s1 = drive(build_runner(spec=SpeculativeConfig(method="ngram", num_speculative_tokens=2)), P, 8)
    assert got == want and s1.stats == s0.stats


def test_patched_step_equals_stock_step_for_random_sampling_requests(stub_vllm):
    """Seeded temperature sampling next to greedy requests, with drafts to verify ("ngram"): on CPU tensors the patched
    step keeps vLLM's sampler + RejectionSampler and must consume the requests' generators exactly as the stock step does
    (same tokens).  The sampling metadata is what vLLM's InputBatch builds (logit_bias = [None] * num_reqs)."""
    from vllm.config import SpeculativeConfig
    P = prompts_for(5, 4)
    sampling = {"r1": (0.9, 11), "r3": (1.3, 12)}
    want, s0 = drive(build_runner(spec=SpeculativeConfig(method="ngram", num_speculative_tokens=2)), P, 10, sampling=sampling)
    assert s0.stats["drafts"] > 0
    H.install()
    H.load_plugin()
    from vllm.config import SpeculativeConfig
    r = build_runner(spec=SpeculativeConfig(method="ngram", num_speculative_tokens=2))
    got, s1 = drive(r, P, 10, sampling=sampling)
    assert got == want and s1.stats == s0.stats
    sm = r.input_batch.sampling_metadata
    assert not sm.all_greedy and sm.logit_bias == [None] * 4 and set(sm.generators) == {1, 3}
    greedy, _ = drive(build_runner(spec=SpeculativeConfig(method="ngram", num_speculative_tokens=2)), P, 10)
    assert greedy != got, "the random requests must actually sample"


def test_runner_construction_rules(stub_vllm):
    """model_runner.py:99-156: vLLM's constructor never sees the Arctic methods; unknown methods and suffix decoding
    next to a foreign method are refused; the suffix cache exists exactly when asked for."""
    H.load_plugin()
    from vllm.config import SpeculativeConfig
    r = build_runner(spec=SpeculativeConfig(method="suffix"))
    assert r.speculative_config.method == "suffix" and r._suffix_cache is not None and r._suffix_cache.max_depth == 64
    assert r.vllm_config.speculative_config is r.speculative_config and not hasattr(r, "drafter")
    assert type(r.rejection_sampler).__name__ == "RejectionSampler"
    assert build_runner(spec=None)._suffix_cache is None
    with pytest.raises(ValueError, match="Suffix decoding is only supported"):
        build_runner(spec=SpeculativeConfig(method="ngram", num_speculative_tokens=2, enable_suffix_decoding=True))
    from vllm.config import ParallelConfig
    pc = ParallelConfig(ulysses_sequence_parallel_size=2)
    from vllm.config import CompilationConfig, PassConfig, VllmConfig
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    cfg = VllmConfig(parallel_config=pc, compilation_config=CompilationConfig(pass_config=PassConfig(True)))
    with pytest.raises(ValueError, match="incompatible with native sequence parallelism"):
        GPUModelRunner(cfg, torch.device("cpu"))


def test_prepare_inputs_publishes_logits_indices_for_swiftkv(stub_vllm):
    H.load_plugin()
    r = build_runner()
    sched = H.MiniScheduler(16, 256)
    sched.add("a", list(range(30)))
    so = sched.schedule()
    r._update_states(so)
    meta, _, logits_indices, _, _ = r._prepare_inputs(so)
    assert all(m.swiftkv_logits_indices is logits_indices for m in meta.values())


def test_shift_mode_switch_is_reentrant_and_restores(stub_vllm):
    """set_shift_parallel_mode (model_runner.py:57-81): True swaps vLLM's TP group for SP_TP, False pins the original,
    None is a no-op; nesting restores each level."""
    H.load_plugin()
    from vllm.config import ParallelConfig
    from vllm.distributed import parallel_state as ps
    from arcticinference_amd.vllm_plugin.model_runner import is_shift_parallel_mode, set_shift_parallel_mode
    build_runner(parallel=ParallelConfig())                # single process: every group has one rank
    tp, sp_tp = ps._TP, ps._SP_TP
    assert tp is not sp_tp and sp_tp.unique_name == "sp_tp" and ps._SP.unique_name == "sp"
    with set_shift_parallel_mode(None):
        assert ps._TP is tp and not is_shift_parallel_mode()
    with set_shift_parallel_mode(True):
        assert ps._TP is sp_tp and is_shift_parallel_mode() and ps.get_tp_group() is sp_tp
        with set_shift_parallel_mode(False):
            assert ps._TP is tp and not is_shift_parallel_mode()
            with set_shift_parallel_mode(True):
                assert ps._TP is sp_tp
            assert ps._TP is tp
        assert ps._TP is sp_tp and is_shift_parallel_mode()
    assert ps._TP is tp and not is_shift_parallel_mode()
    with pytest.raises(RuntimeError):
        with set_shift_parallel_mode(True):
            raise RuntimeError("boom")
    assert ps._TP is tp and not is_shift_parallel_mode()
