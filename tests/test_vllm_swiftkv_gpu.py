"""GPU: LlamaSwiftKVForCausalLM (arcticinference_amd/vllm_plugin/swiftkv_model.py) inside the patched vLLM worker, through the
stand-in for vLLM 0.9.2 (tests/stubs): the plugin registers the HF config type and the model name, `get_model` builds the
class, an HF-named checkpoint goes through `load_weights` (fused q/k/v, gate/up and the SwiftKV k/v pair), and prefill +
decode steps run through GPUModelRunner.execute_model.  Checked against an fp32 from-scratch recomputation of the SwiftKV
forward (oracle/spec_oracle.swiftkv_llama_last_logits): logits of every sampled row, and the emitted tokens wherever the
oracle's top-2 margin is not within bf16 noise.  The bulk KV write (A16) and the one-launch row gather run for real here;
the later layers' decode attention goes through aic_verify_attention_ex."""
import numpy as np
import pytest
import torch

import vllm_harness as H
from oracle import spec_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
L, N_KV, HID, HQ, HKV, D, INTER, V = 4, 2, 256, 4, 2, 128, 512, 600


def _hf():
    from arcticinference_amd.swiftkv_config import LlamaSwiftKVConfig
    return LlamaSwiftKVConfig(num_hidden_layers=L, num_key_value_layers=N_KV, hidden_size=HID, num_attention_heads=HQ,
                              num_key_value_heads=HKV, head_dim=D, intermediate_size=INTER, vocab_size=V, rms_norm_eps=1e-5,
                              rope_theta=10000.0, hidden_act="silu", architectures=["LlamaSwiftKVForCausalLM"],
                              tie_word_embeddings=False)


def _checkpoint(seed=0):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *shape, std: torch.randn(*shape, generator=g) * std
    w = {"model.embed_tokens.weight": rnd(V, HID, std=1.0), "lm_head.weight": rnd(V, HID, std=0.3),
         "model.norm_swiftkv.weight": 1 + rnd(HID, std=0.1), "model.norm.weight": 1 + rnd(HID, std=0.1)}
    for i in range(L):
        p = f"model.layers.{i}."
        for name, shape in (("self_attn.q_proj", (HQ * D, HID)), ("self_attn.k_proj", (HKV * D, HID)),
                            ("self_attn.v_proj", (HKV * D, HID)), ("self_attn.o_proj", (HID, HQ * D)),
                            ("mlp.gate_proj", (INTER, HID)), ("mlp.up_proj", (INTER, HID)), ("mlp.down_proj", (HID, INTER))):
            w[p + name + ".weight"] = rnd(*shape, std=shape[1] ** -0.5)
        w[p + "input_layernorm.weight"] = 1 + rnd(HID, std=0.1)
        w[p + "post_attention_layernorm.weight"] = 1 + rnd(HID, std=0.1)
        if i >= N_KV:
            for name, shape in (("self_attn.q_proj_swiftkv", (HQ * D, HID)), ("self_attn.k_proj_swiftkv", (HKV * D, HID)),
                                ("self_attn.v_proj_swiftkv", (HKV * D, HID))):
                w[p + name + ".weight"] = rnd(*shape, std=HID ** -0.5)
    w["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.zeros(D // 2)      # HF checkpoints carry these: skipped
    return {k: v.to(torch.bfloat16) for k, v in w.items()}


def test_swiftkv_model_through_the_patched_runner(stub_vllm):
    H.load_plugin()
    from transformers import AutoConfig
    from vllm import ModelRegistry
    from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig,
                             VllmConfig, set_current_vllm_config)
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    from arcticinference_amd.vllm_plugin import step_context, swiftkv_model
    assert AutoConfig.for_model("llama_swiftkv", num_key_value_layers=3, num_hidden_layers=8).num_key_value_layers == 3
    assert ModelRegistry.resolve("LlamaSwiftKVForCausalLM") is swiftkv_model.swiftkv_classes().LlamaSwiftKVForCausalLM
    cfg = VllmConfig(model_config=ModelConfig(hf_config=_hf(), max_model_len=400, dtype=torch.bfloat16),
                     parallel_config=ParallelConfig(), scheduler_config=SchedulerConfig(max_num_seqs=8),
                     cache_config=CacheConfig(block_size=16),
                     compilation_config=CompilationConfig(level=0, cudagraph_capture_sizes=(8, 4, 2, 1)),
                     device_config=DeviceConfig(DEV))
    H.init_single_process_groups(cfg)
    runner = GPUModelRunner(cfg, torch.device(DEV))
    set_current_vllm_config(cfg)
    runner.load_model()
    model = runner.model
    assert type(model).__name__ == "LlamaSwiftKVForCausalLM" and len(model.model.layers) == L
    ck = _checkpoint()
    loaded = model.load_weights(ck.items())
    fused = [n for n in loaded if "kv_proj_swiftkv" in n or "qkv_proj" in n or "gate_up_proj" in n]
    assert len(fused) == L * 2 + (L - N_KV) and "lm_head.weight" in loaded and not any("rotary" in n for n in loaded)
    assert all(getattr(p, "shift_parallel_mode", False) for p in model.model.layers[N_KV:].parameters())
    runner.initialize_kv_cache((120, torch.bfloat16))

    seen = []
    orig_logits = model.compute_logits
    model.compute_logits = lambda hidden, sm=None: seen.append(orig_logits(hidden, sm)) or seen[-1]

    rng = np.random.default_rng(5)
    prompts = {f"r{i}": [int(t) for t in rng.integers(0, V, size=n)] for i, n in enumerate((37, 64, 5, 90))}
    sched = H.MiniScheduler(16, 400)
    for rid, p in prompts.items():
        sched.add(rid, p)
    ck32 = {k: v.float() for k, v in ck.items()}
    verify0 = step_context.calls["verify"]
    checked = agree = 0
    for step in range(6):
        so = sched.schedule()
        before = {rid: list(r["prompt"]) + list(r["out"]) for rid, r in sched.reqs.items()}
        out = runner.execute_model(so)
        emitted = sched.update(out)
        logits = seen[-1].float().cpu()
        assert logits.shape[0] == len(out.req_ids)
        for row, rid in enumerate(out.req_ids):
            want = O.swiftkv_llama_last_logits(ck32, before[rid], L, N_KV, HQ, HKV, D)
            scale = float(want.abs().max())
            assert torch.allclose(logits[row], want, atol=0.04 * scale, rtol=0), (step, rid, float((logits[row] - want).abs().max()), scale)
            top = torch.topk(want, 2).values
            checked += 1
            if float(top[0] - top[1]) > 0.04 * scale:          # outside bf16 noise: the token must be the oracle's
                assert emitted[rid] == [int(want.argmax())], (step, rid)
                agree += 1
    assert checked == 24 and agree >= 12
    # decode steps: the later layers' attention ran on the HIP kernel (their prefill-step call has max_query_len > 33 and
    # stays with the backend); the first half's layers likewise
    assert step_context.calls["verify"] - verify0 >= 5 * L
    # the selection landed in the decode runner's persistent graph buffers (batch 4 <= the largest captured size 8)
    sel = model.model._selector
    assert sel is not None and sel.inputs is not None and model.model.decode_runner.inputs is sel.inputs


def test_swiftkv_runners_trace_fullgraph(stub_vllm):
    """Both runners of the SwiftKV model carry @support_torch_compile like the reference's (llama_swiftkv.py:218, :283); the
    stand-in's decorator is a no-op, so what vLLM would do — trace each runner's forward with Dynamo, fullgraph — is done
    here by hand: prefill runner AND decode runner compiled with fullgraph=True (any graph break raises), driven through the
    patched execute_model for prefill + decode steps; logits equal the eager run's, the patched attention shows up as the
    opaque custom op in every graph (the bulk KV write sits between the runners, in swiftkv_select, as in the reference)."""
    H.load_plugin()
    from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig,
                             VllmConfig, set_current_vllm_config)
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner

    def run(compiled):
        cfg = VllmConfig(model_config=ModelConfig(hf_config=_hf(), max_model_len=400, dtype=torch.bfloat16),
                         parallel_config=ParallelConfig(), scheduler_config=SchedulerConfig(max_num_seqs=8),
                         cache_config=CacheConfig(block_size=16),
                         compilation_config=CompilationConfig(level=0, cudagraph_capture_sizes=()),
                         device_config=DeviceConfig(DEV))
        H.init_single_process_groups(cfg)
        runner = GPUModelRunner(cfg, torch.device(DEV))
        set_current_vllm_config(cfg)
        runner.load_model()
        model = runner.model
        model.load_weights(_checkpoint().items())
        runner.initialize_kv_cache((120, torch.bfloat16))
        graphs = {"prefill": [], "decode": []}
        if compiled:
            core = model.model
            for name, mod in (("prefill", core.prefill_runner), ("decode", core.decode_runner)):
                def backend(gm, example_inputs, _name=name):
                    graphs[_name].append(gm)
                    return gm.forward
                mod.forward = torch.compile(mod.forward, fullgraph=True, backend=backend, dynamic=True)
        seen = []
        orig_logits = model.compute_logits
        model.compute_logits = lambda hidden, sm=None: seen.append(orig_logits(hidden, sm)) or seen[-1]
        rng = np.random.default_rng(11)
        sched = H.MiniScheduler(16, 400)
        for i, n in enumerate((37, 64, 5)):
            sched.add(f"r{i}", [int(t) for t in rng.integers(0, V, size=n)])
        toks = []
        for _ in range(4):
            toks.append(sched.update(runner.execute_model(sched.schedule())))
        return toks, [x.float().cpu() for x in seen], graphs

    torch._dynamo.reset()
    try:
        want_toks, want_logits, _ = run(False)
        got_toks, got_logits, graphs = run(True)
    finally:
        torch._dynamo.reset()
    assert got_toks == want_toks
    for a, b in zip(want_logits, got_logits):
        assert a.shape == b.shape and torch.allclose(a, b, atol=2e-2 * float(a.abs().max()), rtol=0)
    for name in ("prefill", "decode"):
        assert graphs[name], f"the {name} runner was never traced"
        for gm in graphs[name]:
            targets = [str(n.target) for n in gm.graph.nodes if n.op == "call_function"]
            n_attn = sum(t == "arctic_inference.attention" for t in targets)
            assert n_attn == (N_KV if name == "prefill" else L - N_KV), (name, n_attn, targets)


# ---------------------------------------------------------------------------------------------------------------
# SP = 2 with shift parallelism (BASELINE config 3's arrangement): two processes on the one GPU, collectives over gloo
# ---------------------------------------------------------------------------------------------------------------
def _sp_worker(rank, world, port, out_q):
    import os
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        H.install()
        H.load_plugin()
        from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig,
                                 VllmConfig, set_current_vllm_config)
        from vllm.distributed import parallel_state as ps
        from vllm.v1.worker.gpu_model_runner import GPUModelRunner
        kw = dict(ulysses_sequence_parallel_size=world, enable_shift_parallel=True, shift_parallel_threshold=16) if world > 1 else {}
        cfg = VllmConfig(model_config=ModelConfig(hf_config=_hf(), max_model_len=400, dtype=torch.bfloat16),
                         parallel_config=ParallelConfig(**kw), scheduler_config=SchedulerConfig(max_num_seqs=8),
                         cache_config=CacheConfig(block_size=16),
                         compilation_config=CompilationConfig(level=0, cudagraph_capture_sizes=(8, 4, 2, 1)),
                         device_config=DeviceConfig(DEV))
        cfg.parallel_config.rank = rank
        ps.reset_for_tests()
        ps.init_world_group(rank)
        set_current_vllm_config(cfg)
        ps.initialize_model_parallel(1, 1)
        r = GPUModelRunner(cfg, torch.device(DEV))
        r.load_model()
        ck = _checkpoint()
        models = [r.model] + ([r.shift_model] if getattr(r, "shift_model", None) is not None else [])
        for m in models:
            m.load_weights(ck.items())
        r.initialize_kv_cache((120, torch.bfloat16))
        seen = []
        for m in models:
            orig = m.compute_logits
            m.compute_logits = (lambda o: lambda hidden, sm=None: seen.append(o(hidden, sm)) or seen[-1])(orig)
        rng = np.random.default_rng(5)
        sched = H.MiniScheduler(16, 400)
        for i, n in enumerate((37, 64, 5, 91, 20)):      # 217 prompt tokens, 5 per decode step: odd, padded to a multiple of SP
            sched.add(f"r{i}", [int(t) for t in rng.integers(0, V, size=n)])
        toks, logits = [], []
        for _ in range(5):
            so = sched.schedule()
            toks.append(sched.update(r.execute_model(so)))
            logits.append(seen[-1].float().cpu().numpy())
        shared = None
        if len(models) == 2:       # the shift replica shares the decode half with the Ulysses model (model_runner.py:767-773)
            shared = models[0].model.decode_runner is models[1].model.decode_runner
        out_q.put((world, rank, toks, logits, shared))
    except BaseException:
        import traceback
        out_q.put((world, rank, "error", traceback.format_exc(), None))
        raise
    finally:
        dist.destroy_process_group()


def test_swiftkv_model_sp2_shift_matches_single_process():
    """SwiftKV under Ulysses SP = 2 with shift parallelism: the first half runs Ulysses for the 217-token prefill step (padded) and
    the shift replica for decode steps, the C7 all-gather hands every token to the second half, which always runs in TP = 2
    (one kv head per rank): logits equal the single-process run's."""
    import socket
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

    (_, _, ref_toks, ref_logits, _), = launch(1)
    for _, rank, toks, logits, shared in launch(2):
        # a request is compared for as long as both runs fed it the same tokens: two parallel layouts of a bf16 model may
        # break a near-tie differently, after which the sequences (and logits) legitimately differ
        same = {rid: True for rid in ref_toks[0]}
        compared = 0
        for step, (a, b) in enumerate(zip(ref_logits, logits)):
            assert a.shape == b.shape
            for row, rid in enumerate(ref_toks[step]):          # rows follow the runner's request order = insertion order
                if not same[rid]:
                    continue
                scale = float(np.abs(a[row]).max())
                assert np.allclose(a[row], b[row], atol=0.05 * scale, rtol=0), (rank, step, rid, float(np.abs(a[row] - b[row]).max()))
                compared += 1
                if ref_toks[step][rid] != toks[step][rid]:
                    top = np.sort(a[row])[-2:]
                    assert top[1] - top[0] < 0.05 * scale, (rank, step, rid, "tokens differ without a near-tie")
                    same[rid] = False
        assert compared >= 15 and sum(same.values()) >= 3, (compared, same)
        assert shared is True, "the Ulysses model and the shift replica must run ONE decode half (model_runner.py:767-773)"
