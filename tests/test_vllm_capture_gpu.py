"""GPU: (f)-2 — the patched capture_model driven for real (reference: model_runner.py:778-856, ulysses.py:297-316).

One process walks an SP = 2 + shift-parallel layout (the stand-in's VIRTUAL_WORLD: this process is rank 0, the peers
are mirrors, so the three collectives are device-local copies a HIP graph can hold).  With CompilationLevel.PIECEWISE
and full_cuda_graph the patched capture_model captures, under the patched graph_capture (TP, PP and SP_TP communicators
held), the Ulysses model for the graph sizes x SP above the shift threshold and the shift replica (TP = SP x TP, the TP
group swapped) for the sizes at or below it — attention INCLUDED: the plugin's route records aic_verify_attention_win
with the capture-time geometry and persistent metadata buffers.  Decode steps are then replayed from those graphs and must
reproduce the hidden states of the same (equally padded) steps run eagerly, within the kernel tolerance — a replay takes
the device-geometry form of the attention call, an eager step the host-partitioned one; steps vLLM would not put into a
full graph (prefill, multi-token queries) run eagerly in both."""
import numpy as np
import pytest
import torch

import vllm_harness as H

pytestmark = pytest.mark.gpu
DEV = "cuda"
HF = dict(num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=4, hidden_size=512, head_dim=128, vocab_size=2000)
SP, THRESHOLD = 2, 8


def _mirror_collectives(monkeypatch):
    """The collective seam of the product (arcticinference_amd.dist_utils) with device-local stand-ins: every peer of a
    VirtualGroup holds what this rank holds."""
    from arcticinference_amd import dist_utils
    calls = {"a2a": 0, "ag": 0}

    def all_to_all_single(recv, send, group=None):
        calls["a2a"] += 1
        recv.copy_(send)

    def all_gather_into_tensor(out, inp, group=None):
        calls["ag"] += 1
        out.view(-1, *inp.shape).copy_(inp.unsqueeze(0).expand(out.numel() // inp.numel(), *inp.shape))

    def all_reduce(t, group=None):
        t.mul_(group.size)

    monkeypatch.setattr(dist_utils, "all_to_all_single", all_to_all_single)
    monkeypatch.setattr(dist_utils, "all_gather_into_tensor", all_gather_into_tensor)
    monkeypatch.setattr(dist_utils, "all_reduce", all_reduce)
    return calls


def _runner(level, full):
    from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, HfConfig, ModelConfig, ParallelConfig,
                             SchedulerConfig, VllmConfig, set_current_vllm_config)
    from vllm.distributed import parallel_state as ps
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    cfg = VllmConfig(model_config=ModelConfig(hf_config=HfConfig(**HF), max_model_len=400, dtype=torch.bfloat16),
                     parallel_config=ParallelConfig(ulysses_sequence_parallel_size=SP, enable_shift_parallel=True,
                                                    shift_parallel_threshold=THRESHOLD),
                     scheduler_config=SchedulerConfig(max_num_seqs=32), cache_config=CacheConfig(block_size=16),
                     compilation_config=CompilationConfig(level=level, full_cuda_graph=full, cudagraph_num_of_warmups=1,
                                                          cudagraph_capture_sizes=(32, 16, 8, 4)),
                     device_config=DeviceConfig(DEV))
    ps.reset_for_tests()
    ps.VIRTUAL_WORLD = SP
    ps.init_world_group(0)
    set_current_vllm_config(cfg)
    ps.initialize_model_parallel(1, 1, backend="gloo")           # patched: SP, SP_TP over the virtual ranks
    assert ps._SP.world_size == SP and ps._SP_TP.world_size == SP and ps._TP.world_size == 1
    r = GPUModelRunner(cfg, torch.device(DEV))
    r.load_model()
    assert r.shift_model is not None and r.shift_parallel_threshold == THRESHOLD
    r.initialize_kv_cache((700, torch.bfloat16))
    r.execute_dummy_runs = True
    return r


def _drive(r, capture_hidden):
    """32 requests decode (SP steps: 32 tokens > threshold), then 24 finish and 8 decode on (shift steps).  The batches
    fill their graph sizes exactly: with mirror ranks a padded token row of THIS rank stands in for a real row of a peer
    (row 16 + t of the attention output is "rank 1's heads of token t"), and padded rows hold whatever the allocator left
    there — real peers would hand over real rows."""
    from arcticinference_amd.workload import TokenSource
    src = TokenSource(vocab_size=2000, seed=12, n_motifs=3, motif_min=8, motif_max=16, p_motif=0.9)
    sched = H.MiniScheduler(16, 400)
    for i in range(32):
        sched.add(f"r{i}", [int(x) for x in src.stream(24 + i % 4, i)])     # 816 prompt tokens: even, no padded row under SP = 2
    hs, toks, sizes = [], [], []
    from vllm import forward_context
    for step in range(12):
        if step == 6:
            for i in range(8, 32):
                sched.finish(f"r{i}")
        h0 = len(forward_context.history)
        out = r.execute_model(sched.schedule())
        toks.append(sched.update(out))
        hs.append(capture_hidden())
        sizes.append(forward_context.history[h0][0])
    return toks, hs, sizes


def test_sp_and_shift_graphs_are_captured_and_replayed_equal_to_eager(stub_vllm, monkeypatch):
    import torch.distributed as dist
    from vllm.compilation import cuda_graphs
    from vllm.config import CompilationLevel
    from vllm.distributed import parallel_state as ps
    H.load_plugin()
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: SP)
    calls = _mirror_collectives(monkeypatch)
    from arcticinference_amd.vllm_plugin import step_context

    class Hook:
        """Records the sampled rows' hidden states and pins the emitted tokens to a fixed table, so that the two runs are fed
        the same tokens whatever a last-bit difference of the two attention call forms does to a near-tie of the toy logits."""
        hidden = None
        calls = 0

        def __call__(self, hidden_states, logits):
            self.hidden = hidden_states.float().cpu()
            rows = torch.arange(logits.shape[0], device=logits.device)
            logits[rows, (rows * 17 + self.calls * 131) % logits.shape[1]] = 1e4
            self.calls += 1
            return logits

    try:
        # ---- eager reference: same layout, same padding to graph sizes, every forward eager ----
        cuda_graphs.enabled = False
        r0 = _runner(CompilationLevel.PIECEWISE, True)
        hook0 = Hook()
        r0.model.logit_hook = hook0
        want_toks, want_hs, want_sizes = _drive(r0, lambda: hook0.hidden)
        cuda_graphs.enabled = True

        # ---- graphs ----
        for k in cuda_graphs.stats:
            cuda_graphs.stats[k] = 0
        r = _runner(CompilationLevel.PIECEWISE, True)
        hook = Hook()
        r.model.logit_hook = hook
        assert r.use_cuda_graph and r.full_cuda_graph
        step_context.calls.update(verify=0, fallback=0)
        r.capture_model()
        # patched capture_model: Ulysses model for sizes x SP above the threshold, shift replica at or below it
        cap = r.dummy_runs
        base = sorted({c[0] for c in cap if c[1] == "base"})
        shift = sorted({c[0] for c in cap if c[1] == "shift"})
        assert base == [n * SP for n in (8, 16, 32)] and shift == [4, 8], (base, shift)
        assert all(c[2] == SP for c in cap if c[1] == "shift")            # captured under the swapped (SP x TP) group
        assert ps._SP_TP.captures == 1 and ps._TP.captures == 1
        assert cuda_graphs.stats["captured"] == 5                          # one graph per (model, size)
        assert step_context.calls["fallback"] == 0 and step_context.calls["verify"] == 2 * 2 * 5   # warm-up + capture, 2 layers
        replays0 = cuda_graphs.stats["replayed"]
        got_toks, got_hs, got_sizes = _drive(r, lambda: hook.hidden)
        assert got_toks == want_toks and got_sizes == want_sizes
        # steps 1-5: 32 decode tokens > threshold -> Ulysses graphs (16 tokens per rank); 6-11: 8 tokens -> shift graph (8)
        assert got_sizes[1:6] == [32] * 5 and got_sizes[6:] == [8] * 6, got_sizes
        assert cuda_graphs.stats["replayed"] - replays0 == 11
        for i, (a, b) in enumerate(zip(want_hs, got_hs)):
            assert a.shape == b.shape, i
            assert torch.allclose(a, b, atol=3e-2, rtol=3e-2), (i, float((a - b).abs().max()))
        assert calls["a2a"] > 0 and calls["ag"] > 0
    finally:
        cuda_graphs.enabled = True
        ps.VIRTUAL_WORLD = None
        ps.reset_for_tests()
