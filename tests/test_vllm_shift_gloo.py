"""CPU, world_size 2 and 4 over gloo: the patched vLLM worker under Ulysses SP with shift parallelism
(A13 ulysses_forward slice + all-gather, A14 the SP / SP_TP / SP_AA / SP_AG groups made by the patched
initialize_model_parallel, A15 set_shift_parallel_mode + the shift branches of load_model / execute_model /
initialize_kv_cache / capture_model / profile_run), driven through the stand-in for vLLM 0.9.2.

Every rank runs the whole patched runner.  Steps above the threshold run the Ulysses model (token slice per rank,
all-to-all around attention, hidden states all-gathered); steps at or below it run the TP = SP x TP replica built by
load_model, which attends over THE SAME KV cache tensors.  The sampled tokens of every rank must equal those of a
single-process stock runner on the same seeded toy model, across a sequence of steps that switches modes — which only
holds if both layouts own the same heads of the same cache (KV-cache invariance)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _prompts():
    rng = np.random.default_rng(5)
    return {f"r{i}": rng.integers(0, 2000, size=int(rng.integers(24, 50))).tolist() for i in range(3)}


def _configs(sp, hkv, threshold, shift=True):
    from vllm.config import (CacheConfig, CompilationConfig, CompilationLevel, DeviceConfig, HfConfig, ModelConfig,
                             ParallelConfig, SchedulerConfig, VllmConfig)
    kw = dict(ulysses_sequence_parallel_size=sp, enable_shift_parallel=shift, shift_parallel_threshold=threshold) if sp > 1 else {}
    return VllmConfig(model_config=ModelConfig(hf_config=HfConfig(num_key_value_heads=hkv), max_model_len=256, dtype=torch.float32),
                      parallel_config=ParallelConfig(**kw), scheduler_config=SchedulerConfig(max_num_seqs=8),
                      cache_config=CacheConfig(block_size=16),
                      compilation_config=CompilationConfig(level=CompilationLevel.PIECEWISE, cudagraph_capture_sizes=(32, 16, 8, 4)),
                      device_config=DeviceConfig("cpu"))


def _steps(runner, n_steps):
    import vllm_harness as H
    sched = H.MiniScheduler(16, 256)
    for rid, p in _prompts().items():
        sched.add(rid, p)
    trace = []
    for step in range(n_steps):
        if step == 5:
            sched.add("late", _prompts()["r0"][:40])         # a prefill in the middle: back to the Ulysses model
        out = runner.execute_model(sched.schedule())
        trace.append(sched.update(out))
    return trace


def _single_process_reference(hkv, n_steps, out_q):
    import vllm_harness as H
    H.install()
    from vllm.config import set_current_vllm_config
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    cfg = _configs(1, hkv, 0)
    H.init_single_process_groups(cfg)
    r = GPUModelRunner(cfg, torch.device("cpu"))
    set_current_vllm_config(cfg)
    r.load_model()
    r.initialize_kv_cache((160, torch.float32))
    out_q.put(("ref", _steps(r, n_steps)))
    dist.destroy_process_group()


def _worker(rank, world, port, hkv, threshold, n_steps, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _worker_body(rank, world, hkv, threshold, n_steps, out_q)
    except BaseException:
        import traceback
        out_q.put((rank, "error", traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _worker_body(rank, world, hkv, threshold, n_steps, out_q):
    if True:
        import vllm_harness as H
        from oracle import spec_oracle as O
        H.install()
        H.load_plugin()
        import arcticinference_amd.ulysses as U
        # the HIP copy kernels cannot run here: the literal torch expressions of ulysses.py:493-517 stand in
        U.PACK_FNS[0] = lambda q, k, v, sp: O.ulysses_pack(q, k, v, sp, q.shape[1] // sp // 64, k.shape[1] // sp // 64, 64)
        U.PACK_FNS[1] = lambda c, sp: O.ulysses_unpack(c, sp, c.shape[1] // 64, 64)
        U.PACK_FNS[2], U.PACK_FNS[3] = O.ulysses_pack_pair, O.ulysses_reorder_split
        from vllm.config import set_current_vllm_config
        from vllm.distributed import parallel_state as ps
        from vllm.forward_context import history
        from vllm.model_executor import model_loader
        from vllm.v1.worker.gpu_model_runner import GPUModelRunner
        from arcticinference_amd.ulysses import rank_groups
        cfg = _configs(world, hkv, threshold)
        cfg.parallel_config.rank = rank
        ps.init_world_group(rank)
        set_current_vllm_config(cfg)
        ps.initialize_model_parallel(1, 1)                       # patched: creates SP, SP_TP (and SP_AA / SP_AG)
        want = rank_groups(world, 1, 1, world, 1, num_kv_heads=hkv)
        mine = lambda kind: next(g for g in want[kind] if rank in g)
        assert ps._SP.ranks == mine("SP") and ps._SP_TP.ranks == mine("SP_TP") and ps._TP.ranks == [rank]
        assert ps._SP.rank_in_group == rank and ps._TP.use_message_queue_broadcaster
        if hkv < world:
            assert ps._SP_AA.ranks == mine("SP_AA") and ps._SP_AG.ranks == mine("SP_AG")
        else:
            assert ps._SP_AA is None and ps._SP_AG is None
        r = GPUModelRunner(cfg, torch.device("cpu"))
        assert r.use_ulysses
        r.load_model()
        # two replicas: the Ulysses model over TP = 1, the shift model over TP = SP x TP; vLLM's TP group is back
        assert model_loader.loaded == [("ToyLlamaForCausalLM", 1), ("ToyLlamaForCausalLM", world)]
        assert r.shift_model is not None and r.shift_parallel_threshold == threshold and ps._TP.ranks == [rank]
        assert r.model.forward.__name__ == "ulysses_forward"
        r.initialize_kv_cache((160, torch.float32))
        base_attn = [m for m in r.model.modules() if type(m).__name__ == "Attention"]
        shift_attn = [m for m in r.shift_model.modules() if type(m).__name__ == "Attention"]
        assert len(base_attn) == len(shift_attn) == 2
        for a, b in zip(base_attn, shift_attn):
            assert a.kv_cache[0].data_ptr() == b.kv_cache[0].data_ptr()        # one cache, two models
            assert (a.num_heads, a.num_kv_heads) == (b.num_heads, b.num_kv_heads) == (8 // world, max(1, hkv // world))
        r.profile_run()
        r.capture_model()
        # profile: both replicas at max tokens; capture: Ulysses model for sizes x SP above the threshold (TP group of 1),
        # shift model for sizes at or below it, under the full-TP group
        assert r.dummy_runs[:2] == [(1024, "base", 1, True), (1024, "shift", world, True)]
        cap = r.dummy_runs[2:]
        base = [c for c in cap if c[1] == "base"]
        shift = [c for c in cap if c[1] == "shift"]
        assert {c[0] for c in base} == {n * world for n in (32, 16, 8, 4) if threshold < n * world <= 1024}
        assert {c[0] for c in shift} == {n for n in (32, 16, 8, 4) if n <= threshold} and all(c[2] == world for c in shift)
        assert all(len([c for c in cap if c[:2] == k]) == 2 for k in {c[:2] for c in cap})   # one warm-up + the capture
        assert ps._SP_TP.captures == 1 and ps._TP.captures == 1
        h0 = len(history)
        trace = _steps(r, n_steps)
        sizes = [h[0] for h in history[h0:]]
        out_q.put((rank, trace, sizes))
        from vllm.v1.executor.multiproc_executor import WorkerProc
        w = WorkerProc()
        w.shutdown()
        assert ps._SP is None and ps._SP_TP is None


@pytest.mark.parametrize("world,hkv", [(2, 4), (4, 2)])
def test_sp_and_shift_steps_reproduce_the_single_process_tokens(world, hkv):
    """(4, 2): fewer kv heads than SP ranks — the KV-replicated variant (SP_AA x SP_AG)."""
    threshold, n_steps = 16, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ref = ctx.Process(target=_single_process_reference, args=(hkv, n_steps, q))
    ref.start()
    tag, want = q.get(timeout=120)
    ref.join(60)
    assert tag == "ref"
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, hkv, threshold, n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, trace, sizes in got:
        assert trace != "error", sizes
        assert trace == want, rank
        # the first step and step 5 are prefills (Ulysses model: tokens padded to a multiple of SP and, being graph-sized
        # per rank, to a graph size x SP); the decode steps of 3-4 tokens run the shift model (graph size 4)
        assert sizes[0] % world == 0 and sizes[0] > threshold and sizes[5] % world == 0 and sizes[5] > threshold
        assert all(s == 4 for i, s in enumerate(sizes) if i not in (0, 5))
    assert sum(len(t) for step in want for t in step.values()) >= n_steps * 3
