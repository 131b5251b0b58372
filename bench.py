#!/usr/bin/env python3
"""bench.py — generated tokens/s of the spec-decode hot path on MI355X (BASELINE.json metric).

A "step" is one engine step of the hot path over a batch of B live requests (Llama-3.1-8B shapes,
arctic LSTM speculator k=3 + suffix decoding, 4K-token prompts, 256 generated tokens per request,
requests replaced as they finish): bulk KV write of the step's tokens, multi-token verify attention
for all 32 layers over the paged KV cache, greedy rejection acceptance on [sum(n_draft), V] logits,
suffix-tree update + batched device suffix proposal, LSTM draft (3 heads), proposal merge.
The target model's dense layers (vLLM's, not part of this path) are represented by synthetic tensors
of their real shapes; the synthetic target is greedy and follows a seeded ground-truth stream, so
draft acceptance is scored model-free exactly like the reference's simulator.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
N > 1 (launched by torch.distributed.run): Ulysses sequence parallelism over the N ranks (heads sharded,
all-to-all around attention over RCCL, vocab-parallel draft LM head); the global batch is fixed, so
scaling is "strong".  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
# every 4th launch of the dominant kernel is bracketed by a HIP event pair (8 of a step's 32 layers, evenly spread
# over the timed region): an event pair per launch cost the stream ~0.4 ms per step, i.e. changed what it measured
PROFILE_STRIDE = 4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt-len", type=int, default=4096)
    ap.add_argument("--gen-len", type=int, default=256)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lstm", action="store_true")
    ap.add_argument("--kv-dtype", default="auto", choices=["auto", "fp8"], help="KV cache dtype (auto = bf16, the headline config)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-shift-parallel", action="store_true",
                    help="N > 1: keep every step on the Ulysses all-to-all path (default: shift parallelism on, threshold "
                         "512 tokens, the reference's --enable-shift-parallel / --shift-parallel-threshold)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the measured configuration) or gloo: a multi-process REHEARSAL of the N > 1 control "
                         "flow with the ranks sharing the visible GPU(s) and collectives staged through the host "
                         "(numbers are not a multi-GPU measurement)")
    ap.add_argument("--rehearse-sp", type=int, default=0,
                    help="single-GPU rehearsal of the SP=N code path: real pack/unpack/attention shapes of rank 0, "
                         "the all-to-all replaced by a local copy (numbers are NOT a multi-GPU measurement)")
    return ap.parse_args()


def cpu_baseline(args, src, shape, spec):
    """The oracle (CPU restatements; kind "port") timed on this box's host cores on a bounded sample of
    the same workload: 2 requests, one engine step each way (suffix oracle update+speculate per request,
    torch-CPU verify attention for 2 of the 32 layers scaled x16, greedy rejection on [6, V] logits,
    LSTM draft with Ds=H=4096 and the full 128256-row head).  Reported beside the GPU number only."""
    from oracle import spec_oracle as O
    from oracle.suffix_oracle import OracleSuffixCache
    t_all = time.perf_counter()
    nreq, k = 2, spec.num_speculative_tokens
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    D, Hq, Hkv, bs = shape.head_size, shape.num_q_heads, shape.num_kv_heads, shape.block_size
    ctx = args.prompt_len + 64
    nblk = (ctx + bs - 1) // bs
    kc = torch.randn(nreq * nblk, bs, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(nreq * nblk, bs, Hkv, D, generator=g).to(torch.bfloat16)
    bt = torch.arange(nreq * nblk, dtype=torch.int32).view(nreq, nblk)
    q = torch.randn(nreq * (k + 1), Hq, D, generator=g).to(torch.bfloat16)
    qsl = np.arange(nreq + 1, dtype=np.int32) * (k + 1)
    layers_sample = 2
    t0 = time.perf_counter()
    for _ in range(layers_sample):
        O.verify_attention(q, kc, vc, bt, [ctx] * nreq, qsl, D ** -0.5)
    t_attn = (time.perf_counter() - t0) * (shape.num_layers / layers_sample)
    logits = torch.randn(nreq * k, shape.vocab_size, generator=g).to(torch.bfloat16)
    t0 = time.perf_counter()
    O.rejection_greedy(logits, [1] * (nreq * k), [k] * nreq, [0] * nreq, k)
    t_rej = time.perf_counter() - t0
    # suffix oracle: prompt trees + a few steps
    cache = OracleSuffixCache(spec.suffix_cache_max_depth)
    rows = []
    for r in range(nreq):
        p, gt = src.request(10_000 + r, args.prompt_len, 64)
        cache.cache_prompt(r, [int(x) for x in p])
        rows.append(([int(x) for x in p], [int(x) for x in gt]))
    t0 = time.perf_counter()
    n_sfx_steps = 16
    for stp in range(n_sfx_steps):
        for r in range(nreq):
            p, gt = rows[r]
            cache.update_response(r, gt[2 * stp:2 * stp + 2])
            cache.speculate(r, (p + gt[:2 * stp + 2])[-64:], max_spec_tokens=32)
    t_sfx = (time.perf_counter() - t0) / n_sfx_steps
    # LSTM oracle (bf16 head on the CPU), Ds = H = 4096, V = 128256
    t_lstm = 0.0
    if not args.no_lstm:
        from arcticinference_amd.speculator import LSTMSpeculatorConfig, random_lstm_weights
        cfg = LSTMSpeculatorConfig(vocab_size=shape.vocab_size, input_hidden_dim=shape.hidden_size)
        w = O.merge_lstm_checkpoint(random_lstm_weights(cfg, seed=0))
        hid = torch.randn(nreq, shape.hidden_size, generator=g).to(torch.bfloat16)
        t0 = time.perf_counter()
        O.lstm_generate_proposals(w, torch.tensor([1, 2]), hid, k, cfg.n_predict, True)
        t_lstm = time.perf_counter() - t0
    step_s = t_attn + t_rej + t_sfx + t_lstm
    # tokens emitted per request-step on this workload are measured by the GPU run; use 1 + accept rate later
    return {"step_seconds_2req": step_s, "parts_s": {"attention": t_attn, "rejection": t_rej, "suffix": t_sfx,
                                                     "lstm": t_lstm},
            "cores": threads, "sample_wall_s": time.perf_counter() - t_all, "nreq": nreq}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the hot path)"
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()      # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from arcticinference_amd import _native as N
    from arcticinference_amd.engine import HotPathEngine, ModelShape, SpecConfig
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    from arcticinference_amd.workload import TokenSource

    shape = ModelShape(num_layers=args.layers)
    spec = SpecConfig()
    B, PL, GL = args.batch, args.prompt_len, args.gen_len
    max_model_len = PL + GL + 64
    src = TokenSource(seed=args.seed)

    ulysses = None
    tp_group = None
    if world > 1:
        from arcticinference_amd.ulysses import UlyssesContext
        ulysses = UlyssesContext(world, rank, dist.group.WORLD, shape, device=dev,
                                 enable_shift_parallel=not args.no_shift_parallel, shift_parallel_threshold=512)
        tp_group = dist.group.WORLD

    if world == 1 and args.rehearse_sp > 1:
        from arcticinference_amd.ulysses import UlyssesContext
        ulysses = UlyssesContext(args.rehearse_sp, 0, None, shape, device=dev, all_to_all=lambda recv, send: recv.copy_(send),
                                 enable_shift_parallel=not args.no_shift_parallel, shift_parallel_threshold=512)

    drafter = None
    if not args.no_lstm:
        cfg = LSTMSpeculatorConfig(vocab_size=shape.vocab_size, input_hidden_dim=shape.hidden_size)
        drafter = ArcticLSTMSpeculator(cfg, max_num_seqs=B, tp_size=world, tp_rank=rank, tp_group=tp_group, device=dev,
                                       quantize_lm_head=True)
        drafter.load_weights(random_lstm_weights(cfg, seed=args.seed).items())

    eng = HotPathEngine(shape, spec, B, max_model_len, drafter, device=dev, ulysses=ulysses, seed=args.seed,
                        kv_cache_dtype=args.kv_dtype)

    # ---- workload: B live requests; finished ones are replaced by fresh ones ---------------------------
    streams = {}
    next_id = [0]

    def new_request():
        rid = next_id[0]
        next_id[0] += 1
        s = src.stream(PL + GL + 128, rid)
        streams[rid] = s
        return rid, s[:PL], int(s[PL])

    first = [new_request() for _ in range(B)]
    eng.add_requests(list(range(B)), [f[0] for f in first], [f[1] for f in first], [f[2] for f in first])

    def truth(r, n):
        s = streams[r.req_id]
        p = len(r.tokens)
        return s[p:p + n]

    gen_tokens = [0]

    def run_step():
        emitted = eng.step(truth)
        live = [i for i, r in enumerate(eng.requests) if r is not None]
        for slot, toks in zip(live, emitted):
            r = eng.requests[slot]
            done = len(r.tokens) - r.num_prompt
            over = max(0, done - GL)
            gen_tokens[0] += len(toks) - min(over, len(toks))
            if done >= GL:
                streams.pop(r.req_id, None)
                rid, prompt, ft = new_request()
                eng.add_request(slot, rid, prompt, ft)   # includes the prompt-tree build (model_runner.py:664-671)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    barrier()
    N.lib().aic_profile_enable(0 if os.environ.get("AIC_BENCH_NOPROFILE") else PROFILE_STRIDE)
    gen_tokens[0] = 0
    eng.stats = type(eng.stats)()
    eng.timeline = {}
    if ulysses is not None:
        ulysses.steps_sp = ulysses.steps_shift = 0
    attn_bytes = [0.0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        # algorithmic KV bytes of one launch of the dominant kernel (one launch per layer covers every request: the
        # short-request body and the long-draft body are workgroups of the same grid):
        # sum_i ctx_i * 2 (K,V) * Hkv_local * D * bytes per element
        kvb = 2 if args.kv_dtype == "auto" else 1
        ctx_all = sum(len(r.tokens) + r.num_drafts for r in eng.requests if r is not None)
        attn_bytes[0] += ctx_all * 2 * eng.hkv_local * shape.head_size * kvb * shape.num_layers
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    import ctypes
    tot_us, launches = ctypes.c_double(0), ctypes.c_int(0)
    N.lib().aic_profile_read(ctypes.byref(tot_us), ctypes.byref(launches))
    N.lib().aic_profile_enable(0)

    red_dev = dev if args.dist_backend == "nccl" else "cpu"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    import copy
    gen_total, stats_snapshot, timeline_snapshot = gen_tokens[0], copy.copy(eng.stats), dict(eng.timeline)
    steps_shift = ulysses.steps_shift if ulysses is not None else 0
    steps_sp = ulysses.steps_sp if ulysses is not None else 0

    # N > 1 with shift parallelism: the decode-size steps above ran in shift (TP) mode.  A few extra steps, outside
    # `value`, with shift off put the Ulysses all-to-all path (2 RCCL all_to_all_single per layer) on the record too.
    a2a_ms = None
    if ulysses is not None and ulysses.enable_shift_parallel and steps_shift > 0:
        ulysses.enable_shift_parallel = False
        k2 = max(4, min(8, args.steps))
        run_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            run_step()
        barrier()
        a2a = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([a2a], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            a2a = float(t.item())
        a2a_ms = a2a / k2 * 1e3
        ulysses.enable_shift_parallel = True

    if rank == 0:
        value = gen_total / elapsed
        st = stats_snapshot
        avg_launch_us = tot_us.value / max(launches.value, 1)
        bytes_per_launch = attn_bytes[0] / max(args.steps * shape.num_layers, 1)   # every launch of a step moves the same bytes
        achieved = bytes_per_launch / (avg_launch_us * 1e-6) / 1e9 if launches.value else 0.0
        # PMC-measured HBM bytes per launch (profiles/r01_pmc_attention.json, tools/pmc_summary.py): recorded for the
        # default single-GPU workload only; any other shape reports null rather than a number that is not its own
        traffic = None
        default_shape = (world == 1 and args.rehearse_sp <= 1 and args.kv_dtype == "auto" and B == 64 and PL == 4096
                         and GL == 256 and shape.num_layers == 32 and not args.no_lstm)
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_attention.json")
        if default_shape and os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "gen tokens/sec/GPU + mean accepted draft len, Llama-3.1-8B spec-decode SP=1/8",
            "value": value,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",   # the global batch is fixed (64 live requests); N GPUs share it through SP / shift
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": ("Llama-3.1-8B shapes (L=%d, Hq=32, Hkv=8, D=128, V=128256), arctic LSTM speculator k=3 "
                             "(Ds=4096, fp8 head when padded batch <= 32) + suffix decoding, B=%d live requests, "
                             "%d-token prompts, %d generated tokens each, greedy, KV cache %s; hot path only (verify "
                             "attention, acceptance, suffix + LSTM proposal, KV write); target dense layers synthetic"
                             % (shape.num_layers, B, PL, GL, "bf16" if args.kv_dtype == "auto" else "fp8 e4m3")),
                "global_batch": B, "prompt_len": PL, "gen_len": GL,
                "parallelism": (("sp%d" % world + ("" if args.no_shift_parallel else "+shift(threshold 512 tokens)") +
                                 ("" if args.dist_backend == "nccl" else " REHEARSAL over gloo, ranks share GPUs")) if world > 1
                                else ("tp1" if args.rehearse_sp <= 1 else "REHEARSAL sp%d on one GPU" % args.rehearse_sp)),
            },
            "tokens_per_s_per_gpu": value / world,
            "steps_in_shift_mode": steps_shift, "steps_in_sp_mode": steps_sp,
            "ulysses_all_to_all_path_ms_per_step": a2a_ms,
            "mean_accepted_draft_len": st.accepted / max(st.num_drafts, 1),
            "draft_acceptance_rate": st.accepted / max(st.drafted, 1),
            "tokens_per_request_step": st.emitted / max(args.steps * B, 1),
            "suffix_share_of_drafts": st.suffix_used / max(args.steps * B, 1),
            "host_timeline_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in timeline_snapshot.items()},
            "roofline": {"bound": "hbm", "kernel": "verify_attn_pair_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_us": avg_launch_us, "launches_timed": launches.value,
                         "launches": args.steps * shape.num_layers,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "every %d-th launch timed with a HIP event pair inside the library, on the launch's "
                                 "stream; the kernel is verify_attn_pair_kernel (short-request and long-draft "
                                 "workgroups in one grid) or verify_attn_kernel when a step has no long draft"
                                 % PROFILE_STRIDE},
        }
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(args, src, shape, spec)
            toks_per_req_step = st.emitted / max(args.steps * B, 1)
            line["cpu_baseline"] = {
                "value": cb["nreq"] * toks_per_req_step / cb["step_seconds_2req"], "unit": "tokens/s",
                "cores": cb["cores"], "kind": "port",
                "sample": "oracle (CPU restatements) on 2 requests x 1 engine step: suffix update+speculate, torch-CPU "
                          "verify attention on 2 of 32 layers (scaled x16), greedy rejection on [6, V], LSTM draft "
                          "(Ds=4096, V=128256); tokens per request-step taken from the GPU run",
                "parts_s": cb["parts_s"], "sample_wall_s": cb["sample_wall_s"]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
