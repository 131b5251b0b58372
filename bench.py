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

The batch is in STEADY STATE: the B requests start at generated positions spread evenly over 0..gen_len-1, so
requests finish (and are replaced by fresh ones, prompt-tree build included) inside the timed region at the
rate a long run would see, contexts average prompt + gen/2, and the response trees are populated.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
N > 1: Ulysses sequence parallelism over the N ranks (heads sharded, all-to-all around attention over RCCL,
vocab-parallel draft LM head); the global batch is fixed, so scaling is "strong".  Launched by
torch.distributed.run (WORLD_SIZE set) the process is one rank; started plainly with --gpus N > 1 it starts
the N ranks itself (a child `python -m torch.distributed.run`, before anything here touches the GPU) and
exits with their code.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import collections
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
    ap.add_argument("--no-attn-graph", action="store_true",
                    help="launch the attention layers kernel by kernel instead of as one HIP graph per step (A/B switch)")
    ap.add_argument("--long-splits", type=int, default=0,
                    help="A/B switch: force the split count of the long-draft part of mixed attention calls (0 = the library's rule)")
    ap.add_argument("--no-lstm", action="store_true")
    ap.add_argument("--no-spec", action="store_true",
                    help="comparison run, not the headline: speculation off (no suffix cache, no draft model) — every request "
                         "decodes one token per step through the same attention + greedy sampling kernels")
    ap.add_argument("--no-suffix", action="store_true",
                    help="BASELINE configs[1]: the Arctic LSTM speculator alone (method \"arctic\", no suffix decoding) — the "
                         "draft model runs in every step; a second roofline entry is reported for its kernels")
    ap.add_argument("--proposal-indexing", default=None, choices=["single_advance", "reference"],
                    help="where a request's row ends for the proposers (vllm_plugin/runner_logic.py).  Default: the LIBRARY's "
                         "default (runner_logic.DEFAULT_INDEXING = \"single_advance\" since r04: the row as the step left it, the "
                         "algorithm of the reference's simulator and of the golden fixtures) — the headline is what the plugin "
                         "does out of the box.  \"reference\" is the opt-in switch for the reference plugin's literal arithmetic "
                         "(sampled ids counted twice, model_runner.py:623-636 / :696-718).  The other mode's acceptance is "
                         "measured in a few extra rounds after the timed region and printed beside the headline")
    ap.add_argument("--plant-draft-prob", type=float, default=0.7,
                    help="draft-model legs only (--no-suffix, and the LSTM-only leg of the default run): probability, per draft "
                         "position, that the synthetic target's arg-max token IS the draft model's token (SURVEY 8(d): \"the "
                         "draft token planted as argmax with prob 0.7 per position\"); a random-weight draft model never hits "
                         "the ground-truth stream, so without it that leg's acceptance is identically 0")
    ap.add_argument("--no-lstm-leg", action="store_true",
                    help="skip the LSTM-only leg (BASELINE configs[1]) that the default run measures after the timed region")
    ap.add_argument("--no-replay-check", action="store_true",
                    help="skip the one-request-at-a-time suffix replay on the golden token source (method \"suffix\", every "
                         "draft taken; its avg_accept_toks must equal tests/golden/suffix_replay.json)")
    ap.add_argument("--kv-dtype", default="auto", choices=["auto", "fp8"], help="KV cache dtype (auto = bf16, the headline config)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--qlen-hist", action="store_true", help="diagnostic: histogram of per-request query lengths in the timed steps")
    ap.add_argument("--lanes", type=int, default=0,
                    help="2: the live requests form two lanes of B/2 whose engine steps are interleaved — one lane's host "
                         "chain (tree update, suffix proposal, index build) runs while the GPU attends for the other; a "
                         "round still advances every request by one step.  1: every request in one engine step per round "
                         "(host chain and GPU alternate).  0 (default): 2 up to SP = 4, 1 at SP = 8, where a rank's launches are small "
                         "and a second lane mostly adds their fixed costs (measured, ms per round, 1 / 2 lanes; r02: one GPU "
                         "6.76 / 6.43, rehearsed SP 2: 4.02 / 3.86, SP 4: 2.72 / 2.64, SP 8: 2.10 / 2.19; r03, same box: SP 4: "
                         "2.76 / 2.66, SP 8 three pairs: 2.16 / 2.20, 2.03 / 2.15, 2.21 / 2.38; decode-size steps run in shift "
                         "mode, i.e. without collectives inside attention, so a second lane adds none there)")
    ap.add_argument("--draft-model-per-request", action="store_true",
                    help="extension: requests that suffix decoding did not take still get the draft model's proposal in "
                         "steps where it took others (the reference gives the whole batch none, model_runner.py:616-618)")
    ap.add_argument("--rank-owned-trees", action="store_true",
                    help="N > 1: rank-owned prompt trees (arcticinference_amd/suffix_sharding.py) — a request's prompt tree lives "
                         "on ONE rank (slot %% N), every rank speculates for its own requests only, one int32 all-reduce of "
                         "the [B, 34] result matrix over a gloo control group brings the drafts together.  Bit-identical drafts "
                         "to the replicated control (gloo tests, world sizes 2 and 8); its collective's latency on 8 GPUs is "
                         "unmeasured, so it is opt-in")
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


def _reference_tree_class():
    """The REAL reference suffix tree, if its compiled module travelled with the repo (oracle/_ref/_C*.so, built in the
    build container by oracle/Makefile from the reference's own sources; never in git).  None otherwise."""
    import glob
    import importlib.util
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "_C*.so"))
    if not so:
        return None
    try:
        spec = importlib.util.spec_from_file_location("_C", so[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.SuffixTree
    except Exception:
        return None


def cpu_baseline(args, src, shape, spec, run_stats):
    """The same engine step on this box's host cores, bounded sample, checker code only (oracle/): reported beside
    the GPU number, never part of it.
      * verify attention: torch-CPU scaled_dot_product_attention, ALL B requests batched, bf16, GQA folded into
        the query rows (the fastest CPU form found: K/V are read once per kv head), 2 of the L layers, scaled;
      * acceptance: arg-max over the [B*k, V] bf16 logits (torch-CPU);
      * suffix proposer: the reference's own suffix tree (oracle/_ref, kind "reference") when its compiled module is
        present, else the C++ restatement (oracle/liboracle_suffix.so), under the SuffixCache policy restatement:
        B prompt trees built, then update_response + speculate for all B requests per step;
      * draft model: the torch-CPU bf16 op sequence (oracle.spec_oracle.lstm_generate_proposals) at B rows, weighted by
        the share of steps that used the draft model in the GPU run.
    Tokens per request-step are the GPU run's (identical policy, bit-exact suffix proposer)."""
    import torch.nn.functional as F
    from oracle import spec_oracle as O
    from oracle.suffix_oracle import OracleSuffixCache, OracleSuffixTree
    t_all = time.perf_counter()
    B, k = args.batch, spec.num_speculative_tokens
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    D, Hq, Hkv = shape.head_size, shape.num_q_heads, shape.num_kv_heads
    G = Hq // Hkv
    S = args.prompt_len + args.gen_len // 2
    ql = k + 1
    # (contents do not matter for the timing: a 4-request random block tiled to B keeps the sample's set-up short)
    tile = lambda t: t.repeat((B + t.shape[0] - 1) // t.shape[0], *([1] * (t.dim() - 1)))[:B].contiguous()
    kc = tile(torch.randn(4, Hkv, S, D, generator=g).to(torch.bfloat16))
    vc = tile(torch.randn(4, Hkv, S, D, generator=g).to(torch.bfloat16))
    q = torch.randn(B, Hkv, G * ql, D, generator=g).to(torch.bfloat16)          # rows (g, position) of a kv head
    pos = torch.arange(ql).unsqueeze(1) + (S - ql)
    mask = (torch.arange(S).unsqueeze(0) <= pos).repeat(G, 1)
    F.scaled_dot_product_attention(q[:2], kc[:2], vc[:2], attn_mask=mask)       # warm the thread pool
    layers_sample = 2
    t0 = time.perf_counter()
    for _ in range(layers_sample):
        F.scaled_dot_product_attention(q, kc, vc, attn_mask=mask)
    t_attn = (time.perf_counter() - t0) * (shape.num_layers / layers_sample)
    del kc, vc
    logits = torch.randn(8, shape.vocab_size, generator=g).to(torch.bfloat16).repeat((B * k + 7) // 8, 1)[:B * k].contiguous()
    t0 = time.perf_counter()
    torch.argmax(logits, dim=-1)
    t_rej = time.perf_counter() - t0
    del logits
    # suffix proposer
    ref_tree = _reference_tree_class()
    cache = OracleSuffixCache(spec.suffix_cache_max_depth, tree_cls=ref_tree or OracleSuffixTree)
    rows = [[int(x) for x in src.stream(args.prompt_len + 160, 50_000 + r)] for r in range(B)]
    t0 = time.perf_counter()
    for r in range(B):
        cache.cache_prompt(r, rows[r][:args.prompt_len])
    t_prompt = (time.perf_counter() - t0) / B
    n_sfx_steps = 8
    t0 = time.perf_counter()
    for stp in range(n_sfx_steps):
        for r in range(B):
            e = args.prompt_len + 2 * stp + 2
            cache.update_response(r, rows[r][e - 2:e])
            cache.speculate(r, rows[r][e - 64:e], max_spec_tokens=32)
    t_sfx = (time.perf_counter() - t0) / n_sfx_steps
    del cache
    # draft model
    t_lstm = 0.0
    lstm_share = run_stats["draft_model_steps"] / max(run_stats["steps"], 1)
    if not args.no_lstm:
        from arcticinference_amd.speculator import LSTMSpeculatorConfig, random_lstm_weights
        cfg = LSTMSpeculatorConfig(vocab_size=shape.vocab_size, input_hidden_dim=shape.hidden_size)
        w = O.merge_lstm_checkpoint(run_stats.get("lstm_checkpoint") or random_lstm_weights(cfg, seed=0))
        hid = torch.randn(B, shape.hidden_size, generator=g).to(torch.bfloat16)
        t0 = time.perf_counter()
        O.lstm_generate_proposals(w, torch.arange(B), hid, k, cfg.n_predict, True)
        t_lstm = time.perf_counter() - t0
    repl = run_stats["replacements_per_step"]
    step_s = t_attn + t_rej + t_sfx + t_lstm * lstm_share + t_prompt * repl
    return {"step_seconds": step_s,
            "parts_s": {"attention_32_layers": t_attn, "rejection": t_rej, "suffix_update_and_speculate": t_sfx,
                        "draft_model_x_share": t_lstm * lstm_share, "draft_model_one_call": t_lstm,
                        "prompt_tree_build_x_rate": t_prompt * repl, "prompt_tree_build_one": t_prompt},
            "suffix_tree": "reference (oracle/_ref, compiled from the reference's sources)" if ref_tree else
                           "port (oracle/suffix_tree_oracle.cpp)",
            "cores": threads, "sample_wall_s": time.perf_counter() - t_all}


class TimedDrafter:
    """The draft model with a HIP event pair around every call (on the stream the call launches on: torch's current
    stream, which is what generate_proposals passes to the library) — the second roofline entry of the line."""

    def __init__(self, inner):
        self._inner = inner
        self.pairs = []

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def generate_proposals(self, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self._inner.generate_proposals(*a, **kw)
        e1.record()
        self.pairs.append((e0, e1, int(a[0].size(0))))
        return out

    def reset(self):
        self.pairs = []

    def summary(self, cfg, k: int):
        """(avg us per call, calls, algorithmic weight bytes per call): k heads of gate GEMM (4Ds x H, then 4Ds x Ds, bf16)
        + k passes over the tied LM head (fp8 when the padded batch is <= 32, else bf16), SURVEY.md 8(d) A8 / A9."""
        if not self.pairs:
            return None
        us = [a.elapsed_time(b) * 1e3 for a, b, _ in self.pairs]
        rows = max(n for _, _, n in self.pairs)
        from arcticinference_amd.speculator import padding_size
        m = self._inner
        Ds, H, V = m.inner_dim, m.input_hidden_dim, m.shard_rows       # the head shard of this rank (vocab-parallel under SP)
        head_b = 1 if (m.quantize_lm_head and padding_size(rows) <= 32) else 2      # arctic_speculator.py:726-728
        gate = (4 * Ds * H + (k - 1) * 4 * Ds * Ds) * 2
        head = k * V * Ds * head_b
        return {"avg_us": float(np.mean(us)), "calls": len(us), "rows": rows, "gate_bytes": gate, "head_bytes": head,
                "head_dtype": "fp8 e4m3" if head_b == 1 else "bf16"}


def suffix_replay_check():
    """The golden replay on THIS build's device suffix cache: method "suffix" (every draft taken), one request at a time,
    on the golden token source — oracle/gen_golden.py:replay's loop (simulator.suffix_decode, simulator.py:33-114).  Its
    avg_accept_toks must equal the compiled reference's (tests/golden/suffix_replay.json, the 64 x (4096 + 256) entry)."""
    from arcticinference_amd.suffix_cache import SuffixCache
    from arcticinference_amd.workload import TokenSource
    with open(os.path.join(ROOT, "tests", "golden", "suffix_replay.json")) as f:
        want = json.load(f)[-1]
    c = want["config"]
    src = TokenSource(seed=c["seed"])
    cache = SuffixCache(c["max_depth"])
    acc = steps = spec = 0
    t0 = time.perf_counter()
    for r in range(c["n_req"]):
        prompt, gt = src.request(r, c["prompt_len"], c["gen_len"])
        prompt, gt = [int(x) for x in prompt], [int(x) for x in gt]
        cache.cache_prompt(r, prompt)
        resp = []
        while len(resp) < len(gt):
            res = cache.speculate(r, (prompt + resp)[-c["max_depth"]:], max_spec_tokens=c["max_spec_tokens"],
                                  max_spec_factor=c["factor"], max_spec_offset=c["offset"], min_token_prob=c["min_token_prob"])
            a = 0
            for tok in res.token_ids:
                if len(resp) + a < len(gt) and gt[len(resp) + a] == tok:
                    a += 1
                else:
                    break
            new = gt[len(resp):len(resp) + a]
            resp.extend(new)
            if len(resp) < len(gt):
                new = new + [gt[len(resp)]]
                resp.append(gt[len(resp)])
            cache.update_response(r, new)
            steps += 1
            acc += a
            spec += len(res.token_ids)
        cache.evict_prompt(r)
    got = acc / steps
    return {"avg_accept_toks": got, "sum_accept": acc, "steps": steps, "sum_spec": spec,
            "golden_avg_accept_toks": want["avg_accept_toks"], "golden_sum_accept": want["sum_accept"],
            "golden_steps": want["steps"],
            "equal_to_reference": bool(acc == want["sum_accept"] and steps == want["steps"] and spec == want["sum_spec"]),
            "wall_s": time.perf_counter() - t0,
            "what": "method \"suffix\" (every draft taken), one request at a time, golden token source, run on this build's "
                    "device suffix cache just now; golden = the compiled reference's run (tests/golden/suffix_replay.json)"}


def spawn_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher: start N ranks as a CHILD process tree (never an exec: this process has
    not touched the GPU, and must not before the children exist) and hand back their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the hot path)"
    if args.dist_backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} over RCCL needs {world} visible GPUs, found {torch.cuda.device_count()} "
                         "(use --dist-backend gloo for a rehearsal on fewer)")
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()      # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    # AIC_BENCH_ONE_RANK_GROUP=1 (tests): a ONE-rank process group, so that the N > 1 code — group creation, the Ulysses
    # context over a real group, barrier and the reductions of the result — runs through RCCL on a one-GPU box
    one_rank_group = world == 1 and os.environ.get("AIC_BENCH_ONE_RANK_GROUP") == "1"
    if world > 1 or one_rank_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == world   # n_gpus below is the size of the group that really formed

    from arcticinference_amd import _native as N
    from arcticinference_amd.engine import HotPathEngine, ModelShape, SpecConfig
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    from arcticinference_amd.workload import TokenSource

    if args.no_attn_graph:
        N.check(N.lib().aic_debug_attn_graph(0))
    elif os.environ.get("AIC_ATTN_GRAPH_MODE"):
        N.check(N.lib().aic_debug_attn_graph(int(os.environ["AIC_ATTN_GRAPH_MODE"])))
    if args.long_splits:
        N.check(N.lib().aic_debug_attn_long_splits(args.long_splits))
    shape = ModelShape(num_layers=args.layers)
    spec = SpecConfig(draft_model_per_request=args.draft_model_per_request, proposal_indexing=args.proposal_indexing,
                      enable_suffix_decoding=not args.no_suffix)
    if args.no_spec:
        spec = SpecConfig(method="none", enable_suffix_decoding=False)
        args.no_lstm = True
    B, PL, GL = args.batch, args.prompt_len, args.gen_len
    max_model_len = PL + GL + 64
    src = TokenSource(seed=args.seed)

    ulysses = None
    tp_group = None
    if world > 1 or one_rank_group:
        from arcticinference_amd.ulysses import UlyssesContext
        ulysses = UlyssesContext(world, rank, dist.group.WORLD, shape, device=dev,
                                 enable_shift_parallel=not args.no_shift_parallel, shift_parallel_threshold=512)
        tp_group = dist.group.WORLD

    if world == 1 and args.rehearse_sp > 1:
        from arcticinference_amd.ulysses import UlyssesContext
        ulysses = UlyssesContext(args.rehearse_sp, 0, None, shape, device=dev, all_to_all=lambda recv, send: recv.copy_(send),
                                 enable_shift_parallel=not args.no_shift_parallel, shift_parallel_threshold=512)

    drafter = lstm_ckpt = None
    if not args.no_lstm:
        cfg = LSTMSpeculatorConfig(vocab_size=shape.vocab_size, input_hidden_dim=shape.hidden_size)
        drafter = ArcticLSTMSpeculator(cfg, max_num_seqs=B, tp_size=world, tp_rank=rank, tp_group=tp_group, device=dev,
                                       quantize_lm_head=True)
        lstm_ckpt = random_lstm_weights(cfg, seed=args.seed)
        drafter.load_weights(lstm_ckpt.items())
        if rank != 0 or world > 1 or args.no_cpu_baseline:
            lstm_ckpt = None            # rank 0 of a single-GPU run keeps the host copy for the cpu_baseline leg
        draft_cfg = cfg
        drafter = TimedDrafter(drafter)

    suffix_owner = None
    if args.rank_owned_trees and dist is not None and world > 1:
        from arcticinference_amd.suffix_sharding import gloo_exchange
        ctl = dist.group.WORLD if args.dist_backend == "gloo" else dist.new_group(backend="gloo")
        suffix_owner = (rank, world, gloo_exchange(ctl))
    eng = HotPathEngine(shape, spec, B, max_model_len, drafter, device=dev, ulysses=ulysses, seed=args.seed,
                        kv_cache_dtype=args.kv_dtype, suffix_owner=suffix_owner)

    if args.no_suffix and drafter is not None:
        eng.plant_draft_prob = args.plant_draft_prob          # configs[1]: the synthetic target agrees with the draft at p per position

    # ---- workload: B live requests; finished ones are replaced by fresh ones ---------------------------
    streams = {}
    next_id = [0]

    # the synthetic token source is workload GENERATION, not the path: streams for the live requests and for every
    # replacement the run can need are drawn before the timed region
    n_pool = B + int((args.steps + args.warmup + 16 + 80) * B * 3.0 / GL) + 8   # (+ the legs after the timed region)
    pool = [src.stream(PL + GL + 128, rid) for rid in range(n_pool)]

    def new_request(generated: int = 0):
        """A fresh request; `generated` > 0 admits it mid-generation (its first `generated` + 1 response tokens exist)."""
        rid = next_id[0]
        next_id[0] += 1
        s = pool[rid] if rid < n_pool else src.stream(PL + GL + 128, rid)
        streams[rid] = s
        return rid, s[:PL], s[PL:PL + generated + 1]

    # steady state from step 0: request i has already generated i * GL / B tokens
    first = [new_request((i * GL) // B) for i in range(B)]
    eng.add_requests(list(range(B)), [f[0] for f in first], [f[1] for f in first], [f[2] for f in first])

    def truth(r, n):
        s = streams[r.req_id]
        p = r.num_tokens
        return s[p:p + n]

    gen_tokens = [0]
    replaced = [0]

    attn_bytes = [0.0]
    kvb = 2 if args.kv_dtype == "auto" else 1

    def account(slots_live, emitted):
        """Count the step's tokens (those past a request's gen_len are not generated tokens) and replace the requests
        that finished."""
        done = eng.num_tokens[slots_live] - eng.num_prompt[slots_live]
        counts = emitted.counts
        gen_tokens[0] += int((counts - np.minimum(np.maximum(done - GL, 0), counts)).sum())
        for slot in slots_live[done >= GL].tolist():
            streams.pop(eng.requests[slot].req_id, None)
            rid, prompt, ft = new_request()
            eng.add_request(slot, rid, prompt, ft)   # prompt tree of the new request (model_runner.py:664-671)
            replaced[0] += 1

    sp_ways = world if world > 1 else max(args.rehearse_sp, 1)
    n_lanes = args.lanes if args.lanes > 0 else (2 if sp_ways <= 4 else 1)
    n_lanes = max(1, min(n_lanes, B))
    lane_slots = [list(range(l, B, n_lanes)) for l in range(n_lanes)]
    pending = [None] * n_lanes

    def run_step():
        """One ROUND: every live request advances by one engine step.  With lanes, lane l's host half (finish) runs while
        the GPU works on what the other lanes began."""
        for l in range(n_lanes):
            c = pending[l]
            if c is not None:
                account(c.live, eng.finish(c))
            pending[l] = eng.begin(truth, lane_slots[l] if n_lanes > 1 else None, lane=l)
            # algorithmic KV bytes of the launches just enqueued (one per layer, every request of the lane):
            # sum_i ctx_i * 2 (K,V) * Hkv_local * D * bytes per element
            attn_bytes[0] += eng.last_ctx_sum * 2 * eng.hkv_local * shape.head_size * kvb * shape.num_layers

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    barrier()
    # the host side keeps ~300k long-lived Python objects (token lists of the live requests): a generation-2 garbage
    # collection that walks them costs milliseconds and would land inside a step; they are parked in the permanent
    # generation instead (what vLLM does after start-up)
    import gc
    gc.collect()
    gc.freeze()
    N.lib().aic_profile_enable(0 if os.environ.get("AIC_BENCH_NOPROFILE") else PROFILE_STRIDE)
    graph0 = N.attn_graph_stats()
    gen_tokens[0] = 0
    replaced[0] = 0
    eng.stats = type(eng.stats)()
    eng.timeline = {}
    if args.qlen_hist:
        eng.qlen_hist = np.zeros(40, dtype=np.int64)
        eng.mix_log = []
    if ulysses is not None:
        ulysses.steps_sp = ulysses.steps_shift = 0
    if drafter is not None:
        drafter.reset()
    attn_bytes[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    import ctypes
    tot_us, launches = ctypes.c_double(0), ctypes.c_int(0)
    N.lib().aic_profile_read(ctypes.byref(tot_us), ctypes.byref(launches))
    N.lib().aic_profile_enable(0)
    graph1 = N.attn_graph_stats()
    # what an event pair reads with nothing between its two events (the stream is idle now): the instrument's own offset
    ev_mean, ev_min = ctypes.c_double(0), ctypes.c_double(0)
    N.check(N.lib().aic_profile_event_overhead(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 256,
                                               ctypes.byref(ev_mean), ctypes.byref(ev_min)))

    red_dev = dev if args.dist_backend == "nccl" else "cpu"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    import copy
    gen_total, stats_snapshot, timeline_snapshot = gen_tokens[0], copy.copy(eng.stats), dict(eng.timeline)
    headline_indexing = eng._indexing
    attn_bytes_timed = attn_bytes[0]            # (the legs below keep running steps: the timed region's bytes are these)
    replaced_total = replaced[0]
    steps_shift = ulysses.steps_shift if ulysses is not None else 0
    steps_sp = ulysses.steps_sp if ulysses is not None else 0
    draft_timing = drafter.summary(draft_cfg, spec.num_speculative_tokens) if drafter is not None else None

    # the OTHER reading of the proposal indexing, a few rounds on the same engine outside `value` (every rank runs them:
    # the control flow is replicated): its acceptance and step time go into the line beside the headline mode's
    other_mode = None
    if eng.suffix_cache is not None and not args.no_spec:
        from arcticinference_amd.vllm_plugin.runner_logic import INDEXING_MODES
        other = [m for m in INDEXING_MODES if m != eng._indexing][0]
        headline_mode, eng._indexing = eng._indexing, other
        k3 = max(8, min(24, args.steps))
        for _ in range(4):                      # the drafts in flight were proposed under the headline mode: let them drain
            run_step()
        barrier()
        gen_tokens[0] = 0
        eng.stats = type(eng.stats)()
        t1 = time.perf_counter()
        for _ in range(k3):
            run_step()
        barrier()
        dt = time.perf_counter() - t1
        so = eng.stats
        other_mode = {"proposal_indexing": other, "rounds": k3, "tokens_per_s": gen_tokens[0] / dt, "ms_per_step": dt / k3 * 1e3,
                      "tokens_per_request_step": so.emitted / max(k3 * B, 1),
                      "accepted_per_request_step": so.accepted / max(k3 * B, 1),
                      "mean_accepted_draft_len": so.accepted / max(so.num_drafts, 1),
                      "suffix_share_of_request_steps": so.suffix_used / max(k3 * B, 1),
                      "steps_with_draft_model": so.draft_model_steps, "lane_steps": so.steps}
        eng._indexing = headline_mode
        for _ in range(2):
            run_step()
        barrier()

    # N > 1 with shift parallelism: the decode-size steps above ran in shift (TP) mode.  A few extra steps, outside
    # `value`, with shift off put the Ulysses all-to-all path (2 RCCL all_to_all_single per layer) on the record too.
    a2a_ms, a2a_error = None, None
    if ulysses is not None and ulysses.enable_shift_parallel and steps_shift > 0:
        # an error here (the same on every rank: the inputs are seeded alike) must not cost the run its JSON line
        try:
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
        except Exception as e:                      # noqa: BLE001 - reported in the line, never swallowed silently
            a2a_error = "%s: %s" % (type(e).__name__, e)
            print("bench: the extra all-to-all steps failed: " + a2a_error, file=sys.stderr, flush=True)
        finally:
            ulysses.enable_shift_parallel = True

    # BASELINE configs[1] — the Arctic LSTM speculator ALONE (method "arctic", no suffix decoding), measured on the same
    # engine after the timed region, outside `value`: the draft model runs in every lane step, the synthetic target
    # agrees with each draft token with probability --plant-draft-prob per position, so tokens/s, accepted
    # length and the draft model's roofline entry (every call bracketed by an event pair) are real numbers.  One GPU only.
    lstm_leg = None
    if (world == 1 and drafter is not None and eng.suffix_cache is not None and not args.no_spec and not args.no_suffix
            and not args.no_lstm_leg):
        eng.suffix_cache = None                     # nothing below uses it again (the replay check builds its own)
        eng.spec.enable_suffix_decoding = False
        eng.plant_draft_prob = args.plant_draft_prob
        k4 = max(12, min(32, args.steps))
        for _ in range(6):                          # suffix drafts in flight drain; every request gets a draft-model draft
            run_step()
        barrier()
        gen_tokens[0] = 0
        eng.stats = type(eng.stats)()
        drafter.reset()
        t1 = time.perf_counter()
        for _ in range(k4):
            run_step()
        barrier()
        dt = time.perf_counter() - t1
        so = eng.stats
        kk = spec.num_speculative_tokens
        pp = args.plant_draft_prob
        lstm_leg = {"what": "BASELINE configs[1]: Arctic LSTM speculator k=%d alone (no suffix decoding), same engine and batch, "
                            "%d rounds after the timed region; synthetic target = the draft token with p = %.2f per position, "
                            "else the ground-truth stream's token" % (kk, k4, pp),
                    "rounds": k4, "tokens_per_s": gen_tokens[0] / dt, "ms_per_step": dt / k4 * 1e3,
                    "tokens_per_request_step": so.emitted / max(k4 * B, 1),
                    "mean_accepted_draft_len": so.accepted / max(so.num_drafts, 1),
                    "expected_accepted_draft_len": sum(pp ** j for j in range(1, kk + 1)),
                    "draft_acceptance_rate": so.accepted / max(so.drafted, 1),
                    "steps_with_draft_model": so.draft_model_steps, "lane_steps": so.steps,
                    "draft_timing": drafter.summary(draft_cfg, kk)}

    if rank == 0:
        value = gen_total / elapsed
        st = stats_snapshot
        from arcticinference_amd.vllm_plugin.runner_logic import DEFAULT_INDEXING as RL_DEFAULT
        # a timed launch = event, kernel, event.  `achieved` uses the pairs' reading AS IT IS (an upper bound of the kernel's
        # duration: an empty pair on the same stream reads ~4.5 us by itself); the reading minus the empty pair's is reported
        # beside it, and rocprofv3's kernel durations for this command (profiles/rNN_kernel_stats.csv) lie between the two
        avg_launch_us = tot_us.value / max(launches.value, 1)
        net_launch_us = max(avg_launch_us - ev_mean.value, 0.0) if launches.value else 0.0
        # every launch of a lane's step moves that lane's bytes; n_lanes launches per layer and round
        bytes_per_launch = attn_bytes_timed / max(args.steps * shape.num_layers * n_lanes, 1)
        achieved = bytes_per_launch / (avg_launch_us * 1e-6) / 1e9 if launches.value else 0.0
        # PMC-measured HBM bytes per launch: NOT measured in this run (counters need their own rocprofv3 --pmc passes) —
        # taken from the committed summary of those passes over this same command, and only for the workload they were
        # collected on; any other shape reports null rather than a number that is not its own
        traffic = traffic_source = None
        default_shape = (world == 1 and args.rehearse_sp <= 1 and args.kv_dtype == "auto" and B == 64 and PL == 4096
                         and GL == 256 and shape.num_layers == 32 and not args.no_lstm)
        for name in ("r04_pmc_attention.json", "r03_pmc_attention.json", "r02_pmc_attention.json", "r01_pmc_attention.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if default_shape and os.path.exists(pmc):
                try:
                    with open(pmc) as f:
                        summary = json.load(f)
                    if summary.get("lanes", 1) != n_lanes:      # collected under another schedule: other launches
                        continue
                    traffic = summary.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/" + name + " (separate rocprofv3 --pmc passes, tools/pmc_summary.py; not this run)"
                    break
                except Exception:
                    traffic = None
        replay_check = None
        if world == 1 and not args.no_replay_check and not args.no_spec:
            try:
                replay_check = suffix_replay_check()
            except Exception as e:                      # noqa: BLE001 - reported, not swallowed
                replay_check = {"error": "%s: %s" % (type(e).__name__, e)}
        golden_accept = None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "suffix_replay.json")) as f:
                gr = json.load(f)[-1]
            golden_accept = {"avg_accept_toks": gr["avg_accept_toks"], "sum_accept": gr["sum_accept"], "steps": gr["steps"],
                             "what": "the compiled reference's suffix proposer alone (method \"suffix\": every draft taken), "
                                     "one request at a time, same token source: tests/golden/suffix_replay.json"}
        except Exception:
            pass
        line = {
            "metric": "gen tokens/sec/GPU + mean accepted draft len, Llama-3.1-8B spec-decode SP=1/8",
            "value": value,
            "unit": "tokens/s",
            "n_gpus": (dist.get_world_size() if dist is not None else 1),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",   # the global batch is fixed (64 live requests); N GPUs share it through SP / shift
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": ("Llama-3.1-8B shapes (L=%d, Hq=32, Hkv=8, D=128, V=128256), %sB=%d live requests, "
                             "%d-token prompts, %d generated tokens each, greedy, KV cache %s; hot path only (verify "
                             "attention, acceptance, suffix + LSTM proposal, KV write); target dense layers synthetic — NOT an "
                             "end-to-end serving number; the north-star comparison against vanilla vLLM-ROCm (>= 2x) could "
                             "not be run: vLLM is installed neither in the build container nor on the GPU box"
                             % (shape.num_layers,
                                "SPECULATION OFF (comparison run: one token per request-step), " if args.no_spec else
                                ("arctic LSTM speculator k=3 (Ds=4096, fp8 head when padded batch <= 32), NO suffix decoding "
                                 "(BASELINE configs[1]: the draft model runs every step; synthetic target = the draft token "
                                 "with p = %.2f per position), " % args.plant_draft_prob if args.no_suffix else
                                 "arctic LSTM speculator k=3 (Ds=4096, fp8 head when padded batch <= 32) + suffix decoding, "),
                                B, PL, GL, "bf16" if args.kv_dtype == "auto" else "fp8 e4m3")),
                "global_batch": B, "prompt_len": PL, "gen_len": GL,
                "lanes": n_lanes,
                # attention layers of a step as one HIP graph launch: launches in the timed region, graphs instantiated in it
                "attn_graph": {"launches": graph1[0] - graph0[0], "instantiated": graph1[1] - graph0[1]},
                "schedule": ("one engine step over all %d requests per round" % B if n_lanes == 1 else
                             "%d lanes of %d requests, steps interleaved (one lane's host chain under the other's attention); "
                             "a round = every request advances one step" % (n_lanes, B // n_lanes)),
                "parallelism": (("sp%d" % world + ("" if args.no_shift_parallel else "+shift(threshold 512 tokens)") +
                                 ("" if args.dist_backend == "nccl" else " REHEARSAL over gloo, ranks share GPUs")) if world > 1
                                else ("tp1" if args.rehearse_sp <= 1 else "REHEARSAL sp%d on one GPU" % args.rehearse_sp)),
            },
            "tokens_per_s_per_gpu": value / world,
            "steps_in_shift_mode": steps_shift, "steps_in_sp_mode": steps_sp,
            "ulysses_all_to_all_path_ms_per_step": a2a_ms,
            "ulysses_all_to_all_path_error": a2a_error,
            "proposal_indexing": headline_indexing,
            "proposal_indexing_is_library_default": headline_indexing == RL_DEFAULT,
            "proposal_indexing_note": ("the headline runs the library's default (\"%s\": what the plugin does out of the box); "
                                       "\"single_advance\" takes a request's row as the step left it (the reference simulator's "
                                       "/ golden fixtures' algorithm), \"reference\" is the reference plugin's literal arithmetic, "
                                       "which counts the sampled ids twice (model_runner.py:623-636, :696-718) — the mode that is "
                                       "not the headline is measured in \"other_indexing_mode\"" % RL_DEFAULT),
            "other_indexing_mode": other_mode,
            # per DRAFT (the counters of stats.py:29-33) and per REQUEST-STEP (the simulator's avg_accept_toks,
            # simulator.py:224-229: accepted tokens / steps) — only the latter compares with reference_suffix_replay
            "mean_accepted_draft_len": st.accepted / max(st.num_drafts, 1),
            "accepted_per_request_step": st.accepted / max(args.steps * B, 1),
            "suffix_replay_check": replay_check,
            "reference_suffix_replay": golden_accept,
            "requests_replaced_in_timed_region": replaced_total,
            "suffix_control_plane": ("rank-owned prompt trees (suffix_sharding.py), one int32 all-reduce per step"
                                     if suffix_owner is not None else "replicated on every rank (the reference's arrangement)"),
            "draft_model_policy": ("per request (extension)" if spec.draft_model_per_request else
                                   "reference rule: no draft-model proposal in a step where suffix decoding takes a request"),
            **({"query_len_histogram": {str(i): int(c) for i, c in enumerate(eng.qlen_hist) if c},
                "long_drafts_per_lane_step": [[list(k), v] for k, v in collections.Counter(eng.mix_log).most_common(24)]}
               if args.qlen_hist else {}),
            "steps_with_draft_model": st.draft_model_steps, "draft_model_launches_dropped": st.draft_model_dropped,
            "draft_acceptance_rate": st.accepted / max(st.drafted, 1),
            "tokens_per_request_step": st.emitted / max(args.steps * B, 1),
            "suffix_share_of_drafts": st.suffix_used / max(args.steps * B, 1),
            "host_timeline_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in timeline_snapshot.items()},
            "roofline": {"bound": "hbm", "kernel": "verify_attn_pair_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "avg_launch_us": avg_launch_us, "launches_timed": launches.value,
                         "empty_event_pair_us": ev_mean.value, "avg_launch_us_minus_empty_pair": net_launch_us,
                         "frac_minus_empty_pair": (bytes_per_launch / (net_launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                                                   if net_launch_us > 0 else 0.0),
                         "launches": args.steps * shape.num_layers * n_lanes,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "every %d-th launch timed with a HIP event pair inside the library, on the launch's "
                                 "stream%s; avg_launch_us and frac use the pairs' reading as it is; empty_event_pair_us = what a "
                                 "pair with nothing between its events reads (256 pairs, same stream, after the timed region), "
                                 "*_minus_empty_pair = with that subtracted — rocprofv3's kernel durations for the same command "
                                 "(profiles/) lie between the two; the kernel is verify_attn_pair_kernel (short-request and long-draft "
                                 "workgroups in one grid) or verify_attn_kernel when a step has no long draft"
                                 % (PROFILE_STRIDE, "" if args.no_attn_graph else
                                    ", in every 5th engine step (those go out kernel by kernel; the other steps' layers are "
                                    "HIP graph launches, which carry no events)")},
        }
        def draft_roofline(dt_, where):
            db = dt_["gate_bytes"] + dt_["head_bytes"]
            da = db / (dt_["avg_us"] * 1e-6) / 1e9
            return {"bound": "hbm", "kernel": "draft model call: skinny_pair_kernel (gate + LM head) + lstm cell kernels, k=%d"
                                              % spec.num_speculative_tokens,
                    "achieved": da, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": da / HBM_PEAK_GBS, "traffic": None,
                    "avg_call_us": dt_["avg_us"], "calls_timed": dt_["calls"], "rows": dt_["rows"],
                    "algorithmic_bytes_per_call": db, "gate_bytes": dt_["gate_bytes"],
                    "head_bytes": dt_["head_bytes"], "head_dtype": dt_["head_dtype"], "measured_in": where,
                    "note": "weights read once per head (SURVEY 8(d) A8 + A9); every call bracketed by a HIP event pair on "
                            "its launch stream"}
        if lstm_leg is not None:
            dtm = lstm_leg.pop("draft_timing")
            if dtm is not None:
                lstm_leg["roofline_draft_model"] = draft_roofline(dtm, "the LSTM-only leg")
            line["lstm_only_leg"] = lstm_leg
        # the draft model's roofline entry: from the run that timed more of its calls (the default run's timed region uses
        # the draft model in a few lane steps only — the reference's merge rule — the LSTM-only leg in every one)
        cands = [(draft_timing, "the timed region")]
        if lstm_leg is not None and "roofline_draft_model" in lstm_leg:
            cands.append((dtm, "the LSTM-only leg (after the timed region)"))
        cands = [c for c in cands if c[0] is not None]
        if cands:
            best = max(cands, key=lambda c: c[0]["calls"])
            line["roofline_draft_model"] = draft_roofline(*best)
        if not args.no_cpu_baseline and world == 1:
            toks_per_req_step = st.emitted / max(args.steps * B, 1)
            cb = cpu_baseline(args, src, shape, spec, {"steps": st.steps, "draft_model_steps": st.draft_model_steps,
                                                       "replacements_per_step": replaced_total / max(args.steps, 1),
                                                       "lstm_checkpoint": lstm_ckpt})
            ref_cpu = None
            try:
                with open(os.path.join(ROOT, "profiles", "r01_reference_cpu_suffix.json")) as f:
                    rc_ = json.load(f)
                ref_cpu = {"source": "profiles/r01_reference_cpu_suffix.json (the reference's suffix proposer alone, timed in the "
                                     "build container on %d cores; oracle/time_reference_cpu.py)" % rc_["host"]["cpus"],
                           "single_thread": {k_: rc_["single_thread"][k_] for k_ in
                                             ("us_per_speculate", "us_per_updated_token", "us_per_cache_prompt_4096",
                                              "generated_tokens_per_s", "mean_accepted_per_step")},
                           "independent_processes_%d" % rc_["processes"]: {
                               "generated_tokens_per_s": rc_["independent_processes"]["generated_tokens_per_s"]}}
            except Exception:
                pass
            line["cpu_baseline"] = {
                "value": B * toks_per_req_step / cb["step_seconds"], "unit": "tokens/s",
                "cores": cb["cores"], "kind": "port",
                "kind_by_part": {"suffix_update_and_speculate + prompt_tree_build":
                                 "reference (oracle/_ref)" if cb["suffix_tree"].startswith("reference") else "port",
                                 "attention_32_layers": "port (torch-CPU SDPA: the reference has no CPU attention of its own)",
                                 "rejection": "port", "draft_model": "port (oracle.spec_oracle, the reference's op sequence)"},
                "sample": "one engine step of the same B=%d workload on the host cores: torch-CPU SDPA over all %d requests "
                          "(bf16, GQA folded into the query rows, ctx %d) on 2 of %d layers scaled; arg-max acceptance on "
                          "[%d, V]; suffix proposer = %s under the SuffixCache policy restatement, %d prompt trees + 8 steps "
                          "of update+speculate for all requests; torch-CPU bf16 LSTM draft (Ds=4096, V=128256) at %d rows "
                          "weighted by the share of steps that used the draft model; tokens per request-step from the GPU run"
                          % (B, B, PL + GL // 2, shape.num_layers, B * spec.num_speculative_tokens, cb["suffix_tree"], B, B),
                "parts_s": cb["parts_s"], "sample_wall_s": cb["sample_wall_s"], "reference_suffix_proposer_cpu": ref_cpu}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
