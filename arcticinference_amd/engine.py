"""Stand-alone driver of the spec-decode hot loop — the part of GPUModelRunnerPatch.execute_model
(/root/reference/arctic_inference/vllm/model_runner.py:218-524, call stack in SURVEY.md §3.3) that this
repository owns: verify attention for the step's (1 + n_draft) query tokens per request, rejection
acceptance, suffix-cache update + batched suffix proposal, LSTM proposal, and the selection rule
(`suffix wins iff score >= num_speculative_tokens`, :555-566, `suffix_ids[i] or lstm_ids[i]`, :597-601).

The target model's dense layers (QKV / MLP GEMMs, norms, LM head) belong to vLLM and are not part of
the path: their outputs (per-layer q, sample hidden states, verify logits) are synthetic tensors of the
real shapes.  bench.py and the integration tests drive this class; the vLLM patch layer
(arcticinference_amd/vllm_plugin/) calls the same ops from inside vLLM's model runner.

Ordering is MI355X-first: everything the GPU needs for drafting (last accepted token, hidden-state
row) is produced ON the device by the acceptance kernel, so the LSTM draft is enqueued before the
host touches the step's results and runs while the host updates the suffix trees.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from collections.abc import Sequence
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _native as N
from . import ops
from .speculator import ArcticLSTMSpeculator
from .suffix_cache import SuffixCache, SuffixSpecResult
from .vllm_plugin.runner_logic import INDEXING_REFERENCE, proposal_indexing

MAX_SPEC_LEN = 32  # vllm.v1.sample.rejection_sampler.MAX_SPEC_LEN (model_runner.py:42, :716)


@dataclass
class SpecConfig:
    """Names and defaults of ArcticSpeculativeConfig (/root/reference/arctic_inference/vllm/config.py:55-62,:93-102)."""
    method: str = "arctic"
    num_speculative_tokens: int = 3
    enable_suffix_decoding: bool = True
    suffix_cache_max_depth: int = 64
    suffix_max_spec_factor: float = 1.0
    suffix_max_spec_offset: float = 0.0
    suffix_min_token_prob: float = 0.1
    disable_by_batch_size: int = 64
    # The reference's propose_arctic_draft_token_ids returns no draft for ANY request of a step in which some request
    # was taken by suffix decoding (model_runner.py:616-618: an emptied sampled list + enable_suffix_decoding ends the
    # whole batch's draft-model proposal).  False reproduces that; True is this build's extension: the draft model
    # still serves the requests suffix decoding did not take.
    draft_model_per_request: bool = False
    # Where a request's row ends for the proposers (vllm_plugin/runner_logic.py): "reference" = the literal arithmetic of
    # model_runner.py:623-636 / :696-718 run after :469-486 has advanced the row (sampled ids counted twice),
    # "single_advance" = the row as the step left it.  None = the library default ("single_advance" since r04, see
    # runner_logic.py for why; "reference" is the opt-in parity switch).
    proposal_indexing: Optional[str] = None


@dataclass
class ModelShape:
    num_layers: int = 32
    num_q_heads: int = 32
    num_kv_heads: int = 8
    head_size: int = 128
    hidden_size: int = 4096
    vocab_size: int = 128256
    block_size: int = 16


@dataclass
class StepStats:
    emitted: int = 0
    accepted: int = 0
    drafted: int = 0
    num_drafts: int = 0
    suffix_used: int = 0
    steps: int = 0
    draft_model_steps: int = 0      # steps whose draft-model proposal was used
    draft_model_dropped: int = 0    # steps whose draft-model proposal was enqueued and then dropped (suffix decoding won)


class _PendingDrafts:
    """LSTM draft tokens of one step on their way to the host (pinned buffer + event).  The next step does not need
    their VALUES on the host (it fills them into its input ids on the device), so nobody waits for this copy on the
    critical path; whoever reads `RequestState.drafts` first resolves it."""

    def __init__(self, eng, pinned: torch.Tensor, event, slots: np.ndarray, rows: np.ndarray, ks: np.ndarray):
        self.eng, self.pinned, self.event = eng, pinned, event
        self.slots, self.rows, self.ks = slots, rows, ks     # slot, row in the LSTM output, number of tokens
        self.done = False

    def resolve(self) -> None:
        if self.done:
            return
        self.done = True
        self.event.synchronize()
        host = self.pinned.numpy()
        e = self.eng
        for slot, row, k in zip(self.slots.tolist(), self.rows.tolist(), self.ks.tolist()):
            if e._pending[slot] is self:
                e.draft_ids[slot, :k] = host[row, :k]
                e._pending[slot] = None
                e.draft_row[slot] = -1


class RequestState:
    """One live request.  Its state lives in the engine's per-slot arrays (vLLM's InputBatch layout: token_ids_cpu,
    num_tokens, the scheduled spec token ids), which is what lets a step's host work run on whole-batch array
    operations; this object is the per-request view of them."""
    __slots__ = ("req_id", "slot", "num_prompt", "_eng")

    def __init__(self, eng, slot: int, req_id, num_prompt: int):
        self._eng, self.slot, self.req_id, self.num_prompt = eng, slot, req_id, num_prompt

    @property
    def num_tokens(self) -> int:
        return int(self._eng.num_tokens[self.slot])

    @property
    def tokens(self) -> np.ndarray:
        """prompt + every sampled token (a view of the token_ids_cpu row)"""
        return self._eng.token_ids_cpu[self.slot, :self._eng.num_tokens[self.slot]]

    @property
    def blocks(self) -> np.ndarray:
        return self._eng._free_blocks[self.slot]

    @property
    def drafts(self) -> List[int]:
        """Draft token ids scheduled for the next step (spec_token_ids).  LSTM drafts may still be in flight from the
        device; reading them here waits for that copy."""
        e = self._eng
        if e._pending[self.slot] is not None:
            e._pending[self.slot].resolve()
        return e.draft_ids[self.slot, :e.n_draft[self.slot]].tolist()

    @drafts.setter
    def drafts(self, value: Sequence[int]) -> None:
        e = self._eng
        e.n_draft[self.slot] = len(value)
        e.draft_ids[self.slot, :len(value)] = value
        e._pending[self.slot] = None
        e.draft_row[self.slot] = -1

    @property
    def num_drafts(self) -> int:
        return int(self._eng.n_draft[self.slot])


class Emitted(Sequence):
    """Tokens emitted per request of a step: behaves like a list of lists; `flat` / `counts` are the arrays behind it."""

    def __init__(self, flat: np.ndarray, counts: np.ndarray):
        self.flat, self.counts = flat, counts
        self._cu = np.concatenate([[0], np.cumsum(counts)])

    def __len__(self) -> int:
        return len(self.counts)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self.flat[self._cu[i]:self._cu[i + 1]].tolist()

    def __eq__(self, other):
        return list(self) == list(other)


class HotPathEngine:
    """One rank of the hot loop.  With sp_size > 1 the attention runs Ulysses-style on this rank's head
    slice (see arcticinference_amd/ulysses.py); everything else is replicated, as in the reference."""

    def __init__(self, shape: ModelShape, spec: SpecConfig, max_num_seqs: int, max_model_len: int,
                 speculator: Optional[ArcticLSTMSpeculator], device: str = "cuda", ulysses=None, seed: int = 0,
                 kv_cache_dtype: str = "auto", suffix_owner=None):
        """`suffix_owner` = (rank, world, exchange): rank-owned prompt trees (suffix_sharding.RankOwnedSuffix) — this rank
        keeps the prompt trees of the slots it owns (slot % world == rank) and speculates for those requests only; `exchange`
        all-reduces the [B, 34] int32 result matrix.  None: every prompt tree here (the reference's replicated control)."""
        self.shape, self.spec = shape, spec
        self.device = torch.device(device)
        self.max_num_seqs, self.max_model_len = max_num_seqs, max_model_len
        self.drafter = speculator
        self.ulysses = ulysses
        self._indexing = proposal_indexing(spec)
        # index build / parse of a step: native (csrc/engine_host.cpp) unless AIC_ENGINE_NUMPY=1 (the numpy form, kept as the
        # A/B reference)
        import os
        self._native_index = os.environ.get("AIC_ENGINE_NUMPY", "0") != "1"
        # begin()'s phase streams, AIC_ENGINE_PHASE_STREAMS = 2 (default): what precedes the attention launches on a stream
        # of its own; 0: every launch of a step on the caller's stream; 1: the acceptance after the attention on a third
        # stream as well; 3: only that.  MEASURED (bench.py, two lanes, ms per round, two runs each): 0: 6.465 / 6.485,
        # 1: 6.615 / 6.619 (the attention launches themselves slow down: 0.714 -> 0.70 of the HBM peak), 2: 6.409 / 6.436,
        # 3: 6.493 / 6.482.
        self._phase_streams = int(os.environ.get("AIC_ENGINE_PHASE_STREAMS", "2"))
        self.sp = 1 if ulysses is None else ulysses.sp_size
        self.suffix_cache = SuffixCache(spec.suffix_cache_max_depth) if (
            spec.enable_suffix_decoding or spec.method == "suffix") else None
        self._sharded = None
        if suffix_owner is not None and self.suffix_cache is not None:
            from .suffix_sharding import RankOwnedSuffix
            self._sharded = RankOwnedSuffix(self.suffix_cache, int(suffix_owner[0]), int(suffix_owner[1]), suffix_owner[2])
        s = shape
        self.hq_local = s.num_q_heads // self.sp
        self.hkv_local = max(1, s.num_kv_heads // self.sp)
        self.blocks_per_seq = (max_model_len + s.block_size - 1) // s.block_size
        nb = max_num_seqs * self.blocks_per_seq
        g = torch.Generator(device=self.device).manual_seed(seed)
        # per-layer paged KV caches [2, num_blocks, block_size, Hkv_local, D] (llama_swiftkv.py:617 layout)
        self.kv_cache_dtype = kv_cache_dtype
        self.kv_scale = None
        if kv_cache_dtype == "auto":
            self.kv = [torch.empty(2, nb, s.block_size, self.hkv_local, s.head_size, dtype=torch.bfloat16,
                                   device=self.device).normal_(generator=g) for _ in range(s.num_layers)]
        else:   # OCP e4m3 cache (A16's fp8 path): random bytes re-interpreted, NaN codes cleared
            assert kv_cache_dtype in ("fp8", "fp8_e4m3")
            self.kv = []
            for _ in range(s.num_layers):
                raw = torch.randint(0, 256, (2, nb, s.block_size, self.hkv_local, s.head_size), dtype=torch.uint8,
                                    device=self.device, generator=g)
                raw[(raw & 0x7f) == 0x7f] = 0x30
                self.kv.append(raw.view(torch.float8_e4m3fn))
            self.kv_scale = torch.full((1,), 0.02, dtype=torch.float32, device=self.device)
        self.max_tokens = max_num_seqs * (MAX_SPEC_LEN + 1)
        tq = self.max_tokens + 16   # + room for the SP padding of the token count
        # synthetic outputs of the target model's dense layers (real shapes)
        self.q_buf = torch.empty(tq, s.num_q_heads * s.head_size, dtype=torch.bfloat16, device=self.device).normal_(generator=g)
        self.k_buf = torch.empty(tq, s.num_kv_heads * s.head_size, dtype=torch.bfloat16, device=self.device).normal_(generator=g)
        self.v_buf = torch.empty(tq, s.num_kv_heads * s.head_size, dtype=torch.bfloat16, device=self.device).normal_(generator=g)
        self.hidden = torch.empty(tq, s.hidden_size, dtype=torch.bfloat16, device=self.device).normal_(generator=g)
        self.logits = torch.empty(tq, s.vocab_size, dtype=torch.bfloat16, device=self.device).normal_(generator=g)
        self.attn_out = torch.empty(tq, self.hq_local * s.head_size, dtype=torch.bfloat16, device=self.device)
        perm = torch.randperm(nb, generator=torch.Generator().manual_seed(seed)).numpy().astype(np.int32)
        self._free_blocks = [perm[i * self.blocks_per_seq:(i + 1) * self.blocks_per_seq] for i in range(max_num_seqs)]
        self._bt_host = np.stack(self._free_blocks)       # [max_num_seqs, blocks_per_seq]: a slot's pages never change
        self.requests: List[Optional[RequestState]] = [None] * max_num_seqs
        # vLLM's input_batch.token_ids_cpu / num_tokens: prompt + sampled tokens per slot (the suffix patterns are slices
        # of it), and the spec token ids scheduled for the next step
        self.token_ids_cpu = np.zeros((max_num_seqs, max_model_len + MAX_SPEC_LEN + 2), dtype=np.int32)
        self.num_tokens = np.zeros(max_num_seqs, dtype=np.int32)
        self.num_prompt = np.zeros(max_num_seqs, dtype=np.int32)
        self.n_draft = np.zeros(max_num_seqs, dtype=np.int32)
        self.draft_ids = np.zeros((max_num_seqs, MAX_SPEC_LEN), dtype=np.int32)
        self.draft_row = np.full(max_num_seqs, -1, dtype=np.int64)    # row of the in-flight LSTM output, -1: ids are on the host
        self._pending: List[Optional[_PendingDrafts]] = [None] * max_num_seqs
        self._req_ids: List = [None] * max_num_seqs
        self.block_table = torch.from_numpy(self._bt_host).to(self.device)   # int32 [max_num_seqs, blocks_per_seq]
        self.sm_scale = s.head_size ** -0.5
        self._plant_col = torch.full((self.max_tokens, 1), 30.0, dtype=torch.bfloat16, device=self.device)
        self.stats = StepStats()
        # Synthetic target only (bench.py's draft-model leg): with this probability, independently per
        # draft position, the target's arg-max token at that position IS the draft token (planted on the device, where
        # the draft model's ids live); otherwise it is the ground-truth stream's token, which a random-weight draft
        # model never hits.  0 = the stream's token everywhere (model-free scoring of suffix drafts).
        self.plant_draft_prob = 0.0
        self._plant_gen = torch.Generator(device=self.device).manual_seed(seed + 12345)
        self.last_suffix_stats: Dict = {}
        self.timeline: Dict[str, float] = {}   # seconds accumulated per phase (host clock)

    # -- request management ---------------------------------------------------------------------------
    def _admit(self, slot: int, req_id, prompt_arr: np.ndarray, gen: np.ndarray) -> RequestState:
        r = RequestState(self, slot, req_id, len(prompt_arr))
        self.requests[slot] = r
        self._req_ids[slot] = req_id
        n = len(prompt_arr)
        self.token_ids_cpu[slot, :n] = prompt_arr
        self.token_ids_cpu[slot, n:n + len(gen)] = gen
        self.num_tokens[slot] = n + len(gen)
        self.num_prompt[slot] = n
        self.n_draft[slot] = 0
        self.draft_row[slot] = -1
        self._pending[slot] = None
        return r

    def add_request(self, slot: int, req_id, prompt: Sequence[int], first_token) -> None:
        """Admit a request whose prompt has been prefilled (its KV is taken as resident) and whose first
        token has been sampled (`first_token`: that token, or the list of tokens generated so far for a request
        that joins mid-generation).  Mirrors the first pass through _update_suffix_cache (:657-673); the prompt tree
        is built on a host thread while the next step's attention runs (SuffixCache.cache_prompt_async)."""
        old = self.requests[slot]
        gen = np.asarray([first_token] if np.isscalar(first_token) else first_token, dtype=np.int32).reshape(-1)
        prompt_arr = np.asarray(prompt, dtype=np.int32)
        if self._sharded is not None:
            self._sharded.admit(slot, req_id, prompt_arr, gen, old_req_id=None if old is None else old.req_id)
        elif self.suffix_cache is not None:
            if old is not None and self.suffix_cache.has_cached_prompt(old.req_id):
                self.suffix_cache.evict_prompt(old.req_id)   # model_runner.py:675-678
            self.suffix_cache.cache_prompt_async(req_id, prompt_arr, gen)
        self._admit(slot, req_id, prompt_arr, gen)

    def add_requests(self, slots, req_ids, prompts, first_tokens, n_threads: int = 8) -> None:
        if self.suffix_cache is not None:
            for s in slots:
                old = self.requests[s]
                if old is not None and self.suffix_cache.has_cached_prompt(old.req_id):
                    self.suffix_cache.evict_prompt(old.req_id)
            if self._sharded is not None:
                self._sharded.admit_many(list(slots), list(req_ids), prompts, first_tokens, n_threads=n_threads)
                for s, rid, p, ft in zip(slots, req_ids, prompts, first_tokens):
                    gen = np.asarray([ft] if np.isscalar(ft) else ft, dtype=np.int32).reshape(-1)
                    self._admit(s, rid, np.asarray(p, dtype=np.int32), gen)
                return
            self.suffix_cache.cache_prompts(list(req_ids), [list(p) for p in prompts], n_threads=n_threads)
        for s, rid, p, ft in zip(slots, req_ids, prompts, first_tokens):
            gen = np.asarray([ft] if np.isscalar(ft) else ft, dtype=np.int32).reshape(-1)
            if self.suffix_cache is not None:
                self.suffix_cache.update_response(rid, gen.tolist())
            self._admit(s, rid, np.asarray(p, dtype=np.int32), gen)

    # -- one engine step --------------------------------------------------------------------------------
    def step(self, next_truth) -> List[List[int]]:
        """`next_truth(req, n)` -> the target model's greedy tokens for the next n positions of `req`
        (the synthetic target: its verify logits get these tokens planted as arg-max).
        Returns the tokens emitted per live request."""
        ctx = self.begin(next_truth)
        return [] if ctx is None else self.finish(ctx)

    # A step has a device half and a host half.  begin() builds the step's geometry and enqueues everything the GPU does
    # (KV write, attention, acceptance, accepted tokens on their way to the host); finish() waits for the accepted
    # tokens and runs the host chain (parse, tree updates, suffix proposal, merge).  Called back to back they are one
    # step.  A driver may also split the live requests into two LANES (disjoint slot sets, each with its own staging
    # buffers and events) and interleave them — begin(A); finish(B); begin(B); finish(A); ... — so that one lane's
    # host chain runs while the GPU attends for the other (bench.py --lanes 2).  Each lane is then its own sequence of
    # engine steps over its requests: the reference's per-step rules (update all, then propose; the draft-model merge
    # rule) apply per lane step, as they would if vLLM had scheduled those requests in that step.
    def _lane(self, lane: int):
        lanes = self.__dict__.setdefault("_lanes", {})
        if lane not in lanes:
            from types import SimpleNamespace
            L = SimpleNamespace(stage={}, out_pin=None, out_ev=None, lstm_prev=None, lstm_pin=None, lstm_flip=0,
                                suffix_won_last=False)
            # pinned buffers are allocated here, not at first use: hipHostMalloc takes milliseconds, and a lane's first
            # draft-model step may come long after its first step
            L.out_pin = torch.empty(self.max_num_seqs * (MAX_SPEC_LEN + 2), dtype=torch.int32).pin_memory()
            L.out_ev = torch.cuda.Event()
            L.prep_ev, L.attn_ev = torch.cuda.Event(), torch.cuda.Event()     # phase streams (begin())
            if self.drafter is not None:
                k = self.spec.num_speculative_tokens
                L.lstm_pin = [torch.empty(self.max_num_seqs, k, dtype=torch.int64).pin_memory() for _ in range(2)]
                # the device-side fill of pending draft ids uses torch kernels nothing else here uses; their first launch
                # loads the code object (tens of ms, measured inside a 20-step bench): do it now
                idx = torch.zeros(2, dtype=torch.int64, device=self.device)
                torch.zeros(4, dtype=torch.int32, device=self.device).index_copy_(
                    0, idx, torch.zeros(4, dtype=torch.int64, device=self.device).index_select(0, idx).to(torch.int32))
            for which in ("A", "B"):
                pin = torch.empty(1 << 17, dtype=torch.uint8).pin_memory()
                L.stage[which] = (pin, torch.empty(pin.numel(), dtype=torch.uint8, device=self.device))
            # scratch of the native index build / parse (aic_step_build, aic_step_parse)
            L.offs_a, L.offs_b = np.zeros(5, np.int64), np.zeros(7, np.int64)
            L.totals, L.ctx_sum = np.zeros(8, np.int64), np.zeros(1, np.int64)
            L.n_emit = np.zeros(self.max_num_seqs, np.int32)
            L.flat_emit = np.zeros(self.max_num_seqs * (MAX_SPEC_LEN + 2), np.int32)
            L.parse_total = np.zeros(1, np.int64)
            lanes[lane] = L
        return lanes[lane]

    def begin(self, next_truth, only_slots: Optional[Sequence[int]] = None, lane: int = 0):
        """Device half of a step over the live requests (of `only_slots`, default all).  Returns the step's context for
        finish(), or None when there is nothing to do."""
        from types import SimpleNamespace
        import time as _t
        _t0 = _t.perf_counter()
        def _mark(name, _state=[_t0]):
            now = _t.perf_counter()
            self.timeline[name] = self.timeline.get(name, 0.0) + (now - _state[0])
            _state[0] = now
        L = self._lane(lane)
        s, spec, dev = self.shape, self.spec, self.device
        rq = self.requests            # the list is the scheduler's: a slot set to None has left the batch
        cand = range(len(rq)) if only_slots is None else only_slots
        live = np.fromiter((i for i in cand if rq[i] is not None), dtype=np.int64)
        B = len(live)
        if B == 0:
            return None
        reqs = [rq[i] for i in live]
        n_draft = self.n_draft[live]
        G = self.hq_local // self.hkv_local
        prev_lstm = L.lstm_prev
        if not self._native_index:
            return self._begin_numpy(next_truth, L, live, reqs, n_draft, G, _mark)
        # every index array of the step in ONE native call, written straight into the lane's two pinned staging buffers
        # (A: what the KV write and the attention launches need; B: what only the acceptance needs)
        pinA, devA = L.stage["A"]
        pinB, devB = L.stage["B"]
        lib = N.lib()
        while True:
            rc = lib.aic_step_build(B, live.ctypes.data, self.max_num_seqs, self.num_tokens.ctypes.data, self.n_draft.ctypes.data,
                                    self.draft_ids.ctypes.data, MAX_SPEC_LEN,
                                    self.draft_row.ctypes.data if prev_lstm is not None else None,
                                    0 if prev_lstm is None else int(prev_lstm.shape[1]), self._bt_host.ctypes.data,
                                    self.blocks_per_seq, s.block_size, G, pinA.data_ptr(), pinA.numel(), pinB.data_ptr(),
                                    pinB.numel(), L.offs_a.ctypes.data, L.offs_b.ctypes.data, L.totals.ctypes.data,
                                    L.ctx_sum.ctypes.data)
            if rc == 0:
                break
            if rc != N.AIC_ERR_BUFFER_TOO_SMALL:
                N.check(rc)
            need = {"A": int(L.totals[6]), "B": int(L.totals[7])}      # filled whether or not the buffers were large enough
            for which in ("A", "B"):       # grow to what the step needs (at least double) and retry (first steps only)
                if need[which] > L.stage[which][0].numel():
                    pin = torch.empty(max(need[which], 2 * L.stage[which][0].numel()), dtype=torch.uint8).pin_memory()
                    L.stage[which] = (pin, torch.empty(pin.numel(), dtype=torch.uint8, device=dev))
            pinA, devA = L.stage["A"]
            pinB, devB = L.stage["B"]
        T, max_q, max_ctx, n_short_reqs, D, F, bytes_a, bytes_b = (int(x) for x in L.totals)
        oa, ob = L.offs_a, L.offs_b
        if getattr(self, "qlen_hist", None) is not None:      # diagnostic (bench.py --qlen-hist): query lengths seen
            self.qlen_hist += np.bincount(n_draft + 1, minlength=len(self.qlen_hist))[:len(self.qlen_hist)]
            if getattr(self, "mix_log", None) is not None:   # which long drafts share a lane step (the attention call's mix)
                self.mix_log.append(tuple(sorted(int(q) + 1 for q in n_draft if (int(q) + 1) * G > 32)))
        # Phase streams (interleaved lanes only): the step's small launches before the attention — staging copy, block-table
        # gather, KV write — are each a few microseconds of GPU work with 15-40 us of queue latency between them
        # (rocprofv3 timeline, profiles/r03_step_timeline.txt: 0.29 ms of idle queue per round).  On a stream of their own
        # they run beside the OTHER lane's attention launches.  The same for the acceptance behind the attention was
        # measured and is not the default (see __init__).
        main = torch.cuda.current_stream()
        phased = self._phase_streams and len(self.__dict__.get("_lanes", ())) > 1
        prep = self._side_stream("prep") if (phased and self._phase_streams != 3) else main
        view = lambda buf, off, n, dt, isz: buf[off:off + n * isz].view(dt)
        with torch.cuda.stream(prep):
            devA[:bytes_a].copy_(pinA[:bytes_a], non_blocking=True)
            d_seq = view(devA, int(oa[0]), B, torch.int32, 4)
            d_qsl = view(devA, int(oa[1]), B + 1, torch.int32, 4)
            slots = view(devA, int(oa[2]), B, torch.int64, 8)
            d_slots = view(devA, int(oa[3]), T, torch.int64, 8)
            order_dev = view(devA, int(oa[4]), B, torch.int32, 4)
            bt = self.block_table.index_select(0, slots) if B != self.max_num_seqs else self.block_table
            self._write_kv(d_slots, T)
            if prep is not main:
                L.prep_ev.record()
        if prep is not main:
            if bt is not self.block_table:
                bt.record_stream(main)       # allocated on the prep stream, read by the attention launches
            main.wait_event(L.prep_ev)
        _mark('host_prepare')
        self._req_split = (order_dev[:n_short_reqs], n_short_reqs, order_dev[n_short_reqs:], B - n_short_reqs)
        self._stream = int(main.cuda_stream)   # looked up once per step, not once per layer
        self._attention_layers(T, bt, d_seq, d_qsl, max_q, max_ctx)
        self.last_ctx_sum = int(L.ctx_sum[0])
        _mark('enqueue_attention')
        post = main
        if phased and self._phase_streams != 2:
            L.attn_ev.record()
            post = self._side_stream("post")
            post.wait_event(L.attn_ev)       # the acceptance follows the step's attention (and the previous draft-model run)
        with torch.cuda.stream(post):
            return self._begin_tail(next_truth, L, live, reqs, B, T, D, F, n_draft, max_q, bytes_b, prev_lstm, _mark, main, post)

    def _side_stream(self, name: str):
        streams = self.__dict__.setdefault("_side_streams", {})
        if name not in streams:
            streams[name] = torch.cuda.Stream(device=self.device)
        return streams[name]

    def _begin_tail(self, next_truth, L, live, reqs, B, T, D, F, n_draft, max_q, bytes_b, prev_lstm, _mark, main, post):
        """begin() after the attention launches, on the stream the caller made current: staging B, acceptance, the copy of
        the accepted tokens to the host."""
        pinB, devB = L.stage["B"]
        ob = L.offs_b
        view = lambda buf, off, n, dt, isz: buf[off:off + n * isz].view(dt)
        if self.ulysses is not None and self.ulysses.pending_reshard is not None:
            # an SP step's hidden states are on their way over xGMI on the resharding stream (ulysses.ReshardStream): the
            # acceptance and the draft model read them — join here, behind everything enqueued since the layers' launches
            self.ulysses.pending_reshard.wait()
            self.ulysses.pending_reshard = None
        # ---- staging B (the GPU is busy with the attention launches from here on): only the synthetic target's tokens
        # are missing from it — row (request i, position p) gets the target's token for that position
        ql = (n_draft + 1).tolist()
        plant = pinB.numpy()[int(ob[2]):int(ob[2]) + 8 * T].view(np.int64)
        at = 0
        for i, r in enumerate(reqs):
            plant[at:at + ql[i]] = next_truth(r, ql[i])
            at += ql[i]
        devB[:bytes_b].copy_(pinB[:bytes_b], non_blocking=True)
        d_draft = view(devB, int(ob[0]), D, torch.int32, 4)
        d_cu = view(devB, int(ob[1]), B, torch.int32, 4)
        d_plant = view(devB, int(ob[2]), T, torch.int64, 8)
        d_trows = view(devB, int(ob[3]), D, torch.int64, 8)
        d_brows = view(devB, int(ob[4]), B, torch.int64, 8)
        if F:
            d_fpos, d_fsrc = view(devB, int(ob[5]), F, torch.int64, 8), view(devB, int(ob[6]), F, torch.int64, 8)
            d_draft.index_copy_(0, d_fpos, prev_lstm.reshape(-1).index_select(0, d_fsrc).to(torch.int32))
        if self.plant_draft_prob > 0.0 and D:
            # target row j verifies draft j (d_trows and d_draft are both in request order)
            hit = torch.rand(D, device=self.device, generator=self._plant_gen) < self.plant_draft_prob
            d_plant.index_copy_(0, d_trows, torch.where(hit, d_draft.to(torch.int64), d_plant.index_select(0, d_trows)))
        _mark('stage_acceptance')
        c = self._begin_accept(L, live, reqs, B, T, n_draft, max(max_q - 1, 1), d_draft, d_cu, d_plant, d_trows, d_brows,
                               _mark)
        if post is not main:
            # the acceptance's outputs live in the post stream's pool; the draft model reads them on the main stream (after
            # the host has waited for out_ev)
            for t in (c.rej.last_token, c.rej.hidden_index):
                if t is not None:
                    t.record_stream(main)
            c.draft_stream = post if c.lstm_out is not None else None
        return c

    def _begin_numpy(self, next_truth, L, live, reqs, n_draft, G, _mark):
        """begin() with the index arrays built by numpy (AIC_ENGINE_NUMPY=1: the A/B reference of the native build)."""
        s, dev = self.shape, self.device
        B = len(live)
        q_len = n_draft + 1
        T = int(q_len.sum())
        qsl = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(q_len, out=qsl[1:])
        # context after this step's tokens are written: everything sampled so far + the drafts
        ntok = self.num_tokens[live]
        ctx = ntok + n_draft
        max_q, max_ctx = int(q_len.max()), int(ctx.max())
        if getattr(self, "qlen_hist", None) is not None:      # diagnostic (bench.py --qlen-hist): query lengths seen
            self.qlen_hist += np.bincount(q_len, minlength=len(self.qlen_hist))[:len(self.qlen_hist)]
            if getattr(self, "mix_log", None) is not None:   # which long drafts share a lane step (the attention call's mix)
                self.mix_log.append(tuple(sorted(int(q) for q in q_len if q * G > 32)))

        # Two staging copies per step.  (A) what the KV write and the attention launches need — contexts, query offsets,
        # slots, the short / long request lists — goes first, and the 2 x L launches are enqueued right behind it.  (B) what
        # only the acceptance needs — draft ids, the synthetic target's tokens, target / bonus row indices — is built
        # while the GPU is already attending (rocprofv3 timeline of the r02 bench: 0.58 ms of GPU idle sat between the
        # suffix kernels of one step and the staging copy of the next; building B first was a third of it).
        slot_map = self._slot_mapping(live, ntok, q_len, qsl, T)
        order, n_short_reqs = ops.split_order(q_len, G)            # short / long request lists of the attention call
        stA = self._stage(L, "A", [(ctx, np.int32), (qsl, np.int32), (live, np.int64), (slot_map, np.int64), (order, np.int32)])
        d_seq, d_qsl, slots, d_slots, order_dev = stA
        bt = self.block_table.index_select(0, slots) if B != self.max_num_seqs else self.block_table

        # (a) KV of the step's tokens for every layer in one launch (A16), then (b) verify attention per layer
        self._write_kv(d_slots, T)
        _mark('host_prepare')
        self._req_split = (order_dev[:n_short_reqs], n_short_reqs, order_dev[n_short_reqs:], B - n_short_reqs)
        self._stream = int(torch.cuda.current_stream().cuda_stream)   # looked up once per step, not once per layer
        self._attention_layers(T, bt, d_seq, d_qsl, max_q, max_ctx)
        self.last_ctx_sum = int(ctx.sum())
        _mark('enqueue_attention')

        # ---- staging B (the GPU is busy with the attention launches from here on) ----
        # planted verify logits: row (request i, position p) gets the target's token for that position
        ql = q_len.tolist()
        plant_tok = np.concatenate([next_truth(r, ql[i]) for i, r in enumerate(reqs)]).astype(np.int64, copy=False)
        # draft ids: suffix drafts are on the host; LSTM drafts of the previous step are still on the device (their
        # host copy is in flight and nobody waits for it here): placeholders now, filled on the device below
        draft_flat = self.draft_ids[live][np.arange(MAX_SPEC_LEN)[None, :] < n_draft[:, None]]
        cu_draft = np.cumsum(n_draft)
        fill_pos = fill_src = np.zeros(0, np.int64)
        prev_lstm = L.lstm_prev
        if prev_lstm is not None:
            pend_rows = self.draft_row[live]
            sel = np.nonzero(pend_rows >= 0)[0]
            if len(sel):
                k_sel = n_draft[sel].astype(np.int64)
                rep = np.repeat(np.arange(len(sel)), k_sel)
                within = np.arange(int(k_sel.sum())) - np.repeat(np.cumsum(k_sel) - k_sel, k_sel)
                fill_pos = (cu_draft[sel] - k_sel)[rep] + within
                fill_src = pend_rows[sel][rep] * prev_lstm.shape[1] + within
        # target rows = all but the last row of each request; bonus row = the last one (model_runner.py:394-404)
        is_bonus = np.zeros(T, dtype=bool)
        is_bonus[qsl[1:] - 1] = True
        target_rows = np.nonzero(~is_bonus)[0]
        bonus_rows = qsl[1:] - 1
        d_draft, d_cu, d_plant, d_trows, d_brows, d_fpos, d_fsrc = self._stage(
            L, "B", [(draft_flat, np.int32), (cu_draft, np.int32), (plant_tok, np.int64), (target_rows, np.int64),
                  (bonus_rows, np.int64), (fill_pos, np.int64), (fill_src, np.int64)])
        # the LSTM draft ids of the previous step (still on the device) go into this step's draft array
        if len(fill_pos):
            d_draft.index_copy_(0, d_fpos, prev_lstm.reshape(-1).index_select(0, d_fsrc).to(torch.int32))
        if self.plant_draft_prob > 0.0 and len(target_rows):
            hit = torch.rand(len(target_rows), device=self.device, generator=self._plant_gen) < self.plant_draft_prob
            d_plant.index_copy_(0, d_trows, torch.where(hit, d_draft.to(torch.int64), d_plant.index_select(0, d_trows)))
        _mark('stage_acceptance')
        return self._begin_accept(L, live, reqs, B, T, n_draft, int(max(n_draft.max(), 1)), d_draft, d_cu, d_plant, d_trows,
                                  d_brows, _mark)

    def _begin_accept(self, L, live, reqs, B, T, n_draft, max_spec, d_draft, d_cu, d_plant, d_trows, d_brows, _mark):
        """(c)-(d) of begin(): acceptance on the planted verify logits, the accepted tokens on their way to the host, the
        draft model behind them."""
        from types import SimpleNamespace
        spec = self.spec
        # (c) verify logits: plant, accept, un-plant
        lg = self.logits[:T]
        col = d_plant.unsqueeze(1)
        saved = torch.gather(lg, 1, col)
        lg.scatter_(1, col, self._plant_col[:T])   # gather/scatter on device tensors only: nothing here may sync the stream
        # greedy sampler on the bonus rows and the target rows of the acceptance: both read in place from the [T, V] logits
        # through their row indices (bonus_logits_indices / target_logits_indices) inside one launch
        rej = ops.rejection_sample(lg, d_draft, d_cu, None, max_spec, target_row_index=d_trows, bonus_row_index=d_brows)
        lg.scatter_(1, col, saved)

        # (d) accepted tokens start their way to the host (pinned buffer, event) BEFORE the LSTM draft is
        # enqueued: the draft needs nothing from the host (last token / hidden row come from the acceptance
        # kernel on the device), so it runs while the host parses tokens and updates the suffix trees
        out_pin = L.out_pin[:B * (max_spec + 1)].view(B, max_spec + 1)   # contiguous: a strided pinned target makes the copy blocking
        out_pin.copy_(rej.output_token_ids, non_blocking=True)
        L.out_ev.record()
        lstm_out = None
        use_lstm = spec.method in ("arctic", "mlp_speculator") and self.drafter is not None and B <= spec.disable_by_batch_size
        # Under the reference's rule (SpecConfig.draft_model_per_request = False) the draft model's output is dropped
        # whenever suffix decoding takes any request of the step, which the host only learns after the suffix round trip.
        # Steps follow each other closely in what they do, so: if the previous step used the draft model, enqueue it
        # now (it runs while the host updates the trees) and drop it should suffix decoding win; if the previous step
        # was taken by suffix decoding, wait for the suffix result and run the draft model only if nobody was taken.
        # With interleaved lanes the wait costs nothing (the GPU is busy with the other lane), while a dropped launch is
        # 0.45 ms of GPU time — and after a step without a suffix winner the next one usually has one: never early there.
        interleaved = len(self.__dict__.get("_lanes", ())) > 1
        early_lstm = use_lstm and (spec.draft_model_per_request or self.suffix_cache is None or
                                   (not L.suffix_won_last and not interleaved))
        if early_lstm:
            lstm_out = self.drafter.generate_proposals(rej.last_token, self.hidden, spec.num_speculative_tokens,
                                                       hidden_index=rej.hidden_index)
        _mark('enqueue_accept_and_draft')
        return SimpleNamespace(lane=L, live=live, reqs=reqs, B=B, n_draft=n_draft, out_pin=out_pin, rej=rej, lstm_out=lstm_out,
                               use_lstm=use_lstm, draft_stream=None)

    def finish(self, c) -> List[List[int]]:
        """Host half of the step begun as `c`: returns the tokens emitted per request of that step."""
        import time as _t
        _t0 = _t.perf_counter()
        def _mark(name, _state=[_t0]):
            now = _t.perf_counter()
            self.timeline[name] = self.timeline.get(name, 0.0) + (now - _state[0])
            _state[0] = now
        s, spec, dev = self.shape, self.spec, self.device
        L, live, reqs, B, n_draft, out_pin, rej, lstm_out, use_lstm = (c.lane, c.live, c.reqs, c.B, c.n_draft, c.out_pin, c.rej,
                                                                      c.lstm_out, c.use_lstm)
        req_ids = [self._req_ids[i] for i in live.tolist()]
        if self.suffix_cache is not None and B <= spec.disable_by_batch_size:
            # the trees went cold during the step: read what the update will touch while the GPU still works
            # (after the wait instead, the same reads cost 0.055 ms and buy back no more than that)
            self.suffix_cache.warm(req_ids)
        L.out_ev.synchronize()                           # the step's only blocking wait before the proposals
        out_host = out_pin.numpy()
        _mark('wait_gpu_accept')

        # (e) host: parse, commit, update the suffix trees while the LSTM kernels run
        if self._native_index:
            # parse_output (:456-459) + the commit into token_ids_cpu / num_tokens (:469-486), one native call
            N.check(N.lib().aic_step_parse(B, live.ctypes.data, self.max_num_seqs, out_host.ctypes.data, out_host.shape[1], s.vocab_size,
                                           self.token_ids_cpu.ctypes.data, self.token_ids_cpu.shape[1],
                                           self.num_tokens.ctypes.data, L.n_emit.ctypes.data, L.flat_emit.ctypes.data,
                                           L.parse_total.ctypes.data))
            n_total = int(L.parse_total[0])
            n_emit = L.n_emit[:B].copy()
            flat_emit = L.flat_emit[:n_total].copy()
            ntok = self.num_tokens[live]
            within = None
        else:
            valid = (out_host != -1) & (out_host < s.vocab_size)                  # parse_output (:456-459)
            n_emit = valid.sum(axis=1).astype(np.int32)
            flat_emit = out_host[valid]                                            # row-major: request by request
            n_total = len(flat_emit)
            ntok = self.num_tokens[live]
            first = np.cumsum(n_emit) - n_emit
            within = np.arange(n_total) - np.repeat(first, n_emit)
            self.token_ids_cpu[np.repeat(live, n_emit), np.repeat(ntok, n_emit) + within] = flat_emit
            ntok = ntok + n_emit
            self.num_tokens[live] = ntok
        self.n_draft[live] = 0
        self.draft_row[live] = -1
        emitted = Emitted(flat_emit, n_emit)
        self.stats.emitted += n_total
        had = n_draft > 0
        self.stats.num_drafts += int(had.sum())
        self.stats.drafted += int(n_draft.sum())
        self.stats.accepted += int((n_emit[had] - 1).sum())
        if B > spec.disable_by_batch_size:
            return emitted
        _mark('host_parse')
        suffix = None
        # `end_idx` of the proposers (runner_logic.py): the row end as committed above, or — "reference" indexing — the
        # emitted ids counted a second time and written again behind themselves (model_runner.py:696-709)
        end_prop = ntok
        if self._indexing == INDEXING_REFERENCE:
            end_prop = ntok + n_emit
            if within is None:
                within = np.arange(n_total) - np.repeat(np.cumsum(n_emit) - n_emit, n_emit)
            pos = np.repeat(ntok, n_emit) + within
            keep = pos < self.max_model_len                  # :701-707: the write is cut at max_model_len
            self.token_ids_cpu[np.repeat(live, n_emit)[keep], pos[keep]] = flat_emit[keep]
        if self.suffix_cache is not None:
            if self._sharded is not None:
                self._sharded.update(req_ids, flat_emit, n_emit)
            else:
                self.suffix_cache.update_responses(req_ids, flat_emit, n_emit)   # _update_suffix_cache (:657-678)
            _mark('host_suffix_update')
            # the tree mirror update + match kernels need nothing from the main stream (their input is the host
            # tree): on a side stream they run beside the LSTM draft instead of queueing behind it
            if not hasattr(self, "_suffix_stream"):
                self._suffix_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._suffix_stream):
                suffix = self._propose_suffix(live, req_ids, end_prop, n_emit)
            _mark('suffix_speculate_roundtrip')
        # (f) merge (:555-566, :595-601).  The LSTM tokens start their copy to the host (the reference's `.cpu()`,
        # arctic_proposer.py:166) but nothing here waits for it: which requests take the LSTM draft, and how many
        # tokens, is known from the suffix result alone, and the next step fills the ids in on the device.
        min_score = 0 if spec.method == "suffix" else spec.num_speculative_tokens
        took = None
        if suffix is not None:
            took = (suffix[1] > 0) & (suffix[2] >= min_score)
        suffix_won = bool(took is not None and took.any())
        L.suffix_won_last = suffix_won
        if use_lstm and not spec.draft_model_per_request:
            if suffix_won:
                self.stats.draft_model_dropped += lstm_out is not None
                lstm_out = None                       # model_runner.py:616-618: no draft-model proposal this step
            elif lstm_out is None:
                lstm_out = self.drafter.generate_proposals(rej.last_token, self.hidden, spec.num_speculative_tokens,
                                                           hidden_index=rej.hidden_index)
        self.stats.steps += 1
        self.stats.draft_model_steps += lstm_out is not None
        pin = ev = None
        if lstm_out is not None:
            if L.lstm_pin is None or L.lstm_pin[0].shape[1] != lstm_out.shape[1]:
                L.lstm_pin = [torch.empty(self.max_num_seqs, lstm_out.shape[1], dtype=lstm_out.dtype).pin_memory()
                              for _ in range(2)]
            L.lstm_flip ^= 1
            pin = L.lstm_pin[L.lstm_flip][:B]              # two buffers: the previous step's copy may still be unread
            # (an early draft ran on begin()'s post stream: its copy goes behind it there)
            early_on = c.draft_stream if (lstm_out is c.lstm_out and c.draft_stream is not None) else torch.cuda.current_stream()
            with torch.cuda.stream(early_on):
                pin.copy_(lstm_out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
        L.lstm_prev = lstm_out
        _mark('host_draft_copy')
        room = np.maximum(self.max_model_len - end_prop - 1, 0)
        if suffix_won:
            idx = np.flatnonzero(took)
            rows = live[idx]
            w = min(suffix[0].shape[1], MAX_SPEC_LEN)
            self.draft_ids[rows, :w] = suffix[0][idx, :w]
            self.n_draft[rows] = np.minimum(suffix[1][idx], room[idx])
            self.stats.suffix_used += len(idx)
        if pin is not None:
            # the draft model's length clamp is ONE value for the batch in the reference (the running minimum of
            # propose_arctic_draft_token_ids, model_runner.py:629-641); the per-request extension clamps per request
            k_batch = spec.num_speculative_tokens
            if not spec.draft_model_per_request:
                k_batch = max(min(k_batch, self.max_model_len - int(end_prop.max()) - 1), 0)
            gets = n_emit > 0
            if took is not None:
                gets &= ~took
            idx = np.flatnonzero(gets)
            ks = np.minimum(k_batch, room[idx]).astype(np.int32)
            idx, ks = idx[ks > 0], ks[ks > 0]
            rows = live[idx]
            self.n_draft[rows] = ks
            self.draft_ids[rows, :spec.num_speculative_tokens] = 0      # placeholders until the copy lands
            self.draft_row[rows] = idx
            pend = _PendingDrafts(self, pin, ev, rows, idx, ks)
            for slot in rows.tolist():
                self._pending[slot] = pend
        _mark('host_merge')
        return emitted

    # -- pieces -------------------------------------------------------------------------------------------
    _TORCH_OF = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}

    def _stage(self, lane, which: str, arrays):
        """One pinned buffer + ONE host->device copy for a group of small index arrays; returns their device views.
        Buffers belong to a lane: a lane's previous copy has completed (its step was finished) before it stages again."""
        bufs = lane.stage
        offs, nbytes = [], 0
        for a, k in arrays:
            nbytes = (nbytes + 15) & ~15
            offs.append(nbytes)
            nbytes += len(a) * np.dtype(k).itemsize
        pin, dev = bufs.get(which, (None, None))
        if pin is None or pin.numel() < nbytes:
            pin = torch.empty(max(nbytes * 2, 1 << 16), dtype=torch.uint8).pin_memory()
            dev = torch.empty(pin.numel(), dtype=torch.uint8, device=self.device)
            bufs[which] = (pin, dev)
        host = pin.numpy()
        for (a, k), o in zip(arrays, offs):
            host[o:o + len(a) * np.dtype(k).itemsize].view(k)[:] = a
        dev[:nbytes].copy_(pin[:nbytes], non_blocking=True)
        return [dev[o:o + len(a) * np.dtype(k).itemsize].view(self._TORCH_OF[np.dtype(k)]) for (a, k), o in zip(arrays, offs)]

    def _propose_suffix(self, live, req_ids, end, n_emit):
        """propose_suffix_draft_token_ids (model_runner.py:680-744) for the whole batch at once, on arrays.
        `end` = tokens per request after this step's.  Returns (tokens [B, cap], n_tokens [B], score [B]); requests that
        are skipped keep n_tokens = 0."""
        cfg = self.spec
        B = len(live)
        end = end.astype(np.int64)
        depth = cfg.suffix_cache_max_depth
        ask = (n_emit > 0) & (end < self.max_model_len)
        n_tok = np.zeros(B, np.int32)
        score = np.zeros(B, np.float32)
        toks = np.zeros((B, 1), np.int32)
        everyone = bool(ask.all())
        where = np.arange(B) if everyone else np.nonzero(ask)[0]
        if len(where) == 0:
            return toks, n_tok, score
        e = end if everyone else end[where]
        size = np.minimum(e, depth)
        rows = live if everyone else live[where]
        if int(size.min()) == depth:      # the usual case: every pattern is the full last `depth` tokens
            idx = (e - depth)[:, None] + np.arange(depth)[None, :]
            flat = self.token_ids_cpu[rows[:, None], idx].reshape(-1)
        else:
            flat = np.concatenate([self.token_ids_cpu[rw, x - z:x] for rw, x, z in zip(rows, e, size)])
        mst = np.minimum(min(MAX_SPEC_LEN, depth), self.max_model_len - e - 1).astype(np.int32)
        nq = len(where)
        consts = self.__dict__.setdefault("_suffix_consts", {})
        if nq not in consts:
            consts[nq] = (np.full(nq, cfg.suffix_max_spec_factor, np.float32), np.full(nq, cfg.suffix_max_spec_offset, np.float32),
                          np.full(nq, cfg.suffix_min_token_prob, np.float32), np.ones(nq, np.int32))
        asked_ids = req_ids if everyone else [req_ids[i] for i in where]
        if self._sharded is not None:
            # rank-owned prompt trees: this rank speculates for the slots it owns, one all-reduce brings the rest
            o_tok, o_n, o_sc = self._sharded.propose(rows.tolist(), asked_ids, flat, size.astype(np.int32), mst, *consts[nq][:3])
        else:
            o_tok, _, o_n, o_sc, _ = self.suffix_cache.speculate_batch_arrays(asked_ids, flat, size.astype(np.int32), mst,
                                                                              *consts[nq])
        if everyone:
            toks, n_tok, score = o_tok, o_n.astype(np.int32, copy=False), o_sc.astype(np.float32, copy=False)
        else:
            toks = np.zeros((B, o_tok.shape[1]), np.int32)
            toks[where] = o_tok
            n_tok[where] = o_n
            score[where] = o_sc
        self.last_suffix_stats = self.suffix_cache.last_stats()
        return toks, n_tok, score

    def _slot_mapping(self, live, ntok, q_len, qsl, T) -> np.ndarray:
        """KV slot of every token of the step (one vectorised pass: row i covers positions first_i .. first_i + q_len_i - 1,
        first_i = position of the last sampled token, which is not in the cache yet)."""
        bs = self.shape.block_size
        rep = np.repeat(np.arange(len(live)), q_len)
        pos = (ntok.astype(np.int64) - 1)[rep] + (np.arange(T) - qsl[:-1][rep])
        return self._bt_host[live[rep], pos // bs].astype(np.int64) * bs + pos % bs

    def _write_kv(self, d_slots, T) -> None:
        s = self.shape
        # synthetic K/V of the new tokens: one [T, L * Hkv_local * D] activation like SwiftKV's fused projection
        n = self.hkv_local * s.head_size
        if not hasattr(self, "_kv_new"):
            g = torch.Generator(device=self.device).manual_seed(1)
            self._kv_new = torch.empty(2, self.max_tokens, s.num_layers * n, dtype=torch.bfloat16,
                                       device=self.device).normal_(generator=g)
            self._one = [torch.ones(1, device=self.device) for _ in range(s.num_layers)]
            self._kc = [kv[0] for kv in self.kv]
            self._vc = [kv[1] for kv in self.kv]
            scales = self._one if self.kv_scale is None else [self.kv_scale] * s.num_layers
            self._kv_writer = ops.KvBulkWriter(self._kc, self._vc, self.kv_cache_dtype, scales, scales, self.hkv_local, s.head_size)
        self._kv_writer(self._kv_new[0, :T], self._kv_new[1, :T], d_slots)

    def _attention_layers(self, T, bt, d_seq, d_qsl, max_q, max_ctx) -> None:
        s = self.shape
        if self.ulysses is None:
            q = self.q_buf[:T].view(T, s.num_q_heads, s.head_size)
            out = self.attn_out[:T].view(T, s.num_q_heads, s.head_size)
            # one plan per step (the layers share the batch geometry), one foreign call per layer
            plan = ops.VerifyAttentionPlan(q, out, self.kv[0][0], bt, d_seq, d_qsl, max_q, max_ctx, self.sm_scale,
                                           req_split=self._req_split, k_scale=self.kv_scale, v_scale=self.kv_scale,
                                           stream=self._stream)
            plan.run_layers(self.layer_tables(plan))     # all layers: one foreign call
        else:
            self.ulysses.attention_layers(self, T, bt, d_seq, d_qsl, max_q, max_ctx)

    def layer_tables(self, plan):
        """Pointer tables of the per-layer K / V caches (fixed for the engine's life)."""
        if not hasattr(self, "_layer_tables"):
            self._layer_tables = plan.layer_tables([kv[0] for kv in self.kv], [kv[1] for kv in self.kv])
        return self._layer_tables
