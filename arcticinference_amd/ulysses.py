"""Ulysses sequence parallelism + shift parallelism, host side.

Reference: /root/reference/arctic_inference/vllm/ulysses.py
  * process-group layout  ExternalDP x DP x PP x SP x TP            (:150-281)  -> rank_groups()
  * head counts per rank, KV replication when Hkv < SP               (:56-105, :432-455) -> local_heads()
  * attention wrapper: pack -> all-to-all -> attention -> all-to-all -> unpack   (:457-519) -> UlyssesAttention
  * shift parallelism: below a token threshold run the TP = SP*TP replica instead (model_runner.py:57-81,
    :237-247); both layouts own the same KV-cache head slice ("KV-cache invariance")
    -> use_shift_model(), sp_tp_head_slice()

MI355X notes.  One process per GPU; the collectives are torch.distributed calls, i.e. RCCL over xGMI.  The
two dominant collectives are equal-split all-to-alls, which map 1:1 onto the fully connected xGMI mesh
(each peer message rides its own link), so they are issued as single `all_to_all_single` calls on
contiguous buffers; the copies around them are ONE fused HIP kernel each (csrc/ulysses_pack.hip) that
writes the send layout / reads the receive layout directly, and the attention kernel consumes q straight
out of the receive buffer through a token stride — the reference's split() views cost nothing here
either, but its cat/transpose/reshape chains (3-4 copy kernels per layer) are gone.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch


# --------------------------------------------------------------------------------------------------
# group algebra (pure; tested on CPU)
# --------------------------------------------------------------------------------------------------
def rank_groups(world_size: int, dp: int, pp: int, sp: int, tp: int, num_kv_heads: Optional[int] = None
                ) -> Dict[str, List[List[int]]]:
    """Rank lists of every group the reference creates (ulysses.py:160-281).  Global rank of coordinate
    (e, d, p, s, t) is ((((e*dp + d)*pp + p)*sp + s)*tp + t): TP fastest, then SP."""
    assert world_size % (dp * pp * sp * tp) == 0
    ext = world_size // (dp * pp * sp * tp)
    dims = (ext, dp, pp, sp, tp)

    def rank(e, d, p, s, t):
        return (((e * dp + d) * pp + p) * sp + s) * tp + t

    def groups(vary: Tuple[int, ...], order: Optional[Tuple[int, ...]] = None):
        """All groups obtained by varying the axes in `vary` (iteration order `order`, last fastest)."""
        order = order or vary
        fixed = [a for a in range(5) if a not in vary]
        out = []

        def rec_fixed(i, coord):
            if i == len(fixed):
                members = []

                def rec_vary(j, c):
                    if j == len(order):
                        members.append(rank(*c))
                        return
                    ax = order[j]
                    for v in range(dims[ax]):
                        c2 = list(c)
                        c2[ax] = v
                        rec_vary(j + 1, c2)
                rec_vary(0, list(coord))
                out.append(members)
                return
            ax = fixed[i]
            for v in range(dims[ax]):
                c2 = list(coord)
                c2[ax] = v
                rec_fixed(i + 1, c2)
        rec_fixed(0, [0] * 5)
        return out

    g = {
        "TP": groups((4,)),
        "PP": groups((2,)),
        "DP": groups((1,)),
        "EP": groups((1, 4)),          # DP x TP (ulysses.py:200-207)
        "SP": groups((3,)),
        # full-TP group of shift parallelism: TP-major / SP-minor member order ("transpose(3, 4) for the
        # correct attn head order", :225-229): index in group = t * SP + s
        "SP_TP": groups((3, 4), order=(4, 3)),
    }
    if num_kv_heads is not None and num_kv_heads < sp:
        # KV-replicated variant (:251-281): SP = SP_AA (size Hkv, all-to-all of kv heads) x SP_AG (all-gather)
        aa, ag = num_kv_heads, sp // num_kv_heads
        assert aa * ag == sp
        sp_aa, sp_ag = [], []
        for e in range(ext):
            for d in range(dp):
                for p in range(pp):
                    for t in range(tp):
                        for j in range(ag):
                            sp_aa.append([rank(e, d, p, i * ag + j, t) for i in range(aa)])
                        for i in range(aa):
                            sp_ag.append([rank(e, d, p, i * ag + j, t) for j in range(ag)])
        g["SP_AA"], g["SP_AG"] = sp_aa, sp_ag
    return g


@dataclass
class LocalHeads:
    num_q_heads: int
    num_kv_heads: int
    kv_replicated: bool


def local_heads(num_q_heads: int, num_kv_heads: int, sp: int, tp: int = 1, shift_mode: bool = False) -> LocalHeads:
    """Heads a rank owns inside attention.  SP mode divides by SP after TP (:432-455); shift mode is a plain
    TP = SP*TP model.  Either way the rank reads the same KV-cache head slice."""
    ways = sp * tp
    q = num_q_heads // ways
    if num_kv_heads >= ways:
        return LocalHeads(q, num_kv_heads // ways, False)
    return LocalHeads(q, 1, True)


def use_shift_model(num_tokens: int, sp: int, enable_shift_parallel: bool, threshold: int = 512) -> bool:
    """model_runner.py:237-239."""
    return sp > 1 and enable_shift_parallel and num_tokens <= threshold


def pad_tokens_for_sp(num_tokens: int, sp: int) -> int:
    """Tokens are rounded up to a multiple of SP before being split across ranks (model_runner.py:240-247)."""
    return (num_tokens + sp - 1) // sp * sp


def sp_tp_head_slice(num_heads: int, sp: int, tp: int, sp_rank: int, tp_rank: int) -> Tuple[int, int]:
    """[first, last) q-head range owned by (tp_rank, sp_rank): TP-major / SP-minor, identical for the SP
    layout and for the shift (TP over SP_TP) layout — the KV-cache invariance the shift model relies on."""
    per = num_heads // (sp * tp)
    idx = tp_rank * sp + sp_rank
    return idx * per, (idx + 1) * per


def sp_local_head_range(num_heads_in_tp_shard: int, sp: int, sp_rank: int) -> Tuple[int, int]:
    """[first, last) of the TP shard's q heads that SP rank `sp_rank` attends with AFTER the Ulysses all-to-all: the pack
    (`[n, SP, hq*D] -> [SP, n, hq*D]`, ulysses.py:493-499) sends head chunk r of every token to rank r, so the order is
    rank-major.  Per-head parameters that vLLM shards over TP only (gpt-oss attention sinks) are sliced with it."""
    per = num_heads_in_tp_shard // sp
    return sp_rank * per, (sp_rank + 1) * per


# --------------------------------------------------------------------------------------------------
# attention wrapper
# --------------------------------------------------------------------------------------------------
def _hip_pack(q, k, v, sp):
    from . import ops
    return ops.ulysses_pack_qkv(q, k, v, sp)


def _hip_unpack(c, sp):
    from . import ops
    return ops.ulysses_unpack_out(c, sp)


def _hip_pack_pair(a, b, parts):
    from . import ops
    return ops.ulysses_pack_pair(a, b, parts)


def _hip_reorder_split(gathered, sp, order):
    from . import ops
    return ops.ulysses_reorder_split_kv(gathered, sp, order)


# the copy kernels a UlyssesAttention uses unless told otherwise; CPU (gloo) tests put torch expressions here
# (pack q|k|v, unpack out; KV-replicated variant: pack one or two tensors part-major, reorder + split the gathered K|V)
PACK_FNS = [_hip_pack, _hip_unpack, _hip_pack_pair, _hip_reorder_split]


class UlyssesAttention:
    """pack -> all-to-all -> `attn` on (all tokens x local heads) -> all-to-all -> unpack (ulysses.py:491-519).

    `attn(q_, k_, v_)` receives strided views into the receive buffer: q_ [N, hq*D], k_/v_ [N, hkv*D] with
    row stride (hq + 2 hkv) * D, and returns [N, hq*D] contiguous.  `pack` / `unpack` default to the HIP
    kernels; the CPU (gloo) tests inject torch expressions so the group / index logic runs without a GPU.

    KV-replicated variant (fewer kv heads than SP ranks, ulysses.py:437-451,462-490): every rank attends with ONE kv
    head.  q takes the usual all-to-all over SP; K/V take an all-to-all inside the rank's SP_AA group (size = number of
    kv heads: shards the heads) followed by an all-gather inside its SP_AG group (size SP / kv heads: collects the
    tokens), and the gathered token chunks are put back into rank order.  Pass `kv_groups=(aa_group, aa_size,
    ag_group, ag_size)`; its copies are HIP kernels as well (pack_pair / reorder_split_kv, csrc/ulysses_pack.hip)."""

    def __init__(self, sp_size: int, group, num_q_heads_local: int, num_kv_heads_local: int, head_size: int,
                 pack: Optional[Callable] = None, unpack: Optional[Callable] = None, all_to_all: Optional[Callable] = None,
                 kv_groups: Optional[tuple] = None, all_gather: Optional[Callable] = None):
        self.sp_size, self.group = sp_size, group
        self.hq, self.hkv, self.D = num_q_heads_local, num_kv_heads_local, head_size
        self.pack, self.unpack = pack or PACK_FNS[0], unpack or PACK_FNS[1]
        # `all_to_all(recv, send[, group])`: torch.distributed over the SP group unless a stand-in is injected (the
        # single-GPU shape rehearsal of bench.py --rehearse-sp copies send to recv)
        self._a2a = all_to_all
        self._ag = all_gather
        self.kv_groups = kv_groups
        if kv_groups is not None:
            _, aa, _, ag = kv_groups
            assert aa * ag == sp_size and num_kv_heads_local == 1
            # chunk c = j * aa + i of the gathered sequence holds the tokens of SP rank i * ag + j (ulysses.py:449-451)
            self.order = [j * aa + i for i in range(aa) for j in range(ag)]

    def _all_to_all(self, recv, send, group):
        import torch.distributed as dist
        if self._a2a is not None:
            self._a2a(recv, send) if group is self.group else self._a2a(recv, send, group)
        else:
            from .dist_utils import all_to_all_single
            all_to_all_single(recv, send, group=group)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn: Callable) -> torch.Tensor:
        import torch.distributed as dist
        if self.sp_size == 1:
            return attn(query, key, value)
        qw, kw = self.hq * self.D, self.hkv * self.D
        if self.kv_groups is not None:
            aa_group, aa, ag_group, ag = self.kv_groups
            sp = self.sp_size
            pack_pair, reorder_split = PACK_FNS[2], PACK_FNS[3]
            # q: all-to-all 1/2 over SP (ulysses.py:463-469)
            q = pack_pair(query, None, sp)
            q_ = torch.empty_like(q)
            self._all_to_all(q_, q, self.group)
            # K/V: heads sharded inside SP_AA, tokens collected inside SP_AG, chunks back in rank order (:470-490)
            kv = pack_pair(key, value, aa)
            kv_part = torch.empty_like(kv)
            self._all_to_all(kv_part, kv, aa_group)
            kv_ = torch.empty(q_.shape[0], 2 * kw, dtype=query.dtype, device=query.device)
            if self._ag is not None:
                self._ag(kv_, kv_part, ag_group)
            else:
                from .dist_utils import all_gather_into_tensor
                all_gather_into_tensor(kv_, kv_part, group=ag_group)
            k_, v_ = reorder_split(kv_, sp, self.order)
            c_ = attn(q_, k_, v_)
        else:
            send = self.pack(query, key, value, self.sp_size)               # [SP*n, qw + 2kw], rank-major
            recv = torch.empty_like(send)
            self._all_to_all(recv, send, self.group)                        # C1 (ulysses.py:502)
            q_, k_, v_ = recv[:, :qw], recv[:, qw:qw + kw], recv[:, qw + kw:]
            c_ = attn(q_, k_, v_)                                           # all N tokens, local heads
        c = torch.empty_like(c_)
        self._all_to_all(c, c_, self.group)                                 # C2 (ulysses.py:514)
        return self.unpack(c, self.sp_size)                                 # [n, SP*qw]


class UlyssesContext:
    """SP state of one rank for the stand-alone engine (arcticinference_amd/engine.py): the per-layer
    attention of a step runs on this rank's head slice with the two all-to-alls around it."""

    def __init__(self, sp_size: int, sp_rank: int, group, shape, device="cuda", all_to_all: Optional[Callable] = None,
                 enable_shift_parallel: bool = False, shift_parallel_threshold: int = 512):
        self.sp_size, self.sp_rank, self.group = sp_size, sp_rank, group
        # shift parallelism (model_runner.py:57-81,237-239): steps of at most `threshold` tokens run the TP = SP x TP
        # replica, whose attention sees all tokens x this rank's heads straight from the TP-sharded projections (no
        # all-to-all); the KV cache is the same tensor in both modes (same head slice per rank)
        self.enable_shift_parallel = enable_shift_parallel
        self.shift_parallel_threshold = shift_parallel_threshold
        self.steps_sp = 0
        self.steps_shift = 0
        lh = local_heads(shape.num_q_heads, shape.num_kv_heads, sp_size)
        self.heads = lh
        kv_groups, all_gather = None, None
        if lh.kv_replicated:
            # fewer kv heads than ranks: SP = SP_AA (kv heads, all-to-all) x SP_AG (all-gather), ulysses.py:251-281
            aa, ag = shape.num_kv_heads, sp_size // shape.num_kv_heads
            aa_group = ag_group = None
            if group is not None and all_to_all is None:
                import torch.distributed as dist
                g = rank_groups(sp_size, 1, 1, sp_size, 1, num_kv_heads=shape.num_kv_heads)
                for ranks in g["SP_AA"]:          # every rank creates every group, in the same order
                    h = dist.new_group(ranks)
                    if sp_rank in ranks:
                        aa_group = h
                for ranks in g["SP_AG"]:
                    h = dist.new_group(ranks)
                    if sp_rank in ranks:
                        ag_group = h
            else:                                 # single-process rehearsal: the collectives are local copies
                all_gather = lambda out, inp, _group: out.copy_(inp.repeat(ag, 1))
            kv_groups = (aa_group, aa, ag_group, ag)
        self.attn = UlyssesAttention(sp_size, group, lh.num_q_heads, lh.num_kv_heads, shape.head_size, all_to_all=all_to_all,
                                     kv_groups=kv_groups, all_gather=all_gather)
        # the SP -> replicated resharding of a step's hidden states runs on a side stream (ReshardStream); the engine joins
        # it (pending_reshard.wait()) before anything reads the gathered rows
        self.reshard = ReshardStream(sp_size, sp_rank, group if all_to_all is None else None, device)
        self.pending_reshard: Optional[ReshardHandle] = None

    def attention_layers(self, eng, T, bt, d_seq, d_qsl, max_q, max_ctx) -> None:
        from . import ops
        s = eng.shape
        sp = self.sp_size
        Tp = pad_tokens_for_sp(T, sp)
        n = Tp // sp
        lo = self.sp_rank * n
        D = s.head_size
        hq, hkv = self.heads.num_q_heads, self.heads.num_kv_heads
        if use_shift_model(T, sp, self.enable_shift_parallel, self.shift_parallel_threshold):
            # shift (TP) mode: all tokens x local heads, as the TP-sharded qkv projection of the shift replica
            # hands them over; no repartition, no collective inside attention
            self.steps_shift += 1
            h0, h1 = sp_tp_head_slice(s.num_q_heads, sp, 1, self.sp_rank, 0)
            q_loc = eng.q_buf[:T].view(T, s.num_q_heads, D)[:, h0:h1]          # strided view, row stride Hq * D
            out = eng.attn_out[:T].view(T, hq, D)
            plan = ops.VerifyAttentionPlan(q_loc, out, eng.kv[0][0], bt, d_seq, d_qsl, max_q, max_ctx, eng.sm_scale,
                                           req_split=eng._req_split, k_scale=eng.kv_scale, v_scale=eng.kv_scale,
                                           stream=eng._stream)
            plan.run_layers(eng.layer_tables(plan))
            return
        self.steps_sp += 1
        # this rank's token slice x all heads, as the dense layers of the target would hand it over
        q = eng.q_buf[lo:lo + n]
        k = eng.k_buf[lo:lo + n]
        v = eng.v_buf[lo:lo + n]

        def attn(q_, k_, v_, layer):
            kv = eng.kv[layer]
            out = eng.attn_out[:Tp].view(Tp, hq, D)
            qv = q_.unflatten(1, (hq, D))   # strided view into the all-to-all receive buffer
            ops.verify_attention(qv[:T], kv[0], kv[1], bt, d_seq, d_qsl, max_q, max_ctx, eng.sm_scale, out=out[:T],
                                 req_split=eng._req_split, k_scale=eng.kv_scale, v_scale=eng.kv_scale, stream=eng._stream)
            return eng.attn_out[:Tp]

        for layer in range(s.num_layers):
            self.attn.forward(q, k, v, lambda a, b, c, L=layer: attn(a, b, c, L))
        # C6: the model's output rows of this rank's token slice, resharded over SP so that sampling and the draft model
        # see every sampled token (ulysses_forward, model_runner.py:202-209) — on the side stream, joined by the engine
        # before the acceptance; with the step's sampled rows known (every row of a verify step is one) the row form
        # (a verify step samples from every row, so the stand-alone engine asks for all of them: the full form)
        self.pending_reshard = self.reshard.gather(eng.hidden[lo:lo + n], Tp)

    def gather_hidden(self, local_rows: torch.Tensor, num_tokens: int) -> torch.Tensor:
        """[N/SP, hidden] -> [N, hidden] over the SP group (one all-gather per step)."""
        out = torch.empty((num_tokens, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        if self.group is None:          # single-process rehearsal: the collective is a local copy
            out.copy_(local_rows.repeat(self.sp_size, 1))
        else:
            from .dist_utils import all_gather_into_tensor
            all_gather_into_tensor(out, local_rows.contiguous(), group=self.group)
        return out


# --------------------------------------------------------------------------------------------------
# SP -> replicated resharding of a step's hidden states, on a side stream
# --------------------------------------------------------------------------------------------------
class ReshardHandle:
    """A resharding in flight: `wait()` orders the CURRENT stream behind it and hands back the tensor."""

    def __init__(self, out: torch.Tensor, event, keep=()):
        self._out, self._event, self._keep = out, event, keep

    def wait(self) -> torch.Tensor:
        if self._event is not None:
            torch.cuda.current_stream(self._out.device).wait_event(self._event)
            self._event = None
        self._keep = ()
        return self._out


class ReshardStream:
    """The layout change at the end of a Ulysses step — every rank holds the model's output rows of ITS token slice
    ([N/SP, hidden]) and sampling / the draft model need rows of every slice — issued on a side HIP stream of its own
    (BASELINE north_star: "shift-parallel TP<->SP resharding on a side stream").  The reference does it as one synchronous
    all-gather of ALL N rows on the compute stream (ulysses_forward, model_runner.py:202-209), whatever the step samples.

    gather(local_rows, num_tokens)            full form: all_gather_into_tensor -> [N, hidden], the reference's bytes.
    gather(local_rows, num_tokens, rows=idx)  row form: only the global rows `idx` (the step's logits_indices; int64, on
        the device) -> [len(idx), hidden].  Every rank picks the rows of `idx` that lie in its slice (zero rows for the
        others) and ONE all-reduce over the SP group sums them; a row has exactly one owner, and the sum runs on the
        rows' int32 bit patterns, so the result is the owner's bits, exactly (no float add, -0.0 stays -0.0).  Bytes per
        rank: ~2 R H instead of the all-gather's (SP - 1) (N / SP) H — a 32K-token prefill chunk of 8192 tokens, 64
        requests, hidden 8192 (BASELINE configs[3]): 2 MB instead of 117 MB over xGMI per step.

    Both return a ReshardHandle at once: the collective (and, in the row form, the row pick) runs on the side stream
    behind an event recorded on the caller's stream, and `handle.wait()` joins the caller's stream behind it — whatever the
    caller enqueues in between (the next step's staging copies, the other lane's launches, the suffix-tree mirror
    update) overlaps with the xGMI traffic.  On CPU tensors (gloo tests) and with group = None (single-process rehearsal:
    local copies) the same calls run synchronously.
    """

    def __init__(self, sp_size: int, sp_rank: int, group, device=None):
        self.sp_size, self.sp_rank, self.group = sp_size, sp_rank, group
        self.device = torch.device(device) if device is not None else None
        self._stream = None
        self.calls = {"full": 0, "rows": 0}
        self.bytes_moved = 0          # payload this rank contributed / received (diagnostics, not timing)

    def _side(self, device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def gather(self, local_rows: torch.Tensor, num_tokens: int, rows: Optional[torch.Tensor] = None) -> ReshardHandle:
        sp, n = self.sp_size, local_rows.shape[0]
        assert n * sp == num_tokens, "the token count must be padded to a multiple of SP (pad_tokens_for_sp)"
        dev = local_rows.device
        side = self._side(dev)
        event = None
        if side is not None:
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(dev))
            side.wait_event(fork)
        ctx = torch.cuda.stream(side) if side is not None else _NullContext()
        with ctx:
            # the row form pays when its all-reduce (~2 R rows per rank) moves less than the all-gather ((SP - 1) N / SP rows)
            by_rows = rows is not None and 2 * rows.numel() < (sp - 1) * n
            if not by_rows:
                self.calls["full"] += 1
                out = torch.empty((num_tokens, local_rows.shape[1]), dtype=local_rows.dtype, device=dev)
                src = local_rows.contiguous()
                if self.group is None:
                    out.copy_(src.repeat(sp, 1))
                else:
                    from .dist_utils import all_gather_into_tensor
                    all_gather_into_tensor(out, src, group=self.group)
                self.bytes_moved += out.numel() * out.element_size()
                keep = (src,)
                if rows is not None:
                    full, out = out, out.index_select(0, rows)
                    keep = (src, full)
            else:
                self.calls["rows"] += 1
                assert rows.dtype == torch.int64 and rows.device == dev
                lo = self.sp_rank * n
                mine = (rows >= lo) & (rows < lo + n)
                picked = local_rows.index_select(0, (rows - lo).clamp_(0, n - 1))
                out = torch.where(mine.unsqueeze(1), picked, torch.zeros((), dtype=picked.dtype, device=dev)).contiguous()
                if self.group is not None:
                    from .dist_utils import all_reduce
                    bits = out.view(torch.int32) if (out.shape[1] * out.element_size()) % 4 == 0 else None
                    assert bits is not None, "hidden size x element size must be a multiple of 4 bytes"
                    all_reduce(bits, group=self.group)
                elif sp > 1:
                    # single-process rehearsal: the "other ranks" hold this rank's rows too
                    out = local_rows.index_select(0, rows.remainder(n))
                self.bytes_moved += 2 * out.numel() * out.element_size()
                keep = (picked, mine)
            if side is not None:
                event = torch.cuda.Event()
                event.record(side)
                for t in (out,) + tuple(keep):
                    t.record_stream(torch.cuda.current_stream(dev))
        if side is not None:
            # tensors made on the side stream's pool are handed to the caller's stream
            out.record_stream(torch.cuda.current_stream(dev))
        return ReshardHandle(out, event, keep)


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


# --------------------------------------------------------------------------------------------------
# vLLM patches: arcticinference_amd/vllm_plugin/ulysses.py (built only when vLLM is importable)
# --------------------------------------------------------------------------------------------------
def build_ulysses_patches():
    from .vllm_plugin.ulysses import build_ulysses_patches as build
    return build()
