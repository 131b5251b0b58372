"""CPU, world_size 2, 4 and 8 over gloo: the N>1 path of UlyssesAttention (pack -> all-to-all -> attention on
local heads over ALL tokens -> all-to-all -> unpack) reproduces single-process attention over all heads.
The HIP pack/unpack kernels cannot run here, so the test injects the oracle's torch expressions of
ulysses.py:493-517 for those two copies; group logic, buffer shapes, strided q/k/v views and the
collective sequence are the product's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import spec_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, Hq, Hkv, D, N, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from arcticinference_amd.ulysses import UlyssesAttention, local_heads
        g = torch.Generator().manual_seed(0)
        q = torch.randn(N, Hq * D, generator=g)
        k = torch.randn(N, Hkv * D, generator=g)
        v = torch.randn(N, Hkv * D, generator=g)
        lh = local_heads(Hq, Hkv, world)
        hq, hkv = lh.num_q_heads, lh.num_kv_heads
        n = N // world
        sl = slice(rank * n, (rank + 1) * n)

        def attn(q_, k_, v_, sinks=None):
            # plain causal attention over all N tokens with this rank's heads (fp32); inputs are strided views.
            # `sinks` [hq]: one extra soft-max term per head (gpt-oss), no value attached
            assert q_.shape == (N, hq * D) and k_.shape == (N, hkv * D)
            Q = q_.reshape(N, hq, D).transpose(0, 1)
            K = k_.reshape(N, hkv, D).transpose(0, 1).repeat_interleave(hq // hkv, 0)
            V = v_.reshape(N, hkv, D).transpose(0, 1).repeat_interleave(hq // hkv, 0)
            s = Q @ K.transpose(1, 2) / D ** 0.5
            s = s.masked_fill(~torch.tril(torch.ones(N, N, dtype=torch.bool)), float("-inf"))
            if sinks is not None:
                s = torch.cat([s, sinks.view(hq, 1, 1).expand(hq, N, 1)], dim=-1)
                return (torch.softmax(s, -1)[..., :N] @ V).transpose(0, 1).reshape(N, hq * D).contiguous()
            return (torch.softmax(s, -1) @ V).transpose(0, 1).reshape(N, hq * D).contiguous()

        ua = UlyssesAttention(world, dist.group.WORLD, hq, hkv, D,
                              pack=lambda a, b, c, sp: O.ulysses_pack(a, b, c, sp, hq, hkv, D),
                              unpack=lambda c, sp: O.ulysses_unpack(c, sp, hq, D))
        out = ua.forward(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), attn)
        # single-process reference over all heads, then this rank's token slice
        ref_all = UlyssesAttention(1, None, Hq, Hkv, D).forward(q, k, v, lambda a, b, c: _full(a, b, c, Hq, Hkv, D, N))
        ok = torch.allclose(out, ref_all[sl], atol=1e-5)
        # shift (TP) mode of the same rank: all tokens x the rank's heads, no collective; the head slice and the KV
        # head slice are the ones SP mode uses (KV-cache invariance of shift parallelism, model_runner.py:57-81)
        from arcticinference_amd.ulysses import sp_tp_head_slice, use_shift_model
        h0, h1 = sp_tp_head_slice(Hq, world, 1, rank, 0)
        assert (h1 - h0) == hq and use_shift_model(N, world, True, 512) and not use_shift_model(N, world, True, N - 1)
        kv0 = h0 // (Hq // Hkv)
        q_loc = q.view(N, Hq, D)[:, h0:h1].reshape(N, hq * D)
        k_loc = k.view(N, Hkv, D)[:, kv0:kv0 + hkv].reshape(N, hkv * D)
        v_loc = v.view(N, Hkv, D)[:, kv0:kv0 + hkv].reshape(N, hkv * D)
        shift = attn(q_loc, k_loc, v_loc)
        ok = ok and torch.allclose(shift, ref_all.view(N, Hq, D)[:, h0:h1].reshape(N, hq * D), atol=1e-5)
        # per-head attention sinks under Ulysses.  Every head gets a DISTINCT sink; the rank attends with the
        # slice the patched Attention layer takes (vllm_plugin/ulysses.py::_arctic_sinks -> sp_local_head_range); a wrong
        # head-to-rank mapping mis-normalises the heads of every rank > 0.  Dense reference: all heads, all sinks.
        from arcticinference_amd.ulysses import sp_local_head_range
        sinks_all = torch.linspace(-2.0, 3.0, Hq)
        s0, s1 = sp_local_head_range(Hq, world, rank)
        assert (s0, s1) == (h0, h1)            # the same heads in SP mode and in shift mode (TP = 1 here)
        out_s = ua.forward(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(),
                           lambda a, b, c: attn(a, b, c, sinks_all[s0:s1]))
        ref_s = _full(q, k, v, Hq, Hkv, D, N, sinks_all)
        ok = ok and torch.allclose(out_s, ref_s[sl], atol=1e-5) and not torch.allclose(ref_s, ref_all, atol=1e-3)
        if world > 1:                           # and the mapping matters: the neighbouring rank's slice is wrong
            w0, w1 = sp_local_head_range(Hq, world, (rank + 1) % world)
            bad = ua.forward(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(),
                             lambda a, b, c: attn(a, b, c, sinks_all[w0:w1]))
            ok = ok and not torch.allclose(bad, ref_s[sl], atol=1e-4)
        out_q.put((rank, bool(ok), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def _full(q, k, v, Hq, Hkv, D, N, sinks=None):
    Q = q.reshape(N, Hq, D).transpose(0, 1)
    K = k.reshape(N, Hkv, D).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
    V = v.reshape(N, Hkv, D).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
    s = Q @ K.transpose(1, 2) / D ** 0.5
    s = s.masked_fill(~torch.tril(torch.ones(N, N, dtype=torch.bool)), float("-inf"))
    if sinks is not None:
        s = torch.cat([s, sinks.view(Hq, 1, 1).expand(Hq, N, 1)], dim=-1)
        return (torch.softmax(s, -1)[..., :N] @ V).transpose(0, 1).reshape(N, Hq * D)
    return (torch.softmax(s, -1) @ V).transpose(0, 1).reshape(N, Hq * D)


@pytest.mark.parametrize("world,Hq,Hkv", [(2, 8, 2), (4, 8, 4), (8, 32, 8)])     # (8, 32, 8): Llama-3.1-8B heads at SP = 8, one kv head per rank
def test_ulysses_attention_gloo(world, Hq, Hkv):
    ctx = mp.get_context("spawn")
    qout = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, Hq, Hkv, 16, 8 * world, qout)) for r in range(world)]
    for p in procs:
        p.start()
    res = [qout.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert all(r[2] == (8, Hq * 16) for r in res)


def _worker_kv_replicated(rank, world, port, Hq, Hkv, D, N, out_q):
    """Fewer kv heads than ranks (ulysses.py:437-451,462-490): q all-to-all over SP, K/V all-to-all inside SP_AA +
    all-gather inside SP_AG + chunk reorder; every rank then attends with one kv head over all tokens."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from arcticinference_amd.ulysses import UlyssesAttention, local_heads, rank_groups
        g = torch.Generator().manual_seed(1)
        q = torch.randn(N, Hq * D, generator=g)
        k = torch.randn(N, Hkv * D, generator=g)
        v = torch.randn(N, Hkv * D, generator=g)
        lh = local_heads(Hq, Hkv, world)
        assert lh.kv_replicated and lh.num_kv_heads == 1
        hq = lh.num_q_heads
        aa, ag = Hkv, world // Hkv
        groups = rank_groups(world, 1, 1, world, 1, num_kv_heads=Hkv)
        aa_group = ag_group = None
        for ranks in groups["SP_AA"]:
            h = dist.new_group(ranks)
            if rank in ranks:
                aa_group = h
        for ranks in groups["SP_AG"]:
            h = dist.new_group(ranks)
            if rank in ranks:
                ag_group = h
        n = N // world
        sl = slice(rank * n, (rank + 1) * n)
        seen = {}

        def attn(q_, k_, v_):
            assert q_.shape == (N, hq * D) and k_.shape == (N, D) and v_.shape == (N, D)
            seen["k"] = k_.clone()
            Q = q_.reshape(N, hq, D).transpose(0, 1)
            K = k_.reshape(N, 1, D).transpose(0, 1).expand(hq, N, D)
            V = v_.reshape(N, 1, D).transpose(0, 1).expand(hq, N, D)
            s = Q @ K.transpose(1, 2) / D ** 0.5
            s = s.masked_fill(~torch.tril(torch.ones(N, N, dtype=torch.bool)), float("-inf"))
            return (torch.softmax(s, -1) @ V).transpose(0, 1).reshape(N, hq * D).contiguous()

        import arcticinference_amd.ulysses as U
        U.PACK_FNS[2], U.PACK_FNS[3] = O.ulysses_pack_pair, O.ulysses_reorder_split    # (no GPU here: the literal expressions)
        ua = UlyssesAttention(world, dist.group.WORLD, hq, 1, D, unpack=lambda c, sp: O.ulysses_unpack(c, sp, hq, D),
                              kv_groups=(aa_group, aa, ag_group, ag))
        out = ua.forward(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), attn)
        ref_all = _full(q, k, v, Hq, Hkv, D, N)
        ok = torch.allclose(out, ref_all[sl], atol=1e-5)
        # the rank saw ALL tokens, in order, of its one kv head (= rank // ag)
        ok = ok and torch.equal(seen["k"], k.view(N, Hkv, D)[:, rank // ag])
        out_q.put((rank, bool(ok), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def test_ulysses_kv_replicated_gloo():
    world, Hq, Hkv, D = 4, 8, 2, 16
    ctx = mp.get_context("spawn")
    qout = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_kv_replicated, args=(r, world, port, Hq, Hkv, D, 8 * world, qout)) for r in range(world)]
    for p in procs:
        p.start()
    res = [qout.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert all(r[2] == (8, Hq * D) for r in res)
