#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (not part of the bench contract; used while tuning)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from arcticinference_amd import ops
from arcticinference_amd import _native as N

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def attn(B=64, ctx=4096, qlen=4, Hq=32, Hkv=8, D=128, bs=16, layers=4, split=False, kv8=False):
    nblk = (ctx + bs - 1) // bs
    nb = B * nblk
    kvs = [torch.randn(2, nb, bs, Hkv, D, device=dev, dtype=torch.bfloat16) for _ in range(layers)]
    kw = {}
    if kv8:   # e4m3 cache: random codes without the NaN patterns
        kvs = []
        for _ in range(layers):
            raw = torch.randint(0, 256, (2, nb, bs, Hkv, D), dtype=torch.uint8, device=dev)
            raw[(raw & 0x7f) == 0x7f] = 0x30
            kvs.append(raw.view(torch.float8_e4m3fn))
        sc = torch.full((1,), 0.02, dtype=torch.float32, device=dev)
        kw = dict(k_scale=sc, v_scale=sc)
    bt = torch.randperm(nb, device=dev).to(torch.int32).view(B, nblk)
    T = B * qlen
    q = torch.randn(T, Hq, D, device=dev, dtype=torch.bfloat16)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    qsl = (torch.arange(B + 1, device=dev) * qlen).to(torch.int32)
    out = torch.empty_like(q)
    # split=True: go through the request-list path (all requests short) like the engine does
    rs = (torch.arange(B, dtype=torch.int32, device=dev), B, None, 0) if split else None
    if split and qlen * (Hq // Hkv) > 16:
        rs = ops.split_requests([qlen] * B, Hq // Hkv, dev)
    i = [0]

    def f():
        kv = kvs[i[0] % layers]
        i[0] += 1
        ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, qlen, ctx, D ** -0.5, out=out, req_split=rs, **kw)
    us = timeit(f)
    gb = B * ctx * 2 * Hkv * D * (1 if kv8 else 2) / 1e9
    print(f"attn{' fp8kv' if kv8 else ''} B={B} ctx={ctx} qlen={qlen} Hq={Hq} Hkv={Hkv} D={D}: {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s (incl. combine)")


def attn_mix(n_short=59, n_long=5, q_short=1, q_long=33, ctx=4224, Hq=32, Hkv=8, D=128, bs=16, layers=4, kv8=False):
    """The steady-state step of bench.py: most requests carry no draft (q_len 1), a few a long suffix draft — one call
    through the host-partitioned path (pair kernel) against its parts."""
    B = n_short + n_long
    nblk = (ctx + bs - 1) // bs
    nb = B * nblk
    kvs = [torch.randn(2, nb, bs, Hkv, D, device=dev, dtype=torch.bfloat16) for _ in range(layers)]
    kw = {}
    if kv8:
        kvs = []
        for _ in range(layers):
            raw = torch.randint(0, 256, (2, nb, bs, Hkv, D), dtype=torch.uint8, device=dev)
            raw[(raw & 0x7f) == 0x7f] = 0x30
            kvs.append(raw.view(torch.float8_e4m3fn))
        sc = torch.full((1,), 0.02, dtype=torch.float32, device=dev)
        kw = dict(k_scale=sc, v_scale=sc)
    bt = torch.randperm(nb, device=dev).to(torch.int32).view(B, nblk)
    ql = [q_short] * n_short + [q_long] * n_long
    T = sum(ql)
    q = torch.randn(T, Hq, D, device=dev, dtype=torch.bfloat16)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(ql)]).astype(np.int32), device=dev)
    out = torch.empty_like(q)
    rs = ops.split_requests(ql, Hq // Hkv, dev)
    i = [0]

    def f():
        kv = kvs[i[0] % layers]
        i[0] += 1
        ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, max(ql), ctx, D ** -0.5, out=out, req_split=rs, **kw)
    us = timeit(f)
    gb = B * ctx * 2 * Hkv * D * (1 if kv8 else 2) / 1e9
    print(f"attn-mix{' fp8kv' if kv8 else ''} short={n_short}x{q_short} long={n_long}x{q_long} ctx={ctx} Hq={Hq} Hkv={Hkv}: {us:8.1f} us  "
          f"{gb / us * 1e6 / 1e3:6.2f} TB/s (incl. combine)")


def attn_ql(ql, ctx=4224, Hq=32, Hkv=8, D=128, bs=16, layers=4, jitter=0):
    """One host-partitioned call over an arbitrary list of query lengths (contexts ctx - jitter .. ctx)."""
    B = len(ql)
    nblk = (ctx + bs - 1) // bs
    nb = B * nblk
    kvs = [torch.randn(2, nb, bs, Hkv, D, device=dev, dtype=torch.bfloat16) for _ in range(layers)]
    bt = torch.randperm(nb, device=dev).to(torch.int32).view(B, nblk)
    q = torch.randn(sum(ql), Hq, D, device=dev, dtype=torch.bfloat16)
    rng = np.random.default_rng(0)
    ctxs = ctx - (rng.integers(0, jitter + 1, B) if jitter else np.zeros(B, np.int64))
    seq = torch.tensor(ctxs, dtype=torch.int32, device=dev)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(ql)]).astype(np.int32), device=dev)
    out = torch.empty_like(q)
    rs = ops.split_requests(ql, Hq // Hkv, dev)
    i = [0]

    def f():
        kv = kvs[i[0] % layers]
        i[0] += 1
        ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, max(ql), ctx, D ** -0.5, out=out, req_split=rs)
    us = timeit(f)
    gb = float(ctxs.sum()) * 2 * Hkv * D * 2 / 1e9
    hist = {k: ql.count(k) for k in sorted(set(ql))}
    print(f"attn-ql {hist} ctx<={ctx} jitter={jitter} short/long={rs[1]}/{rs[3]}: {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s (incl. combine)")


def attn_trace(n_short=59, n_long=5, q_short=1, q_long=33, ctx=4224, Hq=32, Hkv=8, D=128, bs=16):
    """Where and when the workgroups of the one-grid short + long launch run (aic_debug_attn_trace)."""
    B = n_short + n_long
    nblk = (ctx + bs - 1) // bs
    nb = B * nblk
    kv = torch.randn(2, nb, bs, Hkv, D, device=dev, dtype=torch.bfloat16)
    bt = torch.randperm(nb, device=dev).to(torch.int32).view(B, nblk)
    ql = [q_short] * n_short + [q_long] * n_long
    q = torch.randn(sum(ql), Hq, D, device=dev, dtype=torch.bfloat16)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(ql)]).astype(np.int32), device=dev)
    out = torch.empty_like(q)
    rs = ops.split_requests(ql, Hq // Hkv, dev)
    run = lambda: ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, max(ql), ctx, D ** -0.5, out=out, req_split=rs)
    for _ in range(3):
        run()
    buf = torch.full((2048, 4), -1, dtype=torch.int64, device=dev)
    N.lib().aic_debug_attn_trace(buf.data_ptr(), 2048)
    run()
    torch.cuda.synchronize()
    N.lib().aic_debug_attn_trace(None, 0)
    t = buf.cpu().numpy()
    t = t[t[:, 3] >= 0]
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0       # us
    hw, xcc = t[:, 2] & 0xffffffff, t[:, 2] >> 32
    cu = (xcc & 0xf) * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xf)      # (xcc, se, cu) -> one id
    print(f"trace short={n_short}x{q_short} long={n_long}x{q_long}: {len(t)} workgroups, kernel {end.max():.1f} us")
    for kind, name in ((1, "long"), (0, "short")):
        m = t[:, 3] == kind
        if not m.any():
            continue
        print(f"  {name:5s}: n={m.sum():4d}  start min/med/max {start[m].min():7.1f} {np.median(start[m]):7.1f} {start[m].max():7.1f}   "
              f"end min/med/max {end[m].min():7.1f} {np.median(end[m]):7.1f} {end[m].max():7.1f}   duration med {np.median(end[m] - start[m]):7.1f}   "
              f"distinct CUs {len(set(cu[m]))}")
    both = set(cu[t[:, 3] == 1]) & set(cu[t[:, 3] == 0])
    per_cu = {}
    for c, k in zip(cu, t[:, 3]):
        per_cu.setdefault(int(c), []).append(int(k))
    if os.environ.get("AIC_TRACE_DETAIL"):     # which short workgroups (by launch index) share a CU with a long one, and how long they ran
        long_cus = set(cu[t[:, 3] == 1].tolist())
        idx = np.nonzero(t[:, 3] == 0)[0]
        shared = np.array([int(cu[i]) in long_cus for i in idx])
        dur = (end - start)[idx]
        n = len(idx)
        print(f"  short workgroups in launch order, 16 bins: share of each bin on a CU with a long workgroup / median duration")
        for b in range(16):
            sl = slice(b * n // 16, (b + 1) * n // 16)
            print(f"    bin {b:2d}: shared {shared[sl].mean():.2f}  dur {np.median(dur[sl]):6.1f}  (max {dur[sl].max():6.1f})")
        print(f"  shared: median dur {np.median(dur[shared]) if shared.any() else 0:.1f}; alone: {np.median(dur[~shared]) if (~shared).any() else 0:.1f}")
    mix = sorted((tuple(sorted(v)) for v in per_cu.values()))
    from collections import Counter
    print(f"  CUs used {len(per_cu)}; CUs with both kinds {len(both)}; per-CU mix {Counter(mix).most_common(6)}")


def attn_phases(n_short=32, ctx=4224, kv8=False, Hq=32, Hkv=8, D=128, bs=16):
    """Where a short-only launch's time goes per workgroup (aic_debug_attn_phase_trace): entry -> request geometry known ->
    first tile requested -> [tile loop] -> partials stored, and when the workgroups start and end within the launch."""
    B = n_short
    nblk = (ctx + bs - 1) // bs
    nb = B * nblk
    if kv8:
        raw = torch.randint(0, 256, (2, nb, bs, Hkv, D), dtype=torch.uint8, device=dev)
        raw[(raw & 0x7f) == 0x7f] = 0x30
        kv = raw.view(torch.float8_e4m3fn)
        sc = torch.full((1,), 0.02, dtype=torch.float32, device=dev)
        kw = dict(k_scale=sc, v_scale=sc)
    else:
        kv = torch.randn(2, nb, bs, Hkv, D, device=dev, dtype=torch.bfloat16)
        kw = {}
    # (a second cache to evict the first from the last-level cache between launches)
    other = torch.randn(64 * 1024 * 1024, device=dev)
    bt = torch.randperm(nb, device=dev).to(torch.int32).view(B, nblk)
    q = torch.randn(B, Hq, D, device=dev, dtype=torch.bfloat16)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    qsl = torch.arange(B + 1, device=dev).to(torch.int32)
    out = torch.empty_like(q)
    rs = ops.split_requests([1] * B, Hq // Hkv, dev)
    run = lambda: ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, 1, ctx, D ** -0.5, out=out, req_split=rs, **kw)
    for _ in range(3):
        run()
    other.mul_(1.0001)
    buf = torch.zeros((4096, 8), dtype=torch.int64, device=dev)
    N.lib().aic_debug_attn_phase_trace(buf.data_ptr(), 4096)
    run()
    torch.cuda.synchronize()
    N.lib().aic_debug_attn_phase_trace(None, 0)
    t = buf.cpu().numpy()
    t = t[t[:, 0] > 0]
    us = lambda a: a / 100.0
    t0 = t[:, 0].min()
    tiles = (t[:, 6] >> 32) / 32.0
    med = lambda a: float(np.median(us(a)))
    print(f"attn-phases {'fp8' if kv8 else 'bf16'} cache, {n_short} requests x {ctx} tokens: {len(t)} workgroups, "
          f"{np.median(tiles):.0f} tiles per wave; launch (first start -> last end) {us(t[:, 5].max() - t0):.1f} us")
    print(f"   workgroup start after the first: median {med(t[:, 0] - t0):.2f} max {us((t[:, 0] - t0).max()):.2f} us")
    print(f"   entry -> request geometry known (request list, lengths): median {med(t[:, 1] - t[:, 0]):.2f} us")
    print(f"   geometry -> first tile requested (Q fragments, block table, K / V loads issued): {med(t[:, 2] - t[:, 1]):.2f} us")
    print(f"   tile loop: median {med(t[:, 4] - t[:, 2]):.2f} us = {med(t[:, 4] - t[:, 2]) / np.median(tiles):.3f} us per 32-token tile "
          f"(min {us((t[:, 4] - t[:, 2]).min()):.1f}, max {us((t[:, 4] - t[:, 2]).max()):.1f})")
    print(f"   loop end -> partials stored: {med(t[:, 5] - t[:, 4]):.2f} us;  workgroup total median {med(t[:, 5] - t[:, 0]):.1f} us; "
          f"last end - median end {us(t[:, 5].max() - np.median(t[:, 5])):.1f} us")


def attn_long_phases(B=16, q_long=33, ctx=4096, Hq=32, Hkv=8, D=128, bs=16, n_short=0):
    """Shader-clock cycle accounting of the long-draft body per wave (aic_debug_attn_phase_trace on a long-only call, or —
    n_short > 0 — on the long workgroups of a mixed call, which share their CUs with short-request workgroups)."""
    nblk = (ctx + bs - 1) // bs
    nb = (B + n_short) * nblk
    kv = torch.randn(2, nb, bs, Hkv, D, device=dev, dtype=torch.bfloat16)
    bt = torch.randperm(nb, device=dev).to(torch.int32).view(B + n_short, nblk)
    ql = [1] * n_short + [q_long] * B
    B = B + n_short
    q = torch.randn(sum(ql), Hq, D, device=dev, dtype=torch.bfloat16)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(ql)]).astype(np.int32), device=dev)
    out = torch.empty_like(q)
    rs = ops.split_requests(ql, Hq // Hkv, dev)
    run = lambda: ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, max(ql), ctx, D ** -0.5, out=out, req_split=rs)
    us = timeit(run)
    cap = 32768
    buf = torch.zeros((cap, 8), dtype=torch.int64, device=dev)     # two rows of 8 per wave
    N.lib().aic_debug_attn_phase_trace(buf.data_ptr(), cap)
    us_tr = timeit(run)
    N.lib().aic_debug_attn_phase_trace(None, 0)
    t = buf.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 7] > 0]
    n_iter, tiles = t[:, 6] >> 8, t[:, 6] & 0xff
    dur_us = (t[:, 7] - t[:, 10]) / 100.0
    ghz = float(np.median(t[:, 11] / np.maximum(dur_us, 1e-3))) / 1e3
    start = (t[:, 10] - t[:, 10].min()) / 100.0
    print(f"long-phases {n_short} short + {B - n_short} x {q_long} tokens, ctx {ctx}: call {us:.1f} us (traced build {us_tr:.1f} us); {len(t) // 4} workgroups, "
          f"{int(np.median(n_iter))} KV tiles each; shader clock {ghz:.2f} GHz; wave start after the first: median {np.median(start):.1f} "
          f"max {start.max():.1f} us; wave duration min / median / max {dur_us.min():.1f} / {np.median(dur_us):.1f} / {dur_us.max():.1f} us")
    print("   cycles per KV tile and wave, by the wave's row tiles (MFMA issue is asynchronous: a phase that needs results waits for them)")
    for k in sorted(set(tiles.tolist())):
        m = tiles == k
        it = np.maximum(n_iter[m], 1).astype(np.float64)
        f = lambda c: float(np.median(t[m, c] / it))
        print(f"   {k} row tiles ({m.sum():5d} waves): loop {f(1):6.0f} = vmcnt wait {f(8):5.0f} | barrier {f(9):5.0f} | next tile's DMA issued {f(2):5.0f} | "
              f"K reads + score MFMAs issued {f(3):5.0f} | soft-max {f(4):5.0f} | V reads + PV MFMAs issued {f(5):5.0f};  prologue {float(np.median(t[m, 0])):6.0f}")


def lstm(B, fp8=True):
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=128256, input_hidden_dim=4096)
    m = ArcticLSTMSpeculator(cfg, max_num_seqs=64, device=dev, quantize_lm_head=fp8, use_graph=False)
    m.load_weights(random_lstm_weights(cfg, seed=0).items())
    hid = torch.randn(B, 4096, device=dev, dtype=torch.bfloat16)
    ids = torch.randint(0, 128256, (B,), device=dev)
    head_gb = 128256 * 4096 * (1 if (fp8 and B <= 32) else 2) / 1e9
    gate_gb = 4 * 4096 * 4096 * 2 / 1e9
    # head-by-head schedule, the fused one with the cell as three launches (r03) and as one (r04, the default), fused +
    # on-the-fly fp8 activations (aic_debug_lstm_fused / aic_debug_lstm_cell_launches)
    for name, fused, cell in (("head-by-head", 0, 1), ("fused, 3-launch cell", 1, 3), ("fused, 1-launch cell", 1, 1), ("fused+xq", 2, 1)):
        N.lib().aic_debug_lstm_fused(fused)
        N.lib().aic_debug_lstm_cell_launches(cell)
        us = timeit(lambda: m.generate_proposals(ids, hid, 3), iters=10)
        print(f"lstm B={B} fp8={fp8 and B <= 32} {name}: {us:8.1f} us per 3-head propose; "
              f"weights {3 * (head_gb + gate_gb):.2f} GB -> {3 * (head_gb + gate_gb) / us * 1e6 / 1e3:5.2f} TB/s "
              f"({3 * (head_gb + gate_gb) / us * 1e6 / 1e3 / 8:.2f} of 8 TB/s)")
    N.lib().aic_debug_lstm_fused(1)
    N.lib().aic_debug_lstm_cell_launches(1)
    for graph in (False, True):
        m.use_graph = graph
        us = timeit(lambda: m.generate_proposals(ids, hid, 3), iters=10)
        print(f"lstm B={B} fp8={fp8 and B <= 32} fused, 1-launch cell, {'HIP graph replay' if graph else 'eager launches'}: {us:8.1f} us "
              f"({3 * (head_gb + gate_gb) / us * 1e6 / 1e3 / 8:.2f} of 8 TB/s)")
    del m
    torch.cuda.empty_cache()


def rejection(B=64, k=3, V=128256):
    logits = torch.randn(B * k, V, device=dev, dtype=torch.bfloat16)
    draft = torch.randint(0, V, (B * k,), device=dev, dtype=torch.int32)
    cu = (torch.arange(1, B + 1, device=dev) * k).to(torch.int32)
    bonus = torch.zeros(B, dtype=torch.int32, device=dev)
    us = timeit(lambda: ops.rejection_sample(logits, draft, cu, bonus, k))
    print(f"rejection B={B} k={k}: {us:8.1f} us  {B * k * V * 2 / 1e9 / us * 1e6 / 1e3:5.2f} TB/s")


if __name__ == "__main__":
    what = sys.argv[1:] or ["attn", "lstm", "rej"]
    if "attn" in what:
        attn(split=True)
        attn()
        attn(B=32)
        attn(B=8)
        attn(B=64, Hq=4, Hkv=1)        # SP=8 slice
        attn(B=64, qlen=8)
        attn(B=16, qlen=33, split=True)
        attn(B=4, qlen=33, split=True)
        attn(B=1, qlen=33, split=True)
        attn(B=64, Hq=64, Hkv=8, D=64)           # gpt-oss-120b heads (secondary head size: shared-tile body for all)
        attn(B=64, Hq=8, Hkv=1, D=64)            # its SP = 8 slice
    if "fp8" in what or "attn" in what:
        attn(split=True, kv8=True)
        attn(B=16, qlen=33, split=True, kv8=True)
    if "long" in what:
        attn(B=16, qlen=33, split=True)
        attn(B=16, qlen=17, split=True)
        attn(B=5, qlen=33, split=True)
        attn(B=1, qlen=33, split=True)
    if "layout" in what:     # the short body's layouts (heads per workgroup x cross-workgroup splits), forced one by one
        for B, combos in ((64, ((0, 0), (4, 2), (2, 1), (2, 2), (1, 1))), (32, ((0, 0), (4, 4), (2, 2), (2, 1), (1, 1), (1, 2))),
                          (16, ((0, 0), (4, 8), (2, 4), (1, 2), (1, 1)))):
            for hpw, sp in combos:
                N.lib().aic_debug_attn_layout(hpw, sp)
                print(f"layout hpw={hpw} splits={sp}: ", end="")
                attn_mix(B, 0)
        for hpw, sp in ((0, 0), (4, 2), (2, 1), (1, 1)):
            N.lib().aic_debug_attn_layout(hpw, sp)
            print(f"layout hpw={hpw} splits={sp}: ", end="")
            attn_mix(59, 5)
        for hpw in (2, 1):          # two kv heads per rank (an 8-kv-head model under SP = 4)
            N.lib().aic_debug_attn_layout(hpw, 0)
            print(f"layout hpw={hpw} splits=auto: ", end="")
            attn_mix(64, 0, Hq=8, Hkv=2)
        N.lib().aic_debug_attn_layout(0, 0)
    if "fp8layout" in what:   # the fp8 short body is issue-bound, not HBM-bound: does a second workgroup per CU pay there?
        attn_mix(64, 0, kv8=True)       # (first case of a process: clocks still ramping)
        for B, combos in ((64, ((0, 0), (4, 2), (4, 4), (4, 8))), (32, ((0, 0), (4, 4), (4, 8), (4, 16))), (16, ((0, 0), (4, 16)))):
            for hpw, sp in combos:
                N.lib().aic_debug_attn_layout(hpw, sp)
                print(f"fp8 layout hpw={hpw} splits={sp}: ", end="")
                attn_mix(B, 0, kv8=True)
        N.lib().aic_debug_attn_layout(0, 0)
    if "longdma" in what:     # who issues the tile DMA in a long-draft workgroup (aic_debug_attn_long_dma)
        for pat in (0, 1, 2, 3):
            N.lib().aic_debug_attn_long_dma(pat)
            print(f"-- long_dma pattern {pat}")
            for B, q in ((16, 33), (16, 20), (16, 24), (32, 12), (16, 6)):
                attn(B=B, ctx=4096, qlen=q, split=True)
            attn_mix(31, 1, q_long=33)
            attn_mix(28, 4, q_long=20)
        N.lib().aic_debug_attn_long_dma(1)
    if "mixphases" in what:   # the same account for the long workgroups of mixed calls (one lane of the bench)
        attn_long_phases(1, 33, 4224, n_short=31)
        attn_long_phases(1, 10, 4224, n_short=31)
        attn_long_phases(4, 20, 4224, n_short=28)
    if "longphases" in what:  # cycle accounting of the long-draft body
        attn_long_phases(16, 33)
        attn_long_phases(16, 20)
        attn_long_phases(32, 12)
    if "phases" in what:      # in-kernel phase trace of the short body: where the per-call fixed cost sits
        for kv8 in (False, True):
            for n in (32, 64):
                attn_phases(n, 4224, kv8)
            attn_phases(32, 1056, kv8)
    if "ctxsweep" in what:    # short body over the context length: slope = streaming rate, intercept = fixed cost per call
        for kv8 in (False, True):
            for n in (32, 64):
                for ctx in (544, 1056, 2112, 4224, 8448):
                    attn_mix(n, 0, 1, 33, ctx=ctx, kv8=kv8)
    if "longctx" in what:     # the long-draft body alone over the context length: fixed cost and slope per 32-token tile
        for B in (16, 1):
            for ctx in (128, 512, 1024, 2048, 4096):
                attn(B=B, ctx=ctx, qlen=33, split=True)
        for ctx in (128, 1024, 4096):
            attn(B=16, ctx=ctx, qlen=12, split=True)
    if "manylong" in what:    # a 32-request lane in which suffix decoding hit for many requests (one-grid form out of room)
        for ns, nl, qlong in ((31, 1, 33), (28, 4, 20), (25, 7, 20), (22, 10, 20), (19, 13, 20), (17, 15, 20), (17, 15, 33), (12, 20, 12),
                              (4, 28, 12), (0, 32, 12)):
            for mode, name in ((0, "one grid   "), (1, "two launches"), (-1, "library    ")):
                N.lib().aic_debug_attn_sequential(mode)
                print(name, end=" ")
                attn_mix(ns, nl, q_long=qlong)
        N.lib().aic_debug_attn_sequential(-1)
    if "fp8mix" in what:      # a 32-request lane with long drafts, fp8 cache against bf16
        attn_mix(32, 0, kv8=True)
        for kv8 in (True, False):
            for ns, nl, qlong in ((32, 0, 33), (31, 1, 33), (30, 2, 33), (30, 2, 12), (28, 4, 20), (0, 1, 33), (0, 2, 33)):
                attn_mix(ns, nl, q_long=qlong, kv8=kv8)
    if "pmc" in what:      # the cases whose kernels tools/pmc_kernels.py tells apart by name (few iterations: counters, not time)
        _t = timeit
        timeit = lambda fn, iters=4, warm=1: _t(fn, iters=4, warm=1)
        attn(split=True)                               # verify_attn_kernel<1, true, false, 4>: the bf16 short body (baseline)
        attn(split=True, kv8=True)                     # <1, true, true, 4>: fp8 short body
        attn(B=64, Hq=4, Hkv=1, split=True)            # <1, false, false, 4>: SP = 8 slice, waves = token ranges
        attn(B=16, qlen=33, split=True)                # verify_attn_long4_kernel<false, 128>: long drafts alone
        attn(B=16, qlen=33, split=True, kv8=True)      # verify_attn_long4_kernel<true, 128>
        attn(B=64, Hq=64, Hkv=8, D=64)                 # verify_attn_long4_kernel<false, 64>: head size 64
        timeit = _t
    if "mid" in what:      # suffix drafts of 4-7 tokens (17-32 query rows): two row tiles of the short body since r02
        attn_mix(64, 0)
        attn_mix(59, 5, q_long=8)
        attn_mix(59, 5, q_long=5)
        attn_mix(59, 5, q_long=9)
        attn_mix(59, 5, q_long=33)
        attn_mix(57, 7, q_long=8)
        attn_mix(30, 2, q_long=8)
        attn_mix(30, 2, q_long=33)
        attn_mix(59, 5, q_long=8, Hq=4, Hkv=1)
        attn_mix(59, 5, q_long=33, Hq=4, Hkv=1)
        attn_mix(64, 0, q_short=8)         # every request at two row tiles
        attn_mix(64, 0, q_short=4)
    if "light" in what:    # weight of the short-body splits that share CUs with long-draft workgroups (percent; 100 = equal)
        for pct in (100, 96, 92, 88, 84):
            N.lib().aic_debug_attn_light(pct)
            print("light", pct)
            attn_ql([1] * 31 + [33], jitter=256)
            attn_ql([1] * 30 + [20, 33], jitter=256)
            attn_ql([1] * 29 + [10, 20, 33], jitter=256)
            attn_ql([1] * 63 + [33], jitter=256)
            attn_ql([1] * 62 + [20] * 2, jitter=256)
            attn_ql([1] * 59 + [33] * 5, jitter=256)
            attn_ql([1] * 59 + [12] * 3 + [33] * 2, Hq=4, Hkv=1, jitter=256)
        N.lib().aic_debug_attn_light(0)
    if "ql" in what:
        for jit in (0, 256):
            attn_ql([1] * 64, jitter=jit)
            attn_ql([1] * 59 + [6] * 3 + [20] * 2, jitter=jit)
            attn_ql([1] * 59 + [8] * 3 + [33] * 2, jitter=jit)
            attn_ql([1] * 61 + [6] * 3, jitter=jit)
            attn_ql([1] * 62 + [20] * 2, jitter=jit)
            attn_ql([1] * 63 + [33], jitter=jit)
            attn_ql([1] * 58 + [5, 7, 8, 10, 20, 33], jitter=jit)
    if "trace32" in what:      # one lane of the bench: 32 requests, a few long drafts
        attn_trace(31, 1)
        attn_trace(28, 4, q_long=20)
        attn_trace(27, 5)
    if "trace" in what:
        attn_trace(59, 5)
        attn_trace(56, 8)
        attn_trace(62, 2)
        attn_trace(63, 1)
        attn_trace(30, 2)
        attn_trace(31, 1)
    if "sptrace" in what:        # the SP = 8 slice: which workgroups end the one-grid launch, by long split count
        for sp in (24, 32):
            N.lib().aic_debug_attn_long_splits(sp)
            print("long splits", sp)
            attn_trace(63, 1, Hq=4, Hkv=1)
            attn_trace(31, 1, Hq=4, Hkv=1)
        N.lib().aic_debug_attn_long_splits(0)
        attn_mix(64, 0, Hq=4, Hkv=1)
        attn_mix(32, 0, Hq=4, Hkv=1)
    if "mix" in what:
        attn_mix(64, 0)
        attn_mix(59, 5)
        attn_mix(59, 0)
        attn_mix(0, 5)
        attn_mix(56, 8)
        attn_mix(59, 5, q_long=17)
        attn_mix(59, 5, q_short=4)
        attn_mix(59, 5, kv8=True)
        attn_mix(8, 1, Hq=4, Hkv=1)      # SP = 8 slice
        attn_mix(59, 5, Hq=4, Hkv=1)
    if "mixdma" in what:        # DMA duty pattern of the long part of mixed calls
        for pat in (0, 1, 2):
            N.lib().aic_debug_attn_long_dma(pat)
            print(f"-- long_dma pattern {pat}")
            for ns, nl, ql in ((31, 1, 10), (30, 2, 10), (31, 1, 20), (31, 1, 33), (30, 2, 33), (28, 4, 20), (27, 5, 33)):
                attn_mix(ns, nl, q_long=ql)
        N.lib().aic_debug_attn_long_dma(1)
    if "mixes" in what:         # the bench's common lane mixes under the library's own rules
        attn(B=32, ctx=4224, qlen=1, split=True)
        for ns, nl, ql in ((31, 1, 10), (30, 2, 10), (31, 1, 20), (30, 2, 20), (31, 1, 33), (30, 2, 33), (28, 4, 20), (27, 5, 33), (25, 7, 20)):
            attn_mix(ns, nl, q_long=ql)
    if "longsplits8" in what:   # the same at full width (8 kv heads: one GPU), after the r04 long-body changes
        L = N.lib()
        for ns, nl, ql in ((31, 1, 10), (30, 2, 10), (31, 1, 20), (30, 2, 20), (31, 1, 33), (30, 2, 33), (28, 4, 20), (27, 5, 33), (25, 7, 20)):
            for sp in (0, 2, 3, 4, 5, 6, 8, 10):
                L.aic_debug_attn_long_splits(sp)
                print("long splits %2d: " % sp, end="")
                attn_mix(ns, nl, q_long=ql)
        L.aic_debug_attn_long_splits(0)
    if "longsplits" in what:
        # long-draft split count of a mixed call on the slices a rank sees under SP (kv heads per rank 8 / 4 / 2 / 1)
        L = N.lib()
        for Hkv in (4, 2, 1):
            for ns, nl in ((63, 1), (61, 3), (59, 5), (31, 1), (30, 2), (15, 1)):
                for sp in (0, 12, 16, 20, 24, 32):
                    L.aic_debug_attn_long_splits(sp)
                    print("long splits %2d: " % sp, end="")
                    attn_mix(ns, nl, Hq=4 * Hkv, Hkv=Hkv)
        L.aic_debug_attn_long_splits(0)
    if "splight" in what:       # lighter trailing splits on SP slices, where the long workgroups are many and short
        for Hkv in (1, 2, 4):
            for ns, nl in ((63, 1), (61, 3), (59, 5), (31, 1), (29, 3)):
                for pct in (100, 94, 0):
                    N.lib().aic_debug_attn_light(pct)
                    print("light %3d: " % pct, end="")
                    attn_mix(ns, nl, Hq=4 * Hkv, Hkv=Hkv)
        N.lib().aic_debug_attn_light(0)
    if "spsplits" in what:      # short-part split count on SP slices (aic_debug_attn_layout: heads per workgroup, splits)
        for Hkv, hpws in ((1, (1,)), (2, (2, 1)), (4, (4, 2))):
            for ns, nl in ((64, 0), (63, 1), (32, 0), (31, 1)):
                for hpw in hpws:
                    for sp in (0, 2, 3, 4, 6, 8, 12, 16):
                        N.lib().aic_debug_attn_layout(hpw if sp else 0, sp)
                        print("hpw %d splits %2d: " % (hpw, sp), end="")
                        attn_mix(ns, nl, Hq=4 * Hkv, Hkv=Hkv)
        N.lib().aic_debug_attn_layout(0, 0)
    if "lstm" in what:
        lstm(64)
        lstm(32)
        lstm(8)
    if "rej" in what:
        rejection()
