"""The attention oracle (oracle/spec_oracle.verify_attention) cross-checked on CPU against an independent
implementation of the same mathematics, torch.nn.functional.scaled_dot_product_attention on dense per-request
tensors with an explicit mask: query position p of a request with context ctx and q_len query tokens sees keys
0 .. ctx - q_len + p (causal within the chunk, everything before it visible).  This does not pin the
oracle to the reference (vLLM's backend is not available here) but it does pin it to the definition."""
import numpy as np
import pytest
import torch

from oracle import spec_oracle as O


@pytest.mark.parametrize("Hq,Hkv,D,bs", [(8, 2, 64, 16), (4, 4, 128, 32), (16, 2, 128, 16)])
def test_oracle_attention_equals_sdpa(Hq, Hkv, D, bs):
    g = torch.Generator().manual_seed(Hq * 7 + D)
    q_lens, ctxs = [4, 1, 9, 33], [70, 16, 9, 131]
    B = len(q_lens)
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = sum((c + bs - 1) // bs for c in ctxs) + 3
    perm = torch.randperm(nb, generator=g)
    bt = torch.zeros(B, max_blocks, dtype=torch.int32)
    at = 0
    for i, c in enumerate(ctxs):
        k = (c + bs - 1) // bs
        bt[i, :k] = perm[at:at + k].to(torch.int32)
        at += k
    kc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    T = sum(q_lens)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    qsl = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    scale = D ** -0.5
    got = O.verify_attention(q, kc, vc, bt, ctxs, qsl, scale)
    G = Hq // Hkv
    for i, (ql, ctx) in enumerate(zip(q_lens, ctxs)):
        blocks = bt[i, :(ctx + bs - 1) // bs].long()
        K = kc[blocks].reshape(-1, Hkv, D)[:ctx].float().repeat_interleave(G, dim=1).transpose(0, 1)   # [Hq, ctx, D]
        V = vc[blocks].reshape(-1, Hkv, D)[:ctx].float().repeat_interleave(G, dim=1).transpose(0, 1)
        Q = q[qsl[i]:qsl[i + 1]].float().transpose(0, 1)                                                 # [Hq, ql, D]
        pos = torch.arange(ql).unsqueeze(1) + (ctx - ql)
        mask = torch.arange(ctx).unsqueeze(0) <= pos                                                      # [ql, ctx]
        want = torch.nn.functional.scaled_dot_product_attention(Q, K, V, attn_mask=mask, scale=scale).transpose(0, 1)
        assert torch.allclose(got[qsl[i]:qsl[i + 1]], want, atol=2e-5, rtol=1e-5), i


@pytest.mark.parametrize("window,with_sinks", [(24, False), (0, True), (24, True), (200, True)])
def test_oracle_window_and_sinks_equal_an_explicit_softmax(window, with_sinks):
    """gpt-oss layers: the sliding-window bound and the per-head sink term against a literal dense computation
    (mask built from absolute positions; the sink as an extra key with a zero value row)."""
    Hq, Hkv, D, bs = 8, 2, 64, 16
    g = torch.Generator().manual_seed(3)
    q_lens, ctxs = [4, 1, 9, 33], [70, 16, 9, 131]
    B = len(q_lens)
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = sum((c + bs - 1) // bs for c in ctxs) + 3
    perm = torch.randperm(nb, generator=g)
    bt = torch.zeros(B, max_blocks, dtype=torch.int32)
    at = 0
    for i, c in enumerate(ctxs):
        k = (c + bs - 1) // bs
        bt[i, :k] = perm[at:at + k].to(torch.int32)
        at += k
    kc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    q = torch.randn(sum(q_lens), Hq, D, generator=g).to(torch.bfloat16)
    sinks = torch.randn(Hq, generator=g) * 2 if with_sinks else None
    qsl = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    scale = D ** -0.5
    got = O.verify_attention(q, kc, vc, bt, ctxs, qsl, scale, sliding_window=window, sinks=sinks)
    G = Hq // Hkv
    for i, (ql, ctx) in enumerate(zip(q_lens, ctxs)):
        blocks = bt[i, :(ctx + bs - 1) // bs].long()
        K = kc[blocks].reshape(-1, Hkv, D)[:ctx].float().repeat_interleave(G, dim=1)      # [ctx, Hq, D]
        V = vc[blocks].reshape(-1, Hkv, D)[:ctx].float().repeat_interleave(G, dim=1)
        for j in range(ql):
            p_abs = ctx - ql + j
            lo = max(0, p_abs - window + 1) if window else 0
            for h in range(Hq):
                logits = (K[lo:p_abs + 1, h] @ q[qsl[i] + j, h].float()) * scale
                vals = V[lo:p_abs + 1, h]
                if with_sinks:
                    logits = torch.cat([logits, sinks[h].view(1)])
                    vals = torch.cat([vals, torch.zeros(1, D)])
                want = torch.softmax(logits, dim=0) @ vals
                assert torch.allclose(got[qsl[i] + j, h], want, atol=2e-5, rtol=1e-5), (i, j, h)
