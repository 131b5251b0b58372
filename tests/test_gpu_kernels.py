"""GPU parity of the HIP kernels (called through the C ABI via arcticinference_amd.ops) against the CPU
oracle restatements in oracle/spec_oracle.py on seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import spec_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from arcticinference_amd import ops
    return ops


# ------------------------------------------------------------------------------------------------
# A16 bulk KV write
# ------------------------------------------------------------------------------------------------
def _kv_case(L, T, H, D, dtype, kv_dtype, block_size=16, num_blocks=None, neg=False, extra_stride=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = H * D
    num_blocks = num_blocks or (T + block_size - 1) // block_size + 3
    keys = torch.randn(T, L * n + extra_stride, generator=g).to(dtype)
    values = torch.randn(T, L * n + extra_stride, generator=g).to(dtype)
    slots = torch.randperm(num_blocks * block_size, generator=g)[:T].to(torch.int64)
    if neg:
        slots[::3] = -1
    cdt = {"auto": dtype, "fp8": torch.float8_e4m3fn, "fp8_e4m3": torch.float8_e4m3fn,
           "fp8_e5m2": torch.float8_e5m2}[kv_dtype]
    kc = [torch.zeros(num_blocks, block_size, H, D, dtype=cdt) for _ in range(L)]
    vc = [torch.zeros(num_blocks, block_size, H, D, dtype=cdt) for _ in range(L)]
    ks = [torch.tensor(0.5 + 0.25 * l, dtype=torch.float32) for l in range(L)]
    vs = [torch.tensor(0.75 + 0.125 * l, dtype=torch.float32) for l in range(L)]
    return keys, values, kc, vc, slots, ks, vs


@pytest.mark.parametrize("cfg", [
    dict(L=4, T=2, H=2, D=16, dtype=torch.float32, kv="auto"),       # the reference test's shape (test_custom_ops.py:56-99)
    dict(L=3, T=37, H=8, D=128, dtype=torch.bfloat16, kv="auto", neg=True),
    dict(L=2, T=19, H=1, D=128, dtype=torch.float16, kv="auto", extra_stride=24),
    dict(L=5, T=33, H=8, D=128, dtype=torch.bfloat16, kv="fp8_e4m3", neg=True),
    dict(L=2, T=9, H=2, D=64, dtype=torch.float32, kv="fp8_e5m2"),
    dict(L=2, T=7, H=3, D=20, dtype=torch.bfloat16, kv="auto"),      # not 16-byte divisible -> scalar path
    dict(L=40, T=5, H=2, D=32, dtype=torch.bfloat16, kv="auto"),     # more than 32 layers: two launches
])
def test_kv_bulk_write(cfg):
    keys, values, kc, vc, slots, ks, vs = _kv_case(cfg["L"], cfg["T"], cfg["H"], cfg["D"], cfg["dtype"], cfg["kv"],
                                                   neg=cfg.get("neg", False), extra_stride=cfg.get("extra_stride", 0))
    kc_ref = [c.clone() for c in kc]
    vc_ref = [c.clone() for c in vc]
    O.kv_bulk_write(keys, values, kc_ref, vc_ref, slots, cfg["kv"], ks, vs, cfg["H"], cfg["D"])
    d = lambda t: t.to(DEV)
    kc_d, vc_d = [d(c) for c in kc], [d(c) for c in vc]
    _ops().reshape_and_cache_flash_bulk(d(keys), d(values), kc_d, vc_d, d(slots), cfg["kv"], [d(s) for s in ks],
                                        [d(s) for s in vs], cfg["H"], cfg["D"])
    torch.cuda.synchronize()
    for a, b in zip(kc_d + vc_d, kc_ref + vc_ref):
        assert torch.equal(a.cpu().view(torch.uint8), b.view(torch.uint8))  # bit-exact, fp8 included


def test_kv_bulk_write_errors_and_noop():
    ops = _ops()
    keys, values, kc, vc, slots, ks, vs = _kv_case(2, 4, 2, 16, torch.bfloat16, "auto")
    d = lambda t: t.to(DEV)
    ops.reshape_and_cache_flash_bulk(d(keys), d(values), [], [], d(slots), "auto", [], [], 2, 16)  # num_layers == 0
    with pytest.raises(RuntimeError):
        ops.reshape_and_cache_flash_bulk(d(keys), d(values), [d(c) for c in kc], [d(c) for c in vc][:1], d(slots),
                                         "auto", [d(s) for s in ks], [d(s) for s in vs], 2, 16)
    with pytest.raises(RuntimeError):
        ops.reshape_and_cache_flash_bulk(d(keys), d(values), [d(c) for c in kc], [d(c) for c in vc], d(slots),
                                         "int4", [d(s) for s in ks], [d(s) for s in vs], 2, 16)
    with pytest.raises(RuntimeError):
        ops.reshape_and_cache_flash_bulk(keys, values, kc, vc, slots, "auto", ks, vs, 2, 16)  # CPU tensors: no fallback


def test_kv_bulk_write_full_size_roundtrip():
    """SURVEY §8d size: T=4096, Lkv=16, Hkv=8, D=128 bf16; property: gather(cache, slots) == source."""
    T, L, H, D, bs = 4096, 16, 8, 128, 16
    n = H * D
    g = torch.Generator(device=DEV).manual_seed(1)
    keys = torch.randn(T, L * n, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    values = torch.randn(T, L * n, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    nb = T // bs + 8
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int64)
    kv = torch.zeros(L, 2, nb, bs, H, D, dtype=torch.bfloat16, device=DEV)
    one = torch.ones(1, device=DEV)
    _ops().reshape_and_cache_flash_bulk(keys, values, [kv[l, 0] for l in range(L)], [kv[l, 1] for l in range(L)],
                                        slots, "auto", [one] * L, [one] * L, H, D)
    flat = kv.view(L, 2, nb * bs, n)
    assert torch.equal(flat[:, 0, slots].transpose(0, 1).reshape(T, L * n), keys)
    assert torch.equal(flat[:, 1, slots].transpose(0, 1).reshape(T, L * n), values)
    assert int((flat.abs().sum(-1) != 0).sum()) == 2 * L * T  # nothing else was touched


# ------------------------------------------------------------------------------------------------
# A6 rejection acceptance
# ------------------------------------------------------------------------------------------------
def _rej_case(B, V, max_n, dtype, seed, plant=0.7):
    rng = np.random.default_rng(seed)
    n = rng.integers(0, max_n + 1, size=B)
    n[0] = max_n
    rows = int(n.sum())
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(rows, V, generator=g).to(dtype)
    draft = rng.integers(0, V, size=rows)
    for r in range(rows):
        if rng.random() < plant:
            logits[r, draft[r]] = 30.0
    # exact ties: the first maximum must win
    if rows:
        logits[0, 5] = 40.0
        logits[0, V - 3] = 40.0
    bonus = rng.integers(0, V, size=B)
    return n, logits, draft, bonus


@pytest.mark.parametrize("B,V,max_n,dtype", [(1, 1000, 3, torch.float32), (7, 32000, 5, torch.bfloat16),
                                             (64, 128256, 3, torch.bfloat16), (5, 4099, 33, torch.float16),
                                             (3, 130, 2, torch.bfloat16)])
def test_rejection_greedy(B, V, max_n, dtype):
    n, logits, draft, bonus = _rej_case(B, V, max_n, dtype, seed=B + V)
    want = O.rejection_greedy(logits, draft, n, bonus, max_n)
    cu = torch.tensor(np.cumsum(n), dtype=torch.int32, device=DEV)
    res = _ops().rejection_sample(logits.to(DEV), torch.tensor(draft, dtype=torch.int32, device=DEV), cu,
                                  torch.tensor(bonus, dtype=torch.int32, device=DEV), max_n)
    got = res.output_token_ids.cpu().numpy()
    assert np.array_equal(got, want)
    # proposer inputs (arctic_proposer.py:133-147)
    assert np.array_equal(res.hidden_index.cpu().numpy(), O.hidden_state_index(want, n))
    nacc = (want != -1).sum(1)
    assert np.array_equal(res.num_accepted.cpu().numpy(), nacc)
    last = want[np.arange(B), nacc - 1]
    assert np.array_equal(res.last_token.cpu().numpy(), last)
    # the same rows read in place from a larger [T, V] tensor through target_row_index (target_logits_indices)
    rows = logits.shape[0]
    if rows:
        pos = torch.randperm(rows + B, generator=torch.Generator().manual_seed(1))[:rows]
        full = torch.randn(rows + B, V, generator=torch.Generator().manual_seed(2)).to(dtype)
        full[pos] = logits
        res2 = _ops().rejection_sample(full.to(DEV), torch.tensor(draft, dtype=torch.int32, device=DEV), cu,
                                       torch.tensor(bonus, dtype=torch.int32, device=DEV), max_n,
                                       target_row_index=pos.to(torch.int64).to(DEV))
        assert np.array_equal(res2.output_token_ids.cpu().numpy(), want)
        # ... and with the bonus tokens taken as the arg-max of B more rows of that tensor, in the same launch
        rest = [i for i in range(rows + B) if i not in set(pos.tolist())]
        brow = torch.tensor(rest[:B], dtype=torch.int64)
        bonus2 = torch.argmax(full[brow].float(), dim=-1).numpy()
        want2 = O.rejection_greedy(logits, draft, n, bonus2, max_n)
        res3 = _ops().rejection_sample(full.to(DEV), torch.tensor(draft, dtype=torch.int32, device=DEV), cu, None, max_n,
                                       target_row_index=pos.to(torch.int64).to(DEV), bonus_row_index=brow.to(DEV))
        assert np.array_equal(res3.output_token_ids.cpu().numpy(), want2)


def test_rejection_random_mixed():
    B, V, max_n = 9, 5000, 4
    n, logits, draft, bonus = _rej_case(B, V, max_n, torch.float32, seed=3, plant=0.5)
    rng = np.random.default_rng(0)
    temp = np.array([0.0, 1.0, 0.7, 1.3, 0.0, 1.0, 2.0, 0.5, 1.0], dtype=np.float32)
    rows = int(n.sum())
    uniform = rng.random(rows)
    noise = torch.empty(B, V).exponential_(generator=torch.Generator().manual_seed(1))
    want = O.rejection_random(logits, draft, n, bonus, max_n, temp, uniform, noise)
    cu = torch.tensor(np.cumsum(n), dtype=torch.int32, device=DEV)
    res = _ops().rejection_sample(logits.to(DEV), torch.tensor(draft, dtype=torch.int32, device=DEV), cu,
                                  torch.tensor(bonus, dtype=torch.int32, device=DEV), max_n,
                                  temperature=torch.tensor(temp, device=DEV),
                                  uniform_probs=torch.tensor(uniform, dtype=torch.float64, device=DEV),
                                  exp_noise=noise.to(DEV))
    assert np.array_equal(res.output_token_ids.cpu().numpy(), want)


# ------------------------------------------------------------------------------------------------
# A12 Ulysses repartition
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sp,n,hq,hkv,D", [(2, 5, 2, 1, 64), (4, 16, 2, 1, 128), (8, 64, 4, 1, 128), (8, 33, 8, 1, 128)])
def test_ulysses_pack_unpack(sp, n, hq, hkv, D):
    g = torch.Generator().manual_seed(sp * 100 + n)
    q = torch.randn(n, sp * hq * D, generator=g).to(torch.bfloat16)
    k = torch.randn(n, sp * hkv * D, generator=g).to(torch.bfloat16)
    v = torch.randn(n, sp * hkv * D, generator=g).to(torch.bfloat16)
    want = O.ulysses_pack(q, k, v, sp, hq, hkv, D)
    ops = _ops()
    got = ops.ulysses_pack_qkv(q.to(DEV), k.to(DEV), v.to(DEV), sp)
    assert torch.equal(got.cpu(), want)
    q_, k_, v_ = ops.ulysses_split_qkv(got, hq * D, hkv * D)
    wq, wk, wv = want.split([hq * D, hkv * D, hkv * D], dim=-1)
    assert torch.equal(q_.cpu(), wq) and torch.equal(k_.cpu(), wk) and torch.equal(v_.cpu(), wv)
    c = torch.randn(sp * n, hq * D, generator=g).to(torch.bfloat16)
    assert torch.equal(ops.ulysses_unpack_out(c.to(DEV), sp).cpu(), O.ulysses_unpack(c, sp, hq, D))


@pytest.mark.parametrize("sp,aa,n,hq,D", [(4, 2, 16, 2, 128), (8, 2, 33, 4, 128), (8, 4, 64, 8, 64), (8, 1, 7, 8, 128)])
def test_ulysses_kv_replicated_copies(sp, aa, n, hq, D):
    """Fewer kv heads than SP ranks (ulysses.py:462-490): q packed alone for the SP all-to-all, K|V packed over the aa kv
    heads for the SP_AA all-to-all, gathered chunks put back into rank order and split — bit-exact against the literal
    torch expressions."""
    ag = sp // aa
    g = torch.Generator().manual_seed(sp * 10 + aa)
    q = torch.randn(n, sp * hq * D, generator=g).to(torch.bfloat16)
    k = torch.randn(n, aa * D, generator=g).to(torch.bfloat16)
    v = torch.randn(n, aa * D, generator=g).to(torch.bfloat16)
    ops = _ops()
    assert torch.equal(ops.ulysses_pack_pair(q.to(DEV), None, sp).cpu(), O.ulysses_pack_pair(q, None, sp))
    assert torch.equal(ops.ulysses_pack_pair(k.to(DEV), v.to(DEV), aa).cpu(), O.ulysses_pack_pair(k, v, aa))
    wide = torch.randn(n, aa * D + 64, generator=g).to(torch.bfloat16)          # a strided source
    assert torch.equal(ops.ulysses_pack_pair(wide.to(DEV)[:, :aa * D], v.to(DEV), aa).cpu(),
                       O.ulysses_pack_pair(wide[:, :aa * D].contiguous(), v, aa))
    order = [j * aa + i for i in range(aa) for j in range(ag)]
    gathered = torch.randn(sp * n, 2 * D, generator=g).to(torch.bfloat16)
    gk, gv = ops.ulysses_reorder_split_kv(gathered.to(DEV), sp, order)
    wk, wv = O.ulysses_reorder_split(gathered, sp, order)
    assert torch.equal(gk.cpu(), wk) and torch.equal(gv.cpu(), wv)
    from arcticinference_amd._native import NativeError
    with pytest.raises((NativeError, RuntimeError)):
        ops.ulysses_reorder_split_kv(gathered.to(DEV), sp, [0] * sp)             # not a permutation


# ------------------------------------------------------------------------------------------------
# A5 verify attention
# ------------------------------------------------------------------------------------------------
def _attn_case(B, Hq, Hkv, D, q_lens, ctxs, bs, seed):
    g = torch.Generator().manual_seed(seed)
    T = int(sum(q_lens))
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = sum((c + bs - 1) // bs for c in ctxs) + 4
    perm = torch.randperm(nb, generator=g)
    bt = torch.zeros(B, max_blocks, dtype=torch.int32)
    p = 0
    for i, c in enumerate(ctxs):
        k = (c + bs - 1) // bs
        bt[i, :k] = perm[p:p + k].to(torch.int32)
        p += k
    kc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    qsl = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    return q, kc, vc, bt, qsl


@pytest.mark.parametrize("cfg", [
    dict(B=1, Hq=4, Hkv=1, q_lens=[4], ctxs=[37], bs=16),
    dict(B=3, Hq=32, Hkv=8, q_lens=[4, 1, 3], ctxs=[300, 17, 1025], bs=16),
    dict(B=2, Hq=8, Hkv=8, q_lens=[2, 5], ctxs=[64, 96], bs=16),          # G = 1
    dict(B=2, Hq=32, Hkv=8, q_lens=[33, 9], ctxs=[700, 40], bs=16),        # suffix-length drafts, several row groups
    dict(B=2, Hq=16, Hkv=2, q_lens=[4, 4], ctxs=[515, 33], bs=32),         # G = 8, block_size 32
    dict(B=4, Hq=4, Hkv=1, q_lens=[4, 4, 4, 4], ctxs=[4100, 4099, 5, 4], bs=16),  # SP=8 slice of Llama-8B; ctx == q_len
    dict(B=5, Hq=32, Hkv=8, q_lens=[4, 33, 2, 17, 4], ctxs=[900, 1300, 64, 2100, 33], bs=16),  # mixed LSTM / suffix drafts
    dict(B=2, Hq=64, Hkv=8, q_lens=[33, 3], ctxs=[500, 70], bs=16),   # G = 8: 264 rows -> two 192-row groups
    dict(B=4, Hq=4, Hkv=1, q_lens=[4, 33, 9, 2], ctxs=[1500, 2600, 130, 64], bs=16),   # SP=8 slice with suffix drafts: short + long in one launch, waves = token ranges
    dict(B=3, Hq=16, Hkv=2, q_lens=[2, 20, 1], ctxs=[777, 1111, 48], bs=32),           # G = 8, two kv heads, one launch
    dict(B=40, Hq=32, Hkv=8, q_lens=[4] * 30 + [12, 33, 7, 20, 9, 33, 5, 16, 11, 6], ctxs=[260 + 37 * i for i in range(40)], bs=16),  # many items: the one-launch form does not fit, two launches
    dict(B=3, Hq=32, Hkv=8, q_lens=[50, 4, 63], ctxs=[640, 64, 2049], bs=64),          # block_size 64; 200 and 252 rows: two 192-row groups
    dict(B=2, Hq=8, Hkv=2, q_lens=[4, 1], ctxs=[128, 256], bs=128),                      # block_size 128, contexts at tile / page boundaries
    dict(B=5, Hq=32, Hkv=8, q_lens=[4, 33, 1, 20, 24], ctxs=[700, 1300, 95, 2100, 333], bs=48),  # pages of 48 tokens: not a power of two (the bodies divide instead of shifting)
    dict(B=4, Hq=32, Hkv=8, q_lens=[18, 33, 22, 36], ctxs=[1100, 1057, 2079, 97], bs=16),   # 5 / 9 / 6 / 9 row tiles: waves with one tile more issue no DMA; contexts end inside a tile
])
def test_verify_attention(cfg):
    D = 128
    q, kc, vc, bt, qsl = _attn_case(cfg["B"], cfg["Hq"], cfg["Hkv"], D, cfg["q_lens"], cfg["ctxs"], cfg["bs"], seed=7)
    scale = 1.0 / D ** 0.5
    want = O.verify_attention(q, kc, vc, bt, cfg["ctxs"], qsl, scale)
    got = _ops().verify_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV),
                                  torch.tensor(cfg["ctxs"], dtype=torch.int32, device=DEV),
                                  torch.tensor(qsl, device=DEV), max(cfg["q_lens"]), max(cfg["ctxs"]), scale)
    # the same batch with host-known query lengths: long drafts go through the shared-tile kernel
    got2 = _ops().verify_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV),
                                   torch.tensor(cfg["ctxs"], dtype=torch.int32, device=DEV),
                                   torch.tensor(qsl, device=DEV), max(cfg["q_lens"]), max(cfg["ctxs"]), scale,
                                   q_lens_host=cfg["q_lens"])
    assert torch.allclose(got2.float().cpu(), want, atol=1e-3, rtol=2 ** -8), f"split path: {(got2.float().cpu() - want).abs().max()}"
    err = (got.float().cpu() - want).abs()
    # tolerance: BASELINE.json north_star asks for 1e-3 in bf16 -> atol 1e-3 plus one bf16 rounding of the
    # output itself (relative 2^-8); internally P.V runs at ~fp32 accuracy (bf16 hi+lo split)
    assert torch.allclose(got.float().cpu(), want, atol=1e-3, rtol=2 ** -8), f"max abs err {err.max()}"


@pytest.mark.parametrize("cfg", [
    dict(B=3, Hq=32, Hkv=8, q_lens=[4, 1, 3], ctxs=[300, 17, 1025], bs=16),
    dict(B=4, Hq=4, Hkv=1, q_lens=[4, 4, 2, 4], ctxs=[4100, 99, 5, 4], bs=16),           # token-split variant
    dict(B=4, Hq=32, Hkv=8, q_lens=[4, 33, 2, 9], ctxs=[900, 1300, 64, 700], bs=32),   # long drafts, both paths
])
def test_verify_attention_fp8_kv(cfg):
    """fp8 (e4m3) KV cache as written by the bulk KV op: K = k8 * k_scale, V = v8 * v_scale."""
    D = 128
    q, kc, vc, bt, qsl = _attn_case(cfg["B"], cfg["Hq"], cfg["Hkv"], D, cfg["q_lens"], cfg["ctxs"], cfg["bs"], seed=21)
    ks, vs = 0.043, 0.021
    k8 = O.fp8_sat(kc.float() / ks, "e4m3")
    v8 = O.fp8_sat(vc.float() / vs, "e4m3")
    scale = 1.0 / D ** 0.5
    want = O.verify_attention(q, k8, v8, bt, cfg["ctxs"], qsl, scale, ks, vs)
    args = (q.to(DEV), k8.to(DEV), v8.to(DEV), bt.to(DEV), torch.tensor(cfg["ctxs"], dtype=torch.int32, device=DEV),
            torch.tensor(qsl, device=DEV), max(cfg["q_lens"]), max(cfg["ctxs"]), scale)
    kw = dict(k_scale=torch.tensor([ks], device=DEV), v_scale=torch.tensor([vs], device=DEV))
    for extra in ({}, {"q_lens_host": cfg["q_lens"]}):
        got = _ops().verify_attention(*args, **kw, **extra)
        assert torch.allclose(got.float().cpu(), want, atol=1e-3, rtol=2 ** -8), (extra, (got.float().cpu() - want).abs().max())


@pytest.mark.parametrize("cfg", [
    dict(B=3, Hq=64, Hkv=8, q_lens=[4, 1, 3], ctxs=[300, 17, 1025], bs=16),          # gpt-oss heads (G = 8), k = 3 drafts
    dict(B=4, Hq=8, Hkv=1, q_lens=[4, 33, 9, 2], ctxs=[1500, 2600, 130, 64], bs=16),  # its SP = 8 slice, suffix drafts
    dict(B=2, Hq=16, Hkv=4, q_lens=[33, 20], ctxs=[700, 95], bs=32),                  # G = 4, block_size 32
    dict(B=5, Hq=8, Hkv=8, q_lens=[1, 2, 1, 5, 1], ctxs=[1, 2, 33, 64, 4097], bs=16),  # G = 1; ctx == q_len; one-token contexts
])
def test_verify_attention_head_size_64(cfg):
    """head_size 64 (gpt-oss-120b, BASELINE configs[4]): every request takes the shared-tile body."""
    D = 64
    q, kc, vc, bt, qsl = _attn_case(cfg["B"], cfg["Hq"], cfg["Hkv"], D, cfg["q_lens"], cfg["ctxs"], cfg["bs"], seed=13)
    scale = 1.0 / D ** 0.5
    want = O.verify_attention(q, kc, vc, bt, cfg["ctxs"], qsl, scale)
    args = (q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV), torch.tensor(cfg["ctxs"], dtype=torch.int32, device=DEV),
            torch.tensor(qsl, device=DEV), max(cfg["q_lens"]), max(cfg["ctxs"]), scale)
    for extra in ({}, {"q_lens_host": cfg["q_lens"]}):
        got = _ops().verify_attention(*args, **extra)
        assert torch.allclose(got.float().cpu(), want, atol=1e-3, rtol=2 ** -8), (extra, (got.float().cpu() - want).abs().max())


@pytest.mark.parametrize("cfg", [
    # gpt-oss-120b heads (Hq 64, Hkv 8, D 64) and their SP = 8 slice (Hq 8, Hkv 1): window 128 + sinks, k = 3 and suffix drafts
    dict(B=4, Hq=64, Hkv=8, D=64, q_lens=[4, 1, 3, 4], ctxs=[300, 17, 1025, 4100], bs=16, window=128, sinks=True),
    dict(B=4, Hq=8, Hkv=1, D=64, q_lens=[4, 33, 9, 2], ctxs=[1500, 2600, 130, 64], bs=16, window=128, sinks=True),
    dict(B=4, Hq=8, Hkv=1, D=64, q_lens=[4, 33, 9, 2], ctxs=[1500, 2600, 130, 64], bs=16, window=0, sinks=True),   # full-attention layer
    dict(B=3, Hq=64, Hkv=8, D=64, q_lens=[33, 20, 5], ctxs=[700, 95, 160], bs=32, window=128, sinks=False),
    # the same features at head size 128 (short body, long body, one-grid pair launch, token-split waves, fp8 cache)
    dict(B=5, Hq=32, Hkv=8, D=128, q_lens=[4, 33, 2, 17, 4], ctxs=[900, 1300, 64, 2100, 33], bs=16, window=128, sinks=True),
    dict(B=4, Hq=4, Hkv=1, D=128, q_lens=[4, 4, 2, 4], ctxs=[4100, 99, 5, 4], bs=16, window=64, sinks=True),
    dict(B=4, Hq=32, Hkv=8, D=128, q_lens=[4, 33, 2, 9], ctxs=[900, 1300, 64, 700], bs=32, window=200, sinks=True, fp8=True),
    dict(B=3, Hq=16, Hkv=2, D=128, q_lens=[1, 1, 1], ctxs=[31, 32, 33], bs=16, window=32, sinks=True),    # window == a tile, contexts around it
    dict(B=2, Hq=8, Hkv=2, D=128, q_lens=[5, 2], ctxs=[5, 2], bs=16, window=3, sinks=True),              # window shorter than the draft; ctx == q_len
    dict(B=2, Hq=8, Hkv=8, D=128, q_lens=[3, 1], ctxs=[4000, 2500], bs=16, window=0, sinks=True, sink_shift=8.0),   # sinks over many splits (counted once)
])
def test_verify_attention_sliding_window_and_sinks(cfg):
    """gpt-oss layers (BASELINE configs[4]): the sliding-window bound and the per-head sink term, against the fp32 oracle
    (parity unpinned against vLLM's backends: semantics recalled, oracle cross-checked in tests/test_oracle_attention.py).
    Both call forms (with and without the host partition), forced split counts included: a sink must be counted once
    however a row's range is split, and a window must shorten the range without moving its upper edge."""
    from arcticinference_amd import _native as N
    D = cfg["D"]
    q, kc, vc, bt, qsl = _attn_case(cfg["B"], cfg["Hq"], cfg["Hkv"], D, cfg["q_lens"], cfg["ctxs"], cfg["bs"], seed=31)
    g = torch.Generator().manual_seed(77)
    sinks = (torch.randn(cfg["Hq"], generator=g) * 3 + cfg.get("sink_shift", 0.0)).float() if cfg["sinks"] else None
    ks = vs = 1.0
    kw = {}
    if cfg.get("fp8"):
        ks, vs = 0.043, 0.021
        kc, vc = O.fp8_sat(kc.float() / ks, "e4m3"), O.fp8_sat(vc.float() / vs, "e4m3")
        kw = dict(k_scale=torch.tensor([ks], device=DEV), v_scale=torch.tensor([vs], device=DEV))
    scale = D ** -0.5
    want = O.verify_attention(q, kc, vc, bt, cfg["ctxs"], qsl, scale, ks, vs, sliding_window=cfg["window"], sinks=sinks)
    plain = O.verify_attention(q, kc, vc, bt, cfg["ctxs"], qsl, scale, ks, vs)
    assert not torch.allclose(want, plain, atol=1e-2), "the case does not exercise the feature"
    args = (q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV), torch.tensor(cfg["ctxs"], dtype=torch.int32, device=DEV),
            torch.tensor(qsl, device=DEV), max(cfg["q_lens"]), max(cfg["ctxs"]), scale)
    dsinks = None if sinks is None else sinks.to(DEV)
    try:
        for splits in (0, 1, 3):
            N.lib().aic_debug_attn_layout(0, splits)
            for lens in (cfg["q_lens"], None):
                out = torch.full((sum(cfg["q_lens"]), cfg["Hq"], D), float("nan"), dtype=torch.bfloat16, device=DEV)
                got = _ops().verify_attention(*args, out=out, q_lens_host=lens, sliding_window=cfg["window"], sinks=dsinks,
                                              **kw).float().cpu()
                assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), (splits, lens is None, float((got - want).abs().max()))
    finally:
        N.lib().aic_debug_attn_layout(0, 0)


def test_verify_attention_window_reads_only_the_window():
    """A windowed layer must not depend on anything before the window: poisoning every cache page that lies wholly below
    (ctx - q_len - window + 1) with NaN leaves the result unchanged (the range really is shortened, not just masked)."""
    D, Hq, Hkv, W = 64, 64, 8, 128
    q_lens, ctxs = [4, 1, 7], [3000, 1500, 700]
    q, kc, vc, bt, qsl = _attn_case(3, Hq, Hkv, D, q_lens, ctxs, 16, seed=5)
    args = lambda k, v: (q.to(DEV), k.to(DEV), v.to(DEV), bt.to(DEV), torch.tensor(ctxs, dtype=torch.int32, device=DEV),
                         torch.tensor(qsl, device=DEV), max(q_lens), max(ctxs), D ** -0.5)
    clean = _ops().verify_attention(*args(kc, vc), q_lens_host=q_lens, sliding_window=W)
    kp, vp = kc.clone(), vc.clone()
    for i, (ql, ctx) in enumerate(zip(q_lens, ctxs)):
        first_needed = ((ctx - ql - W + 1) // 32) * 32          # the kernel starts at a 32-token tile boundary
        for blk in range(first_needed // 16):
            kp[int(bt[i, blk])] = float("nan")
            vp[int(bt[i, blk])] = float("nan")
    poisoned = _ops().verify_attention(*args(kp, vp), q_lens_host=q_lens, sliding_window=W)
    assert torch.equal(clean, poisoned) and not torch.isnan(clean.float()).any()


@pytest.mark.parametrize("seed", list(range(16)))
def test_verify_attention_random_shapes(seed):
    """Seeded random batches over the head geometries the path serves (group size 1 / 2 / 4 / 8, 1 / 2 / 4 / 8 kv heads per
    rank), query lengths 1..33 (row counts on both sides of the 16- and 32-row boundaries of the short body and of the
    192-row group of the long one), contexts from q_len itself to a few thousand tokens, block sizes 16 / 32 / 64, bf16 and
    fp8 caches, with and without the host-side partition: all against the fp32 oracle, 1e-3."""
    rng = np.random.default_rng(1000 + seed)
    Hkv, G = [(1, 4), (8, 4), (4, 2), (2, 1), (1, 8), (2, 8), (8, 1), (4, 4), (2, 4), (1, 1), (4, 8), (8, 8), (8, 8), (1, 8),
              (2, 4), (8, 8)][seed]
    fp8 = bool(seed % 3 == 2)
    Hq, D = Hkv * G, (64 if (seed % 4 == 3 and not fp8) else 128)      # head size 64 (bf16 cache only) in a quarter of the cases
    bs = int(rng.choice([16, 32, 64]))
    B = int(rng.integers(1, 10))
    q_lens = [int(x) for x in rng.choice([1, 1, 2, 4, 4, 5, 8, 9, 16, 17, 24, 33], size=B)]
    ctxs = [int(max(q, rng.choice([q, q + 1, 31, 32, 33, 200, 1000, 2500]) + rng.integers(0, 40))) for q in q_lens]
    q, kc, vc, bt, qsl = _attn_case(B, Hq, Hkv, D, q_lens, ctxs, bs, seed=seed)
    ks = vs = 1.0
    kw = {}
    if fp8:
        ks, vs = 0.05, 0.03
        kc, vc = O.fp8_sat(kc.float() / ks, "e4m3"), O.fp8_sat(vc.float() / vs, "e4m3")
        kw = dict(k_scale=torch.tensor([ks], device=DEV), v_scale=torch.tensor([vs], device=DEV))
    want = O.verify_attention(q, kc, vc, bt, ctxs, qsl, D ** -0.5, ks, vs)
    args = (q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV), torch.tensor(ctxs, dtype=torch.int32, device=DEV),
            torch.tensor(qsl, device=DEV), max(q_lens), max(ctxs), D ** -0.5)
    for lens in (q_lens, None):
        out = torch.full((sum(q_lens), Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        got = _ops().verify_attention(*args, out=out, q_lens_host=lens, **kw).float().cpu()
        assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), (seed, Hq, Hkv, bs, q_lens, ctxs, fp8, lens is None,
                                                                    float((got - want).abs().max()))


@pytest.mark.parametrize("kv", ["bf16", "fp8"])
def test_verify_attention_every_short_layout(kv):
    """The short body's layouts for host-partitioned calls — 4 / 2 / 1 kv heads per workgroup (the waves of a head merge
    their token ranges through LDS) x 1 / 2 / 3 cross-workgroup splits; one split writes the finished row itself (no
    partials, no combine launch) — forced through the debug entry point: every one must give the oracle's result, with
    and without long drafts in the call (rows finished by the attention launch must survive the combine launch)."""
    from arcticinference_amd import _native as N
    D, Hq, Hkv, bs = 128, 32, 8, 16
    cases = [([4, 1, 3, 4, 2, 1], [300, 17, 1025, 64, 2049, 33]),              # all short: direct mode skips the combine
             ([4, 33, 2, 17, 4, 1], [900, 1300, 64, 2100, 33, 16]),           # mixed: short rows direct, long rows combined
             # 17-32 query rows (q_len 5-8 at Hq/Hkv = 4): the two-row-tile form of the short body, chosen per workgroup
             ([8, 1, 5, 4, 7, 6], [300, 17, 1025, 64, 2049, 33]),
             ([8, 33, 5, 9, 4, 1, 6], [900, 1300, 64, 2100, 33, 16, 700])]
    ops = _ops()
    try:
        for q_lens, ctxs in cases:
            q, kc, vc, bt, qsl = _attn_case(len(ctxs), Hq, Hkv, D, q_lens, ctxs, bs, seed=17)
            ks = vs = 1.0
            kw = {}
            if kv == "fp8":
                ks, vs = 0.04, 0.02
                kc, vc = O.fp8_sat(kc.float() / ks, "e4m3"), O.fp8_sat(vc.float() / vs, "e4m3")
                kw = dict(k_scale=torch.tensor([ks], device=DEV), v_scale=torch.tensor([vs], device=DEV))
            want = O.verify_attention(q, kc, vc, bt, ctxs, qsl, D ** -0.5, ks, vs)
            args = (q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV), torch.tensor(ctxs, dtype=torch.int32, device=DEV),
                    torch.tensor(qsl, device=DEV), max(q_lens), max(ctxs), D ** -0.5)
            for hpw in (4, 2, 1):
                for splits in (1, 2, 3):
                    N.lib().aic_debug_attn_layout(hpw, splits)
                    out = torch.full((sum(q_lens), Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
                    got = ops.verify_attention(*args, out=out, q_lens_host=q_lens, **kw).float().cpu()
                    assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), (kv, q_lens, hpw, splits, (got - want).abs().max())
    finally:
        N.lib().aic_debug_attn_layout(0, 0)


def test_verify_attention_plan_equals_direct_call():
    """VerifyAttentionPlan (arguments built once, one foreign call per layer) gives the bits of verify_attention()."""
    D, Hq, Hkv = 128, 32, 8
    q_lens, ctxs = [4, 33, 2, 17, 4], [900, 1300, 64, 2100, 33]
    q, kc, vc, bt, qsl = _attn_case(5, Hq, Hkv, D, q_lens, ctxs, 16, seed=3)
    ops = _ops()
    dq, dk, dv, dbt = q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    dqsl = torch.tensor(qsl, device=DEV)
    rs = ops.split_requests(q_lens, Hq // Hkv, DEV)
    want = ops.verify_attention(dq, dk, dv, dbt, seq, dqsl, max(q_lens), max(ctxs), D ** -0.5, req_split=rs)
    out = torch.empty_like(dq)
    plan = ops.VerifyAttentionPlan(dq, out, dk, dbt, seq, dqsl, max(q_lens), max(ctxs), D ** -0.5, req_split=rs)
    k2, v2 = dk.clone(), dv.clone()          # another "layer": same shapes, other addresses
    plan.run(k2, v2)
    assert torch.equal(out, want)
    with pytest.raises(ValueError):
        plan.run(dk[:-1], dv[:-1])
    # every layer of a step in one foreign call: the last layer's result stays in `out`
    k3, v3 = torch.randn_like(dk), torch.randn_like(dv)
    want3 = ops.verify_attention(dq, k3, v3, dbt, seq, dqsl, max(q_lens), max(ctxs), D ** -0.5, req_split=rs)
    plan.run_layers(plan.layer_tables([dk, k2, k3], [dv, v2, v3]))
    assert torch.equal(out, want3)
    with pytest.raises(ValueError):
        plan.layer_tables([dk[:-1]], [dv[:-1]])


def test_verify_attention_layers_graph_equals_kernel_launches():
    """aic_verify_attention_layers sends >= 4 layers out as one HIP graph whose kernel nodes are re-parameterised per call:
    same bits as kernel-by-kernel launches, across calls that change the batch (same and different kernel sequences), and
    the graph really is what ran (launch / instantiate counters)."""
    import ctypes
    from arcticinference_amd import _native as N
    D, Hq, Hkv = 128, 32, 8
    ops = _ops()
    lib = N.lib()

    def stats():
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        lib.aic_debug_attn_graph_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    cases = [([4, 33, 2, 17, 4], [900, 1300, 64, 2100, 33], 5),     # short + long drafts: the one-grid launch + combine
             ([5, 40, 1, 9, 2], [700, 2300, 640, 100, 3300], 5),     # same kernels, other geometry: parameters only
             ([4, 4, 2, 7, 4], [900, 1300, 64, 2100, 33], 5),        # short requests only: another kernel sequence
             ([4, 33, 2, 17, 4], [900, 1300, 64, 2100, 33], 5),      # the first shape again: its graph is reused
             ([4, 33, 2, 17, 4], [900, 1300, 64, 2100, 33], 13)]     # 12 layers or more: a 4-layer graph, then the rest
    try:
        l0, b0 = stats()
        for i, (q_lens, ctxs, L) in enumerate(cases):
            q, kc, vc, bt, qsl = _attn_case(5, Hq, Hkv, D, q_lens, ctxs, 16, seed=11 + i)
            dq, dbt = q.to(DEV), bt.to(DEV)
            ks = [kc.to(DEV)] + [torch.randn_like(kc).to(DEV) for _ in range(L - 1)]
            vs = [vc.to(DEV)] + [torch.randn_like(vc).to(DEV) for _ in range(L - 1)]
            seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
            dqsl = torch.tensor(qsl, device=DEV)
            rs = ops.split_requests(q_lens, Hq // Hkv, DEV)
            # q / out advance per layer here (layer stride = one q), so every layer's result is checked
            qs = torch.stack([dq * (1.0 + 0.25 * (l % 5)) for l in range(L)]).contiguous()
            outs = {}
            for mode in (0, 1):
                lib.aic_debug_attn_graph(mode)
                out = torch.full_like(qs, float("nan"))
                plan = ops.VerifyAttentionPlan(qs[0], out[0], ks[0], dbt, seq, dqsl, max(q_lens), max(ctxs), D ** -0.5, req_split=rs)
                a = plan._args
                kt, vt, n, _keep = plan.layer_tables(ks, vs)
                N.check(lib.aic_verify_attention_layers(a[0], a[1], qs.stride(0), kt, vt, n, *a[4:20], a[20], a[21],
                                                        out.stride(0), *a[22:]))
                torch.cuda.synchronize()
                outs[mode] = out
            assert torch.equal(outs[0], outs[1]), i
            for l in (0, 3, 4, L - 1):
                want = ops.verify_attention(qs[l], ks[l], vs[l], dbt, seq, dqsl, max(q_lens), max(ctxs), D ** -0.5, req_split=rs)
                assert torch.equal(outs[1][l], want), (i, l)
        l1, b1 = stats()
        # one launch per 5-layer call, two for the 13-layer one; graphs: mixed x 5, short-only x 5, mixed x 4, mixed x 9
        assert l1 - l0 == len(cases) + 1 and b1 - b0 == 4, (l0, b0, l1, b1)
    finally:
        lib.aic_debug_attn_graph(1)


def test_verify_attention_layers_graph_back_to_back_calls_keep_their_own_parameters():
    """interleaved lanes enqueue a same-shape call while the previous one is still running.  Calls of one
    shape with different geometry, queries and caches go out back to back WITHOUT a sync in between (several more than
    the per-shape exec pool holds); every call's output must be bit-identical to its kernel-by-kernel result — a call
    must never see the node parameters of the call behind it."""
    from arcticinference_amd import _native as N
    D, Hq, Hkv, L, B = 128, 32, 8, 6, 16
    ops = _ops()
    lib = N.lib()
    g = torch.Generator().manual_seed(5)
    calls = []
    for c in range(7):
        q_lens = [int(x) for x in torch.randint(1, 5, (B,), generator=g)]
        q_lens[c % B] = 33                                   # one long draft: the one-grid launch + combine (same shape)
        ctxs = [int(x) for x in torch.randint(2500, 4000, (B,), generator=g)]
        q, kc, vc, bt, qsl = _attn_case(B, Hq, Hkv, D, q_lens, ctxs, 16, seed=40 + c)
        dq = q.to(DEV)
        calls.append(dict(q_lens=q_lens, ctxs=ctxs, qs=torch.stack([dq * (1.0 + 0.5 * l) for l in range(L)]).contiguous(),
                          k=kc.to(DEV), v=vc.to(DEV), bt=bt.to(DEV), seq=torch.tensor(ctxs, dtype=torch.int32, device=DEV),
                          qsl=torch.tensor(qsl, device=DEV), rs=ops.split_requests(q_lens, Hq // Hkv, DEV)))

    def issue(c, out):
        plan = ops.VerifyAttentionPlan(c["qs"][0], out[0], c["k"], c["bt"], c["seq"], c["qsl"], max(c["q_lens"]),
                                       max(c["ctxs"]), D ** -0.5, req_split=c["rs"])
        a = plan._args
        kt, vt, n, keep = plan.layer_tables([c["k"]] * L, [c["v"]] * L)
        N.check(lib.aic_verify_attention_layers(a[0], a[1], c["qs"].stride(0), kt, vt, n, *a[4:20], a[20], a[21],
                                                out.stride(0), *a[22:]))
        return plan, keep

    try:
        want = []
        lib.aic_debug_attn_graph(0)
        for c in calls:
            out = torch.full_like(c["qs"], float("nan"))
            issue(c, out)
            torch.cuda.synchronize()
            want.append(out)
        lib.aic_debug_attn_graph(1)
        l0, _ = N.attn_graph_stats()
        outs = [torch.full_like(c["qs"], float("nan")) for c in calls]
        torch.cuda.synchronize()
        keep = [issue(c, o) for c, o in zip(calls, outs)]      # seven same-shape graph launches in flight, no sync
        torch.cuda.synchronize()
        assert N.attn_graph_stats()[0] - l0 == len(calls)
        for i, (o, w) in enumerate(zip(outs, want)):
            assert torch.equal(o, w), i
    finally:
        lib.aic_debug_attn_graph(1)


def test_verify_attention_ignores_empty_trailing_requests():
    """What a full-graph replay looks like to the kernel: the launch was recorded for MORE requests than the step has —
    the entries behind the live ones have an empty query (query_start_loc repeated), a zero length and a zeroed block-table
    row — and for an upper bound of the context length.  The live requests' rows equal the oracle; nothing faults."""
    D, Hq, Hkv = 128, 8, 2
    q_lens, ctxs = [1, 1, 4, 1, 2], [300, 33, 64, 17, 1025]
    q, kc, vc, bt, qsl = _attn_case(5, Hq, Hkv, D, q_lens, ctxs, 16, seed=9)
    want = O.verify_attention(q, kc, vc, bt, ctxs, qsl, D ** -0.5)
    B_cap = 12                                             # captured batch
    bt_pad = torch.zeros(B_cap, bt.shape[1], dtype=torch.int32)
    bt_pad[:5] = bt
    seq = torch.tensor(ctxs + [0] * (B_cap - 5), dtype=torch.int32)
    qsl_pad = torch.tensor(list(qsl) + [int(qsl[-1])] * (B_cap - 5), dtype=torch.int32)
    T_cap = 16                                             # captured token count: rows behind the live ones are padding
    q_pad = torch.zeros(T_cap, Hq, D, dtype=torch.bfloat16)
    q_pad[:q.shape[0]] = q
    out = torch.full((T_cap, Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    _ops().verify_attention(q_pad.to(DEV), kc.to(DEV), vc.to(DEV), bt_pad.to(DEV), seq.to(DEV), qsl_pad.to(DEV), 4, 4096,
                            D ** -0.5, out=out)
    torch.cuda.synchronize()
    got = out[:q.shape[0]].float().cpu()
    assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), float((got - want).abs().max())


@pytest.mark.parametrize("cfg", [
    dict(Hq=32, Hkv=8, D=128, fp8=False, window=0, sinks=False),    # Llama-3.1-8B heads
    dict(Hq=4, Hkv=1, D=128, fp8=False, window=0, sinks=False),     # their SP = 8 slice (waves = token ranges)
    dict(Hq=32, Hkv=8, D=128, fp8=True, window=0, sinks=False),     # fp8 e4m3 cache
    dict(Hq=64, Hkv=8, D=64, fp8=False, window=128, sinks=True),    # gpt-oss layer
])
def test_verify_attention_device_geometry_form_replayed_from_a_hip_graph(cfg):
    """The call form a full-graph capture records (vllm_plugin/ulysses.py::_arctic_verify under a capturing stream): no
    host-side request partition, request count / token count / max_query_len / max_seq_len frozen at capture-time upper
    bounds, per-request lengths, query offsets, block table and q read from PERSISTENT device buffers.  The launch is
    captured ONCE into a HIP graph and replayed over three different steps written into those buffers (fewer live requests
    than captured, mixed query lengths incl. suffix-length drafts, contexts from 1 token to the bound); every replay's live
    rows must equal the fp32 oracle at the kernel tolerance — atol 1e-3 + one bf16 ulp, the same bar as the eager forms,
    so graph replay has its own guard besides tests/test_vllm_capture_gpu.py's hidden-state comparison (3e-2)."""
    Hq, Hkv, D, bs = cfg["Hq"], cfg["Hkv"], cfg["D"], 16
    B_cap, T_cap, MAXQ, MAXS = 16, 96, 33, 2048
    blocks_per = MAXS // bs
    nb = B_cap * blocks_per + 3
    g = torch.Generator().manual_seed(23)
    kc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, generator=g).to(torch.bfloat16)
    ks = vs = 1.0
    kw = {}
    if cfg["fp8"]:
        ks, vs = 0.037, 0.019
        kc, vc = O.fp8_sat(kc.float() / ks, "e4m3"), O.fp8_sat(vc.float() / vs, "e4m3")
        kw = dict(k_scale=torch.tensor([ks], device=DEV), v_scale=torch.tensor([vs], device=DEV))
    sinks = (torch.randn(Hq, generator=g) * 2).float() if cfg["sinks"] else None
    if sinks is not None:
        kw["sinks"] = sinks.to(DEV)
    scale = D ** -0.5
    # persistent buffers (what vLLM refreshes before a replay)
    d_q = torch.zeros(T_cap, Hq, D, dtype=torch.bfloat16, device=DEV)
    d_out = torch.zeros(T_cap, Hq, D, dtype=torch.bfloat16, device=DEV)
    d_seq = torch.zeros(B_cap, dtype=torch.int32, device=DEV)
    d_qsl = torch.zeros(B_cap + 1, dtype=torch.int32, device=DEV)
    d_bt = torch.zeros(B_cap, blocks_per, dtype=torch.int32, device=DEV)
    kcd, vcd = kc.to(DEV), vc.to(DEV)

    def call():
        _ops().verify_attention(d_q, kcd, vcd, d_bt, d_seq, d_qsl, MAXQ, MAXS, scale, out=d_out, sliding_window=cfg["window"], **kw)

    steps = [
        dict(q_lens=[1] * 9, ctxs=[1500, 33, 1, 700, 16, 17, 2048, 64, 999]),
        dict(q_lens=[4, 4, 1, 33, 2, 4, 9], ctxs=[1200, 40, 5, 1999, 2, 310, 777]),
        dict(q_lens=[4] * 16, ctxs=[100 + 121 * i for i in range(16)]),
    ]

    def load(step):
        q_lens, ctxs = step["q_lens"], step["ctxs"]
        B, T = len(q_lens), sum(q_lens)
        assert B <= B_cap and T <= T_cap and max(ctxs) <= MAXS and all(c >= n for c, n in zip(ctxs, q_lens))
        gg = torch.Generator().manual_seed(1000 + T)
        q = torch.randn(T, Hq, D, generator=gg).to(torch.bfloat16)
        perm = torch.randperm(nb, generator=gg)
        bt = torch.zeros(B_cap, blocks_per, dtype=torch.int32)
        p = 0
        for i, c in enumerate(ctxs):
            k = (c + bs - 1) // bs
            bt[i, :k] = perm[p:p + k].to(torch.int32)
            p += k
        qsl = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
        d_q.zero_()
        d_q[:T].copy_(q)
        d_seq.copy_(torch.tensor(ctxs + [0] * (B_cap - B), dtype=torch.int32))
        d_qsl.copy_(torch.tensor(list(qsl) + [int(qsl[-1])] * (B_cap - B), dtype=torch.int32))
        d_bt.copy_(bt)
        d_out.fill_(float("nan"))
        return q, bt[:B], qsl, T

    # warm-up outside the capture (workspace allocation, module load), then ONE capture
    load(steps[0])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call()
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        call()
    for step in steps:
        q, bt, qsl, T = load(step)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = O.verify_attention(q, kc, vc, bt, step["ctxs"], qsl, scale, ks, vs, sliding_window=cfg["window"], sinks=sinks)
        got = d_out[:T].float().cpu()
        assert not torch.isnan(got).any()
        assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), (step["q_lens"], float((got - want).abs().max()))
        # and the replay is the eager device-geometry call, bit for bit
        d_out.fill_(float("nan"))
        call()
        torch.cuda.synchronize()
        assert torch.equal(d_out[:T].float().cpu(), got)


def test_verify_attention_unsupported_shapes():
    from arcticinference_amd._native import NativeError
    q, kc, vc, bt, qsl = _attn_case(1, 4, 1, 96, [2], [40], 16, seed=1)
    args = lambda q, kc, vc: (q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV), torch.tensor([40], dtype=torch.int32, device=DEV),
                              torch.tensor(qsl, device=DEV), 2, 40, 0.1)
    with pytest.raises((NativeError, RuntimeError)):
        _ops().verify_attention(*args(q, kc, vc))                       # head_size 96
    q, kc, vc, bt, qsl = _attn_case(1, 4, 1, 64, [2], [40], 16, seed=1)
    k8 = O.fp8_sat(kc.float(), "e4m3")
    one = torch.ones(1, device=DEV)
    with pytest.raises((NativeError, RuntimeError)):
        _ops().verify_attention(*args(q, k8, k8), k_scale=one, v_scale=one)   # head_size 64 with an fp8 cache


def test_verify_attention_strided_q_and_peaked_softmax():
    """q as a column slice of an all-to-all receive buffer; one key dominates (forces the online-softmax rescale)."""
    D, Hq, Hkv = 128, 8, 2
    q, kc, vc, bt, qsl = _attn_case(2, Hq, Hkv, D, [4, 4], [200, 333], 16, seed=11)
    buf = torch.zeros(q.shape[0], (Hq + 2 * Hkv) * D, dtype=torch.bfloat16)
    buf[:, :Hq * D] = q.reshape(q.shape[0], -1)
    # spike: make key 150 of request 1 line up with query row 0 of head 0
    blk, off = int(bt[1, 150 // 16]), 150 % 16
    kc[blk, off, 0] = q[4, 0] * 4.0
    want = O.verify_attention(q, kc, vc, bt, [200, 333], qsl, 1.0 / D ** 0.5)
    bd = buf.to(DEV)
    qv = bd[:, :Hq * D].view(-1, Hq, D)
    got = _ops().verify_attention(qv, kc.to(DEV), vc.to(DEV), bt.to(DEV), torch.tensor([200, 333], dtype=torch.int32, device=DEV),
                                  torch.tensor(qsl, device=DEV), 4, 333, 1.0 / D ** 0.5)
    assert torch.allclose(got.float().cpu(), want, atol=1e-3, rtol=2 ** -8)


@pytest.mark.parametrize("kv", ["bf16", "fp8"])
def test_verify_attention_full_size_properties(kv):
    """BASELINE size (B=64, ~4096-token contexts, Llama-8B heads, k=3 and suffix-length drafts), where the fp32 oracle
    is too slow to be the checker: size-independent properties instead.
    (1) paging: physically shuffling the KV pages and permuting the block table must not change one bit;
    (2) two independent code paths (one launch with host-known lengths / the generic kernel + row groups) agree
        within the kernel tolerance;
    (3) the output is linear in V: attn(K, V1 + V2) = attn(K, V1) + attn(K, V2) within tolerance;
    (4) a sample of requests agrees with the oracle."""
    torch.manual_seed(5)
    B, Hq, Hkv, D, bs = 64, 32, 8, 128, 16
    rng = np.random.RandomState(3)
    q_lens = [4] * 56 + [int(x) for x in rng.randint(6, 34, size=8)]
    rng.shuffle(q_lens)
    ctxs = [int(x) for x in rng.randint(3900, 4353, size=B)]
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = B * max_blocks
    perm = torch.randperm(nb)
    bt = perm[:B * max_blocks].view(B, max_blocks).to(torch.int32)
    kc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    v2 = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    ks, vs = 1.0, 1.0
    skw = {}
    if kv == "fp8":   # the cache as the bulk KV op writes it: e4m3 codes of x / scale
        ks, vs = 0.031, 0.017
        kc, vc, v2 = (O.fp8_sat(t.float().cpu() / sc, "e4m3").to(DEV) for t, sc in ((kc, ks), (vc, vs), (v2, vs)))
        skw = dict(k_scale=torch.tensor([ks], device=DEV), v_scale=torch.tensor([vs], device=DEV))
    T = sum(q_lens)
    q = torch.randn(T, Hq, D, device=DEV, dtype=torch.bfloat16)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32), device=DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    scale = D ** -0.5
    ops = _ops()

    def run(k, v, table, **kw):
        return ops.verify_attention(q, k, v, table.to(DEV), seq, qsl, max(q_lens), max(ctxs), scale, **skw, **kw).float()

    a = run(kc, vc, bt, q_lens_host=q_lens)
    # (1) shuffle the pages
    shuf = torch.randperm(nb)
    inv = torch.empty_like(shuf)
    inv[shuf] = torch.arange(nb)
    def take(t, idx):   # page gather (through bytes: fp8 tensors have no index kernel)
        return t.view(torch.uint8)[idx].view(t.dtype) if t.dtype not in (torch.bfloat16,) else t[idx]
    a_shuf = run(take(kc, shuf.to(DEV)), take(vc, shuf.to(DEV)), inv[bt.long()].to(torch.int32), q_lens_host=q_lens)
    assert torch.equal(a, a_shuf)
    # (2) generic path
    b = run(kc, vc, bt)
    assert torch.allclose(a, b, atol=1e-3, rtol=2 ** -8), (a - b).abs().max()
    # (3) linearity in V (the V sum is formed in fp32 and rounded once: compare with matching slack)
    if kv == "bf16":
        a2 = run(kc, v2, bt, q_lens_host=q_lens)
        vsum = (vc.float() + v2.float()).to(torch.bfloat16)
        a12 = run(kc, vsum, bt, q_lens_host=q_lens)
        assert torch.allclose(a12, a + a2, atol=6e-3, rtol=2 ** -6), (a12 - a - a2).abs().max()
    # (4) oracle on three requests (one of them a long draft)
    pick = [0, int(np.argmax(q_lens)), B - 1]
    qs = qsl.cpu().numpy()
    for i in pick:
        rows = slice(int(qs[i]), int(qs[i + 1]))
        want = O.verify_attention(q[rows].cpu(), kc.cpu(), vc.cpu(), bt[i:i + 1], [ctxs[i]],
                                  np.array([0, q_lens[i]], dtype=np.int32), scale, ks, vs)
        assert torch.allclose(a[rows].cpu(), want, atol=1e-3, rtol=2 ** -8), (i, (a[rows].cpu() - want).abs().max())


@pytest.mark.parametrize("window", [128, 0])
def test_verify_attention_gpt_oss_full_batch_properties(window):
    """BASELINE configs[4] at batch size: gpt-oss-120b heads (Hq 64, Hkv 8, D 64), B = 64 requests, ~4K-token contexts,
    k = 3 drafts and suffix-length drafts, a sliding-window layer (128) and a full-attention layer (0), per-head sinks in
    both.  The fp32 oracle is too slow for the whole batch: size-independent properties, then an oracle sample.
    (1) paging: shuffled KV pages + permuted block table -> bit-identical;
    (2) the host-partitioned call form and the generic one agree within the kernel tolerance;
    (3) sinks: where the sink dominates the normaliser (sink logit +20: the keys' mass, ~e^9, is 2e-5 of it), raising every
        head's sink by ln 2 halves the output — the sink is counted exactly once however the row's range was split;
    (4) window: pages wholly below every request's window are poisoned with NaN -> unchanged (window layer only);
    (5) linearity in V;
    (6) four requests (first, last, longest draft, longest context) against the oracle."""
    torch.manual_seed(11)
    B, Hq, Hkv, D, bs = 64, 64, 8, 64, 16
    rng = np.random.RandomState(9)
    q_lens = [4] * 54 + [int(x) for x in rng.randint(5, 34, size=10)]
    rng.shuffle(q_lens)
    ctxs = [int(x) for x in rng.randint(3800, 4353, size=B)]
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = B * max_blocks
    bt = torch.randperm(nb).view(B, max_blocks).to(torch.int32)
    kc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    v2 = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    T = sum(q_lens)
    q = torch.randn(T, Hq, D, device=DEV, dtype=torch.bfloat16)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32), device=DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    sinks = (torch.randn(Hq) * 2 + 7.0).float()       # e^7 ~ 1100: comparable with the mass of ~4K keys (~6000) and of a 128-key window (~200)
    scale = D ** -0.5
    ops = _ops()

    def run(k, v, table, snk=sinks, **kw):
        return ops.verify_attention(q, k, v, table.to(DEV), seq, qsl, max(q_lens), max(ctxs), scale, sliding_window=window,
                                    sinks=None if snk is None else snk.to(DEV), **kw).float()

    a = run(kc, vc, bt, q_lens_host=q_lens)
    assert not torch.isnan(a).any()
    # (1)
    shuf = torch.randperm(nb)
    inv = torch.empty_like(shuf)
    inv[shuf] = torch.arange(nb)
    a_shuf = run(kc[shuf.to(DEV)], vc[shuf.to(DEV)], inv[bt.long()].to(torch.int32), q_lens_host=q_lens)
    assert torch.equal(a, a_shuf)
    # (2)
    b = run(kc, vc, bt)
    assert torch.allclose(a, b, atol=1e-3, rtol=2 ** -8), (a - b).abs().max()
    # (3) a dominating sink: doubling its weight halves the rows
    big = torch.full((Hq,), 20.0)
    s1 = run(kc, vc, bt, snk=big, q_lens_host=q_lens)
    s2 = run(kc, vc, bt, snk=big + float(np.log(2.0)), q_lens_host=q_lens)
    assert float(s1.abs().max()) > 1e-7                      # (not a comparison of zeros)
    assert torch.allclose(s2, 0.5 * s1, atol=1e-10, rtol=2 ** -6), (s2 - 0.5 * s1).abs().max()
    # and the sinks matter at all
    nos = run(kc, vc, bt, snk=None, q_lens_host=q_lens)
    assert not torch.allclose(a, nos, atol=2e-3)
    # (4) nothing below the window is read
    if window:
        kp, vp = kc.clone(), vc.clone()
        for i, c in enumerate(ctxs):
            lo = (c - q_lens[i]) - window + 1            # first key the first query row may see
            dead = (max(lo, 0) // 32 * 32) // bs         # pages wholly below its 32-token tile (the kernel starts at a tile boundary)
            kp[bt[i, :dead].long().to(DEV)] = float("nan")
            vp[bt[i, :dead].long().to(DEV)] = float("nan")
        for kw in (dict(q_lens_host=q_lens), {}):
            p = run(kp, vp, bt, **kw)
            assert torch.equal(p, run(kc, vc, bt, **kw))
        del kp, vp
    # (5) linearity in V (sink mass is part of the normaliser and the same in all three)
    a2 = run(kc, v2, bt, q_lens_host=q_lens)
    vsum = (vc.float() + v2.float()).to(torch.bfloat16)
    a12 = run(kc, vsum, bt, q_lens_host=q_lens)
    assert torch.allclose(a12, a + a2, atol=6e-3, rtol=2 ** -6), (a12 - a - a2).abs().max()
    # (6) the oracle on four requests
    qs = qsl.cpu().numpy()
    kcc, vcc = kc.cpu(), vc.cpu()
    for i in sorted({0, B - 1, int(np.argmax(q_lens)), int(np.argmax(ctxs))}):
        rows = slice(int(qs[i]), int(qs[i + 1]))
        want = O.verify_attention(q[rows].cpu(), kcc, vcc, bt[i:i + 1], [ctxs[i]], np.array([0, q_lens[i]], dtype=np.int32),
                                  scale, 1.0, 1.0, sliding_window=window, sinks=sinks)
        for got in (a, b):
            assert torch.allclose(got[rows].cpu(), want, atol=1e-3, rtol=2 ** -8), (i, (got[rows].cpu() - want).abs().max())


def test_verify_attention_lighter_trailing_splits():
    """One-grid launch of a lane-sized batch (30 requests of up to 32 rows + 2 long drafts, ~4100-token contexts): the short
    workgroups that will share a CU with a long-draft workgroup are the last in the grid — whole trailing token-range
    splits — and get a shorter range (AttnParams::light_from / light_pct).  Any weight must give the same attention:
    against equal splits, against the generic path, and against the oracle on a sample of requests."""
    from arcticinference_amd import _native as N
    torch.manual_seed(7)
    B, Hq, Hkv, D, bs = 32, 32, 8, 128, 16
    rng = np.random.RandomState(11)
    q_lens = [1] * 22 + [4] * 6 + [8, 5] + [33, 20]
    rng.shuffle(q_lens)
    ctxs = [int(x) for x in rng.randint(3700, 4353, size=B)]
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = B * max_blocks
    bt = torch.randperm(nb)[:B * max_blocks].view(B, max_blocks).to(torch.int32)
    kc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(sum(q_lens), Hq, D, device=DEV, dtype=torch.bfloat16)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32), device=DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    run = lambda **kw: _ops().verify_attention(q, kc, vc, bt.to(DEV), seq, qsl, max(q_lens), max(ctxs), D ** -0.5, **kw).float()
    try:
        outs = {}
        for pct in (100, 0, 75, 50):                 # equal splits, the built-in weight, two exaggerated ones
            N.lib().aic_debug_attn_light(pct)
            outs[pct] = run(q_lens_host=q_lens)
    finally:
        N.lib().aic_debug_attn_light(0)
    generic = run()
    for pct, o in outs.items():
        assert torch.allclose(o, outs[100], atol=1e-3, rtol=2 ** -8), (pct, (o - outs[100]).abs().max())
        assert torch.allclose(o, generic, atol=1e-3, rtol=2 ** -8), (pct, (o - generic).abs().max())
    assert not torch.equal(outs[50], outs[100])          # the weights did move the split boundaries
    qs = qsl.cpu().numpy()
    for i in (0, int(np.argmax(q_lens)), B - 1):
        rows = slice(int(qs[i]), int(qs[i + 1]))
        want = O.verify_attention(q[rows].cpu(), kc.cpu(), vc.cpu(), bt[i:i + 1], [ctxs[i]],
                                  np.array([0, q_lens[i]], dtype=np.int32), D ** -0.5, 1.0, 1.0)
        assert torch.allclose(outs[0][rows].cpu(), want, atol=1e-3, rtol=2 ** -8), (i, (outs[0][rows].cpu() - want).abs().max())


@pytest.mark.parametrize("n_long,kv", [(15, "bf16"), (20, "bf16"), (31, "bf16"), (16, "fp8")])
def test_verify_attention_many_long_drafts_in_a_lane(n_long, kv):
    """A lane step in which suffix decoding hit for half of the 32 requests or more: the one-grid launch would leave the
    long part a single split (every long workgroup alone with a whole context), so the call goes out as two plain launches
    on one stream — the long part with its own split count, then the short part.  Same attention as the generic path and
    the oracle, and the run of layers is still one HIP graph launch (a plain kernel chain: no side stream)."""
    import ctypes
    from arcticinference_amd import _native as N
    torch.manual_seed(3)
    B, Hq, Hkv, D, bs = 32, 32, 8, 128, 16
    rng = np.random.RandomState(n_long)
    q_lens = [int(x) for x in rng.randint(9, 34, size=n_long)] + [int(x) for x in rng.choice([1, 1, 1, 4, 6, 8], size=B - n_long)]
    rng.shuffle(q_lens)
    ctxs = [int(x) for x in rng.randint(1500, 2400, size=B)]
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = B * max_blocks
    bt = torch.randperm(nb).view(B, max_blocks).to(torch.int32)
    kw = {}
    if kv == "fp8":
        raw = torch.randint(0, 256, (2, nb, bs, Hkv, D), dtype=torch.uint8, device=DEV)
        raw[(raw & 0x7f) == 0x7f] = 0x30
        kc, vc = raw[0].view(torch.float8_e4m3fn), raw[1].view(torch.float8_e4m3fn)
        sc = torch.full((1,), 0.03, dtype=torch.float32, device=DEV)
        kw = dict(k_scale=sc, v_scale=sc)
    else:
        kc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
        vc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(sum(q_lens), Hq, D, device=DEV, dtype=torch.bfloat16)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32), device=DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    ops = _ops()
    dbt = bt.to(DEV)
    part = ops.verify_attention(q, kc, vc, dbt, seq, qsl, max(q_lens), max(ctxs), D ** -0.5, q_lens_host=q_lens, **kw).float()
    generic = ops.verify_attention(q, kc, vc, dbt, seq, qsl, max(q_lens), max(ctxs), D ** -0.5, **kw).float()
    # (two bf16 results, each rounded once: up to one bf16 ulp apart in either direction)
    assert torch.allclose(part, generic, atol=1e-3, rtol=2 ** -7), (part - generic).abs().max()
    qs = qsl.cpu().numpy()
    for i in (int(np.argmax(q_lens)), int(np.argmin(q_lens)), B - 1):
        rows = slice(int(qs[i]), int(qs[i + 1]))
        if kv == "fp8":
            want = O.verify_attention(q[rows].cpu(), kc.cpu(), vc.cpu(), bt[i:i + 1], [ctxs[i]],
                                      np.array([0, q_lens[i]], dtype=np.int32), D ** -0.5, 0.03, 0.03)
        else:
            want = O.verify_attention(q[rows].cpu(), kc.cpu(), vc.cpu(), bt[i:i + 1], [ctxs[i]],
                                      np.array([0, q_lens[i]], dtype=np.int32), D ** -0.5)
        assert torch.allclose(part[rows].cpu(), want, atol=1e-3, rtol=2 ** -8), (i, (part[rows].cpu() - want).abs().max())
    # four layers in one call: one graph launch, bit-identical to the single calls
    lib = N.lib()
    a0, b0 = ctypes.c_uint64(), ctypes.c_uint64()
    lib.aic_debug_attn_graph_stats(ctypes.byref(a0), ctypes.byref(b0))
    L = 4
    rs = ops.split_requests(q_lens, Hq // Hkv, DEV)
    outs = torch.full((L,) + tuple(q.shape), float("nan"), device=DEV, dtype=q.dtype)
    qs4 = torch.stack([q * (1.0 + 0.5 * l) for l in range(L)]).contiguous()
    plan = ops.VerifyAttentionPlan(qs4[0], outs[0], kc, dbt, seq, qsl, max(q_lens), max(ctxs), D ** -0.5, req_split=rs, **kw)
    a = plan._args
    kt, vt, n, _keep = plan.layer_tables([kc] * L, [vc] * L)
    N.check(lib.aic_verify_attention_layers(a[0], a[1], qs4.stride(0), kt, vt, n, *a[4:20], a[20], a[21], outs.stride(0), *a[22:]))
    torch.cuda.synchronize()
    a1, b1 = ctypes.c_uint64(), ctypes.c_uint64()
    lib.aic_debug_attn_graph_stats(ctypes.byref(a1), ctypes.byref(b1))
    assert a1.value - a0.value == 1, "the two-launch form must stay a plain kernel chain (one graph launch)"
    for l in range(L):
        want = ops.verify_attention(qs4[l], kc, vc, dbt, seq, qsl, max(q_lens), max(ctxs), D ** -0.5, req_split=rs, **kw)
        assert torch.equal(outs[l], want), l


@pytest.mark.parametrize("seed", list(range(8)))
def test_verify_attention_mixed_call_forms_agree(seed):
    """Lane- and batch-sized mixes of short requests and long drafts (anything from one long draft to all of them, on both
    sides of the point where the one-grid form runs out of room): the one-grid form, the two-launch form and the library's
    choice between them give the same attention as the generic path."""
    from arcticinference_amd import _native as N
    rng = np.random.RandomState(100 + seed)
    B = [32, 64, 32, 24, 64, 32, 48, 32][seed]
    Hkv = [8, 8, 8, 8, 2, 1, 8, 4][seed]
    G, D, bs = 4, 128, 16
    Hq = Hkv * G
    n_long = int(rng.randint(1, B))
    q_lens = [int(x) for x in rng.randint(9, 34, size=n_long)] + [int(x) for x in rng.choice([1, 1, 2, 4, 7, 8], size=B - n_long)]
    rng.shuffle(q_lens)
    ctxs = [int(x) for x in rng.randint(300, 1400, size=B)]
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = B * max_blocks
    bt = torch.randperm(nb).view(B, max_blocks).to(torch.int32).to(DEV)
    torch.manual_seed(seed)
    kc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(sum(q_lens), Hq, D, device=DEV, dtype=torch.bfloat16)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32), device=DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    run = lambda **kw: _ops().verify_attention(q, kc, vc, bt, seq, qsl, max(q_lens), max(ctxs), D ** -0.5, **kw).float()
    generic = run()
    try:
        for mode in (0, 1, -1):
            N.check(N.lib().aic_debug_attn_sequential(mode))
            got = run(q_lens_host=q_lens)
            assert torch.allclose(got, generic, atol=1e-3, rtol=2 ** -7), (seed, B, Hkv, n_long, mode, float((got - generic).abs().max()))
    finally:
        N.check(N.lib().aic_debug_attn_sequential(-1))


@pytest.mark.parametrize("shape", ["rising", "falling", "spikes", "plateaus"])
@pytest.mark.parametrize("fp8", [False, True])
def test_verify_attention_lazy_maximum_on_structured_scores(shape, fp8):
    """Both attention bodies keep a row's adopted maximum while no score exceeds it by more than 2^6 (lazy maximum): scores that
    RISE tile after tile (every tile takes the exact path), FALL (the first tile's maximum stays: weights down to 2^-120),
    SPIKE (isolated tokens far above a flat floor, at random tiles) or sit on PLATEAUS just below the slack (weights up to 2^6
    before a row adopts a new maximum) must give the oracle's attention — short requests, LSTM-length and suffix-length drafts."""
    torch.manual_seed(3)
    B, Hq, Hkv, D, bs = 6, 32, 8, 128, 16
    q_lens = [1, 4, 33, 20, 1, 9]
    ctxs = [1500, 700, 2100, 900, 64, 333]
    q, kc, vc, bt, qsl = _attn_case(B, Hq, Hkv, D, q_lens, ctxs, bs, seed=23)
    # keys = a common direction u times a per-token profile (+ a little noise), queries = c * u: score(t) ~ c * profile(t)
    g = torch.Generator().manual_seed(5)
    u = torch.randn(D, generator=g)
    u = u / u.norm()
    nb = kc.shape[0]
    tok = torch.arange(nb * bs).view(nb, bs).float()           # a key's profile follows its SLOT: contexts walk random pages, so
    for i, c in enumerate(ctxs):                               # give every request's tokens their position through the block table
        pos = torch.arange(c)
        slot = bt[i, pos // bs].long() * bs + pos % bs
        tok.view(-1)[slot] = pos.float()
    t = tok / 2100.0
    if shape == "rising":
        prof = 40.0 * t
    elif shape == "falling":
        prof = -40.0 * t
    elif shape == "spikes":
        prof = torch.where(torch.rand(nb, bs, generator=g) < 0.01, torch.full((nb, bs), 30.0), torch.zeros(nb, bs))
    else:
        prof = torch.floor(t * 12.0) * 1.7        # steps of 1.7 * c * scale in the exponent: several steps below the slack, then over it
    kc = (prof[..., None, None] * u + 0.05 * torch.randn(nb, bs, Hkv, D, generator=g)).to(torch.bfloat16)
    q = (3.0 * u + 0.05 * torch.randn(q.shape, generator=g)).to(torch.bfloat16) * (D ** 0.5 / 3.0)
    scale = D ** -0.5
    kw = {}
    if fp8:
        ks = float(kc.float().abs().max()) / 448.0
        vs = float(vc.float().abs().max()) / 448.0
        kc8 = O.fp8_sat(kc.float() / ks, "e4m3")
        vc8 = O.fp8_sat(vc.float() / vs, "e4m3")
        want = O.verify_attention(q, kc8, vc8, bt, ctxs, qsl, scale, ks, vs)
        kw = dict(k_scale=torch.tensor([ks], device=DEV), v_scale=torch.tensor([vs], device=DEV))
        kc_dev, vc_dev = kc8.to(DEV), vc8.to(DEV)
    else:
        want = O.verify_attention(q, kc, vc, bt, ctxs, qsl, scale)
        kc_dev, vc_dev = kc.to(DEV), vc.to(DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    for host_lens in (None, q_lens):
        got = _ops().verify_attention(q.to(DEV), kc_dev, vc_dev, bt.to(DEV), seq, torch.tensor(qsl, device=DEV), max(q_lens), max(ctxs),
                                      scale, q_lens_host=host_lens, **kw).float().cpu()
        assert torch.isfinite(got).all()
        assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), (shape, fp8, host_lens is not None, (got - want).abs().max())


def test_verify_attention_long_draft_dma_duty_patterns():
    """Who issues the tile DMA in a long-draft workgroup is a choice (aic_debug_attn_long_dma: every wave its quarter / the waves
    with one row tile more issue nothing / they keep their K pieces / wave 0 keeps one piece): every pattern must give the same
    attention, long drafts alone and beside short requests, at 5, 6, 9 and 10 row tiles (17-20, 21-24, 33-36, 37-40 tokens at
    G = 4), contexts that end inside a tile and inside a page, against the oracle."""
    from arcticinference_amd import _native as N
    B, Hq, Hkv, D, bs = 9, 32, 8, 128, 16
    q_lens = [17, 20, 22, 24, 33, 36, 38, 1, 4]
    ctxs = [1100, 97, 2079, 513, 1057, 4100, 640, 300, 77]
    q, kc, vc, bt, qsl = _attn_case(B, Hq, Hkv, D, q_lens, ctxs, bs, seed=11)
    want = O.verify_attention(q, kc, vc, bt, ctxs, qsl, D ** -0.5)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    outs = {}
    try:
        for pat in (0, 1, 2, 3):
            N.lib().aic_debug_attn_long_dma(pat)
            outs[pat] = _ops().verify_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV), seq, torch.tensor(qsl, device=DEV),
                                                max(q_lens), max(ctxs), D ** -0.5, q_lens_host=q_lens).float().cpu()
    finally:
        N.lib().aic_debug_attn_long_dma(1)
    for pat, got in outs.items():
        assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), (pat, (got - want).abs().max())
        assert torch.equal(got, outs[1]), pat          # the same arithmetic in the same order: only the data movement differs
    # long drafts alone (the two-launch / long-only form)
    sel = [i for i, ql in enumerate(q_lens) if ql > 8]
    rows = np.concatenate([np.arange(qsl[i], qsl[i + 1]) for i in sel])
    ql2 = [q_lens[i] for i in sel]
    qsl2 = np.concatenate([[0], np.cumsum(ql2)]).astype(np.int32)
    got = _ops().verify_attention(q[rows].to(DEV), kc.to(DEV), vc.to(DEV), bt[sel].to(DEV), seq[sel], torch.tensor(qsl2, device=DEV),
                                  max(ql2), max(ctxs), D ** -0.5, q_lens_host=ql2).float().cpu()
    assert torch.allclose(got, want[rows], atol=1e-3, rtol=2 ** -8), (got - want[rows]).abs().max()


@pytest.mark.parametrize("Hkv", [8, 1])
def test_verify_attention_long_draft_split_counts(Hkv):
    """The long-draft part of a one-grid call takes 2-32 token-range splits by the load of the call (a rank of SP = 8 sees
    one kv head: many short ranges).  Any count must give the same attention: forced counts 1, 3, 16, 32 (ranges down to
    4 tiles, ragged last ranges) and the built-in choice, against the generic path and the oracle."""
    from arcticinference_amd import _native as N
    torch.manual_seed(5)
    B, G, D, bs = 24, 4, 128, 16
    Hq = Hkv * G
    rng = np.random.RandomState(3)
    q_lens = [1] * 15 + [4] * 5 + [33, 20, 12, 9]
    rng.shuffle(q_lens)
    ctxs = [int(x) for x in rng.randint(900, 4353, size=B)]
    ctxs[int(np.argmax(q_lens))] = 4301
    max_blocks = max((c + bs - 1) // bs for c in ctxs)
    nb = B * max_blocks
    bt = torch.randperm(nb).view(B, max_blocks).to(torch.int32)
    kc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(sum(q_lens), Hq, D, device=DEV, dtype=torch.bfloat16)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32), device=DEV)
    seq = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    run = lambda **kw: _ops().verify_attention(q, kc, vc, bt.to(DEV), seq, qsl, max(q_lens), max(ctxs), D ** -0.5, **kw).float()
    try:
        outs = {}
        for n in (0, 1, 3, 16, 32):
            N.lib().aic_debug_attn_long_splits(n)
            outs[n] = run(q_lens_host=q_lens)
    finally:
        N.lib().aic_debug_attn_long_splits(0)
    generic = run()
    for n, o in outs.items():
        assert torch.allclose(o, generic, atol=1e-3, rtol=2 ** -8), (n, (o - generic).abs().max())
    assert not torch.equal(outs[1], outs[32])            # the counts did change the reduction order
    qs = qsl.cpu().numpy()
    for i in np.argsort(q_lens)[-4:]:
        rows = slice(int(qs[i]), int(qs[i + 1]))
        want = O.verify_attention(q[rows].cpu(), kc.cpu(), vc.cpu(), bt[i:i + 1], [ctxs[i]],
                                  np.array([0, q_lens[i]], dtype=np.int32), D ** -0.5, 1.0, 1.0)
        for n in (0, 32):
            got = outs[n][rows].cpu()
            assert torch.allclose(got, want, atol=1e-3, rtol=2 ** -8), (i, n, (got - want).abs().max())


# ------------------------------------------------------------------------------------------------
# A7-A10 LSTM speculator
# ------------------------------------------------------------------------------------------------
def _check_tokens(got, want_toks, want_logits, tag, ulps=1, rerun=None, count_by_near_ties=False):
    """EVERY head of EVERY row: the kernel's token must be the oracle's, or sit within `ulps` bf16 steps of the oracle's
    top logit (accumulation order differs between MFMA tiles and the CPU GEMM; on the fp8 head one bf16 ulp in an
    activation can flip its e4m3 code, a 6 % step on that element).  After a row parts from the oracle at such a
    near-tie its later heads follow the kernel's token, so the oracle is re-run teacher-forced with the kernel's tokens
    (`rerun(got) -> logits per head`) and every head is judged against the logits of its own prefix.
    fp8 head: 4 steps (measured: a flipped e4m3 activation code moved a logit by 3 bf16 steps at |logit| = 2.4; with 2 the
    test failed on that); bf16 head: 1 step."""
    B, k = want_toks.shape
    if not torch.equal(got, want_toks) and rerun is not None:
        want_logits = rerun(got)
    bad = near = 0
    for b in range(B):
        for h in range(k):
            lg = want_logits[h][b].float()
            top2 = torch.topk(lg, 2).values
            near += float(top2[0] - top2[1]) <= max(abs(float(top2[0])), 1e-3) * 2 ** -7 * ulps
            if int(got[b, h]) == int(torch.argmax(lg)):
                continue
            top, mine = float(lg.max()), float(lg[int(got[b, h])])
            assert top - mine <= max(abs(top), 1e-3) * 2 ** -7 * ulps, f"{tag}: row {b} head {h}: {mine} vs max {top}"
            bad += 1
    # how many may part: a tenth of the heads — or, at the full vocabulary (128256 candidates: the oracle's own top-2 gap is
    # inside the tolerance for a large share of the rows, counted above as `near`), three quarters of the pairs that CAN
    assert bad <= (max(1, B * k // 10, near * 3 // 4) if count_by_near_ties else max(1, B * k // 10)), \
        f"{tag}: {bad} near-tie mismatches ({near} of {B * k} oracle pairs are near-ties)"


@pytest.mark.parametrize("B,fp8", [(1, False), (5, True), (16, True), (33, False), (64, False), (24, True)])
def test_lstm_speculator_small(B, fp8):
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=3000, input_hidden_dim=768, inner_dim="512", emb_dim="512", proj_dim="512",
                               n_predict=3, num_lookahead_tokens=3)
    ck = random_lstm_weights(cfg, seed=1, std=0.05)
    m = ArcticLSTMSpeculator(cfg, max_num_seqs=64, device=DEV, quantize_lm_head=fp8)
    m.load_weights(ck.items())
    g = torch.Generator().manual_seed(B)
    hidden = torch.randn(B, 768, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, 3000, (B,), generator=g)
    use_fp8 = fp8 and (16 if B <= 16 else 32 if B <= 32 else 64) <= 32
    want, logits = O.lstm_generate_proposals(O.merge_lstm_checkpoint(ck), ids, hidden, 3, 3, True, fp8_head=use_fp8,
                                             return_logits=True)
    got = m.generate_proposals(ids.to(DEV), hidden.to(DEV), 3).cpu()
    assert got.shape == (B, 3) and got.dtype == torch.int64
    rerun = lambda forced: O.lstm_generate_proposals(O.merge_lstm_checkpoint(ck), ids, hidden, 3, 3, True, fp8_head=use_fp8,
                                                     return_logits=True, forced_tokens=forced)[1]
    _check_tokens(got, want, logits, f"B={B} fp8={use_fp8}", ulps=4 if use_fp8 else 1, rerun=rerun)


@pytest.mark.parametrize("B,fp8,Ds,H,V", [(3, False, 512, 768, 3000), (20, True, 512, 768, 3000), (20, False, 512, 768, 3000),
                                           (64, True, 512, 768, 3000), (13, True, 1024, 512, 50000), (40, False, 1024, 1024, 50000)])
def test_lstm_fused_schedule_equals_head_by_head(B, fp8, Ds, H, V):
    """The whole-draft entry point's fused schedule (LM head of head h + gate projection of head h + 1 in one launch, the
    arg-max finished inside the next cell launch: 12-15 launches at k = 3) and the head-by-head one (16-19) are the same
    arithmetic operation by operation: tokens AND bf16-rounded maximum logits are bit-identical, with and without the
    hidden-state row index."""
    import ctypes
    from arcticinference_amd import _native as N
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=V, input_hidden_dim=H, inner_dim=str(Ds), emb_dim=str(Ds), proj_dim=str(Ds),
                               n_predict=3, num_lookahead_tokens=3)
    m = ArcticLSTMSpeculator(cfg, max_num_seqs=64, device=DEV, quantize_lm_head=fp8)
    m.load_weights(random_lstm_weights(cfg, seed=5, std=0.05).items())
    g = torch.Generator().manual_seed(B)
    hidden = torch.randn(2 * B, H, generator=g).to(torch.bfloat16).to(DEV)
    ids = torch.randint(0, V, (B,), generator=g).to(torch.int32).to(DEV)
    hidx = torch.randperm(2 * B, generator=g)[:B].to(torch.int32).to(DEV)
    lib = N.lib()

    def run(mode, index):
        lib.aic_debug_lstm_fused(mode)
        toks = torch.full((B, 3), -1, dtype=torch.int64, device=DEV)
        vals = torch.full((B, 3), float("nan"), dtype=torch.float32, device=DEV)
        N.check(lib.aic_lstm_propose(m._h, hidden.data_ptr(), index.data_ptr() if index is not None else None, ids.data_ptr(),
                                     B, 3, toks.data_ptr(), vals.data_ptr(), N.current_stream_ptr()))
        torch.cuda.synchronize()
        return toks.cpu(), vals.cpu()

    try:
        for index in (None, hidx):
            t0, v0 = run(0, index)
            for mode in (1, 2):        # 2: the fp8 head quantises its activations on the way into LDS (built, not the default)
                # r04: the cell as ONE launch (parts of a row and of the batch meet at device-memory counters inside it;
                # the default) against gates / state / quantisation as three launches: same code cut at the counters
                for cell_launches in (3, 1):
                    lib.aic_debug_lstm_cell_launches(cell_launches)
                    t1, v1 = run(mode, index)
                    assert torch.equal(t0, t1) and torch.equal(v0, v1), (B, fp8, mode, cell_launches, index is not None)
            assert (t0 >= 0).all() and (t0 < V).all() and not torch.isnan(v0).any()
    finally:
        lib.aic_debug_lstm_fused(1)
        lib.aic_debug_lstm_cell_launches(1)


def test_lstm_speculator_full_size():
    """The 8B speculator's shapes (Ds = 4096, hidden 4096, vocab 128256), B = 8 rows, bf16 head: tokens against the
    CPU oracle (a few seconds of CPU GEMM)."""
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=128256, input_hidden_dim=4096)
    ck = random_lstm_weights(cfg, seed=4, std=0.02)
    m = ArcticLSTMSpeculator(cfg, max_num_seqs=8, device=DEV, quantize_lm_head=False)
    m.load_weights(ck.items())
    g = torch.Generator().manual_seed(8)
    hidden = torch.randn(8, 4096, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, 128256, (8,), generator=g)
    want, logits = O.lstm_generate_proposals(O.merge_lstm_checkpoint(ck), ids, hidden, 3, 3, True, fp8_head=False,
                                             return_logits=True)
    got = m.generate_proposals(ids.to(DEV), hidden.to(DEV), 3).cpu()
    rerun = lambda forced: O.lstm_generate_proposals(O.merge_lstm_checkpoint(ck), ids, hidden, 3, 3, True, fp8_head=False,
                                                     return_logits=True, forced_tokens=forced)[1]
    _check_tokens(got, want, logits, "full size", rerun=rerun)


@pytest.mark.parametrize("B", [8, 32])
def test_lstm_speculator_full_size_fp8_head(B):
    """BASELINE configs[1]'s own arithmetic at its own size: the per-tensor fp8 e4m3 LM head (W8A8, dynamic activation
    scale: arctic_speculator.py:706-751, fp8.py:207-223, :276-308) at V = 128256, Ds = 4096 — the 1.576 GB per call that
    bench.py's `roofline_draft_model` times — for B = 8 and B = 32 rows (32 = the widest batch that still takes the fp8
    head, arctic_speculator.py:726-728), every head of every row against the CPU oracle, teacher-forced after a
    near-tie.  Parity unpinned against a running reference (vLLM not importable); the oracle restates its op sequence."""
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=128256, input_hidden_dim=4096)
    ck = random_lstm_weights(cfg, seed=6, std=0.02)
    m = ArcticLSTMSpeculator(cfg, max_num_seqs=32, device=DEV, quantize_lm_head=True)
    m.load_weights(ck.items())
    g = torch.Generator().manual_seed(100 + B)
    hidden = torch.randn(B, 4096, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, 128256, (B,), generator=g)
    w = O.merge_lstm_checkpoint(ck)
    want, logits = O.lstm_generate_proposals(w, ids, hidden, 3, 3, True, fp8_head=True, return_logits=True)
    # the whole-draft entry point, with the bf16-rounded maximum logit of every head beside its token
    from arcticinference_amd import _native as N
    got_d = torch.full((B, 3), -1, dtype=torch.int64, device=DEV)
    vals_d = torch.full((B, 3), float("nan"), dtype=torch.float32, device=DEV)
    hid_d, ids_d = hidden.to(DEV), ids.to(torch.int32).to(DEV)
    N.check(N.lib().aic_lstm_propose(m._h, hid_d.data_ptr(), None, ids_d.data_ptr(), B, 3, got_d.data_ptr(), vals_d.data_ptr(),
                                     N.current_stream_ptr()))
    torch.cuda.synchronize()
    got, vals = got_d.cpu(), vals_d.cpu()
    assert torch.equal(got, m.generate_proposals(ids.to(DEV), hid_d, 3).cpu())
    assert got.shape == (B, 3) and (got >= 0).all() and (got < 128256).all()
    rerun = lambda forced: O.lstm_generate_proposals(w, ids, hidden, 3, 3, True, fp8_head=True, return_logits=True,
                                                     forced_tokens=forced)[1]
    _check_tokens(got, want, logits, f"full size fp8 head B={B}", ulps=4, rerun=rerun, count_by_near_ties=True)
    # the VALUE of the fused head: the kernel's maximum logit against the oracle's logit of the SAME token, teacher-forced
    # with the kernel's tokens (so every head is judged on its own prefix) — the arithmetic of the W8A8 GEMM + arg-max
    # itself, whatever a near-tie did to the token.  4 bf16 steps, as for the tokens (a flipped e4m3 activation code).
    forced_logits = logits if torch.equal(got, want) else rerun(got)
    for b in range(B):
        for h in range(3):
            ref = float(forced_logits[h][b][int(got[b, h])])
            assert abs(float(vals[b, h]) - ref) <= max(abs(ref), 1e-3) * 2 ** -7 * 4, (b, h, float(vals[b, h]), ref)
    assert m.quantize_lm_head


@pytest.mark.parametrize("B,tie,scale_input,fp8", [(3, False, False, False), (20, True, True, False), (64, False, True, False),
                                                    (12, True, False, True)])
def test_mlp_speculator(B, tie, scale_input, fp8):
    """ArcticMLPSpeculator (arctic_speculator.py:102-401) against the CPU bf16 op-sequence oracle: tied and untied
    stages, with and without the input norm, bf16 and fp8 LM head."""
    from arcticinference_amd.speculator import ArcticMLPSpeculator, MLPSpeculatorConfig, random_mlp_weights
    cfg = MLPSpeculatorConfig(vocab_size=3000, emb_dim=768, inner_dim=512, n_predict=3, num_lookahead_tokens=3,
                              tie_weights=tie, scale_input=scale_input)
    ck = random_mlp_weights(cfg, seed=2, std=0.05)
    m = ArcticMLPSpeculator(cfg, max_num_seqs=64, device=DEV, quantize_lm_head=fp8)
    m.load_weights(ck.items())
    g = torch.Generator().manual_seed(B)
    hidden = torch.randn(B, 768, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, 3000, (B,), generator=g)
    use_fp8 = fp8 and (16 if B <= 16 else 32 if B <= 32 else 64) <= 32
    want, logits = O.mlp_generate_proposals(ck, ids, hidden, 3, 3, 512, tie, scale_input, fp8_head=use_fp8, return_logits=True)
    got = m.generate_proposals(ids.to(DEV), hidden.to(DEV), 3).cpu()
    assert got.shape == (B, 3) and got.dtype == torch.int64
    rerun = lambda forced: O.mlp_generate_proposals(ck, ids, hidden, 3, 3, 512, tie, scale_input, fp8_head=use_fp8,
                                                    return_logits=True, forced_tokens=forced)[1]
    _check_tokens(got, want, logits, f"mlp B={B} tie={tie}", ulps=4 if use_fp8 else 1, rerun=rerun)
    with pytest.raises(ValueError):
        m.generate_proposals(ids.to(DEV), hidden.to(DEV), 4)


@pytest.mark.parametrize("tie", [True, False])
def test_lstm_speculator_method_sum_rnn(tie):
    """ArcticLSTMSpeculator with method "sum_rnn" (the reference's default, arctic_speculator.py:441,691-703): the
    Sequential-named checkpoint (`emb.{i}.0.weight`, `proj.{i}.0.weight`, `ln.{i}.0.*`) loads through the LSTM class's
    registry name and gives the op-sequence oracle's tokens (stacked stages: the next test)."""
    from arcticinference_amd.speculator import (ArcticSumRNNSpeculator, LSTMSpeculatorConfig, MLPSpeculatorConfig,
                                                lstm_family_speculator, random_mlp_weights)
    V, H, Ds, B = 3000, 768, 512, 9
    ck = random_mlp_weights(MLPSpeculatorConfig(vocab_size=V, emb_dim=H, inner_dim=Ds, n_predict=3, num_lookahead_tokens=3,
                                                tie_weights=tie, scale_input=True), seed=4, std=0.05)
    seq = {}
    for k, v in ck.items():          # the names the reference's nn.Sequential wrappers give the same tensors
        parts = k.split(".")
        seq["speculator." + (".".join(parts[:2] + ["0"] + parts[2:]) if parts[0] in ("emb", "proj", "ln") else k)] = v
    cfg = LSTMSpeculatorConfig(vocab_size=V, input_hidden_dim=H, inner_dim=str(Ds), emb_dim=str(Ds), proj_dim=str(Ds),
                               n_predict=3, num_lookahead_tokens=3, tie_weights=tie, scale_input=True, method="sum_rnn")
    m = lstm_family_speculator(cfg, max_num_seqs=16, device=DEV, quantize_lm_head=False)
    assert isinstance(m, ArcticSumRNNSpeculator)
    m.load_weights(seq.items())
    g = torch.Generator().manual_seed(3)
    hidden = torch.randn(B, H, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, V, (B,), generator=g)
    want, logits = O.mlp_generate_proposals(ck, ids, hidden, 3, 3, Ds, tie, True, fp8_head=False, return_logits=True)
    got = m.generate_proposals(ids.to(DEV), hidden.to(DEV), 3).cpu()
    rerun = lambda forced: O.mlp_generate_proposals(ck, ids, hidden, 3, 3, Ds, tie, True, fp8_head=False, return_logits=True,
                                                    forced_tokens=forced)[1]
    _check_tokens(got, want, logits, f"sum_rnn tie={tie}", ulps=1, rerun=rerun)


@pytest.mark.parametrize("stacks,tie,B,fp8", [((1, 1, 1), True, 9, False), ((2, 0, 1), False, 20, False), ((0, 1, 2), True, 5, True),
                                              ((1, 0, 0), True, 33, False)])
def test_lstm_speculator_sum_rnn_stacked_stages(stacks, tie, B, fp8):
    """sum_rnn with multi-entry dimension lists (arctic_speculator.py:478-542): extra (LayerNorm, GELU, Linear) stages
    inside emb / proj and (GELU, Linear, LayerNorm) stages inside ln, with the reference's Sequential parameter names,
    against the CPU bf16 op-sequence oracle — tied and untied, bf16 and fp8 head."""
    from arcticinference_amd.speculator import (ArcticSumRNNSpeculator, LSTMSpeculatorConfig, MLPSpeculatorConfig,
                                                lstm_family_speculator, random_mlp_weights)
    V, H, Ds, k = 3000, 768, 512, 3
    ck = random_mlp_weights(MLPSpeculatorConfig(vocab_size=V, emb_dim=H, inner_dim=Ds, n_predict=k, num_lookahead_tokens=k,
                                                tie_weights=tie, scale_input=True), seed=7, std=0.05)
    g = torch.Generator().manual_seed(11)
    lin = lambda: (torch.randn(Ds, Ds, generator=g) * 0.06).to(torch.bfloat16)
    lnw = lambda: (1.0 + 0.1 * torch.randn(Ds, generator=g)).to(torch.bfloat16)
    lnb = lambda: (0.1 * torch.randn(Ds, generator=g)).to(torch.bfloat16)
    stages = lambda kind: (range(2) if kind == "proj" else range(1)) if tie else range(k)
    for kind, n in (("emb", stacks[0]), ("proj", stacks[1])):
        for i in stages(kind):
            for j in range(1, n + 1):
                ck[f"{kind}.{i}.{3 * j - 2}.weight"], ck[f"{kind}.{i}.{3 * j - 2}.bias"] = lnw(), lnb()
                ck[f"{kind}.{i}.{3 * j}.weight"] = lin()
    for i in stages("ln"):
        for j in range(1, stacks[2] + 1):
            ck[f"ln.{i}.{3 * j - 1}.weight"] = lin()
            ck[f"ln.{i}.{3 * j}.weight"], ck[f"ln.{i}.{3 * j}.bias"] = lnw(), lnb()
    seq = {}
    for name, v in ck.items():          # the reference's names: the base module of a Sequential is its index 0
        parts = name.split(".")
        seq["speculator." + (".".join(parts[:2] + ["0"] + parts[2:]) if parts[0] in ("emb", "proj", "ln") and len(parts) == 3
                             else name)] = v
    dims = lambda n: ".".join([str(Ds)] * (n + 1))
    cfg = LSTMSpeculatorConfig(vocab_size=V, input_hidden_dim=H, inner_dim=dims(stacks[2]), emb_dim=dims(stacks[0]),
                               proj_dim=dims(stacks[1]), n_predict=k, num_lookahead_tokens=k, tie_weights=tie,
                               scale_input=True, method="sum_rnn")
    m = lstm_family_speculator(cfg, max_num_seqs=64, device=DEV, quantize_lm_head=fp8)
    assert isinstance(m, ArcticSumRNNSpeculator) and m.stacks == stacks
    m.load_weights(seq.items())
    hidden = torch.randn(B, H, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, V, (B,), generator=g)
    use_fp8 = fp8 and B <= 32
    kw = dict(fp8_head=use_fp8, return_logits=True, stacks=stacks)
    want, logits = O.mlp_generate_proposals(ck, ids, hidden, k, k, Ds, tie, True, **kw)
    got = m.generate_proposals(ids.to(DEV), hidden.to(DEV), k).cpu()
    rerun = lambda forced: O.mlp_generate_proposals(ck, ids, hidden, k, k, Ds, tie, True, forced_tokens=forced, **kw)[1]
    _check_tokens(got, want, logits, f"stacked sum_rnn {stacks} tie={tie}", ulps=4 if use_fp8 else 1, rerun=rerun)


def test_mlp_speculator_sharded_embedding_c9():
    """C9 (vocab_parallel_embedding.py:161-178,425-444): the MLP speculator's token embedding sharded over the TP group.
    (1) every rank's masked lookup, summed (the all-reduce), is the full table's row, bit for bit; (2) a model that is fed
    looked-up rows gives the tokens of the model that gathers from its own table; (3) two ranks (vocab-sharded LM head AND
    embedding, both "ranks" on this GPU, collectives emulated) reproduce the single-rank tokens."""
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, ArcticMLPSpeculator, MLPSpeculatorConfig, random_mlp_weights
    from arcticinference_amd import _native as N
    cfg = MLPSpeculatorConfig(vocab_size=3001, emb_dim=768, inner_dim=512, n_predict=3, num_lookahead_tokens=3,
                              tie_weights=False, scale_input=True)
    ck = random_mlp_weights(cfg, seed=6, std=0.05)
    B, k = 10, 3
    g = torch.Generator().manual_seed(1)
    hidden = torch.randn(B, 768, generator=g).to(torch.bfloat16).to(DEV)
    ids = torch.randint(0, 3001, (B,), generator=g).to(DEV)
    ids[0], ids[1] = 0, 3000                      # first row of rank 0's shard, last real row of the last shard
    full = ArcticMLPSpeculator(cfg, max_num_seqs=16, device=DEV, quantize_lm_head=False)
    full.load_weights(ck.items())
    want = full.generate_proposals(ids, hidden, k).cpu()
    # (2) one rank, rows looked up outside the fused kernel
    ext = ArcticMLPSpeculator(cfg, max_num_seqs=16, device=DEV, quantize_lm_head=False, shard_embedding=True)
    ext.load_weights(ck.items())
    assert torch.equal(ext.generate_proposals(ids, hidden, k).cpu(), want)
    # (1) + (3) two ranks
    ranks = []
    for r in range(2):
        m = ArcticMLPSpeculator(cfg, max_num_seqs=16, tp_size=2, tp_rank=r, device=DEV, quantize_lm_head=False,
                                all_reduce=lambda z: None)
        m.load_weights(ck.items())
        assert m.shard_embedding and m.weights["emb"][0].shape[0] == m.shard_rows < 3001
        ranks.append(m)
    for head in range(k):
        parts = [m.embedding_rows(head, ids.to(torch.int32), B).clone() for m in ranks]
        assert torch.equal((parts[0].float() + parts[1].float()).to(torch.bfloat16).cpu(), ck[f"emb.{head}.weight"][ids.cpu()])
        assert int((parts[0].abs().sum(1) > 0).sum() + (parts[1].abs().sum(1) > 0).sum()) == B     # each row from ONE rank
    for m in ranks:
        m.begin(hidden, None, B)
    last, outs = ids.to(torch.int32), []
    for head in range(k):
        total = sum(m.embedding_rows(head, last, B).float() for m in ranks).to(torch.bfloat16)       # the all-reduce
        packed = []
        for m in ranks:
            m._z[:B].copy_(total)
            N.check(N.lib().aic_mlp_set_embedding_rows(m._h, m._z.data_ptr()))
            tok, val = ArcticLSTMSpeculator.head_step(m, head, last, B)
            packed.append(torch.stack([val.to(torch.float64).view(torch.int64), tok]))
        nxt = ArcticLSTMSpeculator.pick_global(torch.stack(packed))
        outs.append(nxt.unsqueeze(1))
        last = nxt.to(torch.int32)
    assert torch.equal(torch.cat(outs, dim=-1).cpu(), want)


def test_lstm_hidden_index_and_errors():
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=1000, input_hidden_dim=512, inner_dim="512", emb_dim="512", proj_dim="512",
                               n_predict=3, num_lookahead_tokens=3)
    ck = random_lstm_weights(cfg, seed=2, std=0.05)
    m = ArcticLSTMSpeculator(cfg, max_num_seqs=8, device=DEV, quantize_lm_head=False)
    with pytest.raises(RuntimeError):
        m.generate_proposals(torch.zeros(2, dtype=torch.long, device=DEV), torch.zeros(2, 512, device=DEV), 3)
    m.load_weights(ck.items())
    with pytest.raises(ValueError):
        m.generate_proposals(torch.zeros(2, dtype=torch.long, device=DEV), torch.zeros(2, 512, device=DEV), 4)
    g = torch.Generator().manual_seed(0)
    pool = torch.randn(20, 512, generator=g).to(torch.bfloat16)
    idx = torch.tensor([7, 0, 19, 3])
    ids = torch.randint(0, 1000, (4,), generator=g)
    a = m.generate_proposals(ids.to(DEV), pool.to(DEV), 3, hidden_index=idx.to(DEV)).cpu()
    b = m.generate_proposals(ids.to(DEV), pool[idx].to(DEV), 3).cpu()
    assert torch.equal(a, b)
    # HIP-graph replay (opt-in) gives the same tokens, with and without the row index
    assert not m.use_graph
    m.use_graph = True
    c = m.generate_proposals(ids.to(DEV), pool[idx].to(DEV), 3).cpu()
    c2 = m.generate_proposals(ids.to(DEV), pool.to(DEV), 3, hidden_index=idx.to(DEV)).cpu()
    assert torch.equal(b, c) and torch.equal(c, c2)


@pytest.mark.parametrize("fp8", [False, True])
def test_lstm_graph_replay_equals_eager(fp8):
    """The draft as one HIP graph launch (use_graph=True; arctic_speculator.py:806-842 always replays a graph — here it is
    opt-in, measured slower than eager launches on this runtime, see speculator.py::_replay_graph): bit-identical tokens to the eager launches for changing batch sizes (exact-size keys: no stale padded rows in
    the fp8 head's activation scale), with and without the hidden-state row index, for two different hidden tensors (the
    index form's graphs read the caller's tensor in place: its address is part of the key), for k = 2 and 3, interleaved
    so that every graph is replayed after others ran on the same static buffers."""
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=5000, input_hidden_dim=768, inner_dim="512", emb_dim="512", proj_dim="512",
                               n_predict=3, num_lookahead_tokens=3)
    ck = random_lstm_weights(cfg, seed=9, std=0.05)
    mg = ArcticLSTMSpeculator(cfg, max_num_seqs=64, device=DEV, quantize_lm_head=fp8, use_graph=True)
    me = ArcticLSTMSpeculator(cfg, max_num_seqs=64, device=DEV, quantize_lm_head=fp8)
    for m in (mg, me):
        m.load_weights(ck.items())
    assert mg.use_graph and not me.use_graph
    g = torch.Generator().manual_seed(3)
    pools = [torch.randn(200, 768, generator=g).to(torch.bfloat16).to(DEV) for _ in range(2)]
    cases = []
    for rnd in range(3):
        for B in (1, 7, 32, 33, 20, 64, 7):
            for k in (3, 2):
                cases.append((B, k, rnd % 2, (B + rnd) % 3 != 0))
    for B, k, which, with_index in cases:
        ids = torch.randint(0, 5000, (B,), generator=g).to(DEV)
        idx = torch.randperm(200, generator=g)[:B].to(DEV)
        if with_index:
            a = mg.generate_proposals(ids, pools[which], k, hidden_index=idx)
            b = me.generate_proposals(ids, pools[which], k, hidden_index=idx)
        else:
            rows = pools[which][idx.long()]
            a = mg.generate_proposals(ids, rows, k)
            b = me.generate_proposals(ids, rows, k)
        assert a.shape == (B, k) and torch.equal(a.cpu(), b.cpu()), (B, k, which, with_index)
    assert 0 < len(mg._graphs) <= mg._MAX_GRAPHS and len(me._graphs) == 0


def test_quantize_fp8_per_tensor():
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(257, 513, generator=g) * 3).to(torch.bfloat16)
    q_ref, s_ref = O.fp8_quant_per_tensor(x)
    q, s = _ops().quantize_fp8_per_tensor(x.to(DEV))
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))


def test_rejection_no_drafts():
    """A step in which no request has drafts (the first decode step): only bonus tokens come out."""
    B, V = 4, 1000
    cu = torch.zeros(B, dtype=torch.int32, device=DEV)
    res = _ops().rejection_sample(torch.zeros(0, V, dtype=torch.bfloat16, device=DEV),
                                  torch.zeros(0, dtype=torch.int32, device=DEV), cu,
                                  torch.tensor([5, 6, 7, 8], dtype=torch.int32, device=DEV), 1)
    assert res.output_token_ids.cpu().tolist() == [[5, -1], [6, -1], [7, -1], [8, -1]]
    assert res.hidden_index.cpu().tolist() == [0, 1, 2, 3] and res.last_token.cpu().tolist() == [5, 6, 7, 8]


def test_lstm_vocab_parallel_shards_agree_with_single_rank():
    """A11: the LM head sharded over tp=2 / tp=4 ranks (vocab padded to 64, even split, zero-filled tail), each
    shard's local (value, index) merged the way the all-gather + arg-max of arctic_speculator.py:733-744 does,
    must give the single-rank tokens.  Both "ranks" run on this one GPU; the collective is emulated by a stack."""
    from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
    cfg = LSTMSpeculatorConfig(vocab_size=3001, input_hidden_dim=512, inner_dim="512", emb_dim="512", proj_dim="512")
    ck = random_lstm_weights(cfg, seed=5, std=0.05)
    B, k = 6, 3
    g = torch.Generator().manual_seed(9)
    hidden = torch.randn(B, 512, generator=g).to(torch.bfloat16).to(DEV)
    ids = torch.randint(0, 3001, (B,), generator=g).to(DEV)
    single = ArcticLSTMSpeculator(cfg, max_num_seqs=8, device=DEV, quantize_lm_head=False)
    single.load_weights(ck.items())
    want = single.generate_proposals(ids, hidden, k).cpu()
    for tp in (2, 4):
        ranks = []
        for r in range(tp):
            m = ArcticLSTMSpeculator(cfg, max_num_seqs=8, tp_size=tp, tp_rank=r, device=DEV, quantize_lm_head=False)
            m.load_weights(ck.items())
            ranks.append(m)
        assert sum(m.shard_rows for m in ranks) == 3001 and ranks[0].shard_size * tp >= 3001
        for m in ranks:
            m.begin(hidden, None, B)
        last = ids.to(torch.int32)
        outs = []
        for head in range(k):
            parts = []
            for m in ranks:
                tok, val = m.head_step(head, last, B)
                parts.append(torch.stack([val.to(torch.float64).view(torch.int64), tok]))
            nxt = ArcticLSTMSpeculator.pick_global(torch.stack(parts))
            outs.append(nxt.unsqueeze(1))
            last = nxt.to(torch.int32)
        got = torch.cat(outs, dim=-1).cpu()
        assert torch.equal(got, want), (tp, got, want)


def test_profile_event_pairs_time_the_attention_launches():
    """bench.py's roofline instrument: with aic_profile_enable(n) every n-th attention launch is bracketed by an event pair
    on its stream; aic_profile_event_overhead is what a pair reads with nothing inside."""
    import ctypes
    from arcticinference_amd import _native as N
    ops = _ops()
    lib = N.lib()
    torch.manual_seed(0)
    B, ctx, Hq, Hkv, D, bs = 8, 512, 8, 2, 128, 16
    nblk = ctx // bs
    kv = torch.randn(2, B * nblk, bs, Hkv, D, device=DEV, dtype=torch.bfloat16)
    bt = torch.arange(B * nblk, device=DEV, dtype=torch.int32).view(B, nblk)
    q = torch.randn(B, Hq, D, device=DEV, dtype=torch.bfloat16)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    qsl = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    N.check(lib.aic_profile_enable(2))
    try:
        for _ in range(6):
            ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, 1, ctx, D ** -0.5)
        tot, n = ctypes.c_double(0), ctypes.c_int(0)
        N.check(lib.aic_profile_read(ctypes.byref(tot), ctypes.byref(n)))
    finally:
        N.check(lib.aic_profile_enable(0))
    assert n.value == 3 and 0.0 < tot.value / n.value < 5e3           # every 2nd of 6 launches, microseconds each
    mean, lo = ctypes.c_double(0), ctypes.c_double(0)
    N.check(lib.aic_profile_event_overhead(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 64, ctypes.byref(mean),
                                           ctypes.byref(lo)))
    assert 0.0 <= lo.value <= mean.value < 200.0
    assert lib.aic_profile_event_overhead(None, 0, ctypes.byref(mean), None) != 0      # pairs must be >= 1
