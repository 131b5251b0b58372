"""GPU: the SwiftKV hot-path pieces (arcticinference_amd/swiftkv.py) against the oracle's literal
torch expressions of llama_swiftkv.py:418-431,573-685: bulk KV write of all decode-half layers straight from the
strided [T, Lkv*H*D] projection output into real 4-D paged caches (through py_custom_ops.reshape_and_cache_flash_bulk,
the reference's own call site :599-628), metadata rewrite, and the one-launch gather of the five per-token tensors."""
import numpy as np
import pytest
import torch

from oracle import spec_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Meta:
    pass


def _case(T, Lkv, H, D, hidden, dtype, n_req, seed, pad=0):
    g = torch.Generator().manual_seed(seed)
    n = Lkv * H * D
    # the fused projection output: K/V states are column ranges of one wider activation (row stride > row width)
    wide = torch.randn(T, 2 * n + pad, generator=g).to(dtype)
    k_states, v_states = wide[:, :n], wide[:, n:2 * n]
    hid = torch.randn(T, hidden, generator=g).to(dtype)
    res = torch.randn(T, hidden, generator=g).to(dtype)
    pos = torch.randint(0, 5000, (T,), generator=g)
    bounds = np.sort(np.random.default_rng(seed).choice(np.arange(1, T), size=n_req - 1, replace=False))
    qsl = torch.tensor(np.concatenate([[0], bounds, [T]]), dtype=torch.int32)
    # sampled rows: the last token of every request, plus a few verify rows before it for some requests
    li = []
    for i in range(n_req):
        a, b = int(qsl[i]), int(qsl[i + 1])
        li += list(range(max(a, b - 1 - (i % 3)), b))
    li = torch.tensor(li, dtype=torch.int64)
    bs = 16
    nb = (T + bs - 1) // bs + 5
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    return wide, k_states, v_states, hid, res, pos, qsl, li, slots, nb, bs


@pytest.mark.parametrize("cfg", [
    dict(T=2, Lkv=4, H=2, D=16, hidden=32, dtype=torch.float32, kv="auto", n_req=2),          # test_custom_ops.py:56-99's shape
    dict(T=300, Lkv=16, H=8, D=128, hidden=512, dtype=torch.bfloat16, kv="auto", n_req=37),  # SwiftKV-8B: 16 of 32 layers
    dict(T=260, Lkv=16, H=1, D=128, hidden=512, dtype=torch.bfloat16, kv="fp8_e4m3", n_req=64, pad=8),   # SP = 8 slice, fp8 cache
    dict(T=97, Lkv=40, H=2, D=64, hidden=256, dtype=torch.float16, kv="fp8_e5m2", n_req=9),   # > 32 layers: two launches
])
def test_swiftkv_select_equals_oracle(cfg):
    from arcticinference_amd import py_custom_ops
    from arcticinference_amd.swiftkv import SwiftKVSelector, swiftkv_select
    T, Lkv, H, D = cfg["T"], cfg["Lkv"], cfg["H"], cfg["D"]
    wide, ks, vs, hid, res, pos, qsl, li, slots, nb, bs = _case(T, Lkv, H, D, cfg["hidden"], cfg["dtype"], cfg["n_req"], 3,
                                                                  cfg.get("pad", 0))
    cdt = {"auto": cfg["dtype"], "fp8_e4m3": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}[cfg["kv"]]
    caches = [torch.zeros(2, nb, bs, H, D, dtype=cdt) for _ in range(Lkv)]
    k_sc = [torch.tensor(0.5 + 0.01 * l) for l in range(Lkv)]
    v_sc = [torch.tensor(0.25 + 0.02 * l) for l in range(Lkv)]
    want_c = [c.clone() for c in caches]
    want_sel, want_qsl, want_slots = O.swiftkv_select(hid, res, pos, ks, vs, qsl, slots, li, [c[0] for c in want_c],
                                                      [c[1] for c in want_c], cfg["kv"], k_sc, v_sc, H, D)
    d = lambda t: t.to(DEV)
    wide_d = d(wide)
    n = Lkv * H * D
    ks_d, vs_d = wide_d[:, :n], wide_d[:, n:2 * n]
    assert ks_d.stride(0) == 2 * n + cfg.get("pad", 0)
    caches_d = [d(c) for c in caches]
    meta = Meta()
    meta.query_start_loc, meta.slot_mapping, meta.swiftkv_logits_indices = d(qsl), d(slots), d(li)
    meta.num_actual_tokens, meta.use_cascade, meta.cu_prefix_query_lens = T, True, "x"
    sel = SwiftKVSelector(cfg["hidden"], Lkv, H, D, cfg["dtype"], DEV)
    got = swiftkv_select(sel, d(hid), d(res), d(pos), ks_d, vs_d, meta, caches_d, cfg["kv"], [d(s) for s in k_sc],
                         [d(s) for s in v_sc])
    for a, b in zip(caches_d, want_c):
        assert torch.equal(a.cpu().view(torch.uint8), b.view(torch.uint8))          # bit-exact, fp8 included
    for a, b in zip(got, want_sel):
        assert a.shape == b.shape and torch.equal(a.cpu(), b)
    assert torch.equal(meta.query_start_loc.cpu(), want_qsl) and meta.query_start_loc.dtype == torch.int32
    assert torch.equal(meta.slot_mapping.cpu(), want_slots) and meta.num_actual_tokens == li.numel()
    assert meta.use_cascade is False and meta.cu_prefix_query_lens is None
    # the same write through the reference's own entry point (py_custom_ops wrapper, per call)
    caches2 = [d(c) for c in caches]
    py_custom_ops.reshape_and_cache_flash_bulk(ks_d, vs_d, [c[0] for c in caches2], [c[1] for c in caches2], d(slots), cfg["kv"],
                                               [d(s) for s in k_sc], [d(s) for s in v_sc], H, D)
    for a, b in zip(caches2, want_c):
        assert torch.equal(a.cpu().view(torch.uint8), b.view(torch.uint8))


def test_swiftkv_select_graph_buffers_and_capture_mode():
    from arcticinference_amd.swiftkv import SwiftKVSelector, swiftkv_select
    T, Lkv, H, D, hidden = 120, 4, 2, 64, 128
    wide, ks, vs, hid, res, pos, qsl, li, slots, nb, bs = _case(T, Lkv, H, D, hidden, torch.bfloat16, 12, 7)
    pad = lambda n: next(s for s in (8, 16, 32, 64) if s >= n)
    sel = SwiftKVSelector(hidden, Lkv, H, D, torch.bfloat16, DEV, cuda_graph_max_batch_size=64, pad_for_cudagraph=pad)
    d = lambda t: t.to(DEV)
    caches = [torch.zeros(2, nb, bs, H, D, dtype=torch.bfloat16, device=DEV) for _ in range(Lkv)]
    one = [torch.ones((), device=DEV)] * Lkv
    meta = Meta()
    meta.query_start_loc, meta.slot_mapping, meta.swiftkv_logits_indices = d(qsl), d(slots), d(li)
    out = swiftkv_select(sel, d(hid), d(res), d(pos), d(ks.contiguous()), d(vs.contiguous()), meta, caches, "auto", one, one)
    n = li.numel()
    assert n <= 64 and all(o.shape[0] == pad(n) for o in out)
    for o, name, src in zip(out, sel.NAMES, (hid, res, pos, ks, vs)):
        assert o.data_ptr() == sel.inputs[name].data_ptr()            # the decode runner's persistent buffers
        assert torch.equal(o[:n].cpu(), src.index_select(0, li))
    # graph capture / profile run: no metadata -> the buffers themselves, padded to the graph size
    cap = swiftkv_select(sel, d(hid)[:20], d(res)[:20], d(pos)[:20], d(ks.contiguous())[:20], d(vs.contiguous())[:20], None,
                         caches, "auto", one, one)
    assert all(c.shape[0] == 32 and c.data_ptr() == sel.inputs[nm].data_ptr() for c, nm in zip(cap, sel.NAMES))
    big = swiftkv_select(sel, d(hid), d(res), d(pos), d(ks.contiguous()), d(vs.contiguous()), None, caches, "auto", one, one)
    assert big[0].shape[0] == T                                         # larger than any captured size: passed through
    # a selection larger than the graph sizes: fresh tensors
    li2 = torch.arange(0, T - 10, dtype=torch.int64)
    got = sel.select((d(hid), d(res), d(pos), d(ks.contiguous()), d(vs.contiguous())), d(li2))
    assert got[0].shape[0] == T - 10 and torch.equal(got[3].cpu(), ks.index_select(0, li2))
    with pytest.raises(RuntimeError):
        from arcticinference_amd.swiftkv import row_gather
        row_gather([hid], [torch.empty_like(hid)], d(li))              # CPU tensors: no fallback


def test_bulk_kv_write_torch_op_compiles_fullgraph_on_the_gpu():
    """torch.ops.arctic_inference.reshape_and_cache_flash_bulk (reference schema, torch_bindings.cpp:5-18) on the MI355X:
    a caller compiled with fullgraph=True (Dynamo + AOT functionalisation of the mutated cache lists) writes the same
    bytes as the eager call and as the oracle, bf16 and fp8 caches; SwiftKVSelector.write_kv goes through the same op."""
    from arcticinference_amd import py_custom_ops
    from arcticinference_amd.swiftkv import SwiftKVSelector
    assert py_custom_ops.try_load_torch_library()
    L, H, D, nb, bs, T = 6, 8, 128, 40, 16, 200
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(T, 2 * L * H * D + 8, generator=g).to(torch.bfloat16)
    n = L * H * D
    slots = torch.randperm(nb * bs, generator=g)[:T]
    slots[5] = -1                                           # a padded token: skipped
    for kv, cdt in (("auto", torch.bfloat16), ("fp8_e4m3", torch.float8_e4m3fn)):
        ksc = [torch.tensor(0.5 + 0.03 * l) for l in range(L)]
        vsc = [torch.tensor(0.25 + 0.02 * l) for l in range(L)]
        want = [torch.zeros(2, nb, bs, H, D, dtype=cdt) for _ in range(L)]
        O.kv_bulk_write(wide[:, :n], wide[:, n:2 * n], [c[0] for c in want], [c[1] for c in want], slots, kv, ksc, vsc, H, D)
        wd = wide.to(DEV)
        kd, vd, sd = wd[:, :n], wd[:, n:2 * n], slots.to(DEV)
        ks_d, vs_d = [t.to(DEV) for t in ksc], [t.to(DEV) for t in vsc]

        def step(k, v, kcs, vcs, sl):
            py_custom_ops.reshape_and_cache_flash_bulk(k, v, kcs, vcs, sl, kv, ks_d, vs_d, H, D)
            return k.float().sum()

        torch._dynamo.reset()
        mk = lambda: [torch.zeros(2, nb, bs, H, D, dtype=cdt, device=DEV) for _ in range(L)]
        for fn in (step, torch.compile(step, fullgraph=True, backend="aot_eager")):
            caches = mk()
            fn(kd, vd, [c[0] for c in caches], [c[1] for c in caches], sd)
            for a, b in zip(caches, want):
                assert torch.equal(a.cpu().view(torch.uint8), b.view(torch.uint8)), kv
        caches = mk()
        sel = SwiftKVSelector(64, L, H, D, torch.bfloat16, DEV)
        sel.write_kv(kd, vd, caches, sd, kv, ks_d, vs_d)
        for a, b in zip(caches, want):
            assert torch.equal(a.cpu().view(torch.uint8), b.view(torch.uint8)), kv
    torch._dynamo.reset()
