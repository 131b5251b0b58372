"""GPU parity: the HIP suffix matcher (through the C ABI) against the golden vectors produced by the
real reference, and against the oracle on seeded replays up to the BASELINE size."""
import random

import pytest

import golden_utils as gu
from arcticinference_amd.suffix_cache import SuffixCache, SuffixTree
from arcticinference_amd.workload import TokenSource

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["suffix_traces.json", "suffix_ties.json", "suffix_clamps.json",
                                  "suffix_treespec.json"])
def test_device_tree_matches_golden(name):
    total = 0
    for case in gu.load(name):
        total += gu.replay_tree_case(case, SuffixTree)
    assert total > 100


def test_device_cache_matches_golden():
    total = sum(gu.replay_cache_case(c, SuffixCache) for c in gu.load("suffix_cache.json"))
    assert total > 100


def test_device_cache_errors_and_edges():
    c = SuffixCache(8)
    c.cache_prompt("a", [1, 2, 3])
    with pytest.raises(ValueError):
        c.speculate("zzz", [1])
    with pytest.raises(ValueError):
        c.speculate("a", [])
    # nothing cached in the global tree yet: empty result, not an error
    r = c.speculate("b", [1, 2], use_cached_prompt=False)
    assert r.token_ids == [] and r.score == 0.0 and r.match_len == 0
    r = c.speculate("a", [9, 9, 1, 2], max_spec_tokens=4)
    assert r.token_ids == [3] and r.match_len == 2
    # max_spec_tokens == 0 and patterns longer than max_depth
    assert c.speculate("a", [1, 2], max_spec_tokens=0).token_ids == []
    assert c.speculate("a", list(range(100, 130)) + [1, 2], max_spec_tokens=4).token_ids == [3]


def _replay(cache, cfg, batched):
    """simulator.suffix_decode-style replay (oracle/gen_golden.py:replay) on the device path."""
    import hashlib
    import struct
    src = TokenSource(seed=cfg["seed"])
    n_req = cfg["n_req"]
    reqs = [src.request(r, cfg["prompt_len"], cfg["gen_len"]) for r in range(n_req)]
    prompts = [[int(x) for x in p] for p, _ in reqs]
    gts = [[int(x) for x in g] for _, g in reqs]
    out = []
    # sequential per request reproduces the golden digests exactly (the global tree evolves in the same order)
    for r in range(n_req):
        cache.cache_prompt(r, prompts[r])
        h = hashlib.sha256()
        resp, steps, acc, spec = [], 0, 0, 0
        gt = gts[r]
        while len(resp) < len(gt):
            text = (prompts[r] + resp)[-cache.max_depth:]
            res = cache.speculate(r, text, max_spec_tokens=cfg["max_spec_tokens"], max_spec_factor=cfg["factor"],
                                  max_spec_offset=cfg["offset"], min_token_prob=cfg["min_token_prob"])
            h.update(struct.pack("<i", res.match_len))
            h.update(struct.pack("<f", res.score))
            h.update(struct.pack(f"<{len(res.token_ids)}i", *res.token_ids))
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
        out.append({"steps": steps, "accepted": acc, "speculated": spec, "sha256": h.hexdigest()})
    return out


def test_device_replay_small_digest():
    want = gu.load("suffix_replay.json")[0]
    got = _replay(SuffixCache(64), want["config"], batched=False)
    assert got == want["per_request"]


def test_device_replay_full_size_digest():
    """BASELINE size: 64 requests x (4096 prompt + 256 generated), digests from the real reference."""
    want = gu.load("suffix_replay.json")[1]
    got = _replay(SuffixCache(64), want["config"], batched=False)
    assert got == want["per_request"]
    assert sum(g["accepted"] for g in got) == want["sum_accept"]


def test_batched_step_equals_oracle():
    """The engine-step pattern: B live requests, one speculate_batch per step; vs the oracle cache."""
    from oracle.suffix_oracle import OracleSuffixCache
    src = TokenSource(seed=5)
    B, PL, GL = 16, 512, 48
    dev, orc = SuffixCache(64), OracleSuffixCache(64)
    data = [src.request(r, PL, GL) for r in range(B)]
    rows = [[int(x) for x in p] for p, _ in data]
    gts = [[int(x) for x in g] for _, g in data]
    dev.cache_prompts(list(range(B)), rows, n_threads=4)
    for r in range(B):
        orc.cache_prompt(r, rows[r])
    done = [0] * B
    rng = random.Random(0)
    for step in range(40):
        live = [r for r in range(B) if done[r] < GL]
        if not live:
            break
        for r in live:
            n_new = min(rng.randint(1, 4), GL - done[r])
            new = gts[r][done[r]:done[r] + n_new]
            dev.update_response(r, new)
            orc.update_response(r, new)
            rows[r].extend(new)
            done[r] += n_new
        pats = [rows[r][-64:] for r in live]
        res = dev.speculate_batch(live, pats, [32] * len(live), [1.0] * len(live), [0.0] * len(live),
                                  [0.1] * len(live), [True] * len(live))
        for r, p, got in zip(live, pats, res):
            want = orc.speculate(r, p, max_spec_tokens=32)
            gu.assert_cand(got, gu.cand_dict(want), ctx=f"step {step} req {r}")
    assert dev._global_tree().selfcheck() == 0


@pytest.mark.parametrize("depth", [3, 17, 128])
def test_device_matches_oracle_other_depths(depth):
    """max_depth other than 64: more than 64 suffix starts per query (the select kernel strides its lanes),
    budgets up to max_depth - 1, pool / hash growth in the HBM mirror while the tree grows."""
    from oracle.suffix_oracle import OracleSuffixTree
    rng = random.Random(depth)
    dev, orc = SuffixTree(depth), OracleSuffixTree(depth)
    vocab = 6 if depth > 16 else 3
    hist = {s: [] for s in range(3)}
    n = 0
    for step in range(1500 if depth > 16 else 300):
        s = rng.randrange(3)
        toks = [rng.randrange(vocab) for _ in range(rng.randint(1, 5))]
        dev.extend(s, toks)
        orc.extend(s, toks)
        hist[s].extend(toks)
        if step % 25 == 0:
            src = hist[rng.randrange(3)]
            if not src:
                continue
            pat = src[-rng.randint(1, min(len(src), depth + 5)):]
            args = (rng.choice([1, 8, depth, 2 * depth]), rng.choice([1.0, 2.0, 8.0]), rng.choice([0.0, 3.0]),
                    rng.choice([0.0, 0.05, 0.2]))
            gu.assert_cand(dev.speculate(pat, *args), gu.cand_dict(orc.speculate(pat, *args)), ctx=f"step {step}")
            n += 1
    assert n > 10 and dev.selfcheck() == 0


def test_device_large_batch_of_queries():
    """256 queries in one round trip (more than one residency of 4-wave blocks), prompt + global trees."""
    from oracle.suffix_oracle import OracleSuffixCache
    src = TokenSource(vocab_size=500, seed=9, n_motifs=10, motif_min=4, motif_max=10, p_motif=0.6)
    B = 256
    dev, orc = SuffixCache(32), OracleSuffixCache(32)
    rows = []
    for r in range(B):
        p, g = src.request(r, 48, 24)
        rows.append([int(x) for x in p] + [int(x) for x in g])
        for c in (dev, orc):
            c.cache_prompt(r, rows[r][:48])
            c.update_response(r, rows[r][48:60])
    res = dev.speculate_batch(list(range(B)), [r[:60] for r in rows], [16] * B, [2.0] * B, [1.0] * B, [0.05] * B,
                              [i % 3 != 0 for i in range(B)])
    hits = 0
    for r in range(B):
        want = orc.speculate(r, rows[r][:60], max_spec_tokens=16, max_spec_factor=2.0, max_spec_offset=1.0,
                             min_token_prob=0.05, use_cached_prompt=(r % 3 != 0))
        gu.assert_cand(res[r], gu.cand_dict(want), ctx=f"req {r}")
        hits += len(want.token_ids) > 0
    assert hits > 100


def test_simulator_path_mode_matches_oracle_driver():
    """The simulator with use_tree_spec=false (the HIP matcher) against the same driver on the oracle cache."""
    import pandas as pd
    from arcticinference_amd import simulator as S
    from oracle.suffix_oracle import OracleSuffixCache
    src = TokenSource(vocab_size=400, seed=12, n_motifs=6, motif_min=4, motif_max=10, p_motif=0.7)
    rows = []
    for r in range(16):
        p, g = src.request(r, 64, 40)
        rows.append({"prompt": [int(x) for x in p], "response": [int(x) for x in g]})
    data = pd.DataFrame(rows)
    cfg = dict(task_id=0, num_eval=6, num_train=9, seed=3, max_depth=32, max_spec_tokens=0, max_spec_factor=1.5,
               min_token_prob=0.1, use_tree_spec=False, use_cached_prompt=True)
    a = pd.DataFrame(S.run_task(SuffixCache, data, None, **cfg)).drop(columns=["spec_ms", "update_ms"])
    b = pd.DataFrame(S.run_task(OracleSuffixCache, data, None, **cfg)).drop(columns=["spec_ms", "update_ms"])
    assert a["num_accept_toks"].sum() > 0
    pd.testing.assert_frame_equal(a, b)


def test_zero_copy_round_trip_gives_the_golden_digests():
    """The alternative round trip (AIC_SUFFIX_ZEROCOPY=1: the apply kernel reads the deltas from pinned host memory, the
    select kernel writes the winners there and raises a flag the host polls; measured, not the default) against the same
    reference-generated digests.  The switch is read once per process: run in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import golden_utils as gu, test_gpu_suffix as T\n"
            "from arcticinference_amd.suffix_cache import SuffixCache\n"
            "want = gu.load('suffix_replay.json')[0]\n"
            "got = T._replay(SuffixCache(64), want['config'], batched=False)\n"
            "assert got == want['per_request'], 'digests differ'\n"
            "got = T._replay(SuffixCache(64), want['config'], batched=True)\n"
            "print('ok')\n") % (os.path.join(root, "tests"), root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, AIC_SUFFIX_ZEROCOPY="1"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


# ------------------------------------------------------------------------------------------------------------------
# tree-mode speculation on the device (SuffixTree::_speculate_tree, suffix_tree.cc:226-274; r04)
# ------------------------------------------------------------------------------------------------------------------
def _tree_stats():
    import ctypes
    from arcticinference_amd import _native as N
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    N.check(N.lib().aic_debug_tree_mode_stats(ctypes.byref(a), ctypes.byref(b)))
    return int(a.value), int(b.value)


def test_tree_mode_goldens_are_answered_by_the_device():
    """The reference-generated tree-mode vectors (suffix_treespec.json: use_tree_spec True on branching trees, min_token_prob
    down to 0, budgets up to 32 tokens) replayed through the DEVICE path — kid lists mirrored in the host container's order,
    one priority queue per wave reproducing std::priority_queue — bit for bit (tokens, PARENTS, f32 probabilities and
    scores), with not one query handed back to the host trees."""
    dev0, back0 = _tree_stats()
    n_tree = 0
    for case in gu.load("suffix_treespec.json"):
        n_tree += sum(1 for ev in case["events"] if ev[0] == "spec" and ev[6])
        gu.replay_tree_case(case, SuffixTree)
    dev1, back1 = _tree_stats()
    assert n_tree > 500 and dev1 - dev0 == n_tree and back1 == back0, (n_tree, dev1 - dev0, back1 - back0)


@pytest.mark.parametrize("seed,vocab,depth", [(0, 5, 12), (1, 12, 16), (2, 40, 24), (3, 3, 64), (4, 200, 32)])
def test_tree_mode_device_equals_host_trees(seed, vocab, depth):
    """Fuzz: growing trees (several sequences, interleaved appends: every update case incl. re-keying and splits; vocabularies
    from 3 to 200, so nodes range from 2 to MORE than 15 children — the latter are handed back to the host and must still
    give the host's answer), tree-mode queries after every few appends, device against the host trees' own expansion
    (aic_debug_tree_mode_on_host: suffix_host.hpp grow_tree, itself pinned by the goldens on CPU)."""
    from arcticinference_amd import _native as N
    rng = random.Random(100 + seed)
    t = SuffixTree(depth)
    hist = {s: [] for s in range(5)}
    dev0, back0 = _tree_stats()
    n_q = 0
    for step in range(260):
        s = rng.randrange(5)
        # skewed token choice so that counts differ between siblings, with repeats so that deep matches exist
        tok = min(int(rng.expovariate(0.6)), vocab - 1) if rng.random() < 0.8 else rng.randrange(vocab)
        if hist[s] and rng.random() < 0.3:
            k = rng.randrange(len(hist[s]))
            tok = hist[s][k]
        hist[s].append(tok)
        t.append(s, tok)
        if step % 3 != 2:
            continue
        src = hist[rng.randrange(5)]
        if not src:
            continue
        e = rng.randint(1, len(src))
        pat = src[max(0, e - rng.randint(1, depth)):e]
        args = (rng.choice([1, 4, 8, 16, 32]), rng.choice([1.0, 2.0, 4.0]), rng.choice([0.0, 2.0]),
                rng.choice([0.0, 0.02, 0.1, 0.25]), True)
        got = t.speculate(pat, *args)
        N.lib().aic_debug_tree_mode_on_host(1)
        try:
            want = t.speculate(pat, *args)
        finally:
            N.lib().aic_debug_tree_mode_on_host(0)
        gu.assert_cand(got, gu.cand_dict(want), ctx=f"seed {seed} step {step}: pattern={pat} args={args}")
        n_q += 1
    dev1, back1 = _tree_stats()
    assert (dev1 - dev0) + (back1 - back0) == n_q and dev1 > dev0
    if vocab <= 12:
        assert back1 == back0           # every node has at most 15 children: nothing may be handed back
    if vocab >= 200:
        assert back1 > back0            # the root-level nodes overflow the 15-entry lists: some queries were handed back
    assert t.selfcheck() == 0


def test_tree_mode_through_the_suffix_cache_on_the_device():
    """SuffixCache.speculate(use_tree_spec=True): prompt tree and global tree in ONE device round trip, the prompt tree winning
    ties, against the oracle's SuffixCache (the reference's policy over oracle trees) — and the simulator's call form."""
    from oracle.suffix_oracle import OracleSuffixCache
    rng = random.Random(11)
    a, b = SuffixCache(12), OracleSuffixCache(12)
    prompts = {r: [rng.randrange(6) for _ in range(40)] for r in range(3)}
    for c in (a, b):
        for r, p in prompts.items():
            c.cache_prompt(r, p)
    dev0, back0 = _tree_stats()
    hist = {r: [] for r in prompts}
    n = 0
    for step in range(150):
        r = rng.randrange(3)
        tok = rng.randrange(6)
        hist[r].append(tok)
        for c in (a, b):
            c.update_response(r, tok)
        if step % 4 == 0:
            pat = (prompts[r] + hist[r])[-rng.randint(1, 12):]
            kw = dict(max_spec_tokens=rng.choice([4, 8, 16]), max_spec_factor=rng.choice([1.0, 2.0]),
                      max_spec_offset=0.0, min_token_prob=rng.choice([0.0, 0.1]), use_tree_spec=True,
                      use_cached_prompt=rng.random() < 0.7)
            gu.assert_cand(a.speculate(r, pat, **kw), gu.cand_dict(b.speculate(r, pat, **kw)), ctx=f"step {step} {kw}")
            n += 1
    dev1, back1 = _tree_stats()
    assert dev1 - dev0 == n and back1 == back0
