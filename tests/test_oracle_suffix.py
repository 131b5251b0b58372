"""CPU: pins the oracle (oracle/suffix_tree_oracle.cpp) against the golden vectors produced by the real
reference build (oracle/gen_golden.py), and — when the reference is present (build container only) —
cross-checks it live against oracle/_ref."""
import random

import pytest

import golden_utils as gu
from oracle import ref_loader
from oracle.suffix_oracle import OracleSuffixCache, OracleSuffixTree
from arcticinference_amd.workload import TokenSource


@pytest.mark.parametrize("name", ["suffix_traces.json", "suffix_ties.json", "suffix_clamps.json",
                                  "suffix_treespec.json"])
def test_oracle_tree_matches_golden(name):
    total = 0
    for case in gu.load(name):
        total += gu.replay_tree_case(case, OracleSuffixTree)
    assert total > 100


def test_oracle_cache_matches_golden():
    total = sum(gu.replay_cache_case(c, OracleSuffixCache) for c in gu.load("suffix_cache.json"))
    assert total > 100


def test_oracle_cache_errors():
    c = OracleSuffixCache(8)
    c.cache_prompt("a", [1, 2, 3])
    with pytest.raises(ValueError):
        c.cache_prompt("a", [1])
    with pytest.raises(ValueError):
        c.evict_prompt("b")
    with pytest.raises(ValueError):
        c.speculate("b", [1])
    with pytest.raises(ValueError):
        c.speculate("a", [])
    assert c.speculate("b", [1, 2], use_cached_prompt=False).score == 0.0


def _replay_digest(cache, cfg):
    import hashlib
    import struct
    src = TokenSource(seed=cfg["seed"])
    out = []
    for r in range(cfg["n_req"]):
        prompt, gt = src.request(r, cfg["prompt_len"], cfg["gen_len"])
        prompt, gt = [int(x) for x in prompt], [int(x) for x in gt]
        cache.cache_prompt(r, prompt)
        h = hashlib.sha256()
        resp, steps, acc, spec = [], 0, 0, 0
        while len(resp) < len(gt):
            text = (prompt + resp)[-cache.max_depth:]
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


@pytest.mark.parametrize("idx", [0, 1])
def test_oracle_replay_digests(idx):
    """(6) of SURVEY §8c: the seeded replay, incl. the full 64 x (4096 + 256) BASELINE-size one."""
    want = gu.load("suffix_replay.json")[idx]
    got = _replay_digest(OracleSuffixCache(64), want["config"])
    assert got == want["per_request"]
    assert sum(g["accepted"] for g in got) == want["sum_accept"]


@pytest.mark.skipif(not ref_loader.available(), reason="reference build only exists in the build container")
def test_oracle_vs_live_reference_fuzz():
    RefTree, _, _, _ = ref_loader.load()
    n = 0
    for trial in range(60):
        rng = random.Random(trial)
        depth = rng.choice([2, 3, 4, 8, 16, 64])
        vocab = rng.choice([2, 3, 5, 20, 200])
        rt, ot = RefTree(depth), OracleSuffixTree(depth)
        hist = {s: [] for s in range(3)}
        for step in range(rng.randint(20, 300)):
            s, t = rng.randrange(3), rng.randrange(vocab)
            rt.append(s, t)
            ot.append(s, t)
            hist[s].append(t)
            if step % 3 == 0:
                src = hist[s]
                pat = src[-rng.randint(1, min(len(src), depth + 3)):]
                args = (rng.choice([0, 1, 3, 8, 32]), rng.choice([0.5, 1.0, 2.0]), rng.choice([-1.0, 0.0, 1.0]),
                        rng.choice([0.0, 0.1, 0.5]), rng.random() < 0.4)
                gu.assert_cand(ot.speculate(pat, *args), gu.cand_dict(rt.speculate(pat, *args)))
                n += 1
    assert n > 1000
