"""Replays the committed golden vectors (tests/golden/*.json, produced from the real reference by
oracle/gen_golden.py) against any implementation with the reference's SuffixTree / SuffixCache
surface.  Floats are compared bit-for-bit (float32 little-endian hex)."""
from __future__ import annotations

import json
import os
import struct

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name: str):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def f2h(x: float) -> str:
    return struct.pack("<f", x).hex()


def h2f(h: str) -> float:
    return struct.unpack("<f", bytes.fromhex(h))[0]


def cand_dict(c) -> dict:
    return {"t": [int(x) for x in c.token_ids], "p": [int(x) for x in c.parents],
            "pr": [f2h(x) for x in c.probs], "s": f2h(c.score), "m": int(c.match_len)}


def assert_cand(got, want: dict, ctx: str = "", check_parents: bool = True) -> None:
    g = cand_dict(got)
    if not check_parents:
        g = dict(g, p=want["p"])
    assert g == want, f"{ctx}\n got  {g}\n want {want}"


def replay_tree_case(case: dict, make_tree, speculate=None, only_path: bool = False) -> int:
    """`make_tree(max_depth)` -> object with extend(seq, toks) and
    speculate(pattern, max_spec_tokens, factor, offset, min_prob, use_tree)."""
    t = make_tree(case["max_depth"])
    n = 0
    for ei, ev in enumerate(case["events"]):
        if ev[0] == "ext":
            t.extend(ev[1], ev[2])
        elif ev[0] == "spec":
            _, pat, mst, factor, offset, mp, tree, want = ev
            if only_path and tree:
                continue
            got = (speculate or (lambda tt, *a: tt.speculate(*a)))(t, pat, mst, factor, offset, mp, tree)
            assert_cand(got, want, ctx=f"event {ei}: pattern={pat} args={(mst, factor, offset, mp, tree)}")
            n += 1
    return n


def replay_cache_case(case: dict, make_cache) -> int:
    c = make_cache(case["max_depth"])
    n = 0
    for ei, ev in enumerate(case["events"]):
        kind = ev[0]
        if kind == "cache_prompt":
            c.cache_prompt(ev[1], ev[2])
        elif kind == "update":
            c.update_response(ev[1], ev[2])
        elif kind == "evict":
            c.evict_prompt(ev[1])
        elif kind == "speculate":
            _, rid, pattern, kw, want = ev
            got = c.speculate(rid, pattern, **kw)
            assert_cand(got, want, ctx=f"event {ei}: req={rid} kw={kw}")
            n += 1
    return n
