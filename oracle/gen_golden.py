"""ORACLE — fixture generator.  Build-container use only.

Drives the REAL reference (oracle/_ref, built from /root/reference/csrc/suffix_cache by
oracle/Makefile, plus the reference's own suffix_cache.py imported from where it lies) and writes
golden input/output vectors to tests/golden/.  The fixtures are data only (events in, candidates
out; float32 values as little-endian hex so they are compared bit-for-bit).

    python -m oracle.gen_golden            # regenerates every tests/golden/suffix_*.json

Vector families (SURVEY.md §8c list):
  (1) suffix_traces     append traces covering the four update cases (leaf extend, new leaf,
                        fuse / extend-without-overlap with re-key, count++ / split), speculate after
                        every token
  (2) suffix_ties       equal-count children, 2..40 children (crosses libstdc++ rehashes), large keys,
                        re-keyed children
  (3) suffix_clamps     factor x offset x max_spec_tokens x min_token_prob grid
  (4) suffix_treespec   use_tree_spec True/False on branching trees
  (5) suffix_cache      SuffixCache (prompt trees + global tree) driven with the model-runner call
                        pattern of model_runner.py:680-744 (pattern slicing, offset rewrite, eviction)
  (6) suffix_replay     seeded 64-request x (4096 prompt + 256 gen) replay a la
                        simulator.suffix_decode: per-request digests + mean accepted length
"""
from __future__ import annotations

import hashlib
import json
import os
import random
import struct
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
sys.path.insert(0, _ROOT)

from oracle import ref_loader  # noqa: E402
from arcticinference_amd.workload import TokenSource  # noqa: E402

OUT = os.path.join(_ROOT, "tests", "golden")


def f2h(x: float) -> str:
    return struct.pack("<f", x).hex()


def cand(c) -> dict:
    return {"t": [int(x) for x in c.token_ids], "p": [int(x) for x in c.parents],
            "pr": [f2h(x) for x in c.probs], "s": f2h(c.score), "m": int(c.match_len)}


def dump(name: str, obj) -> None:
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


ARGSETS = [(8, 1.0, 0.0, 0.1, False), (32, 2.0, 0.0, 0.0, False), (3, 1.0, 1.0, 0.3, False),
           (8, 1.0, 0.0, 0.1, True), (16, 1.5, -1.0, 0.0, True)]


def gen_traces(Tree):
    cases = []
    hand = [
        # leaf extension then branching
        (4, [(0, [1, 2, 3, 4, 5, 6])]),
        # repeated char: active nodes that are ancestors of each other; fuse paths
        (8, [(0, [7, 7, 7, 7, 7, 7, 7, 7, 7, 7])]),
        (8, [(0, [1, 2, 1, 2, 1, 2, 1, 3, 1, 2, 1, 3])]),
        # two sequences sharing prefixes: count++ and split
        (6, [(0, [1, 2, 3, 4]), (1, [1, 2, 3, 5]), (0, [1, 2, 9]), (1, [2, 3, 4, 4])]),
        # extend-without-overlap with re-key (child first token changes)
        (5, [(0, [1, 2, 3]), (1, [1, 2, 3, 4]), (0, [4, 1, 2]), (1, [3, 4, 1])]),
        (3, [(0, [1, 2, 3, 4])]),
        (64, [(0, [5, 6, 7, 8, 5, 6, 7, 9, 5, 6, 7, 8, 5, 6])]),
    ]
    for depth, script in hand:
        ev = []
        t = Tree(depth)
        hist = {}
        for seq, toks in script:
            for tok in toks:
                t.append(seq, tok)
                hist.setdefault(seq, []).append(tok)
                ev.append(["ext", seq, [tok]])
                h = hist[seq]
                pat = h[-(depth + 2):]
                for a in ARGSETS[:2] + ARGSETS[3:4]:
                    ev.append(["spec", pat, *a, cand(t.speculate(pat, *a))])
        cases.append({"max_depth": depth, "events": ev})
    for trial in range(24):
        rng = random.Random(1000 + trial)
        depth = rng.choice([2, 3, 4, 6, 8, 16])
        vocab = rng.choice([2, 3, 4, 6])
        nseq = rng.randint(1, 3)
        t = Tree(depth)
        hist = {s: [] for s in range(nseq)}
        ev = []
        for step in range(rng.randint(30, 70)):
            s = rng.randrange(nseq)
            tok = rng.randrange(vocab)
            t.append(s, tok)
            hist[s].append(tok)
            ev.append(["ext", s, [tok]])
            src = hist[rng.randrange(nseq)]
            if src:
                L = rng.randint(1, min(len(src), depth + 2))
                pat = src[-L:]
                a = ARGSETS[rng.randrange(len(ARGSETS))]
                ev.append(["spec", pat, *a, cand(t.speculate(pat, *a))])
        cases.append({"max_depth": depth, "events": ev})
    dump("suffix_traces.json", cases)


def gen_ties(Tree):
    cases = []
    for nchild in [2, 3, 5, 9, 12, 13, 14, 29, 30, 40]:
        for keymode in ["small", "large", "mixed"]:
            rng = random.Random(nchild * 7 + len(keymode))
            if keymode == "small":
                keys = rng.sample(range(2, 200), nchild)
            elif keymode == "large":
                keys = rng.sample(range(100000, 128256), nchild)
            else:
                keys = rng.sample(range(2, 128256), nchild)
            depth = 8
            t = Tree(depth)
            ev = []
            # every sequence is "1 <key> <key+1>" so node "1" gets nchild equal-count children
            for i, k in enumerate(keys):
                toks = [1, k, (k + 1) % 128256]
                t.extend(i, toks)
                ev.append(["ext", i, toks])
                for a in [(8, 2.0, 1.0, 0.0, False), (8, 2.0, 1.0, 0.0, True)]:
                    ev.append(["spec", [1], *a, cand(t.speculate([1], *a))])
            # bump some children to make partial ties at a higher count
            for j in range(0, nchild, 3):
                toks = [1, keys[j]]
                t.extend(1000 + j, toks)
                ev.append(["ext", 1000 + j, toks])
                ev.append(["spec", [1], 8, 2.0, 1.0, 0.0, False, cand(t.speculate([1], 8, 2.0, 1.0, 0.0, False))])
            cases.append({"max_depth": depth, "events": ev})
    # re-keyed children: extend-without-overlap changes the key under which a child is stored
    for trial in range(12):
        rng = random.Random(500 + trial)
        depth = rng.choice([4, 6, 8])
        t = Tree(depth)
        ev = []
        hist = {}
        for step in range(120):
            s = rng.randrange(3)
            tok = rng.choice([1, 2, 3]) if rng.random() < 0.8 else rng.randrange(4, 40)
            t.append(s, tok)
            hist.setdefault(s, []).append(tok)
            ev.append(["ext", s, [tok]])
            if step % 2 == 0:
                pat = hist[s][-rng.randint(1, min(len(hist[s]), depth)):]
                a = (16, 3.0, 2.0, 0.0, rng.random() < 0.5)
                ev.append(["spec", pat, *a, cand(t.speculate(pat, *a))])
        cases.append({"max_depth": depth, "events": ev})
    dump("suffix_ties.json", cases)


def gen_clamps(Tree):
    rng = random.Random(7)
    depth = 16
    t = Tree(depth)
    ev = []
    base = [rng.randrange(5) for _ in range(200)]
    t.extend(0, base)
    ev.append(["ext", 0, base])
    other = [rng.randrange(5) for _ in range(150)]
    t.extend(1, other)
    ev.append(["ext", 1, other])
    pats = [base[-k:] for k in (1, 2, 3, 5, 8, 16, 20)] + [other[40:40 + k] for k in (2, 4, 9)]
    for pat in pats:
        for factor in (0.5, 1.0, 2.0):
            for offset in (-1.0, 0.0, 1.0):
                for mst in (0, 1, 3, 8, 32):
                    for mp in (0.0, 0.1, 0.5):
                        a = (mst, factor, offset, mp, False)
                        ev.append(["spec", pat, *a, cand(t.speculate(pat, *a))])
    dump("suffix_clamps.json", [{"max_depth": depth, "events": ev}])


def gen_treespec(Tree):
    cases = []
    for trial in range(10):
        rng = random.Random(300 + trial)
        depth = rng.choice([6, 8, 12, 64])
        vocab = rng.choice([3, 4, 8])
        t = Tree(depth)
        ev = []
        hist = {}
        for s in range(4):
            toks = [rng.randrange(vocab) for _ in range(rng.randint(40, 90))]
            t.extend(s, toks)
            hist[s] = toks
            ev.append(["ext", s, toks])
        for q in range(60):
            src = hist[rng.randrange(4)]
            e = rng.randint(1, len(src))
            L = rng.randint(1, min(e, depth))
            pat = src[e - L:e]
            for tree in (False, True):
                a = (rng.choice([4, 8, 16, 32]), rng.choice([1.0, 2.0, 4.0]), rng.choice([0.0, 2.0]),
                     rng.choice([0.0, 0.05, 0.1, 0.25]), tree)
                ev.append(["spec", pat, *a, cand(t.speculate(pat, *a))])
        cases.append({"max_depth": depth, "events": ev})
    dump("suffix_treespec.json", cases)


def runner_call(cache, req_id, row, end_idx, spec_ids, cfg, max_model_len, MAX_SPEC_LEN=32):
    """The call pattern of propose_suffix_draft_token_ids (model_runner.py:709-740)."""
    depth = cfg["suffix_cache_max_depth"]
    size = min(end_idx, depth)
    pattern = list(row[end_idx - size:end_idx]) + list(spec_ids)
    if len(pattern) > depth:
        pattern = pattern[-depth:]
    max_spec_tokens = min(MAX_SPEC_LEN - len(spec_ids), depth, max_model_len - end_idx - 1)
    factor = cfg["suffix_max_spec_factor"]
    offset = cfg["suffix_max_spec_offset"] - len(spec_ids) * (factor + 1)
    kw = dict(max_spec_tokens=max_spec_tokens, max_spec_factor=factor, max_spec_offset=offset,
              min_token_prob=cfg["suffix_min_token_prob"])
    return pattern, kw, cache.speculate(req_id, pattern, **kw)


def gen_cache(Cache):
    cases = []
    src = TokenSource(vocab_size=4096, seed=11, n_motifs=12, motif_min=6, motif_max=20)
    for ci, cfg in enumerate([
        dict(suffix_cache_max_depth=64, suffix_max_spec_factor=1.0, suffix_max_spec_offset=0.0, suffix_min_token_prob=0.1),
        dict(suffix_cache_max_depth=16, suffix_max_spec_factor=2.0, suffix_max_spec_offset=-1.0, suffix_min_token_prob=0.05),
        dict(suffix_cache_max_depth=8, suffix_max_spec_factor=1.5, suffix_max_spec_offset=1.0, suffix_min_token_prob=0.3),
    ]):
        rng = random.Random(40 + ci)
        cache = Cache(cfg["suffix_cache_max_depth"])
        ev = []
        max_model_len = 400
        nreq = 5
        rows, ends, gts = {}, {}, {}
        for r in range(nreq):
            p, g = src.request(100 * ci + r, rng.randint(60, 200), 120)
            rid = f"req-{r}"
            rows[rid] = [int(x) for x in p]
            ends[rid] = len(p)
            gts[rid] = [int(x) for x in g]
        live = []
        order = list(rows.keys())
        for step in range(70):
            # admit a new request now and then; drop finished ones
            if order and (not live or rng.random() < 0.2):
                live.append(order.pop(0))
            for rid in list(live):
                k = len(rows[rid]) - ends[rid]
                if k >= len(gts[rid]) - 6:
                    live.remove(rid)
            seen = list(live)
            for rid in seen:
                k = len(rows[rid]) - ends[rid]
                n_new = rng.randint(1, 4)
                new = gts[rid][k:k + n_new]
                if not cache.has_cached_prompt(rid):
                    cache.cache_prompt(rid, rows[rid][:ends[rid]])
                    ev.append(["cache_prompt", rid, rows[rid][:ends[rid]]])
                cache.update_response(rid, new)
                ev.append(["update", rid, new])
                rows[rid].extend(new)
            for rid in cache.cached_prompt_ids():
                if rid not in seen:
                    cache.evict_prompt(rid)
                    ev.append(["evict", rid])
            for rid in seen:
                end_idx = len(rows[rid])
                spec_ids = [] if rng.random() < 0.7 else gts[rid][end_idx - ends[rid]:end_idx - ends[rid] + rng.randint(1, 3)]
                pattern, kw, res = runner_call(cache, rid, rows[rid], end_idx, spec_ids, cfg, max_model_len)
                ev.append(["speculate", rid, pattern, kw, cand(res)])
        # also: use_cached_prompt=False and the error surface is exercised by the test itself
        cases.append({"max_depth": cfg["suffix_cache_max_depth"], "cfg": cfg, "events": ev})
    dump("suffix_cache.json", cases)


def replay(cache, src, n_req, prompt_len, gen_len, max_spec_tokens, factor, offset, min_prob):
    """simulator.suffix_decode restated as a driver (simulator.py:33-114); path mode."""
    per_req = []
    tot_accept = tot_spec = tot_steps = tot_out = 0
    for r in range(n_req):
        prompt, gt = src.request(r, prompt_len, gen_len)
        prompt = [int(x) for x in prompt]
        gt = [int(x) for x in gt]
        cache.cache_prompt(r, prompt)
        h = hashlib.sha256()
        resp = []
        steps = acc = spec = 0
        while len(resp) < len(gt):
            text = (prompt + resp)[-cache.max_depth:]
            res = cache.speculate(r, text, max_spec_tokens=max_spec_tokens, max_spec_factor=factor,
                                  max_spec_offset=offset, min_token_prob=min_prob)
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
            tot_out += len(new)
        cache.evict_prompt(r)
        per_req.append({"steps": steps, "accepted": acc, "speculated": spec, "sha256": h.hexdigest()})
        tot_accept += acc
        tot_spec += spec
        tot_steps += steps
    return {"per_request": per_req, "sum_accept": tot_accept, "sum_spec": tot_spec, "steps": tot_steps,
            "sum_out": tot_out, "avg_accept_toks": tot_accept / tot_steps}


def gen_replay(Cache):
    out = []
    for (n_req, pl, gl, seed) in [(8, 512, 64, 3), (64, 4096, 256, 0)]:
        src = TokenSource(seed=seed)
        cache = Cache(64)
        r = replay(cache, src, n_req, pl, gl, 32, 1.0, 0.0, 0.1)
        r["config"] = {"n_req": n_req, "prompt_len": pl, "gen_len": gl, "seed": seed, "max_depth": 64,
                       "max_spec_tokens": 32, "factor": 1.0, "offset": 0.0, "min_token_prob": 0.1}
        print(f"replay {n_req}x({pl}+{gl}): avg_accept_toks={r['avg_accept_toks']:.4f} steps={r['steps']}")
        out.append(r)
    dump("suffix_replay.json", out)


def main():
    if not ref_loader.available():
        raise SystemExit("the reference build (oracle/_ref) is not available here")
    Tree, _, Cache, _ = ref_loader.load()
    gen_traces(Tree)
    gen_ties(Tree)
    gen_clamps(Tree)
    gen_treespec(Tree)
    gen_cache(Cache)
    gen_replay(Cache)


if __name__ == "__main__":
    main()
