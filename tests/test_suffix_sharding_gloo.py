"""CPU, world sizes 2 and 8 over gloo: rank-owned prompt trees (arcticinference_amd/suffix_sharding.py).  Every rank runs
the sharded control plane — prompt trees only for the requests it owns, every request's tokens into its global-tree
replica, speculation for its own requests, one int32 all-reduce — next to a REPLICATED cache (the reference's arrangement:
every prompt tree on every rank, model_runner.py:657-744), over the oracle's suffix trees (the product's device matcher
needs a GPU; the sharding logic is the same code either way).  Step by step, on every rank, the sharded drafts must equal
the replicated ones bit for bit (tokens, counts, score bits), requests being replaced along the way; and a rank must hold
no prompt tree it does not own."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from arcticinference_amd.suffix_sharding import RankOwnedSuffix, gloo_exchange
        from arcticinference_amd.workload import TokenSource
        from oracle.suffix_oracle import OracleSuffixCache

        depth, B, PL, GL, steps = 16, 16, 120, 40, 14
        src = TokenSource(vocab_size=300, seed=3, n_motifs=6, motif_min=5, motif_max=10)

        def rows_oracle(cache):
            def speculate_rows(req_ids, flat, lens, mst, fac, off, mpr):
                n = len(req_ids)
                toks, n_tok, score = np.zeros((n, 32), np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
                at = 0
                for i, rid in enumerate(req_ids):
                    pat = [int(t) for t in flat[at:at + lens[i]]]
                    at += int(lens[i])
                    r = cache.speculate(rid, pat, max_spec_tokens=int(mst[i]), max_spec_factor=float(fac[i]),
                                        max_spec_offset=float(off[i]), min_token_prob=float(mpr[i]))
                    k = len(r.token_ids)
                    toks[i, :k], n_tok[i], score[i] = r.token_ids, k, r.score
                return toks, n_tok, score
            return speculate_rows

        sharded_cache, full_cache = OracleSuffixCache(depth), OracleSuffixCache(depth)
        sh = RankOwnedSuffix(sharded_cache, rank, world, gloo_exchange(dist.group.WORLD), speculate_rows=rows_oracle(sharded_cache))
        full = RankOwnedSuffix(full_cache, 0, 1, lambda m: m, speculate_rows=rows_oracle(full_cache))   # world 1 = replicated
        streams, req_of, n_gen = {}, [None] * B, np.zeros(B, np.int64)
        next_id = [0]

        def admit(slot):
            rid = next_id[0]
            next_id[0] += 1
            s = src.stream(PL + GL + 64, rid)
            streams[rid] = s
            old = req_of[slot]
            first = int(s[PL])
            for x in (sh, full):
                x.admit(slot, rid, [int(t) for t in s[:PL]], [first], old_req_id=old)
            req_of[slot], n_gen[slot] = rid, 1

        # first fill in one call (the engine's add_requests form), later replacements one by one
        ids = list(range(B))
        next_id[0] = B
        for rid in ids:
            streams[rid] = src.stream(PL + GL + 64, rid)
        for x in (sh, full):
            x.admit_many(list(range(B)), ids, [streams[r][:PL] for r in ids], [[int(streams[r][PL])] for r in ids])
        req_of, n_gen = list(ids), np.ones(B, np.int64)
        ok, compared, nonempty = True, 0, 0
        rng = np.random.default_rng(7)                   # same seed on every rank: the control flow is replicated
        for step in range(steps):
            slots = list(range(B))
            rids = [req_of[s] for s in slots]
            # the step's accepted tokens: 1-4 of the request's ground-truth stream
            n_emit = rng.integers(1, 5, size=B)
            flat = np.concatenate([streams[r][PL + n_gen[s]:PL + n_gen[s] + n_emit[s]] for s, r in zip(slots, rids)]).astype(np.int32)
            for x in (sh, full):
                x.update(rids, flat, n_emit.astype(np.int32))
            n_gen += n_emit
            pats = [streams[r][:PL + n_gen[s]][-depth:] for s, r in zip(slots, rids)]
            lens = np.asarray([len(p) for p in pats], np.int32)
            args = (np.concatenate(pats).astype(np.int32), lens, np.full(B, 8, np.int32), np.ones(B, np.float32),
                    np.zeros(B, np.float32), np.full(B, 0.1, np.float32))
            a = sh.propose(slots, rids, *args)
            b = full.propose(slots, rids, *args)
            ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.int32), b[2].view(np.int32))
            compared += B
            nonempty += int((a[1] > 0).sum())
            for s in slots:                               # requests that are done leave; a new one takes the slot
                if n_gen[s] >= GL:
                    admit(s)
        # ownership: prompt trees only for the owned slots' CURRENT requests, none evicted twice, none left behind
        held = set(sharded_cache.cached_prompt_ids())
        want = {req_of[s] for s in range(B) if s % world == rank}
        ok = ok and held == want and set(full_cache.cached_prompt_ids()) == set(req_of)
        st = sh.stats
        ok = ok and st["queries_owned"] * world == st["queries_total"] and st["prompt_trees_built"] < full.stats["prompt_trees_built"]
        out_q.put((rank, bool(ok), compared, nonempty, st["prompt_trees_built"], full.stats["prompt_trees_built"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_rank_owned_prompt_trees_give_the_replicated_drafts(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert all(r[2] == 14 * 16 for r in res) and all(r[3] > 20 for r in res), res      # drafts were really produced and compared
    # the prompt trees of the run were built once across the ranks, not once per rank
    assert sum(r[4] for r in res) == res[0][5], res
