"""CPU: the product's host suffix tree (arcticinference_amd/csrc/suffix_host.hpp) and the HBM image
it mirrors, checked against the golden vectors from the real reference.  No compute call is made on
the library (there is no GPU here): the image is exported and walked by tests/flat_matcher.py."""
import random

import pytest

import flat_matcher
import golden_utils as gu
from arcticinference_amd.suffix_cache import SuffixCache, SuffixTree


class _ImgTree:
    def __init__(self, depth):
        self.t = SuffixTree(depth)
        self.depth = depth

    def extend(self, seq, toks):
        self.t.extend(seq, toks)

    def speculate(self, pat, mst, factor, offset, mp, tree):
        if tree:  # tree mode is a host feature of the library
            return self.t.speculate(pat, mst, factor, offset, mp, True)
        assert self.t.selfcheck() == 0
        return flat_matcher.speculate(self.t.export(), list(pat), mst, factor, offset, mp, self.depth)


@pytest.mark.parametrize("name", ["suffix_traces.json", "suffix_ties.json", "suffix_treespec.json"])
def test_host_image_matches_golden(name):
    total = 0
    for case in gu.load(name):
        total += gu.replay_tree_case(case, _ImgTree)
    assert total > 100


def test_host_image_clamps_subset():
    case = gu.load("suffix_clamps.json")[0]
    case = dict(case, events=case["events"][:2] + case["events"][2::7])
    assert gu.replay_tree_case(case, _ImgTree) > 100


def test_best_child_bookkeeping_fuzz():
    """NodeRec.best must always equal the reference's container-order scan (selfcheck)."""
    for trial in range(40):
        rng = random.Random(trial)
        t = SuffixTree(rng.choice([3, 5, 8, 64]))
        vocab = rng.choice([2, 3, 6, 30, 3000])
        for step in range(rng.randint(200, 1500)):
            t.append(rng.randrange(4), rng.randrange(vocab))
            if step % 97 == 0:
                assert t.selfcheck() == 0
        assert t.selfcheck() == 0


def test_cache_surface_and_errors():
    c = SuffixCache(8)
    assert c.max_depth == 8
    c.cache_prompt("a", [1, 2, 3])
    assert c.has_cached_prompt("a") and c.cached_prompt_ids() == ["a"]
    with pytest.raises(ValueError):
        c.cache_prompt("a", [1])
    with pytest.raises(ValueError):
        c.evict_prompt("b")
    with pytest.raises(ValueError):
        c.speculate("b", [1])
    with pytest.raises(ValueError):
        c.speculate("a", [])
    c.update_response("a", 4)
    c.update_response("b", [1, 2, 3])
    assert c._global_tree().num_seqs() == 2
    assert c._prompt_tree("a").export()["tokens"][:4].tolist() == [1, 2, 3, 4]
    c.evict_prompt("a")
    assert not c.has_cached_prompt("a")
    c.cache_prompts(["x", "y"], [[1, 2, 3, 1, 2], [5, 5, 5]], n_threads=2)
    assert c.cached_prompt_ids() == ["x", "y"]
    assert c._prompt_tree("y").selfcheck() == 0


def test_update_responses_equals_per_request_loop():
    """The batched update (one native call) leaves every tree byte-identical to the per-request update_response loop
    of the reference's _update_suffix_cache."""
    import numpy as np
    rng = random.Random(11)
    a, b = SuffixCache(16), SuffixCache(16)
    n = 24
    prompts = [[rng.randrange(1, 9) for _ in range(rng.randint(5, 60))] for _ in range(n)]
    for c in (a, b):
        c.cache_prompts(list(range(n - 4)), prompts[:n - 4], n_threads=2)     # the last four have no prompt tree
    for step in range(40):
        runs = [[rng.randrange(1, 9) for _ in range(rng.choice([0, 1, 1, 2, 4, 9]))] for _ in range(n)]
        for r in range(n):
            if runs[r]:
                a.update_response(r, runs[r])
        live = [r for r in range(n) if runs[r] or step % 2]                    # zero-length runs are allowed
        b.update_responses(live, np.asarray([t for r in live for t in runs[r]], dtype=np.int32),
                           np.asarray([len(runs[r]) for r in live], dtype=np.int32))
    def same(x, y):
        ex, ey = x.export(), y.export()
        assert ex.keys() == ey.keys()
        for k in ex:
            assert np.array_equal(np.asarray(ex[k]), np.asarray(ey[k])), k
    same(a._global_tree(), b._global_tree())
    for r in range(n - 4):
        same(a._prompt_tree(r), b._prompt_tree(r))
    assert b._global_tree().selfcheck() == 0
    with pytest.raises(ValueError):
        b.update_responses([0, 1], np.zeros(3, np.int32), np.asarray([1, 1], np.int32))


def test_cache_tree_mode_matches_oracle():
    from oracle.suffix_oracle import OracleSuffixCache
    rng = random.Random(5)
    a, b = SuffixCache(8), OracleSuffixCache(8)
    for c in (a, b):
        c.cache_prompt(0, [1, 2, 3, 1, 2, 4, 1, 2, 3])
    hist = []
    for step in range(200):
        tok = rng.randrange(1, 5)
        hist.append(tok)
        for c in (a, b):
            c.update_response(0, tok)
        if step % 5 == 0:
            pat = ([1, 2, 3, 1, 2, 4, 1, 2, 3] + hist)[-rng.randint(1, 10):]
            ra = a.speculate(0, pat, 8, 2.0, 0.0, 0.05, use_tree_spec=True)
            rb = b.speculate(0, pat, 8, 2.0, 0.0, 0.05, use_tree_spec=True)
            gu.assert_cand(ra, gu.cand_dict(rb))


def test_async_prompt_build_equals_synchronous_calls():
    """cache_prompt_async(prompt, response) must leave exactly the trees that cache_prompt + update_response leave:
    same global-tree image (the response enters it in call order), same prompt-tree image, also when the request is
    extended, queried (export joins the build) or evicted while other builds are in flight."""
    import numpy as np
    from arcticinference_amd.workload import TokenSource
    src = TokenSource(vocab_size=500, seed=9, n_motifs=4, motif_min=6, motif_max=12)
    a, b = SuffixCache(16), SuffixCache(16)
    streams = [src.stream(700, r) for r in range(6)]
    for r in range(6):
        a.cache_prompt(r, streams[r][:600])
        a.update_response(r, streams[r][600:600 + r + 1])
        b.cache_prompt_async(r, streams[r][:600], streams[r][600:600 + r + 1])
        assert b.has_cached_prompt(r)
        with pytest.raises(ValueError):
            b.cache_prompt_async(r, [1, 2])
    for c in (a, b):
        c.update_responses([0, 3], np.asarray([7, 8, 9], np.int32), np.asarray([2, 1], np.int32))
        c.evict_prompt(4)
    for r in (0, 1, 2, 3, 5):
        ea, eb = a._prompt_tree(r).export(), b._prompt_tree(r).export()
        assert all(np.array_equal(ea[k], eb[k]) for k in ea), r
        assert b._prompt_tree(r).selfcheck() == 0
    ga, gb = a._global_tree().export(), b._global_tree().export()
    assert all(np.array_equal(ga[k], gb[k]) for k in ga)
    # a cache destroyed while builds are in flight joins them
    c = SuffixCache(64)
    for r in range(4):
        c.cache_prompt_async(r, src.stream(3000, 10 + r))
    del c
