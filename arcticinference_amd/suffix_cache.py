"""SuffixCache / SuffixSpecResult / SuffixTree with the reference's names, arguments, defaults and
errors (/root/reference/arctic_inference/common/suffix_cache/suffix_cache.py:24-222 and the pybind
class of csrc/suffix_cache/pybind.cc:24-38), backed by libarctic_hip.so.

Tree updates run in the library's host arena; candidate matching runs on the MI355X.  The extra
method `speculate_batch` is what the model-runner patch uses: one device round trip for the whole
engine step instead of one host call per request (model_runner.py:680-744 loops on the CPU).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Hashable, List, Optional, Sequence, Union

import numpy as np

from . import _native as N


@dataclass
class SuffixSpecResult:
    """Same fields as the reference dataclass (suffix_cache.py:24-54)."""
    token_ids: List[int] = field(default_factory=list)
    parents: List[int] = field(default_factory=list)
    probs: List[float] = field(default_factory=list)
    score: float = 0.0
    match_len: int = 0

    @staticmethod
    def from_candidate(candidate) -> "SuffixSpecResult":
        return SuffixSpecResult(token_ids=list(candidate.token_ids), parents=list(candidate.parents),
                                probs=list(candidate.probs), score=candidate.score,
                                match_len=candidate.match_len)


Candidate = SuffixSpecResult  # the pybind `Candidate` has the same five fields (pybind.cc:25-30)


def _i32(seq) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(seq, dtype=np.int32).reshape(-1))


def _stream() -> int:
    try:
        return N.current_stream_ptr()
    except Exception:
        return 0


class SuffixTree:
    """Drop-in for `arctic_inference.common.suffix_cache._C.SuffixTree`."""

    def __init__(self, max_depth: int, _handle=None, _owner=None):
        self._owner = _owner
        if _handle is not None:
            self._h = ctypes.c_void_p(_handle)
        else:
            h = N.lib().aic_st_create(int(max_depth))
            if not h:
                raise ValueError(N.lib().aic_last_error().decode())
            self._h = ctypes.c_void_p(h)
        self._max_depth = int(max_depth)

    def __del__(self):
        if getattr(self, "_owner", None) is None and getattr(self, "_h", None):
            N.lib().aic_st_destroy(self._h)
            self._h = None

    def num_seqs(self) -> int:
        return N.lib().aic_st_num_seqs(self._h)

    def append(self, seq_id: int, token: int) -> None:
        N.check(N.lib().aic_st_append(self._h, int(seq_id), int(token)))

    def extend(self, seq_id: int, tokens: Sequence[int]) -> None:
        a = _i32(tokens)
        N.check(N.lib().aic_st_extend(self._h, int(seq_id), a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), a.size))

    def speculate(self, pattern: Sequence[int], max_spec_tokens: int, max_spec_factor: float = 1.0,
                  max_spec_offset: float = 0.0, min_token_prob: float = 0.1,
                  use_tree_spec: bool = False) -> SuffixSpecResult:
        pat = _i32(pattern)
        if pat.size == 0:
            return SuffixSpecResult()
        cap = max(int(max_spec_tokens), 1)
        toks = np.empty(cap, np.int32)
        pars = np.empty(cap, np.int32)
        prbs = np.empty(cap, np.float32)
        score = ctypes.c_float(0.0)
        mlen = ctypes.c_int32(0)
        P32 = ctypes.POINTER(ctypes.c_int32)
        m = N.check(N.lib().aic_st_speculate(
            self._h, pat.ctypes.data_as(P32), pat.size, int(max_spec_tokens), float(max_spec_factor),
            float(max_spec_offset), float(min_token_prob), int(bool(use_tree_spec)),
            toks.ctypes.data_as(P32), pars.ctypes.data_as(P32), prbs.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            cap, ctypes.byref(score), ctypes.byref(mlen), _stream()))
        return SuffixSpecResult(toks[:m].tolist(), pars[:m].tolist(), [float(x) for x in prbs[:m]],
                                float(score.value), int(mlen.value))

    # ---- test / debug helpers ---------------------------------------------------------------------
    def selfcheck(self) -> int:
        return N.check(N.lib().aic_st_selfcheck(self._h))

    def export(self) -> dict:
        """The flattened HBM image as numpy arrays, taken from the host mirror (no GPU needed)."""
        c = [ctypes.c_int32(0) for _ in range(4)]
        N.check(N.lib().aic_st_export(self._h, *[ctypes.byref(x) for x in c], None, None, None, None, None))
        nn, ns, nt, nq = [int(x.value) for x in c]
        nodes = np.zeros((nn, 8), np.int32)
        hsh = np.zeros((ns, 4), np.int32)
        toks = np.zeros(max(nt, 1), np.int32)
        base = np.zeros(max(nq, 1), np.int32)
        ids = np.zeros(max(nq, 1), np.int32)
        N.check(N.lib().aic_st_export(self._h, *[ctypes.byref(x) for x in c], nodes.ctypes.data, hsh.ctypes.data,
                                      toks.ctypes.data, base.ctypes.data, ids.ctypes.data))
        return {"nodes": nodes, "hash": hsh, "tokens": toks[:nt], "seq_base": base[:nq], "seq_ids": ids[:nq]}


class SuffixCache:
    """Same surface as the reference class (suffix_cache.py:57-222) + `speculate_batch`."""

    def __init__(self, max_depth: int = 64):
        self._max_depth = max_depth
        h = N.lib().aic_sc_create(int(max_depth))
        if not h:
            raise ValueError(N.lib().aic_last_error().decode())
        self._h = ctypes.c_void_p(h)
        self._keys = {}        # req_id -> int64 key handed to the native side (never reused)
        self._prompt_ids = {}  # insertion-ordered set of req_ids with a cached prompt
        self._next_key = 0

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                N.lib().aic_sc_destroy(self._h)
            except TypeError:       # interpreter shutdown: the module globals are already gone
                pass
            self._h = None

    def _key(self, req_id: Hashable) -> int:
        k = self._keys.get(req_id)
        if k is None:
            k = self._next_key
            self._next_key += 1
            self._keys[req_id] = k
        return k

    @property
    def max_depth(self) -> int:
        return self._max_depth

    def has_cached_prompt(self, req_id: Hashable) -> bool:
        return req_id in self._prompt_ids

    def cached_prompt_ids(self) -> List[Hashable]:
        return list(self._prompt_ids.keys())

    def cache_prompt(self, req_id: Hashable, prompt_token_ids: Sequence[int]):
        if req_id in self._prompt_ids:
            raise ValueError(f"Prompt already exists for request '{req_id}'")
        a = _i32(prompt_token_ids)
        N.check(N.lib().aic_sc_cache_prompt(self._h, self._key(req_id), a.ctypes.data, a.size))
        self._prompt_ids[req_id] = True

    def cache_prompt_async(self, req_id: Hashable, prompt_token_ids: Sequence[int],
                           response_token_ids: Sequence[int] = ()):
        """cache_prompt(req_id, prompt) + update_response(req_id, response) with the prompt tree built on a host
        thread (the caller knows the prompt before it needs the tree: while the request is being prefilled).  The
        first call that touches the request's prompt tree waits for the build; results equal the synchronous calls."""
        if req_id in self._prompt_ids:
            raise ValueError(f"Prompt already exists for request '{req_id}'")
        a = _i32(prompt_token_ids)
        r = _i32(response_token_ids)
        N.check(N.lib().aic_sc_cache_prompt_async(self._h, self._key(req_id), a.ctypes.data, a.size,
                                                  r.ctypes.data if r.size else None, r.size))
        self._prompt_ids[req_id] = True

    def cache_prompts(self, req_ids: Sequence[Hashable], prompts: Sequence[Sequence[int]], n_threads: int = 8):
        """Several prompts at once; the independent prompt trees are built on host threads."""
        for r in req_ids:
            if r in self._prompt_ids:
                raise ValueError(f"Prompt already exists for request '{r}'")
        arrs = [_i32(p) for p in prompts]
        keys = np.asarray([self._key(r) for r in req_ids], np.int64)
        lens = np.asarray([a.size for a in arrs], np.int32)
        flat = np.concatenate(arrs) if arrs else np.zeros(0, np.int32)
        N.check(N.lib().aic_sc_cache_prompts(self._h, len(arrs), keys.ctypes.data, flat.ctypes.data,
                                             lens.ctypes.data, int(n_threads)))
        for r in req_ids:
            self._prompt_ids[r] = True

    def evict_prompt(self, req_id: Hashable):
        if req_id not in self._prompt_ids:
            raise ValueError(f"Prompt does not exist for request '{req_id}'")
        N.check(N.lib().aic_sc_evict_prompt(self._h, self._key(req_id)))
        del self._prompt_ids[req_id]

    def update_response(self, req_id: Hashable, token_ids: Union[int, Sequence[int]]):
        a = _i32([token_ids] if isinstance(token_ids, (int, np.integer)) else token_ids)
        N.check(N.lib().aic_sc_update_response(self._h, self._key(req_id), a.ctypes.data, a.size))

    def update_responses(self, req_ids: Sequence[Hashable], flat_tokens: np.ndarray, lens: np.ndarray):
        """update_response for several requests in one native call: request i appends lens[i] tokens of the
        concatenated int32 array (the per-request loop of model_runner.py:657-678)."""
        keys = np.asarray([self._key(r) for r in req_ids], np.int64)
        flat = np.ascontiguousarray(flat_tokens, dtype=np.int32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if len(keys) != len(lens) or int(lens.sum()) != flat.size:
            raise ValueError("update_responses: lens do not add up to the token array")
        N.check(N.lib().aic_sc_update_responses(self._h, len(keys), keys.ctypes.data, flat.ctypes.data, lens.ctypes.data))

    def warm(self, req_ids: Sequence[Hashable]) -> None:
        """Read-only CPU-cache warm-up of the trees of these requests (aic_sc_warm): call while waiting for the GPU."""
        keys = np.asarray([self._key(r) for r in req_ids], np.int64)
        N.check(N.lib().aic_sc_warm(self._h, len(keys), keys.ctypes.data))

    def speculate(self, req_id: Hashable, pattern: Sequence[int], max_spec_tokens: Optional[int] = None,
                  max_spec_factor: float = 1.0, max_spec_offset: float = 0.0, min_token_prob: float = 0.1,
                  use_tree_spec: bool = False, use_cached_prompt: bool = True) -> SuffixSpecResult:
        if use_cached_prompt and req_id not in self._prompt_ids:
            raise ValueError(f"Prompt does not exist for request '{req_id}'")
        if len(pattern) == 0:
            raise ValueError("Pattern must not be empty")
        if max_spec_tokens is None:
            max_spec_tokens = self.max_depth
        if use_tree_spec:
            return self._speculate_tree_mode(req_id, pattern, max_spec_tokens, max_spec_factor, max_spec_offset,
                                             min_token_prob, use_cached_prompt)
        return self.speculate_batch([req_id], [pattern], [max_spec_tokens], [max_spec_factor], [max_spec_offset],
                                    [min_token_prob], [use_cached_prompt])[0]

    def _speculate_tree_mode(self, req_id, pattern, max_spec_tokens, factor, offset, min_prob, use_prompt):
        """use_tree_spec = True (SuffixTree::_speculate_tree, suffix_tree.cc:226-274; the reference's simulator default, never
        the serving path): on the device since r04 (aic_sc_speculate_batch_tree: both trees, every suffix start, a priority
        queue per wave that reproduces std::priority_queue's order); a query the device gives back (a node with more than 15
        children on its way, a queue beyond 256 entries) is evaluated by the host trees, as every tree-mode query was before."""
        pattern = list(pattern)[-self._max_depth:]
        if N.lib().aic_device_count() <= 0:       # the host trees alone (CPU tests of the tree code; as before r04)
            return self._speculate_tree_mode_host(req_id, pattern, max_spec_tokens, factor, offset, min_prob, use_prompt)
        a = _i32(pattern)
        cap = max(1, int(max_spec_tokens))        # a tree candidate has branches: bounded by max_spec_tokens, not by max_depth
        key = np.asarray([self._key(req_id)], np.int64)
        lens = np.asarray([a.size], np.int32)
        mst = np.asarray([max_spec_tokens], np.int32)
        fac, off, mpr = (np.asarray([x], np.float32) for x in (factor, offset, min_prob))
        upr = np.asarray([1 if use_prompt else 0], np.int32)
        o_tok, o_par = np.zeros((1, cap), np.int32), np.zeros((1, cap), np.int32)
        o_prb = np.zeros((1, cap), np.float32)
        o_n, o_ml = np.zeros(1, np.int32), np.zeros(1, np.int32)
        o_sc = np.zeros(1, np.float32)
        N.check(N.lib().aic_sc_speculate_batch_tree(
            self._h, 1, key.ctypes.data, a.ctypes.data, lens.ctypes.data, mst.ctypes.data, fac.ctypes.data, off.ctypes.data,
            mpr.ctypes.data, upr.ctypes.data, cap, o_tok.ctypes.data, o_par.ctypes.data, o_prb.ctypes.data, o_n.ctypes.data,
            o_sc.ctypes.data, o_ml.ctypes.data, _stream()))
        k = int(o_n[0])
        if k >= 0:
            return SuffixSpecResult(o_tok[0, :k].tolist(), o_par[0, :k].tolist(), [float(x) for x in o_prb[0, :k]],
                                    float(o_sc[0]), int(o_ml[0]))
        return self._speculate_tree_mode_host(req_id, pattern, max_spec_tokens, factor, offset, min_prob, use_prompt)

    def _speculate_tree_mode_host(self, req_id, pattern, max_spec_tokens, factor, offset, min_prob, use_prompt):
        lib = N.lib()
        lib.aic_debug_tree_mode_on_host(1)
        try:
            result = SuffixSpecResult()
            if use_prompt:
                h = lib.aic_sc_prompt_tree(self._h, self._key(req_id))
                result = SuffixTree(self._max_depth, _handle=h, _owner=self).speculate(
                    pattern, max_spec_tokens, factor, offset, min_prob, True)
            g = SuffixTree(self._max_depth, _handle=lib.aic_sc_global_tree(self._h), _owner=self).speculate(
                pattern, max_spec_tokens, factor, offset, min_prob, True)
            return g if g.score > result.score else result
        finally:
            lib.aic_debug_tree_mode_on_host(0)

    def speculate_batch(self, req_ids: Sequence[Hashable], patterns: Sequence[Sequence[int]],
                        max_spec_tokens: Sequence[int], max_spec_factor: Sequence[float],
                        max_spec_offset: Sequence[float], min_token_prob: Sequence[float],
                        use_cached_prompt: Sequence[bool]) -> List[SuffixSpecResult]:
        """All requests of one engine step in one device round trip (path candidates)."""
        n = len(req_ids)
        if n == 0:
            return []
        for p in patterns:
            if len(p) == 0:
                raise ValueError("Pattern must not be empty")
        arrs = [_i32(p)[-self._max_depth:] for p in patterns]
        o_tok, o_prb, o_n, o_sc, o_ml = self.speculate_batch_arrays(
            req_ids, np.concatenate(arrs), np.asarray([a.size for a in arrs], np.int32), max_spec_tokens,
            max_spec_factor, max_spec_offset, min_token_prob, use_cached_prompt)
        out = []
        for i in range(n):
            k = int(o_n[i])
            out.append(SuffixSpecResult(o_tok[i, :k].tolist(), list(range(-1, k - 1)),
                                        [float(x) for x in o_prb[i, :k]], float(o_sc[i]), int(o_ml[i])))
        return out

    def speculate_batch_arrays(self, req_ids, flat_patterns: np.ndarray, pattern_lens: np.ndarray, max_spec_tokens,
                               max_spec_factor, max_spec_offset, min_token_prob, use_cached_prompt):
        """speculate_batch on flat arrays (the engine's per-step form: no per-request Python objects).  Patterns
        are concatenated int32 (at most max_depth tokens each).  Returns (tokens [n, cap], probs [n, cap],
        n_tokens [n], score [n], match_len [n]); candidate i is the path tokens[i, :n_tokens[i]]."""
        n = len(req_ids)
        for r, up in zip(req_ids, use_cached_prompt):
            if up and r not in self._prompt_ids:
                raise ValueError(f"Prompt does not exist for request '{r}'")
        flat = np.ascontiguousarray(flat_patterns, dtype=np.int32)
        lens = np.ascontiguousarray(pattern_lens, dtype=np.int32)
        if n and (int(lens.min()) <= 0 or int(lens.max()) > self._max_depth or int(lens.sum()) != flat.size):
            raise ValueError("Pattern must not be empty (and at most max_depth tokens, lens adding up)")
        keys = np.asarray([self._key(r) for r in req_ids], np.int64)
        mst = np.ascontiguousarray(max_spec_tokens, np.int32)
        fac = np.ascontiguousarray(max_spec_factor, np.float32)
        off = np.ascontiguousarray(max_spec_offset, np.float32)
        mpr = np.ascontiguousarray(min_token_prob, np.float32)
        upr = np.ascontiguousarray(use_cached_prompt, np.int32)
        cap = max(1, min(int(mst.max()), self._max_depth)) if n else 1
        o_tok = np.zeros((n, cap), np.int32)
        o_prb = np.zeros((n, cap), np.float32)
        o_n = np.zeros(n, np.int32)
        o_sc = np.zeros(n, np.float32)
        o_ml = np.zeros(n, np.int32)
        if n:
            N.check(N.lib().aic_sc_speculate_batch(
                self._h, n, keys.ctypes.data, flat.ctypes.data, lens.ctypes.data, mst.ctypes.data, fac.ctypes.data,
                off.ctypes.data, mpr.ctypes.data, upr.ctypes.data, cap, o_tok.ctypes.data, o_prb.ctypes.data,
                o_n.ctypes.data, o_sc.ctypes.data, o_ml.ctypes.data, _stream()))
        return o_tok, o_prb, o_n, o_sc, o_ml

    def last_stats(self) -> dict:
        us = ctypes.c_float(0)
        mb = ctypes.c_int64(0)
        nn = ctypes.c_int64(0)
        N.check(N.lib().aic_sc_last_stats(self._h, ctypes.byref(us), ctypes.byref(mb), ctypes.byref(nn)))
        return {"match_us": float(us.value), "mirrored_bytes": int(mb.value), "n_nodes": int(nn.value)}

    # test helpers
    def _global_tree(self) -> SuffixTree:
        return SuffixTree(self._max_depth, _handle=N.lib().aic_sc_global_tree(self._h), _owner=self)

    def _prompt_tree(self, req_id) -> SuffixTree:
        return SuffixTree(self._max_depth, _handle=N.lib().aic_sc_prompt_tree(self._h, self._key(req_id)), _owner=self)
