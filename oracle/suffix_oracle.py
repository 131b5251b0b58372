"""ORACLE — test infrastructure only (never imported by the product package).

ctypes binding of oracle/liboracle_suffix.so (suffix_tree_oracle.cpp) plus a
pure-Python restatement of the reference's SuffixCache policy
(/root/reference/arctic_inference/common/suffix_cache/suffix_cache.py:57-222)
driving it.  Used by tests/ as the checker, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import Hashable, List, Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_suffix.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C++ restatement (g++ only; no reference sources involved)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "suffix_tree_oracle.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_st_create.restype = ctypes.c_void_p
        L.orc_st_create.argtypes = [ctypes.c_int]
        L.orc_st_destroy.argtypes = [ctypes.c_void_p]
        L.orc_st_num_seqs.argtypes = [ctypes.c_void_p]
        L.orc_st_append.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.orc_st_extend.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
        L.orc_st_speculate.restype = ctypes.c_int
        L.orc_st_speculate.argtypes = [
            ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int,
            ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
            ctypes.POINTER(ctypes.c_float), ctypes.c_int,
            ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)]
        _lib = L
    return _lib


@dataclass
class OracleCandidate:
    token_ids: List[int] = field(default_factory=list)
    parents: List[int] = field(default_factory=list)
    probs: List[float] = field(default_factory=list)
    score: float = 0.0
    match_len: int = 0


class OracleSuffixTree:
    """Same surface as the reference pybind class (csrc/suffix_cache/pybind.cc:32-37)."""

    def __init__(self, max_depth: int):
        self._h = ctypes.c_void_p(lib().orc_st_create(int(max_depth)))
        self._max_depth = int(max_depth)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_st_destroy(self._h)
            self._h = None

    def num_seqs(self) -> int:
        return lib().orc_st_num_seqs(self._h)

    def append(self, seq_id: int, token: int) -> None:
        lib().orc_st_append(self._h, int(seq_id), int(token))

    def extend(self, seq_id: int, tokens: Sequence[int]) -> None:
        n = len(tokens)
        arr = (ctypes.c_int32 * n)(*[int(t) for t in tokens])
        lib().orc_st_extend(self._h, int(seq_id), arr, n)

    def speculate(self, pattern: Sequence[int], max_spec_tokens: int, max_spec_factor: float = 1.0,
                  max_spec_offset: float = 0.0, min_token_prob: float = 0.1,
                  use_tree_spec: bool = False) -> OracleCandidate:
        n = len(pattern)
        pat = (ctypes.c_int32 * max(n, 1))(*[int(t) for t in pattern])
        cap = max(int(max_spec_tokens), 1)
        toks = (ctypes.c_int32 * cap)()
        pars = (ctypes.c_int32 * cap)()
        prbs = (ctypes.c_float * cap)()
        score = ctypes.c_float(0.0)
        mlen = ctypes.c_int32(0)
        m = lib().orc_st_speculate(self._h, pat, n, int(max_spec_tokens), float(max_spec_factor),
                                   float(max_spec_offset), float(min_token_prob), int(bool(use_tree_spec)),
                                   toks, pars, prbs, cap, ctypes.byref(score), ctypes.byref(mlen))
        return OracleCandidate(list(toks[:m]), list(pars[:m]), [float(p) for p in prbs[:m]],
                               float(score.value), int(mlen.value))


class OracleSuffixCache:
    """Restatement of the SuffixCache policy (suffix_cache.py:57-222): one global
    tree of responses + one prompt tree per live request; the prompt-tree result
    is kept unless the global tree scores strictly higher (:220-221)."""

    def __init__(self, max_depth: int = 64, tree_cls=OracleSuffixTree):
        self._max_depth = max_depth
        self._tree_cls = tree_cls
        self._global = tree_cls(max_depth)
        self._prompt = {}
        self._seq_of = {}

    @property
    def max_depth(self) -> int:
        return self._max_depth

    def has_cached_prompt(self, req_id: Hashable) -> bool:
        return req_id in self._prompt

    def cached_prompt_ids(self):
        return list(self._prompt.keys())

    def cache_prompt(self, req_id: Hashable, prompt_token_ids: Sequence[int]) -> None:
        if req_id in self._prompt:
            raise ValueError(f"Prompt already exists for request '{req_id}'")
        t = self._tree_cls(self._max_depth)
        t.extend(0, list(prompt_token_ids))
        self._prompt[req_id] = t

    def evict_prompt(self, req_id: Hashable) -> None:
        if req_id not in self._prompt:
            raise ValueError(f"Prompt does not exist for request '{req_id}'")
        del self._prompt[req_id]

    def update_response(self, req_id: Hashable, token_ids) -> None:
        if req_id not in self._seq_of:
            self._seq_of[req_id] = len(self._seq_of)
        sid = self._seq_of[req_id]
        toks = [token_ids] if isinstance(token_ids, int) else list(token_ids)
        self._global.extend(sid, toks)
        if req_id in self._prompt:
            self._prompt[req_id].extend(0, toks)

    def speculate(self, req_id: Hashable, pattern: Sequence[int], max_spec_tokens: Optional[int] = None,
                  max_spec_factor: float = 1.0, max_spec_offset: float = 0.0, min_token_prob: float = 0.1,
                  use_tree_spec: bool = False, use_cached_prompt: bool = True) -> OracleCandidate:
        if use_cached_prompt and req_id not in self._prompt:
            raise ValueError(f"Prompt does not exist for request '{req_id}'")
        if not len(pattern):
            raise ValueError("Pattern must not be empty")
        if max_spec_tokens is None:
            max_spec_tokens = self._max_depth
        pattern = list(pattern)
        if len(pattern) > self._max_depth:
            pattern = pattern[-self._max_depth:]
        best = OracleCandidate()
        if use_cached_prompt:
            best = self._prompt[req_id].speculate(pattern, max_spec_tokens, max_spec_factor,
                                                  max_spec_offset, min_token_prob, use_tree_spec)
        g = self._global.speculate(pattern, max_spec_tokens, max_spec_factor, max_spec_offset,
                                   min_token_prob, use_tree_spec)
        if g.score > best.score:
            best = g
        return best
