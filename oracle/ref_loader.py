"""ORACLE — test infrastructure only.  Build-container use only.

Loads the REAL reference suffix tree (oracle/_ref/_C*.so, compiled from
/root/reference/csrc/suffix_cache by oracle/Makefile `ref`) and the reference's
own Python policy class (arctic_inference/common/suffix_cache/suffix_cache.py),
imported from /root/reference where it lies.  Nothing from the reference is
copied into the repo: this module exists so gen_golden.py can produce fixtures
and tests can cross-check the restatement when the reference is present.

/root/reference does not exist on the GPU box; `available()` is False there and
every caller must skip.
"""
from __future__ import annotations

import glob
import importlib.util
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("ARCTIC_REFERENCE_ROOT", "/root/reference")

_cached = None


def available() -> bool:
    return bool(glob.glob(os.path.join(_HERE, "_ref", "_C*.so"))) and os.path.isdir(REF_ROOT)


def load():
    """Returns (SuffixTree, Candidate, SuffixCache, SuffixSpecResult) of the reference."""
    global _cached
    if _cached is not None:
        return _cached
    so = glob.glob(os.path.join(_HERE, "_ref", "_C*.so"))
    if not so:
        raise RuntimeError("oracle/_ref is not built; run `make -C oracle ref` in the build container")
    spec = importlib.util.spec_from_file_location("_C", so[0])
    cmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cmod)

    # The reference policy file does `from arctic_inference.common.suffix_cache._C import ...`;
    # give it that name without importing the reference package (whose vllm parts need vLLM).
    names = ["arctic_inference", "arctic_inference.common", "arctic_inference.common.suffix_cache"]
    saved = {n: sys.modules.get(n) for n in names + [names[-1] + "._C"]}
    try:
        for n in names:
            m = types.ModuleType(n)
            m.__path__ = []  # mark as package
            sys.modules[n] = m
        sys.modules[names[-1] + "._C"] = cmod
        path = os.path.join(REF_ROOT, "arctic_inference", "common", "suffix_cache", "suffix_cache.py")
        pspec = importlib.util.spec_from_file_location("_arctic_ref_suffix_cache", path)
        pmod = importlib.util.module_from_spec(pspec)
        sys.modules["_arctic_ref_suffix_cache"] = pmod  # dataclasses look the module up by name
        pspec.loader.exec_module(pmod)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    _cached = (cmod.SuffixTree, cmod.Candidate, pmod.SuffixCache, pmod.SuffixSpecResult)
    return _cached
