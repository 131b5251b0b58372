"""Stands where the reference's pybind11 module sits (csrc/suffix_cache/pybind.cc:24-38)."""
from arcticinference_amd.suffix_cache import Candidate, SuffixTree  # noqa: F401
