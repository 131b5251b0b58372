from arcticinference_amd.suffix_cache import SuffixCache, SuffixSpecResult  # noqa: F401
