"""CPU timing of the suffix-tree appends of one engine step (no GPU): 64 prompt trees of 4096 tokens each (depth 64) and
one global tree, every step appends `per_step` tokens per request to its prompt tree and to the global tree — what
aic_sc_update_responses does — with the trees gone cold in between (a 256 MiB buffer is streamed through the caches)."""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from arcticinference_amd import _native as N          # noqa: E402
from arcticinference_amd.workload import TokenSource   # noqa: E402


def main(n_req=64, prompt=4096, steps=40, per_step=2, depth=64):
    lib = N.lib()
    I32P = ctypes.POINTER(ctypes.c_int32)
    src = TokenSource(seed=0)
    streams = [np.asarray(src.stream(prompt + steps * per_step + 8, r), dtype=np.int32) for r in range(n_req)]
    trees = [lib.aic_st_create(depth) for _ in range(n_req)]
    glob = lib.aic_st_create(depth)
    t0 = time.perf_counter()
    for r, t in enumerate(trees):
        a = np.ascontiguousarray(streams[r][:prompt])
        N.check(lib.aic_st_extend(t, 0, a.ctypes.data_as(I32P), prompt))
    print(f"build: {time.perf_counter() - t0:.2f} s for {n_req} prompt trees")
    evict = np.zeros(256 << 20, dtype=np.uint8)
    tot = 0.0
    for s in range(steps):
        evict += 1                      # the trees go cold, as between two engine steps
        at = prompt + s * per_step
        t0 = time.perf_counter()
        for r in range(n_req):
            for j in range(per_step):
                tok = int(streams[r][at + j])
                lib.aic_st_append(glob, r, tok)
                lib.aic_st_append(trees[r], 0, tok)
        tot += time.perf_counter() - t0
    n = steps * n_req * per_step
    print(f"append: {tot / steps * 1e3:.3f} ms per step of {n_req} x {per_step} tokens ({tot / n * 1e6:.2f} us per token, both trees; "
          f"includes ~0.3 us of ctypes per call)")
    assert lib.aic_st_selfcheck(glob) == 0 and all(lib.aic_st_selfcheck(t) == 0 for t in trees[:4])


if __name__ == "__main__":
    main()
