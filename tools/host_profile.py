#!/usr/bin/env python3
"""cProfile of the engine's host side on the bench workload (which Python / ctypes calls the step's host chain spends
its time in).  usage: python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from arcticinference_amd.engine import HotPathEngine, ModelShape, SpecConfig
from arcticinference_amd.workload import TokenSource

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, PL, GL = 64, 4096, 256
shape, spec = ModelShape(), SpecConfig(proposal_indexing="single_advance")
src = TokenSource(seed=0)
eng = HotPathEngine(shape, spec, B, PL + GL + 64, None, device="cuda", seed=0)
pool = [src.stream(PL + GL + 128, r) for r in range(B + 40)]
nxt = [B]
streams = {r: pool[r] for r in range(B)}
eng.add_requests(list(range(B)), list(range(B)), [pool[r][:PL] for r in range(B)],
                 [pool[r][PL:PL + (r * GL) // B + 1] for r in range(B)])


def truth(r, n):
    s = streams[r.req_id]
    p = len(r.tokens)
    return s[p:p + n]


lane_slots = [list(range(l, B, n_lanes)) for l in range(n_lanes)]
pending = [None] * n_lanes


def run_step():
    for l in range(n_lanes):
        if pending[l] is not None:
            eng.finish(pending[l])
            for slot in lane_slots[l]:
                r = eng.requests[slot]
                if len(r.tokens) - r.num_prompt >= GL:
                    rid = nxt[0]
                    nxt[0] += 1
                    streams[rid] = pool[rid]
                    eng.add_request(slot, rid, pool[rid][:PL], pool[rid][PL:PL + 1])
        pending[l] = eng.begin(truth, lane_slots[l] if n_lanes > 1 else None, lane=l)


for _ in range(8):
    run_step()
import gc
gc.collect()
gc.freeze()
pr = cProfile.Profile()
pr.enable()
import ctypes
from arcticinference_amd import _native as N
spec_build = spec_dev = 0.0
for _ in range(steps):
    run_step()
    b, d = ctypes.c_float(0), ctypes.c_float(0)
    N.lib().aic_sc_last_timing(eng.suffix_cache._h, ctypes.byref(b), ctypes.byref(d))
    spec_build += b.value
    spec_dev += d.value
pr.disable()
print(f"aic_sc_speculate_batch per call: host collection {spec_build / steps:.1f} us, staging copy .. stream sync {spec_dev / steps:.1f} us")
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
print({k: round(v / (steps + 8) * 1e3, 3) for k, v in eng.timeline.items()})
