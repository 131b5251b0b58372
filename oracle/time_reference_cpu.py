"""ORACLE — test infrastructure, build-container use only (reads /root/reference through oracle/ref_loader.py).

BASELINE.md §3.1: the REAL reference suffix tree (oracle/_ref, compiled from the reference's sources) driven with the
seeded 4096-prompt / 256-generation replay in the style of simulator.suffix_decode, timed on this container's host
cores: microseconds per cache_prompt / speculate / update_response and the mean accepted length.  One thread, and
`--procs` independent processes (the reference's only form of parallelism for this code).  Writes one JSON document;
the committed copy is profiles/r01_reference_cpu_suffix.json.

    python oracle/time_reference_cpu.py --requests 16 --procs 8 > profiles/r01_reference_cpu_suffix.json
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    first, n_req, prompt_len, gen_len = args
    from oracle import ref_loader
    from arcticinference_amd.workload import TokenSource
    _, _, SuffixCache, _ = ref_loader.load()
    src = TokenSource(seed=0)
    cache = SuffixCache(64)
    t_prompt = t_spec = t_upd = 0.0
    n_spec = n_upd_tok = steps = accepted = speculated = 0
    for r in range(first, first + n_req):
        prompt, gt = src.request(r, prompt_len, gen_len)
        prompt, gt = [int(x) for x in prompt], [int(x) for x in gt]
        t0 = time.perf_counter()
        cache.cache_prompt(r, prompt)
        t_prompt += time.perf_counter() - t0
        resp = []
        while len(resp) < len(gt):
            text = (prompt + resp)[-cache.max_depth:]
            t0 = time.perf_counter()
            res = cache.speculate(r, text, max_spec_tokens=32, max_spec_factor=1.0, max_spec_offset=0.0, min_token_prob=0.1)
            t_spec += time.perf_counter() - t0
            n_spec += 1
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
            t0 = time.perf_counter()
            cache.update_response(r, new)
            t_upd += time.perf_counter() - t0
            n_upd_tok += len(new)
            steps += 1
            accepted += a
            speculated += len(res.token_ids)
        cache.evict_prompt(r)
    return dict(requests=n_req, steps=steps, accepted=accepted, speculated=speculated, t_prompt=t_prompt, t_spec=t_spec,
                t_upd=t_upd, n_spec=n_spec, n_upd_tok=n_upd_tok)


def summarize(parts, wall):
    tot = {k: sum(p[k] for p in parts) for k in parts[0]}
    return {"requests": tot["requests"], "steps": tot["steps"],
            "mean_accepted_per_step": tot["accepted"] / tot["steps"],
            "acceptance_rate": tot["accepted"] / max(tot["speculated"], 1),
            "us_per_cache_prompt_4096": tot["t_prompt"] / tot["requests"] * 1e6,
            "us_per_speculate": tot["t_spec"] / tot["n_spec"] * 1e6,
            "us_per_updated_token": tot["t_upd"] / tot["n_upd_tok"] * 1e6,
            "wall_s": wall, "generated_tokens_per_s": tot["requests"] * 256 / wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=16, help="requests per process")
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    t0 = time.perf_counter()
    one = summarize([run((0, a.requests, 4096, 256))], time.perf_counter() - t0)
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(a.procs) as pool:
        parts = pool.map(run, [(i * a.requests, a.requests, 4096, 256) for i in range(a.procs)])
    many = summarize(parts, time.perf_counter() - t0)
    json.dump({"what": "the reference's csrc/suffix_cache + suffix_cache.py (oracle/_ref), seeded replay of 4096-token prompts "
                       "+ 256 generated tokens per request, max_depth 64, path candidates, suffix proposer alone",
               "host": {"cpus": os.cpu_count(), "note": "build container, not the GPU box"},
               "single_thread": one, "processes": a.procs, "independent_processes": many}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
