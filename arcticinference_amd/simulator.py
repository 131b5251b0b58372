"""Offline suffix-decoding simulator on the MI355X suffix cache: the user-facing acceptance-length tool of the
reference (`arctic_inference/common/suffix_cache/simulator.py`), same command line, same per-step records and summary
columns, so sweeps written for the reference run unchanged:

    python -m arcticinference_amd.simulator data.jsonl --num-train 200 --num-eval 50 --max-depth 32 64 \
        --max-spec-factor 1.0 2.0 --use-tree-spec false --output steps.csv

Rows of the dataset need a prompt and a response column (names configurable); both may be token-id lists already
(no tokenizer needed) or strings, in which case `--tokenizer` must name a tokenizer that is available locally.
The model is the reference's (`simulator.py:33-114`): the target emits the recorded response; every step asks the
cache for a candidate, accepts the longest root path that matches the recorded continuation, appends one bonus token
and feeds the step's tokens back into the cache.  Both candidate forms run the HIP matcher: `use_tree_spec=false` (what the
vLLM plugin uses) and, since r04, `use_tree_spec=true` (the priority-queue expansion; a query that meets a node with more than
15 children is evaluated on the host trees instead)."""
from __future__ import annotations

import argparse
import itertools
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import pandas as pd

CONFIG_COLUMNS = ["num_eval", "num_train", "seed", "max_depth", "max_spec_tokens", "max_spec_factor", "min_token_prob",
                  "use_tree_spec", "use_cached_prompt"]


def simulate_request(cache, request_id, prompt: List[int], truth: List[int], max_spec_tokens: int, max_spec_factor: float,
                     min_token_prob: float, use_tree_spec: bool, use_cached_prompt: bool) -> List[Dict]:
    """One request against `cache` (any object with the SuffixCache surface).  Mirrors suffix_decode (simulator.py:33-114)."""
    if not max_spec_tokens:
        max_spec_tokens = cache.max_depth
    if use_cached_prompt:
        cache.cache_prompt(request_id, prompt)
    steps: List[Dict] = []
    done = 0
    history = list(prompt)
    while done < len(truth):
        t0 = time.perf_counter()
        cand = cache.speculate(request_id, history, max_spec_tokens=max_spec_tokens, max_spec_factor=max_spec_factor,
                               min_token_prob=min_token_prob, use_tree_spec=use_tree_spec,
                               use_cached_prompt=use_cached_prompt)
        spec_s = time.perf_counter() - t0
        # walk the candidate tree from its root along the recorded continuation
        accepted: List[int] = []
        node = -1
        for tok in truth[done:]:
            nxt = next((c for c, p in enumerate(cand.parents) if p == node and cand.token_ids[c] == tok), None)
            if nxt is None:
                break
            accepted.append(tok)
            node = nxt
        emitted = list(accepted)
        done += len(accepted)
        if done < len(truth):
            emitted.append(truth[done])       # the target's own (bonus) token
            done += 1
        history.extend(emitted)
        t0 = time.perf_counter()
        cache.update_response(request_id, emitted)
        update_s = time.perf_counter() - t0
        steps.append({"step": len(steps), "match_len": cand.match_len, "score": cand.score,
                      "num_spec_toks": len(cand.token_ids), "num_accept_toks": len(accepted),
                      "num_out_toks": len(emitted), "spec_ms": spec_s * 1e3, "update_ms": update_s * 1e3})
    if use_cached_prompt:
        cache.evict_prompt(request_id)
    return steps


def split_data(data: pd.DataFrame, train: Optional[pd.DataFrame], num_eval: Optional[int], num_train: Optional[int],
               seed: int) -> Tuple[pd.DataFrame, pd.DataFrame]:
    """(eval rows, train rows), the reference's sampling rule (simulator.py:117-146)."""
    if train is None:
        if num_eval is None:
            num_eval = len(data) - num_train
        if num_train is None:
            num_train = len(data) - num_eval
        if num_train + num_eval > len(data):
            raise ValueError("num_train + num_eval exceeds the dataset")
        mixed = data.sample(frac=1, random_state=seed)
        return mixed.tail(num_eval), mixed.head(num_train)
    ev = data if num_eval is None else data.sample(frac=1, random_state=seed).head(num_eval)
    tr = train if num_train is None else train.sample(frac=1, random_state=seed).head(num_train)
    return ev, tr


def run_task(make_cache, data, train, task_id, num_eval, num_train, seed, max_depth, max_spec_tokens, max_spec_factor,
             min_token_prob, use_tree_spec, use_cached_prompt, progress: bool = False) -> List[Dict]:
    ev, tr = split_data(data, train, num_eval, num_train, seed)
    cache = make_cache(max_depth)
    for rid, row in tr.iterrows():
        # recorded responses warm the global tree under ids that cannot collide with the evaluation ids (:171-176)
        cache.update_response(-1 - rid + 1, row["response"])
    out: List[Dict] = []
    rows = ev.iterrows()
    if progress:
        from tqdm import tqdm
        rows = tqdm(rows, total=len(ev), desc=f"task {task_id}")
    for rid, row in rows:
        # the reference hands `max_depth` to suffix_decode as its max_spec_tokens (:181-186); `max_spec_tokens` is
        # recorded in the output but not applied — kept, so numbers stay comparable
        steps = simulate_request(cache, rid, row["prompt"], row["response"], max_depth, max_spec_factor, min_token_prob,
                                 use_tree_spec, use_cached_prompt)
        for s in steps:
            s.update(task_id=task_id, request_id=rid, num_eval=len(ev), num_train=len(tr), seed=seed, max_depth=max_depth,
                     max_spec_tokens=max_spec_tokens, max_spec_factor=max_spec_factor, min_token_prob=min_token_prob,
                     use_tree_spec=use_tree_spec, use_cached_prompt=use_cached_prompt)
        out.extend(steps)
    return out


def summarize(df: pd.DataFrame, config_cols: Sequence[str] = CONFIG_COLUMNS) -> pd.DataFrame:
    """One row per task: the reference's summary columns (simulator.py:212-248)."""
    per_req = df.groupby(["task_id", "request_id"]).agg(out=("num_out_toks", "sum"), steps=("step", "count"))
    per_req["speedup"] = per_req["out"] / per_req["steps"]
    req_speedup = per_req.groupby("task_id")["speedup"].mean()
    cols = ["task_id"] + list(config_cols)
    s = df.groupby(cols).agg(sum_accept=("num_accept_toks", "sum"), sum_spec=("num_spec_toks", "sum"),
                             sum_out=("num_out_toks", "sum"), avg_accept_toks=("num_accept_toks", "mean"),
                             avg_spec_toks=("num_spec_toks", "mean"), sum_spec_ms=("spec_ms", "sum"),
                             sum_update_ms=("update_ms", "sum")).reset_index()
    s["accept_rate"] = s["sum_accept"] / s["sum_spec"]
    s["req_speedup"] = s["task_id"].map(req_speedup)
    s["spec_ms_per_tok"] = s["sum_spec_ms"] / s["sum_spec"]
    s["update_ms_per_tok"] = s["sum_update_ms"] / s["sum_out"]
    drop = [c for c in cols if c != "task_id" and s[c].nunique() == 1]
    drop += ["sum_accept", "sum_spec", "sum_out", "sum_spec_ms", "sum_update_ms"]
    return s.drop(columns=drop).set_index("task_id")


# ---- data --------------------------------------------------------------------------------------------
def read_table(path: str, fmt: Optional[str], prompt_col: str, response_col: str) -> pd.DataFrame:
    fmt = (fmt or os.path.splitext(path)[1].lstrip(".")).lower()
    readers = {"json": lambda p: pd.read_json(p), "jsonl": lambda p: pd.read_json(p, lines=True),
               "csv": pd.read_csv, "parquet": pd.read_parquet}
    if fmt not in readers:
        raise ValueError(f"unsupported dataset format '{fmt}' (json, jsonl, csv, parquet)")
    df = readers[fmt](path)
    for c in (prompt_col, response_col):
        if c not in df.columns:
            raise ValueError(f"column '{c}' not in {path}")
    return df[[prompt_col, response_col]].rename(columns={prompt_col: "prompt", response_col: "response"})


def tokenized(df: pd.DataFrame, tokenizer_name: Optional[str]) -> pd.DataFrame:
    def is_ids(v):
        return isinstance(v, (list, tuple)) or hasattr(v, "tolist")
    if all(is_ids(v) for col in ("prompt", "response") for v in df[col]):
        out = df.copy()
        for col in ("prompt", "response"):
            out[col] = [[int(t) for t in (v.tolist() if hasattr(v, "tolist") else v)] for v in df[col]]
        return out
    if not tokenizer_name:
        raise ValueError("the dataset holds text: pass --tokenizer (a tokenizer available locally), or token-id lists")
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(tokenizer_name)
    out = df.copy()
    out["prompt"] = [tok.encode(str(v)) for v in df["prompt"]]
    out["response"] = [tok.encode(str(v), add_special_tokens=False) for v in df["response"]]
    return out


# ---- command line ------------------------------------------------------------------------------------
def _flag(v: str) -> bool:
    if v.lower() in ("true", "1", "yes"):
        return True
    if v.lower() in ("false", "0", "no"):
        return False
    raise argparse.ArgumentTypeError(f"expected a boolean, got '{v}'")


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="suffix-decoding simulator (reference CLI, MI355X suffix cache)")
    ap.add_argument("dataset", type=str, help="Path to the dataset file")
    ap.add_argument("--format", type=str, help="Format of the dataset file, uses its extension if not provided")
    ap.add_argument("--train-dataset", type=str, help="Path to a separate dataset file for training")
    ap.add_argument("--prompt-column", type=str, default="prompt")
    ap.add_argument("--response-column", type=str, default="response")
    ap.add_argument("--num-train", type=int, nargs="+", default=[None])
    ap.add_argument("--num-eval", type=int, nargs="+", default=[None])
    ap.add_argument("--seed", type=int, nargs="+", default=[0])
    ap.add_argument("--tokenizer", type=str, help="Name of a locally available HuggingFace tokenizer")
    ap.add_argument("--output", "-o", type=str, help="The path to the output CSV file (per-step records)")
    ap.add_argument("--parallel", "-p", type=int, default=1,
                    help="accepted for compatibility; tasks run one after another (one GPU suffix cache at a time)")
    ap.add_argument("--max-depth", type=int, nargs="+", default=[64])
    ap.add_argument("--max-spec-tokens", type=int, nargs="+", default=[0])
    ap.add_argument("--max-spec-factor", type=float, nargs="+", default=[1.0])
    ap.add_argument("--min-token-prob", type=float, nargs="+", default=[0.1])
    ap.add_argument("--use-tree-spec", type=_flag, nargs="+", default=[True])
    ap.add_argument("--use-cached-prompt", type=_flag, nargs="+", default=[True])
    return ap


def main(args: argparse.Namespace, make_cache=None) -> pd.DataFrame:
    if make_cache is None:
        from .suffix_cache import SuffixCache
        make_cache = SuffixCache
    data = tokenized(read_table(args.dataset, args.format, args.prompt_column, args.response_column), args.tokenizer)
    train = None
    if args.train_dataset:
        train = tokenized(read_table(args.train_dataset, args.format, args.prompt_column, args.response_column), args.tokenizer)
    if train is None and args.num_train == [None] and args.num_eval == [None]:
        raise ValueError("give --num-train and/or --num-eval (or a --train-dataset)")
    grid = itertools.product(args.num_eval, args.num_train, args.seed, args.max_depth, args.max_spec_tokens,
                             args.max_spec_factor, args.min_token_prob, args.use_tree_spec, args.use_cached_prompt)
    records: List[Dict] = []
    for task_id, cfg in enumerate(grid):
        records.extend(run_task(make_cache, data, train, task_id, *cfg, progress=True))
    df = pd.DataFrame.from_records(records)
    summary = summarize(df)
    with pd.option_context("display.max_columns", None, "display.width", 200):
        print(summary)
    if args.output:
        df.to_csv(args.output, index=False)
    return summary


if __name__ == "__main__":
    main(get_parser().parse_args())
