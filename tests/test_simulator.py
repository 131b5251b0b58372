"""The simulator CLI (arcticinference_amd/simulator.py = the reference's simulator.py surface) on CPU: tree-mode
candidates are evaluated on the host trees, so no GPU is needed.  Records must equal the ones the same driver
produces with the oracle cache (restatement pinned against the real reference), timings aside."""
import json

import numpy as np
import pandas as pd

from arcticinference_amd import simulator as S
from arcticinference_amd.suffix_cache import SuffixCache
from arcticinference_amd.workload import TokenSource
from oracle.suffix_oracle import OracleSuffixCache

TIMING = ["spec_ms", "update_ms"]


def _dataset(n=14, seed=4):
    src = TokenSource(vocab_size=300, seed=seed, n_motifs=5, motif_min=4, motif_max=9, p_motif=0.7)
    rows = []
    for r in range(n):
        p, g = src.request(r, 40, 30)
        rows.append({"prompt": [int(x) for x in p], "response": [int(x) for x in g]})
    return pd.DataFrame(rows)


def test_records_equal_oracle_driver():
    data = _dataset()
    cfg = dict(task_id=0, num_eval=5, num_train=8, seed=1, max_depth=16, max_spec_tokens=0, max_spec_factor=2.0,
               min_token_prob=0.1, use_tree_spec=True, use_cached_prompt=True)
    a = pd.DataFrame(S.run_task(SuffixCache, data, None, **cfg)).drop(columns=TIMING)
    b = pd.DataFrame(S.run_task(OracleSuffixCache, data, None, **cfg)).drop(columns=TIMING)
    assert len(a) > 20 and a["num_accept_toks"].sum() > 0
    pd.testing.assert_frame_equal(a, b)
    # every request reproduced its recorded response: out tokens add up to the response lengths
    ev, _ = S.split_data(data, None, 5, 8, 1)
    assert a.groupby("request_id")["num_out_toks"].sum().to_dict() == {i: len(r) for i, r in ev["response"].items()}


def test_cli_sweep_and_summary(tmp_path):
    data = _dataset(12, seed=9)
    path = tmp_path / "d.jsonl"
    with open(path, "w") as f:
        for _, row in data.iterrows():
            f.write(json.dumps({"prompt": row["prompt"], "response": row["response"]}) + "\n")
    out = tmp_path / "steps.csv"
    args = S.get_parser().parse_args([str(path), "--num-train", "6", "--num-eval", "4", "--max-depth", "8", "16",
                                      "--max-spec-factor", "1.0", "--use-tree-spec", "true", "--output", str(out)])
    summary = S.main(args)
    assert list(summary.index) == [0, 1] and "max_depth" in summary.columns           # the one column that varies
    for col in ("avg_accept_toks", "avg_spec_toks", "accept_rate", "req_speedup", "spec_ms_per_tok", "update_ms_per_tok"):
        assert col in summary.columns
    assert (summary["req_speedup"] >= 1.0).all()
    steps = pd.read_csv(out)
    assert set(S.CONFIG_COLUMNS) <= set(steps.columns) and steps["task_id"].nunique() == 2
    assert np.isclose(summary.loc[0, "avg_accept_toks"], steps[steps.task_id == 0]["num_accept_toks"].mean())
