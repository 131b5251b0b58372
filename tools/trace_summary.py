"""Summarise a rocprofv3 kernel trace (rocpd sqlite) of bench.py: per-step kernel time, idle gaps, one layer's timeline.
usage: python tools/trace_summary.py <results.db> [n_rounds_to_average] [lanes] [rounds_to_skip_at_the_end]
A round = every request advances one engine step; with `lanes` interleaved lanes (bench.py --lanes) it holds that many lane
steps, each ending with a suffix_match_kernel launch (the step marker used here)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # e.g. the extra all-to-all steps bench.py runs after the timed region
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "suffix_match_kernel" in r[0]]
a, b = idx[-(nst + skip) * lanes - 1], idx[-skip * lanes - 1]
seg = rows[a:b]
T = (seg[-1][1] - seg[0][1]) / 1e3
print(f"wall per round {T / nst:.1f} us over {nst} rounds of {lanes} lane step(s)")
by = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, st in seg:
    k = n.split("(")[0][-48:]
    by[k][0] += 1
    by[k][1] += (e - s) / 1e3
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{v[1] / nst:9.1f} us/round {v[0] / nst:6.1f} calls/round avg {v[1] / v[0]:7.1f}  {k}")
gaps = collections.defaultdict(lambda: [0, 0.0])
ce, prev, busy = seg[0][2], seg[0][0], 0
cs = seg[0][1]
for n, s, e, st in seg[1:]:
    if s > ce:
        k = (prev.split("(")[0][-28:], n.split("(")[0][-28:])
        gaps[k][0] += 1
        gaps[k][1] += (s - ce) / 1e3
        busy += ce - cs
        cs = s
    if e > ce:
        ce, prev = e, n
busy += ce - cs
print(f"busy {busy / 1e3 / nst:.1f} us/round, idle {(T - busy / 1e3) / nst:.1f} us/round")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"{v[1] / nst:8.1f} us/round n/round {v[0] / nst:5.1f} avg {v[1] / v[0]:6.1f}   {k[0]} -> {k[1]}")
a = idx[-skip * lanes - 2]
t0 = rows[a][1]
for n, s, e, st in rows[a:a + 24]:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} st{st} {n.split('(')[0][-44:]}")
