#!/bin/bash
# usage: tools/ab_bench.sh "<label>:<ENV=.. ENV=..>" ...   — runs bench.py once per variant on the same box
for spec in "$@"; do
  label="${spec%%:*}"; envs="${spec#*:}"
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null)
  python3 - "$label" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d["roofline"]
print("run", sys.argv[1], round(d["value"], 1), "tok/s", round(d["ms_per_step"], 3), "ms", round(r["avg_launch_us"] or 0, 1), "us", round(r["frac"] or 0, 3))
print("   host", {k: round(v, 2) for k, v in d.get("host_timeline_ms_per_step", {}).items()})
PY
done
