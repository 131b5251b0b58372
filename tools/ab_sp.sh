#!/bin/bash
# bench.py at N = 1 and rehearsed SP 2 / 4 / 8 (one and two lanes): value, ms per round, roofline fraction
set -e
run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-replay-check --no-lstm-leg "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['config']['lanes'])"; }
run
run --lanes 1
for sp in 2 4 8; do run --rehearse-sp $sp --lanes 1; run --rehearse-sp $sp --lanes 2; done
