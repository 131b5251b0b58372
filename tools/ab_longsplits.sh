#!/bin/bash
# same-box A/B of the long-draft split count on the bench (1 and 2 lanes)
set -e
run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['config']['lanes'])"; }
for ls in 0 3 4 6 8; do run --lanes 1 --long-splits $ls; done
for ls in 0 4 6 8 12; do run --lanes 2 --long-splits $ls; done
run --lanes 1
run --lanes 2
