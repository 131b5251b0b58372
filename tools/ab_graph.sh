#!/bin/bash
# A/B of the attention-layers graph launch (bench.py --no-attn-graph): default bench, rehearsed SP 8 with 1 and 2 lanes.
set -e
mkdir -p gpurun_out
run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('attn_graph'), d['config']['lanes'])"; }
run
run --no-attn-graph
run --rehearse-sp 8 --lanes 1
run --rehearse-sp 8 --lanes 1 --no-attn-graph
run --rehearse-sp 8 --lanes 2
run --rehearse-sp 8 --lanes 2 --no-attn-graph
