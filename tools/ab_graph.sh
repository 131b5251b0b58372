#!/bin/bash
# same-box A/B of the attention-layers graph launch: AIC_ATTN_GRAPH_MODE 0 = kernel by kernel, 1 = head (4 layers) + rest,
# 2 = one graph; rehearsed SP 8 (one lane: the host chain is serial with the GPU) and the default bench
set -e
run() { echo "== mode $M $*"; AIC_ATTN_GRAPH_MODE=$M timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['config'].get('attn_graph'), d['config']['lanes'])"; }
for rep in 1 2 3; do for M in 0 1 2; do run --rehearse-sp 8 --lanes 1; done; done
for rep in 1 2; do for M in 0 1 2; do run; done; done
