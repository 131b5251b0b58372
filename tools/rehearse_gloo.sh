#!/bin/bash
# Multi-process rehearsals of bench.py on a one-GPU box: the ranks share the card, collectives go over gloo staged through
# the host (arcticinference_amd/dist_utils.py).  Not a multi-GPU measurement: they show that the N > 1 control flow
# (group set-up, vocab-parallel draft head, shift mode, the extra all-to-all steps) runs to completion in lockstep.
# (The box lets at most 6 processes touch the card: 2 and 4 ranks.)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() {
  n=$1; shift
  echo "== $n ranks: $*"
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
    bench.py --gpus $n --dist-backend gloo --steps 6 --warmup 2 --no-cpu-baseline --no-replay-check "$@" 2> gpurun_out/rehearse_gloo_$n.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['n_gpus'], round(d['value']), round(d['ms_per_step'],2), d['config']['parallelism'], 'shift', d['steps_in_shift_mode'], 'sp', d['steps_in_sp_mode'],
      'a2a ms', d['ulysses_all_to_all_path_ms_per_step'], d['ulysses_all_to_all_path_error'], 'acc/req-step', round(d['accepted_per_request_step'],3),
      'other', (d.get('other_indexing_mode') or {}).get('accepted_per_request_step'))"
  echo "rc=$?"
}
run 2
run 4 --no-shift-parallel
