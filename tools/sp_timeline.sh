#!/bin/bash
# GPU busy / idle per round of the rehearsed SP step (bench.py --rehearse-sp N --lanes L), from a rocprofv3 kernel trace.
# usage: bash tools/sp_timeline.sh <sp> <lanes> [extra bench flags]; output in gpurun_out/sp_timeline_<sp>_<lanes>.txt
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
SP=$1; L=$2; shift 2
O=$R/gpurun_out/spt_${SP}_${L}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d $O -o p -- python3 $R/bench.py --rehearse-sp $SP --lanes $L --no-cpu-baseline "$@" > $O/log 2>&1
cd $R
grep '^{' $O/log | cut -c1-200
DB=$(find $O -name '*.db' | head -1)
python tools/trace_summary.py $DB 16 $L 10 > gpurun_out/sp_timeline_${SP}_${L}.txt 2>&1
find $O -name '*.db' -delete
cat gpurun_out/sp_timeline_${SP}_${L}.txt | head -48
