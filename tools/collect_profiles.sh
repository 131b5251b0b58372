#!/bin/bash
# Round artifacts, on the GPU box: tests, the bench line (default schedule and --lanes 1), micro-benchmarks, rocprofv3
# kernel stats + trace of the bench command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE in their own runs, with
# --kernel-trace only).  Results land in gpurun_out/art/; tools/pmc_summary.py and tools/trace_summary.py digest them.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/art
rm -rf $O
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
timeout -k 10 600 python bench.py 2> $O/bench.err | grep '^{' > $O/bench.json
cut -c1-300 $O/bench.json
timeout -k 10 600 python bench.py --lanes 1 --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_1lane.json
cut -c1-300 $O/bench_1lane.json
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_20steps.json
timeout -k 10 600 python bench.py --kv-dtype fp8 --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_fp8kv.json
# BASELINE configs[1]: the LSTM speculator alone (draft model every step), and the headline under the reference's indexing
timeout -k 10 600 python bench.py --no-suffix --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_nosuffix.json
timeout -k 10 600 python bench.py --proposal-indexing reference --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_reference_indexing.json
timeout -k 10 600 python tools/microbench.py attn fp8 mix mid ql lstm rej > $O/microbench.txt 2>&1
bash tools/ab_sp.sh > $O/rehearsal_sp.txt 2>&1 || true
echo microbench done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/stats -o p -- python3 $R/bench.py --no-cpu-baseline --no-replay-check > $O/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-replay-check > $O/pmc_fetch.log 2>&1
grep '^{' $O/pmc_fetch.log > $O/pmc_bench.json || true
echo fetch pass done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-replay-check > $O/pmc_write.log 2>&1
echo write pass done
cd $R
DB=$(find $O/stats -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/trace_summary.py $DB 8 2 > $O/step_timeline.txt 2>&1 || true; fi
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py $F $W $O/pmc_bench.json > $O/pmc_attention.json
cat $O/pmc_attention.json | head -50
ls $O $O/stats
# the .db is large: keep the CSVs and summaries only
find $O -name '*.db' -delete
