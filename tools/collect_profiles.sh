#!/bin/bash
# Round artifacts, on the GPU box: tests, the bench line, micro-benchmarks, rocprofv3 kernel stats and the two PMC
# passes (FETCH_SIZE / WRITE_SIZE in their own runs, with --kernel-trace only).  Results land in gpurun_out/art/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/art
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
timeout -k 10 600 python bench.py 2> $O/bench.err | grep '^{' > $O/bench.json
cut -c1-300 $O/bench.json
timeout -k 10 300 python tools/microbench.py > $O/microbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
ls $O $O/stats $O/pmc_fetch
