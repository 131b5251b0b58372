#!/bin/bash
# SQ counters of the attention bodies over `tools/microbench.py pmc` (three separate --pmc passes with --kernel-trace only, as the
# guide prescribes), digested by tools/pmc_kernels.py into gpurun_out/art/pmc_kernels.txt.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/art/pmck
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o p -- python3 $R/tools/microbench.py pmc > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/b -o p -- python3 $R/tools/microbench.py pmc > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/c -o p -- python3 $R/tools/microbench.py pmc > $O/c.log 2>&1
cd $R
python tools/pmc_kernels.py $(find $O -name '*counter_collection.csv') > $R/gpurun_out/art/pmc_kernels.txt
cat $R/gpurun_out/art/pmc_kernels.txt
find $O -name '*kernel_trace.csv' -delete
