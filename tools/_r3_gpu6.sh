python tools/debug_capture.py > gpurun_out/r3_debug_capture3.log 2>&1; echo "rc=$?" >> gpurun_out/r3_debug_capture3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_lstm -o lstm -- python3 $GRAFT_REPO_ROOT/tools/microbench.py lstm > $GRAFT_REPO_ROOT/gpurun_out/r3_mb_lstm_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_lstm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_lstm_kernel_stats.csv
head -30 gpurun_out/r3_lstm_kernel_stats.csv
