python -m pytest tests -m gpu -x -q > gpurun_out/r3_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t4.log; tail -4 gpurun_out/r3_t4.log
python bench.py --rehearse-sp 8 --no-cpu-baseline --no-replay-check > gpurun_out/r3_bench_sp8_b.json 2> gpurun_out/r3_bench_sp8_b.err
AIC_ENGINE_NUMPY=1 AIC_SUFFIX_STAGED=1 python bench.py --rehearse-sp 8 --no-cpu-baseline --no-replay-check > gpurun_out/r3_bench_sp8_b_old.json 2> gpurun_out/r3_bench_sp8_b_old.err
python bench.py --no-cpu-baseline > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lstm -o lstm -- python3 $GRAFT_REPO_ROOT/tools/microbench.py lstm > $GRAFT_REPO_ROOT/gpurun_out/r3_mb_lstm_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_lstm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_lstm_kernel_stats.csv
cut -c1-160 gpurun_out/r3_lstm_kernel_stats.csv | head -24
