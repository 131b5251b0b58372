set -x
python tools/debug_capture.py > gpurun_out/r3_debug_capture.log 2>&1; echo "debug rc=$?" >> gpurun_out/r3_debug_capture.log
python bench.py > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err
python bench.py --no-suffix --no-cpu-baseline --no-replay-check > gpurun_out/r3_bench_nosuffix_a.json 2> gpurun_out/r3_bench_nosuffix_a.err
python bench.py --rehearse-sp 8 --no-cpu-baseline --no-replay-check > gpurun_out/r3_bench_sp8_a.json 2> gpurun_out/r3_bench_sp8_a.err
tail -c 600 gpurun_out/r3_bench_a.json
