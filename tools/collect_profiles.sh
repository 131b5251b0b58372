#!/bin/bash
# Round artifacts, on the GPU box, in stages that each fit one gpurun call (20 minutes):
#   stage bench   : the bench line in its configurations (default, --lanes 1, the driver's --steps 20, fp8 cache, configs[1]
#                   --no-suffix, the reference's proposal indexing, speculation off), micro-benchmarks, SP rehearsal
#   stage profile : rocprofv3 kernel stats + trace of the bench command, the two PMC passes (FETCH_SIZE / WRITE_SIZE in their
#                   own runs, with --kernel-trace only), kernel stats of the LSTM-only configuration
#   stage tests   : pytest -m gpu
# Results land in gpurun_out/art/; tools/pmc_summary.py and tools/trace_summary.py digest them.
#   bash tools/collect_profiles.sh bench|profile|tests
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/art
mkdir -p $O
cd $R
STAGE=${1:-bench}
Q="--no-cpu-baseline --no-replay-check --no-lstm-leg"

if [ "$STAGE" = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 || true
  tail -3 $O/pytest_gpu.txt
fi

if [ "$STAGE" = bench ]; then
  timeout -k 10 600 python bench.py 2> $O/bench.err | grep '^{' > $O/bench.json
  cut -c1-300 $O/bench.json
  timeout -k 10 300 python bench.py --lanes 1 $Q 2>> $O/bench.err | grep '^{' > $O/bench_1lane.json
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 2>> $O/bench.err | grep '^{' > $O/bench_20steps.json
  timeout -k 10 300 python bench.py --kv-dtype fp8 $Q 2>> $O/bench.err | grep '^{' > $O/bench_fp8kv.json
  # speculation against plain decode on the same engine, same box, one after the other (the stand-in for the north star's
  # ">= 2x vanilla": vLLM is on neither box): configs[1] (LSTM alone, drafts planted at p = 0.7), default, speculation off
  timeout -k 10 300 python bench.py --no-suffix --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_nosuffix.json
  timeout -k 10 300 python bench.py --no-spec --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_nospec.json
  timeout -k 10 300 python bench.py --no-spec --lanes 1 --no-cpu-baseline --no-replay-check 2>> $O/bench.err | grep '^{' > $O/bench_nospec_1lane.json
  timeout -k 10 300 python bench.py --proposal-indexing reference $Q 2>> $O/bench.err | grep '^{' > $O/bench_reference_indexing.json
  python - <<'EOF' > $O/bench_spec_vs_nospec.txt
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "art")
rows = [("default (arctic LSTM k=3 + suffix decoding, library-default indexing)", "bench.json"),
        ("configs[1]: LSTM alone (--no-suffix), drafts planted at p=0.7 per position", "bench_nosuffix.json"),
        ("speculation off (--no-spec), two lanes", "bench_nospec.json"),
        ("speculation off (--no-spec --lanes 1)", "bench_nospec_1lane.json"),
        ("reference plugin's literal indexing (--proposal-indexing reference)", "bench_reference_indexing.json")]
base = None
print("# same box, one run after the other (tools/collect_profiles.sh bench); hot path only, target dense layers synthetic")
for name, f in rows:
    try:
        d = json.load(open(os.path.join(O, f)))
    except Exception as e:
        print(f"{name}: missing ({e})")
        continue
    if "nospec.json" in f:
        base = d["value"]
    print(f"{name}:\n    {d['value']:9.0f} tokens/s  {d['ms_per_step']:.3f} ms/round  tokens/request-step {d['tokens_per_request_step']:.3f}  "
          f"accepted/draft {d['mean_accepted_draft_len']:.2f}  attention frac {d['roofline']['frac']:.3f}")
    leg = d.get("lstm_only_leg")
    if leg:
        print(f"    + LSTM-only leg of the same run: {leg['tokens_per_s']:.0f} tokens/s, {leg['ms_per_step']:.3f} ms/round, "
              f"accepted/draft {leg['mean_accepted_draft_len']:.2f} (expected {leg['expected_accepted_draft_len']:.2f}), "
              f"draft call {leg['roofline_draft_model']['avg_call_us']:.1f} us = {leg['roofline_draft_model']['frac']:.3f}")
if base:
    for name, f in rows[:2]:
        try:
            d = json.load(open(os.path.join(O, f)))
            print(f"speed-up over speculation off (two lanes): {name.split(':')[0].split(' (')[0]}: {d['value'] / base:.2f}x")
        except Exception:
            pass
EOF
  cat $O/bench_spec_vs_nospec.txt
  timeout -k 10 600 python tools/microbench.py attn fp8 mix mid ql lstm rej ctxsweep > $O/microbench.txt 2>&1 || true
  timeout -k 10 300 python tools/microbench.py longctx manylong > $O/microbench_long.txt 2>&1 || true
  bash tools/ab_sp.sh > $O/rehearsal_sp.txt 2>&1 || true
  echo bench stage done
fi

if [ "$STAGE" = profile ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/stats -o p -- python3 $R/bench.py $Q > $O/stats.log 2>&1
  echo stats done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nosuffix -o p -- python3 $R/bench.py --no-suffix --no-cpu-baseline --no-replay-check > $O/stats_nosuffix.log 2>&1
  echo nosuffix stats done
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 4 --warmup 2 $Q > $O/pmc_fetch.log 2>&1
  grep '^{' $O/pmc_fetch.log > $O/pmc_bench.json || true
  echo fetch pass done
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 4 --warmup 2 $Q > $O/pmc_write.log 2>&1
  echo write pass done
  cd $R
  DB=$(find $O/stats -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/trace_summary.py $DB 8 2 > $O/step_timeline.txt 2>&1 || true; fi
  F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py $F $W $O/pmc_bench.json > $O/pmc_attention.json
  head -50 $O/pmc_attention.json
  # the .db and the kernel traces are large: keep the stats CSVs, the counter CSVs and the summaries
  find $O -name '*.db' -delete
  find $O -name '*kernel_trace.csv' -delete
  ls $O $O/stats
fi
