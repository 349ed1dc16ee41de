#!/bin/bash
# Evidence for bench.py's line, collected in ONE gpurun call from the same tree:
#   1. the plain default bench (JSON line + per-kernel stage report)
#   2. rocprofv3 --kernel-trace --stats of the same command (per-kernel durations)
#   3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (they do not fit one pass on gfx950)
# then tools/pmc_summarise.py folds 3 into one JSON.  Copy what is to be judged from gpurun_out/$TAG into profiles/.
# BENCH_ARGS (e.g. "--c-isdf 12") goes to every bench invocation: the PMC passes of the default c = 18 headline did not finish inside
# gpurun's 7-minute silence limit (round 3); the c = 12 variant's do (about a minute each).
set -o pipefail
TAG=${1:-r03_prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
if [ -z "$PMC_ONLY" ]; then
python3 bench.py $BENCH_ARGS --steps 2 --warmup 1 --stage-report $OUT/stage_report.txt > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done: $(cut -c1-160 $OUT/bench.json)"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py $BENCH_ARGS --steps 1 --warmup 1 --no-cpu-baseline --no-accuracy > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
echo "stats done"
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f --output-format csv -- python3 bench.py $BENCH_ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-accuracy > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w --output-format csv -- python3 bench.py $BENCH_ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-accuracy > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err || exit 1
echo "pmc write done"
python3 tools/pmc_summarise.py $OUT/pmc_summary.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt
# keep the merged-back volume small: the raw traces are large
head -60 $OUT/stats/*kernel_stats.csv > $OUT/kernel_stats_head.csv 2>/dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats/*kernel_trace.csv
ls -la $OUT
