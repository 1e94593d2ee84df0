#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py (kernel trace + stats, then HBM counters in their own passes).
# usage: scripts/profile_round.sh <tag>      (writes gpurun_out/prof_<tag>/...)
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the stats pass runs the DEFAULT bench command (what the driver runs), so that the per-kernel averages are the ones of the judged line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py > $OUT/bench_line.json 2> $OUT/trace_err.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recon > /dev/null 2> $OUT/pmc_fetch_err.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recon > /dev/null 2> $OUT/pmc_write_err.log
python3 $R/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the raw traces are large: keep stats + counter rows of our kernels only
find $OUT -name "*kernel_trace.csv" -delete
