#!/bin/bash
# GPU box: rocprofv3 kernel stats of the ReconModel (int8-sim) forward only.
# usage: scripts/profile_recon.sh <tag> [batch] [resident]     (writes gpurun_out/prof_<tag>/...)
set -u
TAG=${1:-reconXX}
B=${2:-128}
MODE=${3:-plain}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/recon_probe.py $B 60 $MODE > $OUT/probe.txt 2> $OUT/trace_err.log
cat $OUT/probe.txt
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$S" $OUT/kernel_stats.csv
head -16 $OUT/kernel_stats.csv | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete
