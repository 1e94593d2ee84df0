#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py (kernel trace + stats, then HBM counters in their own passes).
# usage: scripts/profile_round.sh <tag>      (writes gpurun_out/prof_<tag>/...)
# The traced command is the driver's (`--steps 20 --warmup 5`: 5 120 images per GPU, batch 256) minus the fresh-process
# run behind value_cold (--no-cold): a child process of a profiled program would be profiled as well, and the box does
# not allow a program that has initialised the GPU (the profiler's preload has) to start another one that way.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
make -C $R/oracle -s                     # the cpu_baseline leg must not compile under the profiler
cd /tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cold"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_line.json 2> $OUT/trace_err.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-recon --no-file-input > $OUT/bench_line_pmc.json 2> $OUT/pmc_fetch_err.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-recon --no-file-input > /dev/null 2> $OUT/pmc_write_err.log
python3 $R/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the raw traces are large: keep stats + counter rows of our kernels only
python3 - <<PYEOF
import csv, glob, os
out = "$OUT"
for tag in ("pmc_fetch", "pmc_write"):
    rows = []
    for path in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            rd = csv.DictReader(fh)
            rows += [r for r in rd if "fq::" in r.get("Kernel_Name", "")]
            fields = rd.fieldnames
    if rows:
        with open(os.path.join(out, tag + "_fq_kernels.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=fields)
            w.writeheader()
            w.writerows(rows)
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    os.replace(path, os.path.join(out, "bench_kernel_stats.csv"))
PYEOF
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
