#!/bin/bash
# GPU box: rocprofv3 kernel stats of the per-channel calibration (scripts/per_channel_probe.py K B), with and without the owner flush.
# usage: scripts/per_channel_profile.sh <outdir> [K=4] [B=256]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p $1 && cd $1 && pwd)
K=${2:-4}; B=${3:-256}
export TMPDIR=/tmp
for own in 1 0; do
  export FQ_CHAN_OWN_FLUSH=$own
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/own$own -o t -- python3 $R/scripts/per_channel_probe.py $K $B > $OUT/probe_own$own.txt 2> $OUT/err_own$own.txt
  python3 - $OUT/own$own > $OUT/kernels_own$own.txt <<'PYEOF'
import csv, glob, os, sys
d = sys.argv[1]
for p in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(p, newline="")))
    for r in rows[:25]:
        print("%-90s calls %6s  total %10.3f ms  avg %9.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
# the individual launches of the histogram kernel
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(p, newline="")) if "hist2048_chan" in r["Kernel_Name"] or "absmax_chan" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows[-8:]:
        print("%-40s grid %8s  %9.1f us" % (r["Kernel_Name"][:40], r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PYEOF
  find $OUT/own$own -name "*.csv" -delete
  cat $OUT/probe_own$own.txt
done
