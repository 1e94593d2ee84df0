#!/bin/bash
# GPU box: HBM-side bytes of every kernel of one pass-1 forward of the calibration (scripts/float_forward_table.py) at 256 images:
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (FETCH doubled: MI355X_MICROARCH.md), beside the kernel's duration.
# usage: scripts/float_traffic_by_kernel.sh <outdir> [batch=256]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p $1 && cd $1 && pwd)
B=${2:-256}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o t -- python3 $R/scripts/float_forward_table.py $B > /dev/null 2> $OUT/err_$c.txt
done
python3 - $OUT > $OUT/float_traffic_b$B.txt <<'PYEOF'
import csv, glob, os, sys
out = sys.argv[1]
def load(c):
    rows = []
    for p in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(p, newline="")) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
def last_forward(rows):
    stems = [i for i, r in enumerate(rows) if "conv_stem_f32" in r["Kernel_Name"]]
    a = stems[-1]
    return rows[a:]
F, W = last_forward(fe), last_forward(wr)
n = min(len(F), len(W))
print("%3s %-64s %9s %9s %9s %7s" % ("#", "kernel (the last forward of the run)", "read MB", "write MB", "us", "TB/s"))
tr = tw = tt = 0.0
for j in range(n):
    name = F[j]["Kernel_Name"].split("(")[0][-64:]
    rd = float(F[j]["Counter_Value"]) * 1024 * 2 / 1e6
    wt = float(W[j]["Counter_Value"]) * 1024 / 1e6
    us = (int(F[j]["End_Timestamp"]) - int(F[j]["Start_Timestamp"])) / 1e3
    tr += rd; tw += wt; tt += us
    print("%3d %-64s %9.1f %9.1f %9.1f %7.2f" % (j, name, rd, wt, us, (rd + wt) / us))
print("one forward: read %.0f MB, written %.0f MB, %.0f us under the profiler: %.2f TB/s" % (tr, tw, tt, (tr + tw) / tt))
PYEOF
find $OUT -name "*.csv" -delete
tail -1 $OUT/float_traffic_b$B.txt
