#!/bin/bash
# GPU box: HBM-side bytes of every kernel of a resident ReconModel forward at 256 images -- rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
# separate passes (FETCH_SIZE doubled: gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md) -- beside the kernel's duration.
# usage: scripts/int8_traffic_by_kernel.sh <outdir> [batch=256]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p $1 && cd $1 && pwd)
B=${2:-256}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o t -- python3 $R/scripts/int8_layer_table.py $B 5 > /dev/null 2> $OUT/err_$c.txt
done
python3 - $OUT > $OUT/traffic_b$B.txt <<'PYEOF'
import csv, glob, os, sys, statistics
out = sys.argv[1]
def load(c):
    rows = []
    for p in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(p, newline="")) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
def forwards(rows):
    names = [r["Kernel_Name"] for r in rows]
    stems = [i for i, n in enumerate(names) if "stem_conv_i8" in n]
    last = stems[-5:]
    return [rows[a:a + 52] for a in last]
F, W = forwards(fe), forwards(wr)
print("%3s %-52s %9s %9s %9s %8s" % ("#", "kernel", "read MB", "write MB", "us", "TB/s"))
tr = tw = tt = 0.0
for j in range(52):
    name = F[0][j]["Kernel_Name"].split("(")[0][-52:]
    rd = statistics.median(float(f[j]["Counter_Value"]) for f in F) * 1024 * 2 / 1e6
    wt = statistics.median(float(w[j]["Counter_Value"]) for w in W) * 1024 / 1e6
    us = statistics.median((int(f[j]["End_Timestamp"]) - int(f[j]["Start_Timestamp"])) / 1e3 for f in F)
    tr += rd; tw += wt; tt += us
    print("%3d %-52s %9.1f %9.1f %9.1f %8.2f" % (j, name, rd, wt, us, (rd + wt) / us))
print("one forward: read %.0f MB, written %.0f MB, %.0f us under the profiler: %.2f TB/s" % (tr, tw, tt, (tr + tw) / tt))
PYEOF
find $OUT -name "*.csv" -delete
tail -1 $OUT/traffic_b$B.txt
