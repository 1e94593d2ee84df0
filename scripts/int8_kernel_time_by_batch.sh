#!/bin/bash
# GPU box: device time of the integer kernels of one resident ReconModel forward by batch size, from rocprofv3's kernel trace
# (HIP events around eager launches are host bound below ~128 images: scripts/int8_layer_table.py's rows are then launch gaps).
# usage: scripts/int8_kernel_time_by_batch.sh <outdir> [batches...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p $1 && cd $1 && pwd); shift      # absolute: the traced script changes its directory, rocprofv3 writes at exit
BATCHES=${@:-"32 64 128 256"}
export TMPDIR=/tmp
mkdir -p $OUT
for b in $BATCHES; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/b$b -o t -- python3 $R/scripts/int8_layer_table.py $b 7 > $OUT/table_b$b.txt 2> $OUT/err_b$b.txt
  python3 - $OUT/b$b $b > $OUT/kernels_b$b.txt <<'PYEOF'
import csv, glob, os, sys, collections
d, b = sys.argv[1], int(sys.argv[2])
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(p, newline="") as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 7 forwards are the timed ones of the script; a forward starts at the stem kernel
names = [r["Kernel_Name"] for r in rows]
stems = [i for i, n in enumerate(names) if "stem_conv_i8" in n]
last = stems[-7:]
per = collections.OrderedDict()
tot = []
for a, e in zip(last, last[1:] + [len(rows)]):
    seq = rows[a:e]
    # cut the tail of the last forward at the first kernel that is not part of a forward (none expected)
    t = 0.0
    for j, r in enumerate(seq[:80]):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (j, r["Kernel_Name"].split("(")[0][-70:])
        per.setdefault(key, []).append(us)
        t += us
    tot.append(t)
import statistics
print("batch %d: kernels per forward %d, device time per forward (median of %d) %.1f us" % (b, len(per), len(tot), statistics.median(tot)))
for (j, n), v in per.items():
    print("%3d %9.1f  %s" % (j, statistics.median(v), n))
PYEOF
  find $OUT/b$b -name "*.csv" -delete
  head -1 $OUT/kernels_b$b.txt
done
