#!/bin/bash
# which part of bench.py launches MIOpen's naive convolution?  (scratch)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/naive
mkdir -p $OUT
for V in 0 1; do
  export FQ_OWN_CONV1X1=$V
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/own$V -- python3 $R/bench.py --steps 4 --warmup 2 --images 512 --no-cold --no-cpu-baseline --no-recon --no-per-channel > $OUT/own$V.json 2> $OUT/own$V.err
  echo "== FQ_OWN_CONV1X1=$V"; grep -h -i "naive\|grouped_conv" $(find $OUT/own$V -name "*kernel_stats.csv") | cut -c1-60,200-260
  python3 - $OUT/own$V <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "naive_conv" in r["Kernel_Name"]]
print("naive launches:", len(idx))
for i in idx[:3] + idx[-2:]:
    print("  ...", [rows[j]["Kernel_Name"][:50] for j in range(max(0, i - 3), min(len(rows), i + 3))], rows[i].get("Grid_Size_X"), rows[i].get("Workgroup_Size_X"))
PY
  find $OUT/own$V -name "*.csv" -delete
done
