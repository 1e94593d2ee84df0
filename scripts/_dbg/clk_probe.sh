#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/clk
mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/p -- python3 $R/scripts/conv1x1_one.py 2048 2048 14 1 256 none 5 > $OUT/run.txt 2> $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv1x1" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = {}
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv1x1" in r["Kernel_Name"]:
            cnt.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(dur, key=int):
    c = cnt.get(d, {})
    g = c.get("GRBM_GUI_ACTIVE", 0)
    print("dispatch %s: %.1f us, GRBM_GUI_ACTIVE %.0f (/8 = %.0f) -> %.3f GHz; MFMA busy %.0f -> util %.3f" % (
        d, dur[d] / 1e3, g, g / 8, g / 8 / dur[d], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g / 8 * 1024 + 1)))
PY
