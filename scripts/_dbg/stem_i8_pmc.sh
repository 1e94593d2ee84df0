#!/bin/bash
# GPU box: SQ counters of the int8 stem kernel alone (scripts/_dbg/stem_i8_probe.py): what its waves do with their cycles
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$1
mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/scripts/_dbg/stem_i8_probe.py > $OUT/run$i.txt 2> $OUT/err$i.log
  i=$((i+1))
done
python3 - "$OUT" <<'PY' | tee $OUT/counters.txt
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem_conv_i8" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-28s %16.0f per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
PY
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
