#!/bin/bash
# GPU box: SQ counters of conv2d_i8_kernel on one layer shape.  usage: profile_conv_pmc.sh <tag> C H K R stride pad [i8|f32]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"
P4="TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum"
P5="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
P6="TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
P7="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
P8="TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
P9="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
P10="TA_BUFFER_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"
P11="TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7" "$P8" "$P9" "$P10" "$P11"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/scripts/conv_one.py "$@" 5 > $OUT/run$i.txt 2> $OUT/err$i.log
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv2d_i8" not in r["Kernel_Name"] and "conv3x3_i8" not in r["Kernel_Name"] and "conv1x1_i8" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-28s %16.0f per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
PY
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
