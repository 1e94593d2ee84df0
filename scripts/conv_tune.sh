#!/bin/bash
# GPU box: scripts/conv_bench.py under every combination of the dispatch knobs of fq_conv2d_i8 (one process each: the
# knobs are read once), to check the per-layer heuristics of conv2d_i8_dispatch at a given batch.
# usage: scripts/conv_tune.sh <batch> <out dir>
B=${1:-256}
OUT=${2:-gpurun_out/conv_tune}
mkdir -p $OUT
python scripts/conv_bench.py $B i8 > $OUT/default.txt 2>/dev/null
for tk in 64 128; do for dma in 0 1; do for st in 2 3; do
  [ $dma = 0 ] && [ $st = 3 ] && continue
  FQ_CONV_TK=$tk FQ_CONV_DMA=$dma FQ_CONV_STAGES=$st python scripts/conv_bench.py $B i8 > $OUT/tk${tk}_dma${dma}_st${st}.txt 2>/dev/null
done; done; done
python - <<PY
import glob, os
rows = {}
for path in sorted(glob.glob("$OUT/*.txt")):
    tag = os.path.basename(path)[:-4]
    for line in open(path):
        parts = line.split()
        if len(parts) > 3 and "," in parts[0] and parts[0][0].isdigit():
            rows.setdefault(parts[0], {})[tag] = float(parts[1])
print("%-22s %8s  best (config)                others" % ("layer", "default"))
tot_d = tot_b = 0.0
cnt = {l.split()[0]: int(l.split()[-1][1:]) for l in open("$OUT/default.txt") if "," in l.split()[0:1][0:1][0] if l.split()[-1].startswith("x")} if False else {}
for line in open("$OUT/default.txt"):
    p = line.split()
    if len(p) > 3 and "," in p[0] and p[0][0].isdigit():
        cnt[p[0]] = int(p[-1][1:])
for layer, d in rows.items():
    best = min(d, key=d.get)
    print("%-22s %8.1f  %8.1f (%s)  %s" % (layer, d["default"], d[best], best, " ".join("%s=%.1f" % (k, v) for k, v in sorted(d.items()) if k != "default")))
    tot_d += d["default"] * cnt[layer]; tot_b += d[best] * cnt[layer]
print("whole net: default %.1f us, best-per-layer %.1f us" % (tot_d, tot_b))
PY
