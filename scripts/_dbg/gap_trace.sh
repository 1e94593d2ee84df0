#!/bin/bash
# GPU box: GPU idle gaps inside one warm calibration (the driver's workload): rocprofv3 kernel trace of a process that calibrates
# four times; the last calibration's kernels: wall, busy sum, and the gaps above 150 us with the kernels on either side.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$(mkdir -p $1 && cd $1 && pwd)
export TMPDIR=/tmp
cat > /tmp/gap_job.py <<PY
import os, sys, time, torch
ROOT = "$R"
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"), os.path.join(ROOT, "tests")]
import bench
from tools import Quantity
dev = torch.device("cuda")
sys.stdout = open(os.devnull, "w")
model = bench.build_model("r50", 224, dev)
bench.make_workdir(19, "1,3,224,224", 0)
data = bench.DeviceBatches(20, 256, 224, 0, 1, dev)
# the warm allocator pool of the driver command (bench.py grows it to 80 % of HBM before the timed call)
free_b, total_b = torch.cuda.mem_get_info(dev)
pool = torch.empty(int(total_b * 0.80) - torch.cuda.memory_allocated(), dtype=torch.uint8, device=dev); del pool
for rep in range(4):
    q = Quantity(model)
    q.profile_phases = True
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q.activation_quantize(data)
    torch.cuda.synchronize()
    sys.stderr.write("calibration %d: %.4f s  pass1 %.4f pass2 %.4f cache plan %s, %.1f GB cached, pool %.1f GB\n" % (
        rep, time.perf_counter() - t0, q.timings.get("pass1_s", 0), q.timings.get("pass2_s", 0), q.timings.get("cache_plan"),
        q.timings.get("cache_bytes", 0) / 2.0 ** 30, torch.cuda.memory_reserved() / 2.0 ** 30))
    torch.zeros(1, device=dev).add_(1)          # marker kernel between calibrations
    torch.cuda.synchronize(); time.sleep(0.05)
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 /tmp/gap_job.py 2> $OUT/err.txt
grep calibration $OUT/err.txt
python3 - $OUT/tr <<'PYEOF' | tee $OUT/gaps.txt
import csv, glob, os, sys
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(p, newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split at gaps > 30 ms (the sleeps between calibrations)
segs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 30e6:
        segs.append(cur); cur = []
    cur.append(b)
segs.append(cur)
seg = max(segs[-3:], key=len) if len(segs) >= 3 else segs[-1]
seg = segs[-1] if len(segs[-1]) > 1000 else seg
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("last calibration: %d kernels, wall %.1f ms, busy %.1f ms (%.1f %%)" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
gaps = []
end = int(seg[0]["End_Timestamp"])
for a, b in zip(seg, seg[1:]):
    end = max(end, int(a["End_Timestamp"]))
    g = int(b["Start_Timestamp"]) - end
    if g > 150e3:
        gaps.append((g / 1e3, (int(b["Start_Timestamp"]) - t0) / 1e6, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
print("gaps > 150 us: %d, total %.1f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e3))
for g in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f us at %7.1f ms   after %-50s before %s" % g)
small = 0.0
end = int(seg[0]["End_Timestamp"])
for a, b in zip(seg, seg[1:]):
    end = max(end, int(a["End_Timestamp"]))
    g = int(b["Start_Timestamp"]) - end
    if 0 < g <= 150e3: small += g
print("sum of the gaps <= 150 us: %.1f ms" % (small / 1e6))
# where the busy time goes, pass by pass (pass 2 starts at the first histogram launch)
import collections, re
first_hist = next((i for i, r in enumerate(seg) if "hist" in r["Kernel_Name"]), len(seg))
for label, part in (("pass 1", seg[:first_hist]), ("pass 2 + KL", seg[first_hist:])):
    if not part:
        continue
    tot = collections.defaultdict(lambda: [0, 0])
    for r in part:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0][:70]
        tot[name][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot[name][1] += 1
    w0, w1 = int(part[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in part)
    print("%s: %d kernels, wall %.1f ms, busy %.1f ms" % (label, len(part), (w1 - w0) / 1e6, sum(v[0] for v in tot.values()) / 1e6))
    for name, (ns, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:22]:
        print("  %8.2f ms %5d x %8.1f us  %s" % (ns / 1e6, n, ns / 1e3 / n, name))
PYEOF
find $OUT/tr -name "*.csv" -delete
