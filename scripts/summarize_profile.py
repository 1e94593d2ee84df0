#!/usr/bin/env python3
"""Condense a scripts/profile_round.sh output directory: per-kernel stats of the fq:: kernels and the
HBM traffic counters (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 streaming reads)."""
import csv
import glob
import os
import sys

out = sys.argv[1]


def rows(pattern):
    for path in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


# ---- the timed region's histogram launches, one per row (profiles/rNN_hist_timed_launches.csv): bench.py's line says which
# dispatches of hist2048_seg_kernel they are (roofline.trace_slice), the kernel trace holds each dispatch's start / end
def timed_hist_launches():
    import json
    line = None
    for name in ("bench_line.json",):
        path = os.path.join(out, name)
        if os.path.isfile(path):
            cand = [ln for ln in open(path).read().split("\n") if ln.startswith("{")]
            line = json.loads(cand[-1]) if cand else None
    if not line or "trace_slice" not in line.get("roofline", {}):
        print("== no roofline.trace_slice in bench_line.json: timed launches not isolated ==")
        return
    sl, roof = line["roofline"]["trace_slice"], line["roofline"]
    disp = [r for r in rows("trace/**/*kernel_trace.csv") if sl["kernel"] in r.get("Kernel_Name", "")]
    disp.sort(key=lambda r: int(r["Start_Timestamp"]))
    picked = disp[sl["first"]:sl["first"] + sl["count"]]
    if len(picked) != sl["count"]:
        print("== kernel trace holds %d dispatches of %s, the slice wants [%d, %d) ==" % (len(disp), sl["kernel"], sl["first"],
                                                                                        sl["first"] + sl["count"]))
        return
    path = os.path.join(out, "hist_timed_launches.csv")
    durs = []
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["dispatch_index_of_this_kernel", "start_ns", "end_ns", "duration_us", "grid_size", "workgroup_size",
                    "algorithmic_bytes", "GB_per_s", "frac_of_8000_GB_per_s"])
        for i, r in enumerate(picked):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            durs.append(d)
            gbs = roof["algorithmic_bytes_per_launch"] / (d * 1e-6) / 1e9
            w.writerow([sl["first"] + i, r["Start_Timestamp"], r["End_Timestamp"], "%.3f" % d, r.get("Grid_Size", r.get("Grid_Size_X", "")),
                        r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), int(roof["algorithmic_bytes_per_launch"]), "%.1f" % gbs, "%.4f" % (gbs / 8000.0)])
    mean = sum(durs) / len(durs)
    print("== the %d timed launches of %s (dispatches %d..%d of the trace) -> %s ==" % (len(durs), sl["kernel"], sl["first"],
                                                                                      sl["first"] + len(durs) - 1, path))
    print("   mean %.1f us (min %.1f, max %.1f); %.0f algorithmic bytes per launch -> %.1f GB/s = %.4f of 8000; bench.py's HIP events in "
          "the same run: mean %.1f us -> frac %.4f" % (mean, min(durs), max(durs), roof["algorithmic_bytes_per_launch"],
                                                       roof["algorithmic_bytes_per_launch"] / (mean * 1e-6) / 1e9,
                                                       roof["algorithmic_bytes_per_launch"] / (mean * 1e-6) / 1e9 / 8000.0,
                                                       roof["mean_launch_ms"] * 1e3, roof["frac"]))


timed_hist_launches()


# ---- HBM traffic of exactly those launches: the PMC passes' own bench line (bench_line_pmc.json) says which dispatches of the
# kernel its timed region was; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md), KB -> bytes
def timed_hist_traffic():
    import json
    path = os.path.join(out, "bench_line_pmc.json")
    if not os.path.isfile(path):
        return
    cand = [ln for ln in open(path).read().split("\n") if ln.startswith("{")]
    line = json.loads(cand[-1]) if cand else None
    if not line or "trace_slice" not in line.get("roofline", {}):
        return
    sl, alg = line["roofline"]["trace_slice"], line["roofline"]["algorithmic_bytes_per_launch"]
    means = {}
    for counter, tag in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        per = {}
        for r in rows(tag + "/**/*counter_collection.csv"):
            if sl["kernel"] in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        ids = sorted(per)[sl["first"]:sl["first"] + sl["count"]]
        if len(ids) != sl["count"]:
            print("== %s: %d dispatches of %s in the counter pass, the slice wants %d ==" % (counter, len(per), sl["kernel"], sl["count"]))
            return
        means[counter] = sum(per[i] for i in ids) / len(ids)
    hbm = means["FETCH_SIZE"] * 2 * 1024 + means["WRITE_SIZE"] * 1024
    print("== HBM traffic of the %d timed launches of %s (separate --pmc passes): FETCH_SIZE %.1f KB x 2 + WRITE_SIZE %.1f KB = %.0f bytes "
          "per launch against %.0f algorithmic: ratio %.4f ==" % (sl["count"], sl["kernel"], means["FETCH_SIZE"], means["WRITE_SIZE"], hbm, alg, hbm / alg))


timed_hist_traffic()

print("== kernel stats (fq:: kernels and top 8 overall) ==")
stats = list(rows("trace/**/*kernel_stats.csv"))
for i, r in enumerate(stats):
    if "fq::" in r["Name"] or i < 8:
        print("%-90s calls %6s avg %12.1f us  %6s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))

for counter, tag in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = {}
    for r in rows(tag + "/**/*counter_collection.csv"):
        name = r.get("Kernel_Name", "")
        if "fq::" not in name or r.get("Counter_Name") != counter:
            continue
        acc.setdefault(name, []).append(float(r["Counter_Value"]))
    print("== %s per launch (KB as reported) ==" % counter)
    for name, vals in acc.items():
        if "hist2048_seg" in name or "hist2048_chan" in name:
            # the per-tensor kernel's launches of the warm-up and of the un-cached extra run only see the image (nothing is
            # cached there); the timed region's launches are the big ones: report those separately
            tail = [v for v in vals if v > 0.5 * max(vals)] if "hist2048_seg" in name else vals
            print("%-90s timed launches %4d mean %14.1f KB  -> %.3f GB%s" %
                  ((name[:60] + " [timed region]"), len(tail), sum(tail) / len(tail),
                   sum(tail) / len(tail) * 1024 / 1e9 * (2 if counter == "FETCH_SIZE" else 1),
                   " (x2 gfx950 correction)" if counter == "FETCH_SIZE" else ""))
        mean = sum(vals) / len(vals)
        print("%-90s launches %4d mean %14.1f KB  -> %.3f GB%s" %
              (name[:90], len(vals), mean, mean * 1024 / 1e9 * (2 if counter == "FETCH_SIZE" else 1),
               " (x2 gfx950 correction)" if counter == "FETCH_SIZE" else ""))
