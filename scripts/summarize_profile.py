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
