#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("# rocprofv3 summary for", out)
for f in find("trace/**/*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, out))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")[:90]
            print("%-90s calls=%s total_ns=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
                name, row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("MinNs"), row.get("MaxNs"), row.get("Percentage")))
for f in find("trace/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    if rows:
        r = [x for x in rows if "aai_" in x.get("Kernel_Name", "")]
        # per launch shape: the plan's one-off launch-shape measurement (aai_engine.cpp: tune_axis_plan) shows up as a few
        # launches of other grids; the shape with most launches is the one the timed steps ran
        shapes = defaultdict(list)
        for x in r:
            key = (x["Kernel_Name"][:70], x.get("Grid_Size_X"), x.get("Grid_Size_Y"), x.get("Grid_Size_Z"),
                   x.get("Workgroup_Size_X"), x.get("Workgroup_Size_Y"), x.get("Workgroup_Size_Z"),
                   x.get("VGPR_Count"), x.get("SGPR_Count"), x.get("LDS_Block_Size"), x.get("Scratch_Size"))
            shapes[key].append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
        print("\n## launch geometry and duration per shape (%s)" % os.path.relpath(f, out))
        for key, d in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
            print("%s grid=(%s,%s,%s) wg=(%s,%s,%s) vgpr=%s sgpr=%s lds=%s scratch=%s  launches=%d avg_ns=%.0f min_ns=%d max_ns=%d" % (
                key + (len(d), sum(d) / len(d), min(d), max(d))))
for tag in ("pmc_fetch", "pmc_write"):
    for f in find(tag + "/**/*counter_collection.csv"):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row.get("Kernel_Name", "")[:70], row.get("Counter_Name"))].append(float(row.get("Counter_Value", 0)))
        print("\n## %s (%s)" % (tag, os.path.relpath(f, out)))
        for (k, c), v in sorted(acc.items()):
            if "aai_" not in k:
                continue
            print("%-70s %-12s launches=%d mean=%.6g min=%.6g max=%.6g" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
