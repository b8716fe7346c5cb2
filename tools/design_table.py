#!/usr/bin/env python3
"""DESIGN.md section 7's table from a default bench.py line: python tools/design_table.py profiles/rNN_bench_default.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = {e["name"]: e for e in d["configs"]}
rf = d["roofline"]
rows = ["| headline: config 2 x 4 images | `aai_axis_kernel` | %.4f ms | %s | **%.3f**; traffic %s B = %.3fx algorithmic, measured by the run | %.4f |"
        % (d["ms_per_step"], format(int(round(d["value"])), ","), rf["frac"], format(int(rf["traffic"]), ","), rf["traffic"] / rf["algorithmic_bytes_per_launch"], d["cpu_baseline"]["value"])]
for name in ("cfg1", "cfg2", "cfg3", "cfg3fast", "cfg4", "cfg5", "cfg5fast", "cfg5bilinear", "cfg5bicubic", "refdefault", "refdefaultfast", "wide8", "wide8fast"):
    e = c[name]
    ms = e["ms"]
    t = ("%.1f us" % (ms * 1e3)) if ms < 0.1 else ("%.3f ms" % ms)
    if "graph_ms" in e and not isinstance(e["graph_ms"], str):
        t += " (%.1f us replayed from a graph)" % (e["graph_ms"] * 1e3)
    cpu = e.get("cpu")
    rows.append("| %s x %d | `%s` | %s | %s | %.3f | %s |" % (name, e["images"], e["kernel"].split("<")[0], t, format(int(round(e["mpix_s"])), ","), e["frac"], ("%.3g" % cpu[0]) if cpu else "-"))
print("| config | kernel | per launch | Mpix/s | HBM frac | reference CPU Mpix/s, same host |\n|---|---|---|---|---|---|")
print("\n".join(rows))
