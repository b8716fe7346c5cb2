#!/bin/bash
# The rotated workloads of the round's kernel work on one box (kernel-only): tools/rot_matrix.sh <tag> [ENV=VAL ...]
TAG=${1:-r04}; shift
OUT=gpurun_out/rot_$TAG.jsonl
: > $OUT
for kv in "$@"; do export "$kv"; done
for w in cfg3 cfg3x8 cfg5 cfg3fast cfg5fast wide8 wide8fast refdefault; do
  b=1; s=5
  case $w in cfg5*) s=3;; cfg3x8) b=8; w=cfg3;; esac
  timeout -k 10 240 python bench.py --workload $w --no-cpu-baseline --traffic off --configs off --steps $s --warmup 1 --batch $b --min-seconds 1.0 >> $OUT 2>> gpurun_out/rot_$TAG.err || echo "{\"failed\": \"$w\"}" >> $OUT
done
python - <<'PY' $OUT
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "failed" in d: print("FAILED", d["failed"]); continue
    r = d["roofline"]
    print("%-44s x%-2d %9.0f Mpix/s %8.1f us/launch %6.0f GB/s (%.1f%%) %s" % (d["config"]["workload"][:44], d["config"]["images_per_gpu_per_step"], d["value"], r["kernel_ms_per_launch"]*1e3, r["achieved"], 100*r["frac"], r["kernel"]))
PY
