#!/bin/bash
# Runs bench.py over every BASELINE workload (kernel-only, no CPU baseline) and a few K1 generality cases.
OUT=gpurun_out/matrix_${1:-r01}.jsonl
: > $OUT
for w in cfg1 cfg1x1024 cfg2 cfg3 cfg3fast cfg4 cfg5s cfg5 cfg5fast cfg5bilinear cfg5bicubic refdefault refdefaultfast wide8 wide8fast; do
  b=4; s=10
  case $w in cfg5*) b=1; s=3;; wide8*|refdefault*) b=1; s=5;; cfg3*) b=2; s=5;; cfg4) b=64;; cfg1x1024) b=1024; w=cfg1;; esac      # cfg4 is BASELINE's batch of 64 images; cfg1 also as 1024 images per launch
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --traffic off --configs off --steps $s --warmup 1 --batch $b --min-seconds 0.5 >> $OUT 2>> gpurun_out/matrix.err || echo "{\"failed\": \"$w\"}" >> $OUT
done
python - <<'PY' $OUT
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "failed" in d: print("FAILED", d["failed"]); continue
    r = d["roofline"]
    print("%-60s x%-4d %10.0f Mpix/s  %8.1f us/launch  %7.0f GB/s (%.1f%%)  %s" % (d["config"]["workload"][:60], d["config"]["images_per_gpu_per_step"], d["value"], r["kernel_ms_per_launch"]*1e3, r["achieved"], 100*r["frac"], r["kernel"]))
PY
