#!/bin/bash
# A/B of the rotated area kernels on one box: cell formulation (default) against the quad formulation (AAI_CELL=0)
OUT=gpurun_out/cell_ab_${1:-r03}.txt
: > $OUT
for w in cfg3 cfg5s cfg5; do
  b=1; s=5
  for cell in 1 0; do
    AAI_CELL=$cell timeout -k 10 240 python bench.py --workload $w --no-cpu-baseline --steps $s --warmup 1 --batch $b --min-seconds 0.5 2>> gpurun_out/cell_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-8s cell=%s  %9.1f us/launch  %8.0f Mpix/s  %6.0f GB/s  %s' % ('$w', '$cell', r['kernel_ms_per_launch']*1e3, d['value'], r['achieved'], r['kernel']))" >> $OUT || echo "FAILED $w cell=$cell" >> $OUT
  done
done
cat $OUT
