#!/bin/bash
# A/B of the rotated area kernels on one box: cell formulation (default) against the quad formulation (AAI_CELL=0);
# extra arguments: values of AAI_CELL_ROWS to try for the cell kernel (rows per strip)
TAG=${1:-r03}; shift
OUT=gpurun_out/cell_ab_$TAG.txt
: > $OUT
one() {   # workload, cell, rows
  AAI_CELL=$2 AAI_CELL_ROWS=$3 timeout -k 10 240 python bench.py --workload $1 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.5 2>> gpurun_out/cell_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-8s cell=%s rows=%-3s %9.1f us/launch  %8.0f Mpix/s  %6.0f GB/s  %s' % ('$1', '$2', '$3', r['kernel_ms_per_launch']*1e3, d['value'], r['achieved'], r['kernel']))" >> $OUT || echo "FAILED $1 cell=$2" >> $OUT
}
for w in cfg3 cfg5s cfg5; do
  one $w 0 0
  one $w 1 0
  for rows in "$@"; do one $w 1 $rows; done
done
cat $OUT
