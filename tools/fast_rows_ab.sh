#!/bin/bash
# fast mode: 16 x 4 wave (AAI_FAST_ROWS=0) against the row-shaped wave with whole-line stores (1: under replication, the default;
# 2: for every plain image) -- config 5 and geometries without replication at ratios 1 ... 3
one() { timeout -k 10 240 python bench.py $1 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s %-12s %9.1f us/launch  %s' % ('$1', '$2', r['kernel_ms_per_launch']*1e3, r['kernel']))"; }
AAI_FAST_ROWS=0 one "--workload cfg5fast" "16x4 wave"; one "--workload cfg5fast" "64x1 wave"
for g in 8192,8192,1.0,1.0,17.5 8192,8192,1.5,1.0,17.5 8192,8192,2.0,1.0,17.5 8192,8192,3.0,1.0,17.5 8192,8192,1.0,1.0,45 8192,8192,2.0,1.0,45 4096,4096,1.0,1.5,30; do
  AAI_FAST_ROWS=0 one "--custom $g,fast" "16x4 wave"; AAI_FAST_ROWS=2 one "--custom $g,fast" "64x1 wave"
done
