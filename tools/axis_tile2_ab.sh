#!/bin/bash
# K1 in the transposed quadrants at ratios 2 ... 4: the four-column register path (AAI_AXIS_TUNE=tile=1: tile kernel only below 2:1) against
# the pipelined tile kernel where no output row needs more than four source rows (default)
for c in "8192,8192,2,1,90,area" "8192,8192,3,1,270,area" "8192,8192,2.5,1,90,area" "8192,8192,3.2,1,90,area" "8192,8192,3.9,1,90,area" "8191,8193,3,1,90,area" "8192,8192,8192,2731,90,area"; do for t in "tile=1" ""; do
  AAI_AXIS_TUNE=$t timeout -k 10 240 python bench.py --custom $c --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 2 --batch 4 --min-seconds 0.3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %-8s %9.1f us/launch  %7.0f GB/s  %s' % ('$c', '$t' or 'default', r['kernel_ms_per_launch']*1e3, r['achieved'], r['kernel']))"
done; done
