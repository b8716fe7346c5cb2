#!/bin/bash
# K1 in the transposed quadrants at ratios below 2 (aai_axis_tile_kernel): independent waves with 64-byte store segments
# (AAI_AXIS_TUNE=tile=1) against the cooperative store of 256-byte segments (default)
for c in "8192,8192,1,1,270,area" "8192,8192,1.5,1,90,area" "8192,8192,1,2,90,area" "4096,4096,1,4,270,area" "8192,8192,1.9,1,90,area" "8191,8193,1,1,90,area"; do for t in "tile=1" ""; do
  AAI_AXIS_TUNE=$t timeout -k 10 240 python bench.py --custom $c --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 2 --batch 4 --min-seconds 0.3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-26s %-8s %9.1f us/launch  %7.0f GB/s  %s' % ('$c', '$t' or 'coop', r['kernel_ms_per_launch']*1e3, r['achieved'], r['kernel']))"
done; done
