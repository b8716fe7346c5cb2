#!/bin/bash
# fast mode at wide footprints (the double-precision line-walking kernel): kernel time per launch
for c in "8192,8192,6,1,17.5,fast" "8192,8192,8,1,17.5,fast" "8192,8192,8,1,45,fast" "8192,8192,12,1,33,fast" "8192,8192,16,1,45,fast" "8192,8192,24,1,10,fast"; do
  timeout -k 10 240 python bench.py --custom $c --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --batch 1 --min-seconds 0.3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %9.1f us/launch  %7.0f GB/s  %s' % ('$c', r['kernel_ms_per_launch']*1e3, r['achieved'], r['kernel']))"
done
