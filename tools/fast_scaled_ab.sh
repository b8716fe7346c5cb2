#!/bin/bash
# fast mode with replicated source pixels (up-sampling, ratios near 1): kernel time per launch
for c in "4096,4096,1,4,45,fast" "4096,4096,1,2,30,fast" "4096,4096,1,1,61,fast" "2048,2048,1,3,17.5,fast"; do
  timeout -k 10 240 python bench.py --custom $c --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --batch 1 --min-seconds 0.3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %9.1f us/launch  %s' % ('$c', r['kernel_ms_per_launch']*1e3, r['kernel']))"
done
