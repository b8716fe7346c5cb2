#!/bin/bash
# wide footprints (ratios above ~5.5 : 1 at an angle): the double-precision runs kernel (AAI_WIDE=0) against the fp32 formulation
# over a window split into parts, a lane per part (aai_wide_kernel, default)
for c in "8192,8192,8,1,17.5,area" "8192,8192,6,1,45,area" "8192,8192,12,1,33,area" "8192,8192,16,1,45,area" "8192,8192,24,1,10,area" "4096,4096,8,1,17.5,area"; do for w in 0 1; do
  AAI_WIDE=$w timeout -k 10 240 python bench.py --custom $c --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --batch 1 --min-seconds 0.3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s wide=%s  %9.1f us/launch  %7.0f GB/s  %s  %s' % ('$c', '$w', r['kernel_ms_per_launch']*1e3, r['achieved'], r['kernel'], d.get('plan','')))"
done; done
