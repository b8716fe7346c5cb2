#!/bin/bash
# cell kernel: tile width A/B (AAI_CELL_TW = 16 | 64) per workload and batch
for w in cfg3 cfg5; do for tw in 16 64; do for b in 1 4; do
  [ $w = cfg5 -a $b = 4 ] && continue
  AAI_CELL_TW=$tw timeout -k 10 240 python bench.py --workload $w --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $b --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w tile=$tw batch=$b  %9.1f us/launch  %9.1f us/image  %s' % (r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$b, r['kernel']))"
done; done; done
