#!/bin/bash
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
for b in 1 8; do for x in 0 1 2 4; do
  AAI_XCD_ROWS=$x timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $b --min-seconds 0.6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg3 x$b xcd=$x %9.1f us/launch %8.1f us/image' % (r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$b))"
done; done
for x in 0 1 2 4; do
  AAI_XCD_ROWS=$x timeout -k 10 200 python bench.py --custom 8192,8192,4,1,45 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 2 --min-seconds 0.6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('4:1@45 x2 xcd=$x %9.1f us/launch' % (r['kernel_ms_per_launch']*1e3))"
done
