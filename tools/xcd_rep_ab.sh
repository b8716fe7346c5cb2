#!/bin/bash
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
for rep in 1 2 3; do for g in cfg3 "8192,8192,3,1,30" "8192,8192,4,1,45"; do for x in 0 2; do
  if [[ "$g" == *,* ]]; then W="--custom $g"; else W="--workload $g"; fi
  AAI_XCD_ROWS=$x timeout -k 10 200 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('rep$rep %-20s x1 xcd=$x %9.1f us' % ('$g', r['kernel_ms_per_launch']*1e3))"
done; done; done
