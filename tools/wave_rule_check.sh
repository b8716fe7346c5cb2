#!/bin/bash
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
OUT=gpurun_out/wave_rule_check.txt; : > $OUT
for g in "8192,8192,2,1,17.5" "8192,8192,2.2,1,75" "8192,8192,2.5,1,45" "8192,8192,3.5,1,10" "8192,8192,5.5,1,30" "6000,4000,2.1,1,200" "8192,8192,2.39,1,2" "4096,4096,3,1,45"; do
  for wv in 1 2; do
    AAI_CELL_WAVE=$wv timeout -k 10 200 python bench.py --custom $g --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-24s wave=$wv %9.1f us  %s' % ('$g', r['kernel_ms_per_launch']*1e3, r['kernel']))" >> $OUT
  done
done
cat $OUT
