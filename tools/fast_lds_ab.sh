#!/bin/bash
# fast mode: the LDS-staged kernel (AAI_FAST_LDS=1, experiments build only: make exp) against the shipped register-window kernels
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
OUT=gpurun_out/fast_lds_ab_${1:-r04}.txt; : > $OUT
one() {   # geometry lds
  AAI_FAST_LDS=$2 timeout -k 10 240 python bench.py --custom $1 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.7 2>> gpurun_out/fast_lds_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-30s lds=%s %9.1f us/launch  %s' % ('$1', '$2', r['kernel_ms_per_launch']*1e3, r['kernel']))" >> $OUT || echo "FAILED $1 $2" >> $OUT
}
for g in "8192,8192,8192,2731,17.5,fast" "8192,8192,1,1,30,fast" "8192,8192,1.5,1,61,fast" "8192,8192,2,1,45,fast" "8192,8192,4,1,17.5,fast" "8192,8192,5,1,100,fast" "8192,8192,3,1,200,fast" "8192,8192,3,1,290,fast" "2048,2048,3,1,17.5,fast"; do
  one $g 1; one $g 0
done
cat $OUT
