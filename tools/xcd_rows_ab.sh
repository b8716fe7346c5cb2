#!/bin/bash
# XCD-aware workgroup order (AAI_XCD_ROWS = tile rows / row blocks per XCD band; 0 = launch order), experiments build
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
OUT=gpurun_out/xcd_rows_ab_${1:-r04}.txt; : > $OUT
one() {   # workload-or-geometry xcd batch
  if [[ "$1" == *,* ]]; then W="--custom $1"; else W="--workload $1"; fi
  AAI_XCD_ROWS=$2 timeout -k 10 240 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $3 --min-seconds 0.7 2>> gpurun_out/xcd_rows_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-30s x%s xcd=%s %9.1f us/launch  %s' % ('$1', '$3', '$2', r['kernel_ms_per_launch']*1e3, r['kernel']))" >> $OUT || echo "FAILED $1 $2" >> $OUT
}
for w in cfg3fast cfg3 "8192,8192,2,1,45" "8192,8192,4,1,30" "8192,8192,1,1,30" cfg5; do
  for x in 0 1 2 4; do one $w $x 1; done
done
for x in 0 1 2; do one cfg3 $x 8; done
cat $OUT
