#!/bin/bash
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
OUT=gpurun_out/w4_ab.txt; : > $OUT
one() {   # geometry batch wave rows
  if [[ "$1" == *,* ]]; then W="--custom $1"; else W="--workload $1"; fi
  local R=""; [ "$4" != "0" ] && R="AAI_CELL_ROWS=$4"
  env AAI_CELL_WAVE=$3 $R timeout -k 10 240 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $2 --min-seconds 0.6 2>> gpurun_out/w4_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-22s x%-2s wave=%s rows=%-3s %9.1f us/launch %8.1f us/image' % ('$1', '$2', '$3', '$4', r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$2))" >> $OUT || echo "FAILED $1 $2 $3 $4" >> $OUT
}
for g in cfg3 "8192,8192,3,1,30" "8192,8192,4,1,45" "8192,8192,5,1,17.5"; do
  one $g 1 2 0; for rows in 8 16 32; do one $g 1 4 $rows; done
done
one cfg3 8 2 0; one cfg3 8 4 16; one cfg3 8 4 32
cat $OUT
