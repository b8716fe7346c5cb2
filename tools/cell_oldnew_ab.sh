#!/bin/bash
# old (previous commit's) against new cell kernel on one box: AAI_LIB = libaai_hip_oldexp.so / libaai_hip_exp.so, rows per wave 4 / 8
OUT=gpurun_out/cell_oldnew_ab_${1:-r04}.txt; : > $OUT
one() {   # lib workload batch rows
  if [[ "$2" == *,* ]]; then W="--custom $2"; else W="--workload $2"; fi
  AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_$1.so AAI_CELL_ROWS=$4 timeout -k 10 240 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $3 --min-seconds 0.7 2>> gpurun_out/cell_oldnew_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-7s %-20s x%-2s rows=%-3s %9.1f us/launch %8.1f us/image  %s' % ('$1', '$2', '$3', '$4', r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$3, r['kernel']))" >> $OUT || echo "FAILED $1 $2 $3 $4" >> $OUT
}
for lib in oldexp exp; do for b in 1 8; do for rows in 4 8; do one $lib cfg3 $b $rows; done; done; done
for lib in oldexp exp; do for rows in 4 8; do one $lib "8192,8192,2,1,45" 4 $rows; done; done
cat $OUT
