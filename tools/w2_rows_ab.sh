#!/bin/bash
OUT=gpurun_out/w2_rows_ab.txt; : > $OUT
one() {   # lib workload batch rows
  if [[ "$2" == *,* ]]; then W="--custom $2"; else W="--workload $2"; fi
  AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_$1.so AAI_CELL_ROWS=$4 timeout -k 10 240 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $3 --min-seconds 0.7 2>> gpurun_out/w2_rows_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-4s %-20s x%-2s rows=%-3s %9.1f us/launch %8.1f us/image' % ('$1', '$2', '$3', '$4', r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$3))" >> $OUT || echo "FAILED $1 $2 $3 $4" >> $OUT
}
one exp cfg3 1 4; one w2 cfg3 1 4; one w2 cfg3 1 8; one w2 cfg3 1 16
one exp cfg3 8 4; one w2 cfg3 8 4; one w2 cfg3 8 8; one w2 cfg3 8 16
one exp cfg5 1 32; one w2 cfg5 1 32; one w2 cfg5 1 64
one exp "8192,8192,2,1,45" 4 4; one w2 "8192,8192,2,1,45" 4 8; one w2 "8192,8192,2,1,45" 4 16
one exp "8192,8192,1,1,30" 2 16; one w2 "8192,8192,1,1,30" 2 16; one w2 "8192,8192,1,1,30" 2 32
cat $OUT
