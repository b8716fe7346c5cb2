#!/bin/bash
# The cell kernel on batches of distinct images (no help from the Infinity Cache): rows per wave x XCD band height, experiments build
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
OUT=gpurun_out/cell_batch_ab_${1:-r04}.txt; : > $OUT
one() {   # workload-or-geometry batch rows xcd
  if [[ "$1" == *,* ]]; then W="--custom $1"; else W="--workload $1"; fi
  AAI_CELL_ROWS=$3 AAI_XCD_ROWS=$4 timeout -k 10 240 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $2 --min-seconds 0.7 2>> gpurun_out/cell_batch_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-26s x%-2s rows=%-3s xcd=%-2s %9.1f us/launch %8.1f us/image  %s' % ('$1', '$2', '$3', '$4', r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$2, r['kernel']))" >> $OUT || echo "FAILED $1 $2 $3 $4" >> $OUT
}
for rows in 4 8; do for x in 0 1 2 4; do one cfg3 8 $rows $x; done; done
for rows in 4 8; do for x in 0 2; do one cfg3 1 $rows $x; done; done
for rows in 4 8 16; do for x in 0 2; do one "8192,8192,2,1,45" 4 $rows $x; done; done
for rows in 8 16 32; do one "8192,8192,1,1,30" 4 $rows 2; done
for rows in 8 16 32; do one cfg5 4 $rows 0; done
cat $OUT
