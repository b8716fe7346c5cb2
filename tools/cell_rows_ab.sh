#!/bin/bash
# rows per wave of the cell kernel (AAI_CELL_ROWS) over geometries; needs the experiments build: make -C area_average_interpolation_amd/csrc exp
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
OUT=gpurun_out/cell_rows_ab_${1:-r04}.txt; : > $OUT
one() {   # custom-geometry batch rows
  AAI_CELL_ROWS=$3 timeout -k 10 240 python bench.py --custom $1 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $2 --min-seconds 0.7 2>> gpurun_out/cell_rows_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s x%-2s rows=%-3s %9.1f us/launch  %s' % ('$1', '$2', '$3', r['kernel_ms_per_launch']*1e3, r['kernel']))" >> $OUT || echo "FAILED $1 $3" >> $OUT
}
for g in "8192,8192,1,1,30" "8192,8192,2,1,30" "8192,8192,1.5,1,61" "8192,8192,4,1,45" "8192,8192,3,1,100" "4096,4096,1,2,30" "4096,4096,3,1,17.5"; do
  for rows in 4 8 16 32; do one $g 1 $rows; done
done
cat $OUT
