#!/bin/bash
# library variants against each other on one box: tools/cell_libs_ab.sh <tag> <lib suffix>...   (libaai_hip_<suffix>.so beside the product)
TAG=$1; shift
OUT=gpurun_out/cell_libs_ab_$TAG.txt; : > $OUT
one() {   # lib workload batch
  if [[ "$2" == *,* ]]; then W="--custom $2"; else W="--workload $2"; fi
  AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_$1.so timeout -k 10 240 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $3 --min-seconds 0.7 2>> gpurun_out/cell_libs_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-7s %-20s x%-2s %9.1f us/launch %8.1f us/image  %s' % ('$1', '$2', '$3', r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$3, r['kernel']))" >> $OUT || echo "FAILED $1 $2 $3" >> $OUT
}
for lib in "$@"; do one $lib cfg3 1; one $lib cfg3 8; one $lib cfg5 1; one $lib "8192,8192,2,1,45" 4; one $lib "8192,8192,1,1,30" 2; one $lib wide8 1; done
cat $OUT
