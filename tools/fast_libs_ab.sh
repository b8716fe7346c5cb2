#!/bin/bash
# fast-mode workloads, library variants against each other on one box: tools/fast_libs_ab.sh <tag> <lib suffix>...
TAG=$1; shift
OUT=gpurun_out/fast_libs_ab_$TAG.txt; : > $OUT
one() {   # lib workload batch
  if [[ "$2" == *,* ]]; then W="--custom $2"; else W="--workload $2"; fi
  AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_$1.so timeout -k 10 240 python bench.py $W --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $3 --min-seconds 0.7 2>> gpurun_out/fast_libs_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-7s %-26s x%-2s %9.1f us/launch  %s' % ('$1', '$2', '$3', r['kernel_ms_per_launch']*1e3, r['kernel']))" >> $OUT || echo "FAILED $1 $2 $3" >> $OUT
}
for lib in "$@"; do for w in cfg3fast cfg5fast wide8fast refdefaultfast "8192,8192,16,1,33,fast" "8192,8192,5,1,45,fast" "4096,4096,1,2,30,fast"; do one $lib $w 1; done; done
cat $OUT
