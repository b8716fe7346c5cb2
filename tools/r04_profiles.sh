#!/bin/bash
# round-4 evidence run: headline profile (kernel stats + PMC), kernel stats / traffic / SQ counters of the rotated workloads
tools/profile_bench.sh r04_cfg2 > gpurun_out/r04_cfg2_rocprofv3_summary.txt 2>&1
echo "profiled cfg2"
for w in cfg3 cfg3fast cfg5 cfg5fast wide8; do
  tools/profile_bench.sh r04_$w --workload $w --batch 1 --steps 5 > gpurun_out/r04_${w}_rocprofv3_summary.txt 2>&1
  tools/profile_counters2.sh r04_$w --workload $w --batch 1 > gpurun_out/r04_${w}_sq_counters.txt 2>&1
  echo "profiled $w"
done
