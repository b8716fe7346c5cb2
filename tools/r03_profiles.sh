#!/bin/bash
# round-3 evidence run: headline profile (kernel stats + PMC), traffic of the rotated kernels, SQ counters
tools/profile_bench.sh r03_cfg2 > gpurun_out/r03_cfg2_rocprofv3_summary.txt 2>&1
for w in cfg3 cfg3fast cfg5 cfg5fast wide8; do
  tools/profile_bench.sh r03_$w --workload $w --batch 1 --steps 5 > gpurun_out/r03_${w}_rocprofv3_summary.txt 2>&1
  tools/profile_counters2.sh r03_$w --workload $w --batch 1 > gpurun_out/r03_${w}_sq_counters.txt 2>&1
  echo "profiled $w"
done
tools/profile_bench.sh r03_cfg5bilinear --workload cfg5bilinear --batch 1 --steps 5 > gpurun_out/r03_cfg5bilinear_rocprofv3_summary.txt 2>&1
