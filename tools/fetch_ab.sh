#!/bin/bash
# FETCH_SIZE of the cell kernel, old / new library, 8 images per launch
export TMPDIR=/tmp
for lib in oldexp exp; do
  AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_$lib.so rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch8_$lib -- python3 bench.py --workload cfg3 --batch 8 --no-cpu-baseline --traffic off --configs off --steps 4 --warmup 1 --repeats 1 > gpurun_out/fetch8_$lib.json 2> gpurun_out/fetch8_$lib.err
  python3 - <<PY
import csv,glob
for f in glob.glob('gpurun_out/pmc_fetch8_$lib/*/*counter_collection.csv'):
    rows=[r for r in csv.DictReader(open(f)) if 'aai_cell_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
    v=[float(r['Counter_Value']) for r in rows]
    print('$lib cell kernel FETCH_SIZE x8: launches', len(v), 'mean KiB', sum(v)/max(1,len(v)))
PY
done
