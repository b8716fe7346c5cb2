#!/bin/bash
# SQ + cache counters of config 3's fast mode: the shipped register-window kernel against the LDS-staged kernel (experiments build)
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
export TMPDIR=/tmp
for lds in 0 1; do
  export AAI_FAST_LDS=$lds
  OUT=gpurun_out/pmc_fastlds$lds; mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --workload cfg3fast --batch 1 --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 > $OUT/a.json 2> $OUT/a.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 bench.py --workload cfg3fast --batch 1 --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 > $OUT/f.json 2> $OUT/f.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/c -- python3 bench.py --workload cfg3fast --batch 1 --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 > $OUT/c.json 2> $OUT/c.err
  python3 - $OUT $lds <<'PY'
import csv, glob, sys, collections
out, lds = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/a/**/*kernel_trace.csv", recursive=True))[:1]:
    d = collections.defaultdict(list)
    for x in csv.DictReader(open(f)):
        if "aai_quad_fast" in x["Kernel_Name"]: d[x["Kernel_Name"][:60]].append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
    for k, v in d.items(): print("lds=%s %-60s launches=%d avg_ns=%.0f" % (lds, k, len(v), sum(v) / len(v)))
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "aai_quad_fast" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print("lds=%s   %-30s n=%d mean=%.6g" % (lds, c, len(v), sum(v) / len(v)))
PY
done
