#!/bin/bash
# fast mode: launch order against the XCD-aware tile order (tile rows dealt cyclically over the XCDs; AAI_XCD_ROWS, experiments build)
export AAI_LIB=$PWD/area_average_interpolation_amd/libaai_hip_exp.so
export TMPDIR=/tmp
OUT=gpurun_out/fast_xcd_ab_${1:-r04}.txt; : > $OUT
one() {   # geometry xcd batch
  AAI_XCD_ROWS=$2 timeout -k 10 240 python bench.py --custom $1 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $3 --min-seconds 0.7 2>> gpurun_out/fast_xcd_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-30s x%s xcd=%s %9.1f us/launch  %s' % ('$1', '$3', '$2', r['kernel_ms_per_launch']*1e3, r['kernel']))" >> $OUT || echo "FAILED $1 $2" >> $OUT
}
for g in "8192,8192,8192,2731,17.5,fast" "8192,8192,2,1,45,fast" "8192,8192,4,1,17.5,fast" "8192,8192,5,1,100,fast" "8192,8192,3,1,200,fast" "8192,8192,3,1,290,fast" "8192,8192,2.5,1,33,fast" "2048,2048,3,1,17.5,fast"; do
  one $g 0 1; one $g 1 1
done
one "8192,8192,8192,2731,17.5,fast" 0 4; one "8192,8192,8192,2731,17.5,fast" 1 4
for x in 0 1; do
  export AAI_XCD_ROWS=$x
  P=gpurun_out/pmc_fastxcd$x; mkdir -p $P
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/f -- python3 bench.py --workload cfg3fast --batch 1 --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 > $P/f.json 2> $P/f.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $P/c -- python3 bench.py --workload cfg3fast --batch 1 --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 > $P/c.json 2> $P/c.err
  python3 - $P $x >> $OUT <<'PY'
import csv, glob, sys, collections
out, x = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/f/**/*kernel_trace.csv", recursive=True))[:1]:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "aai_quad_fast" in r["Kernel_Name"]]
    print("xcd=%s cfg3fast under rocprofv3: launches=%d avg_ns=%.0f" % (x, len(d), sum(d) / max(len(d), 1)))
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "aai_quad_fast" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, v in sorted(acc.items()): print("xcd=%s   %-26s mean=%.6g" % (x, c, sum(v) / len(v)))
PY
done
cat $OUT
