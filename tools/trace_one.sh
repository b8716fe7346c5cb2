#!/bin/bash
# rocprofv3 --kernel-trace of one bench.py workload; prints the per-kernel average durations.  usage: tools/trace_one.sh <tag> <bench args...>
# (a profiler preload initialises the GPU in the process it wraps: bench.py must not self-launch ranks from there)
for a in "$@"; do if [ "$a" = "--gpus" ]; then echo "$0 refuses --gpus: profile one rank (bench.py would have to exec workers from a GPU-initialised process)" >&2; exit 2; fi; done
TAG=$1; shift
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 2 --repeats 1 "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    d = collections.defaultdict(list)
    for x in csv.DictReader(open(f)):
        if "aai_" in x["Kernel_Name"]:
            d[(x["Kernel_Name"][:64], x["Grid_Size_X"], x["Grid_Size_Y"], x["Grid_Size_Z"], x["VGPR_Count"], x["LDS_Block_Size"], x["Scratch_Size"])].append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
    for k, v in d.items(): print(k, "launches", len(v), "avg_ns %.0f" % (sum(v) / len(v)))
PY
