#!/bin/bash
# SQ counters (two passes) of one bench.py workload; prints per-kernel means.  usage: tools/profile_counters2.sh <tag> <bench args...>
# (a profiler preload initialises the GPU in the process it wraps: bench.py must not self-launch ranks from there)
for a in "$@"; do if [ "$a" = "--gpus" ]; then echo "$0 refuses --gpus: profile one rank (bench.py would have to exec workers from a GPU-initialised process)" >&2; exit 2; fi; done
TAG=$1; shift
OUT=gpurun_out/pmc2_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES SQ_IFETCH --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 "$@" > $OUT/b.json 2> $OUT/b.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*kernel_trace.csv", recursive=True))[:1]:
    d = collections.defaultdict(list)
    for x in csv.DictReader(open(f)):
        if "aai_" in x["Kernel_Name"]: d[x["Kernel_Name"][:70]].append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
    for k, v in d.items(): print("%-70s launches=%d avg_ns=%.0f" % (k, len(v), sum(v) / len(v)))
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if any(t in row["Kernel_Name"] for t in ("aai_cell_kernel", "aai_quad_kernel", "aai_quad_fast", "aai_wide_kernel", "aai_rotated_runs")):
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-60s %-24s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
