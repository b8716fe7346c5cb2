#!/bin/bash
# SQ-level counters for one workload (own run, counters only): instruction mix, lane utilisation, stalls.
# usage: tools/profile_counters.sh <tag> <bench args...>
# (a profiler preload initialises the GPU in the process it wraps: bench.py must not self-launch ranks from there)
for a in "$@"; do if [ "$a" = "--gpus" ]; then echo "$0 refuses --gpus: profile one rank (bench.py would have to exec workers from a GPU-initialised process)" >&2; exit 2; fi; done
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1 "$@" > $OUT/b.json 2> $OUT/b.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "aai_" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-60s %-28s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
