#!/bin/bash
# L2 / fabric read counters of one bench.py workload (three passes of <= 4 TCC counters): request sizes on the memory side,
# DRAM share, L2 hit / miss, L1 -> L2 requests.  usage: tools/profile_cache.sh <tag> <bench args...>
for a in "$@"; do if [ "$a" = "--gpus" ]; then echo "$0 refuses --gpus" >&2; exit 2; fi; done
TAG=$1; shift
OUT=gpurun_out/cache_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --traffic off --configs off --steps 3 --warmup 1 --repeats 1"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/a -- $B "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $OUT/b -- $B "$@" > $OUT/b.json 2> $OUT/b.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/c -- $B "$@" > $OUT/c.json 2> $OUT/c.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if any(k in row["Kernel_Name"] for k in ("aai_cell_kernel", "aai_quad_kernel", "aai_quad_fast", "aai_axis_kernel")):
            acc[(row["Kernel_Name"][:56], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-56s %-30s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
