#!/bin/bash
# Profile bench.py on the GPU box with rocprofv3: per-kernel time first, then HBM counters in passes of
# their own (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile_bench.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
# (a profiler preload initialises the GPU in the process it wraps: bench.py must not self-launch ranks from there)
for a in "$@"; do if [ "$a" = "--gpus" ]; then echo "$0 refuses --gpus: profile one rank (bench.py would have to exec workers from a GPU-initialised process)" >&2; exit 2; fi; done
set -e
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --traffic off --configs off --steps 10 --warmup 2 --repeats 1 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
