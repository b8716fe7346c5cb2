#!/bin/bash
# round-4 fuzz: 2 x 2,500 default cases, 2 x 1,500 with the cell kernel forced, 700 larger images (wide footprints)
OUT=gpurun_out/r04_fuzz.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( "$@" 2>&1 | tail -6 ) >> $OUT; echo "rc=$?" >> $OUT; echo "done $*"; }
run timeout -k 10 400 python tools/fuzz_parity.py 2500 401
run timeout -k 10 400 python tools/fuzz_parity.py 2500 402
FUZZ_CELL=1 run timeout -k 10 400 python tools/fuzz_parity.py 1500 411
FUZZ_CELL=1 run timeout -k 10 400 python tools/fuzz_parity.py 1500 412
FUZZ_MAX=420 run timeout -k 10 400 python tools/fuzz_parity.py 700 421
FUZZ_MAX=420 FUZZ_CELL=1 run timeout -k 10 400 python tools/fuzz_parity.py 500 422
cat $OUT
