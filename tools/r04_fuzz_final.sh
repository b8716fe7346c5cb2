#!/bin/bash
OUT=gpurun_out/r04_fuzz_final.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( "$@" 2>&1 | tail -3 ) >> $OUT; echo "done $*"; }
run timeout -k 10 500 python tools/fuzz_parity.py 2500 441
run timeout -k 10 500 python tools/fuzz_parity.py 2500 442
run timeout -k 10 500 python tools/fuzz_parity.py 2500 443
FUZZ_CELL=1 run timeout -k 10 500 python tools/fuzz_parity.py 2500 444
FUZZ_CELL=1 run timeout -k 10 500 python tools/fuzz_parity.py 2500 445
FUZZ_MAX=480 run timeout -k 10 500 python tools/fuzz_parity.py 900 446
FUZZ_MAX=480 FUZZ_CELL=1 run timeout -k 10 500 python tools/fuzz_parity.py 900 447
grep -h 'cases\|==' $OUT
