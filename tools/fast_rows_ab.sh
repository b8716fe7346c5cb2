#!/bin/bash
# fast mode under replication: 16 x 4 wave (AAI_FAST_ROWS=0) against the row-shaped wave with whole-line stores (default)
one() { timeout -k 10 240 python bench.py --workload $1 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s %-16s %9.1f us/launch  %s' % ('$1', '$2', r['kernel_ms_per_launch']*1e3, r['kernel']))"; }
for w in cfg5fast; do AAI_FAST_ROWS=0 one $w "16x4 wave"; one $w "64x1 wave"; done
for w in cfg5bilinear cfg5bicubic cfg5; do one $w ""; done
