#!/bin/bash
# cell kernel, config 3, one image: rows per strip x tail shape
for rows in 8 12 16; do for t in "10,4" "20,4" "30,4" "30,8"; do
  AAI_CELL_ROWS=$rows AAI_CELL_TAIL=$t timeout -k 10 240 python bench.py --workload cfg3 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg3 rows=%-3s tail=%-6s %9.1f us/launch' % ('$rows', '$t', r['kernel_ms_per_launch']*1e3))"
done; done
