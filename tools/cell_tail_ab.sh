#!/bin/bash
# cell kernel: shorter strips for the last rows of a launch (AAI_CELL_TAIL="<percent of rows>,<rows per tail strip>")
for w in cfg3 cfg5; do for t in "0,0" "10,2" "10,4" "20,2" "20,4" "30,4" "40,4"; do
  AAI_CELL_TAIL=$t timeout -k 10 240 python bench.py --workload $w --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch 1 --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-6s tail=%-6s %9.1f us/launch' % ('$w', '$t', r['kernel_ms_per_launch']*1e3))"
done; done
