#!/bin/bash
# fast mode (K3) over ratios and angles: fp32 dst-frame kernel vs the round-1 line-walking kernel (AAI_ROT_TUNE=quad=0)
for ratio in 1.5 2 2.5 3 3.5 4 4.5 5 5.5 6 6.4; do for ang in 1.5 17.5 45; do
  c="8192,8192,$ratio,1,$ang"
  line="$c"
  for q in 1 0; do
    t=$(AAI_ROT_TUNE="quad=$q" python bench.py --custom $c,fast --no-cpu-baseline --steps 4 --warmup 1 --batch 2 --min-seconds 0.2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4f %s' % (r['kernel_ms_per_launch']/2, r['kernel'].split('<')[0][4:14]))")
    line="$line | quad=$q $t"
  done
  echo "$line"
done; done
