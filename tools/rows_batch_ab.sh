for rows in 8 16 32; do for b in 4; do
AAI_CELL_ROWS=$rows timeout -k 10 240 python bench.py --workload cfg3 --no-cpu-baseline --traffic off --configs off --steps 5 --warmup 1 --batch $b --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg3 rows=$rows batch=$b  %9.1f us/launch  %9.1f us/image' % (r['kernel_ms_per_launch']*1e3, r['kernel_ms_per_launch']*1e3/$b))"
done; done
