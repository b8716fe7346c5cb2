for c in 8192,8192,1,1,270 4096,4096,1,2,90 4096,4096,1,4,270 8192,8192,1.5,1,90 8192,8192,2,1,90 8192,8192,3,1,270 8192,8192,8192,2731,90 8000,6000,5,2,90; do for tile in 0 1; do
  echo -n "$c tile=$tile : "
  AAI_AXIS_TUNE="tile=$tile" python bench.py --custom $c --batch 4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ms/launch', round(d['ms_per_step'],4), 'GB/s', round(r['achieved']), r['kernel'])"
done; done
