for dt in f32 u8 u16; do for wl in cfg2 cfg1; do
  echo -n "$dt $wl : "
  python bench.py --workload $wl --src-dtype $dt --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline'))"
done; done
