for dt in u8 u16; do for vec in 8 16; do for rows in 1 2 4 8; do
  if [ $dt = u16 ] && [ $vec = 16 ]; then continue; fi
  echo -n "$dt vec=$vec rows=$rows : "
  AAI_AXIS_TUNE="vec=$vec,rows=$rows" python bench.py --src-dtype $dt --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done; done
