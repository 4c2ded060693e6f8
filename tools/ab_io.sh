for v in pre prod pre prod; do
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=variants/lib_$v.so; fi
  db=$(python tools/door_b_breakdown.py 2>/dev/null | tail -n 2 | head -n 1)
  pr=$(python tools/door_b_probe.py 2>/dev/null | tail -n 2 | tr '\n' ' ' | cut -c1-120)
  f=$(python tools/config_fps.py 1 2 3 2>/dev/null | cut -d">" -f2 | tr "\n" " ")
  b=$(python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['door_a_sync']['value'],1), round(d['single_view_async']['value'],1))")
  echo "$v | doorb $db | $pr | fps $f | bench,sync,async $b"
done
