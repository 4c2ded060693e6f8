# tools/ab_seg.sh: first guess of the tile segments (SAS_SEG_FACTOR x the mean list; config 3: 16 -> 16 384 keys, 8 -> 8 192, 4 -> 4 096 + one regrow) on one box, twice
for r in 1 2; do
for sf in 16 8 4 32; do
  export SAS_SEG_FACTOR=$sf
  st=$(timeout -k 10 200 python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(round(s['project'],4), round(s['blend'],4), round(s['total'],4))")
  fps=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "SAS_SEG_FACTOR=$sf project,tile,total ms=$st bench_fps=$fps"
done; done
