# tools/ab_cull.sh: SAS_CULL=1 / 0 (exact tile culling on / off, one library) on one GPU box: isolated stages at config 3 and the 300-step bench, twice
for r in 1 2; do
for cu in 1 0; do
  export SAS_CULL=$cu
  st=$(timeout -k 10 200 python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(round(s['project'],4), round(s['blend'],4), round(s['total'],4))")
  fps=$(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "SAS_CULL=$cu project,tile,total ms=$st bench_fps=$fps"
done; done
