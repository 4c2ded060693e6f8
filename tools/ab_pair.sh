# tools/ab_pair.sh: view pairing on / off x four / six frame slots, 300-step bench, twice (one GPU box)
for r in 1 2; do
for pv in 1 0; do
  for slots in 4 6; do
  export SAS_PAIR=$pv SAS_SLOTS=$slots
  fps=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "SAS_PAIR=$pv SAS_SLOTS=$slots bench_fps=$fps"
  done
done; done
