#!/bin/bash
# A/B of environment switches on ONE box (boxes differ by ~10 %): usage tools/ab.sh "VAR=a VAR=b ..." [reps]
reps=${2:-2}
for r in $(seq $reps); do
  for kv in $1; do
    ms=$(env $kv python bench.py --no-cpu --steps 300 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $r $kv ms_per_step=$ms"
  done
done
