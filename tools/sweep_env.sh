#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> ms/step of the default bench for each value (2 runs each, same box)
VAR=$1; shift
for v in "$@"; do
  for i in 1 2; do
    r=$(env $VAR=$v python3 bench.py --no-cpu --no-roofline --steps 400 --warmup 30 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$VAR=$v run$i ms_per_step=$r"
  done
done
