#!/bin/bash
# usage: tools/sweep_lib.sh lib1.so lib2.so ...  -> ms/step of the default bench with each build of the library (RBVAE_LIB), 2 runs each
for l in "$@"; do
  for i in 1 2; do
    r=$(RBVAE_LIB=$l python3 bench.py --no-cpu --no-roofline --steps 400 --warmup 30 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$l run$i ms_per_step=$r"
  done
done
