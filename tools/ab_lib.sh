#!/bin/bash
# usage: tools/ab_lib.sh name1 name2 ...   -> ms/step of the default bench with librbvae_hip_<name>.so ("base" = the product library), 2 rounds
run() { python3 bench.py --no-cpu --no-roofline --steps 400 --warmup 30 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = base ]; then echo "base $(run)"; else echo "$n $(RBVAE_LIB=$PWD/symbols-from-video_amd/librbvae_hip_$n.so run)"; fi
  done
done
