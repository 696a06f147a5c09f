#!/bin/bash
# A/B of the 128x256 weight-gradient tile: ms/step of the default bench (2 runs each, same box)
run() { python3 bench.py --no-cpu --no-roofline --steps 400 --warmup 30 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for cfg in "RBVAE_WG_NT4=0" "RBVAE_WG_NT4=1 RBVAE_WG_NT4_SLAB=8 RBVAE_WG_KS_SMALL=3" "RBVAE_WG_NT4=1 RBVAE_WG_NT4_SLAB=8 RBVAE_WG_KS_SMALL=6" "RBVAE_WG_NT4=1 RBVAE_WG_NT4_SLAB=4 RBVAE_WG_KS_SMALL=3" "RBVAE_WG_NT4=1 RBVAE_WG_NT4_SLAB=6 RBVAE_WG_KS_SMALL=4" "RBVAE_WG_NT4=0"; do
  for i in 1 2; do echo "$cfg run$i ms_per_step=$(env $cfg bash -c "$(declare -f run); run")"; done
done
