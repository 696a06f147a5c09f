#!/bin/bash
# usage: tools/ab_tool.sh "<command>" name1 name2 ...  -> runs the command once per library variant ("base" = product library), 2 rounds
cmd=$1; shift
for i in 1 2; do
  for n in "$@"; do
    if [ "$n" = base ]; then echo "== base"; bash -c "$cmd"; else echo "== $n"; RBVAE_LIB=$PWD/symbols-from-video_amd/librbvae_hip_$n.so bash -c "$cmd"; fi
  done
done
