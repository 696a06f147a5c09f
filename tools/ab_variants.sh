#!/bin/bash
# Build librbvae_hip_<name>.so variants of ONE source file with different -D flags (same-box A/B through RBVAE_LIB).
# usage: tools/ab_variants.sh <file.hip> name1:"-DX=1 -DY=0" name2:"..."
set -e
R=$(cd $(dirname $0)/.. && pwd); P=$R/symbols-from-video_amd
src=$1; shift
base=$(basename $src .hip)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $flags -c $P/csrc/$src -o /tmp/${base}_$name.o
  ls $P/build/*.o | grep -v "\.dbg\.o$" | grep -v "/dbg.o$" | grep -v "/$base.o" | xargs /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/librbvae_hip_$name.so /tmp/${base}_$name.o
  echo built $P/librbvae_hip_$name.so
done
