#!/bin/bash
# Build a kernel-variant copy of the library for A/B runs on one box: tools/build_variant.sh <name> "<extra hipcc flags>" <unit.hip> [...]
# -> ethereum_consensus_amd/lib/libecgpu_<name>.so (the named translation units recompiled with the flags, every other object
# shared with the product build).  Select it with ECGPU_LIB=<path>.  Development only; the product is lib/libecgpu.so.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
CS=ethereum_consensus_amd/csrc; OBJ=ethereum_consensus_amd/lib/obj; VOBJ=ethereum_consensus_amd/lib/obj_$name
mkdir -p $VOBJ
objs=""
for o in $OBJ/*.o; do
  b=$(basename $o .o)
  hit=0
  for u in "$@"; do [ "$u" = "$b.hip" ] && hit=1; done
  if [ $hit = 1 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 \
      -Iinclude -I$CS $flags -c $CS/$b.hip -o $VOBJ/$b.o &
    objs="$objs $VOBJ/$b.o"
  else
    objs="$objs $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ethereum_consensus_amd/lib/libecgpu_$name.so $objs
echo built ethereum_consensus_amd/lib/libecgpu_$name.so
