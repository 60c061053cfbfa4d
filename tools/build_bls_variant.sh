#!/bin/bash
# development helper: libecgpu variant with bls.hip compiled under extra -D flags
#   tools/build_bls_variant.sh w1 -DECG_BLS_WAVES=1
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=ethereum_consensus_amd/lib/variants
mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function ${OPT:--O3} "$@" -Iinclude -Iethereum_consensus_amd/csrc \
    -c ethereum_consensus_amd/csrc/bls.hip -o $d/bls_$name.o
objs=$(ls ethereum_consensus_amd/lib/obj/*.o | grep -v "/bls.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libecgpu_$name.so $objs $d/bls_$name.o
echo built $d/libecgpu_$name.so
