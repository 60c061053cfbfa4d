#!/bin/bash
# development helper: libecgpu variant with different Fp2-VM generator arguments
#   tools/build_vm2_variant.sh NAME --lanes 8 --window 60
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=ethereum_consensus_amd/lib/variants
mkdir -p $d/inc_$name
python tools/gen_bls_vm2.py "$@" > $d/inc_$name/bls_vm2_prog.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -DECG_VM2_PROG_HEADER="\"$PWD/$d/inc_$name/bls_vm2_prog.h\"" -Iinclude -Iethereum_consensus_amd/csrc \
    -c ethereum_consensus_amd/csrc/bls_vm2.hip -o $d/bls_vm2_$name.o
objs=$(ls ethereum_consensus_amd/lib/obj/*.o | grep -v bls_vm2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libecgpu_$name.so $objs $d/bls_vm2_$name.o
echo built $d/libecgpu_$name.so
