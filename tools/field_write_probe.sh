#!/bin/bash
# builds and runs tools/field_write_probe.c on a 2^20-validator mainnet state (run on the GPU box, from the repo root)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from ethereum_consensus_amd import synthetic
f = synthetic.state_fields(1 << 20, "mainnet", seed=5)
open("/tmp/ecgpu_state.ssz", "wb").write(synthetic.serialize_state(f))
PY
gcc -O2 -std=gnu99 -Iinclude tools/field_write_probe.c -o /tmp/field_write_probe -Lethereum_consensus_amd/lib -lecgpu -Wl,-rpath,$PWD/ethereum_consensus_amd/lib
/tmp/field_write_probe /tmp/ecgpu_state.ssz 1048576
