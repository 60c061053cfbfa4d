"""Fixtures for bench/reference_rs (the unmodified reference timed off-box): the K = 1 tuples and the deneb state bench.py
feeds the MI355X backend, written to files.  Needs the GPU (keys and signatures are generated on the device)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ethereum_consensus_amd import bls, ssz, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/fixtures")
    ap.add_argument("--tuples", type=int, default=65536)
    ap.add_argument("--validators", type=int, default=1 << 20)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    sks, msgs = bench.bls_inputs(a.tuples, 0)
    pks = bls.sk_to_pk_batch(sks)
    sigs = bls.sign_batch(sks, [msgs[32 * i:32 * i + 32] for i in range(a.tuples)])
    msgs = bytearray(msgs)
    for i in range(0, a.tuples, 64):  # the same fault injection as bench.py: every 64th message was not signed
        msgs[32 * i] ^= 1
    with open(os.path.join(a.out, f"tuples_{a.tuples}.bin"), "wb") as f:
        for i in range(a.tuples):
            f.write(pks[48 * i:48 * i + 48] + bytes(msgs[32 * i:32 * i + 32]) + sigs[96 * i:96 * i + 96])
    enc = synthetic.beacon_state_deneb(a.validators, "mainnet", seed=1)
    with open(os.path.join(a.out, f"state_deneb_mainnet_{a.validators}.ssz"), "wb") as f:
        f.write(enc)
    print("state root (this backend):", ssz.hash_tree_root_beacon_state_deneb(enc, 0).hex())
    print("wrote", a.out)


if __name__ == "__main__":
    main()
