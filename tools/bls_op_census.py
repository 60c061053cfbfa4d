#!/usr/bin/env python3
"""Multiplier census of one K = 1 verification on the lane programs (tests/hostsim hs_op_census), mean over the first 16
tuples of bench.py's workload: the numbers bench.py's BLS_OPS_BY_BUILD carries.  `--calls`: the compact-code tower."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ethereum_consensus_amd import build  # noqa: E402
from oracle import cbls  # noqa: E402  (workload generation only)
import bench  # noqa: E402

variant = "calls" if "--calls" in sys.argv else ""
L = ctypes.CDLL(build.build_hostsim(verbose=False, variant=variant))
n = 16
sks, msgs = bench.bls_inputs(n, 0)
tot = [0] * 12
for i in range(n):
    sk = int.from_bytes(sks[32 * i:32 * i + 32], "big")
    msg = msgs[32 * i:32 * i + 32]
    pk = cbls.sk_to_pk(sk)
    sig = cbls.sign(sk, msg)
    out = (ctypes.c_uint64 * 12)()
    L.hs_op_census(pk, msg, ctypes.c_uint64(32), sig, out)
    for j in range(12):
        tot[j] += out[j]
names = ["bls_pk_validate", "bls_sig", "bls_h2c", "bls_pairing"]
ops = {nm: tuple(round(tot[3 * s + j] / n) for j in range(3)) for s, nm in enumerate(names)}
print(ops)
print("multiplies per signature:", sum(m * 351 + s * 273 + x for m, s, x in ops.values()))
