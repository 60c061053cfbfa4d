#!/usr/bin/env python
"""tests/golden/*.json -- small committed fixtures: inputs and the outputs the ORACLE gives for them, plus every fixed vector the
reference itself holds offline.  The reference is Rust (blst, ssz_rs and sha2 are un-vendored crates, no toolchain here), so
nothing in this file is the output of the reference's own code except the vectors quoted from its tests:
  * crypto/bls.rs:530-544 `test_can_sign` (secret key, message, signature), bin/ec/validator/keystores.rs:240-249 (EIP-2335
    key), bin/ec/bls.rs:6-7 (group order), deneb/blob_sidecar.rs:70-132 (sepolia inclusion proof), deneb/beacon_block.rs:139-154
    (generalized indices);
everything else is `oracle/` on seeded inputs -- regression pins: a change of the oracle OR of the kernels shows up as a diff against
a committed file, and the GPU suite compares the kernels with the same files (tests/test_golden.py).

    python tools/make_golden.py            # rewrites tests/golden/ (CPU only, ~1 min)
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def S(tag: bytes, i: int) -> bytes:
    return hashlib.sha256(b"ecgpu/v1/" + tag + b"/" + i.to_bytes(4, "little")).digest()


# literals of the reference's tests (crypto/bls.rs:532,537,541; bin/ec/validator/keystores.rs:242,245)
REF_CAN_SIGN_SK = "40094c5c6c378857eac09b8ec64c87182f58700c056a8b371ad0eb0a5b983d50"
REF_CAN_SIGN_MSG = b"blst is such a blast"
REF_CAN_SIGN_SIG = ("a01e49276730e4752eef31b0570c8707de501398dac70dd144438cd1bd05fb9b9bb3e1a9ceef0a68cc08904362cafa3f"
                    "1005e5b699a41847fff6f5552260468846de5bdbf94a9aedeb29bc6cdb2c1d34922d9e9af4c0593a69ae978a90b5aba6")
REF_EIP2335_SK = "000000000019d6689c085ae165831e934ff763ae46a2a6c172b3f1b60a8ce26f"
REF_EIP2335_PK = "9612d7a727c9d0a22e185a1c768478dfe919cada9266988cb32359c11f2b7b27f4ae4040902382ae2910c15e2b420d07"


def assert_oracles_reproduce_the_reference_literals(B, cbls):
    """the pins: a golden file that stored oracle OUTPUT would pin nothing"""
    sk = int(REF_CAN_SIGN_SK, 16)
    assert B.sign(sk, REF_CAN_SIGN_MSG).hex() == REF_CAN_SIGN_SIG, "Python oracle does not reproduce test_can_sign"
    assert cbls.sign(sk, REF_CAN_SIGN_MSG).hex() == REF_CAN_SIGN_SIG, "C++ oracle does not reproduce test_can_sign"
    assert B.fast_aggregate_verify([B.sk_to_pk(sk)], REF_CAN_SIGN_MSG, bytes.fromhex(REF_CAN_SIGN_SIG)) == 0
    assert B.sk_to_pk(int(REF_EIP2335_SK, 16)).hex() == REF_EIP2335_PK, "Python oracle does not reproduce the EIP-2335 key"
    assert cbls.sk_to_pk(int(REF_EIP2335_SK, 16)).hex() == REF_EIP2335_PK, "C++ oracle does not reproduce the EIP-2335 key"


def bls_fixture():
    """SURVEY.md 8(d) config 2 in miniature: 24 K = 1 tuples, every 3rd one damaged (the eight fault classes of the bench workload
    in order), statuses by both oracles; 12 randomly mutated tuples (tests/_blsmutate.py); a 3-key aggregate and its sum"""
    from oracle import bls12_381 as B, cbls
    from ethereum_consensus_amd import synthetic as syn
    from tests import _blsmutate as M
    assert_oracles_reproduce_the_reference_literals(B, cbls)
    n = 24
    sks = [1 + int.from_bytes(S(b"sk", i), "big") % (B.R - 1) for i in range(n)]
    msgs = bytearray(b"".join(S(b"msg", i) for i in range(n)))
    pks = bytearray(b"".join(cbls.sk_to_pk(s) for s in sks))
    sigs = bytearray(b"".join(cbls.sign(sks[i], bytes(msgs[32 * i:32 * i + 32])) for i in range(n)))
    want, kinds = syn.bls_inject_faults(pks, msgs, sigs, n, period=3)
    py = [B.fast_aggregate_verify([bytes(pks[48 * i:48 * i + 48])], bytes(msgs[32 * i:32 * i + 32]), bytes(sigs[96 * i:96 * i + 96])) for i in range(n)]
    cpp = list(cbls.fast_aggregate_verify_batch_k1(bytes(pks), bytes(msgs), bytes(sigs)))
    assert py == cpp == list(want)
    m = 12
    sk2 = [1 + int.from_bytes(S(b"sk", 1000 + i), "big") % (B.R - 1) for i in range(m)]
    msg2 = bytearray(b"".join(S(b"msg", 1000 + i) for i in range(m)))
    pk2 = bytearray(b"".join(cbls.sk_to_pk(s) for s in sk2))
    sig2 = bytearray(b"".join(cbls.sign(sk2[i], bytes(msg2[32 * i:32 * i + 32])) for i in range(m)))
    kind2 = M.mutate_tuples(pk2, msg2, sig2, m, every=1, seed=21)
    st2 = [B.fast_aggregate_verify([bytes(pk2[48 * i:48 * i + 48])], bytes(msg2[32 * i:32 * i + 32]), bytes(sig2[96 * i:96 * i + 96])) for i in range(m)]
    assert st2 == list(cbls.fast_aggregate_verify_batch_k1(bytes(pk2), bytes(msg2), bytes(sig2)))
    agg_msg = S(b"att", 0)
    agg_sks = sks[:3]
    agg_sig = B.aggregate([B.sign(s, agg_msg) for s in agg_sks])[1]
    agg_pks = [B.sk_to_pk(s) for s in agg_sks]
    assert B.fast_aggregate_verify(agg_pks, agg_msg, agg_sig) == 0
    return {
        "source": "oracle/bls12_381.py == oracle/c/bls12_381.cpp (both asserted equal when this file was made)",
        "reference_vectors": {
            # the LITERALS of the reference's own tests, copied from the files cited (not oracle output): the generator fails if
            # the oracles do not reproduce them
            "test_can_sign (crypto/bls.rs:530-544)": {
                "secret_key": REF_CAN_SIGN_SK, "message": REF_CAN_SIGN_MSG.hex(), "signature": REF_CAN_SIGN_SIG,
                "origin": "literal of crypto/bls.rs:532,537; both oracles asserted to reproduce it",
                "public_key (derived)": B.sk_to_pk(int(REF_CAN_SIGN_SK, 16)).hex()},
            "EIP-2335 key (bin/ec/validator/keystores.rs:240-249)": {
                "secret_key": REF_EIP2335_SK, "public_key": REF_EIP2335_PK,
                "origin": "literals of bin/ec/validator/keystores.rs:242,245; both oracles asserted to reproduce the key"},
            "group order (bin/ec/bls.rs:6-7)": "%x" % B.R},
        "k1_tuples": {"n": n, "fault_period": 3, "public_keys": bytes(pks).hex(), "messages": bytes(msgs).hex(), "signatures": bytes(sigs).hex(),
                      "statuses": list(want), "fault_kind": [int(k) for k in kinds]},
        "mutated_tuples": {"n": m, "public_keys": bytes(pk2).hex(), "messages": bytes(msg2).hex(), "signatures": bytes(sig2).hex(),
                           "statuses": st2, "kind": [M.KINDS[k] for k in kind2]},
        "aggregate": {"public_keys": [p.hex() for p in agg_pks], "message": agg_msg.hex(), "signature": agg_sig.hex(),
                      "eth_aggregate_public_keys": B.eth_aggregate_public_keys(agg_pks)[1].hex()},
    }


def ssz_fixture():
    """BeaconState roots of every fork (minimal preset, 37 validators; oracle/ssz.py), a header, a validator list, a shuffling"""
    from ethereum_consensus_amd import synthetic
    from oracle import ssz as O, shuffle as SH
    from tests.test_gpu_merkle import _fork_state_value
    out = {"source": "oracle/ssz.py, oracle/shuffle.py", "states": {}}
    for k, fork in enumerate(("phase0", "altair", "bellatrix", "capella", "deneb", "electra")):
        rnd = random.Random(300 + k)
        f = synthetic.state_fields(37, "minimal", seed=300 + k, extra_data=b"golden")
        f["_preset"] = "minimal"
        t, v = _fork_state_value(fork, f, rnd)
        enc = t.serialize(v)
        out["states"][fork] = {"preset": "minimal", "validators": 37, "ssz": enc.hex(), "hash_tree_root": t.htr(v).hex(),
                               "field_roots": [r.hex() for r in t.field_roots(v)]}
    hdr = {"slot": 8626175, "proposer_index": 123456, "parent_root": S(b"hdr", 0), "state_root": S(b"hdr", 1), "body_root": S(b"hdr", 2)}
    out["beacon_block_header"] = {"ssz": O.BeaconBlockHeader.serialize(hdr).hex(), "hash_tree_root": O.BeaconBlockHeader.htr(hdr).hex()}
    vals = synthetic.validators(100).tobytes()
    vt = O.SSZList(O.Validator, 1 << 40)
    from tests._statevalue import oracle_state_value
    fv = synthetic.state_fields(100, "minimal", seed=1)
    out["validators_100"] = {"ssz121": vals.hex(), "hash_tree_root": vt.htr(oracle_state_value(fv)["validators"]).hex()}
    # a deneb and an electra block (minimal preset, seeded random values: tests/_sszrand.py), with the body's root as the
    # independent half (the block container from its five field roots)
    from tests._sszrand import random_value
    out["blocks"] = {}
    for k, (fork, t) in enumerate((("deneb", O.BeaconBlockDeneb(O.BLOCK_MINIMAL)), ("electra", O.BeaconBlockElectra(O.BLOCK_ELECTRA_MINIMAL)))):
        v = random_value(t, random.Random(500 + k), None)
        body_t = dict(t.fields)["body"]
        out["blocks"][fork] = {"preset": "minimal", "ssz": t.serialize(v).hex(), "hash_tree_root": t.htr(v).hex(),
                               "body_root": body_t.htr(v["body"]).hex()}
    seed = S(b"shuffle", 0)
    out["shuffling"] = {"seed": seed.hex(), "n": 333, "rounds": 10, "permutation": SH.compute_shuffled_indices(list(range(333)), seed, 10)}
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]  # e.g. `tools/make_golden.py ssz.json`: the BLS fixture takes minutes of pure-Python pairings
    for name, fn in (("bls.json", bls_fixture), ("ssz.json", ssz_fixture)):
        if only and name not in only:
            continue
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(fn(), f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote tests/golden/" + name)


if __name__ == "__main__":
    main()
