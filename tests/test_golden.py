"""tests/golden/*.json (tools/make_golden.py): committed inputs with the outputs the oracle gave for them, plus the vectors the
reference itself holds.  CPU: the oracles still produce the files' outputs (both BLS restatements, the SSZ one, the shuffling) --
a change of the oracle shows up as a diff against history.  GPU: the kernels, through the C ABI, produce them too."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def _tuples(t):
    return bytes.fromhex(t["public_keys"]), bytes.fromhex(t["messages"]), bytes.fromhex(t["signatures"]), t["statuses"]


def test_oracles_reproduce_the_bls_fixture():
    from oracle import bls12_381 as B, cbls
    g = _load("bls.json")
    kat = g["reference_vectors"]["test_can_sign (crypto/bls.rs:530-544)"]
    # the one signature the reference's own test pins, bit for bit
    assert kat["signature"].startswith("a01e49276730e4752eef31b0570c8707") and kat["signature"].endswith("90b5aba6")
    assert B.sign(int(kat["secret_key"], 16), bytes.fromhex(kat["message"])).hex() == kat["signature"]
    assert cbls.sign(int(kat["secret_key"], 16), bytes.fromhex(kat["message"])).hex() == kat["signature"]
    eip = g["reference_vectors"]["EIP-2335 key (bin/ec/validator/keystores.rs:240-249)"]
    assert eip["public_key"].startswith("9612d7a727c9d0a2") and B.sk_to_pk(int(eip["secret_key"], 16)).hex() == eip["public_key"]
    assert int(g["reference_vectors"]["group order (bin/ec/bls.rs:6-7)"], 16) == B.R
    for key in ("k1_tuples", "mutated_tuples"):
        pks, msgs, sigs, want = _tuples(g[key])
        assert list(cbls.fast_aggregate_verify_batch_k1(pks, msgs, sigs)) == want
        for i in range(0, g[key]["n"], 5):
            assert B.fast_aggregate_verify([pks[48 * i:48 * i + 48]], msgs[32 * i:32 * i + 32], sigs[96 * i:96 * i + 96]) == want[i]
    a = g["aggregate"]
    pks = [bytes.fromhex(p) for p in a["public_keys"]]
    assert cbls.fast_aggregate_verify(pks, bytes.fromhex(a["message"]), bytes.fromhex(a["signature"])) == 0
    assert B.eth_aggregate_public_keys(pks)[1].hex() == a["eth_aggregate_public_keys"]


def test_oracles_reproduce_the_ssz_fixture():
    from oracle import cref, shuffle as SH, ssz as O
    g = _load("ssz.json")
    assert len(g["states"]) == 6
    hdr = g["beacon_block_header"]
    h = bytes.fromhex(hdr["ssz"])  # 5 leaves: two uint64 chunks, three roots (phase0/beacon_block.rs:83-91)
    leaves = [h[0:8].ljust(32, b"\0"), h[8:16].ljust(32, b"\0"), h[16:48], h[48:80], h[80:112]]
    assert O.merkleize_chunks(leaves, 5).hex() == hdr["hash_tree_root"]
    v = g["validators_100"]
    assert cref.htr_validators(bytes.fromhex(v["ssz121"]))[0].hex() == v["hash_tree_root"]
    s = g["shuffling"]
    assert SH.compute_shuffled_indices(list(range(s["n"])), bytes.fromhex(s["seed"]), s["rounds"]) == s["permutation"]
    # a block's root from its header fields and the body root (phase0/beacon_block.rs:66-81: five leaves)
    for fork, b in g["blocks"].items():
        enc = bytes.fromhex(b["ssz"])
        leaves = [enc[0:8].ljust(32, b"\0"), enc[8:16].ljust(32, b"\0"), enc[16:48], enc[48:80], bytes.fromhex(b["body_root"])]
        assert O.merkleize_chunks(leaves, 5).hex() == b["hash_tree_root"], fork
    # the state container's root from its field roots (the independent half of each state entry)
    for fork, st in g["states"].items():
        roots = [bytes.fromhex(r) for r in st["field_roots"]]
        assert O.merkleize_chunks(roots, len(roots)).hex() == st["hash_tree_root"], fork


def test_lane_simulator_reproduces_the_block_vectors():
    """the generic SSZ plan (the product's own planner and lane programs, on the CPU simulator) on the committed block encodings"""
    from ethereum_consensus_amd import ssz_types as T
    from tests.test_hostsim_ssz import sim_htr
    g = _load("ssz.json")
    for fork, b in g["blocks"].items():
        pt = T.BeaconBlockDeneb(T.MINIMAL) if fork == "deneb" else T.BeaconBlockElectra(T.ELECTRA_MINIMAL)
        rc, root, _ = sim_htr(pt, bytes.fromhex(b["ssz"]))
        assert rc == 0 and root.hex() == b["hash_tree_root"], fork


@pytest.fixture(scope="module")
def gpu():
    from ethereum_consensus_amd import _lib
    L = _lib.load(build_if_missing=False)
    assert L.ecgpu_init(0) == 0, "no gfx950 device: the GPU tests need one"
    return L


@pytest.mark.gpu
def test_kernels_reproduce_the_bls_fixture(gpu):
    from ethereum_consensus_amd import bls
    g = _load("bls.json")
    kat = g["reference_vectors"]["test_can_sign (crypto/bls.rs:530-544)"]
    sk, msg = bytes.fromhex(kat["secret_key"]), bytes.fromhex(kat["message"])
    assert bls.sk_to_pk_batch(sk).hex() == kat["public_key (derived)"]
    assert bls.sign_batch(sk, [msg]).hex() == kat["signature"]
    bls.verify_signature(bytes.fromhex(kat["public_key (derived)"]), msg, bytes.fromhex(kat["signature"]))
    eip = g["reference_vectors"]["EIP-2335 key (bin/ec/validator/keystores.rs:240-249)"]
    assert bls.sk_to_pk_batch(bytes.fromhex(eip["secret_key"])).hex() == eip["public_key"]
    for key in ("k1_tuples", "mutated_tuples"):
        pks, msgs, sigs, want = _tuples(g[key])
        assert list(bls.fast_aggregate_verify_batch(pks, None, msgs, sigs)) == want, key
    a = g["aggregate"]
    pks = [bytes.fromhex(p) for p in a["public_keys"]]
    bls.fast_aggregate_verify(pks, bytes.fromhex(a["message"]), bytes.fromhex(a["signature"]))
    assert bls.eth_aggregate_public_keys(pks).hex() == a["eth_aggregate_public_keys"]


@pytest.mark.gpu
def test_kernels_reproduce_the_ssz_fixture(gpu):
    from ethereum_consensus_amd import shuffling, ssz
    g = _load("ssz.json")
    for fork, st in g["states"].items():
        assert ssz.hash_tree_root_beacon_state(fork, bytes.fromhex(st["ssz"]), ssz.MINIMAL).hex() == st["hash_tree_root"], fork
    from ethereum_consensus_amd import ssz_types as T
    for fork, b in g["blocks"].items():
        pt = T.BeaconBlockDeneb(T.MINIMAL) if fork == "deneb" else T.BeaconBlockElectra(T.ELECTRA_MINIMAL)
        assert ssz.hash_tree_root(pt, bytes.fromhex(b["ssz"])).hex() == b["hash_tree_root"], fork
    hdr = g["beacon_block_header"]
    assert ssz.hash_tree_root_beacon_block_header(bytes.fromhex(hdr["ssz"])).hex() == hdr["hash_tree_root"]
    v = g["validators_100"]
    assert ssz.hash_tree_root_validators(bytes.fromhex(v["ssz121"])).hex() == v["hash_tree_root"]
    s = g["shuffling"]
    assert list(shuffling.compute_shuffled_indices(list(range(s["n"])), bytes.fromhex(s["seed"]), s["rounds"])) == s["permutation"]
