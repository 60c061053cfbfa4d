"""GPU parity tests for the SHA-256 / SSZ Merkleization path: HIP kernels through the C ABI
(ethereum_consensus_amd.ssz -> libecgpu.so) against the oracle on the same seeded inputs."""
import ctypes
import hashlib
import random

import numpy as np
import pytest

from oracle import cref, ssz as ossz
from tests import test_oracle_ssz as fx
from tests._statevalue import oracle_state_root_fast, oracle_state_value
from tests.test_hostsim_merkle import make_validators

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from ethereum_consensus_amd import _lib, ssz
    L = _lib.load(build_if_missing=False)
    assert L.ecgpu_init(-1) == 0, L.ecgpu_last_error()
    return ssz


def rnd(n, seed):
    return random.Random(seed).randbytes(n)


def test_sha256(gpu):
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000):
        d = rnd(n, n)
        assert gpu.hash(d) == hashlib.sha256(d).digest()
    # crypto::hash takes any length (crypto/bls.rs:12-20): one lane walks 16 385 blocks of a 1 MiB input; block-boundary lengths
    for n in (1 << 20, (1 << 20) - 9, (1 << 20) + 55):
        d = rnd(n, 7)
        assert gpu.hash(d) == hashlib.sha256(d).digest(), n


def test_reference_fixtures(gpu):
    # sepolia BlobSidecar inclusion proof, deneb/blob_sidecar.rs:47-64,108-132
    leaf = gpu.merkleize(fx.KZG_COMMITMENT, 2)
    assert leaf == ossz.BlsPublicKey.htr(fx.KZG_COMMITMENT)
    assert gpu.is_valid_merkle_branch(leaf, fx.PROOF, 17, 221184 % (1 << 17), fx.BODY_ROOT)
    assert not gpu.is_valid_merkle_branch(leaf, fx.PROOF, 17, 221185 % (1 << 17), fx.BODY_ROOT)
    # config 1: BeaconBlockHeader of the same fixture (blob_sidecar.rs:78-84)
    enc = ossz.BeaconBlockHeader.serialize(fx.HEADER)
    assert gpu.hash_tree_root_beacon_block_header(enc) == ossz.BeaconBlockHeader.htr(fx.HEADER)
    assert gpu.hash_tree_root_beacon_block_header(bytes(112)) == ossz.BeaconBlockHeader.htr(ossz.BeaconBlockHeader.default())
    dom = rnd(32, 5)
    assert gpu.compute_signing_root(leaf, dom) == ossz.SigningData.htr({"object_root": leaf, "domain": dom})


def test_headers_batch_of_random(gpu):
    r = random.Random(3)
    for _ in range(20):
        h = {"slot": r.getrandbits(64), "proposer_index": r.getrandbits(64), "parent_root": r.randbytes(32),
             "state_root": r.randbytes(32), "body_root": r.randbytes(32)}
        assert gpu.hash_tree_root_beacon_block_header(ossz.BeaconBlockHeader.serialize(h)) == ossz.BeaconBlockHeader.htr(h)


@pytest.mark.parametrize("nbytes,limit", [
    (0, 0), (0, 8), (0, 1 << 40), (1, 1), (32, 1), (33, 2), (64, 2), (5 * 32, 8), (1000, 1 << 20),
    (32 * 511, 512), (32 * 512, 512), (32 * 513, 1024), (32 * 777, 1 << 38), (8 * 4097, 1 << 38),
    (32 * 100_003, 1 << 35), (1 << 22, 1 << 17), ((1 << 24) + 8, 1 << 38)])
def test_merkleize_vs_oracle(gpu, nbytes, limit):
    d = rnd(nbytes, nbytes + 1)
    want, h = cref.merkleize_bytes(d, limit, None)
    assert gpu.merkleize(d, limit) == want
    assert gpu.last_hash64_count() == h
    want2, _ = cref.merkleize_bytes(d, limit, nbytes // 8)
    assert gpu.merkleize(d, limit, mix_in_length=nbytes // 8) == want2
    if nbytes <= 32 * 777:
        assert want == ossz.merkleize_bytes(d, limit if limit else None)


def test_merkleize_rejects_over_limit(gpu):
    with pytest.raises(gpu.MerkleizationError):
        gpu.merkleize(bytes(96), 2)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 64, 255, 256, 257, 1000, 4099, 1 << 14])
def test_validator_list(gpu, n):
    from ethereum_consensus_amd import synthetic as S
    enc = S.validators(n, seed=n).tobytes()
    want, h = cref.htr_validators(enc)
    assert gpu.hash_tree_root_validators(enc) == want
    assert gpu.last_hash64_count() == h
    if n <= 64:
        vs = make_validators(n, n)
        enc = b"".join(ossz.Validator.serialize(v) for v in vs)
        assert gpu.hash_tree_root_validators(enc) == ossz.SSZList(ossz.Validator, 1 << 40).htr(vs)


def test_staged_registry_pass_at_every_alignment_and_with_ragged_ends(gpu):
    """The registry pass of 2^20 .. 2^21 validators (csrc/merkle.hip k_merkle_pass<2, ValidatorLeaves>: 16-byte lane loads into
    LDS, whole waves of 256 records; the ragged end goes to the generic pass): the same records at all 16 byte alignments of the
    device pointer, and lengths that end inside a wave, inside a lane's four records, and on a wave boundary."""
    import torch
    from ethereum_consensus_amd import _lib, synthetic as S
    L = _lib.load()
    n_max = (1 << 20) + 256 + 3
    data = S.validators(n_max, seed=11).tobytes()
    t = torch.frombuffer(bytearray(bytes(32) + data + bytes(32)), dtype=torch.uint8).cuda()
    shifted = torch.empty(32 + len(data) + 64, dtype=torch.uint8, device="cuda")
    out = torch.zeros(32, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lens = [1 << 20, (1 << 20) + 1, (1 << 20) + 255, (1 << 20) + 256, n_max]
    want = {n: cref.htr_validators(data[:121 * n])[0] for n in lens}
    for mis in range(16):
        shifted[mis:mis + len(data)] = t[32:32 + len(data)]
        torch.cuda.synchronize()  # (torch's default stream is the null handle here: the library would run on a stream of its own)
        for n in (lens if mis in (0, 5) else lens[:1] + lens[-1:]):
            rc = L.ecgpu_htr_validators_dev(shifted.data_ptr() + mis, n, 1 << 40, out.data_ptr(), st)
            assert rc == 0
            torch.cuda.synchronize()
            assert bytes(out.cpu().numpy()) == want[n], (mis, n)


def test_staged_registry_pass_beyond_two_million_validators(gpu):
    """2^21 + 77 validators (mainnet's registry is about to pass 2^21): the leaf pass stays the staged one, four records per lane"""
    from ethereum_consensus_amd import synthetic as S
    n = (1 << 21) + 77
    enc = S.validators(n, seed=21).tobytes()
    want, h = cref.htr_validators(enc)
    assert gpu.hash_tree_root_validators(enc) == want
    assert gpu.last_hash64_count() == h


def test_device_resident_unaligned_slices(gpu):
    """_dev entry points on byte-unaligned slices of a device buffer (what the state driver does)."""
    import torch
    from ethereum_consensus_amd import _lib
    L = _lib.load()
    data = rnd(121 * 1000 + 77, 42)
    t = torch.frombuffer(bytearray(b"\xa5" * 64 + data + b"\xa5" * 64), dtype=torch.uint8).cuda()
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for mis in (0, 1, 2, 3, 7):
        sl = data[mis: mis + 121 * 999]
        rc = L.ecgpu_htr_validators_dev(t.data_ptr() + 64 + mis, 999, 1 << 40, out.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        assert bytes(out[:32].cpu().numpy()) == cref.htr_validators(sl)[0]
        n = 32 * 300 + 5
        rc = L.ecgpu_merkleize_dev(t.data_ptr() + 64 + mis, n, 1 << 20, 1, 9, out.data_ptr() + 32, st)
        assert rc == 0
        torch.cuda.synchronize()
        assert bytes(out[32:].cpu().numpy()) == cref.merkleize_bytes(data[mis: mis + n], 1 << 20, 9)[0]


@pytest.mark.parametrize("preset,n", [("minimal", 0), ("minimal", 1), ("minimal", 37), ("minimal", 2000), ("mainnet", 5),
                                      ("mainnet", 3000)])
def test_beacon_state_root_vs_python_oracle(gpu, preset, n):
    from ethereum_consensus_amd import synthetic as S
    f = S.state_fields(n, preset, seed=n + 3, n_votes=n % 7, n_hist_roots=n % 5, n_hist_summaries=n % 3,
                       extra_data=b"x" * (n % 33))
    enc = S.serialize_state(f)
    P = ossz.MINIMAL if preset == "minimal" else ossz.MAINNET
    want = ossz.BeaconStateDeneb(P).htr(oracle_state_value(f))
    assert gpu.hash_tree_root_beacon_state_deneb(enc, S.PRESETS[preset]["id"]) == want
    assert oracle_state_root_fast(f, preset) == want


@pytest.mark.parametrize("n", [70_001, 300_001, 600_000])
def test_beacon_state_root_of_ragged_mid_sized_registries(gpu, n):
    """mainnet states between the small ones above and the 2^20 of config 3: registries that end inside a tile and inside a
    lane's subtree, lists whose last chunk is partial, tile counts that are not powers of two -- the arrival tickets of the fused
    tail count exactly these.  Against the C restatement (oracle/c/sha256_merkle.c over the oracle's own type tree)."""
    from ethereum_consensus_amd import synthetic as S
    f = S.state_fields(n, "mainnet", seed=n % 97, n_votes=n % 11, n_hist_roots=n % 5, n_hist_summaries=n % 7, extra_data=b"y" * (n % 33))
    enc = S.serialize_state(f)
    want = oracle_state_root_fast(f, "mainnet")
    assert gpu.hash_tree_root_beacon_state_deneb(enc, 0) == want
    # the device-resident entry (the one bench.py times), twice back to back on one stream: the second root's upload must not
    # disturb the first one's tail (one arena, a ring of pinned plan slots)
    import torch
    L = gpu._lib.load()
    st = torch.cuda.current_stream().cuda_stream
    d = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
    fixed = int(L.ecgpu_beacon_state_deneb_fixed_size(0))
    fixed_part = ctypes.create_string_buffer(bytes(enc[:fixed]), fixed)
    outs = [torch.zeros(32, dtype=torch.uint8, device="cuda") for _ in range(2)]
    for o in outs:
        rc = L.ecgpu_htr_beacon_state_deneb_dev(d.data_ptr(), len(enc), fixed_part, 0, o.data_ptr(), st)
        assert rc == 0, (rc, L.ecgpu_last_error())
    torch.cuda.synchronize()
    assert [bytes(o.cpu().numpy()) for o in outs] == [want, want]


def test_beacon_state_root_full_size(gpu):
    """config 3: mainnet preset, 2^20 validators (BASELINE.json configs[2])."""
    from ethereum_consensus_amd import synthetic as S
    n = 1 << 20
    f = S.state_fields(n, "mainnet")
    enc = S.serialize_state(f)
    want = oracle_state_root_fast(f, "mainnet")
    got = gpu.hash_tree_root_beacon_state_deneb(enc, 0)
    assert got == want
    hashes = gpu.last_hash64_count()
    assert 10_000_000 < hashes < 10_300_000
    # idempotence + sensitivity: flipping one balance bit changes the root, flipping back restores it
    b = bytearray(enc)
    assert gpu.hash_tree_root_beacon_state_deneb(enc, 0) == got
    b[-1] ^= 1
    assert gpu.hash_tree_root_beacon_state_deneb(bytes(b), 0) != got


def test_malformed_state_is_rejected(gpu):
    from ethereum_consensus_amd import synthetic as S
    enc = S.beacon_state_deneb(3, "minimal")
    with pytest.raises(gpu.MerkleizationError):
        gpu.hash_tree_root_beacon_state_deneb(enc + b"\0", 1)
    with pytest.raises(gpu.MerkleizationError):
        gpu.hash_tree_root_beacon_state_deneb(enc[:100], 1)


# ---- generic SSZ hash_tree_root (ecgpu_htr_ssz): deneb BeaconBlock and the SSZ kinds ---------------
def test_generic_ssz_kinds_and_beacon_block(gpu):
    import random
    from ethereum_consensus_amd import ssz_types as T
    from tests._sszrand import random_value
    from tests.test_hostsim_ssz import PAIRS
    r = random.Random(41)
    for pt, ot in PAIRS:
        for fill in ("empty", "full", None, None):
            v = random_value(ot, r, fill)
            assert gpu.hash_tree_root(pt, ot.serialize(v)) == ot.htr(v)
    ot, pt = ossz.Bitlist(2048), T.bitlist(2048)
    for nbits in (0, 1, 7, 8, 9, 255, 256, 257, 2047, 2048):
        v = [r.random() < 0.5 for _ in range(nbits)]
        assert gpu.hash_tree_root(pt, ot.serialize(v)) == ot.htr(v)
    for preset in ("mainnet", "minimal"):
        pt = T.BeaconBlockDeneb(T.MAINNET if preset == "mainnet" else T.MINIMAL)
        ot = ossz.BeaconBlockDeneb(ossz.BLOCK_MAINNET if preset == "mainnet" else ossz.BLOCK_MINIMAL)
        for fill in ("empty", "full", None, None, None):
            v = random_value(ot, r, fill)
            assert gpu.hash_tree_root(pt, ot.serialize(v)) == ot.htr(v), (preset, fill)
    for preset in ("mainnet", "minimal"):  # electra/beacon_block.rs:17-63 (widening: the block that goes with the electra state)
        pt = T.BeaconBlockElectra(T.ELECTRA_MAINNET if preset == "mainnet" else T.ELECTRA_MINIMAL)
        ot = ossz.BeaconBlockElectra(ossz.BLOCK_ELECTRA_MAINNET if preset == "mainnet" else ossz.BLOCK_ELECTRA_MINIMAL)
        for fill in ("empty", "full", None, None):
            v = random_value(ot, r, fill)
            assert gpu.hash_tree_root(pt, ot.serialize(v)) == ot.htr(v), ("electra", preset, fill)
    # wide sequences: pass + tile kernels under the generic plan
    for pt, ot, v in [(T.list_(T.uint64, 1 << 20), ossz.SSZList(ossz.uint64, 1 << 20), [r.randrange(1 << 64) for _ in range(70000)]),
                      (T.list_(T.bytevector(48), 4096), ossz.SSZList(ossz.ByteVector(48), 4096), [r.randbytes(48) for _ in range(3000)]),
                      (T.list_(T.bytelist(1 << 30), 1 << 20), ossz.SSZList(ossz.ByteList(1 << 30), 1 << 20),
                       [r.randbytes(r.randrange(0, 200)) for _ in range(1500)] + [r.randbytes(70000)])]:
        assert gpu.hash_tree_root(pt, ot.serialize(v)) == ot.htr(v)
    # malformed encodings: rejected, never a crash
    from ethereum_consensus_amd import _lib
    with pytest.raises(_lib.EcgpuError):
        gpu.hash_tree_root(T.bitlist(8), b"\x00")
    with pytest.raises(_lib.EcgpuError):
        gpu.hash_tree_root(T.BeaconBlockDeneb(T.MAINNET), bytes(50))


def test_resident_state_patches(gpu):
    """ecgpu_resident_state_*: upload once, overwrite the bytes a block changed, re-Merkleize on the device.  After
    every batch of patches the root equals the root of the patched encoding computed from scratch (whose parity with the
    Python oracle the tests above establish), for patches in the big lists, the small fields and the fixed part."""
    from ethereum_consensus_amd import synthetic
    r = random.Random(12)
    for preset, n in (("minimal", 3000), ("mainnet", 5000)):
        f = synthetic.state_fields(n, preset, seed=21)
        enc = bytearray(synthetic.serialize_state(f))
        pid = synthetic.PRESETS[preset]["id"]
        st = gpu.ResidentBeaconStateDeneb(bytes(enc), pid)
        assert st.hash_tree_root() == gpu.hash_tree_root_beacon_state_deneb(bytes(enc), pid)
        want0 = ossz.BeaconStateDeneb(ossz.MINIMAL if preset == "minimal" else ossz.MAINNET).htr(oracle_state_value(f))
        assert st.hash_tree_root() == want0
        from ethereum_consensus_amd import _lib
        L = _lib.load()
        fixed_size = int(L.ecgpu_beacon_state_deneb_fixed_size(pid))
        for slot in range(3):
            patches, used = [], set()
            for _ in range(200):
                ln = r.choice([1, 8, 8, 8, 32])
                off = r.randrange(fixed_size, len(enc) - ln)  # somewhere in the variable part: validators, balances, flags ...
                if any(b in used for b in range(off, off + ln)):
                    continue  # the patches of one call must not overlap
                used.update(range(off, off + ln))
                patches.append((off, r.randbytes(ln)))
            patches.append((40, (8_700_000 + slot).to_bytes(8, "little")))  # the `slot` field (fixed part, a gathered chunk)
            patches.append((48 + 16 + 112 + 32 * 5, r.randbytes(32)))        # block_roots[5] (fixed part, big vector)
            for off, b in patches:
                enc[off:off + len(b)] = b
            st.patch(patches)
            assert st.hash_tree_root() == gpu.hash_tree_root_beacon_state_deneb(bytes(enc), pid), (preset, slot)
        # the cached validator roots: one patch across many records (the cache is rebuilt), then a few single-byte ones
        # (only those records are re-hashed), then a patch that straddles two records
        vals_off = fixed_size + len(f["historical_roots"].tobytes()) + 72 * len(f["eth1_data_votes"])  # 3rd variable field
        assert bytes(enc[vals_off:vals_off + 121]) == f["validators"].tobytes()[:121]
        big = r.randbytes(121 * (n // 2) + 17)
        enc[vals_off + 121 * 5 + 3:vals_off + 121 * 5 + 3 + len(big)] = big
        st.patch([(vals_off + 121 * 5 + 3, big)])
        assert st.hash_tree_root() == gpu.hash_tree_root_beacon_state_deneb(bytes(enc), pid)
        few = [(vals_off + 121 * v + r.randrange(121), r.randbytes(1)) for v in (0, 1, n // 3, n - 1)]
        few.append((vals_off + 121 * 77 - 4, r.randbytes(8)))  # last 4 bytes of record 76, first 4 of record 77
        for off, b in few:
            enc[off:off + len(b)] = b
        st.patch(few)
        assert st.hash_tree_root() == gpu.hash_tree_root_beacon_state_deneb(bytes(enc), pid)
        assert st.hash_tree_root() == gpu.hash_tree_root_beacon_state_deneb(bytes(enc), pid)  # nothing dirty: cache as is
        with pytest.raises(gpu.MerkleizationError):
            st.patch([(len(enc) - 4, bytes(8))])          # runs past the end
        st.close()


def test_concurrent_calls_from_several_host_threads(gpu):
    """The reference functions are pure and re-entrant and its spec-test harness runs trials on several threads
    (SURVEY.md 8b): every host thread gets its own stream + arena, so concurrent calls must not disturb each other."""
    import threading
    from ethereum_consensus_amd import bls, synthetic
    from tests import _blscases as C
    vals = [synthetic.validators(3000 + 17 * t).tobytes() for t in range(6)]
    want = [cref.htr_validators(v)[0] for v in vals]
    pk = bls.sk_to_pk_batch(C.CAN_SIGN_SK.to_bytes(32, "big"))
    errors = []

    def worker(t):
        try:
            for it in range(6):
                assert gpu.hash_tree_root_validators(vals[t]) == want[t]
                d = rnd(32 * (100 + t), t)
                assert gpu.merkleize(d, 1 << 20, 7) == ossz.mix_in_length(ossz.merkleize_bytes(d, 1 << 20), 7)
                if t % 2 == 0:
                    bls.verify_signature(pk, C.CAN_SIGN_MSG, C.CAN_SIGN_SIG)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_sharded_big_lists_equal_the_unsharded_roots(gpu):
    """validators / packed lists cut into aligned subtrees (one per rank), sub-roots combined by the
    top-of-tree call -- here the "ranks" run one after the other on the one GPU; the collective itself is covered by
    tests/test_dist_gloo.py."""
    from ethereum_consensus_amd import shard
    import random
    r = random.Random(9)
    for n in (0, 1, 5, 1000, 4096, 70001):
        vals = bytes(r.getrandbits(8) for _ in range(121 * min(n, 64))) * (n // 64 + 1)
        vals = vals[:121 * n]
        full = gpu.hash_tree_root_validators(vals)
        bal = bytes(r.getrandbits(8) for _ in range(256)) * (n * 8 // 256 + 1)
        bal = bal[:8 * n]
        n_chunks = (8 * n + 31) // 32
        limit_chunks = (1 << 40) * 8 // 32
        full_bal = gpu.merkleize(bal, limit_chunks, n)
        for world in (1, 2, 4, 8):
            w = shard.subtree_width(n, world)
            subs = b""
            for rank in range(world):
                lo, hi = shard.subtree_range(n, rank, world)
                subs += gpu.validators_subtree_root(vals[121 * lo:121 * hi], w)
            assert gpu.merkleize_subtree_roots(subs, w, 1 << 40, n) == full
            wc = shard.subtree_width(n_chunks, world)
            subs = b""
            for rank in range(world):
                lo, hi = shard.subtree_range(n_chunks, rank, world)
                subs += gpu.merkleize(bal[32 * lo:32 * hi], wc)
            assert gpu.merkleize_subtree_roots(subs, wc, limit_chunks, n) == full_bal
    # single-process form of the host helper (world 1: no collective)
    assert gpu.hash_tree_root_validators_sharded(None, vals, n) == full
    assert gpu.merkleize_sharded(None, bal, n_chunks, limit_chunks, n) == full_bal


@pytest.mark.parametrize("fork", ["altair", "bellatrix", "capella", "deneb", "electra"])
def test_one_beacon_state_sharded_over_emulated_ranks(gpu, fork):
    """SURVEY.md 8e row 2 / north_star "2^20-validator batch at 1, 2, 4 and 8 GPUs": ONE state over `world` ranks through the
    two-phase device entries (ecgpu_beacon_state_shard_subroots_dev per rank, the 5 x 32-byte exchange, ecgpu_htr_beacon_state_
    sharded_dev) equals the oracle's root of the same state -- ragged, tiny and empty registries, worlds that are not powers of
    two, both presets; the ranks run one after the other on the one GPU (the collective: tests/test_dist_gloo.py,
    tests/test_gpu_dist.py)."""
    import random
    import torch
    from ethereum_consensus_amd import synthetic
    ssz = gpu
    L = ssz._lib.load()
    rnd = random.Random(5)
    st = torch.cuda.current_stream().cuda_stream
    nl = L.ecgpu_beacon_state_shard_lists()
    assert nl == 5
    for preset_name, preset, n in (("minimal", ssz.MINIMAL, 0), ("minimal", ssz.MINIMAL, 1), ("minimal", ssz.MINIMAL, 37), ("mainnet", ssz.MAINNET, 300),
                                   ("minimal", ssz.MINIMAL, 4097), ("minimal", ssz.MINIMAL, 70001)):
        f = synthetic.state_fields(n, preset_name, seed=rnd.randrange(1000))
        f["_preset"] = preset_name
        if n <= 300:
            t, v = _fork_state_value(fork, f, rnd)
            enc, want = t.serialize(v), t.htr(v)
        else:  # the pure-Python oracle would take minutes: the unsharded GPU root (itself pinned to the oracle at the small sizes)
            if fork != "deneb":
                continue
            enc = synthetic.serialize_state(f)
            want = ssz.hash_tree_root_beacon_state(fork, enc, preset)
            assert want == oracle_state_root_fast(f, preset_name)
        fixed = L.ecgpu_beacon_state_fixed_size(ssz.FORKS[fork], preset)
        d = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
        h_fixed = ctypes.create_string_buffer(bytes(enc[:fixed]), fixed)
        out = torch.zeros(32, dtype=torch.uint8, device="cuda")
        for world in (1, 2, 3, 8):
            for keep in (False, True):  # phase B computes the other fields itself / takes the roots phase A left behind
                d_all = torch.full((32 * nl * world,), 0xAB, dtype=torch.uint8, device="cuda")
                d_keep = torch.full((64 * 32,), 0xCD, dtype=torch.uint8, device="cuda")
                kp = d_keep.data_ptr() if keep else None
                for rank in range(world):
                    rc = L.ecgpu_beacon_state_shard_subroots_dev(ssz.FORKS[fork], d.data_ptr(), len(enc), h_fixed, preset, rank, world,
                                                                 d_all.data_ptr() + 32 * nl * rank, kp, st)
                    assert rc == 0, (rc, L.ecgpu_last_error())
                out.zero_()
                rc = L.ecgpu_htr_beacon_state_sharded_dev(ssz.FORKS[fork], d.data_ptr(), len(enc), h_fixed, preset, d_all.data_ptr(), world, kp,
                                                          out.data_ptr(), st)
                assert rc == 0, (rc, L.ecgpu_last_error())
                torch.cuda.synchronize()
                assert bytes(out.cpu().numpy()) == want, (fork, preset_name, n, world, keep)
    # argument checks: rank outside the world, no ranks, phase0 (host entry only)
    assert L.ecgpu_beacon_state_shard_subroots_dev(ssz.FORKS[fork], d.data_ptr(), len(enc), h_fixed, preset, 2, 2, d_all.data_ptr(), None, st) == -3
    assert L.ecgpu_htr_beacon_state_sharded_dev(ssz.FORKS[fork], d.data_ptr(), len(enc), h_fixed, preset, d_all.data_ptr(), 0, None, out.data_ptr(), st) == -3
    assert L.ecgpu_beacon_state_shard_subroots_dev(0, d.data_ptr(), len(enc), h_fixed, preset, 0, 1, d_all.data_ptr(), None, st) == -3


def test_box_selfcheck_runs_and_reports_positive_times(gpu):
    """ecgpu_selfcheck_ifetch[_sweep]: the instruction-fetch probe bench.py reports next to its numbers."""
    from ethereum_consensus_amd import _lib
    L = _lib.load()
    a, b = ctypes.c_double(0), ctypes.c_double(0)
    assert L.ecgpu_selfcheck_ifetch(ctypes.byref(a), ctypes.byref(b)) == 0
    assert 0.5 < a.value < 100 and 0.5 < b.value < 1000
    sweep = (ctypes.c_double * 4)()
    assert L.ecgpu_selfcheck_ifetch_sweep(sweep) == 0
    assert all(0.5 < x < 1000 for x in sweep)


from tests._statevalue import fork_state_value as _fork_state_value  # noqa: E402


@pytest.mark.parametrize("fork", ["phase0", "altair", "bellatrix", "capella", "deneb", "electra"])
def test_beacon_state_root_of_every_fork(gpu, fork):
    """hash_tree_root(BeaconState) for phase0 / altair / bellatrix / capella / deneb, both presets,
    against the oracle's independent restatement of each fork's container (oracle/ssz.py BeaconState); the encodings are the
    oracle's own serializations of random states (phase0: with pending attestations whose bit lists end in every byte
    position), plus malformed encodings."""
    import random
    from ethereum_consensus_amd import synthetic
    from oracle import ssz as O
    ssz = gpu
    rnd = random.Random({"phase0": 1, "altair": 2, "bellatrix": 3, "capella": 4, "deneb": 5, "electra": 6}[fork])
    for preset_name, preset, n in (("minimal", ssz.MINIMAL, 37), ("mainnet", ssz.MAINNET, 300), ("minimal", ssz.MINIMAL, 0)):
        f = synthetic.state_fields(n, preset_name, seed=rnd.randrange(1000), extra_data=rnd.randbytes(rnd.choice([0, 5, 32])))
        f["_preset"] = preset_name
        t, v = _fork_state_value(fork, f, rnd)
        enc = t.serialize(v)
        assert ssz.hash_tree_root_beacon_state(fork, enc, preset) == t.htr(v), (fork, preset_name, n)
        assert len(enc) >= ssz._lib.load().ecgpu_beacon_state_fixed_size(ssz.FORKS[fork], preset) > 0
        if fork == "deneb":
            assert ssz.hash_tree_root_beacon_state_deneb(enc, preset) == t.htr(v)
    # malformed: truncated below the fixed part; a first offset that does not match; (bellatrix+) a wrong extra_data offset
    with pytest.raises(ssz.MerkleizationError):
        ssz.hash_tree_root_beacon_state(fork, enc[:1000], ssz.MINIMAL)
    bad = bytearray(enc)
    fixed = ssz._lib.load().ecgpu_beacon_state_fixed_size(ssz.FORKS[fork], ssz.MINIMAL)
    pos = bytes(enc).find(fixed.to_bytes(4, "little"))
    bad[pos] ^= 1
    with pytest.raises(ssz.MerkleizationError):
        ssz.hash_tree_root_beacon_state(fork, bytes(bad), ssz.MINIMAL)
    if fork in ("bellatrix", "capella", "deneb", "electra"):
        hdr_fixed = {"bellatrix": 536, "capella": 568, "deneb": 584, "electra": 648}[fork]
        t = O.BeaconState(fork, O.MINIMAL)
        names = [n for n, _ in t.fields]
        # the payload header starts at the offset stored in its slot of the fixed part
        off_pos = sum((ft.fixed_size if ft.fixed_size is not None else 4) for _, ft in t.fields[:names.index("latest_execution_payload_header")])
        h = int.from_bytes(enc[off_pos:off_pos + 4], "little")
        assert int.from_bytes(enc[h + 436:h + 440], "little") == hdr_fixed
        bad = bytearray(enc)
        bad[h + 436] ^= 4
        with pytest.raises(ssz.MerkleizationError):
            ssz.hash_tree_root_beacon_state(fork, bytes(bad), ssz.MINIMAL)
        # the device entry (the encoding never visits the host) makes the same check on the device: a good encoding gives the
        # same root, the malformed one gives the poisoned root of include/ecgpu.h (32 x 0xFF) -- round-2 advisor: one entry
        # used to return a root for what the other rejected
        import torch
        L = ssz._lib.load()
        st = torch.cuda.current_stream().cuda_stream
        out = torch.zeros(32, dtype=torch.uint8, device="cuda")
        for blob, want in ((enc, t.htr(v)), (bytes(bad), b"\xff" * 32)):
            d = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            fixed_part = ctypes.create_string_buffer(bytes(blob[:fixed]), fixed)
            rc = L.ecgpu_htr_beacon_state_dev(ssz.FORKS[fork], d.data_ptr(), len(blob), fixed_part, ssz.MINIMAL, out.data_ptr(), st)
            assert rc == 0, (rc, L.ecgpu_last_error())
            torch.cuda.synchronize()
            assert bytes(out.cpu().numpy()) == want
            # ... and the checked form says so with a status next to the root (round-3 verdict item 8): 0, or ECGPU_ERR_BAD_ARG
            # -- the code the host entries and the resident state return for the same encoding
            status = torch.full((1,), 77, dtype=torch.int32, device="cuda")
            out.zero_()
            rc = L.ecgpu_htr_beacon_state_dev_checked(ssz.FORKS[fork], d.data_ptr(), len(blob), fixed_part, ssz.MINIMAL, out.data_ptr(),
                                                      status.data_ptr(), st)
            assert rc == 0, (rc, L.ecgpu_last_error())
            torch.cuda.synchronize()
            assert bytes(out.cpu().numpy()) == want and int(status.item()) == (0 if blob is enc else -3)


@pytest.mark.parametrize("fork", ["phase0", "altair", "bellatrix", "capella", "deneb", "electra"])
def test_device_entry_of_every_fork(gpu, fork):
    """ecgpu_htr_beacon_state_dev from phase0 (whose two PendingAttestation lists are copied back and planned on the host: one
    synchronisation, include/ecgpu.h) to electra: the root of a device-resident encoding equals the oracle's"""
    import random
    import torch
    from ethereum_consensus_amd import synthetic
    ssz = gpu
    L = ssz._lib.load()
    rnd = random.Random(40 + ssz.FORKS[fork])
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(32, dtype=torch.uint8, device="cuda")
    for preset_name, preset, n in (("minimal", ssz.MINIMAL, 37), ("mainnet", ssz.MAINNET, 9), ("minimal", ssz.MINIMAL, 0)):
        f = synthetic.state_fields(n, preset_name, seed=rnd.randrange(1000))
        f["_preset"] = preset_name
        t, v = _fork_state_value(fork, f, rnd)
        enc = t.serialize(v)
        fixed = L.ecgpu_beacon_state_fixed_size(ssz.FORKS[fork], preset)
        d = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
        h_fixed = ctypes.create_string_buffer(bytes(enc[:fixed]), fixed)
        out.zero_()
        rc = L.ecgpu_htr_beacon_state_dev(ssz.FORKS[fork], d.data_ptr(), len(enc), h_fixed, preset, out.data_ptr(), st)
        assert rc == 0, (rc, L.ecgpu_last_error())
        torch.cuda.synchronize()
        assert bytes(out.cpu().numpy()) == t.htr(v), (fork, preset_name, n)


def test_resident_state_of_an_older_fork(gpu):
    """a capella state kept resident: patches + roots equal the from-scratch roots of the patched encoding"""
    import random
    from ethereum_consensus_amd import synthetic
    from oracle import ssz as O
    ssz = gpu
    rnd = random.Random(77)
    f = synthetic.state_fields(500, "minimal", seed=3)
    f["_preset"] = "minimal"
    t, v = _fork_state_value("capella", f, rnd)
    enc = bytearray(t.serialize(v))
    st = ssz.ResidentBeaconStateDeneb(bytes(enc), ssz.MINIMAL, fork="capella")
    assert st.hash_tree_root() == t.htr(v)
    for _ in range(3):
        patches = []
        for _ in range(20):
            off = rnd.randrange(2736 + 200, len(enc) - 700)  # somewhere in the variable part, away from offset words
            b = rnd.randbytes(rnd.choice([1, 8]))
            patches.append((off, b))
        patches = sorted(dict(patches).items())
        ok = all(patches[i][0] + len(patches[i][1]) <= patches[i + 1][0] for i in range(len(patches) - 1))
        if not ok:
            continue
        try:
            st.patch(patches)
        except ssz.MerkleizationError:
            continue  # touched an offset word
        for off, b in patches:
            enc[off:off + len(b)] = b
        try:
            want = ssz.hash_tree_root_beacon_state("capella", bytes(enc), ssz.MINIMAL)
        except ssz.MerkleizationError:
            break  # a patch made the encoding itself invalid (e.g. a slashed byte is still a byte: never here)
        assert st.hash_tree_root() == want
    st.close()


def test_merkle_proof_of_a_chunk_array(gpu):
    """ssz.merkle_proof (ecgpu_merkle_proof): branches of chunks against the oracle's merkleize and is_valid_merkle_branch --
    with a limit, with a limit that is not a power of two, and with limit_chunks == 0 (the tree of the chunks themselves:
    the round-2 advisor found the Python wrapper sizing its buffer for depth 0 there)."""
    from ethereum_consensus_amd import ssz
    r = random.Random(77)
    for n, limit in ((1, 0), (2, 0), (5, 0), (64, 0), (100, 0), (5, 8), (5, 100), (33, 4096), (1, 1), (0, 16)):
        chunks = r.randbytes(32 * n)
        eff = limit or max(n, 1)
        depth = (eff - 1).bit_length()
        root = ossz.merkleize_chunks([chunks[32 * i:32 * i + 32] for i in range(n)], limit or None)
        for index in sorted({0, max(n - 1, 0), min(n, (1 << depth) - 1), (1 << depth) - 1}):
            branch = ssz.merkle_proof(chunks, limit, index)
            assert len(branch) == depth
            leaf = chunks[32 * index:32 * index + 32] if index < n else bytes(32)
            assert ossz.is_valid_merkle_branch(leaf, branch, depth, index, root), (n, limit, index)
            assert ssz.is_valid_merkle_branch(leaf, branch, depth, index, root)
    with pytest.raises(ssz.MerkleizationError):
        ssz.merkle_proof(bytes(32 * 9), 8, 0)   # more chunks than the limit
    with pytest.raises(ssz.MerkleizationError):
        ssz.merkle_proof(bytes(32 * 4), 0, 4)   # index outside the tree
    with pytest.raises(ssz.MerkleizationError):
        ssz.merkle_proof(bytes(33), 0, 0)


def test_proof_into_a_registry_of_more_than_65536_validators(gpu):
    """state.validators[i].exit_epoch-style proofs through the generic prover on a List[Validator, 2^40] of 70 001 records
    (round 2 refused sequences beyond 65 536 composite elements and made one host round trip per element): one plan per tree
    level now.  Checked without the (slow) Python prover: the witness root equals the registry root of the dedicated entry,
    the generalized index equals the oracle's, the leaf is the field's chunk, and the branch folds to the root."""
    from ethereum_consensus_amd import ssz_types as T
    from ethereum_consensus_amd import synthetic as syn
    ssz = gpu
    n = 70001
    enc = syn.validators(n, seed=9).tobytes()
    Validator_g = T.container(("public_key", T.bytevector(48)), ("withdrawal_credentials", T.Bytes32), ("effective_balance", T.uint64),
                              ("slashed", T.uint(8)), ("activation_eligibility_epoch", T.uint64), ("activation_epoch", T.uint64),
                              ("exit_epoch", T.uint64), ("withdrawable_epoch", T.uint64))
    reg_g = T.list_(Validator_g, 1 << 40)
    reg_o = ossz.SSZList(ossz.Validator, 1 << 40)
    want_root = ssz.hash_tree_root_validators(enc)
    for i, field, lo in ((0, "exit_epoch", 105), (n - 1, "effective_balance", 80), (65536, "withdrawable_epoch", 113), (40000, "withdrawal_credentials", 48)):
        leaf, branch, g, root = ssz.prove(reg_g, enc, [i, field])
        assert root == want_root
        assert g == ossz.generalized_index(reg_o, [i, field])
        size = 32 if field == "withdrawal_credentials" else 8
        assert leaf == enc[121 * i + lo:121 * i + lo + size].ljust(32, b"\0")
        depth = len(branch)
        assert depth == 40 + 1 + 3 and ossz.is_valid_merkle_branch(leaf, branch, depth, g - (1 << depth), root)
    # the length node of the same list, and an element past the end
    leaf, branch, g, root = ssz.prove(reg_g, enc, [ssz.LENGTH])
    assert leaf == n.to_bytes(32, "little") and len(branch) == 1 and ossz.is_valid_merkle_branch(leaf, branch, 1, 1, root) and root == want_root
    with pytest.raises(ssz.MerkleizationError):
        ssz.prove(reg_g, enc, [n, "exit_epoch"])


def test_proofs_and_generalized_indices(gpu):
    """`prove` / `generalized_index` over the generic SSZ description against the oracle's restatement
    (oracle/ssz.py prove), the reference's pinned indices (deneb/beacon_block.rs:139-154) and ecgpu_is_valid_merkle_branch
    with the index arithmetic of deneb/blob_sidecar.rs:56-63."""
    import random
    from ethereum_consensus_amd import ssz_types as T
    from tests import _sszrand
    ssz = gpu
    for p_gpu, p_or in ((T.MAINNET, ossz.BLOCK_MAINNET), (T.MINIMAL, ossz.BLOCK_MINIMAL)):
        body_g = T.BeaconBlockBodyDeneb(p_gpu)
        body_o = dict(ossz.BeaconBlockDeneb(p_or).fields)["body"]
        if p_gpu is T.MAINNET:
            idx = [ssz.generalized_index(body_g, ["blob_kzg_commitments"])] + [ssz.generalized_index(body_g, ["blob_kzg_commitments", i]) for i in range(6)]
            assert idx == [27, 221184, 221185, 221186, 221187, 221188, 221189]
        r = random.Random(11)
        for trial in range(3):
            v = _sszrand.random_value(body_o, r, fill=["full", None, None][trial])
            enc = body_o.serialize(v)
            paths = [["execution_payload"], ["eth1_data", "deposit_count"], ["attestations", ssz.LENGTH], ["graffiti"], ["randao_reveal"],
                     ["sync_aggregate", "sync_committee_bits", 5], ["execution_payload", "block_hash"], ["execution_payload", "extra_data", ssz.LENGTH],
                     ["execution_payload", "logs_bloom", 100]]
            if v["blob_kzg_commitments"]:
                paths += [["blob_kzg_commitments", 0], ["blob_kzg_commitments", len(v["blob_kzg_commitments"]) - 1]]
            if v["execution_payload"]["transactions"]:
                paths += [["execution_payload", "transactions", 0]]
            if v["attestations"]:
                paths += [["attestations", 0, "data", "target", "root"], ["attestations", len(v["attestations"]) - 1, "aggregation_bits", ssz.LENGTH]]
            if v["deposits"]:
                paths += [["deposits", 0, "proof", 32], ["deposits", 0, "data", "amount"]]
            for path in paths:
                opath = [ossz.LENGTH if p == ssz.LENGTH else p for p in path]
                want = ossz.prove(body_o, v, opath)
                got = ssz.prove(body_g, enc, path)
                assert got == want, path
                leaf, branch, g, root = got
                depth = g.bit_length() - 1
                assert g == ssz.generalized_index(body_g, path) and len(branch) == depth
                assert ssz.is_valid_merkle_branch(leaf, branch, depth, g - (1 << depth), root)
                bad = bytes([leaf[0] ^ 1]) + leaf[1:]
                assert not ssz.is_valid_merkle_branch(bad, branch, depth, g - (1 << depth), root)
    with pytest.raises(ssz.MerkleizationError):
        ssz.generalized_index(T.BeaconBlockBodyDeneb(T.MAINNET), ["graffiti", 40])
    with pytest.raises(ssz.MerkleizationError):
        ssz.prove(T.BeaconBlockBodyDeneb(T.MAINNET), enc, ["blob_kzg_commitments", 4095, 3])


def test_light_client_branches_of_a_beacon_state(gpu):
    """The light-client proofs of spec-tests/runners/light_client.rs:32-40 on a BeaconState: current / next sync committee
    and finalized_checkpoint -> root, from the field roots the state plan computes anyway; against the oracle on a small
    state and by verification against the state root at 2^17 validators."""
    from ethereum_consensus_amd import ssz_types as T
    from ethereum_consensus_amd import synthetic
    ssz = gpu
    f = synthetic.state_fields(37, "minimal", seed=4)
    t = ossz.BeaconStateDeneb(ossz.MINIMAL)
    v = oracle_state_value(f)
    enc = synthetic.serialize_state(f)
    names = [n for n, _ in t.fields]
    for name in ("current_sync_committee", "next_sync_committee", "finalized_checkpoint", "validators", "slot"):
        pos = names.index(name)
        leaf, branch, g, root = ssz.prove_beacon_state_field("deneb", enc, ssz.MINIMAL, pos)
        assert (leaf, branch, g, root) == ossz.prove(t, v, [name])
    # finalized_checkpoint -> root: the field's branch continued by a proof inside the 40-byte Checkpoint
    pos = names.index("finalized_checkpoint")
    cp_enc = ossz.Checkpoint.serialize(v["finalized_checkpoint"])
    leaf2, br2, g2, cp_root = ssz.prove(T.Checkpoint, cp_enc, ["root"])
    leaf, branch, g, root = ssz.prove_beacon_state_field("deneb", enc, ssz.MINIMAL, pos)
    assert cp_root == leaf
    full = ossz.prove(t, v, ["finalized_checkpoint", "root"])
    assert (leaf2, br2 + branch, (g << 1) + (g2 - 2), root) == full and full[2] == 105
    # mainnet size: branches verify against the state root
    f = synthetic.state_fields(1 << 17, "mainnet", seed=9)
    enc = synthetic.serialize_state(f)
    want_root = ssz.hash_tree_root_beacon_state_deneb(enc, ssz.MAINNET)
    for pos in (22, 23, 20, 11):
        leaf, branch, g, root = ssz.prove_beacon_state_field("deneb", enc, ssz.MAINNET, pos)
        assert root == want_root and g == 32 + pos and ssz.is_valid_merkle_branch(leaf, branch, 5, pos, root)


def test_resident_state_lists_change_length(gpu):
    """add_validator_to_registry (phase0/block_processing.rs:317-349) and the other length changes on a
    RESIDENT state -- appends (within the buffer's slack and beyond it), a truncation, patches before and after -- against
    from-scratch roots of the re-serialized state."""
    import numpy as np
    from ethereum_consensus_amd import synthetic
    ssz = gpu
    f = synthetic.state_fields(100, "minimal", seed=12)
    enc = synthetic.serialize_state(f)
    st = ssz.ResidentBeaconStateDeneb(enc, ssz.MINIMAL)
    assert st.hash_tree_root() == ssz.hash_tree_root_beacon_state_deneb(enc, ssz.MINIMAL) and len(st) == len(enc)

    def grow(k, seed):
        new = synthetic.validators(k, seed=seed)
        f["validators"] = np.concatenate([f["validators"], new])
        f["balances"] = np.concatenate([f["balances"], np.full(k, 32 * 10**9 + seed, dtype="<u8")])
        for name in ("previous_epoch_participation", "current_epoch_participation"):
            f[name] = np.concatenate([f[name], np.zeros(k, dtype=np.uint8)])
        f["inactivity_scores"] = np.concatenate([f["inactivity_scores"], np.zeros(k, dtype="<u8")])
        return new

    for step, k in enumerate((1, 1, 7)):  # one validator at a time, as a deposit does
        new = grow(k, 100 + step)
        for i in range(k):
            st.add_validator(new[i:i + 1].tobytes(), 32 * 10**9 + 100 + step)
        enc = synthetic.serialize_state(f)
        assert len(st) == len(enc)
        assert st.hash_tree_root() == ssz.hash_tree_root_beacon_state_deneb(enc, ssz.MINIMAL), step
    # a patch addressed in the NEW encoding: the balance of the last validator
    n = len(f["validators"])
    tail = sum(len(x) for x in (f["balances"].tobytes(), f["previous_epoch_participation"].tobytes(), f["current_epoch_participation"].tobytes(),
                                f["inactivity_scores"].tobytes(), synthetic.serialize_payload_header(f["payload_header"]),
                                f["historical_summaries"].tobytes()))
    bal_off = len(enc) - tail
    f["balances"][n - 1] = 31 * 10**9
    st.patch([(bal_off + 8 * (n - 1), int(31 * 10**9).to_bytes(8, "little"))])
    # beyond the slack of the device buffer: 1 500 validators in one append per list
    new = grow(1500, 7)
    st.append(st.VALIDATORS, new.tobytes())
    st.append(st.BALANCES, f["balances"][-1500:].tobytes())
    st.append(st.PREVIOUS_EPOCH_PARTICIPATION, bytes(1500))
    st.append(st.CURRENT_EPOCH_PARTICIPATION, bytes(1500))
    st.append(st.INACTIVITY_SCORES, bytes(8 * 1500))
    # the eth1_data_votes reset and one more historical summary
    f["eth1_data_votes"] = []
    st.truncate(st.ETH1_DATA_VOTES, 0)
    extra = np.frombuffer(bytes(range(64)), dtype=np.uint8).reshape(1, 64)
    f["historical_summaries"] = np.concatenate([f["historical_summaries"], extra])
    st.append(st.HISTORICAL_SUMMARIES, extra.tobytes())
    enc = synthetic.serialize_state(f)
    assert len(st) == len(enc)
    assert st.hash_tree_root() == ssz.hash_tree_root_beacon_state_deneb(enc, ssz.MINIMAL)
    with pytest.raises(ssz.MerkleizationError):
        st.append(st.VALIDATORS, bytes(120))  # not a whole record
    with pytest.raises(ssz.MerkleizationError):
        st.append(7, bytes(8))                 # the payload header is not a list
    st.close()


# ---- the resident state's trees re-hash dirty paths only (csrc/state_tree.h) --------------------------
VAR_INDEX = {"historical_roots": 0, "eth1_data_votes": 1, "validators": 2, "balances": 3, "previous_epoch_participation": 4,
             "current_epoch_participation": 5, "inactivity_scores": 6, "historical_summaries": 8, "pending_balance_deposits": 9,
             "pending_partial_withdrawals": 10, "pending_consolidations": 11}
ELEM = {"historical_roots": 32, "eth1_data_votes": 72, "validators": 121, "balances": 8, "previous_epoch_participation": 1,
        "current_epoch_participation": 1, "inactivity_scores": 8, "historical_summaries": 64, "pending_balance_deposits": 16,
        "pending_partial_withdrawals": 24, "pending_consolidations": 16}


def ssz_resident():
    from ethereum_consensus_amd import ssz
    return ssz.ResidentBeaconStateDeneb


def _random_step(r, st, m, fork):
    """one randomly chosen mutation applied to the resident state `st` and to the host model `m`; returns its name"""
    from ethereum_consensus_amd import synthetic
    lists = [n for n in m.names if n in VAR_INDEX]
    op = r.choice(["patch"] * 6 + ["add_validator", "add_validator", "append", "truncate", "rewrite", "rotate", "fixed", "nothing"])
    if fork == "phase0" and op == "rotate":
        op = r.choice(["attest", "attest", "rotate_attestations"])
    if op == "attest":  # process_attestation: a PendingAttestation pushed onto current_epoch_attestations (the list re-serialized)
        from oracle import ssz as O
        t = O.SSZList(O.PendingAttestation(), 4096 if m.preset == "mainnet" else 1024)
        cur = m.att["current_epoch_attestations"]
        for _ in range(r.choice([1, 1, 4])):
            cur.append({"aggregation_bits": [r.random() < 0.5 for _ in range(r.choice([0, 1, 8, 9, 64, 333]))],
                        "data": {"slot": r.randrange(1 << 40), "index": r.randrange(64), "beacon_block_root": r.randbytes(32),
                                 "source": {"epoch": r.randrange(1 << 30), "root": r.randbytes(32)},
                                 "target": {"epoch": r.randrange(1 << 30), "root": r.randbytes(32)}},
                        "inclusion_delay": r.randrange(1, 33), "proposer_index": r.randrange(1 << 20)})
        enc = t.serialize(cur)
        st.replace(ssz_resident().CURRENT_EPOCH_ATTESTATIONS, enc)
        m.var["current_epoch_attestations"] = bytearray(enc)
        return op
    if op == "rotate_attestations":  # process_participation_record_updates: previous = current, current = []
        st.replace(ssz_resident().PREVIOUS_EPOCH_ATTESTATIONS, bytes(m.var["current_epoch_attestations"]))
        st.replace(ssz_resident().CURRENT_EPOCH_ATTESTATIONS, b"")
        m.var["previous_epoch_attestations"] = bytearray(m.var["current_epoch_attestations"])
        m.var["current_epoch_attestations"] = bytearray()
        m.att["previous_epoch_attestations"], m.att["current_epoch_attestations"] = m.att["current_epoch_attestations"], []
        return op
    if op == "patch":  # a block's worth of small writes: balances, flags, scores, records, roots
        patches, used = [], set()
        for _ in range(r.choice([1, 3, 40, 400])):
            name = r.choice(lists)
            if not m.var[name]:
                continue
            ln = r.choice([1, 8]) if ELEM[name] != 1 else 1
            if name == "validators":
                ln = r.choice([1, 8, 32, 121, 130])
            ln = min(ln, len(m.var[name]))
            off = m.start(name) + r.randrange(0, len(m.var[name]) - ln + 1)
            if any(b in used for b in range(off, off + ln)):
                continue
            used.update(range(off, off + ln))
            patches.append((off, r.randbytes(ln)))
        st.patch(patches)
        for off, b in patches:
            m.write(off, b)
    elif op == "fixed":  # slot, a block root, a state root, a randao mix, a slashing: the fixed part's big vectors and basic fields
        patches = []
        for name in r.sample(["slot", "block_roots", "state_roots", "randao_mixes", "slashings", "latest_block_header"], 3):
            lo, hi = m.fixed_ranges[name]
            ln = 8 if name in ("slot", "slashings") else 32
            off = lo + ln * r.randrange((hi - lo) // ln)
            patches.append((off, r.randbytes(ln)))
        st.patch(patches)
        for off, b in patches:
            m.write(off, b)
    elif op == "add_validator":  # a deposit
        for _ in range(r.choice([1, 1, 2, 9])):
            rec = synthetic.validators(1, seed=r.randrange(1 << 30)).tobytes()
            bal = r.randrange(1 << 40)
            st.add_validator(rec, bal)
            m.var["validators"] += rec
            m.var["balances"] += bal.to_bytes(8, "little")
            if fork == "phase0":
                continue
            m.var["previous_epoch_participation"] += b"\x00"
            m.var["current_epoch_participation"] += b"\x00"
            m.var["inactivity_scores"] += bytes(8)
    elif op == "append":  # an eth1 vote per block, a summary / root per period
        name = r.choice([n for n in ("eth1_data_votes", "historical_roots", "historical_summaries", "pending_balance_deposits",
                                     "pending_partial_withdrawals", "pending_consolidations") if n in m.var])
        data = r.randbytes(ELEM[name] * r.choice([1, 1, 3]))
        if len(m.var[name]) + len(data) <= ELEM[name] * min(getattr(m, "limits", {}).get(name, 1 << 20), 32 if name == "eth1_data_votes" else 1 << 20):
            st.append(VAR_INDEX[name], data)
            m.var[name] += data
    elif op == "truncate":  # the eth1_data_votes reset; a list cut to a shorter length
        name = r.choice([n for n in ("eth1_data_votes", "historical_roots", "balances", "pending_balance_deposits", "pending_consolidations")
                         if n in m.var])
        if name == "balances":
            return op  # (balances never shrink on their own: only together with the registry -- not modelled)
        keep = ELEM[name] * r.randrange(0, len(m.var[name]) // ELEM[name] + 1)
        st.truncate(VAR_INDEX[name], keep)
        del m.var[name][keep:]
    elif op == "rewrite":  # an epoch's rewards: every balance changes in one patch (the tree is rebuilt, not climbed)
        name = r.choice(["balances", "inactivity_scores"] if fork != "phase0" else ["balances"])
        data = r.randbytes(len(m.var[name]))
        if data:
            st.patch([(m.start(name), data)])
            m.write(m.start(name), data)
    elif op == "rotate":  # the participation rotation of an epoch boundary
        cur = bytes(m.var["current_epoch_participation"])
        if cur:
            st.patch([(m.start("previous_epoch_participation"), cur), (m.start("current_epoch_participation"), bytes(len(cur)))])
            m.write(m.start("previous_epoch_participation"), cur)
            m.write(m.start("current_epoch_participation"), bytes(len(cur)))
    return op


@pytest.mark.parametrize("fork,preset,n_val,steps", [("phase0", "minimal", 600, 170), ("phase0", "mainnet", 3000, 120),
                                                      ("altair", "minimal", 700, 170), ("bellatrix", "minimal", 1100, 170),
                                                      ("capella", "mainnet", 2500, 170), ("deneb", "minimal", 37, 170),
                                                      ("deneb", "mainnet", 5000, 170), ("deneb", "minimal", 2040, 170),
                                                      ("electra", "minimal", 900, 170),
                                                      ("capella", "mainnet", 300_000, 30)])  # (trees of height 19: 1 024-entry regions, several active per patch set)
def test_resident_state_randomised_patch_append_truncate_sequences(gpu, fork, preset, n_val, steps):
    """1 510 randomised steps over every resident fork (phase0 ... electra; phase0 with attestations pushed and rotated): after EVERY step the resident root (dirty paths climbed, rebuilt fields,
    finishing jobs over the cached levels) equals ecgpu_htr_beacon_state of the re-serialized state, computed from scratch."""
    from ethereum_consensus_amd import synthetic
    from tests._statemodel import EncodingModel
    ssz = gpu
    import zlib
    r = random.Random(zlib.crc32(f"{fork}/{preset}/{n_val}".encode()))
    f = synthetic.state_fields(n_val, preset, seed=n_val)
    f["_preset"] = preset
    t, v = _fork_state_value(fork, f, r)
    enc = t.serialize(v)
    pid = ssz.MINIMAL if preset == "minimal" else ssz.MAINNET
    st = ssz.ResidentBeaconStateDeneb(enc, pid, fork=fork)
    m = EncodingModel(t, enc)
    m.limits = {n: ty.limit for n, ty in t.fields if hasattr(ty, "limit")}  # (an append past a list's limit is an error, tested elsewhere)
    m.preset = preset
    m.att = {k: list(v[k]) for k in ("previous_epoch_attestations", "current_epoch_attestations") if k in v}
    assert m.encoding() == enc
    assert st.hash_tree_root() == t.htr(v)
    seen = set()
    for k in range(steps):
        op = _random_step(r, st, m, fork)
        seen.add(op)
        if r.random() < 0.25:
            continue  # several mutations between two roots
        cur = m.encoding()
        assert len(st) == len(cur), (k, op)
        assert st.hash_tree_root() == ssz.hash_tree_root_beacon_state(fork, cur, pid), (k, op)
    cur = m.encoding()
    assert st.hash_tree_root() == ssz.hash_tree_root_beacon_state(fork, cur, pid)
    assert {"patch", "add_validator", "rewrite", "fixed"} <= seen and ({"rotate"} <= seen or {"attest", "rotate_attestations"} <= seen)
    st.close()


def test_resident_phase0_state_attestation_lists_change_by_replacement_only(gpu):
    """phase0/beacon_state.rs:80-81: the two lists of variable-size PendingAttestation.  A patch reaching into them, an append,
    and a malformed replacement are refused and leave the state as it was; a replacement is rooted inside the call."""
    from ethereum_consensus_amd import synthetic
    from oracle import ssz as O
    ssz = gpu
    R = ssz.ResidentBeaconStateDeneb
    r = random.Random(5)
    f = synthetic.state_fields(200, "minimal", seed=8)
    f["_preset"] = "minimal"
    t, v = _fork_state_value("phase0", f, r)
    enc = t.serialize(v)
    st = R(enc, ssz.MINIMAL, fork="phase0")
    root0 = st.hash_tree_root()
    assert root0 == t.htr(v)
    with pytest.raises(Exception):
        st.patch([(len(enc) - 4, b"\x01\x02\x03\x04")])
    with pytest.raises(Exception):
        st.append(R.CURRENT_EPOCH_ATTESTATIONS, bytes(148))
    lt = O.SSZList(O.PendingAttestation(), 1024)
    good = lt.serialize(v["current_epoch_attestations"] + v["current_epoch_attestations"][:1])
    with pytest.raises(Exception):
        st.replace(R.CURRENT_EPOCH_ATTESTATIONS, good[:-3] if len(good) > 8 else b"\x07")  # a truncated serialization
    assert len(st) == len(enc) and st.hash_tree_root() == root0
    st.replace(R.CURRENT_EPOCH_ATTESTATIONS, good)
    v["current_epoch_attestations"] = v["current_epoch_attestations"] + v["current_epoch_attestations"][:1]
    assert st.hash_tree_root() == t.htr(v)
    st.replace(R.PREVIOUS_EPOCH_ATTESTATIONS, good)
    st.replace(R.CURRENT_EPOCH_ATTESTATIONS, b"")
    v["previous_epoch_attestations"], v["current_epoch_attestations"] = v["current_epoch_attestations"], []
    assert st.hash_tree_root() == t.htr(v) and len(st) == len(t.serialize(v))
    st.replace(R.BALANCES, bytes(8 * 200))  # lists of fixed-size elements replaced whole: same length, another length
    v["balances"] = [0] * 200
    st.replace(R.HISTORICAL_ROOTS, bytes(range(96)))
    v["historical_roots"] = [bytes(range(32 * i, 32 * i + 32)) for i in range(3)]
    assert st.hash_tree_root() == t.htr(v)
    st.close()


def test_resident_electra_state_refuses_an_append_past_a_list_limit_and_stays_as_it_was(gpu):
    """electra/beacon_state.rs:73-145 (minimal preset: PENDING_PARTIAL_WITHDRAWALS_LIMIT = 64): the 65th record is refused
    before anything moves; the root and the size afterwards are those before the call, and the next legal append works."""
    from ethereum_consensus_amd import synthetic
    ssz = gpu
    r = random.Random(77)
    f = synthetic.state_fields(300, "minimal", seed=3)
    f["_preset"] = "minimal"
    t, v = _fork_state_value("electra", f, r)
    v["pending_partial_withdrawals"] = [{"index": i, "amount": 5, "withdrawable_epoch": 9} for i in range(63)]
    enc = t.serialize(v)
    st = ssz.ResidentBeaconStateDeneb(enc, ssz.MINIMAL, fork="electra")
    root0 = st.hash_tree_root()
    assert root0 == t.htr(v)
    with pytest.raises(Exception):
        st.append(ssz.ResidentBeaconStateDeneb.PENDING_PARTIAL_WITHDRAWALS, bytes(48))  # 63 + 2 > 64
    assert len(st) == len(enc) and st.hash_tree_root() == root0
    st.append(ssz.ResidentBeaconStateDeneb.PENDING_PARTIAL_WITHDRAWALS, (7).to_bytes(8, "little") * 3)
    v["pending_partial_withdrawals"].append({"index": 7, "amount": 7, "withdrawable_epoch": 7})
    assert st.hash_tree_root() == t.htr(v)
    st.truncate(ssz.ResidentBeaconStateDeneb.PENDING_PARTIAL_WITHDRAWALS, 24)  # the queue processed down to one record
    v["pending_partial_withdrawals"] = v["pending_partial_withdrawals"][:1]
    assert st.hash_tree_root() == t.htr(v)
    st.close()


def test_resident_root_after_a_blocks_patches_rehashes_dirty_paths_only(gpu):
    """BASELINE configs[4] at full size: a 2^20-validator mainnet state, 4 096 balances + 4 096 participation bytes patched.
    The root equals the from-scratch root and costs <= 150 k hash64 (from scratch: 10.1 M; round 4's validator-root cache:
    1.2 M) -- the count is taken on the device by the climbs themselves."""
    from ethereum_consensus_amd import synthetic, _lib
    ssz = gpu
    L = _lib.load()
    n = 1 << 20
    f = synthetic.state_fields(n, "mainnet", seed=5)
    enc = bytearray(synthetic.serialize_state(f))
    st = ssz.ResidentBeaconStateDeneb(bytes(enc), ssz.MAINNET)
    root0 = st.hash_tree_root()
    assert root0 == ssz.hash_tree_root_beacon_state_deneb(bytes(enc), ssz.MAINNET)
    fixed = int(L.ecgpu_beacon_state_deneb_fixed_size(ssz.MAINNET))
    vals_off = fixed + len(f["historical_roots"].tobytes()) + 72 * len(f["eth1_data_votes"])
    bal_off = vals_off + 121 * n
    part_off = bal_off + 8 * n + n  # current_epoch_participation
    r = random.Random(9)
    for rnd_ in range(3):
        idx = r.sample(range(n), 4096)
        patches = [(bal_off + 8 * i, r.randrange(1 << 40).to_bytes(8, "little")) for i in idx]
        patches += [(part_off + i, bytes([r.randrange(1, 8)])) for i in r.sample(range(n), 4096)]
        if rnd_ == 2:  # ... and a few registry records (an exit, a slashing): 8 + 11 hash64 each
            patches += [(vals_off + 121 * i + 88, b"\x01") for i in r.sample(range(n), 16)]
        for off, b in patches:
            enc[off:off + len(b)] = b
        st.patch(patches)
        root = st.hash_tree_root()
        hashes = int(L.ecgpu_last_hash64_count())
        assert root == ssz.hash_tree_root_beacon_state_deneb(bytes(enc), ssz.MAINNET)
        assert hashes <= 150_000, hashes
        assert hashes >= 4096 * 5  # (it did climb)
    # nothing dirty: the root costs the finishing jobs and the small fields only
    assert st.hash_tree_root() == root
    assert int(L.ecgpu_last_hash64_count()) <= 12_000
    st.close()


# ---- round 6: field-addressed entries (csrc/state_fields.h) on the device, anchored to the oracle at the VALUE level -----------------
def _fresh_state(fork, preset_name, n, seed):
    from ethereum_consensus_amd import synthetic
    r = random.Random(seed)
    f = synthetic.state_fields(n, preset_name, seed=seed, extra_data=r.randbytes(r.choice([0, 5, 32])))
    f["_preset"] = preset_name
    t, v = _fork_state_value(fork, f, r)
    return t, {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in v.items()}


@pytest.mark.parametrize("fork,preset,n_val,steps", [("phase0", "minimal", 300, 150), ("altair", "minimal", 700, 170), ("bellatrix", "minimal", 1100, 150),
                                                      ("capella", "mainnet", 2500, 90), ("deneb", "minimal", 37, 200), ("deneb", "minimal", 2040, 170),
                                                      ("deneb", "mainnet", 5000, 90), ("electra", "minimal", 900, 170)])
def test_resident_state_follows_the_state_transition_against_the_oracle(gpu, fork, preset, n_val, steps):
    """VERDICT round 5, 1(b) + 2: the resident root after every batch of mutations equals oracle/ssz.py's hash_tree_root of the
    VALUE (tests/_statefields.py mutates the oracle value the way the reference mutates its struct and tells the resident state
    the same in (field, index) coordinates through ecgpu_resident_state_patch_field / _patch_elements / _push / _truncate_field /
    _set_field / _add_validator / _rotate_participation).  Neither the product's from-scratch kernels nor a byte model of the
    encoding stand between the dirty-path climbs and the oracle here."""
    import zlib
    from tests import _statefields as SF
    ssz = gpu
    r = random.Random(zlib.crc32(f"fields/{fork}/{preset}/{n_val}".encode()))
    t, v = _fresh_state(fork, preset, n_val, seed=n_val + 1)
    pid = ssz.MINIMAL if preset == "minimal" else ssz.MAINNET
    st = ssz.ResidentBeaconStateDeneb(t.serialize(v), pid, fork=fork)
    assert st.hash_tree_root() == t.htr(v)
    seen = set()
    for k in range(steps):
        op = SF.random_step(r, st, t, v, fork, preset)
        seen.add(op)
        if r.random() < 0.4:
            continue  # several mutations between two roots
        assert st.hash_tree_root() == t.htr(v), (k, op)
        if k % 16 == 0:
            assert len(st) == len(t.serialize(v)), (k, op)
    assert st.hash_tree_root() == t.htr(v)
    assert ssz.hash_tree_root_beacon_state(fork, t.serialize(v), pid) == t.htr(v)
    assert {"balance", "slot"} <= seen and seen & {"deposit", "deposit_then_balance"} and len(seen - {None}) >= 12, seen
    st.close()


def test_resident_state_deposit_then_balance_and_votes_across_a_reset(gpu):
    """the two scenarios in which round 4/5's never-compiled Rust StateMirror computed wrong offsets (ADVICE rounds 4 and 5),
    driven through the C ABI: (1) deposits, then balance / flag / record writes to old AND new validators in the same slot;
    (2) an eth1 vote per block across the voting-period reset, a balance write and a root every block."""
    from oracle import ssz as O
    from tests import _statefields as SF
    ssz = gpu
    r = random.Random(11)
    t, v = _fresh_state("deneb", "minimal", 500, seed=3)
    st = ssz.ResidentBeaconStateDeneb(t.serialize(v), ssz.MINIMAL)
    assert st.hash_tree_root() == t.htr(v)
    for slot in range(4):
        for _ in range(3):
            rec = SF.random_validator(r)
            v["validators"].append(rec)
            v["balances"].append(32 * 10**9)
            for name in ("previous_epoch_participation", "current_epoch_participation", "inactivity_scores"):
                v[name].append(0)
            st.add_validator(O.Validator.serialize(rec), 32 * 10**9)
        m = len(v["validators"])
        for i in (7, m - 1, m - 3, r.randrange(m)):
            v["balances"][i] = r.randrange(1 << 40)
            st.patch_elements("balances", i, v["balances"][i].to_bytes(8, "little"))
        v["current_epoch_participation"][m - 2] = 5
        st.patch_elements("current_epoch_participation", m - 2, b"\x05")
        v["validators"][m - 1] = dict(v["validators"][m - 1], slashed=True)
        st.patch_field("validators", 121 * (m - 1) + 88, b"\x01")
        assert st.field_size("balances") == 8 * m
        assert st.hash_tree_root() == t.htr(v), slot
    st.close()
    t, v = _fresh_state("capella", "minimal", 400, seed=9)
    v["eth1_data_votes"] = []
    st = ssz.ResidentBeaconStateDeneb(t.serialize(v), ssz.MINIMAL, fork="capella")
    for period in range(2):
        for blk in range(32):
            e = {"deposit_root": r.randbytes(32), "deposit_count": blk, "block_hash": r.randbytes(32)}
            v["eth1_data_votes"].append(e)
            st.push("eth1_data_votes", O.Eth1Data.serialize(e))
            i = r.randrange(400)
            v["balances"][i] = r.randrange(1 << 40)
            st.patch_elements("balances", i, v["balances"][i].to_bytes(8, "little"))
            assert st.hash_tree_root() == t.htr(v), (period, blk)
        with pytest.raises(ssz.MerkleizationError):
            st.push("eth1_data_votes", bytes(72))  # the 33rd vote of a 32-slot period
        v["eth1_data_votes"] = []
        st.truncate_field("eth1_data_votes", 0)
        v["balances"][0] = period
        st.patch_elements("balances", 0, period.to_bytes(8, "little"))
        assert st.hash_tree_root() == t.htr(v), period
    # refused calls leave the state as it was
    for call in (lambda: st.patch_elements("balances", 400, bytes(8)), lambda: st.patch_elements("balances", 0, bytes(7)),
                 lambda: st.push("slot", bytes(8)), lambda: st.set_field("slot", bytes(4)), lambda: st.patch_elements(34, 0, bytes(16)),
                 lambda: st.set_field("latest_execution_payload_header", bytes(700)), lambda: st.truncate_field("balances", 8 * 401)):
        with pytest.raises(ssz.MerkleizationError):
            call()
    assert st.hash_tree_root() == t.htr(v)
    st.close()


def test_resident_state_field_writes_at_2_pow_20_validators_against_the_c_oracle(gpu):
    """BASELINE configs[4] at full size in field coordinates: 4 096 balances + 4 096 participation flags + 16 exits + 2 deposits per
    slot on a 2^20-validator mainnet state; expected roots from the C restatement (oracle/c) over the VALUE arrays -- not from
    the product's from-scratch kernels."""
    import numpy as np
    from ethereum_consensus_amd import synthetic, _lib
    ssz = gpu
    L = _lib.load()
    n = (1 << 20) - 6  # (the six deposits below fill the registry's tree of height 20 exactly; a 2^20 + 1st validator makes it one level taller: a rebuild)
    f = synthetic.state_fields(n, "mainnet", seed=6)
    st = ssz.ResidentBeaconStateDeneb(synthetic.serialize_state(f), ssz.MAINNET)
    assert st.hash_tree_root() == oracle_state_root_fast(f, "mainnet")
    r = random.Random(10)
    for slot in range(3):
        m = len(f["validators"])
        for i in r.sample(range(m), 4096):
            f["balances"][i] = r.randrange(1 << 40)
            st.patch_elements("balances", i, int(f["balances"][i]).to_bytes(8, "little"))
        for i in r.sample(range(m), 4096):
            f["current_epoch_participation"][i] = r.randrange(1, 8)
            st.patch_elements("current_epoch_participation", i, bytes([int(f["current_epoch_participation"][i])]))
        for i in r.sample(range(m), 16):  # initiate_validator_exit: exit_epoch, withdrawable_epoch
            f["validators"][i]["exit_epoch"] = 300_000 + slot
            f["validators"][i]["withdrawable_epoch"] = 300_256 + slot
            st.patch_field("validators", 121 * i + 105, (300_000 + slot).to_bytes(8, "little") + (300_256 + slot).to_bytes(8, "little"))
        new = synthetic.validators(2, seed=900 + slot)
        for k in range(2):
            st.add_validator(new[k:k + 1].tobytes(), 32 * 10**9 + k)
        f["validators"] = np.concatenate([f["validators"], new])
        f["balances"] = np.concatenate([f["balances"], np.array([32 * 10**9, 32 * 10**9 + 1], dtype="<u8")])
        for name in ("previous_epoch_participation", "current_epoch_participation"):
            f[name] = np.concatenate([f[name], np.zeros(2, dtype=np.uint8)])
        f["inactivity_scores"] = np.concatenate([f["inactivity_scores"], np.zeros(2, dtype="<u8")])
        f["balances"][m + 1] = 5  # ... and a write to the validator just deposited
        st.patch_elements("balances", m + 1, (5).to_bytes(8, "little"))
        f["slot"] += 1
        st.patch_elements("slot", 0, int(f["slot"]).to_bytes(8, "little"))
        root = st.hash_tree_root()
        hashes = int(L.ecgpu_last_hash64_count())
        assert root == oracle_state_root_fast(f, "mainnet"), slot
        assert hashes <= 160_000, hashes  # dirty paths only
    st.close()


def test_resident_state_handed_between_threads_keeps_host_order(gpu):
    """a state may be handed from one host thread to another -- each has its own stream -- and every
    change / root must then see the host's order without a host synchronisation in between: patches return as soon as their
    bytes are staged.  Two threads alternate on ONE state (never at once): A writes balances and flushes (asynchronous), B at
    once pushes a validator, writes the new validator's balance and roots; then the roles swap.  Every root == oracle."""
    import threading
    from oracle import ssz as O
    from tests import _statefields as SF
    ssz = gpu
    r = random.Random(31)
    t, v = _fresh_state("deneb", "minimal", 800, seed=12)
    st = ssz.ResidentBeaconStateDeneb(t.serialize(v), ssz.MINIMAL)
    assert st.hash_tree_root() == t.htr(v)
    turn = threading.Semaphore(0), threading.Semaphore(0)
    errors = []

    def worker(me):
        try:
            for step in range(40):
                turn[me].acquire()
                n = len(v["validators"])
                if step % 2 == me:  # writer of this step: many small writes, flushed, NOT synchronised
                    for _ in range(300):
                        i = r.randrange(n)
                        v["balances"][i] = r.randrange(1 << 40)
                        st.patch_elements("balances", i, v["balances"][i].to_bytes(8, "little"))
                    st.flush()
                else:  # the other thread carries on at once: a length change, a write behind it, a root
                    rec = SF.random_validator(r)
                    v["validators"].append(rec)
                    v["balances"].append(7)
                    for name in ("previous_epoch_participation", "current_epoch_participation", "inactivity_scores"):
                        v[name].append(0)
                    st.add_validator(O.Validator.serialize(rec), 7)
                    v["balances"][n] = 9
                    st.patch_elements("balances", n, (9).to_bytes(8, "little"))
                    if st.hash_tree_root() != t.htr(v):
                        errors.append((me, step))
                turn[1 - me].release()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            turn[1 - me].release()

    th = [threading.Thread(target=worker, args=(k,), daemon=True) for k in (0, 1)]
    for x in th:
        x.start()
    turn[0].release()
    for x in th:
        x.join(timeout=600)
    assert not errors, errors[:5]
    assert st.hash_tree_root() == t.htr(v)
    st.close()
