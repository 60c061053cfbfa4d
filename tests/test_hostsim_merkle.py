"""The gfx950 Merkle lane programs (csrc/sha256.h, csrc/merkle.h), compiled for the host by
tests/hostsim, against oracle/ssz.py.  Same source the GPU runs; CPU only."""
import ctypes
import hashlib
import random

import pytest

from oracle import ssz
from tests import _hostsim as hs


def rnd(n, seed):
    return random.Random(seed).randbytes(n)


def test_zero_table_and_hash64():
    L = hs.lib()
    out = ctypes.create_string_buffer(65 * 32)
    L.hs_zero_table(out)
    for d in range(65):
        assert out.raw[32 * d: 32 * d + 32] == ssz.ZERO_HASHES[d]
    a, b = rnd(32, 1), rnd(32, 2)
    o = ctypes.create_string_buffer(32)
    L.hs_hash64(a, b, o)
    assert o.raw == hashlib.sha256(a + b).digest()


@pytest.mark.parametrize("n", [0, 1, 3, 55, 56, 63, 64, 65, 119, 120, 200, 1000])
def test_sha256_stream(n):
    L = hs.lib()
    data = rnd(n, n)
    o = ctypes.create_string_buffer(32)
    L.hs_sha256(data, n, o)
    assert o.raw == hashlib.sha256(data).digest()


@pytest.mark.parametrize("nbytes,limit,ds", [
    (32, 1, [0]), (64, 2, [1]), (33, 2, [1]), (5 * 32, 8, [1]), (5 * 32, 8, [3]), (5 * 32, 8, [2, 1]),
    (1000, 1 << 20, [3, 1]), (32 * 777, 1 << 38, [2, 2, 1]), (32 * 1024, 1024, [6, 1]), (32 * 1025, 2048, [4, 3]),
    (8 * 4097, 1 << 38, [5]), (1, 1 << 35, [1]),
])
@pytest.mark.parametrize("misalign", [0, 1, 3, 9])
def test_merkleize_chunks(nbytes, limit, ds, misalign):
    data = rnd(nbytes, nbytes * 7 + misalign)
    n0 = (nbytes + 31) // 32
    depth = ssz._depth_for(limit)
    for mix in (False, True):
        want = ssz.merkleize_bytes(data, limit)
        if mix:
            want = ssz.mix_in_length(want, nbytes // 8)
        got = hs.merkleize(0, data, n0, depth, mix, nbytes // 8, ds, misalign)
        assert got == want


def test_empty_trees():
    for limit in (0, 1, 8, 1 << 40):
        depth = ssz._depth_for(limit)
        assert hs.merkleize(0, b"", 0, depth, False, 0, [1]) == ssz.merkleize_chunks([], limit)
        assert hs.merkleize(0, b"", 0, depth, True, 0, [1]) == ssz.mix_in_length(ssz.merkleize_chunks([], limit), 0)


def make_validators(n, seed=0):
    r = random.Random(seed)
    vs = []
    for i in range(n):
        vs.append({
            "public_key": r.randbytes(48), "withdrawal_credentials": r.randbytes(32),
            "effective_balance": r.choice([32 * 10**9, 31 * 10**9, r.getrandbits(64)]),
            "slashed": r.random() < 0.3,
            "activation_eligibility_epoch": r.getrandbits(64), "activation_epoch": r.getrandbits(20),
            "exit_epoch": r.choice([2**64 - 1, r.getrandbits(30)]),
            "withdrawable_epoch": r.choice([2**64 - 1, r.getrandbits(64)]),
        })
    return vs


@pytest.mark.parametrize("n,ds", [(1, [1]), (2, [1]), (3, [2]), (4, [2]), (5, [2, 1]), (17, [0, 3]), (64, [3, 2]), (100, [2])])
@pytest.mark.parametrize("misalign", [0, 1, 2, 3])
def test_validator_list_root(n, ds, misalign):
    vs = make_validators(n, n)
    ser = b"".join(ssz.Validator.serialize(v) for v in vs)
    assert len(ser) == 121 * n
    want = ssz.SSZList(ssz.Validator, 1 << 40).htr(vs)
    got = hs.merkleize(2, ser, n, 40, True, n, ds, misalign)
    assert got == want


@pytest.mark.parametrize("misalign", range(16))
def test_staged_registry_wave_addressing(misalign):
    """The registry's leaf pass staged through LDS (csrc/merkle.hip k_merkle_pass<2, ValidatorLeaves>), on the host with the kernel's
    own addressing (merkle.h StagedRecord / staged_record / staged_root_dword): a wave's 256 records at every byte alignment --
    16-byte vectors into the stage, the spill vector only where the step is misaligned, words fetched through the funnel, the
    roots transposed word-major -- give the 64 nodes of the generic lane program, and never read what lies behind the records."""
    import ctypes
    import random
    from ethereum_consensus_amd import synthetic as S
    L = hs.lib()
    L.hs_staged_validator_wave.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    data = S.validators(256 + 3, seed=100 + misalign).tobytes()
    lead = random.Random(misalign).randbytes(16 + 121 * 2)  # (records in front of the wave: the aligned head reads into them)
    raw, addr = hs.unaligned_buffer(lead + data[:121 * 256], (misalign - len(lead)) % 16)
    out = ctypes.create_string_buffer(32 * 64)
    L.hs_staged_validator_wave(addr, len(lead), out)
    assert (addr + len(lead)) % 16 == misalign
    want = ctypes.create_string_buffer(32 * 64)
    raw2, addr2 = hs.unaligned_buffer(data[:121 * 256], 0)
    L.hs_pass(2, 2, addr2, 121 * 256, 256, want, 0)
    assert out.raw == want.raw


def test_bytes48_pair64_eth1data_leaves():
    r = random.Random(5)
    pks = [r.randbytes(48) for _ in range(33)]
    want = ssz.Vector(ssz.BlsPublicKey, 64).htr(pks + [bytes(48)] * 31)
    got = hs.merkleize(3, b"".join(pks) + bytes(48) * 31, 64, 6, False, 0, [3, 2], 1)
    assert got == want
    hsum = [{"block_summary_root": r.randbytes(32), "state_summary_root": r.randbytes(32)} for _ in range(7)]
    want = ssz.SSZList(ssz.HistoricalSummary, 1 << 24).htr(hsum)
    got = hs.merkleize(4, b"".join(ssz.HistoricalSummary.serialize(x) for x in hsum), 7, 24, True, 7, [2], 3)
    assert got == want
    votes = [{"deposit_root": r.randbytes(32), "deposit_count": r.getrandbits(64), "block_hash": r.randbytes(32)}
             for _ in range(11)]
    want = ssz.SSZList(ssz.Eth1Data, 2048).htr(votes)
    got = hs.merkleize(5, b"".join(ssz.Eth1Data.serialize(x) for x in votes), 11, 11, True, 11, [1, 2], 2)
    assert got == want


def test_scheduled_merkleize_counts_hashes_like_the_oracle():
    for nbytes, limit in [(32 * 5, 8), (32 * 700, 1 << 20), (32 * 3000, 4096), (8 * 1234, 1 << 38)]:
        data = rnd(nbytes, nbytes)
        n0 = (nbytes + 31) // 32
        got, h = hs.merkleize_scheduled(0, data, n0, ssz._depth_for(limit), True, 77, 1)
        assert got == ssz.mix_in_length(ssz.merkleize_bytes(data, limit), 77)
        assert h == ssz.hash64_count(n0, limit) + 1


@pytest.mark.parametrize("preset,n", [("minimal", 0), ("minimal", 1), ("minimal", 37), ("minimal", 700), ("mainnet", 5)])
def test_beacon_state_deneb_plan(preset, n):
    """The product's state plan (csrc/state_plan.h) executed on the lane simulator == oracle."""
    from ethereum_consensus_amd import synthetic as S
    from tests._statevalue import oracle_state_value
    f = S.state_fields(n, preset, seed=n + 3, n_votes=n % 7, n_hist_roots=n % 5, n_hist_summaries=n % 3,
                       extra_data=b"x" * (n % 33))
    enc = S.serialize_state(f)
    P = ssz.MINIMAL if preset == "minimal" else ssz.MAINNET
    t = ssz.BeaconStateDeneb(P)
    v = oracle_state_value(f)
    assert t.serialize(v) == enc
    rc, root, hashes = hs.state_root_deneb(enc, S.PRESETS[preset]["id"])
    assert rc == 0
    assert root == t.htr(v)
    assert hashes > 8 * n


@pytest.mark.parametrize("fork", ["altair", "bellatrix", "capella", "deneb", "electra"])
def test_beacon_state_plan_of_every_fork(fork):
    """csrc/state_plan.h for altair .. electra on the lane simulator == the oracle's container of that fork
    (oracle/ssz.py BeaconState; electra: 37 fields in a 64-leaf container, a 19-field payload header, three lists of two- /
    three-uint64 containers: electra/beacon_state.rs:73-145), both presets, empty / short / tile-sized pending lists."""
    import random
    from ethereum_consensus_amd import synthetic as S
    from tests._statevalue import fork_state_value as _fork_state_value
    fork_id = {"altair": 1, "bellatrix": 2, "capella": 3, "deneb": 4, "electra": 5}[fork]
    rnd = random.Random(100 + fork_id)
    for preset, n in (("minimal", 0), ("minimal", 37), ("mainnet", 5), ("minimal", 3), ("mainnet", 2)):
        f = S.state_fields(n, preset, seed=n + 11, extra_data=b"y" * (n % 33))
        f["_preset"] = preset
        t, v = _fork_state_value(fork, f, rnd)
        enc = t.serialize(v)
        rc, root, hashes = hs.state_root_fork(fork_id, enc, S.PRESETS[preset]["id"])
        assert rc == 0 and root == t.htr(v), (fork, preset, n)
    assert hs.state_root_fork(fork_id, enc[:200], S.PRESETS[preset]["id"])[0] == -3


def test_beacon_state_plan_rejects_malformed():
    from ethereum_consensus_amd import synthetic as S
    enc = bytearray(S.beacon_state_deneb(3, "minimal"))
    assert hs.state_root_deneb(bytes(enc[:100]), 1)[0] == -3
    assert hs.state_root_deneb(bytes(enc) + b"\0", 1)[0] == -3  # historical_summaries not a multiple of 64


@pytest.mark.parametrize("n0,limit", [(513, 1024), (1023, 1024), (1024, 1024), (1025, 2048), (2049, 1 << 40), (5000, 8192),
                                      (3 * 1024, 1 << 38), (600, 1 << 10)])
def test_tile_stage_schedule(n0, limit):
    """Trees wider than one finishing job go through the tile stage (merkle.h TileDesc: 1024 nodes per workgroup,
    virtual pairs taken from the zero ladder): ragged tails, exact powers of two, shallow and deep limits."""
    data = rnd(32 * n0 - 5, n0)
    depth = ssz._depth_for(limit)
    got, h = hs.merkleize_scheduled(0, data, n0, depth, True, n0, 3)
    assert got == ssz.mix_in_length(ssz.merkleize_bytes(data, limit), n0)
    assert h == ssz.hash64_count(n0, limit) + 1
    # record functors inside the tile stage: 48-byte keys and Eth1Data records
    keys = rnd(48 * n0, n0 + 1)
    got, _ = hs.merkleize_scheduled(3, keys, n0, depth, False, 0)
    assert got == ssz.merkleize_chunks([ssz.merkleize_bytes(keys[48 * i: 48 * i + 48], 2) for i in range(n0)], limit)


# ---- resident field trees: dirty-path re-hashing (csrc/state_tree.h) -----------------------------------
REC = {0: 32, 2: 121, 3: 48, 4: 64, 5: 72}


def _full_root(kind, data, n0, depth, mix, mix_len):
    L = hs.lib()
    L.hs_merkleize.restype = ctypes.c_uint64
    L.hs_merkleize.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                               ctypes.c_uint64, ctypes.c_void_p]
    out = ctypes.create_string_buffer(32)
    L.hs_merkleize(kind, data, len(data), n0, depth, 1 if mix else 0, mix_len, out)
    return out.raw


def _canon(kind, buf: bytearray) -> bytearray:
    """Validator records: the `slashed` member is an SSZ boolean, one byte that is 0 or 1 (any other value is not an encoding
    the reference's deserializer accepts), so that the oracle's typed hash_tree_root applies to the bytes"""
    if kind == 2:
        for i in range(88, len(buf), 121):
            buf[i] &= 1
    return buf


def _oracle_root(kind, data, n0, depth, mix, mix_len):
    """the same tree by oracle/ssz.py: typed hash_tree_root of every record, merkleize to 2^depth leaves, length mix-in -- nothing
    of the product's (passes, tiles, cached levels, climbs) is involved"""
    data = bytes(data)
    if kind == 0:
        root = ssz.merkleize_bytes(data, 1 << depth)
    else:
        rec = REC[kind]
        assert len(data) == rec * n0
        if kind == 2:
            offs, o = [], 0
            for name, ty in ssz.Validator.fields:
                offs.append((name, ty, o))
                o += ty.fixed_size

            def val(b):
                d = {}
                for name, ty, o in offs:
                    raw = b[o:o + ty.fixed_size]
                    d[name] = raw if isinstance(ty, ssz.ByteVector) else (bool(raw[0]) if isinstance(ty, ssz.Boolean) else int.from_bytes(raw, "little"))
                return d
            leaves = [ssz.Validator.htr(val(data[rec * i:rec * i + rec])) for i in range(n0)]
        elif kind == 3:
            leaves = [ssz.BlsPublicKey.htr(data[rec * i:rec * i + rec]) for i in range(n0)]
        elif kind == 4:
            leaves = [ssz.HistoricalSummary.htr({"block_summary_root": data[rec * i:rec * i + 32], "state_summary_root": data[rec * i + 32:rec * i + 64]})
                      for i in range(n0)]
        else:
            leaves = [ssz.Eth1Data.htr({"deposit_root": data[rec * i:rec * i + 32], "deposit_count": int.from_bytes(data[rec * i + 32:rec * i + 40], "little"),
                                        "block_hash": data[rec * i + 40:rec * i + 72]}) for i in range(n0)]
        root = ssz.merkleize_chunks(leaves, 1 << depth)
    return ssz.mix_in_length(root, mix_len) if mix else root


def _tree_update(kind, before, n0_before, after, n0_after, depth, mix, mix_len, marks, seed):
    L = hs.lib()
    L.hs_tree_update.restype = ctypes.c_uint64
    L.hs_tree_update.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64,
                                 ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32,
                                 ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    arr = (ctypes.c_uint64 * max(1, len(marks)))(*marks)
    out = ctypes.create_string_buffer(32)
    left = ctypes.c_uint32(99)
    rebuilt = ctypes.c_uint64(0)
    h = L.hs_tree_update(kind, before, len(before), n0_before, after, len(after), n0_after, depth, 1 if mix else 0, mix_len, arr, len(marks),
                         seed, out, ctypes.byref(left), ctypes.byref(rebuilt))
    assert h != 2 ** 64 - 1
    return out.raw, h, left.value, rebuilt.value


@pytest.mark.parametrize("kind,n", [(0, 2), (0, 3), (0, 5), (0, 64), (0, 1023), (0, 1024), (0, 1025), (0, 5000), (0, 70001),
                                    (2, 2), (2, 3), (2, 511), (2, 513), (2, 1500), (2, 9000), (5, 7), (5, 2048), (4, 700), (3, 512), (3, 1536)])
def test_resident_tree_rehashes_dirty_paths_only(kind, n):
    r = random.Random(1000 * kind + n)
    rec = REC[kind]
    # a packed field's last chunk may be partial
    nbytes = rec * n - (r.randrange(1, 31) if kind == 0 and n > 2 else 0)
    before = _canon(kind, bytearray(r.randbytes(nbytes)))
    depth = max(1, (n - 1).bit_length()) + r.randrange(0, 21)
    mix = kind != 3
    for trial, n_dirty in enumerate((0, 1, 2, min(n, 40), min(n, 700))):
        after = bytearray(before)
        touched = sorted(r.sample(range(n), n_dirty))
        for e in touched:
            lo, hi = rec * e, min(rec * e + rec, nbytes)
            pos = r.randrange(lo, hi)
            after[pos] ^= 1 + r.randrange(255)
            if kind == 2 and pos % 121 == 88:
                after[pos] = 1 - (before[pos] & 1)  # `slashed` flips between its two values
        marks = list(touched) + [r.choice(touched) for _ in range(len(touched) // 3)]  # duplicates are welcome
        r.shuffle(marks)
        root, hashes, left, rebuilt = _tree_update(kind, bytes(before), n, bytes(after), n, depth, mix, n, marks, seed=trial * 77 + n)
        assert left == 0  # every counter and flag back at zero
        assert root == _full_root(kind, bytes(after), n, depth, mix, n)
        if n <= 9000:  # ... and the ORACLE's root of the same records (the climb was only ever compared with the product's own from-scratch schedule)
            assert root == _oracle_root(kind, after, n, depth, mix, n), (kind, n, trial)
        # only dirty paths: at most (leaf work + T) hash64 per dirty entry, and no more than a rebuild
        H = max(1, (n - 1).bit_length())
        T = max(H - 9, 1 if kind == 0 else 0)
        per_leaf = {0: 0, 2: 8, 3: 1, 4: 1, 5: 3}[kind]
        assert hashes <= n_dirty * (per_leaf + T)
        assert hashes <= rebuilt
        if n_dirty == 0:
            assert hashes == 0
        before = after


@pytest.mark.parametrize("kind,n,grow", [(0, 1030, 7), (0, 1500, 548), (2, 600, 3), (2, 1025, 1023), (5, 100, 28), (4, 513, 1)])
def test_resident_tree_follows_an_append(kind, n, grow):
    """entries appended without changing the tree's height: the new entries (and a packed list's last partial chunk) are marked"""
    r = random.Random(5000 + n)
    rec = REC[kind]
    unit = 8 if kind == 0 else rec  # balances: 8-byte elements packed four to a chunk
    nb0 = unit * (n * (4 if kind == 0 else 1) - (1 if kind == 0 else 0))
    nb1 = nb0 + unit * grow * (4 if kind == 0 else 1)
    data = bytes(_canon(kind, bytearray(r.randbytes(nb1))))
    n0_0, n0_1 = (nb0 + rec - 1) // rec, (nb1 + rec - 1) // rec
    assert (n0_0 - 1).bit_length() == (n0_1 - 1).bit_length()
    depth = 40
    marks = list(range(nb0 // rec, (nb1 - 1) // rec + 1))
    root, hashes, left, _ = _tree_update(kind, data[:nb0], n0_0, data, n0_1, depth, True, nb1 // unit, marks, seed=n)
    assert left == 0
    assert root == _full_root(kind, data, n0_1, depth, True, nb1 // unit)
    assert root == _oracle_root(kind, data, n0_1, depth, True, nb1 // unit)
