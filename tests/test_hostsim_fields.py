"""csrc/state_fields.h -- the field-addressed write queue behind ecgpu_resident_state_patch_field / _patch_elements / _push /
_truncate_field / _set_field / _add_validator / _rotate_participation -- executed on the CPU (tests/hostsim/hostsim_fields.cpp:
the same FieldWriter template the product instantiates over the device-resident encoding, here over a host byte array) and
compared with oracle/ssz.py at the VALUE level: every mutation is made on the oracle value the way the reference's state
transition makes it on its struct (phase0/helpers.rs:979-1030, phase0/block_processing.rs:317-349, 689-700,
altair/block_processing.rs:98-213, phase0/slot_processing.rs:58-86) and told to the queue in (field, index) coordinates; the
resulting ENCODING must be the oracle's serialization of the value, byte for byte, and (altair+) the lane simulator's root of it
the oracle's hash_tree_root.

Up to round 5 this arithmetic lived in the Rust shim's StateMirror, which nothing in this image can compile; the advisor found
two wrong-root bugs in it by reading (a balance written after a deposit of the same slot; eth1 votes across a reset).  Both are
scenarios below."""
import random
import zlib

import pytest

from oracle import ssz as O
from tests import _statefields as SF
from tests._statevalue import fork_state_value


def _state(fork, preset_name, n, seed):
    from ethereum_consensus_amd import synthetic
    r = random.Random(seed)
    f = synthetic.state_fields(n, preset_name, seed=seed, extra_data=r.randbytes(r.choice([0, 5, 32])))
    f["_preset"] = preset_name
    t, v = fork_state_value(fork, f, r)
    v = {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in v.items()}
    return t, v


def _check(drv, t, v, fork, why):
    enc = drv.encoding()
    assert enc == t.serialize(v), why
    if fork != "phase0":  # (the lane simulator's state root starts at altair; phase0 is covered byte for byte above)
        assert drv.hash_tree_root() == t.htr(v), why


@pytest.mark.parametrize("fork,preset_name,n,steps", [("phase0", "minimal", 90, 160), ("altair", "minimal", 130, 200), ("bellatrix", "mainnet", 70, 120),
                                                       ("capella", "minimal", 257, 200), ("deneb", "minimal", 64, 200), ("deneb", "mainnet", 300, 120),
                                                       ("electra", "minimal", 100, 200)])
def test_randomised_state_transition_mutations_in_field_coordinates(fork, preset_name, n, steps):
    r = random.Random(zlib.crc32(f"{fork}/{preset_name}/{n}".encode()))
    t, v = _state(fork, preset_name, n, seed=n)
    drv = SF.HostsimDriver(fork, 1 if preset_name == "minimal" else 0, t.serialize(v))
    _check(drv, t, v, fork, "start")
    seen = set()
    for k in range(steps):
        op = SF.random_step(r, drv, t, v, fork, preset_name)
        seen.add(op)
        if r.random() < 0.4:
            continue  # several mutations between two roots: they travel in one block
        assert drv.field_size("validators") == 121 * len(v["validators"]) and drv.field_size("balances") == 8 * len(v["balances"])
        _check(drv, t, v, fork, (k, op))
    _check(drv, t, v, fork, "end")
    # (the seeded sequence must have exercised the queue's main cases: element writes, length changes of both kinds, writes on top of
    # each other, a whole-list rewrite)
    assert {"balance", "vote", "slot", "validator_field"} <= seen and seen & {"deposit", "deposit_then_balance"} and "twice" in seen
    assert seen & {"epoch_balances", "epoch_boundary"} and len(seen - {None}) >= 14, seen
    drv.close()


def test_a_deposit_then_a_balance_write_in_one_slot():
    """advisor, round 4 (rust/ecgpu-shim StateMirror): absolute offsets taken before an append land 121 n bytes early once a
    deposit has grown the registry.  Here: 3 deposits, then top-ups of an OLD validator and of the NEWEST one, then flags of the
    newest -- all before one root."""
    t, v = _state("deneb", "minimal", 50, seed=3)
    drv = SF.HostsimDriver("deneb", 1, t.serialize(v))
    r = random.Random(1)
    for _ in range(3):
        rec = SF.random_validator(r)
        v["validators"].append(rec)
        v["balances"].append(32 * 10**9)
        for name in ("previous_epoch_participation", "current_epoch_participation", "inactivity_scores"):
            v[name].append(0)
        drv.add_validator(O.Validator.serialize(rec), 32 * 10**9)
    v["balances"][7] += 5
    drv.patch_elements("balances", 7, v["balances"][7].to_bytes(8, "little"))
    v["balances"][52] = 31 * 10**9
    drv.patch_elements("balances", 52, v["balances"][52].to_bytes(8, "little"))
    v["current_epoch_participation"][52] = 5
    drv.patch_elements("current_epoch_participation", 52, b"\x05")
    v["validators"][51] = dict(v["validators"][51], slashed=True)
    drv.patch_field("validators", 121 * 51 + 88, b"\x01")
    assert drv.counters() == (0, 0, 0)  # nothing has reached the encoding yet
    _check(drv, t, v, "deneb", "deposits then writes")
    calls = drv.counters()
    assert calls[1] == 5 and calls[0] == 1  # five lists grew once each (three deposits merged); ONE patch block (the writes into queued elements rode in the pushes)
    drv.close()


def test_an_eth1_vote_per_block_across_a_voting_period_reset():
    """advisor, round 5: `eth1_data_votes` grows by 72 bytes per block and is emptied at the period boundary -- everything behind
    it (the registry, balances ...) moves each time.  32 blocks of a minimal-preset period, a balance write per block, a root per
    block; then the reset, then votes again."""
    t, v = _state("capella", "minimal", 40, seed=9)
    v["eth1_data_votes"] = []
    drv = SF.HostsimDriver("capella", 1, t.serialize(v))
    r = random.Random(2)
    for period in range(2):
        for blk in range(32):
            e = {"deposit_root": r.randbytes(32), "deposit_count": blk, "block_hash": r.randbytes(32)}
            v["eth1_data_votes"].append(e)
            drv.push("eth1_data_votes", O.Eth1Data.serialize(e))
            i = r.randrange(40)
            v["balances"][i] = r.randrange(1 << 40)
            drv.patch_elements("balances", i, v["balances"][i].to_bytes(8, "little"))
            if blk % 5 == 0:
                _check(drv, t, v, "capella", (period, blk))
        with pytest.raises(ValueError):
            drv.push("eth1_data_votes", bytes(72))  # EPOCHS_PER_ETH1_VOTING_PERIOD * SLOTS_PER_EPOCH = 32 on minimal: the 33rd is refused
        _check(drv, t, v, "capella", "full period")
        v["eth1_data_votes"] = []
        drv.truncate_field("eth1_data_votes", 0)
        v["balances"][0] = period
        drv.patch_elements("balances", 0, period.to_bytes(8, "little"))
        _check(drv, t, v, "capella", "after the reset")
    drv.close()


def test_queue_semantics_are_program_order():
    t, v = _state("altair", "minimal", 20, seed=5)
    drv = SF.HostsimDriver("altair", 1, t.serialize(v))
    u64 = lambda x: int(x).to_bytes(8, "little")
    drv.truncate_field("historical_roots", 0)
    # a write into an element that is still queued, then a truncate that drops it again, then a push in its place
    drv.push("historical_roots", bytes([1]) * 32 + bytes([2]) * 32)
    drv.patch_elements("historical_roots", 1, bytes([3]) * 32)
    drv.patch_elements("historical_roots", 0, bytes([1]) * 16 + bytes([11]) * 16)
    drv.truncate_field("historical_roots", 32)
    drv.push("historical_roots", bytes([4]) * 32)
    v["historical_roots"] = [bytes([1]) * 16 + bytes([11]) * 16, bytes([4]) * 32]
    _check(drv, t, v, "altair", "queued elements edited and dropped")
    # a write, then a truncate below it (the write dies with its bytes), then the list regrows
    drv.patch_elements("historical_roots", 1, bytes([9]) * 32)
    drv.truncate_field("historical_roots", 32)
    drv.push("historical_roots", bytes([5]) * 32)
    v["historical_roots"] = [bytes([1]) * 16 + bytes([11]) * 16, bytes([5]) * 32]
    _check(drv, t, v, "altair", "truncate below a queued write")
    # a write that straddles applied and queued elements
    drv.push("historical_roots", bytes([6]) * 32)
    drv.patch_field("historical_roots", 48, bytes([12]) * 32)
    v["historical_roots"] = [v["historical_roots"][0], bytes([5]) * 16 + bytes([12]) * 16, bytes([12]) * 16 + bytes([6]) * 16]
    _check(drv, t, v, "altair", "a write across the applied / queued boundary")
    drv.truncate_field("historical_roots", 64)
    v["historical_roots"] = v["historical_roots"][:2]
    # set_field over a list with queued writes and pushes: what was queued for it is superseded; other fields keep theirs
    drv.patch_elements("balances", 3, u64(77))
    drv.push("historical_roots", bytes([6]) * 32)
    drv.patch_elements("historical_roots", 0, bytes([7]) * 32)
    drv.set_field("historical_roots", bytes([8]) * 96)
    v["balances"][3] = 77
    v["historical_roots"] = [bytes([8]) * 32] * 3
    assert drv.field_size("historical_roots") == 96
    _check(drv, t, v, "altair", "set_field supersedes")
    # same-length set_field = a write; later writes land on top
    drv.set_field("balances", b"".join(u64(1000 + i) for i in range(20)))
    drv.patch_elements("balances", 19, u64(5))
    v["balances"] = [1000 + i for i in range(19)] + [5]
    _check(drv, t, v, "altair", "whole-list write then an element")
    # rotation sees the flags written before it and not those after
    drv.patch_elements("current_epoch_participation", 2, b"\x07")
    drv.rotate_participation()
    drv.patch_elements("current_epoch_participation", 4, b"\x01")
    prev = list(v["current_epoch_participation"])
    prev[2] = 7
    v["previous_epoch_participation"], v["current_epoch_participation"] = prev, [0] * 4 + [1] + [0] * 15
    _check(drv, t, v, "altair", "rotation in program order")
    drv.close()


def test_refused_calls_change_nothing():
    t, v = _state("deneb", "minimal", 10, seed=6)
    drv = SF.HostsimDriver("deneb", 1, t.serialize(v))
    bad = [lambda: drv.patch_elements("balances", 10, bytes(8)),            # one past the end
           lambda: drv.patch_field("balances", 79, bytes(2)),               # straddles the end
           lambda: drv.patch_elements("balances", 0, bytes(7)),             # not whole elements
           lambda: drv.push("validators", bytes(120)),
           lambda: drv.push("slot", bytes(8)),                              # not a list
           lambda: drv.push("latest_execution_payload_header", bytes(8)),
           lambda: drv.truncate_field("balances", 88),                      # longer than the list
           lambda: drv.truncate_field("validators", 60),
           lambda: drv.set_field("slot", bytes(4)),                         # a fixed-size field keeps its size
           lambda: drv.set_field("eth1_data_votes", bytes(72 * 33)),        # past the limit (32 on minimal)
           lambda: drv.patch_elements(34, 0, bytes(16)),                    # electra's field in a deneb state
           lambda: drv.patch_elements(99, 0, bytes(8)),
           lambda: drv.set_field("latest_execution_payload_header", bytes(584 + 33)),  # extra_data longer than 32 bytes
           lambda: drv.set_field("latest_execution_payload_header", bytes(100))]
    for k, call in enumerate(bad):
        with pytest.raises(ValueError):
            call()
        assert drv.counters()[:2] == (0, 0), k
    _check(drv, t, v, "deneb", "after refused calls")
    p0t, p0v = _state("phase0", "minimal", 10, seed=6)
    p0 = SF.HostsimDriver("phase0", 1, p0t.serialize(p0v))
    for call in (lambda: p0.rotate_participation(), lambda: p0.patch_field("current_epoch_attestations", 0, b"\x00"),
                 lambda: p0.push("current_epoch_attestations", bytes(8)), lambda: p0.patch_elements("inactivity_scores", 0, bytes(8))):
        with pytest.raises(ValueError):
            call()
    assert p0.encoding() == p0t.serialize(p0v)
    p0.close()
    drv.close()


def test_a_slots_writes_travel_as_one_block():
    """BASELINE configs[4]: 4 096 balances + 4 096 participation flags per slot -> one patch call, one patch per write"""
    t, v = _state("deneb", "mainnet", 5000, seed=8)
    drv = SF.HostsimDriver("deneb", 0, t.serialize(v))
    r = random.Random(4)
    idx = r.sample(range(5000), 4096)
    for i in idx:
        v["balances"][i] = r.randrange(1 << 40)
        drv.patch_elements("balances", i, v["balances"][i].to_bytes(8, "little"))
    for i in r.sample(range(5000), 4096):
        v["current_epoch_participation"][i] = r.randrange(1, 8)
        drv.patch_elements("current_epoch_participation", i, bytes([v["current_epoch_participation"][i]]))
    assert drv.encoding() == t.serialize(v)
    assert drv.counters() == (1, 0, 8192)
    drv.close()
