"""Value-level model of a BeaconState under the mutations the reference's state transition makes, driving the field-addressed
entries of the C ABI (include/ecgpu.h ECGPU_BS_*, csrc/state_fields.h) -- TESTS ONLY.

Every step changes the ORACLE VALUE (a dict, as oracle/ssz.py containers take it) the way the reference changes its `BeaconState`
struct, and tells the driver the same thing in the reference's own coordinates: (field, index) + the element's serialization, made
by the oracle's serializer.  No byte offset into the state's encoding is computed anywhere on this side; the expected root is
`t.htr(value)` -- oracle/ssz.py, which shares nothing with the product's kernels, plans or layout tables.

Drivers: `HostsimDriver` (csrc/state_fields.h over a host byte array + the lane simulator's root: CPU suite) and the product's
`ethereum_consensus_amd.ssz.ResidentBeaconStateDeneb` (GPU suite) expose the same methods.
"""
import ctypes

from oracle import ssz as O

FIELD_POS = None


def positions():
    global FIELD_POS
    if FIELD_POS is None:
        from ethereum_consensus_amd.ssz import ResidentBeaconStateDeneb as R
        FIELD_POS = dict(R.FIELD_POSITIONS)
    return FIELD_POS


class HostsimDriver:
    """csrc/state_fields.h's queue over a host byte array (tests/hostsim/hostsim_fields.cpp); root = the lane simulator's
    state root of the resulting encoding (altair+) -- no GPU anywhere."""

    def __init__(self, fork: str, preset: int, enc: bytes):
        from tests import _hostsim
        self.L = _hostsim.lib()
        L = self.L
        L.hs_fs_create.restype = ctypes.c_void_p
        L.hs_fs_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64]
        L.hs_fs_destroy.argtypes = [ctypes.c_void_p]
        for name in ("hs_fs_patch_field", "hs_fs_patch_elements"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64]
        for name in ("hs_fs_push", "hs_fs_set_field"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64]
        L.hs_fs_truncate_field.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64]
        L.hs_fs_add_validator.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64]
        L.hs_fs_rotate_participation.argtypes = [ctypes.c_void_p]
        L.hs_fs_flush.argtypes = [ctypes.c_void_p]
        L.hs_fs_field_size.restype = ctypes.c_longlong
        L.hs_fs_field_size.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.hs_fs_encoding.restype = ctypes.c_uint64
        L.hs_fs_encoding.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        L.hs_fs_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.hs_fs_error.restype = ctypes.c_char_p
        L.hs_fs_error.argtypes = [ctypes.c_void_p]
        from ethereum_consensus_amd.ssz import FORKS
        self.fork, self.preset = FORKS[fork], preset
        self.h = L.hs_fs_create(self.fork, preset, enc, len(enc))

    def close(self):
        if self.h:
            self.L.hs_fs_destroy(self.h)
            self.h = None

    def _rc(self, rc):
        if rc:
            raise ValueError((self.L.hs_fs_error(self.h) or b"").decode() or f"rc {rc}")

    def _pos(self, f):
        return positions()[f] if isinstance(f, str) else int(f)

    def patch_field(self, f, off, data):
        self._rc(self.L.hs_fs_patch_field(self.h, self._pos(f), off, bytes(data), len(data)))

    def patch_elements(self, f, first, data):
        self._rc(self.L.hs_fs_patch_elements(self.h, self._pos(f), first, bytes(data), len(data)))

    def push(self, f, data):
        self._rc(self.L.hs_fs_push(self.h, self._pos(f), bytes(data), len(data)))

    def truncate_field(self, f, keep):
        self._rc(self.L.hs_fs_truncate_field(self.h, self._pos(f), keep))

    def set_field(self, f, data):
        self._rc(self.L.hs_fs_set_field(self.h, self._pos(f), bytes(data), len(data)))

    def add_validator(self, rec, balance):
        self._rc(self.L.hs_fs_add_validator(self.h, bytes(rec), balance))

    def rotate_participation(self):
        self._rc(self.L.hs_fs_rotate_participation(self.h))

    def flush(self):
        self._rc(self.L.hs_fs_flush(self.h))

    def field_size(self, f):
        return int(self.L.hs_fs_field_size(self.h, self._pos(f)))

    def encoding(self) -> bytes:
        self.flush()
        n = self.L.hs_fs_encoding(self.h, None, 0)
        buf = ctypes.create_string_buffer(n)
        self.L.hs_fs_encoding(self.h, buf, n)
        return buf.raw

    def counters(self):
        c = (ctypes.c_uint32 * 3)()
        self.L.hs_fs_counters(self.h, c)
        return tuple(c)

    def __len__(self):
        return len(self.encoding())

    def hash_tree_root(self) -> bytes:
        from tests import _hostsim
        rc, root, _ = _hostsim.state_root_fork(self.fork, self.encoding(), self.preset)
        assert rc == 0, rc
        return root


def _elem(t, name):
    """the element type of list / vector field `name` of container t"""
    return dict(t.fields)[name].elem


def _ftype(t, name):
    return dict(t.fields)[name]


def random_validator(r):
    far = (1 << 64) - 1
    return {"public_key": r.randbytes(48), "withdrawal_credentials": bytes([r.choice([0, 1])]) + bytes(11) + r.randbytes(20),
            "effective_balance": r.choice([32, 31, 16, 0]) * 10**9, "slashed": r.random() < 0.1,
            "activation_eligibility_epoch": r.randrange(1 << 20), "activation_epoch": r.choice([far, r.randrange(1 << 20)]),
            "exit_epoch": r.choice([far, far, r.randrange(1 << 20)]), "withdrawable_epoch": r.choice([far, far, r.randrange(1 << 20)])}


# byte offsets of the fields inside a 121-byte Validator record (phase0/validator.rs:10-26), from the ORACLE's field sizes
def _validator_field_offsets():
    off, out = 0, {}
    for name, ty in O.Validator.fields:
        out[name] = (off, ty)
        off += ty.fixed_size
    assert off == 121
    return out


OPS = ["balance", "balance", "balance", "flags", "flags", "score", "validator_field", "validator_record", "deposit", "deposit_then_balance",
       "vote", "votes_reset", "slot", "mix", "slashing", "summary", "checkpoints", "rotate", "epoch_balances", "eth1", "header",
       "withdrawal_indices", "twice", "electra_lists", "attestations", "epoch_boundary", "nothing"]


def random_step(r, drv, t, v, fork: str, preset_name: str):
    """one mutation of the kind the reference's state transition makes; returns the operation's name (or None if not applicable)"""
    op = r.choice(OPS)
    n = len(v["validators"])
    altair = fork != "phase0"
    u64 = lambda x: int(x).to_bytes(8, "little")
    if op == "balance":  # increase_balance / decrease_balance (phase0/helpers.rs:979-1030): a block's rewards and penalties
        for _ in range(r.choice([1, 4, 60, 700])):
            if not n:
                break
            i = r.randrange(n)
            v["balances"][i] = r.randrange(1 << 40)
            drv.patch_elements("balances", i, u64(v["balances"][i]))
    elif op == "flags" and altair:  # process_attestation (altair/block_processing.rs:98-170)
        name = r.choice(["current_epoch_participation", "previous_epoch_participation"])
        for _ in range(r.choice([1, 30, 500])):
            if not n:
                break
            i = r.randrange(n)
            v[name][i] = r.randrange(8)
            drv.patch_elements(name, i, bytes([v[name][i]]))
    elif op == "score" and altair and n:  # process_inactivity_updates
        i = r.randrange(n)
        v["inactivity_scores"][i] = r.randrange(1 << 20)
        drv.patch_elements("inactivity_scores", i, u64(v["inactivity_scores"][i]))
    elif op == "validator_field" and n:  # slash_validator / initiate_validator_exit / effective-balance updates: ONE member of a record
        i = r.randrange(n)
        offs = _validator_field_offsets()
        name = r.choice(["slashed", "exit_epoch", "withdrawable_epoch", "effective_balance", "activation_epoch", "withdrawal_credentials"])
        new = random_validator(r)[name]
        v["validators"][i] = dict(v["validators"][i], **{name: new})
        off, ty = offs[name]
        drv.patch_field("validators", 121 * i + off, ty.serialize(new))
    elif op == "validator_record" and n:  # the whole record
        i = r.randrange(n)
        v["validators"][i] = random_validator(r)
        drv.patch_elements("validators", i, O.Validator.serialize(v["validators"][i]))
    elif op in ("deposit", "deposit_then_balance"):  # add_validator_to_registry (phase0/block_processing.rs:317-349)
        for _ in range(r.choice([1, 1, 2, 16])):
            rec, bal = random_validator(r), r.randrange(1 << 36)
            v["validators"] = v["validators"] + [rec]
            v["balances"] = v["balances"] + [bal]
            if altair:
                for name in ("previous_epoch_participation", "current_epoch_participation", "inactivity_scores"):
                    v[name] = v[name] + [0]
            drv.add_validator(O.Validator.serialize(rec), bal)
            if op == "deposit_then_balance":
                # the advisor's first scenario: a top-up of the validator just added AND of an old one, in the same slot
                m = len(v["validators"])
                for i in (m - 1, r.randrange(m)):
                    v["balances"][i] += 10**9
                    drv.patch_elements("balances", i, u64(v["balances"][i]))
                if altair and r.random() < 0.5:
                    v["current_epoch_participation"][m - 1] = 7
                    drv.patch_elements("current_epoch_participation", m - 1, b"\x07")
    elif op == "vote":  # process_eth1_data (phase0/block_processing.rs:689-700): one vote per block
        lim = _ftype(t, "eth1_data_votes").limit
        if len(v["eth1_data_votes"]) < lim:
            e = {"deposit_root": r.randbytes(32), "deposit_count": r.randrange(1 << 32), "block_hash": r.randbytes(32)}
            v["eth1_data_votes"] = v["eth1_data_votes"] + [e]
            drv.push("eth1_data_votes", O.Eth1Data.serialize(e))
    elif op == "votes_reset":  # process_eth1_data_reset
        v["eth1_data_votes"] = []
        drv.truncate_field("eth1_data_votes", 0)
    elif op == "slot":  # process_slot (phase0/slot_processing.rs:58-86): state root cached, header filled, block root cached, slot += 1
        N = len(v["state_roots"])
        s = v["slot"]
        sr, br = r.randbytes(32), r.randbytes(32)
        v["state_roots"] = list(v["state_roots"])
        v["block_roots"] = list(v["block_roots"])
        v["state_roots"][s % N] = sr
        v["block_roots"][s % N] = br
        v["latest_block_header"] = dict(v["latest_block_header"], state_root=sr)
        v["slot"] = s + 1
        drv.patch_elements("state_roots", s % N, sr)
        drv.patch_field("latest_block_header", 8 + 8 + 32, sr)  # slot, proposer_index, parent_root | state_root
        drv.patch_elements("block_roots", s % N, br)
        drv.patch_elements("slot", 0, u64(s + 1))
    elif op == "mix":  # process_randao
        N = len(v["randao_mixes"])
        i = r.randrange(N)
        v["randao_mixes"] = list(v["randao_mixes"])
        v["randao_mixes"][i] = r.randbytes(32)
        drv.patch_elements("randao_mixes", i, v["randao_mixes"][i])
    elif op == "slashing":
        N = len(v["slashings"])
        i = r.randrange(N)
        v["slashings"] = list(v["slashings"])
        v["slashings"][i] = r.randrange(1 << 50)
        drv.patch_elements("slashings", i, u64(v["slashings"][i]))
    elif op == "summary":  # process_historical_summaries_update / process_historical_roots_update
        if "historical_summaries" in v and r.random() < 0.7:
            e = {"block_summary_root": r.randbytes(32), "state_summary_root": r.randbytes(32)}
            v["historical_summaries"] = v["historical_summaries"] + [e]
            drv.push("historical_summaries", e["block_summary_root"] + e["state_summary_root"])
        else:
            e = r.randbytes(32)
            v["historical_roots"] = v["historical_roots"] + [e]
            drv.push("historical_roots", e)
    elif op == "checkpoints":  # process_justification_and_finalization
        bits = [r.random() < 0.5 for _ in range(4)]
        v["justification_bits"] = bits
        drv.set_field("justification_bits", _ftype(t, "justification_bits").serialize(bits))
        for name in ("previous_justified_checkpoint", "current_justified_checkpoint", "finalized_checkpoint"):
            v[name] = {"epoch": r.randrange(1 << 30), "root": r.randbytes(32)}
            drv.set_field(name, O.Checkpoint.serialize(v[name]))
    elif op == "rotate" and altair:  # process_participation_flag_updates
        v["previous_epoch_participation"] = list(v["current_epoch_participation"])
        v["current_epoch_participation"] = [0] * n
        drv.rotate_participation()
    elif op == "epoch_balances" and n:  # process_rewards_and_penalties: every balance, in one write
        v["balances"] = [r.randrange(1 << 40) for _ in range(n)]
        drv.patch_elements("balances", 0, b"".join(u64(b) for b in v["balances"]))
    elif op == "eth1":
        v["eth1_data"] = {"deposit_root": r.randbytes(32), "deposit_count": r.randrange(1 << 32), "block_hash": r.randbytes(32)}
        v["eth1_deposit_index"] = r.randrange(1 << 32)
        drv.set_field("eth1_data", O.Eth1Data.serialize(v["eth1_data"]))
        drv.set_field("eth1_deposit_index", u64(v["eth1_deposit_index"]))
    elif op == "header" and "latest_execution_payload_header" in v:  # process_execution_payload: a new header, extra_data of any length
        ty = _ftype(t, "latest_execution_payload_header")
        hdr = dict(v["latest_execution_payload_header"])
        hdr["block_hash"], hdr["extra_data"] = r.randbytes(32), r.randbytes(r.choice([0, 1, 17, 32]))
        hdr["block_number"] = r.randrange(1 << 40)
        v["latest_execution_payload_header"] = hdr
        enc = ty.serialize(hdr)
        if r.random() < 0.5 or len(enc) != drv.field_size("latest_execution_payload_header"):
            drv.set_field("latest_execution_payload_header", enc)
        else:  # same length: the members patched where they lie inside the header (offsets from the ORACLE's field sizes)
            off = 0
            for fn, fty in ty.fields:
                size = fty.fixed_size if fty.fixed_size is not None else 4
                if fn in ("block_hash", "block_number"):
                    drv.patch_field("latest_execution_payload_header", off, fty.serialize(hdr[fn]))
                off += size
            drv.patch_field("latest_execution_payload_header", off, hdr["extra_data"])  # extra_data: the header's only variable part
    elif op == "withdrawal_indices" and "next_withdrawal_index" in v:
        v["next_withdrawal_index"] = r.randrange(1 << 40)
        v["next_withdrawal_validator_index"] = r.randrange(1 << 20)
        drv.patch_elements("next_withdrawal_index", 0, u64(v["next_withdrawal_index"]))
        drv.set_field("next_withdrawal_validator_index", u64(v["next_withdrawal_validator_index"]))
    elif op == "twice" and n:  # the same element written several times between two roots: the last value stands; overlapping ranges
        i = r.randrange(n)
        for _ in range(3):
            v["balances"][i] = r.randrange(1 << 40)
            drv.patch_elements("balances", i, u64(v["balances"][i]))
        if n >= 4:
            j = r.randrange(n - 3)
            a = [r.randrange(1 << 40) for _ in range(4)]
            drv.patch_elements("balances", j, b"".join(u64(x) for x in a))      # [j, j + 4)
            b = [r.randrange(1 << 40) for _ in range(2)]
            drv.patch_elements("balances", j + 1, b"".join(u64(x) for x in b))  # [j + 1, j + 3) on top
            v["balances"][j:j + 4] = [a[0], b[0], b[1], a[3]]
            drv.patch_field("balances", 8 * j + 3, b"\x99\x98")                  # two bytes inside element j, on top again
            x = bytearray(u64(a[0]))
            x[3:5] = b"\x99\x98"
            v["balances"][j] = int.from_bytes(x, "little")
    elif op == "electra_lists" and fork == "electra":
        name = r.choice(["pending_balance_deposits", "pending_partial_withdrawals", "pending_consolidations"])
        ety = _elem(t, name)
        lim = _ftype(t, name).limit
        if r.random() < 0.3 and v[name]:
            keep = r.randrange(len(v[name]) + 1)
            v[name] = v[name][:keep]
            drv.truncate_field(name, keep * ety.fixed_size)
        elif len(v[name]) < lim:
            e = {fn: r.randrange(1 << 40) for fn, _ in ety.fields}
            v[name] = v[name] + [e]
            drv.push(name, ety.serialize(e))
    elif op == "attestations" and fork == "phase0":  # process_attestation pushes; the epoch boundary rotates (phase0/epoch_processing.rs)
        ty = _ftype(t, "current_epoch_attestations")
        if r.random() < 0.3:
            v["previous_epoch_attestations"], v["current_epoch_attestations"] = v["current_epoch_attestations"], []
            drv.set_field("previous_epoch_attestations", ty.serialize(v["previous_epoch_attestations"]))
            drv.set_field("current_epoch_attestations", b"")
        elif len(v["current_epoch_attestations"]) < 64:
            a = {"aggregation_bits": [r.random() < 0.5 for _ in range(r.choice([0, 1, 8, 9, 64, 333]))],
                 "data": {"slot": r.randrange(1 << 40), "index": r.randrange(64), "beacon_block_root": r.randbytes(32),
                          "source": {"epoch": r.randrange(1 << 30), "root": r.randbytes(32)},
                          "target": {"epoch": r.randrange(1 << 30), "root": r.randbytes(32)}},
                 "inclusion_delay": r.randrange(1, 33), "proposer_index": r.randrange(1 << 20)}
            v["current_epoch_attestations"] = v["current_epoch_attestations"] + [a]
            drv.set_field("current_epoch_attestations", ty.serialize(v["current_epoch_attestations"]))
    elif op == "epoch_boundary" and altair:
        # what rust/patches/ethereum-consensus-gpu-feature.patch hands over after process_epoch: the participation lists rotate on the
        # device, the registry-sized lists and the vectors process_epoch rewrites travel whole (set_field; same length = a write,
        # large ones are applied from where they lie)
        v["previous_epoch_participation"] = list(v["current_epoch_participation"])
        v["current_epoch_participation"] = [0] * n
        drv.rotate_participation()
        v["balances"] = [r.randrange(1 << 40) for _ in range(n)]
        v["inactivity_scores"] = [r.randrange(1 << 16) for _ in range(n)]
        v["validators"] = [dict(x, effective_balance=r.choice([32, 31, 30]) * 10**9) for x in v["validators"]]
        v["slashings"] = list(v["slashings"])
        v["slashings"][r.randrange(len(v["slashings"]))] = 0
        v["randao_mixes"] = list(v["randao_mixes"])
        v["randao_mixes"][r.randrange(len(v["randao_mixes"]))] = r.randbytes(32)
        for name in ("balances", "inactivity_scores", "validators", "slashings", "randao_mixes"):
            drv.set_field(name, _ftype(t, name).serialize(v[name]))
        if n:  # ... and the first block of the new epoch writes on top of it before the next root
            i = r.randrange(n)
            v["balances"][i] += 1
            drv.patch_elements("balances", i, u64(v["balances"][i]))
    elif op == "nothing":
        pass
    else:
        return None
    return op
