"""CPU oracle for SSZ Merkleization -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference's `hash_tree_root` lives in the un-vendored git dependency `ssz_rs` @
84ef2b71aa004f6767420badb42c902ad56b8b72 (/root/reference/Cargo.toml:20) on top of `sha2`
0.10.8 (Cargo.toml:23).  Neither is on disk, so this restates the published algorithm they
implement -- consensus-specs `ssz/simple-serialize.md` -- over
`hashlib.sha256`, and takes the *type trees* from the reference:

  BeaconBlockHeader  phase0/beacon_block.rs:83-91      Validator   phase0/validator.rs:10-26
  Fork/ForkData      phase0/beacon_state.rs:15-29      Checkpoint  phase0/operations.rs:13-17
  Eth1Data           phase0/operations.rs:66-71        SigningData signing.rs:8-12
  SyncCommittee      altair/sync.rs:17-22              HistoricalSummary capella (phase0/beacon_state.rs:42-45)
  ExecutionPayloadHeader (deneb)  deneb/execution_payload.rs:48-76
  BeaconState (deneb)             deneb/beacon_state.rs:13-64
  presets            phase0/presets/{mainnet,minimal}.rs, altair/presets/mainnet.rs:19,
                     bellatrix/presets/mainnet.rs:23-24

Pinned (tests/test_oracle_ssz.py) by the reference's offline fixtures: the sepolia BlobSidecar
inclusion proof (deneb/blob_sidecar.rs:70-132: htr(ByteVector<48>) + 17-deep branch, which
contains Z0..Z3 and a length mix-in chunk) and the generalized indices of
deneb/beacon_block.rs:139-154.
"""
from __future__ import annotations

import hashlib
from typing import List, Sequence

BYTES_PER_CHUNK = 32


def hash64(a: bytes, b: bytes) -> bytes:
    return hashlib.sha256(a + b).digest()


ZERO_HASHES: List[bytes] = [bytes(32)]
for _ in range(64):
    ZERO_HASHES.append(hash64(ZERO_HASHES[-1], ZERO_HASHES[-1]))


def _depth_for(limit: int) -> int:
    return 0 if limit <= 1 else (limit - 1).bit_length()


def merkleize_chunks(chunks: Sequence[bytes], limit: int | None = None) -> bytes:
    """Appendix A `merkleize`: virtual zero padding to next_pow2(limit); odd tails pair with Z_d."""
    n = len(chunks)
    if limit is None:
        limit = n
    assert n <= max(limit, 0) or (limit == 0 and n == 0), (n, limit)
    depth = _depth_for(limit)
    if n == 0:
        return ZERO_HASHES[depth]
    layer = list(chunks)
    for d in range(depth):
        if len(layer) & 1:
            layer.append(ZERO_HASHES[d])
        layer = [hash64(layer[i], layer[i + 1]) for i in range(0, len(layer), 2)]
    assert len(layer) == 1
    return layer[0]


def merkleize_bytes(data: bytes, limit_chunks: int | None = None) -> bytes:
    """`pack` (right-pad to 32) then merkleize."""
    if len(data) % 32:
        data = data + bytes(32 - len(data) % 32)
    chunks = [data[i : i + 32] for i in range(0, len(data), 32)]
    return merkleize_chunks(chunks, limit_chunks)


def merkleize_subtree_roots(sub_roots: Sequence[bytes], width: int, limit: int) -> bytes:
    """Top of a tree whose first levels were reduced elsewhere (sharded lists): `sub_roots` are the roots
    of consecutive aligned `width`-leaf subtrees; climb from level log2(width) to the `limit`-leaf root, odd tails
    pairing with the zero hash of their own level."""
    level, depth = _depth_for(width), _depth_for(limit)
    assert width == 1 << level and len(sub_roots) * width <= max(limit, 1)
    if not sub_roots:
        return ZERO_HASHES[depth]
    layer = list(sub_roots)
    for d in range(level, depth):
        if len(layer) & 1:
            layer.append(ZERO_HASHES[d])
        layer = [hash64(layer[i], layer[i + 1]) for i in range(0, len(layer), 2)]
    assert len(layer) == 1
    return layer[0]


def mix_in_length(root: bytes, length: int) -> bytes:
    return hash64(root, length.to_bytes(32, "little"))


def is_valid_merkle_branch(leaf: bytes, branch: Sequence[bytes], depth: int, index: int, root: bytes) -> bool:
    """Appendix A; argument order as at phase0/block_processing.rs:433."""
    v = leaf
    for i in range(depth):
        if (index >> i) & 1:
            v = hash64(branch[i], v)
        else:
            v = hash64(v, branch[i])
    return v == root


def hash64_count(n_chunks: int, limit: int) -> int:
    """Number of hash64 the merkleize above performs (for work accounting in bench.py)."""
    depth = _depth_for(limit)
    if n_chunks == 0:
        return 0
    c, total = n_chunks, 0
    for _ in range(depth):
        c = (c + 1) // 2
        total += c
    return total


# --------------------------------------------------------------------------------------
# a tiny SSZ type system: every type has serialize(v) -> bytes and htr(v) -> 32 bytes
# --------------------------------------------------------------------------------------
class SSZType:
    fixed_size: int | None = None

    def serialize(self, v) -> bytes:  # pragma: no cover
        raise NotImplementedError

    def htr(self, v) -> bytes:  # pragma: no cover
        raise NotImplementedError

    def default(self):  # pragma: no cover
        raise NotImplementedError


class UInt(SSZType):
    def __init__(self, bits: int):
        self.bits = bits
        self.fixed_size = bits // 8

    def serialize(self, v):
        return int(v).to_bytes(self.fixed_size, "little")

    def htr(self, v):
        return self.serialize(v).ljust(32, b"\0")

    def default(self):
        return 0


class Boolean(SSZType):
    fixed_size = 1

    def serialize(self, v):
        return b"\x01" if v else b"\x00"

    def htr(self, v):
        return self.serialize(v).ljust(32, b"\0")

    def default(self):
        return False


uint8, uint64, uint256, boolean = UInt(8), UInt(64), UInt(256), Boolean()


class ByteVector(SSZType):
    def __init__(self, n: int):
        self.n = n
        self.fixed_size = n

    def serialize(self, v):
        assert len(v) == self.n
        return bytes(v)

    def htr(self, v):
        return merkleize_bytes(self.serialize(v), (self.n + 31) // 32)

    def default(self):
        return bytes(self.n)


class ByteList(SSZType):
    def __init__(self, limit: int):
        self.limit = limit

    def serialize(self, v):
        assert len(v) <= self.limit
        return bytes(v)

    def htr(self, v):
        return mix_in_length(merkleize_bytes(bytes(v), (self.limit + 31) // 32), len(v))

    def default(self):
        return b""


Bytes32 = ByteVector(32)
Root = Bytes32


def _serialize_sequence(elem: SSZType, v) -> bytes:
    """SSZ sequence encoding: fixed-size elements back to back; variable-size ones behind a table of 4-byte offsets."""
    parts = [elem.serialize(x) for x in v]
    if elem.fixed_size is not None:
        return b"".join(parts)
    off, table = 4 * len(parts), []
    for p in parts:
        table.append(off.to_bytes(4, "little"))
        off += len(p)
    return b"".join(table) + b"".join(parts)


class Vector(SSZType):
    def __init__(self, elem: SSZType, n: int):
        self.elem, self.n = elem, n
        if elem.fixed_size is not None:
            self.fixed_size = elem.fixed_size * n

    def serialize(self, v):
        assert len(v) == self.n
        return _serialize_sequence(self.elem, v)

    def htr(self, v):
        assert len(v) == self.n
        if isinstance(self.elem, (UInt, Boolean)):
            return merkleize_bytes(self.serialize(v), (self.n * self.elem.fixed_size + 31) // 32)
        return merkleize_chunks([self.elem.htr(x) for x in v], self.n)

    def default(self):
        return [self.elem.default() for _ in range(self.n)]


class SSZList(SSZType):
    def __init__(self, elem: SSZType, limit: int):
        self.elem, self.limit = elem, limit

    def serialize(self, v):
        return _serialize_sequence(self.elem, v)

    def htr(self, v):
        assert len(v) <= self.limit
        if isinstance(self.elem, (UInt, Boolean)):
            lim = (self.limit * self.elem.fixed_size + 31) // 32
            return mix_in_length(merkleize_bytes(self.serialize(v), lim), len(v))
        return mix_in_length(merkleize_chunks([self.elem.htr(x) for x in v], self.limit), len(v))

    def default(self):
        return []


class Bitvector(SSZType):
    def __init__(self, n: int):
        self.n = n
        self.fixed_size = (n + 7) // 8

    def serialize(self, v):
        assert len(v) == self.n
        out = bytearray(self.fixed_size)
        for i, b in enumerate(v):
            if b:
                out[i // 8] |= 1 << (i % 8)
        return bytes(out)

    def htr(self, v):
        return merkleize_bytes(self.serialize(v), (self.n + 255) // 256)

    def default(self):
        return [False] * self.n


class Bitlist(SSZType):
    def __init__(self, limit: int):
        self.limit = limit

    def serialize(self, v):
        """bits little-endian within bytes, then one delimiter bit"""
        assert len(v) <= self.limit
        out = bytearray(len(v) // 8 + 1)
        for i, b in enumerate(v):
            if b:
                out[i // 8] |= 1 << (i % 8)
        out[len(v) // 8] |= 1 << (len(v) % 8)
        return bytes(out)

    def htr(self, v):
        out = bytearray((len(v) + 7) // 8)
        for i, b in enumerate(v):
            if b:
                out[i // 8] |= 1 << (i % 8)
        return mix_in_length(merkleize_bytes(bytes(out), (self.limit + 255) // 256), len(v))

    def default(self):
        return []


class Container(SSZType):
    def __init__(self, name: str, fields: Sequence[tuple]):
        self.name = name
        self.fields = list(fields)
        if all(t.fixed_size is not None for _, t in self.fields):
            self.fixed_size = sum(t.fixed_size for _, t in self.fields)

    def serialize(self, v):
        """SSZ container encoding: fixed parts (4-byte offsets for variable-size fields), then the
        variable parts in field order."""
        fixed_len = sum(t.fixed_size if t.fixed_size is not None else 4 for _, t in self.fields)
        fixed, var = [], []
        off = fixed_len
        for n, t in self.fields:
            if t.fixed_size is not None:
                fixed.append(t.serialize(v[n]))
            else:
                body = t.serialize(v[n])
                fixed.append(off.to_bytes(4, "little"))
                var.append(body)
                off += len(body)
        return b"".join(fixed) + b"".join(var)

    def htr(self, v):
        return merkleize_chunks([t.htr(v[n]) for n, t in self.fields], len(self.fields))

    def field_roots(self, v):
        return [t.htr(v[n]) for n, t in self.fields]

    def default(self):
        return {n: t.default() for n, t in self.fields}

    def generalized_index(self, path: Sequence) -> int:
        """ssz_rs `generalized_index` for the paths exercised at deneb/beacon_block.rs:139-154."""
        g, typ = 1, self
        for p in path:
            if isinstance(typ, Container):
                names = [n for n, _ in typ.fields]
                i = names.index(p)
                width = 1 << _depth_for(len(names))
                g = g * width + i
                typ = typ.fields[i][1]
            elif isinstance(typ, SSZList):
                g = g * 2  # data subtree (length is the right child)
                chunks = typ.limit if not isinstance(typ.elem, (UInt, Boolean)) else (typ.limit * typ.elem.fixed_size + 31) // 32
                width = 1 << _depth_for(chunks)
                g = g * width + int(p)
                typ = typ.elem
            elif isinstance(typ, Vector):
                width = 1 << _depth_for(typ.n)
                g = g * width + int(p)
                typ = typ.elem
            else:
                raise TypeError(typ)
        return g


# --------------------------------------------------------------------------------------
# the reference's type trees on the hot path
# --------------------------------------------------------------------------------------
BlsPublicKey = ByteVector(48)
BlsSignature = ByteVector(96)
Version = ByteVector(4)
ExecutionAddress = ByteVector(20)

SigningData = Container("SigningData", [("object_root", Root), ("domain", Bytes32)])
ForkData = Container("ForkData", [("current_version", Version), ("genesis_validators_root", Root)])
Fork = Container("Fork", [("previous_version", Version), ("current_version", Version), ("epoch", uint64)])
Checkpoint = Container("Checkpoint", [("epoch", uint64), ("root", Root)])
Eth1Data = Container("Eth1Data", [("deposit_root", Root), ("deposit_count", uint64), ("block_hash", Bytes32)])
BeaconBlockHeader = Container(
    "BeaconBlockHeader",
    [("slot", uint64), ("proposer_index", uint64), ("parent_root", Root), ("state_root", Root), ("body_root", Root)],
)
AttestationData = Container(
    "AttestationData",
    [("slot", uint64), ("index", uint64), ("beacon_block_root", Root), ("source", Checkpoint), ("target", Checkpoint)],
)
Validator = Container(
    "Validator",
    [
        ("public_key", BlsPublicKey),
        ("withdrawal_credentials", Bytes32),
        ("effective_balance", uint64),
        ("slashed", boolean),
        ("activation_eligibility_epoch", uint64),
        ("activation_epoch", uint64),
        ("exit_epoch", uint64),
        ("withdrawable_epoch", uint64),
    ],
)
assert Validator.fixed_size == 121 and BeaconBlockHeader.fixed_size == 112
DepositMessage = Container("DepositMessage", [("public_key", BlsPublicKey), ("withdrawal_credentials", Bytes32), ("amount", uint64)])
DepositData = Container(
    "DepositData",
    [("public_key", BlsPublicKey), ("withdrawal_credentials", Bytes32), ("amount", uint64), ("signature", BlsSignature)],
)
HistoricalSummary = Container("HistoricalSummary", [("block_summary_root", Root), ("state_summary_root", Root)])


def SyncCommittee(size: int) -> Container:
    return Container("SyncCommittee", [("public_keys", Vector(BlsPublicKey, size)), ("aggregate_public_key", BlsPublicKey)])


def ExecutionPayloadHeaderDeneb(bytes_per_logs_bloom: int = 256, max_extra_data_bytes: int = 32) -> Container:
    return Container(
        "ExecutionPayloadHeader",
        [
            ("parent_hash", Bytes32),
            ("fee_recipient", ExecutionAddress),
            ("state_root", Bytes32),
            ("receipts_root", Bytes32),
            ("logs_bloom", ByteVector(bytes_per_logs_bloom)),
            ("prev_randao", Bytes32),
            ("block_number", uint64),
            ("gas_limit", uint64),
            ("gas_used", uint64),
            ("timestamp", uint64),
            ("extra_data", ByteList(max_extra_data_bytes)),
            ("base_fee_per_gas", uint256),
            ("block_hash", Bytes32),
            ("transactions_root", Root),
            ("withdrawals_root", Root),
            ("blob_gas_used", uint64),
            ("excess_blob_gas", uint64),
        ],
    )


class Preset:
    def __init__(self, name, slots_per_historical_root, historical_roots_limit, eth1_data_votes_bound,
                 validator_registry_limit, epochs_per_historical_vector, epochs_per_slashings_vector,
                 sync_committee_size):
        self.name = name
        self.SLOTS_PER_HISTORICAL_ROOT = slots_per_historical_root
        self.HISTORICAL_ROOTS_LIMIT = historical_roots_limit
        self.ETH1_DATA_VOTES_BOUND = eth1_data_votes_bound
        self.VALIDATOR_REGISTRY_LIMIT = validator_registry_limit
        self.EPOCHS_PER_HISTORICAL_VECTOR = epochs_per_historical_vector
        self.EPOCHS_PER_SLASHINGS_VECTOR = epochs_per_slashings_vector
        self.SYNC_COMMITTEE_SIZE = sync_committee_size


# phase0/presets/mainnet.rs:7-36,82-84 ; altair/presets/mainnet.rs:19 ; minimal.rs equivalents
MAINNET = Preset("mainnet", 8192, 1 << 24, 2048, 1 << 40, 65536, 8192, 512)
MINIMAL = Preset("minimal", 64, 1 << 24, 32, 1 << 40, 64, 64, 32)


def BeaconStateDeneb(p: Preset) -> Container:
    """deneb/beacon_state.rs:25-63, 28 fields."""
    return Container(
        "BeaconState",
        [
            ("genesis_time", uint64),
            ("genesis_validators_root", Root),
            ("slot", uint64),
            ("fork", Fork),
            ("latest_block_header", BeaconBlockHeader),
            ("block_roots", Vector(Root, p.SLOTS_PER_HISTORICAL_ROOT)),
            ("state_roots", Vector(Root, p.SLOTS_PER_HISTORICAL_ROOT)),
            ("historical_roots", SSZList(Root, p.HISTORICAL_ROOTS_LIMIT)),
            ("eth1_data", Eth1Data),
            ("eth1_data_votes", SSZList(Eth1Data, p.ETH1_DATA_VOTES_BOUND)),
            ("eth1_deposit_index", uint64),
            ("validators", SSZList(Validator, p.VALIDATOR_REGISTRY_LIMIT)),
            ("balances", SSZList(uint64, p.VALIDATOR_REGISTRY_LIMIT)),
            ("randao_mixes", Vector(Bytes32, p.EPOCHS_PER_HISTORICAL_VECTOR)),
            ("slashings", Vector(uint64, p.EPOCHS_PER_SLASHINGS_VECTOR)),
            ("previous_epoch_participation", SSZList(uint8, p.VALIDATOR_REGISTRY_LIMIT)),
            ("current_epoch_participation", SSZList(uint8, p.VALIDATOR_REGISTRY_LIMIT)),
            ("justification_bits", Bitvector(4)),
            ("previous_justified_checkpoint", Checkpoint),
            ("current_justified_checkpoint", Checkpoint),
            ("finalized_checkpoint", Checkpoint),
            ("inactivity_scores", SSZList(uint64, p.VALIDATOR_REGISTRY_LIMIT)),
            ("current_sync_committee", SyncCommittee(p.SYNC_COMMITTEE_SIZE)),
            ("next_sync_committee", SyncCommittee(p.SYNC_COMMITTEE_SIZE)),
            ("latest_execution_payload_header", ExecutionPayloadHeaderDeneb()),
            ("next_withdrawal_index", uint64),
            ("next_withdrawal_validator_index", uint64),
            ("historical_summaries", SSZList(HistoricalSummary, p.HISTORICAL_ROOTS_LIMIT)),
        ],
    )


# ---- proofs: an independent restatement of ssz_rs `Prove::prove` / `generalized_index` --------------
LENGTH = "__len__"


def _data_chunks(typ: SSZType, v):
    """(chunks of the data tree of `typ` for value `v`, leaf limit, mixes a length in?, length, child types or None)"""
    if isinstance(typ, Container):
        return [t.htr(v[n]) for n, t in typ.fields], len(typ.fields), False, 0, [t for _, t in typ.fields]
    if isinstance(typ, (Vector, SSZList)):
        is_list = isinstance(typ, SSZList)
        bound = typ.limit if is_list else typ.n
        if isinstance(typ.elem, (UInt, Boolean)):
            data = typ.serialize(v)
            data += bytes(-len(data) % 32)
            return [data[i:i + 32] for i in range(0, len(data), 32)], (bound * typ.elem.fixed_size + 31) // 32, is_list, len(v), None
        return [typ.elem.htr(x) for x in v], bound, is_list, len(v), [typ.elem] * len(v)
    if isinstance(typ, (ByteVector, ByteList)):
        data = bytes(v) + bytes(-len(v) % 32)
        bound = typ.limit if isinstance(typ, ByteList) else typ.n
        return [data[i:i + 32] for i in range(0, len(data), 32)], (bound + 31) // 32, isinstance(typ, ByteList), len(v), None
    if isinstance(typ, (Bitvector, Bitlist)):
        out = bytearray((len(v) + 7) // 8)
        for i, b in enumerate(v):
            if b:
                out[i // 8] |= 1 << (i % 8)
        data = bytes(out) + bytes(-len(out) % 32)
        bound = typ.limit if isinstance(typ, Bitlist) else typ.n
        return [data[i:i + 32] for i in range(0, len(data), 32)], (bound + 255) // 256, isinstance(typ, Bitlist), len(v), None
    raise TypeError("a basic value has no children")


def _branch_in_chunks(chunks, limit: int, index: int):
    """siblings of leaf `index` in merkleize_chunks(chunks, limit), bottom-up, from the full level arrays"""
    depth = _depth_for(max(limit, 1))
    level = list(chunks)
    zero = bytes(32)
    out = []
    for d in range(depth):
        sib = (index >> d) ^ 1
        out.append(level[sib] if sib < len(level) else zero)
        nxt = [hash64(level[i], level[i + 1] if i + 1 < len(level) else zero) for i in range(0, len(level), 2)]
        level, zero = nxt, hash64(zero, zero)
    return out


def _child_position(typ: SSZType, p):
    """position of path element `p` in the data tree of `typ`, and the child's value accessor"""
    if isinstance(typ, Container):
        i = [n for n, _ in typ.fields].index(p) if isinstance(p, str) else int(p)
        return i
    if isinstance(typ, (Vector, SSZList)) and isinstance(typ.elem, (UInt, Boolean)):
        return int(p) * typ.elem.fixed_size // 32
    if isinstance(typ, (ByteVector, ByteList)):
        return int(p) // 32
    if isinstance(typ, (Bitvector, Bitlist)):
        return int(p) // 256
    return int(p)


def generalized_index(typ: SSZType, path) -> int:
    g = 1
    for p in path:
        _, limit, mix, _, _ = _data_chunks(typ, typ.default())
        if p == LENGTH:
            assert mix
            return g * 2 + 1
        if mix:
            g *= 2
        pos = _child_position(typ, p)
        g = (g << _depth_for(max(limit, 1))) + pos
        if isinstance(typ, Container):
            typ = typ.fields[pos][1]
        elif isinstance(typ, (Vector, SSZList)) and not isinstance(typ.elem, (UInt, Boolean)):
            typ = typ.elem
        else:
            typ = None
    return g


def prove(typ: SSZType, v, path):
    """-> (leaf, branch bottom-up, generalized index, witness root)"""
    root = typ.htr(v)
    parts = []
    g = 1
    leaf = None
    for p in path:
        assert leaf is None, "path continues below a leaf"
        chunks, limit, mix, length, child_types = _data_chunks(typ, v)
        depth = _depth_for(max(limit, 1))
        len_chunk = int(length).to_bytes(32, "little")
        if p == LENGTH:
            assert mix
            parts.append([merkleize_chunks(chunks, limit)])
            g = g * 2 + 1
            leaf = len_chunk
            continue
        pos = _child_position(typ, p)
        part = _branch_in_chunks(chunks, limit, pos)
        if mix:
            part.append(len_chunk)
            g *= 2
        g = (g << depth) + pos
        parts.append(part)
        if child_types is None:
            leaf = chunks[pos] if pos < len(chunks) else bytes(32)
        else:
            v = v[typ.fields[pos][0]] if isinstance(typ, Container) else v[pos]
            typ = child_types[pos]
    if leaf is None:
        leaf = typ.htr(v)
    branch = [n for part in reversed(parts) for n in part]
    return leaf, branch, g, root


# ---- the other forks' states (row a14), restated from the reference field for field ----------------------------------
def ExecutionPayloadHeader(fork: str) -> Container:
    """bellatrix/execution_payload.rs:58-81 (14 fields), capella/execution_payload.rs (+ withdrawals_root), deneb/
    execution_payload.rs:48-76 (+ blob_gas_used, excess_blob_gas)"""
    full = ExecutionPayloadHeaderDeneb()
    if fork == "electra":  # electra/execution_payload.rs:54-84: + deposit_receipts_root, withdrawal_requests_root
        return Container("ExecutionPayloadHeader", full.fields + [("deposit_receipts_root", Root), ("withdrawal_requests_root", Root)])
    n = {"bellatrix": 14, "capella": 15, "deneb": 17}[fork]
    return Container("ExecutionPayloadHeader", full.fields[:n])


# electra/beacon_state.rs:27-58 and the limits of electra/presets/{mainnet,minimal}.rs:10-12
PendingBalanceDeposit = Container("PendingBalanceDeposit", [("index", uint64), ("amount", uint64)])
PendingPartialWithdrawal = Container("PendingPartialWithdrawal", [("index", uint64), ("amount", uint64), ("withdrawable_epoch", uint64)])
PendingConsolidation = Container("PendingConsolidation", [("source_index", uint64), ("target_index", uint64)])
ELECTRA_LIMITS = {"mainnet": (1 << 27, 1 << 27, 1 << 18), "minimal": (1 << 27, 1 << 6, 1 << 6)}


def PendingAttestation(max_validators_per_committee: int = 2048) -> Container:
    """phase0/operations.rs:45-52"""
    return Container("PendingAttestation", [("aggregation_bits", Bitlist(max_validators_per_committee)), ("data", AttestationData),
                                            ("inclusion_delay", uint64), ("proposer_index", uint64)])


PENDING_ATTESTATIONS_BOUND = {"mainnet": 128 * 32, "minimal": 128 * 8}  # MAX_ATTESTATIONS * SLOTS_PER_EPOCH, phase0/presets/*.rs:84


def BeaconState(fork: str, p: Preset) -> Container:
    """phase0/beacon_state.rs:50-88, altair/beacon_state.rs:13-55, bellatrix/beacon_state.rs:13-58,
    capella/beacon_state.rs:13-64, deneb/beacon_state.rs:13-64: the common 15 fields, then the fork's own tail"""
    d = BeaconStateDeneb(p).fields
    common, bits_cps = d[:15], d[17:21]
    if fork == "phase0":
        att = SSZList(PendingAttestation(), PENDING_ATTESTATIONS_BOUND[p.name])
        return Container("BeaconState", common + [("previous_epoch_attestations", att), ("current_epoch_attestations", att)] + bits_cps)
    fields = d[:24]
    if fork in ("bellatrix", "capella", "deneb", "electra"):
        fields = fields + [("latest_execution_payload_header", ExecutionPayloadHeader(fork))]
    if fork in ("capella", "deneb", "electra"):
        fields = fields + d[25:28]
    if fork == "electra":  # electra/beacon_state.rs:73-145: six uint64, three lists of pending operations
        lim = ELECTRA_LIMITS[p.name]
        fields = fields + [(n, uint64) for n in ("deposit_receipts_start_index", "deposit_balance_to_consume", "exit_balance_to_consume",
                                                  "earliest_exit_epoch", "consolidation_balance_to_consume", "earliest_consolidation_epoch")]
        fields = fields + [("pending_balance_deposits", SSZList(PendingBalanceDeposit, lim[0])),
                           ("pending_partial_withdrawals", SSZList(PendingPartialWithdrawal, lim[1])),
                           ("pending_consolidations", SSZList(PendingConsolidation, lim[2]))]
    assert len(fields) == {"altair": 24, "bellatrix": 25, "capella": 28, "deneb": 28, "electra": 37}[fork]
    return Container("BeaconState", fields)


# ---- deneb block types (row a15), restated from the reference field for field ------------------------------------
SignedBeaconBlockHeader = Container("SignedBeaconBlockHeader", [("message", BeaconBlockHeader), ("signature", BlsSignature)])  # phase0/beacon_block.rs:93-100
ProposerSlashing = Container("ProposerSlashing", [("signed_header_1", SignedBeaconBlockHeader), ("signed_header_2", SignedBeaconBlockHeader)])  # phase0/operations.rs:97-100
Deposit = Container("Deposit", [("proof", Vector(Root, 33)), ("data", DepositData)])  # phase0/operations.rs:110-122, DEPOSIT_CONTRACT_TREE_DEPTH + 1
VoluntaryExit = Container("VoluntaryExit", [("epoch", uint64), ("validator_index", uint64)])  # :127-132
SignedVoluntaryExit = Container("SignedVoluntaryExit", [("message", VoluntaryExit), ("signature", BlsSignature)])  # :137-140
Withdrawal = Container("Withdrawal", [("index", uint64), ("validator_index", uint64), ("address", ExecutionAddress), ("amount", uint64)])  # capella/withdrawal.rs:9-17
BlsToExecutionChange = Container("BlsToExecutionChange", [("validator_index", uint64), ("from_bls_public_key", BlsPublicKey),
                                                          ("to_execution_address", ExecutionAddress)])  # capella/bls_to_execution_change.rs:9-15
SignedBlsToExecutionChange = Container("SignedBlsToExecutionChange", [("message", BlsToExecutionChange), ("signature", BlsSignature)])  # :20-23


class BlockPreset:
    """phase0/presets/*.rs:7,32-36; altair:19; bellatrix:21-24; capella:18-19; deneb:20"""

    def __init__(self, sync_committee_size, max_withdrawals, max_blob_commitments):
        self.MAX_PROPOSER_SLASHINGS, self.MAX_VALIDATORS_PER_COMMITTEE, self.MAX_ATTESTER_SLASHINGS = 16, 2048, 2
        self.MAX_ATTESTATIONS, self.MAX_DEPOSITS, self.MAX_VOLUNTARY_EXITS = 128, 16, 16
        self.SYNC_COMMITTEE_SIZE = sync_committee_size
        self.BYTES_PER_LOGS_BLOOM, self.MAX_EXTRA_DATA_BYTES = 256, 32
        self.MAX_BYTES_PER_TRANSACTION, self.MAX_TRANSACTIONS_PER_PAYLOAD = 1 << 30, 1 << 20
        self.MAX_WITHDRAWALS_PER_PAYLOAD, self.MAX_BLS_TO_EXECUTION_CHANGES = max_withdrawals, 16
        self.MAX_BLOB_COMMITMENTS_PER_BLOCK = max_blob_commitments


BLOCK_MAINNET, BLOCK_MINIMAL = BlockPreset(512, 16, 4096), BlockPreset(32, 4, 16)


def BeaconBlockDeneb(p: BlockPreset) -> Container:
    """deneb/beacon_block.rs:12-91 with phase0/operations.rs:35-61,105-108, altair/sync.rs:9-12, deneb/execution_payload.rs:13-46"""
    indexed = Container("IndexedAttestation", [("attesting_indices", SSZList(uint64, p.MAX_VALIDATORS_PER_COMMITTEE)),
                                               ("data", AttestationData), ("signature", BlsSignature)])
    attestation = Container("Attestation", [("aggregation_bits", Bitlist(p.MAX_VALIDATORS_PER_COMMITTEE)), ("data", AttestationData),
                                            ("signature", BlsSignature)])
    attester_slashing = Container("AttesterSlashing", [("attestation_1", indexed), ("attestation_2", indexed)])
    sync_aggregate = Container("SyncAggregate", [("sync_committee_bits", Bitvector(p.SYNC_COMMITTEE_SIZE)),
                                                 ("sync_committee_signature", BlsSignature)])
    payload = Container("ExecutionPayload", [
        ("parent_hash", Bytes32), ("fee_recipient", ExecutionAddress), ("state_root", Bytes32), ("receipts_root", Bytes32),
        ("logs_bloom", ByteVector(p.BYTES_PER_LOGS_BLOOM)), ("prev_randao", Bytes32), ("block_number", uint64), ("gas_limit", uint64),
        ("gas_used", uint64), ("timestamp", uint64), ("extra_data", ByteList(p.MAX_EXTRA_DATA_BYTES)), ("base_fee_per_gas", uint256),
        ("block_hash", Bytes32), ("transactions", SSZList(ByteList(p.MAX_BYTES_PER_TRANSACTION), p.MAX_TRANSACTIONS_PER_PAYLOAD)),
        ("withdrawals", SSZList(Withdrawal, p.MAX_WITHDRAWALS_PER_PAYLOAD)), ("blob_gas_used", uint64), ("excess_blob_gas", uint64)])
    body = Container("BeaconBlockBody", [
        ("randao_reveal", BlsSignature), ("eth1_data", Eth1Data), ("graffiti", Bytes32),
        ("proposer_slashings", SSZList(ProposerSlashing, p.MAX_PROPOSER_SLASHINGS)),
        ("attester_slashings", SSZList(attester_slashing, p.MAX_ATTESTER_SLASHINGS)),
        ("attestations", SSZList(attestation, p.MAX_ATTESTATIONS)), ("deposits", SSZList(Deposit, p.MAX_DEPOSITS)),
        ("voluntary_exits", SSZList(SignedVoluntaryExit, p.MAX_VOLUNTARY_EXITS)), ("sync_aggregate", sync_aggregate),
        ("execution_payload", payload), ("bls_to_execution_changes", SSZList(SignedBlsToExecutionChange, p.MAX_BLS_TO_EXECUTION_CHANGES)),
        ("blob_kzg_commitments", SSZList(ByteVector(48), p.MAX_BLOB_COMMITMENTS_PER_BLOCK))])
    return Container("BeaconBlock", [("slot", uint64), ("proposer_index", uint64), ("parent_root", Root), ("state_root", Root),
                                     ("body", body)])


# ---- electra block types: electra/beacon_block.rs:17-63, electra/operations.rs:10-50, electra/execution_payload.rs:13-45,
# electra/beacon_state.rs:16-25 (DepositReceipt), :62-68 (ExecutionLayerWithdrawalRequest); limits electra/presets/*.rs:13-17.
# MAX_VALIDATORS_PER_SLOT: a const parameter the reference never binds; the specification's value is
# MAX_VALIDATORS_PER_COMMITTEE * MAX_COMMITTEES_PER_SLOT (phase0/presets/{mainnet,minimal}.rs:5 -> 64, 4).
class BlockPresetElectra(BlockPreset):
    def __init__(self, base: BlockPreset, committees_per_slot, max_deposit_receipts, max_withdrawal_requests):
        self.__dict__.update(base.__dict__)
        self.MAX_COMMITTEES_PER_SLOT = committees_per_slot
        self.MAX_VALIDATORS_PER_SLOT = self.MAX_VALIDATORS_PER_COMMITTEE * committees_per_slot
        self.MAX_ATTESTER_SLASHINGS_ELECTRA, self.MAX_ATTESTATIONS_ELECTRA, self.MAX_CONSOLIDATIONS = 1, 8, 1
        self.MAX_DEPOSIT_RECEIPTS_PER_PAYLOAD, self.MAX_WITHDRAWAL_REQUESTS_PER_PAYLOAD = max_deposit_receipts, max_withdrawal_requests


BLOCK_ELECTRA_MAINNET = BlockPresetElectra(BLOCK_MAINNET, 64, 8192, 16)
BLOCK_ELECTRA_MINIMAL = BlockPresetElectra(BLOCK_MINIMAL, 4, 4, 2)
DepositReceipt = Container("DepositReceipt", [("public_key", BlsPublicKey), ("withdrawal_credentials", Bytes32), ("amount", uint64),
                                              ("signature", BlsSignature), ("index", uint64)])
ExecutionLayerWithdrawalRequest = Container("ExecutionLayerWithdrawalRequest", [("source_address", ExecutionAddress),
                                                                                ("validator_public_key", BlsPublicKey), ("amount", uint64)])
Consolidation = Container("Consolidation", [("source_index", uint64), ("target_index", uint64), ("epoch", uint64)])
SignedConsolidation = Container("SignedConsolidation", [("message", Consolidation), ("signature", BlsSignature)])


def BeaconBlockElectra(p: BlockPresetElectra) -> Container:
    indexed = Container("IndexedAttestation", [("attesting_indices", SSZList(uint64, p.MAX_VALIDATORS_PER_SLOT)),
                                               ("data", AttestationData), ("signature", BlsSignature)])
    attestation = Container("Attestation", [("aggregation_bits", Bitlist(p.MAX_VALIDATORS_PER_SLOT)), ("data", AttestationData),
                                            ("committee_bits", Bitvector(p.MAX_COMMITTEES_PER_SLOT)), ("signature", BlsSignature)])
    attester_slashing = Container("AttesterSlashing", [("attestation_1", indexed), ("attestation_2", indexed)])
    sync_aggregate = Container("SyncAggregate", [("sync_committee_bits", Bitvector(p.SYNC_COMMITTEE_SIZE)),
                                                 ("sync_committee_signature", BlsSignature)])
    payload = Container("ExecutionPayload", [
        ("parent_hash", Bytes32), ("fee_recipient", ExecutionAddress), ("state_root", Bytes32), ("receipts_root", Bytes32),
        ("logs_bloom", ByteVector(p.BYTES_PER_LOGS_BLOOM)), ("prev_randao", Bytes32), ("block_number", uint64), ("gas_limit", uint64),
        ("gas_used", uint64), ("timestamp", uint64), ("extra_data", ByteList(p.MAX_EXTRA_DATA_BYTES)), ("base_fee_per_gas", uint256),
        ("block_hash", Bytes32), ("transactions", SSZList(ByteList(p.MAX_BYTES_PER_TRANSACTION), p.MAX_TRANSACTIONS_PER_PAYLOAD)),
        ("withdrawals", SSZList(Withdrawal, p.MAX_WITHDRAWALS_PER_PAYLOAD)), ("blob_gas_used", uint64), ("excess_blob_gas", uint64),
        ("deposit_receipts", SSZList(DepositReceipt, p.MAX_DEPOSIT_RECEIPTS_PER_PAYLOAD)),
        ("withdrawal_requests", SSZList(ExecutionLayerWithdrawalRequest, p.MAX_WITHDRAWAL_REQUESTS_PER_PAYLOAD))])
    body = Container("BeaconBlockBody", [
        ("randao_reveal", BlsSignature), ("eth1_data", Eth1Data), ("graffiti", Bytes32),
        ("proposer_slashings", SSZList(ProposerSlashing, p.MAX_PROPOSER_SLASHINGS)),
        ("attester_slashings", SSZList(attester_slashing, p.MAX_ATTESTER_SLASHINGS_ELECTRA)),
        ("attestations", SSZList(attestation, p.MAX_ATTESTATIONS_ELECTRA)), ("deposits", SSZList(Deposit, p.MAX_DEPOSITS)),
        ("voluntary_exits", SSZList(SignedVoluntaryExit, p.MAX_VOLUNTARY_EXITS)), ("sync_aggregate", sync_aggregate),
        ("execution_payload", payload), ("bls_to_execution_changes", SSZList(SignedBlsToExecutionChange, p.MAX_BLS_TO_EXECUTION_CHANGES)),
        ("blob_kzg_commitments", SSZList(ByteVector(48), p.MAX_BLOB_COMMITMENTS_PER_BLOCK)),
        ("consolidations", SSZList(SignedConsolidation, p.MAX_CONSOLIDATIONS))])
    return Container("BeaconBlock", [("slot", uint64), ("proposer_index", uint64), ("parent_root", Root), ("state_root", Root),
                                     ("body", body)])


def compute_signing_root(obj_type: SSZType, obj, domain: bytes) -> bytes:
    """signing.rs:14-22."""
    return SigningData.htr({"object_root": obj_type.htr(obj), "domain": domain})
